#!/usr/bin/env python3
"""Writes tests/golden/gate_counts.json: builder.num_gates() and the compiled shape for the size list of the reference's
`test_encrypt_report_sizes` (aes-gcm/src/circuit_gcm.rs:708-736: AES-GCM-128 and -256, L in 16..2048, TAG = false).

The reference prints these numbers but does not store them, and it cannot be run here (no cargo), so they are NOT a
pin against real plonky2: they are this build's own counts, frozen so that any drift of the builder (gate packing,
constant folding, lookup-row layout) shows up in the CPU test suite.  Regenerate only for an intended change."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
out = {"source": "tools/make_gate_counts.py (self-generated; see docstring)", "sizes": []}
for nk in (4, 8):
    for L in (16, 17, 32, 33, 64, 128, 256, 512, 1024, 2048):
        t0 = time.time()
        b = pkg.CircuitBuilder()
        pkg.AesGcmTarget.build(b, nk, nk + 6, L, False)
        gates = b.num_gates()
        info = b.build().info
        out["sizes"].append({"nk": nk, "L": L, "num_gates": gates, "degree_bits": info["degree_bits"], "num_ops": info["num_ops"],
                             "num_levels": info["num_levels"], "num_slots": info["num_slots"], "proof_bytes": info["proof_bytes"]})
        print(out["sizes"][-1], "%.1fs" % (time.time() - t0))
with open(os.path.join(ROOT, "tests", "golden", "gate_counts.json"), "w") as f:
    json.dump(out, f, indent=1)
