#!/usr/bin/env python3
"""Wall-clock of p2_prove_batch (host buffers in, proofs out) for small batches of the AES-GCM 1 KiB circuit."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
pkg = g.load_package()
import circuits, random
r = random.Random(1)
L = 1024
keys = [(bytes(r.randrange(256) for _ in range(16)), bytes(r.randrange(256) for _ in range(12)), bytes(r.randrange(256) for _ in range(L))) for _ in range(256)]
data, pws, _ = circuits.encrypt(pkg, 4, L, False, keys=keys)
data.gpu()
for B in (1, 1, 2, 4, 8, 16, 32, 64, 256):
    d2 = pkg.CircuitData(data.blob); d2.gpu()
    d2.prove_batch(pws[:B])  # warm-up (allocates the workspace)
    t0 = time.perf_counter(); n = 3
    for _ in range(n): proofs, st = d2.prove_batch(pws[:B])
    dt = (time.perf_counter() - t0) / n
    assert st == [0] * B
    print("B=%2d  %.2f ms per call  %.2f ms per proof  (PCIe-inclusive: host assignments in, %d-byte proofs out)" % (B, 1e3 * dt, 1e3 * dt / B, len(proofs[0])), flush=True)
    del d2
