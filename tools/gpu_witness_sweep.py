#!/usr/bin/env python3
"""HIP-event time of the witness kernel for several macro sizes (P2AES_WITNESS_FUSE, witness_schedule.h) on one circuit:
    python tools/gpu_witness_sweep.py [plaintext bytes = 65536] [batch = 8] [K ...]
The circuit is built once; every K loads its own handle (the schedule is made at load)."""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
import circuits  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
Ks = [int(a) for a in sys.argv[3:]] or [1, 4, 8, 16, 32]
r = random.Random(1)
keys = [(bytes(r.randrange(256) for _ in range(16)), bytes(r.randrange(256) for _ in range(12)), bytes(r.randrange(256) for _ in range(L))) for _ in range(B)]
t0 = time.time()
data, pws, _ = circuits.encrypt(pkg, 4, L, False, keys=keys)
print("circuit built in %.1f s: n = 2^%d, %d ops, %d builder levels" % (time.time() - t0, data.info["degree_bits"], data.info["num_ops"], data.info["num_levels"]), flush=True)
ref = None
for K in Ks:
    os.environ["P2AES_WITNESS_FUSE"] = str(K)
    d = pkg.CircuitData(data.blob)
    t0 = time.time()
    h = d.gpu()
    load = time.time() - t0
    proofs, st = d.prove_batch(pws)   # warm-up, allocates the workspace
    assert st == [0] * B, st
    if ref is None:
        ref = proofs
    assert proofs == ref, "proofs changed with the witness schedule"
    pkg.lib().p2_circuit_set_timing(h, 1)
    d.prove_batch(pws)
    pkg.lib().p2_circuit_synchronize(h)
    arr = (pkg.api._KernelTime * 64)()
    k = pkg.lib().p2_circuit_get_timing(h, arr, 64)
    times = {arr[i].name.decode(): arr[i].ms for i in range(min(k, 64))}
    print("K = %3d: witness %8.3f ms per %d-proof call (load %.1f s), sum of kernels %.1f ms" % (K, times.get("witness", -1), B, load, sum(times.values())), flush=True)
    del d
