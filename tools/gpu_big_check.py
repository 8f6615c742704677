#!/usr/bin/env python3
"""AES-GCM-128 over a 64 KiB plaintext (BASELINE.json configs[4]: n = 2^19 rows): prove a small batch on the GPU and
verify every proof (the oracle needs minutes per proof at this size, so the check is the independent verifier plus
determinism)."""
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
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
TAG = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
r = random.Random(1)
keys = [(bytes(r.randrange(256) for _ in range(16)), bytes(r.randrange(256) for _ in range(12)), bytes(r.randrange(256) for _ in range(L))) for _ in range(B)]
t0 = time.time()
data, pws, _ = circuits.encrypt(pkg, 4, L, TAG, keys=keys)
print("built: n=2^%d ops=%d levels=%d blob=%dMB in %.1fs" % (data.info["degree_bits"], data.info["num_ops"], data.info["num_levels"], len(data.blob) >> 20, time.time() - t0), flush=True)
t0 = time.time()
data.gpu()
print("p2_circuit_load (upload + constants/sigmas commitment): %.2fs" % (time.time() - t0), flush=True)
t0 = time.time()
proofs, status = data.prove_batch(pws)
t1 = time.time()
print("prove_batch(%d): %.3fs  status %s" % (B, t1 - t0, status), flush=True)
t0 = time.time()
proofs2, status2 = data.prove_batch(pws)
t1 = time.time()
print("prove_batch(%d) again: %.3fs -> %.3f s/proof; deterministic: %s" % (B, t1 - t0, (t1 - t0) / B, proofs2 == proofs), flush=True)
vd = data.verifier_data()
t0 = time.time()
for p in proofs[:2]:
    data.verify(p, vd)
print("verified 2 proofs in %.1fs (%d bytes each)" % (time.time() - t0, len(proofs[0])), flush=True)
bad = pkg.PartialWitness()
bad.map = dict(pws[0].map)
k = list(bad.map)[16 + 12 + L + 1000]
bad.map[k] ^= 1
print("wrong ciphertext byte ->", data.prove_batch([bad])[1], "(expect [1])")
print("BIG OK")
