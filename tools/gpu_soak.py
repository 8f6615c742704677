#!/usr/bin/env python3
"""Soak check on the GPU box: the device self-test on 2^24 threads with several seeds (every mad-based field primitive
against the textbook forms, 1.07e9 inputs per seed), then several 256-proof batches of the AES-GCM 1 KiB circuit with every
proof checked by the host verifier and two of them against the oracle.  The inline-asm helpers rely on hand-placed wait
states; a marginal hazard would show up here as a sporadic mismatch."""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
import oracle_lib as O  # noqa: E402

L = pkg.lib()
for seed in (1, 0x5EED, 0xDEADBEEF, 987654321):
    t0 = time.time()
    bad = L.p2_selftest_device(seed, 1 << 24, 0)
    print("selftest seed %#x: %d mismatches (%.1fs)" % (seed, bad, time.time() - t0), flush=True)
    assert bad == 0
b = pkg.CircuitBuilder()
t = pkg.AesGcmTarget.build(b, 4, 10, 1024, False)
data = b.build()
rnd = random.Random(7)
oc = O.OracleCircuit(data.blob)
vd = data.verifier_data()
assert vd == oc.verifier_data()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for rd in range(rounds):
    pws = []
    for i in range(256):
        key, nonce, pt = bytes(rnd.randrange(256) for _ in range(16)), bytes(rnd.randrange(256) for _ in range(12)), bytes(rnd.randrange(256) for _ in range(1024))
        ct, tag = pkg.native.gcm_encrypt(key, nonce, pt)
        pw = pkg.PartialWitness()
        t.set_targets(pw, key, nonce, pt, ct, tag)
        pws.append(pw)
    t0 = time.time()
    proofs, status = data.prove_batch(pws)
    t1 = time.time()
    assert status == [0] * 256
    for p in proofs:
        data.verify(p, vd)
    t2 = time.time()
    for i in (rnd.randrange(256), rnd.randrange(256)):
        st, ref = oc.prove(pws[i].map)
        assert st == 0 and ref == proofs[i], "proof %d of round %d differs from the oracle" % (i, rd)
    print("round %d: 256 proofs in %.2fs, all verified in %.1fs, 2 byte-identical to the oracle (%.1fs)" % (rd, t1 - t0, t2 - t1, time.time() - t2), flush=True)
print("SOAK OK")
