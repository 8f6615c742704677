#!/usr/bin/env python3
"""Per-kernel HIP-event times of ONE proof (B = 1) of the AES-GCM 1 KiB circuit: where single-proof latency goes."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
L = pkg.lib()
b = pkg.CircuitBuilder()
t = pkg.AesGcmTarget.build(b, 4, 10, 1024, False)
data = b.build()
key, nonce, pt = bytes([42] * 16), bytes([111] * 12), bytes([42] * 1024)
ct, tag = pkg.native.gcm_encrypt(key, nonce, pt)
pw = pkg.PartialWitness()
t.set_targets(pw, key, nonce, pt, ct, tag)
data.prove(pw)
t0 = time.time()
for _ in range(5):
    data.prove(pw)
print("prove(pw): %.2f ms per call" % ((time.time() - t0) / 5 * 1e3))
h = data.gpu()
L.p2_circuit_set_timing(h, 1)
data.prove(pw)
L.p2_circuit_synchronize(h)  # collects the event pairs
arr = (pkg.api._KernelTime * 64)()
k = L.p2_circuit_get_timing(h, arr, 64)
rows = sorted(((arr[i].ms, arr[i].count, arr[i].name.decode()) for i in range(min(k, 64))), reverse=True)
print("sum of kernel times %.2f ms over %d launches" % (sum(r[0] for r in rows), sum(r[1] for r in rows)))
for ms, cnt, name in rows[:16]:
    print("  %-18s %7.3f ms  %4d launches" % (name, ms, cnt))
