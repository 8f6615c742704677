#!/usr/bin/env python3
"""iNTT and coset LDE of one random column at 2^20, 2^21, 2^22 points on the GPU (two-pass transform: every shape of the
register-blocked first pass that the pytest sizes 2^15..2^19 do not reach) against the CPU oracle, through numpy buffers."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
import oracle_lib as O  # noqa: E402

P = 0xFFFFFFFF00000001
L, OL = pkg.lib(), O.lib()
u64p = C.POINTER(C.c_uint64)
rng = np.random.default_rng(7)
bad = 0
for bits in [int(a) for a in sys.argv[1:]] or [20, 21, 22]:
    n = 1 << bits
    vals = (rng.integers(0, 1 << 63, size=n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=n, dtype=np.uint64)) % np.uint64(P)
    out = np.zeros(n, dtype=np.uint64)
    t0 = time.time()
    assert L.p2_gpu_intt(vals.ctypes.data_as(u64p), 1, bits, out.ctypes.data_as(u64p), 0) == 0, L.p2_last_error()
    lde = np.zeros(8 * n, dtype=np.uint64)
    assert L.p2_gpu_lde(vals.ctypes.data_as(u64p), 1, bits, 3, lde.ctypes.data_as(u64p), 0) == 0, L.p2_last_error()
    t1 = time.time()
    ref = vals.copy()
    OL.orc_fft(ref.ctypes.data_as(u64p), bits, 1)
    ref_lde = np.zeros(8 * n, dtype=np.uint64)
    OL.orc_lde(vals.ctypes.data_as(u64p), bits, 3, ref_lde.ctypes.data_as(u64p))
    ok_i, ok_l = bool((ref == out).all()), bool((ref_lde == lde).all())
    print("2^%d: intt %s, lde %s (gpu %.1fs, oracle %.1fs)" % (bits, "ok" if ok_i else "MISMATCH", "ok" if ok_l else "MISMATCH", t1 - t0, time.time() - t1), flush=True)
    if not ok_l:
        d = np.nonzero(ref_lde != lde)[0]
        print("   first lde mismatches at", d[:8], "of", len(d))
    if not ok_i:
        d = np.nonzero(ref != out)[0]
        print("   first intt mismatches at", d[:8], "of", len(d))
    bad += (not ok_i) + (not ok_l)
print("NTT BIG", "OK" if not bad else "FAILED")
sys.exit(1 if bad else 0)
