#!/usr/bin/env python3
"""Development aid: run the GPU pipeline on a few circuits and compare every intermediate buffer with the CPU
oracle's trace, in pipeline order, reporting the first divergence per circuit."""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
import ctypes as C  # noqa: E402

import oracle_lib as O  # noqa: E402

L = pkg.lib()
OL = O.lib()
rnd = random.Random(7)
P = 0xFFFFFFFF00000001
ok_all = True


def check(name, a, b):
    global ok_all
    if a == b:
        print("   ok   %-22s (%d words)" % (name, len(a)))
        return True
    ok_all = False
    k = next((i for i in range(min(len(a), len(b))) if a[i] != b[i]), min(len(a), len(b)))
    nbad = sum(1 for i in range(min(len(a), len(b))) if a[i] != b[i])
    print("   FAIL %-22s len %d vs %d, first diff at %d (%d differ): gpu %s oracle %s" % (
        name, len(a), len(b), k, nbad, a[k:k + 2], b[k:k + 2]))
    return False


def primitives():
    print("== primitives")
    n = 1000
    st = [rnd.randrange(P) for _ in range(12 * n)]
    buf = (C.c_uint64 * len(st))(*st)
    assert L.p2_gpu_poseidon(buf, n, 0) == 0, L.p2_last_error()
    ref = []
    for i in range(n):
        s = (C.c_uint64 * 12)(*st[12 * i:12 * i + 12])
        OL.orc_poseidon(s)
        ref += list(s)
    check("poseidon", list(buf), ref)
    for bits in (3, 5, 10, 13, 14, 15, 16):
        cols, nn = (3 if bits <= 14 else 2), 1 << bits
        vals = [rnd.randrange(P) for _ in range(cols * nn)]
        out = (C.c_uint64 * (cols * nn))()
        assert L.p2_gpu_intt((C.c_uint64 * len(vals))(*vals), cols, bits, out, 0) == 0, L.p2_last_error()
        ref = []
        for c in range(cols):
            a = (C.c_uint64 * nn)(*vals[c * nn:(c + 1) * nn])
            OL.orc_fft(a, bits, 1)
            ref += list(a)
        check("intt 2^%d" % bits, list(out), ref)
        lde = (C.c_uint64 * (cols * nn * 8))()
        assert L.p2_gpu_lde((C.c_uint64 * len(vals))(*vals), cols, bits, 3, lde, 0) == 0, L.p2_last_error()
        ref = []
        for c in range(cols):
            o = (C.c_uint64 * (8 * nn))()
            OL.orc_lde((C.c_uint64 * nn)(*vals[c * nn:(c + 1) * nn]), bits, 3, o)
            ref += list(o)
        check("lde 2^%d" % bits, list(lde), ref)
    for cols, leaves in ((3, 64), (11, 512), (135, 2048)):
        colmaj = [rnd.randrange(P) for _ in range(cols * leaves)]
        cap = (C.c_uint64 * 64)()
        assert L.p2_gpu_merkle_cap((C.c_uint64 * len(colmaj))(*colmaj), cols, leaves, 4, cap, 0) == 0, L.p2_last_error()
        rowmaj = [colmaj[c * leaves + i] for i in range(leaves) for c in range(cols)]
        ref = (C.c_uint64 * 64)()
        OL.orc_merkle_cap((C.c_uint64 * len(rowmaj))(*rowmaj), leaves, cols, 4, ref)
        check("merkle %dx%d" % (cols, leaves), list(cap), list(ref))


def pipeline(name, data, pw):
    print("== %s  (n = 2^%d, %d ops, %d levels)" % (name, data.info["degree_bits"], data.info["num_ops"], data.info["num_levels"]))
    oc = O.OracleCircuit(data.blob)
    t0 = time.time()
    st, ref = oc.prove(pw.map, trace=True)
    t1 = time.time()
    proofs, status = data.prove_batch([pw, pw])
    t2 = time.time()
    print("   oracle %.2fs, gpu (2 proofs incl. load) %.2fs, status %s / oracle %d" % (t1 - t0, t2 - t1, status, st))
    n, info = 1 << data.info["degree_bits"], data.info
    check("verifier_data", data.verifier_data(), oc.verifier_data())
    w = oc.trace("wires")
    check("wires", data.debug_read("wires", 1), w[:80 * n])
    check("wires_cap", data.debug_read("wires_cap", 1), oc.trace("wires_cap"))
    ch = data.debug_read("challenges", 1)
    check("betas/gammas", ch[0:4], oc.trace("betas") + oc.trace("gammas"))
    if oc.trace("deltas"):
        check("deltas", ch[4:12], oc.trace("deltas"))
    check("zs", data.debug_read("zs", 1), oc.trace("zs"))
    check("zs_cap", data.debug_read("zs_cap", 1), oc.trace("zs_cap"))
    check("alphas", ch[12:14], oc.trace("alphas"))
    check("quotient_coeffs", data.debug_read("quotient_coeffs", 1), oc.trace("quotient_coeffs"))
    check("quotient_cap", data.debug_read("quotient_cap", 1), oc.trace("quotient_cap"))
    check("zeta", ch[14:16], oc.trace("zeta"))
    check("fri_alpha", ch[16:18], oc.trace("fri_alpha"))
    fin = data.debug_read("fri_final_poly_in", 1)
    refin = oc.trace("fri_final_poly_in")
    check("fri_final_poly_in", fin, refin[0::2] + refin[1::2])
    nr = info["num_fri_rounds"]
    check("fri_betas", ch[18:18 + 2 * nr], oc.trace("fri_betas"))
    check("pow_witness", ch[34:35], oc.trace("pow_witness"))
    check("query_indices", ch[36:36 + 28], oc.trace("query_indices"))
    for i, p in enumerate(proofs):
        if p is None:
            print("   proof %d: status %d" % (i, status[i]))
            continue
        same = p == ref
        print("   proof %d bytes == oracle: %s" % (i, same))
        if not same:
            global ok_all
            ok_all = False
            k = next(j for j in range(len(ref)) if p[j] != ref[j])
            print("      first differing byte", k, "of", len(ref))
        try:
            data.verify(p)
            print("      verifies")
        except pkg.P2Error as e:
            print("      VERIFY FAILED:", e)


def main():
    primitives()
    b = pkg.CircuitBuilder()
    lut = b.sbox_lut()
    t = b.add_virtual_byte_target(lut)
    data = b.build()
    pw = pkg.PartialWitness()
    pw.set_target(t, 77)
    pipeline("assert_byte", data, pw)
    pw = pkg.PartialWitness()
    pw.set_target(t, 256)
    print("   out-of-range byte:", data.prove_batch([pw])[1], "(expect [1])")

    b = pkg.CircuitBuilder()
    lut = b.gf_2_8_mul_lut()
    x, y = b.add_virtual_byte_target_unsafe(), b.add_virtual_byte_target_unsafe()
    xy = b.gf_2_8_mul(lut, x, y)
    data = b.build()
    pw = pkg.PartialWitness()
    for t_, v in ((x, 0x57), (y, 0x13), (xy, 0xFE)):
        pw.set_byte_target(t_, v)
    pipeline("gf_2_8_mul", data, pw)

    if len(sys.argv) > 1:
        Lp = int(sys.argv[1])
        key, nonce, pt = bytes([42] * 16), bytes([111] * 12), bytes([42] * Lp)
        ct, tag = pkg.native.gcm_encrypt(key, nonce, pt)
        b = pkg.CircuitBuilder()
        t = pkg.AesGcmTarget.build(b, 4, 10, Lp, len(sys.argv) > 2)
        data = b.build()
        pw = pkg.PartialWitness()
        t.set_targets(pw, key, nonce, pt, ct, tag)
        pipeline("aes-gcm-128 L=%d" % Lp, data, pw)
    print("ALL OK" if ok_all else "SOME CHECKS FAILED")
    return 0 if ok_all else 1


if __name__ == "__main__":
    sys.exit(main())
