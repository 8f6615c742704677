// ORACLE -- TEST INFRASTRUCTURE ONLY.  Lane-parallel form of the oracle's sparse Poseidon (oracle_poseidon_sparse.h): W
// independent permutations, one per 64-bit SIMD lane, for the two places where the prover hashes many equal-shaped inputs
// at once -- the leaves of a Merkle tree and the pairs of one tree level.  Used only when bench.py's cpu_baseline leg asks for
// the fast hash (orc_set_fast_hash(1)) and the host has the instruction set; tests/test_oracle_kat.py holds it to the textbook
// permutation, and a whole proof has to come out byte-identical either way.
//
// This header is a template over a vector type V supplied by the including translation unit (oracle_simd_avx512.cpp,
// oracle_simd_avx2.cpp; each compiled with its own -m flags and entered only after a cpuid check):
//   V::W lanes;  V::set1, vadd, vsub, vand, vor, vsrl32, vsll32 (64-bit lanes);  vmul32 = low 32 x low 32 -> 64 per lane;
//   vadd_if_lt(r, a, b, x) = r + (a <u b ? x : 0);  vsub_if_lt likewise;  V::gather(base, stride) = lane l <- base[l * stride].
// Arithmetic is the scalar file's: words are ARBITRARY 64-bit representatives inside the permutation, 2^64 = 2^32 - 1 and
// 2^96 = -1 (mod p), every wrap of 2^64 is paid back at once, and the output is canonicalised.
#pragma once
#include "oracle_field.h"

namespace orc {

template <class V>
struct PoseidonLanes {
    static constexpr u64 E = 0xffffffffull;
    // a * b as (hi, lo) from four 32 x 32 products
    static inline void mul_wide(V a, V b, V& hi, V& lo) {
        const V m32 = V::set1(E);
        const V ah = vsrl32(a), bh = vsrl32(b);
        const V ll = vmul32(a, b), lh = vmul32(a, bh), hl = vmul32(ah, b), hh = vmul32(ah, bh);
        const V mid = vadd(lh, vsrl32(ll));              // <= (2^32-1)^2 + 2^32 - 1: no wrap
        const V mid2 = vadd(hl, vand(mid, m32));         // likewise
        lo = vor(vand(ll, m32), vsll32(mid2));
        hi = vadd(hh, vadd(vsrl32(mid), vsrl32(mid2)));  // the true high word: < 2^64
    }
    static inline void sqr_wide(V a, V& hi, V& lo) {
        const V m32 = V::set1(E);
        const V ah = vsrl32(a);
        const V ll = vmul32(a, a), lh = vmul32(a, ah), hh = vmul32(ah, ah);
        const V mid = vadd(lh, vsrl32(ll));
        const V mid2 = vadd(lh, vand(mid, m32));
        lo = vor(vand(ll, m32), vsll32(mid2));
        hi = vadd(hh, vadd(vsrl32(mid), vsrl32(mid2)));
    }
    // lo + 2^64 (hl + 2^32 hh) = lo - hh + hl (2^32 - 1): some representative below 2^64 (oracle_field.h fred, without its last line)
    static inline V red(V hi, V lo) {
        const V eps = V::set1(E);
        const V hh = vsrl32(hi), hl = vand(hi, eps);
        V t0 = vsub(lo, hh);
        t0 = vsub_if_lt(t0, lo, hh, eps);
        const V t1 = vsub(vsll32(hl), hl);
        V r = vadd(t0, t1);
        return vadd_if_lt(r, r, t1, eps);
    }
    static inline V mul(V a, V b) {
        V h, l;
        mul_wide(a, b, h, l);
        return red(h, l);
    }
    static inline V sqr(V a) {
        V h, l;
        sqr_wide(a, h, l);
        return red(h, l);
    }
    // a * b + c for arbitrary words: below 2^128, so the carry of the low word fits the high one
    static inline V mul_add(V a, V b, V c) {
        V h, l;
        mul_wide(a, b, h, l);
        l = vadd(l, c);
        h = vadd_if_lt(h, l, c, V::set1(1));
        return red(h, l);
    }
    static inline V pow7(V x) {
        const V x2_ = sqr(x), x4 = sqr(x2_), x3 = mul(x, x2_);
        return mul(x3, x4);
    }
    // a arbitrary, c CANONICAL (the same in every lane): a representative of a + c
    static inline V add_const(V a, u64 c) {
        const V cv = V::set1(c);
        const V r = vadd(a, cv);
        return vadd_if_lt(r, r, cv, V::set1(E));  // wrapped: r < c <= p - 1, so + (2^32 - 1) cannot wrap again
    }
    static inline V canon(V a) {
        const V p = V::set1(MODULUS);
        return vadd_if_lt(vsub(a, p), a, p, p);
    }
    // sum of 64 x 64 products kept exactly in (carries, high, low) until the end
    struct Wide {
        V lo, hi, clo, chi;
        inline Wide() : lo(V::set1(0)), hi(lo), clo(lo), chi(lo) {}
        inline void mac(V a, V b) {
            V h, l;
            mul_wide(a, b, h, l);
            const V one = V::set1(1);
            lo = vadd(lo, l);
            clo = vadd_if_lt(clo, lo, l, one);
            hi = vadd(hi, h);
            chi = vadd_if_lt(chi, hi, h, one);
        }
        // lo + 2^64 (hi + clo) + 2^128 chi, with 2^128 = -2^32 (mod p); the carry counters stay below 2^5
        inline V reduce() const {
            const V one = V::set1(1), eps = V::set1(E);
            const V h2 = vadd(hi, clo);
            const V top = vadd_if_lt(chi, h2, clo, one);
            const V r0 = red(h2, lo);
            const V x = vsll32(top);
            V r = vsub(r0, x);
            return vsub_if_lt(r, r0, x, eps);  // borrowed 2^64 = 2^32 - 1; r0 - x + 2^64 >= 2^64 - 2^37, no second borrow
        }
    };
    // the MDS layer of a full round: entries below 2^6, so the 32-bit halves of the words are combined separately
    // (sums below 2^42) and the 74-bit result lo + 2^32 hi is folded once
    static inline void mds_small(const V* in, V* out) {
        const V m32 = V::set1(E), one = V::set1(1);
        V lo[24], hi[24];
        for (int i = 0; i < 12; i++) {
            lo[i] = lo[i + 12] = vand(in[i], m32);
            hi[i] = hi[i + 12] = vsrl32(in[i]);
        }
        V circ[12];
        for (int i = 0; i < 12; i++) circ[i] = V::set1(MDS_CIRC[i]);
        for (int r = 0; r < 12; r++) {
            V al = vmul32(lo[r], circ[0]), ah = vmul32(hi[r], circ[0]);
            for (int i = 1; i < 12; i++) {
                al = vadd(al, vmul32(lo[i + r], circ[i]));
                ah = vadd(ah, vmul32(hi[i + r], circ[i]));
            }
            if (MDS_DIAG[r]) {
                const V d = V::set1(MDS_DIAG[r]);
                al = vadd(al, vmul32(lo[r], d));
                ah = vadd(ah, vmul32(hi[r], d));
            }
            const V x = vsll32(ah);
            const V low = vadd(al, x);
            const V top = vadd_if_lt(vsrl32(ah), low, x, one);  // bits 64.. of al + 2^32 ah: below 2^11
            const V t = vsub(vsll32(top), top);                  // top * (2^32 - 1)
            const V s = vadd(low, t);
            out[r] = vadd_if_lt(s, s, t, m32);
        }
    }
    static inline void full_round(V st[12], const u64* rc, const u64* extra) {
        V sb[12];
        for (int i = 0; i < 12; i++) {
            V x = add_const(st[i], rc[i]);
            if (extra) x = add_const(x, extra[i]);
            sb[i] = pow7(x);
        }
        mds_small(sb, st);
    }
    static inline void permute(V st[12], const SparsePoseidon& S) {
        for (int r = 0; r < 4; r++) full_round(st, ROUND_CONSTANTS + 12 * r, nullptr);
        {
            V t[11];
            for (int r = 0; r < 11; r++) {
                Wide a;
                for (int c = 0; c < 11; c++) a.mac(V::set1(S.HEAD[r][c]), st[c + 1]);
                t[r] = a.reduce();
            }
            for (int r = 0; r < 11; r++) st[r + 1] = t[r];
        }
        for (int i = 0; i < 22; i++) {
            const V s0 = pow7(add_const(st[0], S.PRE[i]));
            Wide a;
            a.mac(V::set1(S.m00), s0);
            for (int j = 0; j < 11; j++) a.mac(V::set1(S.ROW[i][j]), st[j + 1]);
            for (int j = 0; j < 11; j++) st[j + 1] = mul_add(V::set1(S.COL[i][j]), s0, st[j + 1]);
            st[0] = a.reduce();
        }
        full_round(st, ROUND_CONSTANTS + 12 * 26, S.TAIL);
        for (int r = 27; r < 30; r++) full_round(st, ROUND_CONSTANTS + 12 * r, nullptr);
        for (int i = 0; i < 12; i++) st[i] = canon(st[i]);
    }
    // ---- canonical-in, canonical-out butterflies for the transform of the low-degree extension (fft_bitrev_out)
    static inline V cadd(V a, V b) {
        const V s = vadd(a, b);
        return canon(vadd_if_lt(s, s, a, V::set1(E)));  // wrapped: a + b - 2^64 < 2^64 - 2^33, so + (2^32 - 1) stays below 2^64
    }
    static inline V csub(V a, V b) { return vsub_if_lt(vsub(a, b), a, b, V::set1(E)); }  // a < b: + p = - (2^32 - 1) mod 2^64
    // one decimation-in-frequency stage with half >= V::W: pairs (k + j, k + j + half), twiddle tw[j] (contiguous per stage)
    static inline void dif_stage(u64* a, size_t n, size_t half, const u64* tw) {
        for (size_t k = 0; k < n; k += 2 * half)
            for (size_t j = 0; j < half; j += V::W) {
                const V u = V::gather(a + k + j, 1), v = V::gather(a + k + j + half, 1), w = V::gather(tw + j, 1);
                vstore(a + k + j, cadd(u, v));
                vstore(a + k + j + half, canon(mul(csub(u, v), w)));
            }
    }
    // test hook (tests/test_oracle_kat.py): the lane arithmetic on arbitrary 64-bit words, canonicalised --
    // out[0] = a b, out[1] = a^2, out[2] = a b + c, out[3] = a b + b c + c a (wide accumulator), out[4] = a^7, out[5] = a + (c mod p)
    static inline void test_arith(const u64* a, const u64* b, const u64* c, u64* out) {
        V va = V::gather(a, 1), vb = V::gather(b, 1), vc = V::gather(c, 1);
        vstore(out, canon(mul(va, vb)));
        vstore(out + V::W, canon(sqr(va)));
        vstore(out + 2 * V::W, canon(mul_add(va, vb, vc)));
        Wide w;
        w.mac(va, vb);
        w.mac(vb, vc);
        w.mac(vc, va);
        vstore(out + 3 * V::W, canon(w.reduce()));
        vstore(out + 4 * V::W, canon(pow7(va)));
        vstore(out + 5 * V::W, canon(add_const(va, c[0] % MODULUS)));
    }
    // hash_no_pad of V::W rows of `width` words, `row_stride` words apart; digests to out[lane]
    static inline void hash_rows(const SparsePoseidon& S, const u64* rows, size_t row_stride, size_t width, Digest* out) {
        V st[12];
        for (int i = 0; i < 12; i++) st[i] = V::set1(0);
        for (size_t off = 0; off < width; off += 8) {
            const size_t k = width - off < 8 ? width - off : 8;
            for (size_t i = 0; i < k; i++) st[i] = V::gather(rows + off + i, row_stride);
            permute(st, S);
        }
        u64 tmp[4][V::W];
        for (int i = 0; i < 4; i++) vstore(tmp[i], st[i]);
        for (int l = 0; l < V::W; l++)
            for (int i = 0; i < 4; i++) out[l].e[i] = tmp[i][l];
    }
    // two-to-one compression of V::W adjacent pairs: dst[l] = H(src[2 l] || src[2 l + 1])
    static inline void compress_pairs(const SparsePoseidon& S, const Digest* src, Digest* dst) {
        V st[12];
        for (int i = 0; i < 8; i++) st[i] = V::gather(&src[0].e[0] + i, 8);
        for (int i = 8; i < 12; i++) st[i] = V::set1(0);
        permute(st, S);
        u64 tmp[4][V::W];
        for (int i = 0; i < 4; i++) vstore(tmp[i], st[i]);
        for (int l = 0; l < V::W; l++)
            for (int i = 0; i < 4; i++) dst[l].e[i] = tmp[i][l];
    }
};

}  // namespace orc
