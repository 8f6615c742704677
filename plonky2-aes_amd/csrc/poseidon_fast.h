// Poseidon-12 for the hashing kernels: same permutation as gl::poseidon (gl.h), restructured for VALU issue slots
// (15.5 k VALU instructions per permutation, PMC-counted: profiles/r02_valu.json, r03_valu.json; the plain 30-round form
// compiles to 41 k, round 1's restructuring to 28 k):
//   * lazy reduction -- state words are arbitrary u64 representatives (not < p) inside the permutation; every product is
//     reduced once from 128 bits without canonicalisation; outputs are canonicalised;
//   * multiply-reduce built from v_mad_u64_u32 (gl::mulr_add_dev): the mad is a 64-bit adder with a free multiplier and a
//     carry-out, and costs what ONE 32-bit add-with-carry costs (tools/microbench/valu_rates.hip);
//   * round constants are never added on their own: every linear layer starts its accumulators from the constants of the
//     layer that follows;
//   * the 22 partial rounds use the sparse-matrix form derived by tools/gen_poseidon_fast.py: 23 multiply-accumulates per
//     round instead of a 144-term MDS, the 12-term dot product in a carry-counting accumulator reduced once, and the dense
//     11x11 layer that opens them merged into the fourth full round's linear step (PF_E);
//   * poseidon_coop: one state over 12 lanes of a 16-lane group, for the sequential Fiat-Shamir chain.
// The host build of the same functions (plain 128-bit arithmetic) is what p2_selftest_host and the CPU tests exercise.
#pragma once
#include "gl.h"

namespace glf {
using gl::u32;
using gl::u64;

#if defined(__HIP_DEVICE_COMPILE__)
#define P2F_DECL __device__ __constant__ static
#else
#define P2F_DECL static
#endif
#include "poseidon_fast.inc"

GL_HD void mul128(u64 a, u64 b, u64& hi, u64& lo) {
#if defined(__HIP_DEVICE_COMPILE__)
    // 4 x (32x32 + 64 -> 64) = 4 v_mad_u64_u32; every partial sum below provably fits 64 bits
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 p00 = (u64)a0 * b0;
    u64 mid = (u64)a0 * b1 + (p00 >> 32);
    u64 mid2 = (u64)a1 * b0 + (u32)mid;
    lo = (mid2 << 32) | (u32)p00;
    hi = (u64)a1 * b1 + (mid >> 32) + (mid2 >> 32);
#else
    unsigned __int128 m = (unsigned __int128)a * b;
    lo = (u64)m;
    hi = (u64)(m >> 64);
#endif
}
// hi*2^64 + lo  ->  some u64 congruent mod p (2^64 = 2^32 - 1, 2^96 = -1; no canonicalisation).
// T = lo + (hl << 32) - (hl + hh) is formed with 32-bit add/subtract-with-carry chains; the number of 2^64 wraps,
// net = carry - borrow in {-1, 0, 1}, is folded back as T - net * p = T - (net << 32) ... + net, again on the halves, so no
// 64-bit compare-and-select is needed: the compiler's version of plonky2's reduce128 spends two of those per reduction
// (v_cmp_lt_u64 + 64-bit add + two v_cndmask, and a v_cndmask on VCC alone costs 23 cycles -- tools/microbench/valu_rates.hip).
// The result T - net * p lies in [0, 2^64) for every input (checked against 128-bit arithmetic on 2*10^8 inputs and all
// combinations of extreme halves).
GL_HD u64 red128(u64 hi, u64 lo) {
#if defined(__HIP_DEVICE_COMPILE__)
    return gl::red128_dev(hi, lo);  // one mad, two subtracts, the wave-uniform correction (gl.h)
#endif
    const u32 l0 = (u32)lo, l1 = (u32)(lo >> 32), hl = (u32)hi, hh = (u32)(hi >> 32);
    u32 C, cv, b, B;
    const u32 u1 = __builtin_addc(l1, hl, 0u, &C);   // lo + (hl << 32): carry C
    const u32 mC = 0u - C;
    const u32 v0 = __builtin_addc(hl, hh, 0u, &cv);  // v = hl + hh, 33 bits
    const u32 w0 = __builtin_subc(l0, v0, 0u, &b);
    u32 w1 = __builtin_subc(u1, cv, b, &B);          // w = u - v: borrow B
    const u32 net = 0u - mC - B;                     // C - B
    w1 += net;                                       // + net * 2^32
    const u32 sx = (u32)((int)net >> 31);
    u32 r0 = __builtin_subc(w0, net, 0u, &b);        // - net (sign-extended)
    u32 r1 = __builtin_subc(w1, sx, b, &B);
#if defined(__HIP_DEVICE_COMPILE__)
    // Value barrier.  ROCm 7.2's AMDGPU backend folds "x - borrow" into a following add-with-carry as "+ 0xFFFFFFFF", which
    // keeps the sum but not the carry-out: a reduction whose result feeds an add-with-carry chain in the same basic block
    // was miscompiled that way (tools/microbench/reduce_check.hip shows it: 4.7 % wrong results when a carry-based
    // canonicalisation follows directly).  The empty asm makes r0, r1 opaque to that combine; it emits nothing.
    asm("" : "+v"(r0), "+v"(r1));
#endif
    return ((u64)r1 << 32) | r0;
}
// a * b mod p as some u64 (a, b arbitrary u64): gl::mulr_add_dev on the device (see gl.h for the sequence).
GL_HD u64 mulr(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return gl::mulr_add_dev<false>(a, b, 0);
#else
    u64 hi, lo;
    mul128(a, b, hi, lo);
    return red128(hi, lo);
#endif
}
GL_HD u64 canon(u64 a) {
#if defined(__HIP_DEVICE_COMPILE__)
    return gl::canon_dev(a);
#else
    return a >= gl::P ? a - gl::P : a;
#endif
}
GL_HD u64 sbox7(u64 x) {
    u64 x2 = mulr(x, x), x3 = mulr(x2, x), x4 = mulr(x2, x2);
    return mulr(x3, x4);
}
// Accumulator for sums of 64x64-bit products (dot products of a sparse/dense matrix row with the state).
// Carries are never propagated between words: each of the four 32x32 partial products is accumulated with one
// v_mad_u64_u32 into its own aligned 64-bit lane (E01: a0*b0, O01: a0*b1 + a1*b0 (weight 2^32), E23: a1*b1 (weight
// 2^64)) and the carry-out of every mad is counted in a 32-bit counter.  8 VALU instructions per term, no moves; the
// four carry SGPR pairs are distinct and each is consumed >= 2 issue slots after it is produced, which is the wait
// gfx950 requires between a VALU SGPR write and its VALU reader (the compiler pads its own carry chains with
// s_nop/v_mov for this reason).  reduce() folds the five words once.
struct Acc {
    u64 e01, o01, e23;
    u32 ce0, co, ce2;
    GL_HD void init() {
        e01 = o01 = e23 = 0;
        ce0 = co = ce2 = 0;
    }
    GL_HD void fma(u64 a, u64 b) {
        u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
#if defined(__HIP_DEVICE_COMPILE__)
        unsigned long long s1, s2, s3, s4;
        asm("v_mad_u64_u32 %[e01], %[s1], %[a0], %[b0], %[e01]\n\t"
            "v_mad_u64_u32 %[o01], %[s2], %[a0], %[b1], %[o01]\n\t"
            "v_mad_u64_u32 %[e23], %[s3], %[a1], %[b1], %[e23]\n\t"
            "v_addc_co_u32_e64 %[ce0], %[s1], 0, %[ce0], %[s1]\n\t"
            "v_mad_u64_u32 %[o01], %[s4], %[a1], %[b0], %[o01]\n\t"
            "v_addc_co_u32_e64 %[co], %[s2], 0, %[co], %[s2]\n\t"
            "v_addc_co_u32_e64 %[ce2], %[s3], 0, %[ce2], %[s3]\n\t"
            "v_addc_co_u32_e64 %[co], %[s4], 0, %[co], %[s4]"
            : [e01] "+v"(e01), [o01] "+v"(o01), [e23] "+v"(e23), [ce0] "+v"(ce0), [co] "+v"(co), [ce2] "+v"(ce2), [s1] "=&s"(s1), [s2] "=&s"(s2),
              [s3] "=&s"(s3), [s4] "=&s"(s4)
            : [a0] "v"(a0), [a1] "v"(a1), [b0] "v"(b0), [b1] "v"(b1));
#else
        u64 p, t;
        p = (u64)a0 * b0; t = e01 + p; ce0 += t < p; e01 = t;
        p = (u64)a0 * b1; t = o01 + p; co += t < p; o01 = t;
        p = (u64)a1 * b1; t = e23 + p; ce2 += t < p; e23 = t;
        p = (u64)a1 * b0; t = o01 + p; co += t < p; o01 = t;
#endif
    }
    // the same with `k` a UNIFORM table constant (PF_E, PF_WHAT rows fetched by scalar loads): its halves are read straight
    // from SGPRs -- every mad here has exactly one scalar source -- instead of being copied into VGPRs first
    GL_HD void fma_k(u64 k, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
        const u32 a0 = (u32)k, a1 = (u32)(k >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
        unsigned long long s1, s2, s3, s4;
        asm("v_mad_u64_u32 %[e01], %[s1], %[a0], %[b0], %[e01]\n\t"
            "v_mad_u64_u32 %[o01], %[s2], %[a0], %[b1], %[o01]\n\t"
            "v_mad_u64_u32 %[e23], %[s3], %[a1], %[b1], %[e23]\n\t"
            "v_addc_co_u32_e64 %[ce0], %[s1], 0, %[ce0], %[s1]\n\t"
            "v_mad_u64_u32 %[o01], %[s4], %[a1], %[b0], %[o01]\n\t"
            "v_addc_co_u32_e64 %[co], %[s2], 0, %[co], %[s2]\n\t"
            "v_addc_co_u32_e64 %[ce2], %[s3], 0, %[ce2], %[s3]\n\t"
            "v_addc_co_u32_e64 %[co], %[s4], 0, %[co], %[s4]"
            : [e01] "+v"(e01), [o01] "+v"(o01), [e23] "+v"(e23), [ce0] "+v"(ce0), [co] "+v"(co), [ce2] "+v"(ce2), [s1] "=&s"(s1), [s2] "=&s"(s2),
              [s3] "=&s"(s3), [s4] "=&s"(s4)
            : [a0] "s"(a0), [a1] "s"(a1), [b0] "v"(b0), [b1] "v"(b1));
#else
        fma(k, b);
#endif
    }
    // a < 2^32: only the two products with a's low half exist
    GL_HD void fma_small(u32 a0, u64 b) {
        u32 b0 = (u32)b, b1 = (u32)(b >> 32);
#if defined(__HIP_DEVICE_COMPILE__)
        unsigned long long s1, s2;
        asm("v_mad_u64_u32 %[e01], %[s1], %[a0], %[b0], %[e01]\n\t"
            "v_mad_u64_u32 %[o01], %[s2], %[a0], %[b1], %[o01]\n\t"
            "s_nop 0\n\t"
            "v_addc_co_u32_e64 %[ce0], %[s1], 0, %[ce0], %[s1]\n\t"
            "v_addc_co_u32_e64 %[co], %[s2], 0, %[co], %[s2]"
            : [e01] "+v"(e01), [o01] "+v"(o01), [ce0] "+v"(ce0), [co] "+v"(co), [s1] "=&s"(s1), [s2] "=&s"(s2)
            : [a0] "v"(a0), [b0] "v"(b0), [b1] "v"(b1));
#else
        u64 p, t;
        p = (u64)a0 * b0; t = e01 + p; ce0 += t < p; e01 = t;
        p = (u64)a0 * b1; t = o01 + p; co += t < p; o01 = t;
#endif
    }
    // value = e01 + 2^32*o01 + 2^64*(e23 + ce0 + 2^32*co) + 2^128*ce2, summed limb by limb with carry chains (no 64-bit
    // compares), then 2^128 = -2^32 (mod p).  Some u64 congruent to the value; not canonical.
    GL_HD u64 reduce() const {
        u32 k1, k2, k2b, k3, k3b, b, b2;
        const u32 L0 = (u32)e01;
        const u32 L1 = __builtin_addc((u32)(e01 >> 32), (u32)o01, 0u, &k1);
        u32 L2 = __builtin_addc((u32)(o01 >> 32), (u32)e23, k1, &k2);
        L2 = __builtin_addc(L2, ce0, 0u, &k2b);
        u32 L3 = __builtin_addc((u32)(e23 >> 32), co, k2, &k3);
        L3 = __builtin_addc(L3, 0u, k2b, &k3b);
        const u32 L4 = ce2 + k3 + k3b;  // multiples of 2^128: tiny
        const u64 r = red128(((u64)L3 << 32) | L2, ((u64)L1 << 32) | L0);
        // r - L4 * 2^32; a borrow is worth -2^64 = -(2^32 - 1)
        const u32 h1 = __builtin_subc((u32)(r >> 32), L4, 0u, &b);
        u32 lo = __builtin_subc((u32)r, 0u - b, 0u, &b2);
        u32 hi = h1 - b2;
#if defined(__HIP_DEVICE_COMPILE__)
        asm("" : "+v"(lo), "+v"(hi));  // value barrier, see red128
#endif
        return ((u64)hi << 32) | lo;
    }
};

// a arbitrary u64, c canonical (< p) -> some u64 congruent to a + c.  The wrap is folded back as + EPS, which cannot
// wrap again (a + c - 2^64 < c <= p - 1).  Carry chains only: a 64-bit compare-and-select costs a v_cmp_*_u64 plus two
// v_cndmask_b32 on VCC, and tools/microbench/valu_rates.hip measures the latter at 23 cycles each on gfx950.
GL_HD u64 add_wrap(u64 a, u64 c) {
    u32 k, K, b, b2;
    const u32 lo = __builtin_addc((u32)a, (u32)c, 0u, &k);
    const u32 hi = __builtin_addc((u32)(a >> 32), (u32)(c >> 32), k, &K);
    // + K * (2^32 - 1):  lo - K, hi + K - borrow
    u32 r0 = __builtin_subc(lo, 0u, K, &b);
    u32 r1 = __builtin_addc(hi, 0u, K, &b2);
    r1 -= b;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+v"(r0), "+v"(r1));  // value barrier, see red128
#endif
    return ((u64)r1 << 32) | r0;
}

// value = al + 2^32 * ah with al, ah < 2^43 (sums of at most 13 products of a 32-bit half with a coefficient < 2^6, plus
// a 32-bit half of a round constant)  ->  some u64 congruent to it.
//   ah' = ah + (al >> 32);  value = x0 + 2^32 x1 + 2^64 x2  with x0 = lo32(al), (x1, x2) = halves of ah', x2 < 2^12
//   = {x1:x0} + x2 * (2^32 - 1): ONE v_mad_u64_u32 (the 64-bit add rides on the multiply; a 32-bit add-with-carry costs
//   the same issue time as the whole mad, valu_rates.hip), and its carry-out -- possible only when x1 >= 2^32 - 2^12 --
//   is folded back by a second mad, which cannot wrap (the wrapped sum is < 2^44).
GL_HD u64 fold_al_ah(u64 al, u64 ah) {
#if defined(__HIP_DEVICE_COMPILE__)
    const u64 ah2 = gl::add_u32(ah, (u32)(al >> 32));  // through the multiplier: no zero-extension of al's high half
#else
    const u64 ah2 = ah + (al >> 32);
#endif
    const u32 x2 = (u32)(ah2 >> 32);
    const u64 base = (ah2 << 32) | (u32)al;
#if defined(__HIP_DEVICE_COMPILE__)
    // the carry needs x1 >= 2^32 - 2^12: once in 2^20 on random data, so its fold sits behind a wave-uniform branch
    gl::sg sc, dead;
    u64 r = gl::mad_eps_co(x2, base, sc);
    if (__builtin_expect(sc != 0, 0)) r = gl::mad_eps_co(gl::one_where(sc), r, dead);
    return r;
#else
    const u64 t = base + (u64)x2 * gl::EPS;
    return t < base ? t + gl::EPS : t;
#endif
}

// MDS layer of a full round, with the NEXT round's constants folded into the accumulators (rc == nullptr: none):
//   out[r] = rc[r] + sum_i circ[i] * s[(i + r) % 12] + 8 * s[0] (r == 0)      as some u64 representative.
// Evaluated on 32-bit halves: every accumulator stays below 2^43, one fold per output word.
GL_HD void mds_full(u64* s, const unsigned long long* rc) {
#if defined(__HIP_DEVICE_COMPILE__)
    // Every term is one v_mad_u64_u32 with the coefficient as an inline constant (gl::madk: the coefficients 2, 8 and 16 must
    // not become shift-adds on zero-extended operands), and the round constant is the ADDEND of the first one, read from its
    // SGPR pair: no separate additions, no moves.
    u32 l[12], h[12];
    u64 res[12];
#pragma unroll
    for (int i = 0; i < 12; i++) {
        l[i] = (u32)s[i];
        h[i] = (u32)(s[i] >> 32);
    }
    // One asm block per output word: 24 (26) mads on two accumulators.  As separate statements each mad is followed by the
    // compiler's boundary pad (it defines an SGPR pair, the unused carry-out, and the compiler cannot see that nobody reads
    // it): 2 500 s_nop per permutation.  29 operands -- the limit is 30.
#define P2_MDS_T(n, K) "v_mad_u64_u32 %[al], %[d], %[l" #n "], " #K ", %[al]\n\tv_mad_u64_u32 %[ah], %[d], %[h" #n "], " #K ", %[ah]\n\t"
#define P2_MDS_TAIL P2_MDS_T(1, 15) P2_MDS_T(2, 41) P2_MDS_T(3, 16) P2_MDS_T(4, 2) P2_MDS_T(5, 28) P2_MDS_T(6, 13) P2_MDS_T(7, 13) \
                    P2_MDS_T(8, 39) P2_MDS_T(9, 18) P2_MDS_T(10, 34) P2_MDS_T(11, 20)
#define P2_MDS_IN(r)                                                                                                                          \
    [l0] "v"(l[(0 + r) % 12]), [h0] "v"(h[(0 + r) % 12]), [l1] "v"(l[(1 + r) % 12]), [h1] "v"(h[(1 + r) % 12]), [l2] "v"(l[(2 + r) % 12]),     \
        [h2] "v"(h[(2 + r) % 12]), [l3] "v"(l[(3 + r) % 12]), [h3] "v"(h[(3 + r) % 12]), [l4] "v"(l[(4 + r) % 12]), [h4] "v"(h[(4 + r) % 12]), \
        [l5] "v"(l[(5 + r) % 12]), [h5] "v"(h[(5 + r) % 12]), [l6] "v"(l[(6 + r) % 12]), [h6] "v"(h[(6 + r) % 12]), [l7] "v"(l[(7 + r) % 12]), \
        [h7] "v"(h[(7 + r) % 12]), [l8] "v"(l[(8 + r) % 12]), [h8] "v"(h[(8 + r) % 12]), [l9] "v"(l[(9 + r) % 12]), [h9] "v"(h[(9 + r) % 12]), \
        [l10] "v"(l[(10 + r) % 12]), [h10] "v"(h[(10 + r) % 12]), [l11] "v"(l[(11 + r) % 12]), [h11] "v"(h[(11 + r) % 12])
#pragma unroll
    for (int r = 0; r < 12; r++) {
        u64 al, ah;
        gl::sg dead;
        if (rc) {
            const u64 rl = (u64)(u32)rc[r], rh = (u64)(rc[r] >> 32);
            if (r == 0)
                asm("v_mad_u64_u32 %[al], %[d], %[l0], 17, %[rl]\n\tv_mad_u64_u32 %[ah], %[d], %[h0], 17, %[rh]\n\t" P2_MDS_TAIL
                    "v_mad_u64_u32 %[al], %[d], %[l0], 8, %[al]\n\tv_mad_u64_u32 %[ah], %[d], %[h0], 8, %[ah]"
                    : [al] "=&v"(al), [ah] "=&v"(ah), [d] "=&s"(dead)
                    : [rl] "s"(rl), [rh] "s"(rh), P2_MDS_IN(r));
            else
                asm("v_mad_u64_u32 %[al], %[d], %[l0], 17, %[rl]\n\tv_mad_u64_u32 %[ah], %[d], %[h0], 17, %[rh]\n\t" P2_MDS_TAIL
                    : [al] "=&v"(al), [ah] "=&v"(ah), [d] "=&s"(dead)
                    : [rl] "s"(rl), [rh] "s"(rh), P2_MDS_IN(r));
        } else {
            if (r == 0)
                asm("v_mad_u64_u32 %[al], %[d], %[l0], 17, 0\n\tv_mad_u64_u32 %[ah], %[d], %[h0], 17, 0\n\t" P2_MDS_TAIL
                    "v_mad_u64_u32 %[al], %[d], %[l0], 8, %[al]\n\tv_mad_u64_u32 %[ah], %[d], %[h0], 8, %[ah]"
                    : [al] "=&v"(al), [ah] "=&v"(ah), [d] "=&s"(dead)
                    : P2_MDS_IN(r));
            else
                asm("v_mad_u64_u32 %[al], %[d], %[l0], 17, 0\n\tv_mad_u64_u32 %[ah], %[d], %[h0], 17, 0\n\t" P2_MDS_TAIL
                    : [al] "=&v"(al), [ah] "=&v"(ah), [d] "=&s"(dead)
                    : P2_MDS_IN(r));
        }
        res[r] = fold_al_ah(al, ah);
    }
#undef P2_MDS_T
#undef P2_MDS_TAIL
#undef P2_MDS_IN
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = res[i];
    return;
#endif
    const u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    u64 lo[12], hi[12], out[12];
#pragma unroll
    for (int i = 0; i < 12; i++) {
        lo[i] = s[i] & gl::EPS;
        hi[i] = s[i] >> 32;
    }
#pragma unroll
    for (int r = 0; r < 12; r++) {
        u64 al = rc ? (u64)(u32)rc[r] : 0, ah = rc ? (u64)(rc[r] >> 32) : 0;
#pragma unroll
        for (int i = 0; i < 12; i++) {
            int j = (i + r) % 12;
            al += lo[j] * C[i];
            ah += hi[j] * C[i];
        }
        if (r == 0) {
            al += lo[0] * 8;
            ah += hi[0] * 8;
        }
        out[r] = fold_al_ah(al, ah);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = out[i];
}

// Row 0 of the MDS alone: rc + sum_i circ[i] * s[i] + 8 * s[0]
GL_HD u64 mds_row0(const u64* s, u64 rc) {
#if defined(__HIP_DEVICE_COMPILE__)
    {
        u64 al = gl::madk_s<17>((u32)s[0], (u64)(u32)rc), ah = gl::madk_s<17>((u32)(s[0] >> 32), rc >> 32);
#define P2_MDS_TERM(i, K)                       \
    al = gl::madk<K>((u32)s[i], al);            \
    ah = gl::madk<K>((u32)(s[i] >> 32), ah);
        P2_MDS_TERM(1, 15) P2_MDS_TERM(2, 41) P2_MDS_TERM(3, 16) P2_MDS_TERM(4, 2) P2_MDS_TERM(5, 28) P2_MDS_TERM(6, 13)
        P2_MDS_TERM(7, 13) P2_MDS_TERM(8, 39) P2_MDS_TERM(9, 18) P2_MDS_TERM(10, 34) P2_MDS_TERM(11, 20) P2_MDS_TERM(0, 8)
#undef P2_MDS_TERM
        return fold_al_ah(al, ah);
    }
#endif
    const u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    u64 al = (u32)rc, ah = rc >> 32;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        al += (s[i] & gl::EPS) * C[i];
        ah += (s[i] >> 32) * C[i];
    }
    al += (s[0] & gl::EPS) * 8;
    ah += (s[0] >> 32) * 8;
    return fold_al_ah(al, ah);
}

// One full round on a state that already carries this round's constants: S-box, then MDS + next constants.
GL_HD void full_round(u64* s, const unsigned long long* rc_next) {
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = sbox7(s[i]);
    mds_full(s, rc_next);
}

// a * b + c mod p as some u64 (a, b, c arbitrary u64): the addend rides on the multiply-adds (gl.h)
GL_HD u64 mulr_add(u64 a, u64 b, u64 c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return gl::mulr_add_dev<true>(a, b, c);
#else
    unsigned __int128 m = (unsigned __int128)a * b + c;
    return red128((u64)(m >> 64), (u64)m);
#endif
}

// the same with a uniform table constant as the first factor
GL_HD u64 mulr_add_k(u64 k, u64 b, u64 c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return gl::mulr_add_dev<true, true>(k, b, c);
#else
    return mulr_add(k, b, c);
#endif
}

// In: canonical or not; out: canonical.
GL_HD void poseidon(u64* s) {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned long long* RC = (const unsigned long long*)gl::D_POSEIDON_RC;
#else
    const unsigned long long* RC = (const unsigned long long*)gl::H_POSEIDON_RC;
#endif
    // Round constants are never added on their own (except the very first ones): each layer's linear step starts its
    // accumulators from the constants of the layer that follows.
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = add_wrap(s[i], RC[i]);
    for (int r = 0; r < 3; r++) full_round(s, RC + 12 * (r + 1));
    {
        // 4th full round with the dense 11x11 layer of the partial-round block merged into its linear step: lane 0 is the
        // MDS row (+ a_0, the first partial S-box's constant), lanes 1.. are rows of E = D0 . MDS[1.., :] (gen_poseidon_fast.py)
        u64 z[12];
#pragma unroll
        for (int i = 0; i < 12; i++) z[i] = sbox7(s[i]);
#pragma nounroll
        for (int r = 0; r < 11; r++) {
            Acc a;
            a.init();
#pragma unroll
            for (int c = 0; c < 12; c++) a.fma_k(PF_E[r * 12 + c], z[c]);
            s[1 + r] = a.reduce();  // straight into the (dead) state: a separate result array costs 34 VGPRs and a wave of occupancy
        }
        s[0] = mds_row0(z, PF_A[0]);
    }
    for (int i = 0; i < 22; i++) {
        u64 s0 = sbox7(s[0]);
        Acc a;
        a.init();
        a.e01 = i < 21 ? PF_A[i + 1] : PF_RC26[0];  // the next S-box's / next full round's constant for lane 0
        a.fma_small(25, s0);
#pragma unroll
        for (int j = 0; j < 11; j++) a.fma_k(PF_WHAT[i * 11 + j], s[1 + j]);
#pragma unroll
        for (int j = 0; j < 11; j++) s[1 + j] = mulr_add_k(PF_V[i * 11 + j], s0, s[1 + j]);
        s[0] = a.reduce();
    }
#pragma unroll
    for (int j = 1; j < 12; j++) s[j] = add_wrap(s[j], PF_RC26[j]);
    for (int r = 26; r < 29; r++) full_round(s, RC + 12 * (r + 1));
    full_round(s, nullptr);
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = canon(s[i]);
}

#if defined(__HIPCC__)
// ---- cooperative form: ONE sponge state spread over 12 lanes of a 16-lane group (lane i holds word i; lanes 12..15 are
// idle and must hold 0).  Same permutation, same constants; the linear layers gather the other words with cross-lane
// reads (ds_bpermute) and the partial rounds' dot product is a 4-step butterfly sum.  4.3 k VALU instructions per lane
// instead of 15.5 k: this is for the Fiat-Shamir chain (k_challenger), where ~110 permutations per proof are strictly
// sequential and a thread-per-sponge kernel leaves a single proof waiting on one lane's instruction stream.
__device__ __forceinline__ u64 shfl64(u64 v, int src_lane) { return (u64)__shfl((unsigned long long)v, src_lane, 64); }
__device__ inline u64 poseidon_coop(u64 w, const u32 i /* lane within the 16-lane group */) {
    const int lane = (int)(threadIdx.x & 63), gbase = lane & ~15;
    const bool live = i < 12;
    const u32 ii = live ? i : 0;  // index clamp for the idle lanes' (unused) constant loads
    const unsigned long long* RC = (const unsigned long long*)gl::D_POSEIDON_RC;
    // circulant MDS row of this lane on the group's words z, plus the next constant
    auto mds = [&](u64 z, u64 rc) -> u64 {
        const u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
        u64 al = (u32)rc, ah = rc >> 32;
#pragma unroll
        for (int k = 0; k < 12; k++) {
            u32 src = i + k;
            src = src >= 12 ? src - 12 : src;  // (i + k) mod 12 for live lanes; idle lanes read some lane of the group, unused
            const u64 x = shfl64(z, gbase + (int)(src & 15));
            const u32 c = C[k] + ((k == 0 && i == 0) ? 8u : 0u);
            al += (x & gl::EPS) * c;
            ah += (x >> 32) * c;
        }
        return fold_al_ah(al, ah);
    };
    w = live ? add_wrap(w, RC[ii]) : 0;
    for (int r = 0; r < 3; r++) w = mds(sbox7(w), live ? RC[12 * (r + 1) + ii] : 0);
    {   // 4th full round with the dense layer merged in (PF_E): lane 0 takes the MDS row, lanes 1..11 a row of E
        const u64 z = sbox7(w);
        const u64 row0 = mds(z, i == 0 ? PF_A[0] : 0);
        const u32 er = (live && i >= 1) ? i - 1 : 0;
        Acc a;
        a.init();
#pragma unroll
        for (int c = 0; c < 12; c++) a.fma(PF_E[er * 12 + c], shfl64(z, gbase + c));
        const u64 rowe = a.reduce();
        w = i == 0 ? row0 : (live ? rowe : 0);
    }
    for (int pr = 0; pr < 22; pr++) {
        const u64 z0 = shfl64(sbox7(w), gbase);  // lane 0's S-box output, to everybody
        const u32 cj = (live && i >= 1) ? pr * 11 + (i - 1) : 0;
        const u64 coef = i == 0 ? 25 : (live ? PF_WHAT[cj] : 0);
        u64 term = canon(mulr(coef, i == 0 ? z0 : w));  // idle lanes: 0
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) term = gl::add(term, shfl64(term, lane ^ d));
        const u64 s0n = gl::add(term, pr < 21 ? PF_A[pr + 1] : PF_RC26[0]);
        const u64 wj = mulr_add((live && i >= 1) ? PF_V[cj] : 0, z0, w);
        w = i == 0 ? s0n : wj;
    }
    if (live && i >= 1) w = add_wrap(w, PF_RC26[ii]);
    for (int r = 26; r < 29; r++) w = mds(sbox7(w), live ? RC[12 * (r + 1) + ii] : 0);
    w = mds(sbox7(w), 0);
    return live ? canon(w) : 0;
}
#endif

}  // namespace glf
