// Goldilocks field (p = 2^64 - 2^32 + 1), its quadratic extension GF(p^2) = F[x]/(x^2 - 7) and the
// Poseidon-12 permutation, written once for host (g++) and device (hipcc, gfx950).
//
// What this replaces: plonky2::field::goldilocks_field::GoldilocksField, field::extension::quadratic and
// hash::poseidon (third-party crate `plonky2`, git rev 109d517d..., /root/reference/Cargo.toml:12 -- source
// not vendored; algorithm restated from its published definition).
//
// All values are kept CANONICAL (< p) after every operation so that buffers can be compared bit-for-bit
// with the CPU oracle.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GL_HD __host__ __device__ __forceinline__
#define GL_D __device__ __forceinline__
#else
#define GL_HD inline
#define GL_D inline
#endif

namespace gl {

typedef uint64_t u64;
typedef uint32_t u32;

static const u64 P = 0xFFFFFFFF00000001ULL;
static const u64 EPS = 0xFFFFFFFFULL;  // 2^64 mod p
// plonky2 GoldilocksField::MULTIPLICATIVE_GROUP_GENERATOR / POWER_OF_TWO_GENERATOR (second pair)
static const u64 MULT_GEN = 14293326489335486720ULL;
static const u64 POW2_GEN = 7277203076849721926ULL;  // order 2^32
static const u64 W_EXT = 7;                          // x^2 = 7

// ---- device building blocks (gfx950).  tools/microbench/valu_rates.hip measures, in cycles per wave-instruction per
// SIMD: plain two-source 32-bit ALU ops 2.3; EVERYTHING else -- v_mad_u64_u32, add/sub with carry, v_cndmask, v_cmp,
// three-source ops, v_lshl_add_u64 -- 4.1 (and v_cndmask_b32 on VCC 23).  So a v_mad_u64_u32 is a 64-bit adder with a free
// multiplier and a carry-out, and the field operations below are built from it rather than from compare-and-select.
// A wave-wide carry lives in an SGPR pair (`sg`).  gfx950 needs two wait states between a VALU write of an SGPR and a
// VALU read of it, and the compiler's hazard recogniser does not look inside inline asm: a helper that READS a carry starts
// with s_nop 1 -- unless its name ends in _settled (the caller's data flow puts two or more instructions of the wave between
// the write and this read; each such call site says why) or _salu (the mask was produced by the scalar unit, which needs no
// wait: the compiler itself issues v_cndmask right behind s_or_b64).  A nop delays only its own wave, about four cycles; other
// resident waves issue meanwhile, but not for free (all of them removed, unsafely: +2.2 % on the whole prover).
// Scalar ALU instructions do not belong in these asm strings: they write SCC, which the compiler tracks only for its own code.
#if defined(__HIP_DEVICE_COMPILE__)
typedef unsigned long long sg;
GL_D u64 mad_co(u32 a, u32 b, u64 c, sg& k) {  // a * b + c, carry-out in k
    u64 r;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(k) : "v"(a), "v"(b), "v"(c));
    return r;
}
// The same with a UNIFORM multiplier kept in an SGPR (table constants fetched with scalar loads): an asm operand declared
// "v" makes the compiler copy such a constant into a VGPR first, one v_mov per 32-bit half per use.  One scalar source per
// VALU instruction is what gfx9-family encodings allow; inline constants (integers 0..64, -1) do not count.
GL_D u64 mad_co_k(u32 k_uniform, u32 b, u64 c, sg& k) {
    u64 r;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(k) : "s"(k_uniform), "v"(b), "v"(c));
    return r;
}
// x * K + acc for a compile-time K in 0..64 (an inline constant): written as asm so that K = 2, 8, 16 stay multiplies -- the
// compiler turns those into v_lshl_add_u64 on a zero-extended operand, which costs two moves to build and the same long slot
template <u32 K>
GL_D u64 madk(u32 x, u64 acc) {
    static_assert(K <= 64, "not an inline constant");
    u64 r;
    sg dead;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(dead) : "v"(x), "n"(K), "v"(acc));
    return r;
}
template <u32 K>
GL_D u64 madk_s(u32 x, u64 acc_uniform) {  // the addend is a uniform 64-bit value in an SGPR pair (a round constant)
    static_assert(K <= 64, "not an inline constant");
    u64 r;
    sg dead;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(dead) : "v"(x), "n"(K), "s"(acc_uniform));
    return r;
}
template <u32 K>
GL_D u64 madk0(u32 x) {
    static_assert(K <= 64, "not an inline constant");
    u64 r;
    sg dead;
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(r), "=s"(dead) : "v"(x), "n"(K));
    return r;
}
GL_D u64 mad_eps_co(u32 a, u64 c, sg& k) {  // a * (2^32 - 1) + c, carry-out in k
    u64 r;
    asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(r), "=s"(k) : "v"(a), "v"(c));
    return r;
}
GL_D u64 add_u32(u64 c, u32 x) {  // c + x through the multiplier (x * 1 + c): one long slot, no zero-extension moves
    u64 r;
    sg dead;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %3" : "=v"(r), "=s"(dead) : "v"(x), "v"(c));
    return r;
}
GL_D u64 add_eps_co(u64 c, sg& k) {  // c + (2^32 - 1) = c - p (mod 2^64); carry-out k <=> c >= p
    u64 r;
    asm("v_mad_u64_u32 %0, %1, 1, -1, %2" : "=v"(r), "=s"(k) : "v"(c));
    return r;
}
GL_D u32 add_co(u32 x, u32 y, sg& cout) {  // x + y, carry-out in cout
    u32 r;
    asm("v_add_co_u32_e64 %0, %1, %2, %3" : "=v"(r), "=s"(cout) : "v"(x), "v"(y));
    return r;
}
GL_D u32 addc_co(u32 x, u32 y, sg cin, sg& cout) {  // x + y + cin, carry-out in cout
    u32 r;
    asm("s_nop 1\n\tv_addc_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(r), "=s"(cout) : "v"(x), "v"(y), "s"(cin));
    return r;
}
GL_D u32 sub_co(u32 x, u32 y, sg& cout) {  // x - y, borrow-out in cout
    u32 r;
    asm("v_sub_co_u32_e64 %0, %1, %2, %3" : "=v"(r), "=s"(cout) : "v"(x), "v"(y));
    return r;
}
GL_D u32 subb_co(u32 x, u32 y, sg cin, sg& cout) {  // x - y - cin, borrow-out in cout
    u32 r;
    asm("s_nop 1\n\tv_subb_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(r), "=s"(cout) : "v"(x), "v"(y), "s"(cin));
    return r;
}
// the same WITHOUT the leading wait states: only where the data flow guarantees that at least two instructions of this wave
// issue between the VALU write of `cin` and this read (in mulr_add_dev the carry k of the third product is read after the fourth
// product, two moves and the reduction's mad, all of which depend on it)
GL_D u32 subb_co_settled(u32 x, u32 y, sg cin, sg& cout) {
    u32 r;
    asm("v_subb_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(r), "=s"(cout) : "v"(x), "v"(y), "s"(cin));
    return r;
}
// Rs + (lanes of C ? 2^32 - 1 : 0), C settled (written at least two instructions earlier): a 0/1 select and one mad, one block
GL_D u64 add_eps_where_settled(sg C, u64 Rs) {
    u64 out;
    u32 c01;
    sg dead;
    asm("v_cndmask_b32_e64 %[c], 0, 1, %[m]\n\t"
        "v_mad_u64_u32 %[o], %[d], %[c], -1, %[r]"
        : [o] "=&v"(out), [c] "=&v"(c01), [d] "=&s"(dead)
        : [m] "s"(C), [r] "v"(Rs));
    return out;
}
GL_D u32 subb0_co(u32 x, sg cin, sg& cout) {  // x - cin, borrow-out in cout
    u32 r;
    asm("s_nop 1\n\tv_subb_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(r), "=s"(cout) : "v"(x), "s"(cin));
    return r;
}
GL_D u32 ones_where(sg m) {  // 0xFFFFFFFF in the lanes of m, else 0
    u32 r;
    asm("s_nop 1\n\tv_cndmask_b32_e64 %0, 0, -1, %1" : "=v"(r) : "s"(m));
    return r;
}
GL_D u32 pick(sg m, u32 yes, u32 no) {  // per lane: m ? yes : no
    u32 r;
    asm("s_nop 1\n\tv_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(no), "v"(yes), "s"(m));
    return r;
}
GL_D u32 one_where(sg m) {  // 1 in the lanes of mask m, 0 elsewhere
    u32 r;
    asm("s_nop 1\n\tv_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(r) : "s"(m));
    return r;
}
// Last step of a reduction.  Rs = R - (H.hi + k) is formed; C = carry of the mad that made R (worth + (2^32 - 1)), B = borrow of
// the subtraction (worth - (2^32 - 1)); neither correction can wrap again: after a carry R < (2^32-1)^2, after a borrow alone
// Rs >= 2^64 - 2^32.  A borrow needs R < H.hi + k <= 2^32 -- one product in 2^32 on random data, but exact zero limbs make it
// certain (2^48 * m 2^48 = m 2^96) -- so it is handled, on a WAVE-UNIFORM branch: B is a lane mask in an SGPR pair, and when it
// is zero for the whole wave (the usual case) what remains is + C (2^32 - 1): one 0/1 select and one mad, 2 long slots instead
// of the 3 long + 1 short of a branch-free two-sided correction.
// BRANCH = false keeps the branch-free two-sided correction: in the NTT kernels, whose waves wait on memory and LDS rather than
// on issue slots, the compare-and-branch per product lengthened every wave's chain (LDE 20.9 -> 22.3 ms per chunk).
template <bool BRANCH = true>
GL_D u64 red_fix(u64 Rs, sg C, sg B) {
    if (!BRANCH) {
        const u32 dh = ones_where(B & ~C);        // - (2^32 - 1) = + {1, 0xFFFFFFFF}
        const u32 dl = ones_where(C & ~B) - dh;   // + (2^32 - 1) = + {0xFFFFFFFF, 0}
        u64 d = ((u64)dh << 32) | dl;
        asm("" : "+v"(d));
        return Rs + d;
    }
    sg Cm = C;
    if (__builtin_expect(B != 0, 0)) {            // if-then only: the lanes that borrowed without a carry get - (2^32 - 1)
        const u32 dh = ones_where(B & ~C);        //   = + {1, 0xFFFFFFFF}; a borrow and a carry cancel
        u64 d = ((u64)dh << 32) | (0u - dh);
        asm("" : "+v"(d));                        // one 64-bit operand: one v_lshl_add_u64, not two adds
        Rs += d;
        Cm = C & ~B;
    }
    // C was written by the reduction's mad, two subtract instructions (and their wait states) ago; Cm on the rare path by SALU
    return add_eps_where_settled(Cm, Rs);
}
// per lane m ? yes : no on both halves of a 64-bit value, one block: the two wait states after the VALU write of m are paid once
GL_D u64 pick64(sg m, u64 yes, u64 no) {
    u32 lo, hi;
    asm("s_nop 1\n\tv_cndmask_b32_e64 %0, %3, %2, %6\n\tv_cndmask_b32_e64 %1, %5, %4, %6"
        : "=&v"(lo), "=&v"(hi)
        : "v"((u32)yes), "v"((u32)no), "v"((u32)(yes >> 32)), "v"((u32)(no >> 32)), "s"(m));
    return ((u64)hi << 32) | lo;
}
// the same for a mask that the SCALAR unit produced (an s_or / s_and of carry masks): no wait states -- the compiler itself
// issues v_cndmask right behind such an instruction
GL_D u64 pick64_salu(sg m, u64 yes, u64 no) {
    u32 lo, hi;
    asm("v_cndmask_b32_e64 %0, %3, %2, %6\n\tv_cndmask_b32_e64 %1, %5, %4, %6"
        : "=&v"(lo), "=&v"(hi)
        : "v"((u32)yes), "v"((u32)no), "v"((u32)(yes >> 32)), "v"((u32)(no >> 32)), "s"(m));
    return ((u64)hi << 32) | lo;
}
// a * b + c mod p as SOME u64 (a, b, c arbitrary u64), 11 long + ~6 short issue slots (the textbook product followed by
// reduce128 compiles to 15 + 6 without the addend):
//   P = a0 b0 + c.lo;  Y = a0 b1 + P.hi + c.hi;  Y = a1 b0 + Y (carry k);  H = a1 b1 + Y.hi
//        exact: lo = (P.lo, Y.lo), hi = H + k 2^32; no intermediate can exceed 64 bits ((2^32-1)^2 + 2 (2^32-1) = 2^64 - 1)
//   R = lo + H.lo (2^32 - 1)   (carry C)       -- 2^64 = 2^32 - 1: one mad
//   R = R - H.hi - k           (borrow B)      -- 2^96 = -1; k rides in as the borrow-in
//   R += (C - B)(2^32 - 1)                     -- red_fix
// A_UNIFORM: `a` is a uniform table constant (its halves stay in SGPRs; see mad_co_k)
template <bool HAS_ADDEND, bool A_UNIFORM = false, bool BRANCH = true>
GL_D u64 mulr_add_dev(u64 a, u64 b, u64 c) {
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u64 P = HAS_ADDEND ? (u64)a0 * b0 + (u32)c : (u64)a0 * b0;
    u64 Y = (u64)a0 * b1 + (P >> 32);
    if (HAS_ADDEND) Y = add_u32(Y, (u32)(c >> 32));
    sg k, C, b1_, B;
    Y = A_UNIFORM ? mad_co_k(a1, b0, Y, k) : mad_co(a1, b0, Y, k);
    const u64 H = (u64)a1 * b1 + (Y >> 32);
    const u64 lo = (Y << 32) | (u32)P;
    const u64 R = mad_eps_co((u32)H, lo, C);
    const u32 r0 = subb_co_settled((u32)R, (u32)(H >> 32), k, b1_);
    const u32 r1 = subb0_co((u32)(R >> 32), b1_, B);
    return red_fix<BRANCH>(((u64)r1 << 32) | r0, C, B);
}
// hi * 2^64 + lo -> some u64 congruent to it: the same reduction for a 128-bit value that is already there (Acc::reduce)
GL_D u64 red128_dev(u64 hi, u64 lo) {
    sg C, b1_, B;
    const u64 R = mad_eps_co((u32)hi, lo, C);
    const u32 r0 = sub_co((u32)R, (u32)(hi >> 32), b1_);
    const u32 r1 = subb0_co((u32)(R >> 32), b1_, B);
    return red_fix(((u64)r1 << 32) | r0, C, B);
}
// any u64 -> the canonical representative: r - p = r + (2^32 - 1) (mod 2^64), and that add carries exactly when r >= p
GL_D u64 canon_dev(u64 r) {
    sg c;
    const u64 t = add_eps_co(r, c);
    return pick64(c, t, r);
}
#endif

GL_HD u64 add(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    // s = a + b (carry c1); t = s - p = s + (2^32 - 1) (carry c2 <=> s >= p); the sum needs the subtraction iff c1 | c2.
    // 5 long issue slots; the compare-and-select form compiles to 7.
    sg k, c1, c2;
    const u32 s0 = add_co((u32)a, (u32)b, k);
    const u32 s1 = addc_co((u32)(a >> 32), (u32)(b >> 32), k, c1);
    const u64 t = add_eps_co(((u64)s1 << 32) | s0, c2);
    const sg m = c1 | c2;  // a uniform 64-bit OR: the compiler emits s_or_b64 (an asm s_or_b64 would clobber SCC behind its back)
    return pick64_salu(m, t, ((u64)s1 << 32) | s0);
#else
    u64 s = a + b;
    // a,b < p so a+b < 2p < 2^65; overflow or s>=p => subtract p once.
    if (s < a || s >= P) s -= P;
    return s;
#endif
}
GL_HD u64 add_ref(u64 a, u64 b) {  // textbook form (self-tests)
    u64 s = a + b;
    if (s < a || s >= P) s -= P;
    return s;
}
GL_HD u64 sub(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    // a - b, and on a borrow - (2^32 - 1) = + {1, 0xFFFFFFFF} (the wrapped difference is 2^64 too large): 4 long slots
    sg b0, B;
    const u32 d0 = sub_co((u32)a, (u32)b, b0);
    const u32 d1 = subb_co((u32)(a >> 32), (u32)(b >> 32), b0, B);
    const u32 e = ones_where(B);
    u64 fix = ((u64)e << 32) | (0u - e);
    asm("" : "+v"(fix));
    return (((u64)d1 << 32) | d0) + fix;
#else
    return a >= b ? a - b : a + (P - b);
#endif
}
GL_HD u64 sub_ref(u64 a, u64 b) { return a >= b ? a - b : a + (P - b); }  // textbook form (self-tests)
GL_HD u64 neg(u64 a) { return a ? P - a : 0; }
GL_HD u64 dbl(u64 a) { return add(a, a); }

GL_HD void mul64(u64 a, u64 b, u64& hi, u64& lo) {
#if defined(__HIP_DEVICE_COMPILE__)
    lo = a * b;
    hi = __umul64hi(a, b);
#else
    unsigned __int128 m = (unsigned __int128)a * b;
    lo = (u64)m;
    hi = (u64)(m >> 64);
#endif
}

// Reduce hi*2^64 + lo mod p to canonical form, using 2^64 = 2^32 - 1 and 2^96 = -1 (mod p).
GL_HD u64 reduce128(u64 hi, u64 lo) {
    u64 hi_hi = hi >> 32, hi_lo = hi & EPS;
    u64 t0 = lo - hi_hi;
    if (lo < hi_hi) t0 -= EPS;  // borrow: add p == subtract 2^32-1 (mod 2^64)
    u64 t1 = hi_lo * EPS;       // < 2^64
    u64 r = t0 + t1;
    if (r < t1) r += EPS;  // carry: 2^64 = EPS mod p
    if (r >= P) r -= P;
    return r;
}
GL_HD u64 mul_ref(u64 a, u64 b) {  // textbook form: the host path, and what the device self-test checks mul against
    u64 hi, lo;
    mul64(a, b, hi, lo);
    return reduce128(hi, lo);
}
GL_HD u64 mul(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return canon_dev(mulr_add_dev<false>(a, b, 0));
#else
    return mul_ref(a, b);
#endif
}
// the NTT kernels' multiply (see red_fix)
GL_HD u64 mul_nb(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return canon_dev(mulr_add_dev<false, false, false>(a, b, 0));
#else
    return mul_ref(a, b);
#endif
}
GL_HD u64 sqr(u64 a) { return mul(a, a); }
// a*b + c
GL_HD u64 mul_add(u64 a, u64 b, u64 c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return canon_dev(mulr_add_dev<true>(a, b, c));
#else
    return add(mul(a, b), c);
#endif
}

// a*b (+ c) as SOME u64 representative, not the canonical one: for a value whose next use is as an operand of a multiplication
// or as the addend of a mul_add (both take arbitrary u64 operands) -- saves the canonicalisation and, with the addend riding
// on the mads, the separate addition (11 long + 6 short slots against 15 + 5 for mul then add)
GL_HD u64 mul_add_nc(u64 a, u64 b, u64 c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return mulr_add_dev<true>(a, b, c);
#else
    return add(mul(a, b), c);
#endif
}
GL_HD u64 mul_nc(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return mulr_add_dev<false>(a, b, 0);
#else
    return mul(a, b);
#endif
}

GL_HD u64 pow(u64 b, u64 e) {
    u64 r = 1;
    while (e) {
        if (e & 1) r = mul(r, b);
        b = sqr(b);
        e >>= 1;
    }
    return r;
}
GL_HD u64 exp_pow2(u64 b, int k) {
    for (int i = 0; i < k; i++) b = sqr(b);
    return b;
}
// Fermat inverse (0 -> 0).
GL_HD u64 inv(u64 a) { return pow(a, P - 2); }

// primitive 2^k-th root of unity, as plonky2 `primitive_root_of_unity(k)`
GL_HD u64 root_of_unity(int k) { return exp_pow2(POW2_GEN, 32 - k); }

GL_HD u32 bitrev(u32 x, int bits) {
    u32 r = 0;
    for (int i = 0; i < bits; i++) {
        r = (r << 1) | (x & 1);
        x >>= 1;
    }
    return r;
}

// ---------------------------------------------------------------- GF(p^2)
struct E2 {
    u64 a, b;  // a + b*x
};
GL_HD E2 e2(u64 a, u64 b = 0) {
    E2 r;
    r.a = a;
    r.b = b;
    return r;
}
GL_HD E2 add(E2 x, E2 y) { return e2(add(x.a, y.a), add(x.b, y.b)); }
GL_HD E2 sub(E2 x, E2 y) { return e2(sub(x.a, y.a), sub(x.b, y.b)); }
GL_HD E2 mul(E2 x, E2 y) {
    // (a0 + a1 x)(b0 + b1 x) = a0b0 + 7 a1b1 + (a0b1 + a1b0) x
    u64 a0b0 = mul(x.a, y.a), a1b1 = mul(x.b, y.b);
    u64 c0 = add(a0b0, mul(a1b1, W_EXT));
    u64 c1 = add(mul(x.a, y.b), mul(x.b, y.a));
    return e2(c0, c1);
}
GL_HD E2 mul(E2 x, u64 s) { return e2(mul(x.a, s), mul(x.b, s)); }
GL_HD E2 sqr(E2 x) { return mul(x, x); }
GL_HD bool eq(E2 x, E2 y) { return x.a == y.a && x.b == y.b; }
GL_HD E2 inv(E2 x) {
    // 1/(a + b x) = (a - b x) / (a^2 - 7 b^2)
    u64 d = sub(sqr(x.a), mul(W_EXT, sqr(x.b)));
    u64 di = inv(d);
    return e2(mul(x.a, di), mul(neg(x.b), di));
}
GL_HD E2 pow(E2 b, u64 e) {
    E2 r = e2(1, 0);
    while (e) {
        if (e & 1) r = mul(r, b);
        b = sqr(b);
        e >>= 1;
    }
    return r;
}
GL_HD E2 exp_pow2(E2 b, int k) {
    for (int i = 0; i < k; i++) b = sqr(b);
    return b;
}

// ---------------------------------------------------------------- Poseidon-12
static const u64 H_POSEIDON_RC[360] = {
#include "poseidon_rc.inc"
};
#if defined(__HIPCC__)
__device__ __constant__ static const u64 D_POSEIDON_RC[360] = {
#include "poseidon_rc.inc"
};
#endif
GL_HD u64 poseidon_rc(int i) {
#if defined(__HIP_DEVICE_COMPILE__)
    return D_POSEIDON_RC[i];
#else
    return H_POSEIDON_RC[i];
#endif
}

static const int SPONGE_WIDTH = 12;
static const int SPONGE_RATE = 8;

GL_HD u64 sbox7(u64 x) {
    u64 x2 = sqr(x), x4 = sqr(x2), x3 = mul(x2, x);
    return mul(x4, x3);
}

// MDS: out[r] = sum_i circ[i] * s[(i+r)%12] + diag[r]*s[r], circ = [17,15,41,16,2,28,13,13,39,18,34,20], diag=[8,0..]
// Evaluated on 32-bit halves so every accumulator fits in 64 bits (coefficients < 2^6, 12 terms).
GL_HD void mds_layer(u64* s) {
    const u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    u64 lo[12], hi[12];
#pragma unroll
    for (int i = 0; i < 12; i++) {
        lo[i] = s[i] & EPS;
        hi[i] = s[i] >> 32;
    }
    u64 out[12];
#pragma unroll
    for (int r = 0; r < 12; r++) {
        u64 al = 0, ah = 0;
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
        // value = al + ah*2^32, al,ah < 2^42.  hi word = ah>>32, lo word = al + (ah<<32) with carry.
        u64 l = al + (ah << 32);
        u64 h = (ah >> 32) + (l < al ? 1 : 0);
        out[r] = reduce128(h, l);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = out[i];
}

// Full permutation, "naive" schedule: 4 full + 22 partial + 4 full rounds; each round = add constants,
// S-box x^7 (all lanes / lane 0), MDS.  plonky2's optimised partial rounds compute the same function.
GL_HD void poseidon(u64* s) {
    int rc = 0;
    for (int r = 0; r < 30; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = add(s[i], poseidon_rc(rc + i));
        rc += 12;
        if (r < 4 || r >= 26) {
#pragma unroll
            for (int i = 0; i < 12; i++) s[i] = sbox7(s[i]);
        } else {
            s[0] = sbox7(s[0]);
        }
        mds_layer(s);
    }
}

// two_to_one(l, r) = permute(l || r || 0^4)[0..4]   (plonky2 hash::hashing::compress)
GL_HD void two_to_one(const u64* l, const u64* r, u64* out) {
    u64 s[12];
    for (int i = 0; i < 4; i++) {
        s[i] = l[i];
        s[4 + i] = r[i];
        s[8 + i] = 0;
    }
    poseidon(s);
    for (int i = 0; i < 4; i++) out[i] = s[i];
}

}  // namespace gl
