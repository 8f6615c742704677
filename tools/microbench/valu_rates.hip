// VALU issue-cost microbenchmark for gfx950 (MI355X): what one SIMD issues per cycle for the integer instructions the
// Goldilocks / Poseidon code is made of.  Replaces int_rates.hip (round 1), whose 0.3 ms launches never let the clock
// settle and which priced everything at an ASSUMED 2.4 GHz.
//
// Method: every wave stamps s_memtime (shader-clock cycles) and s_memrealtime (constant 100 MHz) around its loop, so
//   * the shader clock actually held under this load = d(memtime) / d(memrealtime) * 100 MHz          (printed per run)
//   * cycles per wave-instruction per SIMD = d(memtime) * waves_on_that_SIMD ... measured per wave as
//         d(memtime) / (instructions the wave issued)  /  (waves per SIMD)                            (clock independent)
// Each configuration runs >= 50 ms (the loop count is calibrated), with 1, 2, 4 and 8 waves per SIMD (256 CUs x w blocks
// of 256 threads; one block = one wave per SIMD).  Instructions are independent chains (8 accumulators), emitted as
// inline asm so that the measured stream is exactly the named instruction plus the loop's s_add/s_cmp/s_cbranch.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#define CHK(x)                                                                 \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));            \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

struct Stamp {
    uint64_t t0, t1, r0, r1;
};

// 8 independent accumulators
#define REP4(X) X X X X X X X X X X X X X X X X  // 16 x 8 = 128 instructions per loop trip: the loop's own s_add / s_cmp / s_cbranch stay under 3 %
enum Op {
    ADD_E32, ADD_E64, MOV, ADDCO_CHAIN, ADDCO_SPACED, MAD64, MAD64_CARRY, MUL_LO, MUL_HI, LSHL_ADD_U64, MAD_U32_U24, CNDMASK_VCC, ADD3, ALIGNBIT, SNOP0, SNOP1,
    MIX_MAD_ADD, XOR3, CNDMASK_SGPR, CMP_CNDMASK, SUBB_CHAIN, MAD_EPS_INLINE, MAD_NOP0, MAD_NOP1, MAD_SALU, NUM_OPS
};
static const char* OP_NAME[NUM_OPS] = {"v_add_u32 (e32)", "v_add_u32 (e64, 2 SGPR-free)", "v_mov_b32", "v_add_co/v_addc_co back-to-back (vcc)",
                                       "v_add_co x4 then v_addc_co x4 (4 sgpr pairs)", "v_mad_u64_u32 (no carry use)", "v_mad_u64_u32 + v_addc_co on its carry (2 apart)",
                                       "v_mul_lo_u32", "v_mul_hi_u32", "v_lshl_add_u64", "v_mad_u32_u24", "v_cndmask_b32 (vcc)", "v_add3_u32", "v_alignbit_b32",
                                       "s_nop 0", "s_nop 1", "mad64 : add_e32 = 1 : 1 interleaved", "v_lshl_add_u32",
                                       "v_cndmask_b32_e64 (sgpr pair written before the loop)", "v_cmp_lt_u32 x4 then v_cndmask x4 (4 sgpr pairs)", "v_sub_co x4 then v_subb_co x4 (4 sgpr pairs)", "v_mad_u64_u32 v, s, v, -1, v (inline constant)",
                                       "v_mad_u64_u32 ; s_nop 0 alternating (per mad)", "v_mad_u64_u32 ; s_nop 1 alternating (per mad)", "v_mad_u64_u32 ; s_andn2_b64 alternating (per mad)"};
static const int OP_INSTR_PER_TRIP[NUM_OPS] = {128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128};

template <int OP>
__global__ __launch_bounds__(256) void k(Stamp* stamps, uint32_t* sink, uint32_t seed, int trips) {
    uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;
    uint32_t a0 = t + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 * 11 + 4, a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;
    uint64_t q0 = a0, q1 = a1, q2 = a2, q3 = a3, q4 = a4, q5 = a5, q6 = a6, q7 = a7;
    uint32_t m = seed * 2654435761u + t, c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    unsigned long long s0, s1, s2, s3, mask = __ballot((t * 2654435761u) >> 31);
    if (OP == CNDMASK_VCC) asm volatile("v_cmp_lt_u32_e32 vcc, %0, %1" ::"v"(a0), "v"(m) : "vcc");
    uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < trips; i++) {
        if (OP == ADD_E32) {
            REP4(asm volatile("v_add_u32_e32 %0, %8, %0\n v_add_u32_e32 %1, %8, %1\n v_add_u32_e32 %2, %8, %2\n v_add_u32_e32 %3, %8, %3\n"
                              "v_add_u32_e32 %4, %8, %4\n v_add_u32_e32 %5, %8, %5\n v_add_u32_e32 %6, %8, %6\n v_add_u32_e32 %7, %8, %7"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
        } else if (OP == ADD_E64) {
            REP4(asm volatile("v_add_u32_e64 %0, %8, %0\n v_add_u32_e64 %1, %8, %1\n v_add_u32_e64 %2, %8, %2\n v_add_u32_e64 %3, %8, %3\n"
                              "v_add_u32_e64 %4, %8, %4\n v_add_u32_e64 %5, %8, %5\n v_add_u32_e64 %6, %8, %6\n v_add_u32_e64 %7, %8, %7"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
        } else if (OP == MOV) {
            REP4(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n"
                              "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %8"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
        } else if (OP == ADDCO_CHAIN) {
            // the compiler's 64-bit add: carry produced and consumed by adjacent instructions (hardware interlocks or the
            // assembler's required s_nop? -- none is inserted here, the hazard is handled by the hardware on gfx950 for VCC)
            REP4(asm volatile("v_add_co_u32_e32 %0, vcc, %8, %0\n v_addc_co_u32_e32 %1, vcc, %8, %1, vcc\n v_add_co_u32_e32 %2, vcc, %8, %2\n v_addc_co_u32_e32 %3, vcc, %8, %3, vcc\n"
                              "v_add_co_u32_e32 %4, vcc, %8, %4\n v_addc_co_u32_e32 %5, vcc, %8, %5, vcc\n v_add_co_u32_e32 %6, vcc, %8, %6\n v_addc_co_u32_e32 %7, vcc, %8, %7, vcc"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m) : "vcc");)
        } else if (OP == ADDCO_SPACED) {
            REP4(asm volatile("v_add_co_u32_e64 %0, %8, %12, %0\n v_add_co_u32_e64 %2, %9, %12, %2\n v_add_co_u32_e64 %4, %10, %12, %4\n v_add_co_u32_e64 %6, %11, %12, %6\n"
                              "v_addc_co_u32_e64 %1, %8, %12, %1, %8\n v_addc_co_u32_e64 %3, %9, %12, %3, %9\n v_addc_co_u32_e64 %5, %10, %12, %5, %10\n v_addc_co_u32_e64 %7, %11, %12, %7, %11"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) , "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3) : "v"(m));)
        } else if (OP == MAD64) {
            REP4(asm volatile("v_mad_u64_u32 %0, %8, %12, %12, %0\n v_mad_u64_u32 %1, %9, %12, %12, %1\n v_mad_u64_u32 %2, %10, %12, %12, %2\n v_mad_u64_u32 %3, %11, %12, %12, %3\n"
                              "v_mad_u64_u32 %4, %8, %12, %12, %4\n v_mad_u64_u32 %5, %9, %12, %12, %5\n v_mad_u64_u32 %6, %10, %12, %12, %6\n v_mad_u64_u32 %7, %11, %12, %12, %7"
                              : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7) , "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3) : "v"(m));)
        } else if (OP == MAD64_CARRY) {
            // the Acc::fma pattern: 4 mads, each carry counted two or more issue slots after it was produced
            REP4(asm volatile("v_mad_u64_u32 %0, %8, %12, %12, %0\n v_mad_u64_u32 %1, %9, %12, %12, %1\n v_mad_u64_u32 %2, %10, %12, %12, %2\n v_addc_co_u32_e64 %4, %8, 0, %4, %8\n"
                              "v_mad_u64_u32 %3, %11, %12, %12, %3\n v_addc_co_u32_e64 %5, %9, 0, %5, %9\n v_addc_co_u32_e64 %6, %10, 0, %6, %10\n v_addc_co_u32_e64 %7, %11, 0, %7, %11"
                              : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) , "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3) : "v"(m));)
        } else if (OP == MUL_LO) {
            REP4(asm volatile("v_mul_lo_u32 %0, %8, %0\n v_mul_lo_u32 %1, %8, %1\n v_mul_lo_u32 %2, %8, %2\n v_mul_lo_u32 %3, %8, %3\n"
                              "v_mul_lo_u32 %4, %8, %4\n v_mul_lo_u32 %5, %8, %5\n v_mul_lo_u32 %6, %8, %6\n v_mul_lo_u32 %7, %8, %7"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
        } else if (OP == MUL_HI) {
            REP4(asm volatile("v_mul_hi_u32 %0, %8, %0\n v_mul_hi_u32 %1, %8, %1\n v_mul_hi_u32 %2, %8, %2\n v_mul_hi_u32 %3, %8, %3\n"
                              "v_mul_hi_u32 %4, %8, %4\n v_mul_hi_u32 %5, %8, %5\n v_mul_hi_u32 %6, %8, %6\n v_mul_hi_u32 %7, %8, %7"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
        } else if (OP == LSHL_ADD_U64) {
            REP4(asm volatile("v_lshl_add_u64 %0, %0, 1, %7\n v_lshl_add_u64 %1, %1, 1, %7\n v_lshl_add_u64 %2, %2, 1, %7\n v_lshl_add_u64 %3, %3, 1, %7\n"
                              "v_lshl_add_u64 %4, %4, 1, %7\n v_lshl_add_u64 %5, %5, 1, %7\n v_lshl_add_u64 %6, %6, 1, %7\n v_lshl_add_u64 %0, %0, 2, %7"
                              : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6) : "v"(q7));)
        } else if (OP == MAD_U32_U24) {
            REP4(asm volatile("v_mad_u32_u24 %0, %8, %8, %0\n v_mad_u32_u24 %1, %8, %8, %1\n v_mad_u32_u24 %2, %8, %8, %2\n v_mad_u32_u24 %3, %8, %8, %3\n"
                              "v_mad_u32_u24 %4, %8, %8, %4\n v_mad_u32_u24 %5, %8, %8, %5\n v_mad_u32_u24 %6, %8, %8, %6\n v_mad_u32_u24 %7, %8, %8, %7"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
        } else if (OP == CNDMASK_VCC) {
            REP4(asm volatile("v_cndmask_b32_e32 %0, %8, %0, vcc\n v_cndmask_b32_e32 %1, %8, %1, vcc\n v_cndmask_b32_e32 %2, %8, %2, vcc\n v_cndmask_b32_e32 %3, %8, %3, vcc\n"
                              "v_cndmask_b32_e32 %4, %8, %4, vcc\n v_cndmask_b32_e32 %5, %8, %5, vcc\n v_cndmask_b32_e32 %6, %8, %6, vcc\n v_cndmask_b32_e32 %7, %8, %7, vcc"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m) : "vcc");)
        } else if (OP == ADD3) {
            REP4(asm volatile("v_add3_u32 %0, %8, %8, %0\n v_add3_u32 %1, %8, %8, %1\n v_add3_u32 %2, %8, %8, %2\n v_add3_u32 %3, %8, %8, %3\n"
                              "v_add3_u32 %4, %8, %8, %4\n v_add3_u32 %5, %8, %8, %5\n v_add3_u32 %6, %8, %8, %6\n v_add3_u32 %7, %8, %8, %7"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
        } else if (OP == ALIGNBIT) {
            REP4(asm volatile("v_alignbit_b32 %0, %8, %0, 7\n v_alignbit_b32 %1, %8, %1, 7\n v_alignbit_b32 %2, %8, %2, 7\n v_alignbit_b32 %3, %8, %3, 7\n"
                              "v_alignbit_b32 %4, %8, %4, 7\n v_alignbit_b32 %5, %8, %5, 7\n v_alignbit_b32 %6, %8, %6, 7\n v_alignbit_b32 %7, %8, %7, 7"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
        } else if (OP == SNOP0) {
            REP4(asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0");)
        } else if (OP == SNOP1) {
            REP4(asm volatile("s_nop 1\n s_nop 1\n s_nop 1\n s_nop 1\n s_nop 1\n s_nop 1\n s_nop 1\n s_nop 1");)
        } else if (OP == MIX_MAD_ADD) {
            REP4(asm volatile("v_mad_u64_u32 %0, %8, %12, %12, %0\n v_add_u32_e32 %4, %12, %4\n v_mad_u64_u32 %1, %9, %12, %12, %1\n v_add_u32_e32 %5, %12, %5\n"
                              "v_mad_u64_u32 %2, %10, %12, %12, %2\n v_add_u32_e32 %6, %12, %6\n v_mad_u64_u32 %3, %11, %12, %12, %3\n v_add_u32_e32 %7, %12, %7"
                              : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) , "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3) : "v"(m));)
        } else if (OP == XOR3) {
            REP4(asm volatile("v_lshl_add_u32 %0, %0, 3, %8\n v_lshl_add_u32 %1, %1, 3, %8\n v_lshl_add_u32 %2, %2, 3, %8\n v_lshl_add_u32 %3, %3, 3, %8\n"
                              "v_lshl_add_u32 %4, %4, 3, %8\n v_lshl_add_u32 %5, %5, 3, %8\n v_lshl_add_u32 %6, %6, 3, %8\n v_lshl_add_u32 %7, %7, 3, %8"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
        } else if (OP == CNDMASK_SGPR) {
            REP4(asm volatile("v_cndmask_b32_e64 %0, %8, %0, %9\n v_cndmask_b32_e64 %1, %8, %1, %9\n v_cndmask_b32_e64 %2, %8, %2, %9\n v_cndmask_b32_e64 %3, %8, %3, %9\n"
                              "v_cndmask_b32_e64 %4, %8, %4, %9\n v_cndmask_b32_e64 %5, %8, %5, %9\n v_cndmask_b32_e64 %6, %8, %6, %9\n v_cndmask_b32_e64 %7, %8, %7, %9"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "s"(mask));)
        } else if (OP == CMP_CNDMASK) {
            REP4(asm volatile("v_cmp_lt_u32_e64 %8, %12, %0\n v_cmp_lt_u32_e64 %9, %12, %1\n v_cmp_lt_u32_e64 %10, %12, %2\n v_cmp_lt_u32_e64 %11, %12, %3\n"
                              "v_cndmask_b32_e64 %4, %12, %4, %8\n v_cndmask_b32_e64 %5, %12, %5, %9\n v_cndmask_b32_e64 %6, %12, %6, %10\n v_cndmask_b32_e64 %7, %12, %7, %11"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3) : "v"(m));)
        } else if (OP == SUBB_CHAIN) {
            REP4(asm volatile("v_sub_co_u32_e64 %0, %8, %0, %12\n v_sub_co_u32_e64 %2, %9, %2, %12\n v_sub_co_u32_e64 %4, %10, %4, %12\n v_sub_co_u32_e64 %6, %11, %6, %12\n"
                              "v_subb_co_u32_e64 %1, %8, %1, %12, %8\n v_subb_co_u32_e64 %3, %9, %3, %12, %9\n v_subb_co_u32_e64 %5, %10, %5, %12, %10\n v_subb_co_u32_e64 %7, %11, %7, %12, %11"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3) : "v"(m));)
        } else if (OP == MAD_EPS_INLINE) {
            REP4(asm volatile("v_mad_u64_u32 %0, %8, %12, -1, %0\n v_mad_u64_u32 %1, %9, %12, -1, %1\n v_mad_u64_u32 %2, %10, %12, -1, %2\n v_mad_u64_u32 %3, %11, %12, -1, %3\n"
                              "v_mad_u64_u32 %4, %8, %12, -1, %4\n v_mad_u64_u32 %5, %9, %12, -1, %5\n v_mad_u64_u32 %6, %10, %12, -1, %6\n v_mad_u64_u32 %7, %11, %12, -1, %7"
                              : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7), "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3) : "v"(m));)
        } else if (OP == MAD_NOP0 || OP == MAD_NOP1 || OP == MAD_SALU) {
            // does a wait-state filler (or a scalar op) between two vector ops cost vector issue slots when other waves are ready?
#define MADX(F) "v_mad_u64_u32 %0, %8, %12, -1, %0\n" F "v_mad_u64_u32 %1, %9, %12, -1, %1\n" F "v_mad_u64_u32 %2, %10, %12, -1, %2\n" F "v_mad_u64_u32 %3, %11, %12, -1, %3\n" F \
                "v_mad_u64_u32 %4, %8, %12, -1, %4\n" F "v_mad_u64_u32 %5, %9, %12, -1, %5\n" F "v_mad_u64_u32 %6, %10, %12, -1, %6\n" F "v_mad_u64_u32 %7, %11, %12, -1, %7\n" F
            if (OP == MAD_NOP0) {
                REP4(asm volatile(MADX("s_nop 0\n") : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7), "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3) : "v"(m));)
            } else if (OP == MAD_NOP1) {
                REP4(asm volatile(MADX("s_nop 1\n") : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7), "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3) : "v"(m));)
            } else {
                REP4(asm volatile(MADX("s_andn2_b64 %13, %13, %13\n") : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7), "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3) : "v"(m), "s"(mask) : "scc");)
            }
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        Stamp s{t0, t1, r0, r1};
        stamps[t >> 6] = s;
    }
    sink[t] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint32_t)(q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7) ^ c0 ^ c1 ^ c2 ^ c3;
}

typedef void (*KernelFn)(Stamp*, uint32_t*, uint32_t, int);
template <int OP>
static KernelFn fn() { return k<OP>; }
template <int... I>
static void fill(KernelFn* t, std::integer_sequence<int, I...>) { ((t[I] = fn<I>()), ...); }

int main(int argc, char** argv) {
    const int NUM_CU = 256;
    KernelFn table[NUM_OPS];
    fill(table, std::make_integer_sequence<int, NUM_OPS>());
    Stamp* d_st;
    uint32_t* d_sink;
    CHK(hipMalloc(&d_st, sizeof(Stamp) * NUM_CU * 8 * 4));
    CHK(hipMalloc(&d_sink, 4 * NUM_CU * 8 * 256));
    std::vector<Stamp> h(NUM_CU * 8 * 4);
    double target_ms = argc > 1 ? atof(argv[1]) : 60.0;
    printf("%-52s %5s %9s %9s %11s %11s %12s\n", "instruction stream", "w/SIMD", "ms", "clock GHz", "cyc/instr", "wall cyc/i", "Tlane-op/s");
    const int first_op = argc > 2 ? atoi(argv[2]) : 0;
    for (int op = first_op; op < NUM_OPS; op++) {
        if (argc > 2 && op != 11 && op < 18) continue;
        for (int w : {1, 2, 4, 8}) {
            int blocks = NUM_CU * w, trips = 2000;
            hipEvent_t e0, e1;
            CHK(hipEventCreate(&e0));
            CHK(hipEventCreate(&e1));
            float ms = 0;
            for (int pass = 0; pass < 2; pass++) {  // pass 0 calibrates the trip count (and warms the clock), pass 1 is measured
                CHK(hipEventRecord(e0));
                hipLaunchKernelGGL(table[op], dim3(blocks), dim3(256), 0, 0, d_st, d_sink, 1u + pass, trips);
                CHK(hipEventRecord(e1));
                CHK(hipEventSynchronize(e1));
                CHK(hipEventElapsedTime(&ms, e0, e1));
                if (pass == 0) trips = (int)std::min(2e8, std::max(2000.0, trips * target_ms / std::max(ms, 1e-3f)));
            }
            CHK(hipMemcpy(h.data(), d_st, sizeof(Stamp) * blocks * 4, hipMemcpyDeviceToHost));
            std::vector<double> clk, cyc;
            for (int i = 0; i < blocks * 4; i++) {
                double dt = (double)(h[i].t1 - h[i].t0), dr = (double)(h[i].r1 - h[i].r0);
                clk.push_back(dt / dr * 0.1);  // GHz
                cyc.push_back(dt / ((double)trips * OP_INSTR_PER_TRIP[op]) / w);
            }
            std::sort(clk.begin(), clk.end());
            std::sort(cyc.begin(), cyc.end());
            double lane_ops = (double)blocks * 256 * trips * OP_INSTR_PER_TRIP[op];
            // SIMD-cycles per wave-instruction from the WALL clock (what the chip sustains, whatever the residency was)
            double wall_cyc = ms * 1e-3 * clk[clk.size() / 2] * 1e9 / ((double)trips * OP_INSTR_PER_TRIP[op] * w);
            printf("%-52s %5d %9.2f %9.3f %11.3f %11.3f %12.2f\n", OP_NAME[op], w, ms, clk[clk.size() / 2], cyc[cyc.size() / 2], wall_cyc, lane_ops / (ms * 1e-3) / 1e12);
            fflush(stdout);
        }
    }
    return 0;
}
