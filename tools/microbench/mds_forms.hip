// Round 3, VERDICT item 8: MEASURE the formulations of Poseidon's full-round MDS layer that DESIGN.md section 6 had only priced
// on paper.  out[r] = rc[r] + sum_i circ[i] * s[(i + r) % 12] + 8 s[0] (r == 0), state words arbitrary u64, result "some u64".
//
//   form 0  shipped (glf::mds_full): 32-bit halves, 24 (26) v_mad_u64_u32 per output word with inline-constant coefficients,
//           one mad to fold the 75-bit sum                                                      ~ 29 long slots per word
//   form 1  byte planes + v_dot4_u32_u8: the state as 8 planes of bytes (4 words per dword, v_perm_b32 transposes), per
//           (row, plane) three dot4 against the row's coefficients packed four to a dword, planes recombined with
//           v_mad_u64_u32 by 2^8, 2^16, 2^24 and folded as in form 0
//   form 2  byte planes + v_mfma_i32_4x4x4_16b_i8: the same planes as the B operand (a lane's own four words of one plane),
//           the coefficient rows as the A operand (constant per lane mod 4), D = 4 output rows of the lane's OWN state:
//           no cross-lane movement at all, 72 MFMAs per layer, the products leave the VALU entirely (the matrix cores take
//           SIGNED bytes: the planes are offset by 128 with one XOR per dword and the accumulators start from 128 x row sum)
//
// Every form is checked against form 0 (canonicalised, 2^20 random states incl. all-ones words) before anything is timed.
// Timing: in-kernel s_memtime / s_memrealtime around >= 20 ms of work, 256 CUs x W blocks of 256 threads (W = 4, 5 waves per
// SIMD), (a) the MDS layer alone, (b) a whole full round (twelve x^7 S-boxes + the layer): what the hash kernels run.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o mds_forms mds_forms.hip && ./mds_forms
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "../../plonky2-aes_amd/csrc/gl.h"
#include "../../plonky2-aes_amd/csrc/poseidon_fast.h"
typedef gl::u64 u64;
typedef gl::u32 u32;

#define CHK(x)                                                      \
    do {                                                            \
        hipError_t e_ = (x);                                        \
        if (e_ != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            exit(1);                                                \
        }                                                           \
    } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __constant__ static const unsigned char CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
__device__ __forceinline__ u32 mds_coeff(int r, int c) { return CIRC[(c - r + 12) % 12] + ((r == 0 && c == 0) ? 8 : 0); }

// 4x4 byte transposes: plane[p][q] = byte p of the words 4q .. 4q+3 (p < 4 from the low halves, p >= 4 from the high halves)
__device__ __forceinline__ void byte_planes(const u64* s, u32 (*pl)[3]) {
#pragma unroll
    for (int q = 0; q < 3; q++) {
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const u32 x0 = (u32)(s[4 * q + 0] >> (32 * half)), x1 = (u32)(s[4 * q + 1] >> (32 * half)), x2 = (u32)(s[4 * q + 2] >> (32 * half)),
                      x3 = (u32)(s[4 * q + 3] >> (32 * half));
            // v_perm_b32 D, S0, S1, sel: bytes 0-3 = S1, 4-7 = S0
            const u32 t0 = __builtin_amdgcn_perm(x1, x0, 0x05010400u), t1 = __builtin_amdgcn_perm(x1, x0, 0x07030602u);
            const u32 u0 = __builtin_amdgcn_perm(x3, x2, 0x05010400u), u1 = __builtin_amdgcn_perm(x3, x2, 0x07030602u);
            pl[4 * half + 0][q] = __builtin_amdgcn_perm(u0, t0, 0x05040100u);
            pl[4 * half + 1][q] = __builtin_amdgcn_perm(u0, t0, 0x07060302u);
            pl[4 * half + 2][q] = __builtin_amdgcn_perm(u1, t1, 0x05040100u);
            pl[4 * half + 3][q] = __builtin_amdgcn_perm(u1, t1, 0x07060302u);
        }
    }
}
// ---- form 1 / form 2 share the plane recombination: al = rc.lo + P0 + P1 2^8 + P2 2^16 + P3 2^24, ah likewise from P4..P7 and
// rc.hi -- four mads each (the multipliers 2^8, 2^16, 2^24 sit in SGPRs: VOP3 on gfx9 takes no literal) -- then the shipped fold
__device__ __forceinline__ u64 planes_to_word(u32 p0, u32 p1, u32 p2, u32 p3, u32 p4, u32 p5, u32 p6, u32 p7, u64 rc) {
#if !defined(__HIP_DEVICE_COMPILE__)
    return 0;  // (the host pass only needs the kernels to parse)
#else
    gl::sg d;
    u64 al = gl::mad_co_k(1u << 24, p3, (u64)(u32)rc, d);
    al = gl::mad_co_k(1u << 16, p2, al, d);
    al = gl::mad_co_k(1u << 8, p1, al, d);
    al = gl::add_u32(al, p0);
    u64 ah = gl::mad_co_k(1u << 24, p7, rc >> 32, d);
    ah = gl::mad_co_k(1u << 16, p6, ah, d);
    ah = gl::mad_co_k(1u << 8, p5, ah, d);
    ah = gl::add_u32(ah, p4);
    return glf::fold_al_ah(al, ah);  // al, ah < 2^43 as in the shipped form
#endif
}

__device__ __forceinline__ void mds_dot4(u64* s, const unsigned long long* rc) {
    u32 pl[8][3];
    byte_planes(s, pl);
    u64 res[12];
#pragma unroll
    for (int r = 0; r < 12; r++) {
        u32 cf[3];
#pragma unroll
        for (int q = 0; q < 3; q++) cf[q] = mds_coeff(r, 4 * q) | (mds_coeff(r, 4 * q + 1) << 8) | (mds_coeff(r, 4 * q + 2) << 16) | (mds_coeff(r, 4 * q + 3) << 24);
        u32 P[8];
#pragma unroll
        for (int p = 0; p < 8; p++) {
            u32 acc = __builtin_amdgcn_udot4(pl[p][0], cf[0], 0u, false);
            acc = __builtin_amdgcn_udot4(pl[p][1], cf[1], acc, false);
            P[p] = __builtin_amdgcn_udot4(pl[p][2], cf[2], acc, false);
        }
        res[r] = planes_to_word(P[0], P[1], P[2], P[3], P[4], P[5], P[6], P[7], rc ? rc[r] : 0);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = res[i];
}

// A operand of v_mfma_i32_4x4x4_16b_i8 for (row group g, column group q): lane 4b + i holds M[4g + i][4q .. 4q + 3]
struct MfmaCoeffs {
    int a[3][3];
    __device__ __forceinline__ void init() {
        const int i = threadIdx.x & 3;
#pragma unroll
        for (int g = 0; g < 3; g++)
#pragma unroll
            for (int q = 0; q < 3; q++)
                a[g][q] = (int)(mds_coeff(4 * g + i, 4 * q) | (mds_coeff(4 * g + i, 4 * q + 1) << 8) | (mds_coeff(4 * g + i, 4 * q + 2) << 16) | (mds_coeff(4 * g + i, 4 * q + 3) << 24));
    }
};
__device__ __forceinline__ void mds_mfma(u64* s, const unsigned long long* rc, const MfmaCoeffs& M) {
    u32 pl[8][3];
    byte_planes(s, pl);
    // the matrix cores multiply SIGNED bytes: feed b - 128 (one XOR per packed dword) and start every accumulator from
    // 128 * (sum of the row's coefficients) -- 256 for every row of the circulant, 264 for row 0 with its extra 8 s[0]
#pragma unroll
    for (int p = 0; p < 8; p++)
#pragma unroll
        for (int q = 0; q < 3; q++) pl[p][q] ^= 0x80808080u;
    u64 res[12];
#pragma unroll
    for (int g = 0; g < 3; g++) {
        v4i D[8];
#pragma unroll
        for (int p = 0; p < 8; p++) {
            v4i acc = {g == 0 ? 128 * 264 : 128 * 256, 128 * 256, 128 * 256, 128 * 256};
            acc = __builtin_amdgcn_mfma_i32_4x4x4i8(M.a[g][0], (int)pl[p][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_i32_4x4x4i8(M.a[g][1], (int)pl[p][1], acc, 0, 0, 0);
            D[p] = __builtin_amdgcn_mfma_i32_4x4x4i8(M.a[g][2], (int)pl[p][2], acc, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
            res[4 * g + i] = planes_to_word((u32)D[0][i], (u32)D[1][i], (u32)D[2][i], (u32)D[3][i], (u32)D[4][i], (u32)D[5][i], (u32)D[6][i], (u32)D[7][i], rc ? rc[4 * g + i] : 0);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = res[i];
}

struct Stamp {
    uint64_t t0, t1, r0, r1;
};
#define WAVES __attribute__((amdgpu_waves_per_eu(W, W)))
template <int FORM, bool SBOX, int W>
__global__ __launch_bounds__(256) WAVES void k(const u64* in, u64* out, Stamp* stamps, int trips) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, n = (size_t)gridDim.x * blockDim.x;
    u64 s[12];
#pragma unroll
    for (int c = 0; c < 12; c++) s[c] = in[(size_t)c * n + i];
    MfmaCoeffs M;
    if (FORM == 2) M.init();
    const unsigned long long* rc = (const unsigned long long*)gl::D_POSEIDON_RC;  // any 12 uniform 64-bit constants per layer
    const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int t = 0; t < trips; t++) {
        if (SBOX) {
#pragma unroll
            for (int c = 0; c < 12; c++) s[c] = glf::sbox7(s[c]);
        }
        if (FORM == 0) glf::mds_full(s, rc + 12 * (t & 7));
        if (FORM == 1) mds_dot4(s, rc + 12 * (t & 7));
        if (FORM == 2) mds_mfma(s, rc + 12 * (t & 7), M);
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) stamps[i >> 6] = Stamp{t0, t1, r0, r1};
#pragma unroll
    for (int c = 0; c < 12; c++) out[(size_t)c * n + i] = glf::canon(s[c]);
}

template <int FORM, bool SBOX, int W>
static double run(const u64* d_in, u64* d_out, Stamp* d_st, size_t blocks, int trips, double* clock_ghz, double* cyc_per_layer) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int it = 0; it < 3; it++) {
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<FORM, SBOX, W>), dim3((unsigned)blocks), dim3(256), 0, 0, d_in, d_out, d_st, trips);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    std::vector<Stamp> st(blocks * 4);
    CHK(hipMemcpy(st.data(), d_st, st.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
    std::vector<double> clk, cyc;
    for (auto& x : st) {
        clk.push_back((double)(x.t1 - x.t0) / (double)(x.r1 - x.r0) * 0.1);  // GHz: memrealtime ticks at 100 MHz
        cyc.push_back((double)(x.t1 - x.t0) / trips / W);                   // SIMD-cycles per wave-layer with W waves sharing the SIMD
    }
    std::sort(clk.begin(), clk.end());
    std::sort(cyc.begin(), cyc.end());
    *clock_ghz = clk[clk.size() / 2];
    *cyc_per_layer = cyc[cyc.size() / 2];
    return best;
}

int main() {
    const size_t CUS = 256;
    u64 x = 88172645463325252ull;
    // ---- correctness: one layer of every form on the same 2^20 states
    {
        const size_t blocks = 4096, n = blocks * 256;
        std::vector<u64> h(12 * n);
        for (auto& v : h) {
            x ^= x << 13, x ^= x >> 7, x ^= x << 17;
            v = x;  // arbitrary u64 representatives, as inside a permutation
        }
        for (int c = 0; c < 12; c++) h[(size_t)c * n] = ~0ull, h[(size_t)c * n + 1] = 0, h[(size_t)c * n + 2] = 0xFFFFFFFF00000000ull;
        u64 *d_in, *d_o[3];
        Stamp* d_st;
        CHK(hipMalloc(&d_in, 96 * n));
        CHK(hipMalloc(&d_st, blocks * 4 * sizeof(Stamp)));
        CHK(hipMemcpy(d_in, h.data(), 96 * n, hipMemcpyHostToDevice));
        for (int f = 0; f < 3; f++) CHK(hipMalloc(&d_o[f], 96 * n));
        hipLaunchKernelGGL((k<0, false, 4>), dim3(blocks), dim3(256), 0, 0, d_in, d_o[0], d_st, 1);
        hipLaunchKernelGGL((k<1, false, 4>), dim3(blocks), dim3(256), 0, 0, d_in, d_o[1], d_st, 1);
        hipLaunchKernelGGL((k<2, false, 4>), dim3(blocks), dim3(256), 0, 0, d_in, d_o[2], d_st, 1);
        CHK(hipDeviceSynchronize());
        std::vector<u64> o[3];
        for (int f = 0; f < 3; f++) {
            o[f].resize(12 * n);
            CHK(hipMemcpy(o[f].data(), d_o[f], 96 * n, hipMemcpyDeviceToHost));
        }
        size_t bad1 = 0, bad2 = 0;
        for (size_t i = 0; i < 12 * n; i++) bad1 += o[0][i] != o[1][i], bad2 += o[0][i] != o[2][i];
        printf("check on %zu states: dot4 form %zu mismatching words, mfma form %zu mismatching words (vs the shipped form, canonicalised)\n", n, bad1, bad2);
        if (bad1 || bad2) return 2;
        CHK(hipFree(d_in));
        CHK(hipFree(d_st));
        for (int f = 0; f < 3; f++) CHK(hipFree(d_o[f]));
    }
    const char* names[3] = {"shipped: 24 mads per word on 32-bit halves", "byte planes + v_dot4_u32_u8", "byte planes + v_mfma_i32_4x4x4_16b_i8"};
    for (int W : {4, 5}) {
        const size_t blocks = CUS * W, n = blocks * 256;
        std::vector<u64> h(12 * n);
        for (auto& v : h) {
            x ^= x << 13, x ^= x >> 7, x ^= x << 17;
            v = x;
        }
        u64 *d_in, *d_out;
        Stamp* d_st;
        CHK(hipMalloc(&d_in, 96 * n));
        CHK(hipMalloc(&d_out, 96 * n));
        CHK(hipMalloc(&d_st, blocks * 4 * sizeof(Stamp)));
        CHK(hipMemcpy(d_in, h.data(), 96 * n, hipMemcpyHostToDevice));
        for (int sb = 0; sb < 2; sb++) {
            const int trips = sb ? 6000 : 30000;
            printf("\n%d waves per SIMD, %s, %d layers per thread:\n", W, sb ? "whole full round (12 S-boxes + MDS layer)" : "MDS layer alone", trips);
            double ms[3], clk[3], cyc[3];
#define RUN(F)                                                                                          \
    ms[F] = W == 4 ? (sb ? run<F, true, 4>(d_in, d_out, d_st, blocks, trips, &clk[F], &cyc[F])            \
                         : run<F, false, 4>(d_in, d_out, d_st, blocks, trips, &clk[F], &cyc[F]))          \
                   : (sb ? run<F, true, 5>(d_in, d_out, d_st, blocks, trips, &clk[F], &cyc[F])            \
                         : run<F, false, 5>(d_in, d_out, d_st, blocks, trips, &clk[F], &cyc[F]))
            RUN(0);
            RUN(1);
            RUN(2);
            for (int f = 0; f < 3; f++)
                printf("  form %d  %-46s %8.2f ms  %7.1f SIMD-cycles per wave-%s  clock %.2f GHz  %.3f x shipped\n", f, names[f], ms[f], cyc[f], sb ? "round" : "layer", clk[f],
                       ms[f] / ms[0]);
        }
        CHK(hipFree(d_in));
        CHK(hipFree(d_out));
        CHK(hipFree(d_st));
    }
    return 0;
}
