// Issue-rate microbenchmark for the integer instructions the Goldilocks/Poseidon code leans on (gfx950).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define ITERS 4096
template <int OP>
__global__ __launch_bounds__(256) void k(uint64_t* out, uint32_t seed) {
    uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;
    uint64_t a0 = t + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 * 11 + 4, a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;
    uint32_t m = seed * 2654435761u + t;
    for (int i = 0; i < ITERS; i++) {
        if (OP == 0) {  // v_mad_u64_u32
#define M64(x) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x) : "v"((uint32_t)x), "v"(m) : "vcc")
            M64(a0); M64(a1); M64(a2); M64(a3); M64(a4); M64(a5); M64(a6); M64(a7);
        } else if (OP == 1) {  // v_mul_lo_u32
#define ML(x) { uint32_t r; asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(r) : "v"((uint32_t)x), "v"(m)); x = r; }
            ML(a0); ML(a1); ML(a2); ML(a3); ML(a4); ML(a5); ML(a6); ML(a7);
        } else if (OP == 2) {  // v_mul_hi_u32
#define MH(x) { uint32_t r; asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(r) : "v"((uint32_t)x), "v"(m)); x = r | 1; }
            MH(a0); MH(a1); MH(a2); MH(a3); MH(a4); MH(a5); MH(a6); MH(a7);
        } else if (OP == 3) {  // v_mad_u32_u24
#define M24(x) { uint32_t r; asm volatile("v_mad_u32_u24 %0, %1, %2, %1" : "=v"(r) : "v"((uint32_t)x), "v"(m)); x = r; }
            M24(a0); M24(a1); M24(a2); M24(a3); M24(a4); M24(a5); M24(a6); M24(a7);
        } else if (OP == 4) {  // v_add_u32
#define AD(x) { uint32_t r; asm volatile("v_add_u32 %0, %1, %2" : "=v"(r) : "v"((uint32_t)x), "v"(m)); x = r; }
            AD(a0); AD(a1); AD(a2); AD(a3); AD(a4); AD(a5); AD(a6); AD(a7);
        } else if (OP == 5) {  // v_lshl_add_u64
#define LA(x) asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(x) : "v"(a7))
            LA(a0); LA(a1); LA(a2); LA(a3); LA(a4); LA(a5); LA(a6); a7 += 1;
        } else if (OP == 6) {  // v_add_co + v_addc (64-bit add)
            a0 += a1; a1 += a2; a2 += a3; a3 += a4; a4 += a5; a5 += a6; a6 += a7; a7 += a0;
        } else if (OP == 7) {  // v_mad_u32_u16? use v_mad_u16? -> v_mul_u32_u24
#define MU(x) { uint32_t r; asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"((uint32_t)x), "v"(m)); x = r; }
            MU(a0); MU(a1); MU(a2); MU(a3); MU(a4); MU(a5); MU(a6); MU(a7);
        } else if (OP == 8) {  // v_dot4_u32_u8 (4 x (8x8) + 32 -> 32)
#define D4(x) { uint32_t r; asm volatile("v_dot4_u32_u8 %0, %1, %2, %1" : "=v"(r) : "v"((uint32_t)x), "v"(m)); x = r; }
            D4(a0); D4(a1); D4(a2); D4(a3); D4(a4); D4(a5); D4(a6); D4(a7);
        } else if (OP == 9) {  // v_perm_b32
#define PB(x) { uint32_t r; asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(r) : "v"((uint32_t)x), "v"(m), "v"(0x07020504u)); x = r; }
            PB(a0); PB(a1); PB(a2); PB(a3); PB(a4); PB(a5); PB(a6); PB(a7);
        } else if (OP == 10) {  // v_cmp_lt_u64 + v_cndmask x2 (the compiler's conditional 64-bit fix-up)
#define CF(x, y) x = (x < y) ? x + 0xFFFFFFFFull : x
            CF(a0, a1); CF(a1, a2); CF(a2, a3); CF(a3, a4); CF(a4, a5); CF(a5, a6); CF(a6, a7); CF(a7, a0);
        } else if (OP == 11) {  // v_dot2_u32_u16? not on gfx950 -> v_mad_u16 ; use v_pk_mad_u16
#define PK(x) { uint32_t r; asm volatile("v_pk_mad_u16 %0, %1, %2, %1" : "=v"(r) : "v"((uint32_t)x), "v"(m)); x = r; }
            PK(a0); PK(a1); PK(a2); PK(a3); PK(a4); PK(a5); PK(a6); PK(a7);
        }
    }
    out[t] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
template <int OP>
void run(const char* name, uint64_t* d, int ops_per_iter) {
    dim3 grid(256 * 8), block(256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, d, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, d, 2u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double ops = (double)grid.x * block.x * ITERS * ops_per_iter;
    printf("%-16s %8.3f ms  %8.2f Tlane-ops/s  (%.2f cycles per wave-instr per SIMD @2.4GHz)\n", name, ms, ops / ms / 1e9,
           2.4e9 * (ms * 1e-3) / ((double)grid.x * block.x / 64 / (256 * 4) * ITERS * ops_per_iter));
}
int main() {
    uint64_t* d; hipMalloc(&d, 256 * 8 * 256 * 8);
    run<4>("v_add_u32", d, 8);
    run<0>("v_mad_u64_u32", d, 8);
    run<1>("v_mul_lo_u32", d, 8);
    run<2>("v_mul_hi_u32", d, 8);
    run<3>("v_mad_u32_u24", d, 8);
    run<7>("v_mul_u32_u24", d, 8);
    run<5>("v_lshl_add_u64", d, 7);
    run<6>("add64(compiler)", d, 8);
    run<8>("v_dot4_u32_u8", d, 8);
    run<9>("v_perm_b32", d, 8);
    run<10>("cmp64+add+sel", d, 8);
    run<11>("v_pk_mad_u16", d, 8);
    return 0;
}
