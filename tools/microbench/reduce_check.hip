// Device-side cross-check of reduction variants: textbook reduce128 (compare-and-select fix-ups) against the carry-chain
// form, with and without the final canonicalisation, on random and extreme 128-bit inputs.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../plonky2-aes_amd/csrc/gl.h"
#include "../../plonky2-aes_amd/csrc/poseidon_fast.h"
typedef gl::u64 u64; typedef gl::u32 u32;
__device__ __host__ inline u64 canon_chain(u64 hi, u64 lo) {
    u64 r = glf::red128(hi, lo);
    u32 k, K;
    const u32 r0 = (u32)r, r1 = (u32)(r >> 32);
    const u32 c0 = __builtin_addc(r0, 0xFFFFFFFFu, 0u, &k);
    const u32 c1 = __builtin_addc(r1, 0u, k, &K);
    return K ? (((u64)c1 << 32) | c0) : r;
}
__global__ void k(unsigned long long* bad, u64 seed) {
    u64 x = seed + 0x9E3779B97F4A7C15ull * (blockIdx.x * blockDim.x + threadIdx.x + 1);
    unsigned long long b1 = 0, b2 = 0;
    for (int i = 0; i < 4096; i++) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17; u64 hi = x;
        x ^= x << 13; x ^= x >> 7; x ^= x << 17; u64 lo = x;
        int m = i & 15;
        if (m == 0) hi &= 0xFFFFFFFFull; if (m == 1) lo &= 0xFFFFFFFFull; if (m == 2) hi |= 0xFFFFFFFF00000000ull; if (m == 3) lo |= 0xFFFFFFFF00000000ull;
        if (m == 4) hi = 0; if (m == 5) lo = 0; if (m == 6) { hi = ~0ull; lo = ~0ull - (x & 3); } if (m == 7) { hi &= 0xFFFFFFFF00000000ull; lo &= 0xFFFFFFFFull; }
        u64 want = gl::reduce128(hi, lo);
        u64 a = glf::red128(hi, lo);
        if (glf::canon(a) != want) b1++;
        if (canon_chain(hi, lo) != want) b2++;
    }
    atomicAdd(&bad[0], b1);
    atomicAdd(&bad[1], b2);
}
int main() {
    unsigned long long* d; hipMalloc(&d, 16); hipMemset(d, 0, 16);
    hipLaunchKernelGGL(k, dim3(1024), dim3(256), 0, 0, d, 12345ull);
    unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("device: %llu mismatches (lazy chain + canon), %llu (chain with carry-based canonicalisation) over %llu inputs\n", h[0], h[1], 1024ull * 256 * 4096);
    return (h[0] || h[1]) ? 1 : 0;
}
