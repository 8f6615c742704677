// Calibration for rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 with THIS repo's access pattern: one u64 (8 B) per lane,
// 512 B per wave-instruction, streaming over a buffer far larger than the 256 MiB Infinity Cache.
// k_read8 reads `n` u64 (known bytes = 8n) and writes one u64 per workgroup; k_write8 writes n u64.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__global__ void k_read8(const uint64_t* in, uint64_t* out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    uint64_t acc = 0;
    for (; i < n; i += stride) acc ^= in[i];
    if (acc == 0x1234567) out[blockIdx.x] = acc;  // practically never: keeps the loads alive without write traffic
}
__global__ void k_write8(uint64_t* out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = i;
}
int main() {
    size_t n = (size_t)1 << 28;  // 2 GiB
    uint64_t *a, *b;
    hipMalloc(&a, n * 8); hipMalloc(&b, 1 << 20);
    hipMemset(a, 1, n * 8);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k_read8, dim3(2048), dim3(256), 0, 0, a, b, n);
    hipLaunchKernelGGL(k_write8, dim3(2048), dim3(256), 0, 0, a, n);
    hipDeviceSynchronize();
    printf("known bytes: read %zu, written %zu\n", n * 8, n * 8);
    return 0;
}
