// Micro-benchmark: the mt19937 seeding recurrence as v_mul_lo_u32 + add vs one v_mad_u64_u32.
// hipcc --offload-arch=gfx950 -O3 tools/micro/mtchain.hip -o /tmp/mtchain && /tmp/mtchain
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__device__ __forceinline__ uint32_t step_a(uint32_t x, uint32_t i) { return 1812433253u * (x ^ (x >> 30)) + i; }
__device__ __forceinline__ uint32_t step_c(uint32_t x, uint32_t i) {
    uint32_t t = x ^ (x >> 30);
    unsigned long long acc = i, out, carry;
    asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(out), "=s"(carry) : "v"(t), "v"(1812433253u), "s"(acc));
    return static_cast<uint32_t>(out);
}
template <int V> __global__ void k(uint32_t* p, int n) {
    uint32_t x = p[blockIdx.x * blockDim.x + threadIdx.x];
#pragma unroll 8
    for (int i = 1; i <= n; ++i) x = V == 0 ? step_a(x, i) : step_c(x, i);
    p[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
int main() {
    const int blocks = 8192, threads = 256, n = 397 * 8;
    uint32_t* d;
    hipMalloc(&d, size_t(blocks) * threads * 4);
    hipMemset(d, 1, size_t(blocks) * threads * 4);
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    uint32_t h[2][4];
    for (int v = 0; v < 2; ++v) {
        for (int grid : {blocks, 1}) {
            hipMemset(d, 1, size_t(blocks) * threads * 4);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(a);
                if (v == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(grid == 1 ? 64 : threads), 0, 0, d, n);
                else hipLaunchKernelGGL(k<2>, dim3(grid), dim3(grid == 1 ? 64 : threads), 0, 0, d, n);
                hipEventRecord(b);
                hipEventSynchronize(b);
            }
            float ms;
            hipEventElapsedTime(&ms, a, b);
            printf("variant %s grid %5d: %8.3f ms  (%.2f ns per step%s)\n", v ? "mad_u64" : "mul_lo+add", grid, ms,
                   grid == 1 ? ms * 1e6 / n : ms * 1e6 / n / (double(grid) * threads / 64 / 1024), grid == 1 ? ", one wave: latency" : " per wave-slot per SIMD");
        }
        hipMemcpy(h[v], d, 16, hipMemcpyDeviceToHost);
    }
    printf("results %s\n", (h[0][0] == h[1][0] && h[0][1] == h[1][1]) ? "equal" : "DIFFER");
    return 0;
}
