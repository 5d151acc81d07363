// micro-benchmark: cost of dispatching many small workgroups as a function of LDS / scratch footprint
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int LDS_BYTES, int SCRATCH_WORDS>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ float lds[LDS_BYTES / 4 > 0 ? LDS_BYTES / 4 : 1];
    float priv[SCRATCH_WORDS > 0 ? SCRATCH_WORDS : 1];
    float acc = threadIdx.x;
    for (int i = 0; i < iters; ++i) acc = acc * 1.0001f + 0.5f;
    lds[threadIdx.x % (LDS_BYTES / 4 > 0 ? LDS_BYTES / 4 : 1)] = acc;
    if (SCRATCH_WORDS > 0 && iters == 12345) {  // scratch is allocated for the dispatch but never touched
        for (int i = 0; i < SCRATCH_WORDS; ++i) priv[i] = acc + i;
        acc = priv[(int)(acc) & (SCRATCH_WORDS - 1)];
    }
    __syncthreads();
    if (acc == 12345.678f) out[blockIdx.x] = lds[0] + priv[0];
}

template <int L, int S>
float run(int blocks, int iters, float* d) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<L, S>), dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(a);
    for (int w = 0; w < 10; ++w) hipLaunchKernelGGL((k<L, S>), dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / 10 * 1000;
}
int main() {
    float* d; CK(hipMalloc(&d, 1 << 20));
    for (int blocks : {2048, 8192, 32768}) {
        for (int iters : {0, 200}) {
            printf("blocks %6d iters %3d | lds0/s0 %7.1f us | lds40k/s0 %7.1f | lds40k/s64w %7.1f | lds40k/s128w %7.1f | lds8k/s0 %7.1f | lds8k/s64w %7.1f\n", blocks, iters,
                   run<0, 0>(blocks, iters, d), run<40960, 0>(blocks, iters, d), run<40960, 64>(blocks, iters, d), run<40960, 128>(blocks, iters, d),
                   run<8192, 0>(blocks, iters, d), run<8192, 64>(blocks, iters, d));
        }
    }
    return 0;
}
