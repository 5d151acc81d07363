// Micro-benchmark: what a HIP call costs right after the application has FREED a large pageable buffer that the runtime had
// pinned for a device-to-host copy (a fresh Image per render call, dropped before the next call — the reference's call pattern).
// No library code involved: hipMalloc, hipMemcpy into malloc'ed memory, free, then a tiny hipMemcpy.
//   hipcc -O2 tools/micro/free_pinned.cpp -o /tmp/free_pinned && /tmp/free_pinned
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    const size_t sizes[] = {33177600ull, 132710400ull, 530841600ull};  // 1080p, 4K, 8K float frames
    char* dev = nullptr;
    CK(hipMalloc(&dev, sizes[2]));
    CK(hipMemset(dev, 1, sizes[2]));
    char small_host[256];
    for (size_t bytes : sizes) {
        for (int keep = 0; keep < 2; ++keep) {
            double copy_ms = 0, next_ms = 0, alloc_ms = 0;
            char* kept = nullptr;
            const int reps = 5;
            for (int r = 0; r < reps + 1; ++r) {
                double t0 = now_ms();
                char* img = keep && kept ? kept : static_cast<char*>(malloc(bytes));
                memset(img, 0, bytes);  // the Image's constructor fills its pixels
                double t1 = now_ms();
                CK(hipMemcpy(small_host, dev, 256, hipMemcpyDeviceToHost));  // the first HIP call of the next "render"
                double t2 = now_ms();
                CK(hipMemcpy(img, dev, bytes, hipMemcpyDeviceToHost));
                double t3 = now_ms();
                if (keep) kept = img; else free(img);  // the caller drops the Image (or reuses one buffer)
                if (r > 0) alloc_ms += t1 - t0, next_ms += t2 - t1, copy_ms += t3 - t2;
            }
            if (kept) free(kept);
            printf("%4zu MB, %s: alloc+fill %.2f ms, first HIP call after it %.2f ms, copy %.2f ms\n", bytes >> 20,
                   keep ? "one buffer reused      " : "fresh buffer, freed each", alloc_ms / reps, next_ms / reps, copy_ms / reps);
        }
    }
    return 0;
}
