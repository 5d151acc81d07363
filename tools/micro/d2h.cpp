// Micro-benchmark: device-to-host copy paths for a frame-sized buffer (what the one-shot render path can use).
//   hipcc -O2 tools/micro/d2h.cpp -o /tmp/d2h && /tmp/d2h
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    const size_t MB = 1 << 20;
    const size_t total = 33177600;  // 1920 x 1080 x 16
    char* dev = nullptr;
    CK(hipMalloc(&dev, total));
    CK(hipMemset(dev, 1, total));
    char* pageable = static_cast<char*>(malloc(total));
    memset(pageable, 0, total);
    char* pinned = nullptr;
    CK(hipHostMalloc(&pinned, total, hipHostMallocDefault));
    memset(pinned, 0, total);
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int rep = 0; rep < 2; ++rep) {
        for (size_t bytes : {1 * MB, 4 * MB, 11 * MB, 22 * MB, total}) {
            double t0 = now_ms();
            CK(hipMemcpy(pageable, dev, bytes, hipMemcpyDeviceToHost));
            double t1 = now_ms();
            CK(hipMemcpyAsync(pageable, dev, bytes, hipMemcpyDeviceToHost, st));
            double t2 = now_ms();
            CK(hipStreamSynchronize(st));
            double t3 = now_ms();
            CK(hipMemcpyAsync(pinned, dev, bytes, hipMemcpyDeviceToHost, st));
            double t4 = now_ms();
            CK(hipStreamSynchronize(st));
            double t5 = now_ms();
            memcpy(pageable, pinned, bytes);
            double t6 = now_ms();
            printf("%6.1f MB: blocking pageable %.3f ms | async pageable call %.3f + sync %.3f | async pinned call %.3f + sync %.3f (%.1f GB/s) | 1-thread memcpy %.3f\n",
                   bytes / 1e6, t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, bytes / (t5 - t3) / 1e6, t6 - t5);
        }
    }
    for (int nt : {2, 4, 8, 16}) {
        double t0 = now_ms();
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t) th.emplace_back([&, t] { size_t a = total * t / nt, b = total * (t + 1) / nt; memcpy(pageable + a, pinned + a, b - a); });
        for (auto& x : th) x.join();
        printf("memcpy 33 MB with %2d threads (spawned): %.3f ms\n", nt, now_ms() - t0);
    }
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now_ms();
        CK(hipHostRegister(pageable, total, hipHostRegisterDefault));
        double t1 = now_ms();
        CK(hipMemcpyAsync(pageable, dev, total, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        double t2 = now_ms();
        CK(hipHostUnregister(pageable));
        double t3 = now_ms();
        printf("hipHostRegister 33 MB %.3f ms, copy into it %.3f ms, unregister %.3f ms\n", t1 - t0, t2 - t1, t3 - t2);
    }
    // does the async path alone ever get fast on a buffer no blocking copy has seen?
    {
        char* fresh = static_cast<char*>(malloc(total));
        memset(fresh, 0, total);
        for (int rep = 0; rep < 4; ++rep) {
            double t0 = now_ms();
            CK(hipMemcpyAsync(fresh, dev, total, hipMemcpyDeviceToHost, st));
            CK(hipStreamSynchronize(st));
            printf("fresh buffer, async + sync only, rep %d: %.3f ms\n", rep, now_ms() - t0);
        }
        char* fresh2 = static_cast<char*>(malloc(total));
        memset(fresh2, 0, total);
        for (int rep = 0; rep < 4; ++rep) {
            double t0 = now_ms();
            CK(hipMemcpy(fresh2, dev, total, hipMemcpyDeviceToHost));
            printf("fresh buffer, blocking only, rep %d: %.3f ms\n", rep, now_ms() - t0);
        }
        for (int rep = 0; rep < 2; ++rep) {
            double t0 = now_ms();
            CK(hipMemcpyAsync(fresh2, dev, total, hipMemcpyDeviceToHost, st));
            CK(hipStreamSynchronize(st));
            printf("  ... then async on it, rep %d: %.3f ms\n", rep, now_ms() - t0);
        }
        // with a kernel-busy GPU in between (a memset kernel on another stream)
        hipStream_t st2;
        CK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
        char* dev2 = nullptr;
        CK(hipMalloc(&dev2, 1ull << 30));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemsetAsync(dev2, rep, 1ull << 30, st2));
            double t0 = now_ms();
            CK(hipMemcpyAsync(fresh2, dev, total, hipMemcpyDeviceToHost, st));
            CK(hipStreamSynchronize(st));
            printf("  ... async while another stream runs a fill kernel, rep %d: %.3f ms\n", rep, now_ms() - t0);
            CK(hipStreamSynchronize(st2));
        }
    }
    // which call of the library's per-render sequence makes the runtime forget a pinned user range?
    {
        char* buf = static_cast<char*>(malloc(total));
        memset(buf, 0, total);
        char small[4096];
        auto copy_ms = [&]() {
            double t0 = now_ms();
            (void)hipMemcpyAsync(buf, dev, total, hipMemcpyDeviceToHost, st);
            (void)hipStreamSynchronize(st);
            return now_ms() - t0;
        };
        printf("suspects: first %.3f, again %.3f", copy_ms(), copy_ms());
        CK(hipDeviceSynchronize());
        printf(" | after hipDeviceSynchronize %.3f", copy_ms());
        CK(hipMemcpy(dev, small, 4096, hipMemcpyHostToDevice));
        printf(" | after small pageable H2D %.3f", copy_ms());
        CK(hipMemcpy(small, dev, 4, hipMemcpyDeviceToHost));
        printf(" | after 4-byte pageable D2H %.3f", copy_ms());
        hipEvent_t ev;
        CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        CK(hipEventRecord(ev, st));
        CK(hipEventSynchronize(ev));
        printf(" | after event sync %.3f", copy_ms());
        hipStream_t other;
        CK(hipStreamCreateWithFlags(&other, hipStreamNonBlocking));
        CK(hipMemsetAsync(dev, 0, 1 << 20, other));
        CK(hipStreamSynchronize(other));
        printf(" | after work on another stream %.3f", copy_ms());
        double a0 = copy_ms();
        (void)hipMemcpyAsync(buf, dev, 6 * MB, hipMemcpyDeviceToHost, st);
        (void)hipMemcpyAsync(buf + 28 * MB, dev + 28 * MB, total - 28 * MB, hipMemcpyDeviceToHost, st);
        (void)hipStreamSynchronize(st);
        double t0 = now_ms();
        (void)hipMemcpyAsync(buf + 6 * MB, dev + 6 * MB, 22 * MB, hipMemcpyDeviceToHost, st);
        (void)hipStreamSynchronize(st);
        printf(" | whole %.3f, then as sub-ranges: 22 MB middle part %.3f", a0, now_ms() - t0);
        t0 = now_ms();
        (void)hipMemcpyAsync(buf + 6 * MB, dev + 6 * MB, 22 * MB, hipMemcpyDeviceToHost, st);
        (void)hipStreamSynchronize(st);
        printf(", again %.3f\n", now_ms() - t0);
    }
    // how many separately pinned ranges does the runtime remember?  k disjoint ranges of a fresh buffer, cycled
    for (int k = 1; k <= 4; ++k) {
        char* buf = static_cast<char*>(malloc(total));
        memset(buf, 0, total);
        const size_t part = total / k / 4096 * 4096;
        printf("%d disjoint range(s) of a fresh buffer, cycled:", k);
        for (int rep = 0; rep < 4; ++rep) {
            double t0 = now_ms();
            for (int j = 0; j < k; ++j) {
                (void)hipMemcpyAsync(buf + j * part, dev + j * part, part, hipMemcpyDeviceToHost, st);
                (void)hipStreamSynchronize(st);
            }
            printf(" %.3f", now_ms() - t0);
        }
        printf(" ms per cycle\n");
    }
    // three row-group copies the way the one-shot path would issue them (blocking, pageable)
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now_ms();
        CK(hipMemcpy(pageable, dev, 6 * MB, hipMemcpyDeviceToHost));
        CK(hipMemcpy(pageable + 28 * MB, dev + 28 * MB, total - 28 * MB, hipMemcpyDeviceToHost));
        CK(hipMemcpy(pageable + 6 * MB, dev + 6 * MB, 22 * MB, hipMemcpyDeviceToHost));
        printf("three blocking pageable copies (6 + 3.6 + 22 MB): %.3f ms\n", now_ms() - t0);
    }
    return 0;
}
