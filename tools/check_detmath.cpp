// Exhaustive / sampled comparison of include/mcrt_detmath.h against the system libm.
// usage: check_detmath [stride]   (stride 1 = every float; default 1)
// Build: g++ -O2 -std=c++17 -ffp-contract=off -Iinclude tools/check_detmath.cpp -o /tmp/check_detmath -lpthread -lm
#include "mcrt_detmath.h"
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

static bool same(float a, float b) {
    uint32_t ua, ub;
    memcpy(&ua, &a, 4);
    memcpy(&ub, &b, 4);
    if (ua == ub) return true;
    return std::isnan(a) && std::isnan(b);
}

int main(int argc, char** argv) {
    uint64_t stride = argc > 1 ? strtoull(argv[1], 0, 10) : 1;
    unsigned nt = std::thread::hardware_concurrency();
    if (!nt) nt = 4;
    std::atomic<uint64_t> bad_sin{0}, bad_cos{0}, bad_pow16{0}, bad_powr{0}, bad_sincos{0};
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t)
        th.emplace_back([&, t] {
            uint64_t bs = 0, bc = 0, bp = 0, br = 0, bsc = 0;
            for (uint64_t u = t * stride; u < (1ull << 32); u += (uint64_t)nt * stride) {
                float x;
                uint32_t uu = (uint32_t)u;
                memcpy(&x, &uu, 4);
                if (!same(sinf(x), mcrt_sinf(x))) { if (bs < 3) fprintf(stderr, "sin mismatch %a: %a vs %a\n", x, sinf(x), mcrt_sinf(x)); ++bs; }
                if (!same(cosf(x), mcrt_cosf(x))) { if (bc < 3) fprintf(stderr, "cos mismatch %a: %a vs %a\n", x, cosf(x), mcrt_cosf(x)); ++bc; }
                {  // the fused form against the system libm as well
                    float fs, fc;
                    mcrt_sincosf(x, &fs, &fc);
                    if (!same(sinf(x), fs) || !same(cosf(x), fc)) { if (bsc < 3) fprintf(stderr, "sincos mismatch %a: (%a,%a) vs (%a,%a)\n", x, sinf(x), cosf(x), fs, fc); ++bsc; }
                }
                if (uu <= 0x40000000u) { // x in [0, 2]
                    if (!same(powf(x, 16.0f), mcrt_powf(x, 16.0f))) { if (bp < 3) fprintf(stderr, "pow16 mismatch %a: %a vs %a\n", x, powf(x, 16.0f), mcrt_powf(x, 16.0f)); ++bp; }
                }
            }
            // random (x>=0, y>0) pairs
            std::mt19937 g(1234 + t);
            for (int i = 0; i < 20000000 / (int)stride + 1000; ++i) {
                uint32_t a = g() & 0x7fffffffu, b = g() & 0x7fffffffu;
                if ((g() & 3) == 0) b = (b % 0x06000000u) + 0x3c000000u; // y in a moderate range
                if ((g() & 3) == 0) a = (a % 0x04000000u) + 0x3e000000u;
                float x, y;
                memcpy(&x, &a, 4);
                memcpy(&y, &b, 4);
                if (std::isnan(x) || std::isnan(y) || std::isinf(y) || y == 0.0f) continue;
                if (!same(powf(x, y), mcrt_powf(x, y))) { if (br < 3) fprintf(stderr, "pow mismatch %a^%a: %a vs %a\n", x, y, powf(x, y), mcrt_powf(x, y)); ++br; }
            }
            bad_sin += bs; bad_cos += bc; bad_pow16 += bp; bad_powr += br; bad_sincos += bsc;
        });
    for (auto& x : th) x.join();
    printf("stride=%llu sin_mismatch=%llu cos_mismatch=%llu sincos_mismatch=%llu pow16_mismatch=%llu pow_random_mismatch=%llu\n",
           (unsigned long long)stride, (unsigned long long)bad_sin.load(), (unsigned long long)bad_cos.load(),
           (unsigned long long)bad_sincos.load(), (unsigned long long)bad_pow16.load(), (unsigned long long)bad_powr.load());
    return (bad_sin | bad_cos | bad_sincos | bad_pow16 | bad_powr) ? 1 : 0;
}
