// One-shot mcrt_render from plain C++ (no Python): wall time per call and the library's split.
//   g++ -O2 tools/micro/hostpath.cpp -Iinclude -Lminecraftskin_raytracer_amd -lmcrt -Wl,-rpath,$PWD/minecraftskin_raytracer_amd -o tools/micro/hostpath_bin
#include "mcrt.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
static double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}
int main(int argc, char** argv) {
    float pose[12] = {0};
    mcrt_scene_desc* sd = nullptr;
    if (mcrt_build_default_scene(pose, &sd) != MCRT_OK) { printf("scene: %s\n", mcrt_last_error()); return 1; }
    mcrt_config cfg;
    mcrt_config_init(&cfg);
    cfg.width = 1920, cfg.height = 1080, cfg.max_bounces = 4, cfg.samples_per_pixel = 4;
    std::vector<float> out(static_cast<size_t>(cfg.width) * cfg.height * 4, 1.0f);
    const int n = argc > 1 ? atoi(argv[1]) : 8;
    for (int i = 0; i < n; ++i) {
        const double t0 = now_ms();
        const int rc = mcrt_render(sd, &cfg, out.data(), nullptr, nullptr, 0);
        const double dt = now_ms() - t0;
        mcrt_timings t;
        mcrt_last_timings(&t);
        printf("call %d rc %d: %.3f ms (flatten %.3f, upload+launch %.3f, kernel %.3f, d2h %.3f)\n", i, rc, dt, t.flatten_ms, t.h2d_ms, t.kernel_ms, t.d2h_ms);
        if (rc != MCRT_OK) { printf("%s\n", mcrt_last_error()); return 1; }
    }
    mcrt_scene_desc_free(sd);
    return 0;
}
