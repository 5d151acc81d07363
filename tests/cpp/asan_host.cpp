// ASan/UBSan driver for the pure-host parts: scene builder + flattener over every pose and skin layout,
// plus malformed descriptions.
#include "flatten.h"
#include "mcrt.h"
#include <cstdio>
#include <cstdint>
#include <string>
#include <vector>
int main() {
    int bad = 0;
    for (int legacy = 0; legacy < 2; ++legacy) {
        const int w = 64, h = legacy ? 32 : 64;
        std::vector<uint8_t> skin(static_cast<size_t>(w) * h * 4);
        uint32_t s = 12345;
        for (auto& b : skin) { s = s * 1664525u + 1013904223u; b = static_cast<uint8_t>(s >> 24); }
        for (int pose = 0; pose < 7; ++pose) {
            float p[12];
            if (mcrt_builtin_pose(pose, p) != 0) { ++bad; continue; }
            mcrt_scene_desc* d = nullptr;
            if (mcrt_build_skin_scene(skin.data(), w, h, p, &d) != 0 || !d) { ++bad; continue; }
            std::vector<uint8_t> blob; std::string err;
            if (!mcrt::flatten_scene(d, blob, err)) { std::printf("flatten failed: %s\n", err.c_str()); ++bad; }
            // a truncated description must be rejected, not read out of bounds
            mcrt_scene_desc cut = *d;
            cut.n_meshes = d->n_meshes;  // same meshes, texture table cut short
            cut.n_textures = d->n_textures > 1 ? 1 : 0;
            std::vector<uint8_t> blob2; std::string err2;
            if (mcrt::flatten_scene(&cut, blob2, err2) && d->n_textures > 1) { std::printf("accepted a cut texture table\n"); ++bad; }
            mcrt_scene_desc_free(d);
        }
    }
    float p0[12] = {0};
    mcrt_scene_desc* d = nullptr;
    if (mcrt_build_default_scene(p0, &d) != 0) ++bad; else { std::vector<uint8_t> b; std::string e; if (!mcrt::flatten_scene(d, b, e)) ++bad; mcrt_scene_desc_free(d); }
    std::printf("asan driver: %d problem(s)\n", bad);
    return bad != 0;
}
