// ASan/UBSan driver for the pure-host parts: scene builder + flattener over every pose and skin layout,
// plus malformed descriptions.
#include "flatten.h"
#include "mcrt.h"
#include <cmath>
#include <cstdlib>
#include <limits>
#include <cstdio>
#include <cstdint>
#include <string>
#include <vector>
int mcrt_detail_fail(int code, const char*) { return code; }  // api.cpp's error hook (not linked here)

static int png_checks() {
    int bad = 0;
    const int sizes[][2] = {{1, 1}, {3, 2}, {257, 31}, {16383, 1}, {16384, 1}, {4095, 4}, {4096, 4}, {13107, 5}};
    for (const auto& wh : sizes) {
        const int w = wh[0], h = wh[1];
        std::vector<uint8_t> img(static_cast<size_t>(w) * h * 4);
        uint32_t s = 99;
        for (auto& b : img) { s = s * 1664525u + 1013904223u; b = static_cast<uint8_t>(s >> 24); }
        const size_t need = mcrt_encode_png_rgba8(img.data(), w, h, nullptr, 0);
        std::vector<uint8_t> out(need);  // exactly `need` bytes: any overrun is an ASan error
        if (!need || mcrt_encode_png_rgba8(img.data(), w, h, out.data(), out.size()) != need) ++bad;
        if (need > 8 && mcrt_encode_png_rgba8(img.data(), w, h, out.data(), need - 1) != need) ++bad;  // too small: size only
    }
    // quantiser on awkward values (image_writer.cpp:18-22 semantics: clamp then *255+0.5 then truncate)
    const float inf = std::numeric_limits<float>::infinity();
    const float vals[8] = {-1.0f, 0.0f, 0.5f, 1.0f, 2.0f, inf, -inf, 0.99999994f};
    uint8_t q[8];
    mcrt_quantize_rgba8(vals, q, 2);
    if (q[0] != 0 || q[1] != 0 || q[2] != 128 || q[3] != 255 || q[4] != 255 || q[5] != 255 || q[6] != 0 || q[7] != 255) ++bad;
    std::vector<float> f(4 * 6, 0.25f);
    if (mcrt_write_png_f32("/tmp/mcrt_asan_host.png", f.data(), 3, 2) != MCRT_OK) ++bad;
    if (mcrt_write_png_f32("/nonexistent_dir_mcrt/x.png", f.data(), 3, 2) == MCRT_OK) ++bad;
    if (mcrt_write_png_rgba8("/tmp/mcrt_asan_host.png", nullptr, 3, 2) == MCRT_OK) ++bad;
    std::remove("/tmp/mcrt_asan_host.png");
    return bad;
}

int main() {
    int bad = png_checks();
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
