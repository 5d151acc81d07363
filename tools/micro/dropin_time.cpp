// dropin_time.cpp — wall time of the C++ drop-in `TileRenderer::render()` (csrc/host/tile_renderer_hip.cpp) the way the
// reference's one call site uses it (main_window.cpp:526-540): a `Scene` built by the caller, a fresh `Image` returned
// by value per call.  Prints one JSON object: the cost of `Image(W, H)` alone, the first call of the process (HIP start-up,
// workspace allocation, seed table, code load), then the median / minimum of the following calls — with the returned
// Image dropped before the next call, and with every Image kept alive.
//
//   g++ -std=c++17 -O2 -Iinclude -Iminecraftskin_raytracer_amd/csrc/host tools/micro/dropin_time.cpp
//       minecraftskin_raytracer_amd/csrc/host/tile_renderer_hip.cpp -Lminecraftskin_raytracer_amd -lmcrt -Wl,-rpath,<dir> -o dropin_time
//   ./dropin_time [width height bounces spp calls]
#include "mcskin_types.hpp"

#include "mcrt.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

static double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// SURVEY.md §8(d) skin S64 (same generator as minecraftskin_raytracer_amd/skins.py)
static std::vector<uint8_t> synthetic_skin_s64() {
    std::vector<uint8_t> img(64 * 64 * 4);
    uint32_t s = 12345u;
    auto step = [&] { s = s * 1664525u + 1013904223u; return s; };
    for (int y = 0; y < 64; ++y)
        for (int x = 0; x < 64; ++x) {
            uint8_t* p = &img[(static_cast<size_t>(y) * 64 + x) * 4];
            for (int c = 0; c < 3; ++c) p[c] = static_cast<uint8_t>(step() >> 24);
            const bool outer = (y < 16 && x >= 32) || (y >= 32 && y < 48) || (y >= 48 && (x < 16 || x >= 48));
            p[3] = 255;
            if (outer) p[3] = (step() >> 28) < 6u ? 255 : 0;
        }
    return img;
}

// the description the native scene builder returns → a reference-shaped Scene (what a caller of the reference holds)
static Scene scene_from_desc(const mcrt_scene_desc& d) {
    Scene sc;
    for (int m = 0; m < d.n_meshes; ++m) {
        const mcrt_mesh& mm = d.meshes[m];
        Mesh mesh;
        int used = 0;
        int tex_ids[6] = {-1, -1, -1, -1, -1, -1};
        auto slot = [&](int tex) -> const TextureRegion* {
            if (tex < 0) return nullptr;
            for (int i = 0; i < used; ++i)
                if (tex_ids[i] == tex) return &mesh.ownedTextures[static_cast<size_t>(i)];
            if (used >= 6) return nullptr;
            const mcrt_texture& t = d.textures[tex];
            std::vector<Color> px(static_cast<size_t>(t.n_pixels));
            for (size_t i = 0; i < px.size(); ++i) px[i] = Color(t.rgba[4 * i], t.rgba[4 * i + 1], t.rgba[4 * i + 2], t.rgba[4 * i + 3]);
            mesh.ownedTextures[static_cast<size_t>(used)] = TextureRegion(t.width, t.height, std::move(px));
            tex_ids[used] = tex;
            return &mesh.ownedTextures[static_cast<size_t>(used++)];
        };
        auto tri = [&](const float* v, int tex) {
            Triangle t;
            t.v0 = Vec3(v[0], v[1], v[2]), t.v1 = Vec3(v[3], v[4], v[5]), t.v2 = Vec3(v[6], v[7], v[8]);
            t.texture = slot(tex);
            return t;
        };
        for (int i = 0; i < mm.n_triangles; ++i) mesh.triangles.push_back(tri(mm.tri_vertices + 9 * i, mm.tri_texture[i]));
        for (int i = 0; i < mm.n_local_triangles; ++i) mesh.localTriangles.push_back(tri(mm.local_tri_vertices + 9 * i, i < mm.n_triangles ? mm.tri_texture[i] : -1));
        mesh.isOuterLayer = mm.is_outer_layer != 0;
        mesh.hasRotation = mm.has_rotation != 0;
        mesh.pivot = Vec3(mm.pivot[0], mm.pivot[1], mm.pivot[2]);
        mesh.rotX = mm.rot_x, mesh.rotZ = mm.rot_z;
        sc.meshes.push_back(std::move(mesh));
    }
    sc.light.position = Vec3(d.light_position[0], d.light_position[1], d.light_position[2]);
    sc.light.color = Color(d.light_color[0], d.light_color[1], d.light_color[2], d.light_color[3]);
    sc.light.intensity = d.light_intensity, sc.light.radius = d.light_radius;
    sc.camera.position = Vec3(d.camera_position[0], d.camera_position[1], d.camera_position[2]);
    sc.camera.target = Vec3(d.camera_target[0], d.camera_target[1], d.camera_target[2]);
    sc.camera.up = Vec3(d.camera_up[0], d.camera_up[1], d.camera_up[2]);
    sc.camera.fov = d.camera_fov;
    sc.backgroundColor = Color(d.background_color[0], d.background_color[1], d.background_color[2], d.background_color[3]);
    return sc;
}

int main(int argc, char** argv) {
    RayTracer::Config c;
    c.width = argc > 1 ? std::atoi(argv[1]) : 1920;
    c.height = argc > 2 ? std::atoi(argv[2]) : 1080;
    c.maxBounces = argc > 3 ? std::atoi(argv[3]) : 4;
    c.samplesPerPixel = argc > 4 ? std::atoi(argv[4]) : 4;
    const int calls = argc > 5 ? std::atoi(argv[5]) : 12;
    const std::vector<uint8_t> skin = synthetic_skin_s64();
    float pose[12] = {0};
    mcrt_scene_desc* desc = nullptr;
    if (mcrt_build_skin_scene(skin.data(), 64, 64, pose, &desc) != MCRT_OK) {
        std::printf("{\"error\": \"scene build failed\"}\n");
        return 1;
    }
    const Scene scene = scene_from_desc(*desc);
    mcrt_scene_desc_free(desc);

    // `Image(W, H)` alone, before anything else has touched the allocator or the GPU (33 MB of fresh pages at 1080p)
    std::vector<double> ctor;
    double checksum = 0.0;
    for (int i = 0; i < 5; ++i) {
        const double t = now_ms();
        Image blank(c.width, c.height);
        ctor.push_back(now_ms() - t);
        checksum += blank.pixels[blank.pixels.size() / 3].a;
    }
    double t0 = now_ms();
    Image first = TileRenderer::render(scene, c);
    const double first_ms = now_ms() - t0;
    if (!TileRenderer::lastErrors().empty()) {
        std::printf("{\"error\": \"%s\"}\n", TileRenderer::lastErrors()[0].message.c_str());
        return 1;
    }
    // (a) the returned Image is dropped before the next call — an export loop; (b) every Image is kept — each call gets pages
    // the HIP runtime has never seen
    std::vector<double> dropped, kept_ms;
    mcrt_timings tm{};
    for (int i = 0; i < calls; ++i) {
        t0 = now_ms();
        Image img = TileRenderer::render(scene, c);
        dropped.push_back(now_ms() - t0);
        checksum += img.pixels[img.pixels.size() / 2].r;
    }
    mcrt_last_timings(&tm);
    std::vector<Image> kept;
    for (int i = 0; i < calls; ++i) {
        t0 = now_ms();
        kept.push_back(TileRenderer::render(scene, c));
        kept_ms.push_back(now_ms() - t0);
        checksum += kept.back().pixels[kept.back().pixels.size() / 2].g;
    }
    std::sort(dropped.begin(), dropped.end());
    std::sort(kept_ms.begin(), kept_ms.end());
    std::sort(ctor.begin(), ctor.end());
    std::printf("{\"width\": %d, \"height\": %d, \"first_ms\": %.3f, \"ms\": %.3f, \"min_ms\": %.3f, \"images_kept_ms\": %.3f, \"calls\": %d, \"image_ctor_ms\": %.3f, "
                "\"last_split_ms\": {\"flatten_ms\": %.3f, \"h2d_ms\": %.3f, \"kernel_ms\": %.3f, \"d2h_ms\": %.3f, \"total_ms\": %.3f}, \"checksum\": %.6f}\n",
                c.width, c.height, first_ms, dropped[dropped.size() / 2], dropped.front(), kept_ms[kept_ms.size() / 2], calls, ctor[ctor.size() / 2], tm.flatten_ms,
                tm.h2d_ms, tm.kernel_ms, tm.d2h_ms, tm.total_ms, checksum + first.pixels[0].r);
    return 0;
}
