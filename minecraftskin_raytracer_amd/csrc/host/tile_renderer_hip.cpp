// tile_renderer_hip.cpp — drop-in replacement for the reference's
// src/raytracer/tile_renderer.cpp: the same TileRenderer static interface
// (raytracer/tile_renderer.h:16-47), every render routed through the C ABI of libmcrt.so into the
// gfx950 kernels.  Build it INSTEAD of the reference file:
//
//   with the reference tree:  -DMCRT_USE_REFERENCE_HEADERS -I<reference>/src   
//   stand-alone (this repo):  includes the mirror types in mcskin_types.hpp
//
// Behaviour kept from the reference (tile_renderer.cpp):
//   :18-39   generateTiles: row-major grid, clipped edge tiles, empty on non-positive arguments
//   :129-189 render: returns Image(W,H); never throws for render failures — the message is kept in
//            lastErrors() as TileError{-1, msg} and the pixels stay Color() = (0,0,0,1);
//            progressCallback called exactly totalTiles times (done = 1..total) — here from the
//            calling thread, for each group of tile rows once it has landed in the Image (rows that
//            hold only background right after the first kernel, the others when the render is done;
//            pass by pass for frames that take several);
//            config.threadCount is accepted and ignored; config.tileSize is honoured (it seeds the
//            per-tile RNG streams)
//   :71-127  renderTile: one Tile — any rectangle of the frame — into an existing Image
// The device is chosen with the environment variable MCRT_DEVICE: an index (default 0) or "all" — every
// visible device renders its cyclic share of the tile rows and downloads it straight into the Image
// (mcrt_render_multi; MCRT_GATHER=1 assembles on the first device by peer copies instead).
#ifdef MCRT_USE_REFERENCE_HEADERS
#include "raytracer/tile_renderer.h"
#else
#include "mcskin_types.hpp"
#endif

#include "mcrt.h"
#include "mcrt_scene_adapter.hpp"

#include <cstdlib>
#include <cstring>

std::vector<TileRenderer::TileError> TileRenderer::errors_;

namespace {
int chosen_device() {
    const char* e = std::getenv("MCRT_DEVICE");
    if (e && std::strcmp(e, "all") == 0) return MCRT_DEVICE_ALL;
    return e ? std::atoi(e) : 0;
}
int chosen_gather() {
    const char* e = std::getenv("MCRT_GATHER");
    return e ? std::atoi(e) : 0;
}
void progress_trampoline(int done, int total, void* user) {
    (*static_cast<std::function<void(int, int)>*>(user))(done, total);
}
static_assert(sizeof(Color) == 4 * sizeof(float), "Image::pixels must be a dense float4 array");
}  // namespace

std::vector<Tile> TileRenderer::generateTiles(int imageWidth, int imageHeight, int tileSize) {
    int n = mcrt_generate_tiles(imageWidth, imageHeight, tileSize, nullptr, 0);
    std::vector<mcrt_tile> raw(static_cast<size_t>(n));
    mcrt_generate_tiles(imageWidth, imageHeight, tileSize, raw.data(), n);
    std::vector<Tile> tiles;
    tiles.reserve(raw.size());
    for (const mcrt_tile& t : raw) tiles.push_back(Tile{t.x, t.y, t.width, t.height});
    return tiles;
}

Image TileRenderer::render(const Scene& scene, const RayTracer::Config& config,
                           std::function<void(int, int)> progressCallback) {
    Image output(config.width, config.height);
    errors_.clear();
    if (config.width <= 0 || config.height <= 0 || config.tileSize <= 0) return output;

    mcrt_adapter::SceneDescription desc(scene);
    mcrt_config cfg = mcrt_adapter::to_mcrt_config(config);
    const int device = chosen_device();
    float* pixels = reinterpret_cast<float*>(output.pixels.data());
    mcrt_progress_fn cb = progressCallback ? progress_trampoline : nullptr;
    void* user = progressCallback ? &progressCallback : nullptr;
    int rc = device == MCRT_DEVICE_ALL ? mcrt_render_multi(desc.get(), &cfg, pixels, cb, user, nullptr, 0, chosen_gather())
                                       : mcrt_render(desc.get(), &cfg, pixels, cb, user, device);
    if (rc != MCRT_OK) {
        errors_.push_back({-1, mcrt_last_error()});
        for (Color& c : output.pixels) c = Color();
    }
    return output;
}

void TileRenderer::renderTile(const Tile& tile, const Scene& scene, const RayTracer::Config& config, Image& output) {
    if (config.width <= 0 || config.height <= 0) return;
    mcrt_adapter::SceneDescription desc(scene);
    mcrt_config cfg = mcrt_adapter::to_mcrt_config(config);
    const int device = chosen_device() == MCRT_DEVICE_ALL ? 0 : chosen_device();
    const mcrt_tile t{tile.x, tile.y, tile.width, tile.height};  // any rectangle of the frame, on or off the tile grid
    if (mcrt_render_rect(desc.get(), &cfg, &t, reinterpret_cast<float*>(output.pixels.data()), device) != MCRT_OK)
        errors_.push_back({-1, mcrt_last_error()});
}

const std::vector<TileRenderer::TileError>& TileRenderer::lastErrors() { return errors_; }
