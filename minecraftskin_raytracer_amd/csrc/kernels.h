// kernels.h — launch interface between the C-ABI host code (api.cpp) and the HIP kernels
// (render_kernels.hip).  Plain structs, no HIP types in the signatures beyond hipStream_t.
#ifndef MCRT_KERNELS_H
#define MCRT_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mcrt.h"

namespace mcrt {

// Which tile rows of the frame a launch owns: rows first, first+step, ...
struct Shard {
    int first;
    int step;
    int tiles_x;     // tile columns
    int tiles_y;     // tile rows in the whole frame
    int owned_rows;  // number of tile rows owned
};

struct RenderParams {
    const uint8_t* scene;  // flat blob in HBM
    mcrt_config cfg;
    Shard shard;
    int layout;            // MCRT_LAYOUT_*
    float* out;            // float4 frame or packed rows
    uint32_t* tile_rng;    // owned_tiles x 624 seeded mt19937 words (NULL when no tile draws)
    uint32_t* hit_rng;     // per-thread 624-word slices for long per-hit streams (NULL normally)
    float* deep_stack;     // per-thread (max_bounces) x float4 slices when max_bounces > 16
    int draws_per_sample;  // 0, 2 or 4
    int grid_blocks;
    int parts_per_tile;    // each tile is split into this many pixel-aligned work units
    int lds_draw_floats;   // dynamic LDS: per-hit shadow draws of one sub-batch
    int lds_alpha_words;   // dynamic LDS: alpha-predicate words staged per workgroup (0 = read from HBM)
    int lds_face_entries;  // dynamic LDS: n_meshes * 6 face-table entries (0 = read FlatMesh from HBM)
    int scene_in_lds;      // 1 when both tables fit the LDS budget (lean kernel variant allowed)
};

Shard make_shard(const mcrt_config& cfg, int first, int step);
// diagnostic builds (-DMCRT_STAMPS) only: per-phase wave-cycle sums of the trace kernel
hipError_t read_phase_stamps(unsigned long long out[16], bool reset);

// workspace requirements (bytes) for a given config + shard
size_t tile_rng_bytes(const RenderParams& p);
size_t hit_rng_bytes(const RenderParams& p);
size_t deep_stack_bytes(const RenderParams& p);
int render_grid_blocks(const RenderParams& p);
// chooses parts_per_tile, the dynamic-LDS split and the grid for p.cfg / p.shard
void fill_launch_geometry(RenderParams& p, int target_units);
constexpr int kAlphaLdsWordsMax = 4096;  // 64 Ki texels
constexpr int kFaceLdsEntriesMax = 384;   // 64 meshes

// enqueue: tile-RNG seeding (if needed) + the trace kernel.  If ev_k0/ev_k1 are non-null they are
// recorded on `stream` immediately around the trace kernel launch.
hipError_t launch_render(const RenderParams& p, hipStream_t stream, hipEvent_t ev_k0, hipEvent_t ev_k1);

hipError_t launch_unpack_rows(const mcrt_config& cfg, const Shard& sh, const float* packed, float* frame,
                              hipStream_t stream);
hipError_t launch_quantize(const float* rgba, uint8_t* out, size_t n_pixels, hipStream_t stream);

// probes
hipError_t launch_probe_intersect(const uint8_t* scene, const float* rays, int n, mcrt_hit* out,
                                  hipStream_t stream);
hipError_t launch_probe_trace(const uint8_t* scene, const mcrt_config& cfg, const float* rays, int n,
                              int depth, float* out, uint32_t* hit_rng, float* deep_stack,
                              hipStream_t stream);
hipError_t launch_probe_mt(const uint32_t* seeds, int n_seeds, int n_draws, float* out, uint32_t* storage,
                           hipStream_t stream);
hipError_t launch_probe_detmath(int op, const float* x, const float* y, size_t n, float* out,
                                hipStream_t stream);
hipError_t launch_probe_detmath_range(int op, uint32_t lo_bits, uint64_t count, float y0, float* out,
                                      hipStream_t stream);

}  // namespace mcrt

#endif
