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
    // MCRT_LAYOUT_PACKED: owned row j is stored as packed tile row pack_first + j * pack_step (a lane
    // of a rank's shard writes into the rank's packed buffer; plain shards use 0 / 1)
    int pack_first;
    int pack_step;
};

// Device workspace of the wavefront pipeline (all HBM; sized for one batch of tile rows in which
// every sample may hit).  A *slot* is one sample of a tile that meshes can touch; a *record* is one hit
// of a chain (the primary hit of a sample and every reflection hit below it).
struct WaveSpace {
    float4* scol;           // [cap] per-sample colour, where the sample did not start a chain (miss, background)
    uint32_t* end;          // [cap] per sample: 0 = the colour is in `scol`; else how its chain ended:
                            //       (records of the chain << 1) | (1 = stopped at maxBounces, 0 = the last reflection ray missed)
    uint4* units;           // units of the tiles meshes can touch: {owned tile, first pixel, end pixel, slot base}
    uint32_t* unit_hits;    // per unit: its primary hits (their records sit at the front of the unit's slot range)
    unsigned long long* tile_mask;  // per owned tile: meshes whose screen bound touches it
    float* tile_draws;      // [touched tiles (bg_in_plan) or all tiles of the batch][draws_stride] a tile's mt19937 draws, as uniform floats
    uint32_t draws_stride;  // tile slots * draws_per_sample
    uint32_t unit_cap;      // capacity of `units`
    uint32_t tile_cap;      // tiles of a batch that may be touched by meshes (host-side superset of the device's culling):
                            // touched tile number k of a batch owns slots [k, k+1) * tile_slots
    // Hit records, SoA.  q_*[1] = q_*[0] + cap.
    // General variants: q_*[level & 1], ping-pong by level, every field written:
    //   q_o ray origin (.w = root = the chain's sample slot, bit-cast), q_d ray direction (.w = depth),
    //   q_p hit point, q_n hit normal (as intersectMesh returns it), q_t texel colour.
    // Flat pipeline: q_*[0] is ONE array of rec_cap records — the primary hits at their slot index (< cap), the
    // records of every deeper level densely from index cap on — in a compact form (36 B instead of 80 for an
    // un-posed scene seen through a pinhole):
    //   q_p hit point (.w = root), q_d ray direction (.w = depth | face axis << 8 | min-side << 10 | exit-face << 11),
    //   q_x texel reference of the face (the colour is fetched from the pool where the record is shaded),
    //   q_n only when the scene holds posed meshes (else the normal follows from the face code),
    //   q_o only for records below the primary hits, and for those too under depth of field (else the camera position)
    float4* q_o[2];
    float4* q_d[2];
    float4* q_p[2];
    float4* q_n[2];
    float4* q_t[2];
    int32_t* q_x;           // [rec_cap] flat pipeline: texel reference per record
    float* targets;         // [rec_cap][3*shadowSamples] light sample positions per record (level 0: reused for the AO directions)
    unsigned long long* cand;  // [rec_cap] per record: meshes its soft-shadow rays can meet (conservative first pass, once per hit)
    uint32_t* lit[2];       // [0]: [rec_cap] visible light samples per record; [1]: [cap] occluded AO samples of the primary
                            // hits (general variants: ping-pong by level parity, [cap] each)
    float4* stack;          // [stack_stride][cap] level colours of the chains, plane-major: depth d of the chain of sample slot r
                            // at [d * cap + r] — the level-0 colours of neighbouring samples share cache lines
    uint32_t* counter_base; // the counters' values when the pass began (the previous pass's `resolve` left them): a pass's counts are differences
    uint32_t* frame_info;   // [0] units of the pass, left by `primary` for `resolve`
    uint32_t* counters;     // [0] units in `units`, [1] touched tiles, [8 + L] entries of level L >= 1
                            // ([9] = level-1 records), [last] touched-tile bound exceeded (never, by construction; sticky);
                            // never cleared: running counts (modulo 2^32) against counter_base
    uint32_t* hit_rng;      // general variants: per-thread 624-word mt19937 states (long streams)
    uint32_t cap;           // slot capacity (= samples of the touched tiles of the largest batch)
    uint32_t tile_slots;    // slots per touched tile: min(tile, width) * min(tile, height) * spp
    int stack_stride;       // maxBounces + 1 (at least 1)
};

// Seeding std::mt19937(seed) for a hit's light samples is a 397-step sequential recurrence whose only product
// the truncated engine needs is the state word mt[397] — a pure function of the 32-bit seed.  The shadow seeds of
// a scene at the reference's scale, (unsigned)(P·(12345, 67890, 11111) + depth·99999) (raytracer.cpp:110-112),
// lie in a window around zero (|sum| < 4 M for the character scene): a table of mt[397] for the seeds
// -2^24 .. 2^24 - 1 (wrapping) is 128 MB per device, built once with the same recurrence in ~2 ms, and replaces
// the chain by one 4-byte load.  Seeds outside the window run the chain.
// integer divisors (frame widths and heights) for which rt::div_frame has been checked exhaustively against the general division
constexpr int kDivFrameMax = 16384;

constexpr uint32_t kSeedWindowHalf = 1u << 24;
constexpr uint32_t kSeedWindow = 1u << 25;  // table entries: seed s at index s + kSeedWindowHalf (mod 2^32)

struct RenderParams {
    const uint8_t* scene;  // flat blob in HBM
    const uint32_t* seed_table;  // kSeedWindow words, or NULL: every hit seeds by the recurrence
    const uint32_t* seed_table_full;  // mt[397] for EVERY 32-bit seed (16 GiB, built on a device's first ambient-occlusion
                                      // render: the AO seeds, raytracer.cpp:122-123, cover the whole range), or NULL
    mcrt_config cfg;
    Shard shard;
    int layout;            // MCRT_LAYOUT_*
    float* out;            // float4 frame or packed rows (may be NULL when out8 is given)
    uint8_t* out8;         // RGBA8 frame or packed rows, quantised in the epilogue (may be NULL)
    uint32_t* tile_rng;    // owned_tiles x stream_parts x 624 mt19937 state words (NULL when no tile draws)
    WaveSpace ws;
    int draws_per_sample;  // 0, 2 or 4
    int stream_waves;      // waves per tile in `plan_tiles` (1, 2 or 4, <= stream_parts; choose_grids)
    int stream_parts;      // a tile's mt19937 stream can be generated in this many parts (1, 2 or 4),
    int stream_part_twists;  //   of this many 624-word twists each, from engine states kept with the tile seeds (tile_rng:
                           //   owned_tiles x stream_parts x 624 words)
    int parts_per_tile;    // a tile that meshes can touch is split into this many pixel-aligned work units
    int lds_alpha_words;   // dynamic LDS: alpha-predicate words staged per workgroup
    int lds_face_entries;  // dynamic LDS: n_meshes * 6 face-table entries
    int scene_in_lds;      // 1 when both tables fit the LDS budget (otherwise the kernels read HBM)
    int scene_posed;       // 1 when any mesh has a rotation (selects the kernels that carry the local-frame path)
    int rows_per_batch;    // owned tile rows per pipeline pass
    int bg_in_plan;        // 1: `plan_tiles` renders the background tiles from the draws in LDS and only touched tiles' draws
                           //    go to HBM; 0 (more than 24 draws per pixel): all streams to HBM, `primary` renders them
    int bg_kernel;         // 1 (bg_in_plan == 0 and many samples per pixel): `background_kernel` renders the background tiles from their
                           //    streams, slab by slab through LDS; 0: `primary`'s tail does, a lane per pixel straight from HBM
    int lit_round;         // `lit`: records per round (a block of 256)
    int lit_pass;          // `lit`: traced records per pass (their light samples live in LDS between two phases)
    int lit_lds_offset;    // `lit`: byte offset of that area in dynamic LDS (behind the scene tables, 16-aligned)
    int lit_lds_bytes;     // `lit`: its size
    int flat;              // 1: flat pipeline (all levels' records shaded at once); 0: general variants, one launch set per level
    float inv_width, inv_height;  // 1.0f / width, 1.0f / height (correctly rounded: formed on the host)
    int div_frame;         // 1: width and height lie in rt::div_frame's verified range (rt_core.h)
    int inside_fast;       // shadow / AO rays: 1 — a candidate whose box holds the ray's origin strictly inside takes
                           //    rt::mesh_candidate_inside (the exit face alone); 0 (MCRT_INSIDE_FAST=0) — the general routine
    int bundle_decisions;  // `lit`: 1 — a hit whose whole bundle of shadow rays is decided (rt::bundle_decide) draws no light
                           //    samples and traces no rays; 0 (MCRT_BUNDLE_DECISIONS=0) — every hit's rays are traced
    int shared_device;     // 1: this render shares the device with others (another lane of its frame, or another handle's frame still
                           //    running when it was enqueued): the grids below are then sized for throughput — fewer, longer-lived
                           //    workgroups per kernel leave room for the other frames' kernels —, otherwise for the frame's own latency
    int grid_primary, grid_ao, grid_lit, grid_resolve;  // workgroup caps of the launches (choose_grids; all kernels stride)
    int rect_x, rect_y, rect_w, rect_h;  // rect_w > 0: the launch renders ONE tile, this rectangle (TileRenderer::renderTile for an
                           //    arbitrary Tile, tile_renderer.cpp:71-127: seed rect_y * width + rect_x, pixels in the rectangle's own
                           //    row-major order); the shard is then one tile row of one tile
};

Shard make_shard(const mcrt_config& cfg, int first, int step);

// Sizes of the workspace arrays for p.cfg / p.shard; fills p.parts_per_tile, p.rows_per_batch and
// p.ws.cap / p.ws.stack_stride.  budget_bytes bounds the per-batch workspace (a batch is never
// smaller than one tile row).
struct WorkspaceBytes {
    size_t tile_rng, tile_draws, scol, end, units, unit_hits, tile_mask, queue_each, texel_refs, targets, cand, lit0, lit1, stack, counters, hit_rng;
};
// row_touched[j]: upper bound of the tiles meshes can touch in owned tile row j (NULL: every tile).
WorkspaceBytes plan_workspace(RenderParams& p, size_t budget_bytes, const int* row_touched);
// fills p.shared_device, p.grid_* and p.stream_waves (MCRT_*_GRID / MCRT_STREAM_WAVES override, development knobs)
void choose_grids(RenderParams& p, bool shared_device, bool company);  // company: other frames or lanes run beside this launch set
constexpr int kAlphaLdsWordsMax = 4096;  // 64 Ki texels
constexpr int kFaceLdsEntriesMax = 384;   // 64 meshes
constexpr int kCounterWords = 4096;

// seeds p.tile_rng: one mt19937 per owned tile (tile_renderer.cpp:78).  A function of the frame width, the
// tile size and the shard only — the caller keeps the result across renders and calls this when those change.
hipError_t launch_seed_tiles(const RenderParams& p, hipStream_t stream);
// enqueue the whole pipeline of one lane on `stream` (p.tile_rng already seeded): per batch of tile rows
// plan → primary (+ the primary hits' reflection rays) → (ao →) lit (the rest of the chains, then light and shade) → resolve
// Optional events for a caller that downloads tile rows as they become final (the one-shot host path):
//  after_plan    recorded behind the first pass's plan_tiles: with bg_in_plan every tile row that holds no touched
//                tile is complete then
//  batch_done[b] recorded behind pass b's resolve: the rows of that pass are complete (n_batch_done entries, may be 0)
struct LaunchMarks {
    hipEvent_t after_plan = nullptr;
    hipEvent_t* batch_done = nullptr;
    int n_batch_done = 0;
};
hipError_t launch_render(const RenderParams& p, hipStream_t stream, const LaunchMarks* marks = nullptr);

hipError_t launch_unpack_rows(const mcrt_config& cfg, const Shard& sh, const float* packed, float* frame,
                              hipStream_t stream);
hipError_t launch_unpack_rows8(const mcrt_config& cfg, const Shard& sh, const uint8_t* packed, uint8_t* frame, hipStream_t stream);  // RGBA8 plane
hipError_t launch_assemble_frame(const mcrt_config& cfg, int world, const float* gathered, size_t rank_stride_pixels,
                                 float* frame, hipStream_t stream);
hipError_t launch_quantize(const float* rgba, uint8_t* out, size_t n_pixels, hipStream_t stream);
// fills table[i] = mt[397] of std::mt19937(i - kSeedWindowHalf) for i < kSeedWindow
hipError_t launch_build_seed_table(uint32_t* table, hipStream_t stream);
// fills table[s] = mt[397] of std::mt19937(s) for the seeds first .. first + count - 1 (count a multiple of 256)
hipError_t launch_build_seed_table_range(uint32_t* table, uint32_t first, uint32_t count, hipStream_t stream);
// host_reciprocals: d_count floats in device memory, 1.0f / d as the host rounds it (what the render kernels get)
hipError_t launch_probe_div_const(uint32_t d_first, uint32_t d_count, int mode, const float* host_reciprocals, unsigned long long* counts, hipStream_t stream);

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
