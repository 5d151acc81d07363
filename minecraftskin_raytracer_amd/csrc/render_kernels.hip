// render_kernels.hip — hand-written HIP kernels for gfx950 (MI355X / CDNA4).
//
// The hot path of the reference, TileRenderer::renderTile (tile_renderer.cpp:71-127) with
// RayTracer::traceRay underneath, as a wavefront pipeline: every stage is a small, high-occupancy
// kernel over a dense work list, so the latency-bound pieces (the 397-step mt19937 seeding chain,
// the dependent slab tests of a shadow ray) are hidden by many resident waves instead of stalling a
// workgroup, and hit work is spread over the whole chip no matter which tiles hold the character.
//
//   seed_tiles      1 lane / tile     mt19937 seeding of the per-tile jitter streams (kept across renders: the
//                                     seeds depend on the frame width, the tile size and the shard only)
//   plan_tiles      1 wave / tile     which meshes' screen bounds touch the tile (a tile nothing can touch
//                                     is rendered whole as background; others are split into pixel-aligned
//                                     units with colour slots assigned), then every draw of the tile's
//                                     mt19937 stream — twisted inside the wave, no workgroup barrier — to HBM
//   primary         persistent WGs    touched units: thread per sample, its draws, camera/lens ray, closest
//                                     hit over the tile's mesh mask; misses write their sample colour, hits
//                                     become records at the front of the unit's slot range.  (Background tiles only when
//                                     a pixel takes more than 24 draws: thread per pixel, gradient, ordered sample sum)
//                                     ... and, packed on the block's first threads, the reflection ray of every
//                                     primary hit (geometry only: direction, hit point, normal) → closest hit →
//                                     level-1 record; chains that end are marked
//   ... then ONCE over the records of ALL levels (320 k + 31 k + 3 k + ... at 1080p / 4 spp):
//   ao              (AO on) 1 lane / primary hit   the meshes its hemisphere of rays can meet; mt19937(ao seed), the A
//                                     cosine-weighted directions and their any-hit tests within the radius, in registers
//   lit             first, per block of 256 level-1 records (1 lane / record): the rest of the chain — reflect, closest
//                                     hit, append, until the ray misses or maxBounces is reached (~10 % go on per level) —
//                                     into a region of the queue the block owns, lit and shaded by the same workgroup;
//                   then phases per block of 256 records, handed over through LDS:
//                   1 lane / record   the hit's bundle mask (meshes its shadow rays can meet) and the whole-bundle
//                                     decision (rt::bundle_classify): most hits are provably lit by all S light
//                                     samples or by none and need no samples and no rays; the rest are packed
//                   1 lane / undecided record   register-only truncated mt19937 → 2·S draws → the S disk sample positions
//                   1 lane / (undecided record, light sample)   exact any-hit test on the meshes left open → lit count
//                   1 lane / record   Blinn-Phong (+AO) → the chain's stack of level colours
//   resolve         1 lane / pixel    folds each sample's chain back to front, ordered sum of the pixel's sample
//                                     colours (float addition order is part of the result), coalesced float4 / RGBA8 store
//   (general variants — per-hit RNG streams longer than 227 draws, or more than kFlatMaxBounces bounces — run
//   light_samples / shadow / level_shade once per recursion level instead of lit, and `primary` traces no
//   reflection rays)
// Records live in HBM as SoA float4 arrays.  Every unit owns a fixed slot range (its samples); its
// primary hits are compacted to the front of that range with an LDS prefix sum and a per-unit count —
// NO global atomics on the hot path (a returning atomic on one word sustains only ~88 ops/us on this
// chip; per-wave queue claims made `primary` atomic-bound); deeper records take one atomic per 256.
// A frame is cut into batches of tile rows so that the worst case (every sample of a touched tile hits) fits.
// No MFMA: there is no dense contraction anywhere on this path.
#include "kernels.h"
#include "rt_core.h"

// Verification hooks.  tools/decide_check.sh builds a variant of the library with -DMCRT_KERNEL_HOOKS='"decide_check_hooks.h"'
// (tools/decide_check_hooks.h: every record `lit` decides is traced as well and contradictions are counted and printed);
// the product build compiles the hooks to nothing.
#ifdef MCRT_KERNEL_HOOKS
#include MCRT_KERNEL_HOOKS
#else
#define MCRT_HOOK_LIT_SHARED
#define MCRT_HOOK_LIT_CLASSIFIED(known, undecided, cand, O)
#define MCRT_HOOK_LIT_SHADED(lit, r)
#define MCRT_HOOK_RESOLVE_BEGIN()
#endif

namespace mcrt {

using namespace rt;

constexpr int kBlock = 256;
constexpr int kChunk = 256;        // work items per chunk: one per thread
#ifndef MCRT_PRIMARY_GRID
#define MCRT_PRIMARY_GRID 1280
#endif
constexpr int kPrimaryGrid = MCRT_PRIMARY_GRID; // persistent primary workgroups (5 per CU: the kernel is built for 5 waves per SIMD)
constexpr int kQueueGrid = 2048;   // workgroups of the queue kernels (grid-stride over device-side counts)
constexpr int kLitGridAlone = 4096;  // `lit` of a frame that has the device to itself
constexpr int kResolveGrid = 4096;
constexpr int kSharedGrid = 896;   // every kernel of a frame that shares the device (choose_grids): 3.5 workgroups per CU

// n / d for a divisor that is the same for the whole wave: a shift when it is a power of two (tile widths of 32,
// 4 samples per pixel: the usual case) instead of the ~25-instruction expansion of a 32-bit division.
struct UDiv {
    unsigned d;
    int shift;  // log2(d), or -1
    __device__ __forceinline__ explicit UDiv(unsigned dv) : d(dv), shift((dv & (dv - 1u)) == 0u && dv != 0u ? static_cast<int>(__builtin_ctz(dv)) : -1) {}
    __device__ __forceinline__ unsigned div(unsigned n) const { return shift >= 0 ? n >> shift : n / d; }
};

// (px + jitter) / width and (py + jitter) / height of a sample (tile_renderer.cpp:88-89): rt::div_frame where the
// frame's size lies in its verified range, the general division otherwise
struct FrameDiv {
    float w, h, rw, rh;
    bool fast;
    __device__ __forceinline__ explicit FrameDiv(const RenderParams& p)
        : w(static_cast<float>(p.cfg.width)), h(static_cast<float>(p.cfg.height)), rw(p.inv_width), rh(p.inv_height), fast(p.div_frame != 0) {}
    __device__ __forceinline__ float u(float x) const { return fast ? div_frame(x, w, rw) : x / w; }
    __device__ __forceinline__ float v(float y) const { return fast ? div_frame(y, h, rh) : y / h; }
};

// ---------------------------------------------------------------------------------------------
// tile geometry helpers (TileRenderer::generateTiles, tile_renderer.cpp:18-39)
// ---------------------------------------------------------------------------------------------
struct TileGeom {
    int x, y, w, h;
    int owned_row;  // index of this tile's row among the rows this launch owns
};

__device__ __forceinline__ TileGeom tile_of(const RenderParams& p, int owned_tile) {
    TileGeom t;
    if (p.rect_w > 0) {  // the one tile of a renderTile call: any rectangle of the frame
        t.x = p.rect_x, t.y = p.rect_y, t.w = p.rect_w, t.h = p.rect_h;
        t.owned_row = 0;
        return t;
    }
    int k = owned_tile / p.shard.tiles_x;
    int tx = owned_tile - k * p.shard.tiles_x;
    int ty = p.shard.first + k * p.shard.step;
    int ts = p.cfg.tile_size;
    t.x = tx * ts;
    t.y = ty * ts;
    t.w = min(ts, p.cfg.width - t.x);
    t.h = min(ts, p.cfg.height - t.y);
    t.owned_row = k;
    return t;
}

// ---------------------------------------------------------------------------------------------
// pre-pass: seed one std::mt19937 per owned tile (tile_renderer.cpp:78).  The seeding
// recurrence is strictly sequential, so it is spread over lanes (one tile per lane).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void seed_tiles_kernel(RenderParams p, int n_tiles) {
    int tile = blockIdx.x * 64 + threadIdx.x;
    if (tile >= n_tiles) return;
    TileGeom t = tile_of(p, tile);
    uint32_t x = static_cast<uint32_t>(t.y * p.cfg.width + t.x);
    uint32_t* dst = p.tile_rng + static_cast<size_t>(tile) * p.stream_parts * 624;
    dst[0] = x;
    for (uint32_t j = 1; j < 624; ++j) {
        x = mt_step(x, j);
        dst[j] = x;
    }
}

// ---------------------------------------------------------------------------------------------
// pixel store: the float4 frame and/or its RGBA8 quantisation `(u8)(clamp(c,0,1)*255+0.5)`
// (image_writer.cpp:18-22 ≡ image.cpp:31-36) — the epilogue of `primary` (background tiles) and `resolve`
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uchar4 quantize_pixel(float4 c) {
    uchar4 q;
    q.x = static_cast<unsigned char>(sclamp(c.x, 0.0f, 1.0f) * 255.0f + 0.5f);
    q.y = static_cast<unsigned char>(sclamp(c.y, 0.0f, 1.0f) * 255.0f + 0.5f);
    q.z = static_cast<unsigned char>(sclamp(c.z, 0.0f, 1.0f) * 255.0f + 0.5f);
    q.w = static_cast<unsigned char>(sclamp(c.w, 0.0f, 1.0f) * 255.0f + 0.5f);
    return q;
}
__device__ __forceinline__ void store_pixel(float4* __restrict__ out_frame, uchar4* __restrict__ out8, size_t idx, float4 v) {
    if (out_frame) out_frame[idx] = v;
    if (out8) out8[idx] = quantize_pixel(v);
}

// ---------------------------------------------------------------------------------------------
// tile_streams: every draw renderTile takes from a tile's mt19937 (tile_renderer.cpp:78-99: per pixel
// in row-major order, per sample, 2 jitter draws if spp > 1, then 2 lens draws if DOF is on).  One
// WAVE per tile: the engine's twist runs inside the wave on a ping-pong state in LDS — LDS
// operations of one wave are ordered, a wave barrier keeps the compiler from moving them — so
// there is no workgroup barrier anywhere and all tiles of a batch advance in parallel.
//  * A tile that meshes can touch: the draws go to HBM as uniform floats, for `primary`.
//  * A background tile (93 % of the tiles of the metric frame): nothing can be hit, no ray is needed —
//    the wave renders the tile itself, straight from the draws in LDS: after each twist the pixels whose
//    draws are complete (they lie in the new 624 words and the previous 624, both still in LDS) get
//    their gradient samples, the ordered sample sum (tile_renderer.cpp:111-124) and a coalesced store.
//    Writing those streams out and reading them back in `primary` was 134 MB of the frame's HBM traffic.
//    Pays while a twist completes enough pixels to fill the wave's lanes (spp * draws per sample <= 24:
//    RenderParams::bg_in_plan); at higher sample counts every tile's stream goes to HBM and `primary`
//    renders the background tiles with a thread per pixel.
// (Generating the stream inside `primary` with block-wide twists made that kernel barrier-bound: 3
// barriers per 624 draws, ~60 us of the 1080p / 4 spp frame, milliseconds at 64 spp.)
// ---------------------------------------------------------------------------------------------
constexpr int kStreamWaves = 4;  // tiles per workgroup
// the pixels [lo, hi) of a background tile; draw(g, jx, jy) = draws number g, g + 1 of the tile's stream (g even)
template <class DrawFn>
__device__ __forceinline__ void background_pixels(const SceneView& sc, const RenderParams& p, const TileGeom& tg, float4* __restrict__ out_frame,
                                                  uchar4* __restrict__ out8, unsigned lo, unsigned hi, int lane, DrawFn&& draw) {
    const mcrt_config& cfg = p.cfg;
    const int spp = cfg.samples_per_pixel > 1 ? cfg.samples_per_pixel : 1;
    const unsigned dd = static_cast<unsigned>(p.draws_per_sample);
    const FrameDiv fd(p);
    const float inv_spp = 1.0f / static_cast<float>(spp);
    const UDiv by_w(static_cast<unsigned>(tg.w));
    for (unsigned pix = lo + static_cast<unsigned>(lane); pix < hi; pix += 64u) {
        const unsigned uly = by_w.div(pix);
        const int ly = static_cast<int>(uly);
        const int lx = static_cast<int>(pix - uly * static_cast<unsigned>(tg.w));
        const float fx = static_cast<float>(tg.x + lx), fy = static_cast<float>(tg.y + ly);
        const unsigned g0 = pix * static_cast<unsigned>(spp) * dd;  // (a tile's stream is shorter than 2^32 draws: plan_workspace)
        float ar = 0.0f, ag = 0.0f, ab = 0.0f, aa = 0.0f;
        for (int sidx = 0; sidx < spp; ++sidx) {
            float jx = 0.5f, jy = 0.5f;
            if (spp > 1) draw(g0 + static_cast<unsigned>(sidx) * dd, jx, jy);  // the sample's two jitter draws: one aligned pair of the stream
            const C4 c = background(sc, cfg, fd.u(fx + jx), fd.v(fy + jy));  // tile_renderer.cpp:111-114
            ar += c.r;
            ag += c.g;
            ab += c.b;
            aa += c.a;
        }
        const int row = (p.layout == MCRT_LAYOUT_PACKED) ? ((p.shard.pack_first + tg.owned_row * p.shard.pack_step) * cfg.tile_size + ly) : (tg.y + ly);
        store_pixel(out_frame, out8, static_cast<size_t>(row) * cfg.width + (tg.x + lx),
                    make_float4(ar * inv_spp, ag * inv_spp, ab * inv_spp, aa * inv_spp));
    }
}

// A background tile whose every sample has the same colour needs no draws at all: the flat background colour
// (gradient off, raytracer.cpp:32-33), or the gradient's edge colour when the whole tile lies where the reference's
// `dist` clamps to 1 — sqrt(cx² + cy²)·2·gradientScale >= 1 for every point a jittered sample of the tile can take
// (raytracer.cpp:19-24; 18 % of the metric frame's tiles, its corners).  The test uses the tile's rectangle in u,v with a
// margin of 1e-4, a hundred times the float error of the reference's expression; the colour and the sample sum are
// formed by the reference's own operations (c = center·(1 - t) + edge·t with t = 1·1; spp additions; · 1/spp).
__device__ __forceinline__ bool constant_background(const SceneView& sc, const RenderParams& p, const TileGeom& tg, float4& pixel) {
    const mcrt_config& cfg = p.cfg;
    C4 c;
    if (cfg.gradient_bg) {
        const float fW = static_cast<float>(cfg.width), fH = static_cast<float>(cfg.height);
        const float ulo = static_cast<float>(tg.x) / fW, uhi = static_cast<float>(tg.x + tg.w) / fW;
        const float vlo = static_cast<float>(tg.y) / fH, vhi = static_cast<float>(tg.y + tg.h) / fH;
        const float cx = (ulo <= 0.5f && 0.5f <= uhi) ? 0.0f : fminf(fabsf(ulo - 0.5f), fabsf(uhi - 0.5f));
        const float cy = (vlo <= 0.5f && 0.5f <= vhi) ? 0.0f : fminf(fabsf(vlo - 0.5f), fabsf(vhi - 0.5f));
        if (!(__builtin_sqrtf(cx * cx + cy * cy) * 2.0f * cfg.gradient_scale >= 1.0f + 1e-4f)) return false;
        const float dist = 1.0f;  // sclamp(dist, 0, 1)
        const float t = dist * dist;
        c.r = cfg.bg_center[0] * (1.0f - t) + cfg.bg_edge[0] * t;
        c.g = cfg.bg_center[1] * (1.0f - t) + cfg.bg_edge[1] * t;
        c.b = cfg.bg_center[2] * (1.0f - t) + cfg.bg_edge[2] * t;
        c.a = 1.0f;
    } else {
        const float* b = sc.hdr->background;
        c = C4{b[0], b[1], b[2], b[3]};
    }
    const int spp = cfg.samples_per_pixel > 1 ? cfg.samples_per_pixel : 1;
    const float inv_spp = 1.0f / static_cast<float>(spp);
    float ar = 0.0f, ag = 0.0f, ab = 0.0f, aa = 0.0f;
    for (int sidx = 0; sidx < spp; ++sidx) {  // tile_renderer.cpp:116-124
        ar += c.r;
        ag += c.g;
        ab += c.b;
        aa += c.a;
    }
    pixel = make_float4(ar * inv_spp, ag * inv_spp, ab * inv_spp, aa * inv_spp);
    return true;
}
// the tile's pixels get `pixel`; `nthreads` threads (a wave, or a workgroup), this one being number `tid`
__device__ __forceinline__ void fill_tile(const RenderParams& p, const TileGeom& tg, float4* __restrict__ out_frame, uchar4* __restrict__ out8, float4 pixel,
                                          unsigned tid, unsigned nthreads) {
    const unsigned npix = static_cast<unsigned>(tg.w) * static_cast<unsigned>(tg.h);
    const UDiv by_w(static_cast<unsigned>(tg.w));
    for (unsigned pix = tid; pix < npix; pix += nthreads) {
        const unsigned uly = by_w.div(pix);
        const int ly = static_cast<int>(uly);
        const int lx = static_cast<int>(pix - uly * static_cast<unsigned>(tg.w));
        const int row = (p.layout == MCRT_LAYOUT_PACKED) ? ((p.shard.pack_first + tg.owned_row * p.shard.pack_step) * p.cfg.tile_size + ly) : (tg.y + ly);
        store_pixel(out_frame, out8, static_cast<size_t>(row) * p.cfg.width + (tg.x + lx), pixel);
    }
}

// executed by one wave; `st` = its 2 x 624 words of LDS.  dst != nullptr: the tile's draws go there;
// dst == nullptr: a background tile, rendered from the draws in LDS.
// jitter_only (a background tile under depth of field, 4 draws per sample): only the samples' jitter pairs are stored,
// packed — sample k's at dst[2k], dst[2k + 1]; no ray leaves a background tile, so its lens draws are never read
// (tile_renderer.cpp:99-114), and they are half of the stream `primary` would read back.
//
// A tile's stream is cut into `stream_parts` PARTS of `stream_part_twists` twists each: the engine's state at the start
// of every part — after 0, q, 2q, ... twists — is a function of the tile's seed alone and is kept with the tile seeds
// (advance_tiles_kernel), so `stream_waves` waves (1, 2 or 4, each taking stream_parts / stream_waves consecutive parts)
// can advance side by side instead of ONE wave running the whole chain of dependent twists (13 at 1080p / 4 spp, 210 at
// 64 spp: `plan_tiles` alone on the device is the latency of that chain at two waves per SIMD).  A wave writes its own
// draws, or — background tile — renders the pixels whose FIRST draw lies in its parts; a pixel whose draws run over their
// end is finished by one more twist of the same wave (the next wave makes the same words again).
__device__ __forceinline__ void tile_stream_wave(const SceneView& sc, const uint32_t* __restrict__ tile_rng, float* __restrict__ dst,
                                                 float4* __restrict__ out_frame, uchar4* __restrict__ out8, const RenderParams& p,
                                                 const TileGeom& tg, int tile, int part, int n_parts, uint32_t* st, int lane, bool jitter_only = false) {
    const int spp = p.cfg.samples_per_pixel > 1 ? p.cfg.samples_per_pixel : 1;
    const unsigned npix = static_cast<unsigned>(tg.w) * static_cast<unsigned>(tg.h);
    const unsigned per_pixel = static_cast<unsigned>(spp) * static_cast<unsigned>(p.draws_per_sample);
    const unsigned total = npix * per_pixel;  // < 2^32 - 2^16 (plan_workspace refuses longer streams): 32-bit index arithmetic throughout
    // this wave: parts [part, part + n_parts) — it starts from the state kept for `part` and simply twists on through the others
    const unsigned part_draws = static_cast<unsigned>(p.stream_part_twists) * 624u;
    const unsigned first_draw = static_cast<unsigned>(part) * part_draws;  // (parts x part_draws covers a full tile's stream: no overflow)
    if (first_draw >= total) return;  // a clipped tile's stream ends before this part
    const unsigned wave_draws = static_cast<unsigned>(n_parts) * part_draws;
    const bool last_part = part + n_parts >= p.stream_parts || first_draw + wave_draws >= total;
    const unsigned end_draw = last_part ? total : first_draw + wave_draws;
    // background tile: the pixels whose first draw lies in [first_draw, end_draw)
    const unsigned pix_lo = (first_draw + per_pixel - 1u) / per_pixel;
    const unsigned pix_hi = last_part ? npix : (end_draw + per_pixel - 1u) / per_pixel;
    const uint32_t* src = tile_rng + (static_cast<size_t>(tile) * p.stream_parts + part) * 624;
    for (int e = lane; e < 624; e += 64) st[e] = src[e];
    auto wave_sync = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    wave_sync();
    int cur = 0;
    unsigned pixels_done = pix_lo;
    for (unsigned done = first_draw; dst ? done < end_draw : pixels_done < pix_hi; done += 624u) {  // (a pixel's draws end before `total`)
        const uint32_t* o = st + cur * 624;
        uint32_t* n = st + (cur ^ 1) * 624;
        // mt19937 twist: new[k] from old[k], old[k+1] and old[k+397] (= new[k-227] once k >= 227); the
        // loops are unrolled so that a phase's LDS reads are all in flight before its first write
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = lane + 64 * i;
            if (k < 227) n[k] = mt_twist(o[k], o[k + 1], o[k + 397]);
        }
        wave_sync();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = 227 + lane + 64 * i;
            if (k < 454) n[k] = mt_twist(o[k], o[k + 1], n[k - 227]);
        }
        wave_sync();
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int k = 454 + lane + 64 * i;
            if (k < 624) n[k] = mt_twist(o[k], (k == 623) ? n[0] : o[k + 1], n[k - 227]);
        }
        wave_sync();
        cur ^= 1;
        const unsigned left = total - done;
        const int m = left < 624u ? static_cast<int>(left) : 624;
        if (dst && jitter_only) {  // wave-uniform; done is a multiple of 624 = 4 x 156: draw done + e is a jitter draw iff (e & 2) == 0
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int k = lane + 64 * i;  // the k-th jitter draw of this twist
                const int e = ((k >> 1) << 2) | (k & 1);
                if (e < m) dst[(done >> 1) + static_cast<unsigned>(k)] = mt_to_unit(mt_temper(n[e]));
            }
        } else if (dst) {
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                const int e = lane + 64 * i;
                if (e < m) dst[done + e] = mt_to_unit(mt_temper(n[e]));
            }
        } else {
            // draws [done, done + m) are in n, the 624 before them in o (the previous twist's words).  Pixels
            // go in whole rounds of 64 (a twist completes only 624 / (2 spp) of them — 78 at 4 spp — and a
            // partial round costs as much as a full one); the rest waits for the next twist, except those
            // whose draws reach back into o, which that twist overwrites.
            const unsigned complete = min(pix_hi, (done + static_cast<unsigned>(m)) / per_pixel);
            const unsigned must_end = min(complete, (done + per_pixel - 1u) / per_pixel);  // first draw before `done`
            const unsigned pending = complete - pixels_done;
            unsigned take = (complete == pix_hi) ? pending : (pending / 64u) * 64u;
            if (pixels_done + take < must_end) take = must_end - pixels_done;
            if (take > 0u) {
                // (g and done are even: the pair is 8-byte aligned in either buffer — one LDS read instead of two, half the
                // cycles of the 8-lanes-per-bank stride a lane per pixel reads the stream with)
                background_pixels(sc, p, tg, out_frame, out8, pixels_done, pixels_done + take, lane, [&](unsigned g, float& jx, float& jy) __attribute__((always_inline)) {
                    const uint2 w = *reinterpret_cast<const uint2*>((g >= done) ? n + (g - done) : o + (g + 624u - done));
                    jx = mt_to_unit(mt_temper(w.x));
                    jy = mt_to_unit(mt_temper(w.y));
                });
                pixels_done += take;
                wave_sync();  // the next twist overwrites o
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// primary-ray culling mask of a tile
// ---------------------------------------------------------------------------------------------
// lens_pad: how far a thin-lens ray can displace the image of a point of this mesh, in the bound's
// units (0 for the pinhole camera)
__device__ __forceinline__ bool mesh_touches_tile(const FlatMesh& m, const TileGeom& t, const mcrt_config& cfg,
                                                  float aspect, float lens_pad) {
    float u0 = m.screen[0], v0 = m.screen[1], u1 = m.screen[2], v1 = m.screen[3];
    if (u0 > u1) return true;  // no bound available
    u0 -= lens_pad, v0 -= lens_pad, u1 += lens_pad, v1 += lens_pad;
    const float W = static_cast<float>(cfg.width), H = static_cast<float>(cfg.height);
    // tile extent padded by 2 pixels, in the bound's units (x: (2u-1)*aspect, y: 1-2v, +y up)
    float tu0 = (2.0f * (static_cast<float>(t.x) - 2.0f) / W - 1.0f) * aspect - 1e-3f * aspect - 1e-3f;
    float tu1 = (2.0f * (static_cast<float>(t.x + t.w) + 2.0f) / W - 1.0f) * aspect + 1e-3f * aspect + 1e-3f;
    float tv1 = 1.0f - 2.0f * (static_cast<float>(t.y) - 2.0f) / H + 2e-3f;
    float tv0 = 1.0f - 2.0f * (static_cast<float>(t.y + t.h) + 2.0f) / H - 2e-3f;
    return !(u1 < tu0 || u0 > tu1 || v1 < tv0 || v0 > tv1);
}

// ---------------------------------------------------------------------------------------------
// scene tables staged in LDS: what candidates index PER LANE (face → texture table, alpha bits)
// ---------------------------------------------------------------------------------------------
struct LdsTables {
    const MCRT_LDS uint32_t* abits;
    const MCRT_LDS int* faces;
    const MCRT_LDS float* mtab;
};
// dyn = dynamic LDS base; layout [face table: 4 ints per (mesh, face)][mesh table: kMeshTabWords per
// mesh][alpha words].  Collective.
__device__ __forceinline__ LdsTables stage_tables(const SceneView& g, const RenderParams& p, unsigned char* dyn) {
    int* s_faces = reinterpret_cast<int*>(dyn);
    float* s_mtab = reinterpret_cast<float*>(dyn + static_cast<size_t>(p.lds_face_entries) * 16);
    const int n_meshes = p.lds_face_entries / 6;
    uint32_t* s_abits = reinterpret_cast<uint32_t*>(dyn + static_cast<size_t>(p.lds_face_entries) * 16 +
                                                    static_cast<size_t>(n_meshes) * kMeshTabWords * 4);
    for (int i = threadIdx.x; i < p.lds_alpha_words; i += blockDim.x) s_abits[i] = g.abits[i];
    for (int i = threadIdx.x; i < p.lds_face_entries; i += blockDim.x) {
        const FlatMesh& fm = g.meshes[i / 6];
        const int f = i - (i / 6) * 6;
        s_faces[4 * i + 0] = fm.tex_off[f];
        s_faces[4 * i + 1] = fm.tex_w[f];
        s_faces[4 * i + 2] = fm.tex_h[f];
        s_faces[4 * i + 3] = 0;
    }
    for (int i = threadIdx.x; i < n_meshes; i += blockDim.x) {
        const FlatMesh& fm = g.meshes[i];
        float* t = s_mtab + i * kMeshTabWords;
        t[0] = fm.lo[0], t[1] = fm.lo[1], t[2] = fm.lo[2];
        t[3] = fm.hi[0], t[4] = fm.hi[1], t[5] = fm.hi[2];
        t[6] = __uint_as_float(fm.flags);
        t[7] = 0.0f;
        t[8] = fm.pivot[0], t[9] = fm.pivot[1], t[10] = fm.pivot[2];
        t[11] = 0.0f;
        t[12] = fm.inv_z_cos, t[13] = fm.inv_z_sin, t[14] = fm.inv_x_cos, t[15] = fm.inv_x_sin;
        t[16] = fm.fwd_x_cos, t[17] = fm.fwd_x_sin, t[18] = fm.fwd_z_cos, t[19] = fm.fwd_z_sin;
        t[20] = fm.sphere[0], t[21] = fm.sphere[1], t[22] = fm.sphere[2], t[23] = fm.sphere[3];
    }
    __syncthreads();
    return LdsTables{(const MCRT_LDS uint32_t*)s_abits, (const MCRT_LDS int*)s_faces, (const MCRT_LDS float*)s_mtab};
}
// kernel variants by scene: kViewHbm — tables too large for LDS (reads HBM; any pose);
// kViewLds — tables in LDS, posed meshes present; kViewLdsUnposed — tables in LDS, no posed mesh
constexpr int kViewHbm = 0, kViewLds = 1, kViewLdsUnposed = 2;
template <int kView>
struct ViewSel {
    using type = SceneViewLdsT<kView == kViewLds>;
    static __device__ __forceinline__ type make(const SceneView& g, const RenderParams& p, unsigned char* dyn) {
        LdsTables t = stage_tables(g, p, dyn);
        return view_with_lds<kView == kViewLds>(g, t.abits, t.faces, t.mtab);
    }
};
template <>
struct ViewSel<kViewHbm> {
    using type = SceneView;
    static __device__ __forceinline__ type make(const SceneView& g, const RenderParams&, unsigned char*) { return g; }
};

// ---------------------------------------------------------------------------------------------
// queue helpers
// ---------------------------------------------------------------------------------------------
// counters[0]: number of planned units (one atomic add per touched tile, in plan_tiles)
constexpr int kCntDense = 8;              // general variants: counters[kCntDense + L] = entries of level L >= 1
constexpr int kCntDeep1 = kCntDense + 1;  // flat pipeline: level-1 records
// WaveSpace::end of a sample whose chain has ended: (records of the chain << 1) | stopped at maxBounces
constexpr uint32_t kEndMiss = 0xffffffffu;  // the primary ray missed: the sample's colour is the background at its jittered position
__device__ __forceinline__ uint32_t chain_code(int records, bool stopped_at_max) {
    return (static_cast<uint32_t>(records) << 1) | (stopped_at_max ? 1u : 0u);
}
constexpr int kCntUnits = 0;
constexpr int kCntTiles = 1;     // touched tiles of the batch so far
constexpr int kCntOverflow = kCounterWords - 1;  // set if more tiles are touched than the host planned for (a bug: the host
                                                 // bound is a superset); sticky

// The pass counters are never cleared: they run on from pass to pass (modulo 2^32), and what a pass has counted is the
// difference to `counter_base`, the snapshot `resolve` — the last kernel of every pass, in which nothing counts any more —
// takes for the pass that follows.  (A memset per pass was a dispatch of its own at the head of every frame: 3 us and a
// launch of a lone frame's 195.)  `resolve` itself reads the one count it needs from frame_info, which `primary` leaves.
__device__ __forceinline__ uint32_t counted(const WaveSpace& ws, int i) { return ws.counters[i] - ws.counter_base[i]; }
__device__ __forceinline__ uint32_t count_add(const WaveSpace& ws, int i, uint32_t n) { return atomicAdd(&ws.counters[i], n) - ws.counter_base[i]; }

// Rank of this thread's item among the workgroup's flagged items, and their total: ballot per wave,
// four wave counts through LDS.  Collective (two barriers: the counts are reusable right after).
__device__ __forceinline__ int block_rank(bool flag, int* s_wcnt, int& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long bal = __ballot(flag);
    if (lane == 0) s_wcnt[wave] = __popcll(bal);
    __syncthreads();
    int before = 0;
    total = 0;
#pragma unroll
    for (int wv = 0; wv < kBlock / 64; ++wv) {
        const int c = s_wcnt[wv];
        if (wv < wave) before += c;
        total += c;
    }
    __syncthreads();
    return before + __popcll(bal & ((1ull << lane) - 1ull));
}

__device__ __forceinline__ void push_entry(const WaveSpace& ws, int parity, uint32_t e, const Ray& ray, const Hit& hit,
                                           uint32_t root, int depth) {
    ws.q_o[parity][e] = make_float4(ray.o.x, ray.o.y, ray.o.z, __uint_as_float(root));
    ws.q_d[parity][e] = make_float4(ray.d.x, ray.d.y, ray.d.z, __int_as_float(depth));
    ws.q_p[parity][e] = make_float4(hit.p.x, hit.p.y, hit.p.z, 0.0f);
    ws.q_n[parity][e] = make_float4(hit.n.x, hit.n.y, hit.n.z, 0.0f);
    ws.q_t[parity][e] = make_float4(hit.tex.r, hit.tex.g, hit.tex.b, hit.tex.a);
}

// ---- compact hit records of the flat pipeline (kernels.h: WaveSpace::q_*) -------------------------------
// .w of q_d: depth | face axis << 8 | min-side << 10 | exit-face << 11
__device__ __forceinline__ uint32_t record_code(int depth, const Hit& h) {
    return static_cast<uint32_t>(depth) | (static_cast<uint32_t>(h.axis) << 8) | (h.neg ? 1u << 10 : 0u) | (h.back ? 1u << 11 : 0u);
}
// with_origin: records below the primary hits, and primary hits under depth of field (otherwise the ray
// starts at the camera position)
__device__ __forceinline__ void push_record(const WaveSpace& ws, bool posed, uint32_t e, const Ray& ray, const Hit& hit, uint32_t root, int depth,
                                            bool with_origin) {
    ws.q_p[0][e] = make_float4(hit.p.x, hit.p.y, hit.p.z, __uint_as_float(root));
    ws.q_d[0][e] = make_float4(ray.d.x, ray.d.y, ray.d.z, __uint_as_float(record_code(depth, hit)));
    ws.q_x[e] = hit.texel;
    if (posed) ws.q_n[0][e] = make_float4(hit.n.x, hit.n.y, hit.n.z, 0.0f);
    if (with_origin) ws.q_o[0][e] = make_float4(ray.o.x, ray.o.y, ray.o.z, 0.0f);
}
// what the chain and shadow stages read of a record: hit point, normal (as intersectMesh returned it),
// ray direction, depth, root
struct RecordGeom {
    V3 p, n, d;
    uint32_t root;
    int depth;
};
__device__ __forceinline__ RecordGeom load_geom(const WaveSpace& ws, bool posed, uint32_t e) {
    const float4 qp = ws.q_p[0][e], qd = ws.q_d[0][e];
    const uint32_t code = __float_as_uint(qd.w);
    RecordGeom g;
    g.p = mk(qp.x, qp.y, qp.z);
    g.d = mk(qd.x, qd.y, qd.z);
    g.root = __float_as_uint(qp.w);
    g.depth = static_cast<int>(code & 0xffu);
    if (posed) {
        const float4 qn = ws.q_n[0][e];
        g.n = mk(qn.x, qn.y, qn.z);
    } else {
        g.n = unposed_normal(static_cast<int>((code >> 8) & 3u), ((code >> 10) & 1u) != 0u, ((code >> 11) & 1u) != 0u);
    }
    return g;
}
// only point and normal (a shadow or AO ray lane)
__device__ __forceinline__ void load_point_normal(const WaveSpace& ws, bool posed, uint32_t e, V3& P, V3& N) {
    const float4 qp = ws.q_p[0][e];
    P = mk(qp.x, qp.y, qp.z);
    if (posed) {
        const float4 qn = ws.q_n[0][e];
        N = mk(qn.x, qn.y, qn.z);
    } else {
        const uint32_t code = __float_as_uint(ws.q_d[0][e].w);
        N = unposed_normal(static_cast<int>((code >> 8) & 3u), ((code >> 10) & 1u) != 0u, ((code >> 11) & 1u) != 0u);
    }
}

// ---------------------------------------------------------------------------------------------
// plan_tiles: 1, 2 or 4 waves per tile of the batch — which meshes can touch the tile (tile_mesh_mask), its units and
// slot range (plan_touched_tile), then the tile's draws, a part of the stream per wave (tile_stream_wave)
// ---------------------------------------------------------------------------------------------
// which meshes' screen bounds touch the tile: lane m of the calling wave tests mesh m (pure: no side effects)
__device__ __forceinline__ unsigned long long tile_mesh_mask(const SceneView& sc, const RenderParams& p, const TileGeom& tg, int lane) {
    const mcrt_config& cfg = p.cfg;
    const bool dof = cfg.dof_enabled && cfg.aperture > 1e-6f;
    const bool cull = sc.hdr->cull_ok != 0 && sc.n_meshes < 64;
    const float aspect = static_cast<float>(cfg.width) / static_cast<float>(cfg.height);
    bool touch = lane < sc.n_meshes;
    if (touch && cull) {
        const FlatMesh& m = sc.meshes[lane];
        float lens_pad = 0.0f;
        if (dof) {
            // Thin lens (tile_renderer.cpp:42-69): the ray of screen sample s leaves the lens at offset
            // (lx, ly), |.| <= aperture, towards the focus point of the pinhole ray, so a point at
            // camera depth z on it is seen by the pinhole at s + (lx, ly) * (1/z - 1/zf), where
            // zf = focusDist / |(su, sv, 1)| is the depth of that focus point.  Over the mesh's depth
            // range and every zf of the frame this bounds the displacement; generous float slack.
            const float half_h = sc.hdr->cam_half_h, half_w = half_h * aspect;
            const float focus = cfg.focus_distance > 0.0f ? cfg.focus_distance : sc.hdr->cam_focus_auto;
            const float inv_f_lo = 1.0f / focus;
            const float inv_f_hi = __builtin_sqrtf(1.0f + half_w * half_w + half_h * half_h) / focus;
            const float inv_z_hi = 1.0f / m.depth[0], inv_z_lo = 1.0f / m.depth[1];
            const float d = fmaxf(fmaxf(fabsf(inv_z_hi - inv_f_lo), fabsf(inv_z_hi - inv_f_hi)),
                                  fmaxf(fabsf(inv_z_lo - inv_f_lo), fabsf(inv_z_lo - inv_f_hi)));
            lens_pad = cfg.aperture * d / half_h * 1.02f + 1e-3f;
            if (!(m.depth[0] > 0.0f) || !(focus > 0.0f) || !(lens_pad < 1e6f)) lens_pad = 1e30f;  // no bound
        }
        touch = mesh_touches_tile(m, tg, cfg, aspect, lens_pad);
    }
    unsigned long long mask = __ballot(touch);
    if (!cull && sc.n_meshes > 0) mask = ~0ull;
    return mask;
}
// lane 0 of the tile's wave: units and slot range of a tile that meshes can touch; returns its number
// among the batch's touched tiles
__device__ __forceinline__ uint32_t plan_touched_tile(const RenderParams& p, const TileGeom& tg, int tile, unsigned long long mask) {
    const mcrt_config& cfg = p.cfg;
    const WaveSpace& ws = p.ws;
    ws.tile_mask[tile] = mask;
    if (mask == 0ull) return ~0u;  // background tile: never queued
    const uint32_t spp = cfg.samples_per_pixel > 1 ? cfg.samples_per_pixel : 1;
    const uint32_t npix = static_cast<uint32_t>(tg.w) * static_cast<uint32_t>(tg.h);  // npix * spp <= ws.tile_slots
    const uint32_t parts = static_cast<uint32_t>(p.parts_per_tile);
    const uint32_t per = (npix + parts - 1u) / parts;
    const uint32_t used = (npix + per - 1u) / per;
    // the k-th touched tile of the batch owns slot range k: the workspace is sized for the tiles that can
    // be touched (host-side superset), not for every sample of the batch
    const uint32_t ord = count_add(ws, kCntTiles, 1u);
    if (ord >= ws.tile_cap) {  // cannot happen; keep memory safe and leave a mark if it ever does
        ws.counters[kCntOverflow] = 1u;
        ws.tile_mask[tile] = 0ull;
        return ~0u;
    }
    const uint32_t slot0 = count_add(ws, kCntUnits, used);
    const uint32_t base = ord * ws.tile_slots;
    for (uint32_t i = 0; i < used; ++i) {
        const uint32_t a = i * per, b = min(npix, a + per);
        ws.units[slot0 + i] = make_uint4(static_cast<uint32_t>(tile), a, b, base + a * spp);
    }
    return ord;
}

__global__ __launch_bounds__(64 * kStreamWaves) void plan_tiles_kernel(const uint8_t* __restrict__ scene_blob,
                                                                       const uint32_t* __restrict__ tile_rng,
                                                                       float* __restrict__ tile_draws, float4* __restrict__ out_frame,
                                                                       uchar4* __restrict__ out8, const RenderParams p,
                                                                       const int tile_base, const int n_tiles) {
    __shared__ __align__(16) uint32_t s_state[kStreamWaves][2 * 624];
    __shared__ uint32_t s_ord[kStreamWaves];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // `parts` neighbouring waves of the workgroup share a tile (1, 2 or 4: a power of two that divides kStreamWaves), each
    // taking `per_wave` consecutive parts of the tile's stream
    const int parts = p.stream_waves;
    const int per_wave = p.stream_parts / parts;
    const int slot = static_cast<int>(blockIdx.x) * kStreamWaves + wave;
    const int t = slot / parts, part = slot - t * parts;
    const bool valid = t < n_tiles;  // wave-uniform
    const int tile = tile_base + (valid ? t : 0);
    const TileGeom tg = tile_of(p, tile);
    const SceneView sc = view_of(scene_blob);
    // the tile's mesh mask: every wave of the tile forms it (the same ballot); its slot range and units: the first wave
    const unsigned long long mask = valid ? tile_mesh_mask(sc, p, tg, lane) : 0ull;
    if (valid && part == 0 && lane == 0) s_ord[wave] = plan_touched_tile(p, tg, tile, mask);
    if (parts > 1) {
        __syncthreads();  // the only workgroup barrier of the kernel: the tile's number among the touched tiles
    } else {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (!valid) return;
    const uint32_t ord = s_ord[wave - part];
    const size_t stride = p.ws.draws_stride;
    const unsigned npix = static_cast<unsigned>(tg.w) * static_cast<unsigned>(tg.h);
    if (!p.bg_in_plan) {  // every tile's draws to HBM; `primary` renders the background tiles (those of one colour need no draws)
        float4 unused;
        if (p.draws_per_sample > 0 && !(mask == 0ull && constant_background(sc, p, tg, unused)))
            tile_stream_wave(sc, tile_rng, tile_draws + static_cast<size_t>(t) * stride, nullptr, nullptr, p, tg, tile, part * per_wave, per_wave, s_state[wave], lane,
                             /*jitter_only=*/mask == 0ull && p.draws_per_sample == 4);
    } else if (mask != 0ull) {  // a tile meshes can touch: its draws, at its touched-tile number
        if (p.draws_per_sample > 0 && ord != ~0u)
            tile_stream_wave(sc, tile_rng, tile_draws + static_cast<size_t>(ord) * stride, nullptr, nullptr, p, tg, tile, part * per_wave, per_wave, s_state[wave], lane);
    } else if (float4 pixel; constant_background(sc, p, tg, pixel)) {  // background tile of one colour: no draws, no samples
        fill_tile(p, tg, out_frame, out8, pixel, static_cast<unsigned>(part * 64 + lane), static_cast<unsigned>(parts) * 64u);
    } else if (p.cfg.samples_per_pixel > 1) {  // background tile, jittered samples
        tile_stream_wave(sc, tile_rng, nullptr, out_frame, out8, p, tg, tile, part * per_wave, per_wave, s_state[wave], lane);
    } else {  // background tile, one centred sample per pixel: no draws at all
        background_pixels(sc, p, tg, out_frame, out8, npix * static_cast<unsigned>(part) / static_cast<unsigned>(parts),
                          npix * static_cast<unsigned>(part + 1) / static_cast<unsigned>(parts), lane,
                          [](unsigned, float&, float&) __attribute__((always_inline)) {});
    }
}

// The engine states at the starts of a tile's parts 1 .. parts-1 (tile_stream_wave): one wave per tile twists the seeded
// state part_twists times per part, in LDS, and stores where it stands.  A function of the seeds and of part_twists only:
// run with the tile seeds, kept across renders with them.
__global__ __launch_bounds__(64 * kStreamWaves) void advance_tiles_kernel(RenderParams p, int n_tiles) {
    __shared__ __align__(16) uint32_t s_state[kStreamWaves][2 * 624];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tile = static_cast<int>(blockIdx.x) * kStreamWaves + wave;
    if (tile >= n_tiles) return;  // wave-uniform; no workgroup barrier in this kernel
    uint32_t* st = s_state[wave];
    uint32_t* states = p.tile_rng + static_cast<size_t>(tile) * p.stream_parts * 624;
    for (int e = lane; e < 624; e += 64) st[e] = states[e];
    auto wave_sync = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    wave_sync();
    int cur = 0;
    for (int part = 1; part < p.stream_parts; ++part) {
        for (int k = 0; k < p.stream_part_twists; ++k) {
            const uint32_t* o = st + cur * 624;
            uint32_t* n = st + (cur ^ 1) * 624;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = lane + 64 * i;
                if (e < 227) n[e] = mt_twist(o[e], o[e + 1], o[e + 397]);
            }
            wave_sync();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = 227 + lane + 64 * i;
                if (e < 454) n[e] = mt_twist(o[e], o[e + 1], n[e - 227]);
            }
            wave_sync();
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int e = 454 + lane + 64 * i;
                if (e < 624) n[e] = mt_twist(o[e], (e == 623) ? n[0] : o[e + 1], n[e - 227]);
            }
            wave_sync();
            cur ^= 1;
        }
        const uint32_t* now = st + cur * 624;
        for (int e = lane; e < 624; e += 64) states[static_cast<size_t>(part) * 624 + e] = now[e];
    }
}

// ---------------------------------------------------------------------------------------------
// primary: persistent workgroups; touched units first, then the background tiles.  A sample's draws
// sit in tile_draws at ((pixel in tile) * spp + sample) * draws_per_sample.
// ---------------------------------------------------------------------------------------------
// 5 waves per SIMD (96 VGPRs, four of them spilled) instead of the 4 the unconstrained build settles at (105 VGPRs): same-box
// A/B +1.5 % at four frames in flight and, with 1280 workgroups, -8 us for one frame alone (profiles/r03_experiments)
#ifndef MCRT_PRIMARY_WAVES
#define MCRT_PRIMARY_WAVES 5
#endif
template <int kView>
__global__ __launch_bounds__(kBlock, MCRT_PRIMARY_WAVES) void primary_kernel(const uint8_t* __restrict__ scene_blob,
                                                         const float* __restrict__ tile_draws,
                                                         float4* __restrict__ out_frame, uchar4* __restrict__ out8, const RenderParams p,
                                                         const int tile_base, const int n_tiles) {
    __shared__ int s_wcnt[kBlock / 64];
    __shared__ float4 s_bd[kBlock], s_bp[kBlock], s_bn[kBlock];  // the chunk's primary hits, packed, for their reflection rays
    __shared__ uint32_t s_out_base;
    extern __shared__ __align__(16) unsigned char s_dyn[];

    const SceneView scg = view_of(scene_blob);
    const mcrt_config& cfg = p.cfg;
    const WaveSpace& ws = p.ws;
    const uint32_t n_units = counted(ws, kCntUnits);
    if (blockIdx.x == 0 && threadIdx.x == 0) ws.frame_info[0] = n_units;  // for `resolve`, which takes the next pass's counter base
    if ((p.bg_in_plan || p.bg_kernel) && blockIdx.x >= n_units) return;  // nothing for this workgroup: leave before the collective staging
    // flat pipeline: the reflection ray of every primary hit (raytracer.cpp:133-139) is traced right here, by the first
    // `total` threads of the block on the chunk's packed hits (as a launch of its own this stage re-read every record:
    // 39 + 28 us became 59 us)
    const bool bounce_here = p.flat && cfg.max_bounces >= 1;
    const bool posed = kView != kViewLdsUnposed && p.scene_posed != 0;
    const typename ViewSel<kView>::type sc = ViewSel<kView>::make(scg, p, s_dyn);
    const int tid = threadIdx.x;
    const int spp = cfg.samples_per_pixel > 1 ? cfg.samples_per_pixel : 1;
    const int dd = p.draws_per_sample;
    const bool dof = cfg.dof_enabled && cfg.aperture > 1e-6f;
    const float aspect = static_cast<float>(cfg.width) / static_cast<float>(cfg.height);
    const FrameDiv fd(p);
    float focusDist = cfg.focus_distance;
    if (focusDist <= 0.0f) focusDist = sc.hdr->cam_focus_auto;
    const size_t stride = ws.draws_stride;

    // ================= units of tiles that meshes can touch: thread per sample =================
    for (uint32_t u = blockIdx.x; u < n_units; u += gridDim.x) {
        const uint4 ud = ws.units[u];
        const int tile = static_cast<int>(ud.x);
        const unsigned pp0 = ud.y, pp1 = ud.z;
        const uint32_t slot_base = ud.w;
        const TileGeom tg = tile_of(p, tile);
        const unsigned long long mesh_mask = ws.tile_mask[tile];
        // the tile's draws: at its touched-tile number, or (all tiles' streams in HBM) at its index in the batch
        const float* draws = tile_draws + static_cast<size_t>(p.bg_in_plan ? slot_base / ws.tile_slots : static_cast<uint32_t>(tile - tile_base)) * stride;
        const unsigned n_samples = (pp1 - pp0) * static_cast<unsigned>(spp);

        uint32_t unit_hits = 0;  // hits so far: they occupy slots slot_base .. slot_base + unit_hits
        for (unsigned s0 = 0; s0 < n_samples; s0 += kChunk) {  // uniform trip count
            bool is_hit = false;
            Ray ray{mk(0, 0, 0), mk(0, 0, 0)};
            Hit hit;
            hit.hit = false;
            uint32_t sample_slot = 0;
            const unsigned sidx = s0 + tid;  // sample of the unit, in stream order
            if (sidx < n_samples) {
                const unsigned pix = pp0 + UDiv(static_cast<unsigned>(spp)).div(sidx);
                const unsigned uly = UDiv(static_cast<unsigned>(tg.w)).div(pix);
                const int ly = static_cast<int>(uly);
                const int lx = static_cast<int>(pix - uly * static_cast<unsigned>(tg.w));
                const int px = tg.x + lx, py = tg.y + ly;
                const float* jd = draws + (static_cast<size_t>(pp0) * spp + sidx) * dd;
                float jx = 0.5f, jy = 0.5f;
                int dpos = 0;
                if (spp > 1) {
                    jx = jd[0];
                    jy = jd[1];
                    dpos = 2;
                }
                const float su = fd.u(static_cast<float>(px) + jx);
                const float sv = fd.v(static_cast<float>(py) + jy);
                sample_slot = slot_base + sidx;
                ray = dof ? lens_ray(sc, su, sv, aspect, cfg.aperture, focusDist, jd[dpos], jd[dpos + 1])
                          : camera_ray(sc, su, sv, aspect);
                hit = hit_scene(sc, ray, mesh_mask);
                // A miss: its colour is a function of the sample's jitter pair alone (tile_renderer.cpp:111-114) — `resolve`
                // forms it from the pair in the tile's stream (8 B) instead of a colour stored here and read back there
                // (2 x 16 B per miss of a touched tile: 13 MB of the metric frame's counted HBM traffic).
                uint32_t code = kEndMiss;
                if (hit.hit && cfg.max_bounces < 0) {
                    const C4 col = background(sc, cfg, 0.5f, 0.5f);  // raytracer.cpp:86-90 (depth 0 > maxBounces)
                    ws.scol[sample_slot] = make_float4(col.r, col.g, col.b, col.a);
                    code = 0u;
                } else if (hit.hit) {
                    is_hit = true;
                    // 0: the colour is in scol (general variants: `level_shade` puts the hits' colours there).  Flat pipeline:
                    // the second phase below / `lit` overwrite a hit's word with its chain's end code — unless maxBounces is 0
                    // and the chain is its primary hit alone.
                    code = (p.flat && cfg.max_bounces == 0) ? 3u : 0u;
                }
                ws.end[sample_slot] = code;
            }
            // hits → dense entries at the front of the unit's slot range (no global atomics)
            int total = 0;
            const int rank = block_rank(is_hit, s_wcnt, total);
            if (is_hit) {
                const uint32_t e = slot_base + unit_hits + static_cast<uint32_t>(rank);
                // root of the chain = its sample's colour slot
                if (p.flat)
                    push_record(ws, posed, e, ray, hit, sample_slot, 0, dof);
                else
                    push_entry(ws, 0, e, ray, hit, sample_slot, 0);
                if (bounce_here) {
                    s_bd[rank] = make_float4(ray.d.x, ray.d.y, ray.d.z, __uint_as_float(sample_slot));
                    s_bp[rank] = make_float4(hit.p.x, hit.p.y, hit.p.z, 0.0f);
                    s_bn[rank] = make_float4(hit.n.x, hit.n.y, hit.n.z, 0.0f);
                }
            }
            unit_hits += static_cast<uint32_t>(total);
            if (bounce_here && total > 0) {  // uniform
                __syncthreads();
                bool next_hit = false;
                Ray nray{mk(0, 0, 0), mk(0, 0, 0)};
                Hit nhit;
                nhit.hit = false;
                uint32_t root = 0;
                if (tid < total) {
                    const float4 bd = s_bd[tid], bp = s_bp[tid], bn = s_bn[tid];
                    root = __float_as_uint(bd.w);
                    nray = reflect_ray(mk(bd.x, bd.y, bd.z), mk(bp.x, bp.y, bp.z), mk(bn.x, bn.y, bn.z));
                    nhit = hit_scene<true>(sc, nray, ~0ull);
                    if (nhit.hit)
                        next_hit = true;
                    else
                        ws.end[root] = chain_code(1, false);  // bounced ray missed → flat background (raytracer.cpp:94-102)
                }
                int survivors = 0;
                const int srank = block_rank(next_hit, s_wcnt, survivors);
                if (survivors > 0) {  // uniform
                    if (tid == 0) s_out_base = count_add(ws, kCntDeep1, static_cast<uint32_t>(survivors));
                    __syncthreads();
                    if (next_hit) push_record(ws, posed, ws.cap + s_out_base + static_cast<uint32_t>(srank), nray, nhit, root, 1, true);
                }
                __syncthreads();  // s_bd .. s_out_base are reused by the next chunk
            }
        }
        if (tid == 0) ws.unit_hits[u] = unit_hits;
    }


    // ================= background tiles: nothing can be hit, no ray is needed =================
    // (only when a pixel takes more than 24 draws — otherwise `plan_tiles` has rendered them from LDS — and at most
    // kSlabMinSpp - 1 samples: above that `background_kernel` does it)
    // thread per pixel, its samples in order as renderTile sums them (tile_renderer.cpp:116-124); no barriers
    if (p.bg_in_plan || p.bg_kernel) return;
    const float inv_spp = 1.0f / static_cast<float>(spp);
    for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const int tile = tile_base + t;
        if (ws.tile_mask[tile] != 0ull) continue;  // uniform
        const TileGeom tg = tile_of(p, tile);
        if (float4 pixel; constant_background(scg, p, tg, pixel)) {  // uniform
            fill_tile(p, tg, out_frame, out8, pixel, static_cast<unsigned>(tid), static_cast<unsigned>(kBlock));
            continue;
        }
        const unsigned npix = static_cast<unsigned>(tg.w) * static_cast<unsigned>(tg.h);
        // the tile's jitter pairs, pixel-major (under depth of field a background tile's stream holds them only, packed: tile_stream_wave)
        const float2* pairs = reinterpret_cast<const float2*>(tile_draws + static_cast<size_t>(t) * stride);
        for (unsigned pix = tid; pix < npix; pix += kBlock) {
            const unsigned uly = pix / static_cast<unsigned>(tg.w);
            const int ly = static_cast<int>(uly);
            const int lx = static_cast<int>(pix - uly * static_cast<unsigned>(tg.w));
            const float fx = static_cast<float>(tg.x + lx), fy = static_cast<float>(tg.y + ly);
            const float2* jd = pairs + static_cast<size_t>(pix) * spp;
            float ar = 0.0f, ag = 0.0f, ab = 0.0f, aa = 0.0f;
            for (int sidx = 0; sidx < spp; ++sidx) {
                const float2 j = jd[sidx];
                const C4 c = background(sc, cfg, fd.u(fx + j.x), fd.v(fy + j.y));  // tile_renderer.cpp:111-114
                ar += c.r;
                ag += c.g;
                ab += c.b;
                aa += c.a;
            }
            const int row = (p.layout == MCRT_LAYOUT_PACKED) ? ((p.shard.pack_first + tg.owned_row * p.shard.pack_step) * cfg.tile_size + ly) : (tg.y + ly);
            store_pixel(out_frame, out8, static_cast<size_t>(row) * cfg.width + (tg.x + lx),
                        make_float4(ar * inv_spp, ag * inv_spp, ab * inv_spp, aa * inv_spp));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// background: the tiles no mesh can touch at HIGH sample counts (kSlabMinSpp samples per pixel and more; below that the
// loop above, in `primary`'s tail, does as well).  The tile's stream in HBM is pixel-major — a pixel's spp jitter pairs
// side by side, 8·spp bytes per pixel — so a lane per pixel reads it at a stride of 8·spp bytes: at 64 spp that is 64 cache
// lines per load instruction and 32 KB of lines per wave in flight, more than the L1 holds: the loop was bound by those
// transactions (8.4 of the 8K / 64 spp frame's 14.3 ms per pass).  Here a wave fetches a SLAB — 64 pixels x 16 samples — with
// neighbouring lanes on neighbouring pairs (every line touched once and used whole), parks it in LDS with the pixels' rows
// padded by one pair (a lane per pixel then reads at a stride of 34 words: two lanes per bank) and consumes it, a lane per
// pixel, samples in order (tile_renderer.cpp:116-124); waves work on their own: no workgroup barrier.  8K / 64 spp: 20.4 ->
// 17.1 ms alone; at 16 spp the staging costs 3-5 % more than it saves (same-box A/B, profiles/r03_experiments).
// ---------------------------------------------------------------------------------------------
constexpr int kSlabSamples = 16;
constexpr int kSlabMinSpp = 33;  // `background_kernel` from this many samples per pixel on
#ifndef MCRT_BG_WAVES
#define MCRT_BG_WAVES 3
#endif
__global__ __launch_bounds__(kBlock, MCRT_BG_WAVES) void background_kernel(const uint8_t* __restrict__ scene_blob, const float* __restrict__ tile_draws,
                                                            float4* __restrict__ out_frame, uchar4* __restrict__ out8, const RenderParams p,
                                                            const int tile_base, const int n_tiles) {
    __shared__ __align__(16) float2 s_slab[kBlock / 64][64][kSlabSamples + 1];
    const SceneView sc = view_of(scene_blob);
    const mcrt_config& cfg = p.cfg;
    const WaveSpace& ws = p.ws;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned spp = static_cast<unsigned>(cfg.samples_per_pixel > 1 ? cfg.samples_per_pixel : 1);
    const FrameDiv fd(p);
    const float inv_spp = 1.0f / static_cast<float>(spp);
    const size_t stride = ws.draws_stride;
    float2(*slab)[kSlabSamples + 1] = s_slab[wave];
    auto wave_sync = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const int tile = tile_base + t;
        if (ws.tile_mask[tile] != 0ull) continue;  // uniform
        const TileGeom tg = tile_of(p, tile);
        if (float4 pixel; constant_background(sc, p, tg, pixel)) {  // uniform
            fill_tile(p, tg, out_frame, out8, pixel, threadIdx.x, static_cast<unsigned>(kBlock));
            continue;
        }
        const unsigned npix = static_cast<unsigned>(tg.w) * static_cast<unsigned>(tg.h);
        // the tile's jitter pairs, pixel-major: pair s of pixel q at pairs[q * spp + s] (under depth of field the stream of a
        // background tile holds the jitter pairs only, packed: tile_stream_wave)
        const float2* pairs = reinterpret_cast<const float2*>(tile_draws + static_cast<size_t>(t) * stride);
        const UDiv by_w(static_cast<unsigned>(tg.w));
        for (unsigned g0 = static_cast<unsigned>(wave) * 64u; g0 < npix; g0 += kBlock) {  // this wave's groups of 64 pixels
            const unsigned group = min(64u, npix - g0);
            const unsigned pix = g0 + static_cast<unsigned>(lane);
            const unsigned uly = by_w.div(pix);
            const int ly = static_cast<int>(uly);
            const int lx = static_cast<int>(pix - uly * static_cast<unsigned>(tg.w));
            const float fx = static_cast<float>(tg.x + lx), fy = static_cast<float>(tg.y + ly);
            float ar = 0.0f, ag = 0.0f, ab = 0.0f, aa = 0.0f;
            // A full slab (64 pixels x 16 samples) is fetched into registers one slab AHEAD: lane l takes pair l % 16 of
            // pixels l / 16, l / 16 + 4, ... — 16 loads, a pixel's 128 bytes across 16 lanes — issued before the previous
            // slab is consumed, parked in LDS after it: the loads' latency runs under 1 024 background samples.
            float2 ahead[kSlabSamples];
            const float2* mine = pairs + static_cast<size_t>(g0 + (static_cast<unsigned>(lane) >> 4)) * spp + (static_cast<unsigned>(lane) & 15u);
            auto fetch_ahead = [&](unsigned s0) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < kSlabSamples; ++i) ahead[i] = mine[static_cast<size_t>(4 * i) * spp + s0];
            };
            const bool full_group = group == 64u;
            if (full_group && spp >= static_cast<unsigned>(kSlabSamples)) fetch_ahead(0u);
            for (unsigned s0 = 0; s0 < spp; s0 += kSlabSamples) {  // uniform
                const unsigned ns = min(static_cast<unsigned>(kSlabSamples), spp - s0);
                if (full_group && ns == static_cast<unsigned>(kSlabSamples)) {  // uniform
#pragma unroll
                    for (int i = 0; i < kSlabSamples; ++i) slab[4 * i + (lane >> 4)][lane & 15] = ahead[i];
                    if (s0 + 2u * kSlabSamples <= spp) fetch_ahead(s0 + kSlabSamples);  // the next slab is a full one too
                } else {
                    const UDiv by_ns(ns);
                    const unsigned items = group * ns;
                    for (unsigned c = static_cast<unsigned>(lane); c < items; c += 64u) {  // neighbouring lanes, neighbouring pairs
                        const unsigned q = by_ns.div(c), k = c - q * ns;
                        slab[q][k] = pairs[static_cast<size_t>(g0 + q) * spp + s0 + k];
                    }
                }
                wave_sync();
                if (static_cast<unsigned>(lane) < group) {
                    for (unsigned k = 0; k < ns; ++k) {
                        const float2 j = slab[lane][k];
                        const C4 col = background(sc, cfg, fd.u(fx + j.x), fd.v(fy + j.y));  // tile_renderer.cpp:111-114
                        ar += col.r;
                        ag += col.g;
                        ab += col.b;
                        aa += col.a;
                    }
                }
                wave_sync();  // the next slab overwrites this one
            }
            if (static_cast<unsigned>(lane) < group) {
                const int row = (p.layout == MCRT_LAYOUT_PACKED) ? ((p.shard.pack_first + tg.owned_row * p.shard.pack_step) * cfg.tile_size + ly) : (tg.y + ly);
                store_pixel(out_frame, out8, static_cast<size_t>(row) * cfg.width + (tg.x + lx), make_float4(ar * inv_spp, ag * inv_spp, ab * inv_spp, aa * inv_spp));
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The recursion of RayTracer::traceRay (raytracer.cpp:82-148) as a FLAT pipeline.  The reflection ray
// of a hit depends on geometry only — direction, hit point, normal (:133-139) — not on the colour
// of the hit.  So the chain of hits below a primary hit is chased first (`primary`'s second phase: every
// primary hit's reflection ray; the head of `lit`: the ~10 % that hit again, followed to their end), every hit of every
// level becoming one *record*; then the expensive stages — light samples, shadow rays, shading — run ONCE over
// all records of all levels, and `resolve` folds each chain's level colours back to front (:143-147)
// while it sums the pixel's samples.  One launch set per level cost ~40 us of dependent latency per
// level whatever it held (320 k, 31 k, 3 k, 300, 30 records at 1080p / 4 spp).
//
// Primary hits sit at the front of each unit's slot range (count per unit, no atomics).  Level-1 records are
// appended densely behind index `cap` by `primary` with ONE workgroup-aggregated atomic per 256-entry block (counter
// kCntDeep1); the records of levels >= 2 go to per-block regions behind them and never meet a global counter (`lit`).
// The general variants (per-hit RNG streams longer than the register engine, or more bounces than
// the flat record arrays are laid out for) keep one launch set per level with ping-pong queues.
// ---------------------------------------------------------------------------------------------

// which records a queue kernel walks: flat — every record of the batch; else — the entries of `level`
struct Scope {
    int level;
    int flat;
    __device__ __forceinline__ bool units() const { return flat || level == 0; }  // records at the front of the units' slot ranges
    __device__ __forceinline__ bool dense() const { return flat || level > 0; }   // records in the dense queue
    __device__ __forceinline__ int par() const { return flat ? 0 : (level & 1); }
};
__device__ __forceinline__ uint32_t dense_count(const WaveSpace& ws, Scope s) {
    return s.flat ? counted(ws, kCntDeep1) : counted(ws, kCntDense + s.level);  // (flat: the level-1 records; deeper ones are private to `lit`'s chasing blocks)
}
__device__ __forceinline__ uint32_t dense_base(const WaveSpace& ws, Scope s) { return s.flat ? ws.cap : 0u; }

// Calls body(first_entry, n_valid) for consecutive blocks of up to kBlock entries of the scope;
// every thread of the workgroup makes the same calls (collectives inside the body are allowed).
template <class F>
__device__ __forceinline__ void for_each_unit_block(const WaveSpace& ws, F&& body) {
    const uint32_t n_units = counted(ws, kCntUnits);
    for (uint32_t u = blockIdx.x; u < n_units; u += gridDim.x) {
        const uint32_t base = ws.units[u].w;
        const uint32_t count = ws.unit_hits[u];
        for (uint32_t k0 = 0; k0 < count; k0 += kBlock) body(base + k0, min(static_cast<uint32_t>(kBlock), count - k0));
    }
}
template <class F>
__device__ __forceinline__ void for_each_dense_block(uint32_t base, uint32_t count, F&& body) {
    for (uint32_t k0 = blockIdx.x * kBlock; k0 < count; k0 += gridDim.x * kBlock) body(base + k0, min(static_cast<uint32_t>(kBlock), count - k0));
}
template <class F>
__device__ __forceinline__ void for_each_entry_block(const WaveSpace& ws, Scope s, F&& body) {
    if (s.units()) for_each_unit_block(ws, body);
    if (s.dense()) for_each_dense_block(dense_base(ws, s), dense_count(ws, s), body);
}
// True when this workgroup will get no entry block of the scope: it can leave before it stages the scene
// tables (a deep level holds a few thousand entries, yet every workgroup of a launch would stage 8.7 KB).
__device__ __forceinline__ bool no_entry_blocks(const WaveSpace& ws, Scope s) {
    if (s.units() && blockIdx.x < counted(ws, kCntUnits)) return false;
    if (s.dense() && static_cast<unsigned long long>(blockIdx.x) * kBlock < dense_count(ws, s)) return false;
    return true;
}

struct Record {
    Ray ray;
    Hit hit;
    uint32_t root;
    int depth;
};
__device__ __forceinline__ Record load_record(const WaveSpace& ws, int par, uint32_t e, bool with_texel) {
    Record r;
    const float4 qo = ws.q_o[par][e], qd = ws.q_d[par][e], qp = ws.q_p[par][e], qn = ws.q_n[par][e];
    r.root = __float_as_uint(qo.w);
    r.depth = __float_as_int(qd.w);
    r.ray = Ray{mk(qo.x, qo.y, qo.z), mk(qd.x, qd.y, qd.z)};
    r.hit.hit = true;
    r.hit.outer = false;
    r.hit.t = 0.0f;
    r.hit.p = mk(qp.x, qp.y, qp.z);
    r.hit.n = mk(qn.x, qn.y, qn.z);
    r.hit.tex = C4{0.0f, 0.0f, 0.0f, 1.0f};
    if (with_texel) {
        const float4 qt = ws.q_t[par][e];
        r.hit.tex = C4{qt.x, qt.y, qt.z, qt.w};
    }
    return r;
}

// The S light sample positions of one hit: mt19937(shadow seed) → 2·S draws → disk samples
// (shading.cpp:28-53).  kGeneral: streams longer than 227 draws use the full engine in HBM.
template <bool kGeneral>
struct SampleRng {
    using type = MtShort;
};
template <>
struct SampleRng<true> {
    using type = HitRng;
};

// seeds the engine of one hit (the sequential part: the 397-step mt19937 recurrence)
template <bool kGeneral>
__device__ __forceinline__ void seed_light_rng(typename SampleRng<kGeneral>::type& rng, V3 P, int depth, int S, uint32_t* my_rng) {
    if constexpr (kGeneral)
        rng.seed(shadow_seed(P, depth), 2 * S, my_rng);
    else
        rng.seed(shadow_seed(P, depth));
}

// 2·S draws of a seeded engine → the S light sample positions of entry e
// ... and the hit's bundle mask: which meshes any of its S shadow rays can meet (rt::bundle_candidates)
template <bool kPosed, class Rng>
__device__ __forceinline__ void write_light_samples(const SceneView& sc, const WaveSpace& ws, uint32_t e, V3 P, V3 N, int S, Rng& rng) {
    ws.cand[e] = bundle_candidates<kPosed>(sc, P + N * 1e-3f, ld3(sc.hdr->light_pos), sc.hdr->light_radius);
    const LightFrame frame = light_frame(sc, P);
    float* dst = ws.targets + static_cast<size_t>(e) * 3 * S;
    for (int i = 0; i < S; ++i) {
        const float d0 = rng.uniform();
        const float d1 = rng.uniform();
        const V3 t = light_sample_on_frame(sc, frame, d0, d1);
        dst[3 * i + 0] = t.x;
        dst[3 * i + 1] = t.y;
        dst[3 * i + 2] = t.z;
    }
}

template <bool kGeneral, bool kPosed>
__device__ __forceinline__ void emit_light_samples(const SceneView& sc, const WaveSpace& ws, uint32_t e, V3 P, V3 N, int depth, int S,
                                                   uint32_t* my_rng) {
    typename SampleRng<kGeneral>::type rng;
    seed_light_rng<kGeneral>(rng, P, depth, S, my_rng);
    write_light_samples<kPosed>(sc, ws, e, P, N, S, rng);
}

// light_samples: per record, the 397-step mt19937 seeding recurrence (sequential), then its 2·S
// draws turned into the S light sample positions.  One record per lane: thousands of resident waves
// hide the integer chain, and the S independent cos/sin/sqrt evaluations of a hit interleave.
template <bool kGeneral, bool kPosed>
__global__ __launch_bounds__(kBlock) void light_samples_kernel(const uint8_t* __restrict__ scene_blob, const RenderParams p,
                                                               const int level) {
    const SceneView sc = view_of(scene_blob);
    const WaveSpace& ws = p.ws;
    const Scope scope{level, p.flat};
    const int par = scope.par();
    const int S = p.cfg.shadow_samples;
    uint32_t* my_rng = nullptr;
    if constexpr (kGeneral)
        my_rng = ws.hit_rng + (static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x) * 624;
    for_each_entry_block(ws, scope, [&](uint32_t first, uint32_t n) __attribute__((always_inline)) {
        if (threadIdx.x >= n) return;
        const uint32_t e = first + threadIdx.x;
        const float4 hp = ws.q_p[par][e], hn = ws.q_n[par][e];
        emit_light_samples<kGeneral, kPosed>(sc, ws, e, mk(hp.x, hp.y, hp.z), mk(hn.x, hn.y, hn.z), __float_as_int(ws.q_d[par][e].w), S,
                                             my_rng);
    });
}

// shadow: one (record, light sample) pair per lane
// occupancy targets of the queue kernels (measured: 5 waves/SIMD best for shadow; 6+ spills)
#ifndef MCRT_SHADOW_WAVES
#define MCRT_SHADOW_WAVES 5
#endif
template <int kView>
__global__ __launch_bounds__(kBlock, MCRT_SHADOW_WAVES) void shadow_kernel(const uint8_t* __restrict__ scene_blob, const RenderParams p,
                                                        const int level) {
    extern __shared__ __align__(16) unsigned char s_dyn[];
    const SceneView scg = view_of(scene_blob);
    const WaveSpace& ws = p.ws;
    const Scope scope{level, p.flat};
    const int par = scope.par();
    const int mode = shadow_mode(scg, p.cfg);
    const int S = p.cfg.shadow_samples;
    const uint32_t pairs_per_hit = (mode == SHADOW_SOFT) ? static_cast<uint32_t>(S) : 1u;
    // groups of pairs_per_hit consecutive lanes belong to one hit; when that is a power of two
    // <= 64 the lit count is a ballot + popcount, otherwise atomics on a zeroed counter
    const bool pow2 = (pairs_per_hit & (pairs_per_hit - 1u)) == 0u && pairs_per_hit <= 64u;
    const uint32_t n_dense = scope.dense() ? dense_count(ws, scope) : 0u;
    {  // leave before the collective staging when this workgroup gets nothing
        const bool some_units = scope.units() && blockIdx.x < counted(ws, kCntUnits);
        const unsigned long long dense_items = pow2 ? static_cast<unsigned long long>(n_dense) * pairs_per_hit : n_dense;  // pow2: strided over pairs
        if (!some_units && static_cast<unsigned long long>(blockIdx.x) * kBlock >= dense_items) return;
    }
    const typename ViewSel<kView>::type sc = ViewSel<kView>::make(scg, p, s_dyn);
    uint32_t* lit = ws.lit[par];
    const V3 lpos = ld3(sc.hdr->light_pos);
    const uint32_t lane = threadIdx.x & 63u;
    // one wave-aligned group of up to 64 consecutive (entry, sample) pairs of the range `first` .. +total
    auto trace_pairs = [&](uint32_t first, uint32_t q0, uint32_t total) __attribute__((always_inline)) {
        const uint32_t q = q0 + lane;
        bool visible = false;
        uint32_t e = 0;
        if (q < total) {
            const uint32_t k = q / pairs_per_hit;
            const uint32_t j = q - k * pairs_per_hit;
            e = first + k;
            const float4 hp = ws.q_p[par][e], hn = ws.q_n[par][e];
            const V3 P = mk(hp.x, hp.y, hp.z);
            V3 N = mk(hn.x, hn.y, hn.z);
            if (mode == SHADOW_HARD) N = normalize(N);
            V3 target = lpos;
            if (mode == SHADOW_SOFT) {  // the first pass was done once for the hit's whole bundle of rays
                target = ld3(ws.targets + (static_cast<size_t>(e) * S + j) * 3);
                visible = !in_shadow_masked(sc, P, N, target, ws.cand[e]);
            } else {
                visible = !in_shadow_inline(sc, P, N, target);
            }
        }
        if (pow2) {
            const unsigned long long m = __ballot(visible);
            if (q < total && (lane & (pairs_per_hit - 1u)) == 0u) {
                const unsigned long long grp =
                    (pairs_per_hit == 64u) ? m : ((m >> lane) & ((1ull << pairs_per_hit) - 1ull));
                lit[e] = static_cast<uint32_t>(__popcll(grp));
            }
        } else if (visible) {
            atomicAdd(&lit[e], 1u);
        }
    };
    auto entry_block = [&](uint32_t first, uint32_t n) __attribute__((always_inline)) {
        if (!pow2) {
            if (threadIdx.x < n) lit[first + threadIdx.x] = 0u;
            __syncthreads();
        }
        const uint32_t total = n * pairs_per_hit;
        // every lane of a wave runs the same number of iterations (ballot inside)
        for (uint32_t q0 = threadIdx.x & ~63u; q0 < total; q0 += kBlock) trace_pairs(first, q0, total);
    };
    if (scope.units()) for_each_unit_block(ws, entry_block);
    if (!scope.dense()) return;
    const uint32_t base = dense_base(ws, scope);
    if (pow2) {
        // dense queue: stride over PAIRS so that even a sparse queue spreads over all workgroups
        const unsigned long long total = static_cast<unsigned long long>(n_dense) * pairs_per_hit;
        const unsigned long long step = static_cast<unsigned long long>(gridDim.x) * kBlock;
        for (unsigned long long q0 = static_cast<unsigned long long>(blockIdx.x) * kBlock + (threadIdx.x & ~63u); q0 < total; q0 += step) {
            // 64 consecutive pairs start at entry q0 / pairs_per_hit exactly (pairs_per_hit divides 64)
            const uint32_t first = base + static_cast<uint32_t>(q0 / pairs_per_hit);
            const uint32_t left = static_cast<uint32_t>(min(total - q0, 64ull));
            trace_pairs(first, 0u, left);
        }
    } else {
        for_each_dense_block(base, n_dense, entry_block);
    }
}

// ---------------------------------------------------------------------------------------------
// lit (flat pipeline): light samples, shadow rays AND shading of every record in one launch, handed over
// through LDS.  Per block of up to `lit_round` records:
//   phase A  a lane per record — the 397-step mt19937 seeding recurrence, the 2·S draws, the S disk
//            sample positions and the record's bundle mask (shading.cpp:28-53) — into LDS;
//   phase B  a lane per (record, light sample) — the exact any-hit test on the bundle mask — lit
//            count by ballot (S a power of two) or LDS atomics, into LDS;
//   phase C  a lane per record — Blinn-Phong with that visibility term (shading.cpp:62-96), times the AO
//            factor for primary hits (raytracer.cpp:121-130; the AO stage runs before this kernel) — onto
//            the chain's stack at the record's depth.
// As separate kernels the positions and masks went through HBM (96 + 8 B written and read back per
// record: ~180 MB of the metric frame's counted traffic), and the VALU-bound seeding chains could not
// overlap the latency-bound shadow rays.  Hard shadows / a point light: no phase A, one ray per record.
// ---------------------------------------------------------------------------------------------
#ifndef MCRT_LIT_WAVES
#define MCRT_LIT_WAVES 4
#endif
template <int kView>
__global__ __launch_bounds__(kBlock, MCRT_LIT_WAVES) void lit_kernel(const uint8_t* __restrict__ scene_blob, const RenderParams p) {
    extern __shared__ __align__(16) unsigned char s_dyn[];  // [scene tables][candidate masks: lit_round][inside masks: lit_round][positions: lit_pass x S x 3 floats][lit counts: lit_round][undecided list: lit_round]
    __shared__ int s_wcnt[kBlock / 64];
    MCRT_HOOK_LIT_SHARED
    const SceneView scg = view_of(scene_blob);
    const WaveSpace& ws = p.ws;
    const Scope scope{0, 1};
    // ONE work list — the units' record ranges, then the dense queue in blocks of 256 — so that the three phases
    // exist once in the code object (inlined per list they made 33 KB, half the instruction cache)
    const uint32_t n_units = counted(ws, kCntUnits);
    const uint32_t n_dense = dense_count(ws, scope);
    const uint32_t n_items = n_units + (n_dense + kBlock - 1u) / kBlock;
    // The chase: the chains below the level-1 records are followed HERE, ahead of the work list — a block of 256
    // level-1 records per workgroup turn, a lane per chain — and the records they append (levels >= 2) go to a region
    // of the dense queue that belongs to that block alone: no global counter, and no other workgroup ever waits for
    // them.  The workgroup that made them lights and shades them itself, as one more (sparse) entry of its work list.
    // (As a launch of its own between `primary` and `lit` the chase was 32 us of a lone frame's 225 for 2 us of work —
    // three dependent scene queries on 122 workgroups; in here it costs `lit` 5.)
    const int max_b = p.cfg.max_bounces;
    const uint32_t count1 = counted(ws, kCntDeep1);
    const uint32_t n_chase = max_b >= 1 ? (count1 + kBlock - 1u) / kBlock : 0u;
    const uint32_t n_work = n_chase + n_items;
    if (blockIdx.x >= n_work) return;  // before the collective staging
    const typename ViewSel<kView>::type sc = ViewSel<kView>::make(scg, p, s_dyn);
    constexpr bool kPosed = kView != kViewLdsUnposed;
    const int mode = shadow_mode(scg, p.cfg);
    const int S = p.cfg.shadow_samples;
    const uint32_t pairs_per_hit = (mode == SHADOW_SOFT) ? static_cast<uint32_t>(S) : 1u;
    const bool pow2 = (pairs_per_hit & (pairs_per_hit - 1u)) == 0u && pairs_per_hit <= 64u;
    const uint32_t round = static_cast<uint32_t>(p.lit_round);
    unsigned long long* s_cand = reinterpret_cast<unsigned long long*>(s_dyn + p.lit_lds_offset);  // 16-aligned
    unsigned long long* s_ins = s_cand + round;  // of those, the boxes that hold the record's ray origin strictly inside (rt::mesh_candidate_inside)
    float* s_pos = reinterpret_cast<float*>(s_ins + round);
    const uint32_t pass = static_cast<uint32_t>(p.lit_pass);  // records whose sample positions fit the LDS area at once
    uint32_t* s_lit = reinterpret_cast<uint32_t*>(s_pos + static_cast<size_t>(pass) * pairs_per_hit * 3);
    uint32_t* s_und = s_lit + round;  // the round's undecided records, packed
    const V3 lpos = ld3(scg.hdr->light_pos);
    const float lradius = scg.hdr->light_radius;
    const uint32_t lane = threadIdx.x & 63u;
    const mcrt_config& cfg = p.cfg;
    const bool posed = kView != kViewLdsUnposed && p.scene_posed != 0;  // the un-posed variants never read q_n
    const bool dof = cfg.dof_enabled && cfg.aperture > 1e-6f;
    const V3 cam_pos = ld3(scg.hdr->cam_pos);
    for (uint32_t work = blockIdx.x; work < n_work; work += gridDim.x) {
        uint32_t first, n;
        if (work < n_chase) {
            // the chains of level-1 records [k0, k0 + 256): a lane per chain — reflect, closest hit, append — until the ray
            // misses or the chain reaches maxBounces; ~10 % of the lanes go on per turn, the loop is workgroup-uniform
            const uint32_t k0 = work * kBlock;
            const uint32_t region = ws.cap + n_chase * kBlock + k0 * static_cast<uint32_t>(max_b - 1);  // 256 x (maxBounces - 1) slots per block
            uint32_t filled = 0;  // uniform
            bool active = k0 + threadIdx.x < count1;
            RecordGeom r;
            r.root = 0;
            r.depth = 1;
            r.d = mk(0, 0, 0);
            r.p = mk(0, 0, 0);
            r.n = mk(0, 0, 0);
            if (active) r = load_geom(ws, posed, ws.cap + k0 + threadIdx.x);
            for (;;) {
                bool next_hit = false;
                Ray nray{mk(0, 0, 0), mk(0, 0, 0)};
                Hit nhit;
                nhit.hit = false;
                if (active) {
                    if (r.depth >= max_b) {  // no reflection at the last level
                        ws.end[r.root] = chain_code(r.depth + 1, true);
                        active = false;
                    } else {
                        nray = reflect_ray(r.d, r.p, r.n);
                        nhit = hit_scene<true>(sc, nray, ~0ull);
                        if (nhit.hit) {
                            next_hit = true;
                        } else {
                            ws.end[r.root] = chain_code(r.depth + 1, false);
                            active = false;
                        }
                    }
                }
                int total = 0;
                const int rank = block_rank(next_hit, s_wcnt, total);
                if (total == 0) break;  // uniform: no chain of this block goes on
                if (next_hit) {
                    push_record(ws, posed, region + filled + static_cast<uint32_t>(rank), nray, nhit, r.root, r.depth + 1, true);
                    r.d = nray.d;
                    r.p = nhit.p;
                    r.n = nhit.n;
                    ++r.depth;
                }
                filled += static_cast<uint32_t>(total);
            }
            __syncthreads();  // the block's own records, written by other lanes than the ones that read them below
            first = region;
            n = filled;
        } else if (work - n_chase < n_units) {
            first = ws.units[work - n_chase].w;
            n = ws.unit_hits[work - n_chase];
        } else {
            const uint32_t k0 = (work - n_chase - n_units) * kBlock;
            first = ws.cap + k0;
            n = min(static_cast<uint32_t>(kBlock), n_dense - k0);
        }
        for (uint32_t r0 = 0; r0 < n; r0 += round) {  // uniform
            const uint32_t m = min(round, n - r0);
            const uint32_t base = first + r0;
            // ---- phase A1: a lane per record — the bundle mask, and whether the whole bundle is decided (rt::bundle_classify)
            bool undecided = false;
            if (threadIdx.x < m) {
                if (mode == SHADOW_SOFT) {
                    const RecordGeom g = load_geom(ws, posed, base + threadIdx.x);
                    const V3 O = g.p + g.n * 1e-3f;
                    unsigned long long cand;
                    const int known = bundle_classify<kPosed>(scg, sc, O, lpos, lradius, S, p.bundle_decisions != 0, cand);
                    undecided = known < 0;
                    MCRT_HOOK_LIT_CLASSIFIED(known, undecided, cand, O)
                    s_cand[threadIdx.x] = cand;
                    s_ins[threadIdx.x] = (undecided && p.inside_fast) ? origin_inside_boxes(sc, O, cand) : 0ull;
                    s_lit[threadIdx.x] = undecided ? 0u : static_cast<uint32_t>(known);
                } else {
                    s_lit[threadIdx.x] = 0u;
                }
            }
            uint32_t n_und = m;  // records whose rays are traced
            if (mode == SHADOW_SOFT) {
                int total = 0;
                const int rank = block_rank(undecided, s_wcnt, total);
                if (undecided) s_und[rank] = threadIdx.x;
                n_und = static_cast<uint32_t>(total);
                if (n_und == 0u) {  // uniform: no ray of this round needs tracing
                    __syncthreads();
                    goto shade_round;
                }
                __syncthreads();
            }
            // the traced records in passes of up to `pass` (their sample positions share the LDS area: with the
            // bundle decisions few records of a round are left, and the area is sized for those)
            for (uint32_t u0 = 0; u0 < n_und; u0 += pass) {  // uniform
                const uint32_t nu = min(pass, n_und - u0);
                if (u0) __syncthreads();  // the previous pass's rays have read the positions
                // ---- phase A2: a lane per undecided record — its mt19937 stream and the S disk sample positions
                if (mode == SHADOW_SOFT && threadIdx.x < nu) {
                    const RecordGeom g = load_geom(ws, posed, base + s_und[u0 + threadIdx.x]);
                    const V3 P = g.p;
                    MtShort rng;
                    const uint32_t seed = shadow_seed(P, g.depth);
                    const uint32_t slot = seed + kSeedWindowHalf;  // wraps: the window is centred on seed 0
                    if (p.seed_table && slot < kSeedWindow)
                        rng.seed_known(seed, p.seed_table[slot]);  // mt[397] of this seed, from the device's table
                    else if (p.seed_table_full)
                        rng.seed_known(seed, p.seed_table_full[seed]);  // (a scene at another scale: its seeds leave the window)
                    else
                        rng.seed(seed);  // the 397-step recurrence
                    const LightFrame frame = light_frame(scg, P);
                    float* dst = s_pos + static_cast<size_t>(threadIdx.x) * 3 * S;
                    for (int i = 0; i < S; ++i) {
                        const float d0 = rng.uniform();
                        const float d1 = rng.uniform();
                        const V3 t = light_sample_on_frame(scg, frame, d0, d1);
                        dst[3 * i + 0] = t.x;
                        dst[3 * i + 1] = t.y;
                        dst[3 * i + 2] = t.z;
                    }
                }
                __syncthreads();
                // ---- phase B: a lane per (undecided record, light sample); every lane of a wave runs the same number of turns (ballot inside)
                const uint32_t total = nu * pairs_per_hit;
                for (uint32_t q0 = threadIdx.x & ~63u; q0 < total; q0 += kBlock) {
                    const uint32_t q = q0 + lane;
                    bool visible = false;
                    uint32_t k = 0;
                    if (q < total) {
                        const uint32_t j = u0 + q / pairs_per_hit;
                        k = (mode == SHADOW_SOFT) ? s_und[j] : j;
                        V3 P, N;
                        load_point_normal(ws, posed, base + k, P, N);
                        if (mode == SHADOW_HARD) N = normalize(N);
                        if (mode == SHADOW_SOFT)
                            visible = !in_shadow_masked(sc, P, N, ld3(s_pos + static_cast<size_t>(q) * 3), s_cand[k], s_ins[k]);
                        else
                            visible = !in_shadow_inline(sc, P, N, lpos);
                    }
                    if (pow2) {
                        const unsigned long long bal = __ballot(visible);
                        if (q < total && (lane & (pairs_per_hit - 1u)) == 0u) {
                            const unsigned long long grp = (pairs_per_hit == 64u) ? bal : ((bal >> lane) & ((1ull << pairs_per_hit) - 1ull));
                            s_lit[k] = static_cast<uint32_t>(__popcll(grp));
                        }
                    } else if (visible) {
                        atomicAdd(&s_lit[k], 1u);
                    }
                }
            }
            __syncthreads();  // the counts are complete; the next round may overwrite the positions
        shade_round:
            // ---- phase C: a lane per record
            if (threadIdx.x < m) {
                const uint32_t e = base + threadIdx.x;
                const RecordGeom r = load_geom(ws, posed, e);
                Hit hit;
                hit.hit = true;
                hit.p = r.p;
                hit.n = r.n;
                hit.tex = texel_color(scg, ws.q_x[e]);  // the colour of the face's texel, from the pool
                V3 origin = cam_pos;
                if (r.depth > 0 || dof) {
                    const float4 qo = ws.q_o[0][e];
                    origin = mk(qo.x, qo.y, qo.z);
                }
                const uint32_t lit = s_lit[threadIdx.x];
                MCRT_HOOK_LIT_SHADED(lit, r)
                const float vis = (mode == SHADOW_SOFT) ? static_cast<float>(lit) / static_cast<float>(S) : (lit ? 1.0f : 0.0f);
                C4 c = shade(scg, hit, normalize(origin - hit.p), vis);
                if (cfg.ao_enabled && r.depth == 0) {  // occluded count from the ao stage (e < cap: a primary hit)
                    const float ao = 1.0f - static_cast<float>(ws.lit[1][e]) / static_cast<float>(cfg.ao_samples);
                    const float kk = 1.0f - cfg.ao_intensity * (1.0f - ao);
                    c.r *= kk;
                    c.g *= kk;
                    c.b *= kk;
                }
                ws.stack[static_cast<size_t>(r.depth) * ws.cap + r.root] = make_float4(c.r, c.g, c.b, c.a);  // plane-major: [depth][sample slot]
            }
        }
    }
}

// ambient occlusion (raytracer.cpp:38-78, depth 0 only) as one stage over the primary hits, a lane per hit:
//   1. the meshes any of its rays can meet: within the radius (rt::ball_candidates) and not behind the hit's own
//      face (the rays leave into the hemisphere around the normal; for an axis-aligned normal the tangent frame
//      is made of exact axis vectors, so the component of every direction along the normal is >= 0 and a box that
//      ends before the origin on that axis is missed by the moving-away argument of rt::bundle_decide_mesh);
//   2. hits with no such mesh are done (0 occluded) — the others are packed onto the block's first lanes;
//   3. per packed lane: mt19937(ao seed), and for each of the A directions the two draws, the cosine-weighted
//      direction and the any-hit test, counted in a register.
// (As two kernels — directions to HBM a lane per hit, rays a lane per (hit, direction) — the stage moved
// 2 x 192 B per hit through HBM with 64 cache lines per store instruction, and every ray lane reloaded and
// re-normalised its hit: 5.7 ms of the reference GUI's default frame; see DESIGN.md.)
#ifndef MCRT_AO_WAVES
#define MCRT_AO_WAVES 4
#endif
template <int kView>
__global__ __launch_bounds__(kBlock, MCRT_AO_WAVES) void ao_kernel(const uint8_t* __restrict__ scene_blob, const RenderParams p) {
    __shared__ int s_wcnt[kBlock / 64];
    __shared__ uint32_t s_list[kBlock];
    __shared__ unsigned long long s_mask[kBlock];
    extern __shared__ __align__(16) unsigned char s_dyn[];
    const SceneView scg = view_of(scene_blob);
    const WaveSpace& ws = p.ws;
    if (blockIdx.x >= counted(ws, kCntUnits)) return;
    const typename ViewSel<kView>::type sc = ViewSel<kView>::make(scg, p, s_dyn);
    constexpr bool kPosed = kView != kViewLdsUnposed;
    const bool posed = kPosed && p.scene_posed != 0;
    const int A = p.cfg.ao_samples;
    const float radius = p.cfg.ao_radius;
    uint32_t* occ_out = ws.lit[1];
    for_each_unit_block(ws, [&](uint32_t first, uint32_t n) __attribute__((always_inline)) {
        bool traced = false;
        if (threadIdx.x < n) {
            const uint32_t e = first + threadIdx.x;
            V3 P, Nraw;
            load_point_normal(ws, posed, e, P, Nraw);
            const V3 N = normalize(Nraw);
            const unsigned long long cand = hemisphere_candidates<kPosed>(scg, P + N * 1e-3f, N, radius);
            s_mask[threadIdx.x] = cand;
            traced = cand != 0ull || scg.n_meshes > 64;  // meshes beyond the mask are tested per ray
            if (!traced) occ_out[e] = 0u;
        }
        int total = 0;
        const int rank = block_rank(traced, s_wcnt, total);
        if (traced) s_list[rank] = threadIdx.x;
        __syncthreads();
        if (static_cast<int>(threadIdx.x) < total) {
            const uint32_t k = s_list[threadIdx.x];
            const uint32_t e = first + k;
            V3 P, Nraw;
            load_point_normal(ws, posed, e, P, Nraw);
            const V3 N = normalize(Nraw);
            const V3 T = (__builtin_fabsf(N.x) < 0.9f) ? normalize(cross(mk(1, 0, 0), N)) : normalize(cross(mk(0, 1, 0), N));
            const V3 B = cross(N, T);
            const unsigned long long cand = s_mask[k];
            const V3 O = P + N * 1e-3f;
            const unsigned long long inside = p.inside_fast ? origin_inside_boxes(sc, O, cand) : 0ull;  // every ray of the hit starts there
            MtShort rng;
            const uint32_t seed = ao_seed(P);
            if (p.seed_table_full)
                rng.seed_known(seed, p.seed_table_full[seed]);  // mt[397] of this seed: one load instead of the 397-step recurrence
            else
                rng.seed(seed);
            uint32_t occluded = 0;
            for (int i = 0; i < A; ++i) {
                const float r1 = rng.uniform();
                const float r2 = rng.uniform();
                const float sinT = sqrt_pos(1.0f - r1);
                const float cosT = sqrt_pos(r1);
                float sn, cs;
                mcrt_sincosf(kTwoPi * r2, &sn, &cs);
                const V3 local = mk(sinT * cs, cosT, sinT * sn);
                const V3 world = normalize(T * local.x + N * local.y + B * local.z);
                if (any_hit_masked(sc, Ray{O, world}, radius, cand, inside)) ++occluded;
            }
            occ_out[e] = occluded;
        }
        __syncthreads();  // s_list, s_mask are reused by the next block of hits
    });
}

// level_shade (general variants): colour of the level, reflection ray, closest hit of the next level →
// survivors to the queue of level + 1 with their light samples (packed); ended chains fold their level
// colours back to front and write the sample's colour.
template <int kView>
__global__ __launch_bounds__(kBlock, 2) void level_shade_kernel(const uint8_t* __restrict__ scene_blob, const RenderParams p,
                                                              const int level) {
    constexpr bool kGeneral = true;
    __shared__ int s_wcnt[kBlock / 64];
    __shared__ uint32_t s_out_base;
    __shared__ float4 s_np[kBlock], s_nn[kBlock];  // new hits handed to the packed threads
    extern __shared__ __align__(16) unsigned char s_dyn[];
    const SceneView scg = view_of(scene_blob);
    const WaveSpace& ws = p.ws;
    const Scope scope{level, 0};
    if (no_entry_blocks(ws, scope)) return;  // before the collective staging
    const typename ViewSel<kView>::type sc = ViewSel<kView>::make(scg, p, s_dyn);
    const mcrt_config& cfg = p.cfg;
    const int par = level & 1;
    const int mode = shadow_mode(sc, cfg);
    const int S = cfg.shadow_samples;
    const float* fb = sc.hdr->background;
    const C4 flat_bg{fb[0], fb[1], fb[2], fb[3]};
    uint32_t* my_rng = nullptr;
    if (ws.hit_rng) my_rng = ws.hit_rng + (static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x) * 624;
    for_each_entry_block(ws, scope, [&](uint32_t first, uint32_t n) __attribute__((always_inline)) {
        bool next_hit = false;
        Ray nray{mk(0, 0, 0), mk(0, 0, 0)};
        Hit nhit;
        nhit.hit = false;
        uint32_t root = 0;
        int depth = 0;
        if (threadIdx.x < n) {
            const uint32_t e = first + threadIdx.x;
            const Record r = load_record(ws, par, e, true);
            root = r.root;
            depth = r.depth;
            const uint32_t lit = ws.lit[par][e];
            const float vis =
                (mode == SHADOW_SOFT) ? static_cast<float>(lit) / static_cast<float>(S) : (lit ? 1.0f : 0.0f);
            const C4 c = level_color(sc, cfg, r.ray.o, r.hit, depth, vis, my_rng);
            bool done = false;
            C4 tail = flat_bg;
            if (depth >= cfg.max_bounces) {  // no reflection: `shadedColor.a = originalAlpha; return clamp()`
                tail = clamp4(c);
                done = true;
            } else {
                nray = reflect_ray(r.ray, r.hit);
                nhit = hit_scene<true>(sc, nray, ~0ull);
                if (nhit.hit) {  // the chain goes on: its level colour waits on the stack for the fold
                    ws.stack[static_cast<size_t>(depth) * ws.cap + root] = make_float4(c.r, c.g, c.b, c.a);
                    next_hit = true;
                } else {  // bounced ray missed → flat background (raytracer.cpp:94-102), folded in at once (:143-147)
                    tail = fold_reflection(c, flat_bg);
                    done = true;
                }
            }
            if (done) {  // unwind the recursion: fold the shallower levels' colours back to front
                for (int d = depth - 1; d >= 0; --d) {
                    const float4 sd = ws.stack[static_cast<size_t>(d) * ws.cap + root];
                    tail = fold_reflection(C4{sd.x, sd.y, sd.z, sd.w}, tail);
                }
                ws.scol[root] = make_float4(tail.r, tail.g, tail.b, tail.a);
            }
        }
        // survivors → dense level+1 queue: one atomic per block
        int total = 0;
        const int rank = block_rank(next_hit, s_wcnt, total);
        if (total > 0) {  // uniform
            if (threadIdx.x == 0) s_out_base = count_add(ws, kCntDense + level + 1, static_cast<uint32_t>(total));
            const bool soft = mode == SHADOW_SOFT;
            if (next_hit && soft) {  // hand the new hit to thread `rank`: the survivors' work below runs packed
                s_np[rank] = make_float4(nhit.p.x, nhit.p.y, nhit.p.z, __int_as_float(depth + 1));
                s_nn[rank] = make_float4(nhit.n.x, nhit.n.y, nhit.n.z, 0.0f);
            }
            __syncthreads();
            if (next_hit) push_entry(ws, par ^ 1, s_out_base + static_cast<uint32_t>(rank), nray, nhit, root, depth + 1);
            // The new entries' light samples (seeding chain + S samples) and bundle masks.  Only a few lanes
            // per wave survive; threads 0 .. total-1 do it instead, so the chain runs in ceil(total / 64) waves.
            if (soft && static_cast<int>(threadIdx.x) < total) {
                const float4 np = s_np[threadIdx.x], nn = s_nn[threadIdx.x];
                typename SampleRng<kGeneral>::type rng;
                const V3 P = mk(np.x, np.y, np.z);
                seed_light_rng<kGeneral>(rng, P, __float_as_int(np.w), S, my_rng);
                write_light_samples<kView != kViewLdsUnposed>(scg, ws, s_out_base + threadIdx.x, P, mk(nn.x, nn.y, nn.z), S, rng);
            }
            __syncthreads();
        }
    });
}

// resolve: ordered per-pixel sum of the queued units' sample colours (tile_renderer.cpp:116-124).  A
// sample that started a chain gets its colour here: the chain's level colours folded back to front
// (raytracer.cpp:143-147) from where the chain ended — the flat background when its last reflection ray
// missed (:94-102), the clamped last level colour when it stopped at maxBounces (:146-147).
// A thread per SAMPLE fetches / folds the colour; the pixel's samples meet in LDS and one thread per
// pixel adds them in order (float addition order is part of the result).
__device__ __forceinline__ float4 sample_colour(const WaveSpace& ws, uint32_t slot, uint32_t code, const C4& flat_bg) {
    if (code == 0u) return ws.scol[slot];
    const float4* lv = ws.stack + slot;  // level d of the chain: lv[d * cap]
    const size_t plane = ws.cap;
    C4 tail = flat_bg;
    int dd = static_cast<int>(code >> 1) - 1;  // the chain's last record
    if (code & 1u) {
        const float4 last = lv[dd * plane];
        tail = clamp4(C4{last.x, last.y, last.z, last.w});
        --dd;
    }
    for (; dd >= 0; --dd) {
        const float4 sd = lv[dd * plane];
        tail = fold_reflection(C4{sd.x, sd.y, sd.z, sd.w}, tail);
    }
    return make_float4(tail.r, tail.g, tail.b, tail.a);
}
__global__ __launch_bounds__(kBlock) void resolve_kernel(const uint8_t* __restrict__ scene_blob, const float* __restrict__ tile_draws,
                                                         float4* __restrict__ out_frame, uchar4* __restrict__ out8, const RenderParams p,
                                                         const int tile_base) {
    __shared__ float4 s_col[kBlock];
    const WaveSpace& ws = p.ws;
    const mcrt_config& cfg = p.cfg;
    const SceneView scg = view_of(scene_blob);
    const FrameDiv fd(p);
    const uint32_t dd = static_cast<uint32_t>(p.draws_per_sample);
    // the colour of sample `sidx` (stream order) of unit d: a miss from its jitter pair — the expressions of `primary`'s
    // sample loop (tile_renderer.cpp:93-114) — else the chain's fold
    auto unit_sample = [&](const uint4& d, const TileGeom& tg, const float* draws, uint32_t sidx, uint32_t spp_) __attribute__((always_inline)) -> float4 {
        const uint32_t slot = d.w + sidx;
        const uint32_t code = ws.end[slot];
        if (code != kEndMiss) return sample_colour(ws, slot, code, C4{scg.hdr->background[0], scg.hdr->background[1], scg.hdr->background[2], scg.hdr->background[3]});
        const unsigned pix = d.y + UDiv(spp_).div(sidx);
        const unsigned uly = UDiv(static_cast<unsigned>(tg.w)).div(pix);
        const int px = tg.x + static_cast<int>(pix - uly * static_cast<unsigned>(tg.w)), py = tg.y + static_cast<int>(uly);
        float jx = 0.5f, jy = 0.5f;
        if (spp_ > 1u) {
            const float2 j = *reinterpret_cast<const float2*>(draws + (static_cast<size_t>(d.y) * spp_ + sidx) * dd);  // dd is even: 8-byte aligned
            jx = j.x, jy = j.y;
        }
        const C4 c = background(scg, cfg, fd.u(static_cast<float>(px) + jx), fd.v(static_cast<float>(py) + jy));
        return make_float4(c.r, c.g, c.b, c.a);
    };
    const uint32_t n_units = ws.frame_info[0];  // (not from the counters: this kernel moves their base)
    const uint32_t spp = cfg.samples_per_pixel > 1 ? static_cast<uint32_t>(cfg.samples_per_pixel) : 1u;
    const float inv_spp = 1.0f / static_cast<float>(spp);
    const uint32_t chunk_px = spp <= static_cast<uint32_t>(kBlock) ? static_cast<uint32_t>(kBlock) / spp : 0u;  // pixels per pass (0: a pixel per thread, serially)
    auto put_pixel = [&](const TileGeom& tg, uint32_t i, float4 acc) __attribute__((always_inline)) {
        const uint32_t uly = i / static_cast<uint32_t>(tg.w);
        const int ly = static_cast<int>(uly);
        const int lx = static_cast<int>(i - uly * static_cast<uint32_t>(tg.w));
        const int row = (p.layout == MCRT_LAYOUT_PACKED) ? ((p.shard.pack_first + tg.owned_row * p.shard.pack_step) * cfg.tile_size + ly) : (tg.y + ly);
        store_pixel(out_frame, out8, static_cast<size_t>(row) * cfg.width + (tg.x + lx),
                    make_float4(acc.x * inv_spp, acc.y * inv_spp, acc.z * inv_spp, acc.w * inv_spp));
    };
    MCRT_HOOK_RESOLVE_BEGIN()
    if (blockIdx.x == 0) {  // the pass has counted everything: where the counters stand is the next pass's base
        for (int i = threadIdx.x; i < kCounterWords - 4; i += kBlock) ws.counter_base[i] = ws.counters[i];
    }
    for (uint32_t u = blockIdx.x; u < n_units; u += gridDim.x) {
        const uint4 d = ws.units[u];
        const TileGeom tg = tile_of(p, static_cast<int>(d.x));
        const uint32_t pp0 = d.y, pp1 = d.z;
        // the tile's draws: at its touched-tile number, or (all tiles' streams in HBM) at its index in the batch — as in `primary`
        const float* draws = tile_draws + static_cast<size_t>(p.bg_in_plan ? d.w / ws.tile_slots : static_cast<uint32_t>(static_cast<int>(d.x) - tile_base)) * ws.draws_stride;
        if (chunk_px == 0u) {
            for (uint32_t i = pp0 + threadIdx.x; i < pp1; i += kBlock) {
                float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                for (uint32_t s = 0; s < spp; ++s) {
                    const float4 c = unit_sample(d, tg, draws, (i - pp0) * spp + s, spp);
                    acc.x += c.x, acc.y += c.y, acc.z += c.z, acc.w += c.w;
                }
                put_pixel(tg, i, acc);
            }
            continue;
        }
        for (uint32_t p0 = pp0; p0 < pp1; p0 += chunk_px) {  // uniform
            const uint32_t npx = min(chunk_px, pp1 - p0);
            if (threadIdx.x < npx * spp) s_col[threadIdx.x] = unit_sample(d, tg, draws, (p0 - pp0) * spp + threadIdx.x, spp);
            __syncthreads();
            if (threadIdx.x < npx) {
                float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                for (uint32_t s = 0; s < spp; ++s) {
                    const float4 c = s_col[threadIdx.x * spp + s];
                    acc.x += c.x, acc.y += c.y, acc.z += c.z, acc.w += c.w;
                }
                put_pixel(tg, p0 + threadIdx.x, acc);
            }
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------------
// small kernels
// ---------------------------------------------------------------------------------------------
template <class Pixel>  // float4, or uchar4 for the RGBA8 plane
__global__ void unpack_rows_kernel(mcrt_config cfg, Shard sh, const Pixel* packed, Pixel* frame) {
    // one thread per packed pixel
    size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    int W = cfg.width, T = cfg.tile_size;
    size_t prow = i / W;
    int x = static_cast<int>(i - prow * W);
    int k = static_cast<int>(prow / T);
    int ly = static_cast<int>(prow - static_cast<size_t>(k) * T);
    if (k >= sh.owned_rows) return;
    int y = (sh.first + k * sh.step) * T + ly;
    if (y >= cfg.height) return;
    frame[static_cast<size_t>(y) * W + x] = packed[i];
}

// every rank's packed rows (rank-major, `rank_stride` float4 apart) → the frame, one thread per output
// pixel: tile row r belongs to rank r mod world and is that rank's (r div world)-th packed tile row
__global__ void assemble_frame_kernel(mcrt_config cfg, int world, const float4* __restrict__ gathered, size_t rank_stride,
                                      float4* __restrict__ frame) {
    const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int W = cfg.width, T = cfg.tile_size;
    if (i >= static_cast<size_t>(W) * cfg.height) return;
    const int y = static_cast<int>(i / W);
    const int x = static_cast<int>(i - static_cast<size_t>(y) * W);
    const int r = y / T, ly = y - r * T;
    const int rank = r % world, k = r / world;
    frame[i] = gathered[static_cast<size_t>(rank) * rank_stride + (static_cast<size_t>(k) * T + ly) * W + x];
}

// the device's seed table (kernels.h): entry i = mt[397] of std::mt19937(i - kSeedWindowHalf)
__global__ __launch_bounds__(256) void seed_table_kernel(uint32_t* __restrict__ table) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < kSeedWindow) table[i] = MtShort::word397(i - kSeedWindowHalf);
}

// the full table: entry s = mt[397] of std::mt19937(s), for the seeds first + (this thread)
__global__ __launch_bounds__(256) void seed_table_range_kernel(uint32_t* __restrict__ table, uint32_t first) {
    const uint32_t s = first + blockIdx.x * 256u + threadIdx.x;
    table[s] = MtShort::word397(s);
}

__global__ void quantize_kernel(const float4* rgba, uchar4* out, size_t n) {  // image_writer.cpp:18-22
    size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = quantize_pixel(rgba[i]);
}

__global__ void probe_intersect_kernel(const uint8_t* scene, const float* rays, int n, mcrt_hit* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    SceneView sc = view_of(scene);
    Ray r{ld3(rays + 6 * i), ld3(rays + 6 * i + 3)};
    Hit h = hit_scene(sc, r, ~0ull);
    mcrt_hit o;
    o.hit = h.hit ? 1 : 0;
    o.t = h.t;
    o.point[0] = h.p.x, o.point[1] = h.p.y, o.point[2] = h.p.z;
    o.normal[0] = h.n.x, o.normal[1] = h.n.y, o.normal[2] = h.n.z;
    o.texture_color[0] = h.tex.r, o.texture_color[1] = h.tex.g, o.texture_color[2] = h.tex.b,
    o.texture_color[3] = h.tex.a;
    o.is_outer_layer = h.outer ? 1 : 0;
    out[i] = o;
}

__global__ void probe_trace_kernel(const uint8_t* scene, mcrt_config cfg, const float* rays, int n, int depth,
                                   float* out, uint32_t* hit_rng, float* deep_stack) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    SceneView sc = view_of(scene);
    Ray r{ld3(rays + 6 * i), ld3(rays + 6 * i + 3)};
    C4 local_stack[kMaxStack];
    C4* stack = deep_stack ? reinterpret_cast<C4*>(deep_stack) + static_cast<size_t>(i) * max(cfg.max_bounces, 1)
                           : local_stack;
    uint32_t* rng = hit_rng ? hit_rng + static_cast<size_t>(i) * 624 : nullptr;
    C4 c;
    if (depth > cfg.max_bounces) {
        c = background(sc, cfg, 0.5f, 0.5f);
    } else {
        Hit h = hit_scene(sc, r, ~0ull);
        if (!h.hit) {
            const float* b = sc.hdr->background;
            c = (depth == 0) ? background(sc, cfg, 0.5f, 0.5f) : C4{b[0], b[1], b[2], b[3]};
        } else {
            c = trace_from_hit(sc, cfg, r, h, depth, stack, rng);
        }
    }
    out[4 * i + 0] = c.r, out[4 * i + 1] = c.g, out[4 * i + 2] = c.b, out[4 * i + 3] = c.a;
}

__global__ void probe_mt_kernel(const uint32_t* seeds, int n_seeds, int n_draws, float* out, uint32_t* storage) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_seeds) return;
    HitRng g;
    g.seed(seeds[i], n_draws, storage ? storage + static_cast<size_t>(i) * 624 : nullptr);
    for (int k = 0; k < n_draws; ++k) out[static_cast<size_t>(i) * n_draws + k] = g.uniform();
}

__device__ __forceinline__ float detmath_op(int op, float x, float y) {
    if (op == 3 || op == 4) {  // the fused form: .s / .c
        float sn, cs;
        mcrt_sincosf(x, &sn, &cs);
        return op == 3 ? sn : cs;
    }
    if (op == 5) return rcp_exact(x);  // held against the host's IEEE 1.0f / x
    return op == 0 ? mcrt_sinf(x) : (op == 1 ? mcrt_cosf(x) : mcrt_powf(x, y));
}
__global__ void probe_detmath_kernel(int op, const float* x, const float* y, size_t n, float* out) {
    size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = detmath_op(op, x[i], y ? y[i] : 0.0f);
}
__global__ void probe_detmath_range_kernel(int op, uint32_t lo_bits, uint64_t count, float y0, float* out) {
    uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= count) return;
    out[i] = detmath_op(op, mcrt_u2f(lo_bits + static_cast<uint32_t>(i)), y0);
}

// div_frame against the general division for the divisors d_first .. d_first + gridDim.y - 1 and every float x in
// {0} and [2^-33, d + 1] (by bit pattern).  counts[0] += mismatches, counts[1] = a failing divisor
// rds: the divisors' reciprocals as the HOST forms them (what the render kernels are given: RenderParams::inv_width / inv_height)
__global__ void probe_div_const_kernel(uint32_t d_first, int mode, const float* __restrict__ rds, unsigned long long* counts) {
    const float d = static_cast<float>(d_first + blockIdx.y);
    const float rd = rds[blockIdx.y];
    if (mode == 4) {  // the device's own 1.0f / d against the host's reciprocal
        if (blockIdx.x == 0 && threadIdx.x == 0 && __float_as_uint(1.0f / d) != __float_as_uint(rd)) {
            atomicAdd(&counts[0], 1ull);
            counts[1] = d_first + blockIdx.y;
        }
        return;
    }
    if (mode == 3) {  // rt::sqrt_pos against sqrtf for 0 and every float from 2^-96 to infinity (the divisor plays no part)
        unsigned long long bad = 0;
        const uint32_t lo = 0x0f800000u /* 2^-96 */, hi = 0x7f800000u;
        for (uint64_t b = static_cast<uint64_t>(lo) + static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; b <= hi + 1ull;
             b += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
            const float x = b > hi ? 0.0f : __uint_as_float(static_cast<uint32_t>(b));
            if (__float_as_uint(__builtin_sqrtf(x)) != __float_as_uint(sqrt_pos(x))) ++bad;
        }
        if (bad) atomicAdd(&counts[0], bad);
        return;
    }
    const uint32_t lo = 0x2f000000u /* 2^-33 */, hi = __float_as_uint(d + 1.0f);
    unsigned long long bad = 0;
    for (uint64_t b = static_cast<uint64_t>(lo) + static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; b <= hi + 1ull;
         b += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        const float x = b > hi ? 0.0f : __uint_as_float(static_cast<uint32_t>(b));
        const float want = x / d;
        const float got = mode == 2 ? x * rd : (mode ? div_frame2(x, d, rd) : div_frame(x, d, rd));  // mode 1: two corrections; mode 2: none (the probe's own check)
        if (__float_as_uint(want) != __float_as_uint(got)) ++bad;
    }
    if (bad) {
        atomicAdd(&counts[0], bad);
        counts[1] = d_first + blockIdx.y;
    }
}

// ---------------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------------
hipError_t launch_probe_div_const(uint32_t d_first, uint32_t d_count, int mode, const float* host_reciprocals, unsigned long long* counts, hipStream_t stream) {
    if (d_count == 0) return hipSuccess;
    hipLaunchKernelGGL(probe_div_const_kernel, dim3(mode == 4 ? 1 : 2048, d_count), dim3(256), 0, stream, d_first, mode, host_reciprocals, counts);
    return hipGetLastError();
}

Shard make_shard(const mcrt_config& cfg, int first, int step) {
    Shard s{};
    s.first = first;
    s.step = step < 1 ? 1 : step;
    s.pack_first = 0;
    s.pack_step = 1;
    if (cfg.width <= 0 || cfg.height <= 0 || cfg.tile_size <= 0) return s;
    s.tiles_x = (cfg.width + cfg.tile_size - 1) / cfg.tile_size;
    s.tiles_y = (cfg.height + cfg.tile_size - 1) / cfg.tile_size;
    s.owned_rows = (first < s.tiles_y && first >= 0) ? (s.tiles_y - first + s.step - 1) / s.step : 0;
    return s;
}

// dynamic LDS of the scene tables a kernel stages (stage_tables): face table, mesh table, alpha predicates
static size_t scene_table_bytes(const RenderParams& p) {
    return p.scene_in_lds ? static_cast<size_t>(p.lds_face_entries) * 16 + static_cast<size_t>(p.lds_face_entries / 6) * kMeshTabWords * 4 +
                                static_cast<size_t>(p.lds_alpha_words) * 4
                          : 0;
}
static int owned_tiles(const RenderParams& p) { return p.shard.owned_rows * p.shard.tiles_x; }
static bool soft_sampling(const mcrt_config& c) { return c.soft_shadows && c.shadow_samples > 1; }

// The flat pipeline keeps the records of every level at once: its arrays are laid out for up to
// kFlatMaxBounces reflection levels (1 + maxBounces records per sample slot in the worst case).
constexpr int kFlatMaxBounces = 8;
#ifndef MCRT_LIT_LDS_KB
#define MCRT_LIT_LDS_KB 25
#endif
constexpr size_t kLitLdsBytes = MCRT_LIT_LDS_KB * 1024;  // `lit`: LDS for the sample positions of the records whose rays are traced, per pass
// rare features that need the general kernel variants (one launch set per level, ping-pong queues): per-hit RNG
// streams longer than the register-only engine covers (they run AO inside `level_shade`, sequentially), or
// more bounces than the flat record arrays are laid out for
static bool needs_general_variant(const mcrt_config& c) {
    return (c.ao_enabled && (c.ao_samples <= 0 || 2 * c.ao_samples > kMtShortMax)) || (soft_sampling(c) && 2 * c.shadow_samples > kMtShortMax) ||
           c.max_bounces > kFlatMaxBounces;
}

// Parts of a tile that meshes can touch: one 256-sample chunk each, at most 16 — fine enough that
// the few tiles holding the character spread over the chip.  Every part reads its samples' draws from
// the tile's stream in HBM (plan_tiles).  Background tiles are never split.
static int choose_parts_per_tile(const RenderParams& p) {
    const mcrt_config& cfg = p.cfg;
    const long long spp = cfg.samples_per_pixel > 1 ? cfg.samples_per_pixel : 1;
    const long long tile_items = p.rect_w > 0 ? static_cast<long long>(p.rect_w) * p.rect_h * spp : static_cast<long long>(cfg.tile_size) * cfg.tile_size * spp;
    long long parts = (tile_items + kChunk - 1) / kChunk;
    const long long most = p.rect_w > 0 ? 4096 : 16;  // a renderTile rectangle may be as large as the frame: it is the launch's only tile
    if (parts > most) parts = most;
    if (parts < 1) parts = 1;
    return static_cast<int>(parts);
}

WorkspaceBytes plan_workspace(RenderParams& p, size_t budget_bytes, const int* row_touched) {
    WorkspaceBytes w{};
    const mcrt_config& c = p.cfg;
    const int n_tiles = owned_tiles(p);
    p.parts_per_tile = choose_parts_per_tile(p);
    p.flat = needs_general_variant(c) ? 0 : 1;
    p.ws.stack_stride = c.max_bounces > 0 ? c.max_bounces + 1 : 1;
    const size_t spp = c.samples_per_pixel > 1 ? c.samples_per_pixel : 1;
    // every tile that meshes can touch owns one slot per sample of a full (frame-clipped) tile
    const size_t tile_w = p.rect_w > 0 ? static_cast<size_t>(p.rect_w) : static_cast<size_t>(c.tile_size < c.width ? c.tile_size : c.width);
    const size_t tile_h = p.rect_w > 0 ? static_cast<size_t>(p.rect_h) : static_cast<size_t>(c.tile_size < c.height ? c.tile_size : c.height);
    const size_t tile_slots = tile_w * tile_h * spp;
    const size_t S = soft_sampling(c) ? static_cast<size_t>(c.shadow_samples) : 0;
    const size_t A = c.ao_enabled && c.ao_samples > 0 ? static_cast<size_t>(c.ao_samples) : 0;
    const size_t rays = S > A ? S : A;  // light samples and AO directions share one array
    // records per slot: flat — the primary hit and one per reflection level; general — two ping-pong queues
    const size_t recs = p.flat ? static_cast<size_t>(1 + (c.max_bounces > 0 ? c.max_bounces : 0)) : 2;
    // bytes per slot: colour + end code, per record 5 float4 + light samples + mask + lit, AO counts, stack
    // light sample positions and bundle masks in HBM: general variants (per record) and the AO directions of
    // the primary hits; the flat pipeline's `lit` keeps the light samples in LDS
    const size_t hbm_rays = p.flat ? 0 : rays;  // the flat pipeline keeps light samples in LDS and AO directions in registers
    const size_t ray_recs = p.flat ? 1 : recs;
    const size_t per_entry = 16 + 4 + recs * (5 * 16 + 4) + ray_recs * (12 * hbm_rays + (hbm_rays ? 8 : 0) + 4) + 4 + 16 * static_cast<size_t>(p.ws.stack_stride);
    // `lit`: a round is a block of up to 256 records (masks, counts and the list of traced records in LDS); the sample
    // positions of the records whose rays are traced go through an area of kLitLdsBytes, `lit_pass` records at a time
    {
        const size_t pairs = S ? S : 1;
        size_t pass = kLitLdsBytes / (12 * pairs);
        if (pass > static_cast<size_t>(kBlock)) pass = kBlock;
        if (pass < 1) pass = 1;
        p.lit_round = kBlock;
        p.lit_pass = static_cast<int>(pass);
        p.lit_lds_bytes = static_cast<int>(pass * 12 * pairs + static_cast<size_t>(kBlock) * 24);  // positions + (candidates, inside, lit count, traced list) per record of a round
        p.lit_lds_offset = static_cast<int>((scene_table_bytes(p) + 15) & ~static_cast<size_t>(15));
    }
    const int owned = p.shard.owned_rows;
    auto row_count = [&](int j) -> size_t { return row_touched ? static_cast<size_t>(row_touched[j]) : static_cast<size_t>(p.shard.tiles_x); };
    // touched tiles of the fullest batch when the shard is cut into batches of R owned rows
    auto fullest = [&](int R) -> size_t {
        size_t mx = 0;
        for (int r0 = 0; r0 < owned; r0 += R) {
            size_t sum = 0;
            for (int j = r0; j < owned && j < r0 + R; ++j) sum += row_count(j);
            if (sum > mx) mx = sum;
        }
        return mx;
    };
    // the tiles' jitter / lens draws: of the touched tiles only when `plan_tiles` renders the background tiles
    // itself (few draws per pixel: a 624-word twist then completes >= 26 pixels, enough for whole rounds of
    // the wave's 64 lanes), of every tile of the batch otherwise
    const size_t draws_stride = tile_slots * static_cast<size_t>(p.draws_per_sample);
    p.bg_in_plan = spp * static_cast<size_t>(p.draws_per_sample) <= 24 ? 1 : 0;
    static const size_t slab_min_spp = [] {  // development knob (the parity sweeps run the slab kernel at every sample count with it)
        const char* e = getenv("MCRT_SLAB_MIN_SPP");
        const int v = e ? atoi(e) : 0;
        return static_cast<size_t>(v > 0 ? v : kSlabMinSpp);
    }();
    p.bg_kernel = (!p.bg_in_plan && spp >= slab_min_spp) ? 1 : 0;
    const size_t draws_row_bytes = p.bg_in_plan ? 0 : draws_stride * 4 * static_cast<size_t>(p.shard.tiles_x);
    const size_t draws_tile_bytes = p.bg_in_plan ? draws_stride * 4 : 0;
    p.ws.draws_stride = static_cast<uint32_t>(draws_stride > 0xffffffffull ? 0xffffffffull : draws_stride);
    p.ws.tile_slots = static_cast<uint32_t>(tile_slots > 0xffffffffull ? 0xffffffffull : tile_slots);
    // records are indexed with 32 bits (with room for the 3·S multiplier done in size_t)
    const size_t index_limit = 0x7ffffff0ull / recs;
    size_t tile_budget = budget_bytes / (per_entry * (tile_slots ? tile_slots : 1) + draws_tile_bytes);
    if (tile_slots && tile_budget > index_limit / tile_slots) tile_budget = index_limit / tile_slots;
    int rows = owned > 0 ? owned : 1;
    // a batch of R rows fits when its hit workspace and its draws fit the budget together
    auto fits = [&](int R) -> bool {
        const size_t f = fullest(R);
        if (f > tile_budget) return false;
        return f * (tile_slots * per_entry + draws_tile_bytes) + static_cast<size_t>(R) * draws_row_bytes <= budget_bytes;
    };
    if (owned > 0 && !fits(rows)) {  // largest R that fits (monotone in R)
        int lo = 1, hi = owned;
        while (lo < hi) {
            const int mid = (lo + hi + 1) / 2;
            if (fits(mid)) lo = mid; else hi = mid - 1;
        }
        rows = lo;  // a batch is never smaller than one tile row
    }
    size_t cap_tiles = owned > 0 ? fullest(rows) : 0;
    if (cap_tiles < 1) cap_tiles = 1;
    p.rows_per_batch = rows;
    if (tile_slots == 0 || cap_tiles > index_limit / tile_slots || draws_stride > 0xffff0000ull)
        p.rows_per_batch = 0;  // one tile (row) alone exceeds the 32-bit index ranges: refused by the caller
    const size_t cap = p.rows_per_batch ? cap_tiles * tile_slots : 1;
    // (+ 256 x recs: `lit`'s chase regions start at the level-1 count rounded up to a whole block)
    const size_t rec_cap = cap * recs + (p.flat ? static_cast<size_t>(kBlock) * recs : 0);
    p.ws.cap = static_cast<uint32_t>(cap);
    p.ws.tile_cap = static_cast<uint32_t>(cap_tiles);
    {  // the tile streams' parts (tile_stream_wave): four waves per tile unless the stream is too short for that
        static const int parts_knob = [] {  // development knob: MCRT_STREAM_PARTS=1 / 2 / 4
            const char* e = getenv("MCRT_STREAM_PARTS");
            const int v = e ? atoi(e) : 0;
            return (v == 1 || v == 2 || v == 4) ? v : 0;
        }();
        const size_t twists = (draws_stride + 623) / 624;  // of a full tile's stream
        int parts = parts_knob ? parts_knob : kStreamWaves;
        while (parts > 1 && twists < static_cast<size_t>(2 * parts)) parts >>= 1;  // at least two twists per part
        p.stream_parts = p.draws_per_sample > 0 ? parts : 1;
        p.stream_part_twists = static_cast<int>((twists + static_cast<size_t>(p.stream_parts) - 1) / static_cast<size_t>(p.stream_parts));
        if (p.stream_part_twists < 1) p.stream_part_twists = 1;
    }
    w.tile_rng = p.draws_per_sample > 0 ? static_cast<size_t>(n_tiles) * 624 * 4 * static_cast<size_t>(p.stream_parts) : 0;
    w.tile_draws = static_cast<size_t>(rows) * draws_row_bytes + cap_tiles * draws_tile_bytes;
    w.scol = cap * 16;
    w.end = cap * 4;
    p.ws.unit_cap = static_cast<uint32_t>(cap_tiles * static_cast<size_t>(p.parts_per_tile));
    w.units = static_cast<size_t>(p.ws.unit_cap) * 16;
    w.unit_hits = static_cast<size_t>(p.ws.unit_cap) * 4;
    w.tile_mask = static_cast<size_t>(n_tiles > 0 ? n_tiles : 1) * 8;
    w.queue_each = rec_cap * 16;
    w.texel_refs = p.flat ? rec_cap * 4 : 4;
    w.targets = (p.flat ? cap : rec_cap) * 12 * hbm_rays;
    w.cand = hbm_rays ? (p.flat ? cap : rec_cap) * 8 : 0;
    w.lit0 = p.flat ? 4 : cap * 4;  // the flat pipeline keeps the lit counts in LDS
    w.lit1 = cap * 4;
    w.stack = cap * 16 * static_cast<size_t>(p.ws.stack_stride);
    w.counters = (static_cast<size_t>(kCounterWords) * 2 + 4) * 4;  // the counters, their base (the previous pass's last values), frame_info
    w.hit_rng = p.flat ? 0 : static_cast<size_t>(256) * kBlock * 624 * 4;  // general grids are capped at 256 WGs
    return w;
}

static int grid_knob(const char* name, int fallback) {
    const char* e = getenv(name);
    const int v = e ? atoi(e) : 0;
    return v > 0 ? v : fallback;
}

// Workgroup caps of a render's launches.  Every kernel strides over device-side counts, so a cap changes nothing but
// the schedule.  A frame alone on the device finishes soonest with many workgroups per kernel (`lit`'s rounds differ in
// cost: 16 per CU balance better than 8, -7 us); frames that share the device — four handles in flight, or the lanes of
// one large frame — get through fastest with FEWER workgroups per kernel (4 per CU), which leaves CU slots to the other
// frames' kernels instead of queueing whole kernels behind each other (+4 % frames/s at 1080p; profiles/r03_experiments/grid_sweep*.txt).
void choose_grids(RenderParams& p, bool shared_device, bool company) {
    static const int queue_knob = grid_knob("MCRT_QUEUE_GRID", 0);
    static const int primary_knob = grid_knob("MCRT_PRIMARY_GRID", 0),
                     ao_knob = grid_knob("MCRT_AO_GRID", queue_knob), lit_knob = grid_knob("MCRT_LIT_GRID", queue_knob),
                     resolve_knob = grid_knob("MCRT_RESOLVE_GRID", 0);
    p.shared_device = shared_device ? 1 : 0;
    // `plan_tiles`: a tile's stream by as many waves as it has parts when the chain of twists is what the kernel waits for —
    // long streams (64 spp: 210-420 twists per tile; GUI defaults alone 4.50 -> 4.10 ms, 8K 18.2 -> 16.8), or a frame that has
    // no company at all, neither other frames nor lanes of its own (1080p: -7 us) — and by ONE wave otherwise: four times the waves bring four times the tile set-up, state
    // loads and partial rounds and take the slots that other frames' or lanes' kernels would fill (-8 % frames/s at 1080p
    // with four frames in flight, -7 % for 4K / 4 spp on three lanes; profiles/r03_experiments/stream_waves.txt)
    static const int waves_knob = grid_knob("MCRT_STREAM_WAVES", 0);  // development knob: 1 / 2 / 4
    const bool long_streams = p.stream_part_twists * p.stream_parts >= 128;
    p.stream_waves = (long_streams || !company) ? p.stream_parts : 1;
    if (waves_knob == 1 || waves_knob == 2 || waves_knob == 4) p.stream_waves = waves_knob < p.stream_parts ? waves_knob : p.stream_parts;
    if (p.stream_waves < 1) p.stream_waves = 1;
    p.grid_primary = primary_knob ? primary_knob : (shared_device ? kSharedGrid : kPrimaryGrid);
    p.grid_ao = ao_knob ? ao_knob : (shared_device ? kSharedGrid : kQueueGrid);
    p.grid_lit = lit_knob ? lit_knob : (shared_device ? kSharedGrid : kLitGridAlone);
    p.grid_resolve = resolve_knob ? resolve_knob : (shared_device ? kSharedGrid : kResolveGrid);
}

template <int kView>
static void launch_levels(const RenderParams& p, hipStream_t stream, size_t dyn) {
    const mcrt_config& c = p.cfg;
    const bool soft = soft_sampling(c);
    const int levels = c.max_bounces < 0 ? 0 : c.max_bounces + 1;
    if (levels < 1) return;
    constexpr bool posed = kView != kViewLdsUnposed;
    (void)soft;
    if (!p.flat) {  // general variants: one launch set per level; `level_shade` emits the light samples of the entries it appends
        const int grid = 256;
        for (int L = 0; L < levels; ++L) {
            if (soft && L == 0) hipLaunchKernelGGL((light_samples_kernel<true, posed>), dim3(grid), dim3(kBlock), 0, stream, p.scene, p, L);
            hipLaunchKernelGGL(shadow_kernel<kView>, dim3(grid), dim3(kBlock), dyn, stream, p.scene, p, L);
            hipLaunchKernelGGL(level_shade_kernel<kView>, dim3(grid), dim3(kBlock), dyn, stream, p.scene, p, L);
        }
        return;
    }
    const int ao_grid = p.grid_ao > 0 ? p.grid_ao : kQueueGrid, lit_grid = p.grid_lit > 0 ? p.grid_lit : kQueueGrid;
    if (c.ao_enabled && c.ao_samples > 0) {  // ahead of `lit`, whose last phase applies the AO factor
        hipLaunchKernelGGL(ao_kernel<kView>, dim3(ao_grid), dim3(kBlock), dyn, stream, p.scene, p);
    }
    hipLaunchKernelGGL(lit_kernel<kView>, dim3(lit_grid), dim3(kBlock), static_cast<size_t>(p.lit_lds_offset) + static_cast<size_t>(p.lit_lds_bytes), stream, p.scene, p);
}

hipError_t launch_seed_tiles(const RenderParams& p, hipStream_t stream) {
    const int n = owned_tiles(p);
    if (n <= 0 || p.draws_per_sample <= 0 || !p.tile_rng) return hipSuccess;
    hipLaunchKernelGGL(seed_tiles_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, p, n);
    if (p.stream_parts > 1) hipLaunchKernelGGL(advance_tiles_kernel, dim3((n + kStreamWaves - 1) / kStreamWaves), dim3(64 * kStreamWaves), 0, stream, p, n);
    return hipGetLastError();
}

hipError_t launch_render(const RenderParams& p, hipStream_t stream, const LaunchMarks* marks) {
    const int n = owned_tiles(p);
    if (n <= 0) return hipSuccess;
    const size_t dyn = scene_table_bytes(p);
    float4* out = reinterpret_cast<float4*>(p.out);
    uchar4* out8 = reinterpret_cast<uchar4*>(p.out8);
    if (static_cast<size_t>(p.lit_lds_offset) != ((dyn + 15) & ~static_cast<size_t>(15))) return hipErrorInvalidValue;  // set by plan_workspace's caller
    for (int r0 = 0; r0 < p.shard.owned_rows; r0 += p.rows_per_batch) {
        const int rows = p.rows_per_batch < p.shard.owned_rows - r0 ? p.rows_per_batch : p.shard.owned_rows - r0;
        const int tile_base = r0 * p.shard.tiles_x;
        const int batch_tiles = rows * p.shard.tiles_x;
        hipError_t e = hipSuccess;  // (no counter memset: the counters run on, `resolve` moves their base)
        hipLaunchKernelGGL(plan_tiles_kernel, dim3((batch_tiles * (p.stream_waves > 0 ? p.stream_waves : 1) + kStreamWaves - 1) / kStreamWaves), dim3(64 * kStreamWaves), 0, stream,
                           p.scene, p.tile_rng, p.ws.tile_draws, out, out8, p, tile_base, batch_tiles);
        if (marks && marks->after_plan && r0 == 0) {
            e = hipEventRecord(marks->after_plan, stream);
            if (e != hipSuccess) return e;
        }
        if (p.bg_kernel)  // the background tiles of a high-spp frame from their streams in HBM
            hipLaunchKernelGGL(background_kernel, dim3(batch_tiles < kQueueGrid ? batch_tiles : kQueueGrid), dim3(kBlock), 0, stream, p.scene, p.ws.tile_draws, out, out8, p, tile_base, batch_tiles);
        const int primary_grid = p.grid_primary > 0 ? p.grid_primary : kPrimaryGrid, resolve_grid = p.grid_resolve > 0 ? p.grid_resolve : kResolveGrid;
        const int pgrid = batch_tiles * p.parts_per_tile < primary_grid ? batch_tiles * p.parts_per_tile : primary_grid;
        if (p.scene_in_lds && !p.scene_posed) {
            hipLaunchKernelGGL(primary_kernel<kViewLdsUnposed>, dim3(pgrid), dim3(kBlock), dyn, stream, p.scene, p.ws.tile_draws, out, out8, p, tile_base, batch_tiles);
            launch_levels<kViewLdsUnposed>(p, stream, dyn);
        } else if (p.scene_in_lds) {
            hipLaunchKernelGGL(primary_kernel<kViewLds>, dim3(pgrid), dim3(kBlock), dyn, stream, p.scene, p.ws.tile_draws, out, out8, p, tile_base, batch_tiles);
            launch_levels<kViewLds>(p, stream, dyn);
        } else {
            hipLaunchKernelGGL(primary_kernel<kViewHbm>, dim3(pgrid), dim3(kBlock), 0, stream, p.scene, p.ws.tile_draws, out, out8, p, tile_base, batch_tiles);
            launch_levels<kViewHbm>(p, stream, 0);
        }
        const int rgrid = batch_tiles * p.parts_per_tile < resolve_grid ? batch_tiles * p.parts_per_tile : resolve_grid;
        hipLaunchKernelGGL(resolve_kernel, dim3(rgrid), dim3(kBlock), 0, stream, p.scene, p.ws.tile_draws, out, out8, p, tile_base);
        const int batch = r0 / p.rows_per_batch;
        if (marks && batch < marks->n_batch_done) {
            e = hipEventRecord(marks->batch_done[batch], stream);
            if (e != hipSuccess) return e;
        }
    }
    return hipGetLastError();
}

hipError_t launch_unpack_rows(const mcrt_config& cfg, const Shard& sh, const float* packed, float* frame,
                              hipStream_t stream) {
    size_t n = static_cast<size_t>(sh.owned_rows) * cfg.tile_size * cfg.width;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(unpack_rows_kernel<float4>, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, stream, cfg, sh,
                       reinterpret_cast<const float4*>(packed), reinterpret_cast<float4*>(frame));
    return hipGetLastError();
}
hipError_t launch_unpack_rows8(const mcrt_config& cfg, const Shard& sh, const uint8_t* packed, uint8_t* frame, hipStream_t stream) {
    size_t n = static_cast<size_t>(sh.owned_rows) * cfg.tile_size * cfg.width;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(unpack_rows_kernel<uchar4>, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, stream, cfg, sh,
                       reinterpret_cast<const uchar4*>(packed), reinterpret_cast<uchar4*>(frame));
    return hipGetLastError();
}

hipError_t launch_assemble_frame(const mcrt_config& cfg, int world, const float* gathered, size_t rank_stride_pixels,
                                 float* frame, hipStream_t stream) {
    const size_t n = static_cast<size_t>(cfg.width) * cfg.height;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(assemble_frame_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, stream, cfg, world,
                       reinterpret_cast<const float4*>(gathered), rank_stride_pixels, reinterpret_cast<float4*>(frame));
    return hipGetLastError();
}

hipError_t launch_quantize(const float* rgba, uint8_t* out, size_t n_pixels, hipStream_t stream) {
    if (n_pixels == 0) return hipSuccess;
    hipLaunchKernelGGL(quantize_kernel, dim3(static_cast<unsigned>((n_pixels + 255) / 256)), dim3(256), 0, stream,
                       reinterpret_cast<const float4*>(rgba), reinterpret_cast<uchar4*>(out), n_pixels);
    return hipGetLastError();
}

hipError_t launch_build_seed_table(uint32_t* table, hipStream_t stream) {
    hipLaunchKernelGGL(seed_table_kernel, dim3(kSeedWindow / 256u), dim3(256), 0, stream, table);
    return hipGetLastError();
}

hipError_t launch_build_seed_table_range(uint32_t* table, uint32_t first, uint32_t count, hipStream_t stream) {
    if (count == 0u || (count & 255u)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(seed_table_range_kernel, dim3(count / 256u), dim3(256), 0, stream, table, first);
    return hipGetLastError();
}

hipError_t launch_probe_intersect(const uint8_t* scene, const float* rays, int n, mcrt_hit* out, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(probe_intersect_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, scene, rays, n, out);
    return hipGetLastError();
}
hipError_t launch_probe_trace(const uint8_t* scene, const mcrt_config& cfg, const float* rays, int n, int depth,
                              float* out, uint32_t* hit_rng, float* deep_stack, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(probe_trace_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, scene, cfg, rays, n, depth, out,
                       hit_rng, deep_stack);
    return hipGetLastError();
}
hipError_t launch_probe_mt(const uint32_t* seeds, int n_seeds, int n_draws, float* out, uint32_t* storage,
                           hipStream_t stream) {
    if (n_seeds <= 0) return hipSuccess;
    hipLaunchKernelGGL(probe_mt_kernel, dim3((n_seeds + 63) / 64), dim3(64), 0, stream, seeds, n_seeds, n_draws, out,
                       storage);
    return hipGetLastError();
}
hipError_t launch_probe_detmath(int op, const float* x, const float* y, size_t n, float* out, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(probe_detmath_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, stream, op, x,
                       y, n, out);
    return hipGetLastError();
}
hipError_t launch_probe_detmath_range(int op, uint32_t lo_bits, uint64_t count, float y0, float* out,
                                      hipStream_t stream) {
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(probe_detmath_range_kernel, dim3(static_cast<unsigned>((count + 255) / 256)), dim3(256), 0,
                       stream, op, lo_bits, count, y0, out);
    return hipGetLastError();
}

}  // namespace mcrt
