// render_kernels.hip — hand-written HIP kernels for gfx950 (MI355X / CDNA4).
//
// The hot path of the reference, TileRenderer::renderTile (tile_renderer.cpp:71-127), as a
// tile-per-workgroup kernel:
//   * one workgroup (256 threads = 4 wave64) renders one tile; the work items of a tile are its
//     (pixel, sample) pairs in the reference's RNG stream order w = pixel*spp + s, processed in
//     chunks of kChunk items;
//   * the tile's std::mt19937 jitter/lens stream is regenerated in LDS: the seeded 624-word state
//     comes from a small pre-pass (one lane per tile, the seeding recurrence is sequential), the
//     twist is done cooperatively (3 parallel phases over a ping-pong state), tempered draws for
//     the chunk land in an LDS float buffer;
//   * per item: camera/lens ray → closest hit over the scene (flat blob, wave-uniform scalar
//     loads; meshes whose screen bound misses the tile are skipped for primary rays) → miss:
//     gradient background(u,v); hit: iterative traceRay (soft shadow with a register-only
//     truncated mt19937, Blinn-Phong, AO, reflection loop folded back to front);
//   * sample colours go to LDS, then one thread per pixel adds its samples in sample order
//     (float addition order is part of the result) and stores one coalesced float4.
// No MFMA: there is no dense contraction anywhere on this path.
#include "kernels.h"
#include "rt_core.h"

namespace mcrt {

using namespace rt;

constexpr int kBlock = 256;
constexpr int kChunk = 256;        // work items per chunk: one per thread
constexpr int kMaxDrawsPerItem = 4;
constexpr int kDrawCap = 4096;     // LDS floats for the per-hit shadow draws of one sub-batch
constexpr int kAlphaLdsCap = 4096; // alpha-predicate words staged in LDS (64 Ki texels); larger pools stay in HBM

// ---------------------------------------------------------------------------------------------
// tile geometry helpers (TileRenderer::generateTiles, tile_renderer.cpp:18-39)
// ---------------------------------------------------------------------------------------------
struct TileGeom {
    int x, y, w, h;
    int owned_row;  // index of this tile's row among the rows this launch owns
};

__device__ __forceinline__ TileGeom tile_of(const RenderParams& p, int owned_tile) {
    TileGeom t;
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
    uint32_t* dst = p.tile_rng + static_cast<size_t>(tile) * 624;
    dst[0] = x;
    for (uint32_t j = 1; j < 624; ++j) {
        x = mt_step(x, j);
        dst[j] = x;
    }
}

// ---------------------------------------------------------------------------------------------
// cooperative mt19937 block generation in LDS
// ---------------------------------------------------------------------------------------------
struct TileStream {
    uint32_t* st;  // 2 x 624 words ping-pong
    int cur;       // which half holds the current block
    int block;     // index of the block held (-1: seeded, nothing generated yet)
};

// one twist: st[cur] → st[cur^1]; all threads of the workgroup must call
__device__ __forceinline__ void stream_twist(TileStream& ts) {
    const uint32_t* o = ts.st + ts.cur * 624;
    uint32_t* n = ts.st + (ts.cur ^ 1) * 624;
    const int tid = threadIdx.x;
    // phase A: k in [0,227) — old values only
    if (tid < 227) n[tid] = mt_twist(o[tid], o[tid + 1], o[tid + 397]);
    __syncthreads();
    // phase B: k in [227,454) — needs new[k-227]
    if (tid < 227) {
        int k = tid + 227;
        n[k] = mt_twist(o[k], o[k + 1], n[k - 227]);
    }
    __syncthreads();
    // phase C: k in [454,624) — new[k-227]; k = 623 wraps to new[0]
    if (tid < 170) {
        int k = tid + 454;
        uint32_t nxt = (k == 623) ? n[0] : o[k + 1];
        n[k] = mt_twist(o[k], nxt, n[k - 227]);
    }
    __syncthreads();
    ts.cur ^= 1;
    ts.block += 1;
}

// Fill jit[0 .. count) with `count` consecutive stream draws, starting at draw `off` of block
// `blk`, as uniform floats.  Collective.  A work unit that starts in the middle of a tile's stream
// catches up by twisting from the seeded state.
__device__ __forceinline__ void stream_fill(TileStream& ts, float* jit, int blk, int off, int count) {
    int out = 0;
    while (count > 0) {
        while (ts.block < blk) stream_twist(ts);
        const uint32_t* s = ts.st + ts.cur * 624;
        const int m = min(624 - off, count);
        for (int e = threadIdx.x; e < m; e += kBlock) jit[out + e] = mt_to_unit(mt_temper(s[off + e]));
        out += m;
        count -= m;
        off = 0;
        ++blk;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// primary-ray culling mask of a tile
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool mesh_touches_tile(const FlatMesh& m, const TileGeom& t, const mcrt_config& cfg,
                                                  float aspect) {
    const float u0 = m.screen[0], v0 = m.screen[1], u1 = m.screen[2], v1 = m.screen[3];
    if (u0 > u1) return true;  // no bound available
    const float W = static_cast<float>(cfg.width), H = static_cast<float>(cfg.height);
    // tile extent padded by 2 pixels, in the bound's units (x: (2u-1)*aspect, y: 1-2v, +y up)
    float tu0 = (2.0f * (static_cast<float>(t.x) - 2.0f) / W - 1.0f) * aspect - 1e-3f * aspect - 1e-3f;
    float tu1 = (2.0f * (static_cast<float>(t.x + t.w) + 2.0f) / W - 1.0f) * aspect + 1e-3f * aspect + 1e-3f;
    float tv1 = 1.0f - 2.0f * (static_cast<float>(t.y) - 2.0f) / H + 2e-3f;
    float tv0 = 1.0f - 2.0f * (static_cast<float>(t.y + t.h) + 2.0f) / H - 2e-3f;
    return !(u1 < tu0 || u0 > tu1 || v1 < tv0 || v0 > tv1);
}

// ---------------------------------------------------------------------------------------------
// the trace kernel: one workgroup per work unit (a pixel-aligned part of a tile)
//
// Per chunk of 256 work items (one per thread), traceRay is run as workgroup phases:
//   P  primary ray + closest hit (per item)                       → alive lanes = items that hit
//   loop over recursion levels while any lane is alive:
//     B  per alive hit (owner lane): register-only mt19937 → 2·S shadow draws into LDS
//     C  per (hit, shadow sample) pair, one pair per lane: disk sample → any-hit shadow ray,
//        visible samples counted with LDS atomics          (8 lanes per hit at the default S = 8)
//     D  per alive hit (owner lane): Blinn-Phong (+AO), reflection ray, closest hit of the next
//        level; chains that end fold their level colours back to front
//   A  ordered per-pixel accumulation of the chunk's sample colours, coalesced float4 store
// ---------------------------------------------------------------------------------------------
// In-kernel phase stamps: diagnostic builds only (-DMCRT_STAMPS).  Per wave, lane 0 adds the
// s_memtime delta of each phase to a global table that no other code reads.
#ifdef MCRT_STAMPS
__device__ unsigned long long g_phase_cycles[16];
#define STAMP_BEGIN()                        \
    unsigned long long stamp_t0_ = clock64(); \
    unsigned long long stamp_acc_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define STAMP(slot)                          \
    do {                                     \
        unsigned long long t1_ = clock64();  \
        stamp_acc_[slot] += t1_ - stamp_t0_; \
        stamp_t0_ = t1_;                     \
    } while (0)
#define STAMP_FLUSH()                                                                    \
    do {                                                                                 \
        if ((threadIdx.x & 63) == 0)                                                     \
            for (int i_ = 0; i_ < 10; ++i_) atomicAdd(&g_phase_cycles[i_], stamp_acc_[i_]); \
    } while (0)
#else
#define STAMP_BEGIN() do { } while (0)
#define STAMP(slot) do { } while (0)
#define STAMP_FLUSH() do { } while (0)
#endif

#ifndef MCRT_WAVES_PER_EU
#define MCRT_WAVES_PER_EU 2
#endif
// kGeneral = false is the lean common variant: no AO, per-hit RNG streams of at most 227 draws,
// at most 16 bounces.  It performs no store through a generic or global pointer other than the
// frame, so every uniform read of the scene blob stays a scalar load (s_load) — with the rare
// features compiled in, their possible aliasing turns the mesh loop's loads into vector loads.
template <bool kGeneral>
__global__ __launch_bounds__(kBlock, MCRT_WAVES_PER_EU) void render_units_kernel(
    const uint8_t* __restrict__ scene_blob, const uint32_t* __restrict__ tile_rng, float4* __restrict__ out_frame,
    const RenderParams p, const int n_units) {
    __shared__ uint32_t s_mt[2 * 624];
    __shared__ float s_jit[kChunk * kMaxDrawsPerItem];
    __shared__ float4 s_col[kChunk];
    __shared__ float4 s_hp[kChunk];  // alive hit k: point
    __shared__ float4 s_hn[kChunk];  // alive hit k: normal used by its shadow rays
    __shared__ unsigned int s_lit[kChunk];
    __shared__ float4 s_carry[2];
    __shared__ unsigned long long s_mask;
    __shared__ int s_wave_cnt[kBlock / 64];
    extern __shared__ __align__(16) unsigned char s_dyn[];
    // dynamic LDS: [per-hit shadow draws][face table: 4 ints per (mesh, face)][alpha-predicate words]
    float* s_draws = reinterpret_cast<float*>(s_dyn);
    int* s_faces = reinterpret_cast<int*>(s_dyn + static_cast<size_t>(p.lds_draw_floats) * 4);
    uint32_t* s_abits = reinterpret_cast<uint32_t*>(s_dyn + static_cast<size_t>(p.lds_draw_floats) * 4 +
                                                    static_cast<size_t>(p.lds_face_entries) * 16);

    const SceneView scg = view_of(scene_blob);
    const mcrt_config& cfg = p.cfg;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    // Stage what the per-candidate tests read with per-lane indices — the face → texture table and
    // the alpha predicates — in LDS (the host launches this kernel only when both fit).
    for (int i = tid; i < p.lds_alpha_words; i += kBlock) s_abits[i] = scg.abits[i];
    for (int i = tid; i < p.lds_face_entries; i += kBlock) {
        const FlatMesh& fm = scg.meshes[i / 6];
        const int f = i - (i / 6) * 6;
        s_faces[4 * i + 0] = fm.tex_off[f];
        s_faces[4 * i + 1] = fm.tex_w[f];
        s_faces[4 * i + 2] = fm.tex_h[f];
        s_faces[4 * i + 3] = 0;
    }
    const SceneViewLds sc = view_with_lds(scg, (const MCRT_LDS uint32_t*)s_abits, (const MCRT_LDS int*)s_faces);
    const int spp = cfg.samples_per_pixel > 1 ? cfg.samples_per_pixel : 1;
    const int dd = p.draws_per_sample;
    const bool dof = cfg.dof_enabled && cfg.aperture > 1e-6f;
    const float aspect = static_cast<float>(cfg.width) / static_cast<float>(cfg.height);
    const float fW = static_cast<float>(cfg.width), fH = static_cast<float>(cfg.height);
    float focusDist = cfg.focus_distance;
    if (focusDist <= 0.0f) focusDist = sc.hdr->cam_focus_auto;
    const float inv_spp = 1.0f / static_cast<float>(spp);
    const float* fb = sc.hdr->background;
    const C4 flat_bg{fb[0], fb[1], fb[2], fb[3]};
    const V3 lpos = ld3(sc.hdr->light_pos);

    const int mode = shadow_mode(sc, cfg);
    const int S = cfg.shadow_samples;
    const int pairs_per_hit = (mode == SHADOW_SOFT) ? S : 1;
    // the disk samples of a hit are spread over lanes when their draws fit the LDS budget
    const bool spread = (mode != SHADOW_SOFT) || (2 * S <= p.lds_draw_floats);
    const int batch_hits = (mode == SHADOW_SOFT && spread) ? max(1, min(kChunk, p.lds_draw_floats / (2 * S))) : kChunk;

    uint32_t* my_hit_rng = nullptr;
    C4 local_stack[kMaxStack];
    C4* stack = local_stack;
    if constexpr (kGeneral) {
        if (p.hit_rng) my_hit_rng = p.hit_rng + (static_cast<size_t>(blockIdx.x) * kBlock + tid) * 624;
        if (p.deep_stack)
            stack = reinterpret_cast<C4*>(p.deep_stack) +
                    (static_cast<size_t>(blockIdx.x) * kBlock + tid) * static_cast<size_t>(max(cfg.max_bounces, 1));
    }

    for (int unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
        const int tile = unit / p.parts_per_tile;
        const int part = unit - tile * p.parts_per_tile;
        const TileGeom tg = tile_of(p, tile);
        // all index arithmetic is 32-bit: a work item is (pixel, sample), never a flat 64-bit index
        const unsigned npix = static_cast<unsigned>(tg.w) * static_cast<unsigned>(tg.h);
        const unsigned pix_per_part = (npix + p.parts_per_tile - 1) / p.parts_per_tile;
        const unsigned pp0 = static_cast<unsigned>(part) * pix_per_part;
        const unsigned pp1 = min(npix, pp0 + pix_per_part);
        if (pp0 >= pp1) continue;  // uniform: clipped tiles have fewer parts
        // position of the unit's first draw in the tile's mt19937 stream (one 64-bit division per unit)
        const unsigned long long first_draw = static_cast<unsigned long long>(pp0) * spp * dd;
        int sblk = static_cast<int>(first_draw / 624ull);
        int soff = static_cast<int>(first_draw - static_cast<unsigned long long>(sblk) * 624ull);

        // ---- per-unit setup: RNG state into LDS, culling mask ----
        __syncthreads();
        TileStream ts;
        ts.st = s_mt;
        ts.cur = 0;
        ts.block = -1;
        if (dd > 0) {
            const uint32_t* src = tile_rng + static_cast<size_t>(tile) * 624;
            for (int e = tid; e < 624; e += kBlock) s_mt[e] = src[e];
        }
        if (tid < 64) {
            bool touch = true;
            const bool cull = sc.hdr->cull_ok != 0 && !dof;
            if (tid < sc.n_meshes && cull) touch = mesh_touches_tile(sc.meshes[tid], tg, cfg, aspect);
            unsigned long long m = __ballot(touch && tid < sc.n_meshes);
            if (tid == 0) s_mask = (sc.n_meshes < 64 && cull) ? m : ~0ull;
        }
        __syncthreads();
        const unsigned long long mesh_mask = s_mask;
        STAMP(0);  // prologue + unit setup

        // chunk cursor: the chunk starts at sample cs of pixel cp
        unsigned cp = pp0, cs = 0;
        int parity = 0;
        while (cp < pp1) {
            // items left in the unit, clamped to one chunk (the product cannot overflow: it is only
            // formed when fewer than kChunk pixels remain and then compared after a division)
            const unsigned pix_left = pp1 - cp;
            int n = kChunk;
            if (pix_left <= static_cast<unsigned>(kChunk)) {
                const unsigned long long left = static_cast<unsigned long long>(pix_left) * spp - cs;
                if (left < static_cast<unsigned long long>(kChunk)) n = static_cast<int>(left);
            }
            if (dd > 0) {
                stream_fill(ts, s_jit, sblk, soff, n * dd);
                soff += n * dd;
                sblk += soff / 624;
                soff %= 624;
            }
            STAMP(1);  // tile stream fill (incl. catch-up twists)

            // ---- P: primary ray of this thread's item ----
            bool alive = false, query = false;
            Ray ray{mk(0, 0, 0), mk(0, 0, 0)};
            Hit hit;
            hit.hit = false;
            C4 col{0.0f, 0.0f, 0.0f, 0.0f};
            float su = 0.5f, sv = 0.5f;
            int depth = 0, top = 0;
            if (tid < n) {
                const unsigned sidx = cs + tid;  // < spp + kChunk
                const unsigned pix = cp + sidx / static_cast<unsigned>(spp);
                const unsigned uly = pix / static_cast<unsigned>(tg.w);
                const int ly = static_cast<int>(uly);
                const int lx = static_cast<int>(pix - uly * static_cast<unsigned>(tg.w));
                const int px = tg.x + lx, py = tg.y + ly;
                const float* jd = s_jit + tid * dd;
                float jx = 0.5f, jy = 0.5f;
                int dpos = 0;
                if (spp > 1) {
                    jx = jd[0];
                    jy = jd[1];
                    dpos = 2;
                }
                su = (static_cast<float>(px) + jx) / fW;
                sv = (static_cast<float>(py) + jy) / fH;
                ray = dof ? lens_ray(sc, su, sv, aspect, cfg.aperture, focusDist, jd[dpos], jd[dpos + 1])
                          : camera_ray(sc, su, sv, aspect);
                query = true;
            }
            STAMP(2);  // P: primary
            // ---- recursion levels; round 0 resolves the primary rays, round r the depth-r reflections ----
            for (int round = 0;; ++round) {
                // Q: the ONE closest-hit site of the kernel
                if (query) {
                    hit = hit_scene(sc, ray, round == 0 ? mesh_mask : ~0ull);
                    query = false;
                    if (round == 0) {
                        if (!hit.hit) {
                            col = background(sc, cfg, su, sv);  // tile_renderer.cpp:111-114
                        } else if (cfg.max_bounces < 0) {
                            col = background(sc, cfg, 0.5f, 0.5f);  // raytracer.cpp:86-90 (depth 0 > maxBounces)
                        } else {
                            alive = true;
                        }
                    } else if (hit.hit) {
                        alive = true;
                    } else {  // bounced ray missed → flat background (raytracer.cpp:94-102), chain ends
                        C4 tail = flat_bg;
                        while (top > 0) tail = fold_reflection(stack[--top], tail);
                        col = tail;
                    }
                }
                // compact the alive lanes: k = rank of this lane's hit among the chunk's alive hits
                const unsigned long long bal = __ballot(alive);
                if (lane == 0) s_wave_cnt[wave] = __popcll(bal);
                __syncthreads();
                int base = 0, n_alive = 0;
#pragma unroll
                for (int wv = 0; wv < kBlock / 64; ++wv) {
                    const int c = s_wave_cnt[wv];
                    if (wv < wave) base += c;
                    n_alive += c;
                }
                STAMP(3);  // Q + compaction + its barrier
                if (n_alive == 0) break;  // uniform
                const int k = base + __popcll(bal & ((1ull << lane) - 1ull));

                float vis = 1.0f;
                if (spread) {
                    if (alive) {
                        const V3 sn = (mode == SHADOW_HARD) ? normalize(hit.n) : hit.n;
                        s_hp[k] = make_float4(hit.p.x, hit.p.y, hit.p.z, 0.0f);
                        s_hn[k] = make_float4(sn.x, sn.y, sn.z, 0.0f);
                        s_lit[k] = 0u;
                    }
                    for (int b0 = 0; b0 < n_alive; b0 += batch_hits) {
                        const int nb = min(batch_hits, n_alive - b0);
                        // ---- B: shadow draws of the hits in this sub-batch ----
                        if (mode == SHADOW_SOFT && alive && k >= b0 && k < b0 + nb) {
                            float* dst = s_draws + (k - b0) * 2 * S;
                            if constexpr (kGeneral) {
                                HitRng rng;
                                rng.seed(shadow_seed(hit.p, depth), 2 * S, my_hit_rng);
                                for (int i = 0; i < 2 * S; ++i) dst[i] = rng.uniform();
                            } else {
#ifdef MCRT_ABL_NO_B  // timing ablation only: constant draws instead of the mt19937 chain
                                for (int i = 0; i < 2 * S; ++i) dst[i] = 0.25f + 0.03f * i;
#else
                                MtShort rng;
                                rng.seed(shadow_seed(hit.p, depth));
                                for (int i = 0; i < 2 * S; ++i) dst[i] = rng.uniform();
#endif
                            }
                        }
                        STAMP(4);  // B: mt19937 draws
                        __syncthreads();
                        STAMP(5);  // barrier after B
                        // ---- C: one (hit, light sample) pair per lane; the ONE any-hit site ----
#ifdef MCRT_ABL_NO_C  // timing ablation only: no shadow rays
                        for (int q = tid; q < nb * pairs_per_hit; q += kBlock) atomicAdd(&s_lit[b0 + q / pairs_per_hit], 1u);
                        if (false)
#endif
                        for (int q = tid; q < nb * pairs_per_hit; q += kBlock) {
                            const int kk = q / pairs_per_hit;
                            const int j = q - kk * pairs_per_hit;
                            const float4 hp = s_hp[b0 + kk], hn = s_hn[b0 + kk];
                            const V3 P = mk(hp.x, hp.y, hp.z), N = mk(hn.x, hn.y, hn.z);
                            V3 target = lpos;
                            if (mode == SHADOW_SOFT) {
                                const float* dr = s_draws + kk * 2 * S + 2 * j;
                                target = light_sample_position(sc, P, dr[0], dr[1]);
                            }
                            if (!in_shadow_inline(sc, P, N, target)) atomicAdd(&s_lit[b0 + kk], 1u);
                        }
                        STAMP(6);  // C: shadow rays
                        __syncthreads();
                        STAMP(7);  // barrier after C
                    }
                    if (alive) {
                        const unsigned int lit = s_lit[k];
                        vis = (mode == SHADOW_SOFT) ? static_cast<float>(lit) / static_cast<float>(S)
                                                    : (lit ? 1.0f : 0.0f);
                    }
                } else if (alive) {
                    if constexpr (kGeneral) vis = hit_visibility(sc, cfg, hit, depth, my_hit_rng);  // very large S: sequential
                }

                // ---- D: colour of the level; the reflection ray goes to the next round's Q ----
                if (alive) {
                    C4 c = kGeneral ? level_color(sc, cfg, ray.o, hit, depth, vis, my_hit_rng)
                                    : shade(sc, hit, normalize(ray.o - hit.p), vis);
                    alive = false;
                    if (depth >= cfg.max_bounces) {
                        C4 tail = clamp4(c);
                        while (top > 0) tail = fold_reflection(stack[--top], tail);
                        col = tail;
                    } else {
                        stack[top++] = c;
                        ray = reflect_ray(ray, hit);
                        ++depth;
                        query = true;
                    }
                }
                STAMP(8);  // D: shade + reflection ray
            }

            if (tid < n) s_col[tid] = make_float4(col.r, col.g, col.b, col.a);
            __syncthreads();

            // ---- A: ordered per-pixel accumulation (tile_renderer.cpp:116-124) ----
            // pixel cp+i owns chunk items [i*spp - cs, (i+1)*spp - cs) clipped to [0, n)
            const int n_pix = static_cast<int>((cs + n - 1) / static_cast<unsigned>(spp)) + 1;
            for (int i = tid; i < n_pix; i += kBlock) {
                const long long wb = static_cast<long long>(i) * spp - cs, we = wb + spp;
                const int sb = wb > 0 ? static_cast<int>(wb) : 0;
                const int se = we < n ? static_cast<int>(we) : n;
                float4 acc = (wb >= 0) ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : s_carry[parity ^ 1];
                for (int q = sb; q < se; ++q) {
                    float4 sc4 = s_col[q];
                    acc.x += sc4.x;
                    acc.y += sc4.y;
                    acc.z += sc4.z;
                    acc.w += sc4.w;
                }
                if (we <= n) {
                    const unsigned pix = cp + static_cast<unsigned>(i);
                    const unsigned uly = pix / static_cast<unsigned>(tg.w);
                    const int ly = static_cast<int>(uly);
                    const int lx = static_cast<int>(pix - uly * static_cast<unsigned>(tg.w));
                    const int row = (p.layout == MCRT_LAYOUT_PACKED) ? (tg.owned_row * cfg.tile_size + ly) : (tg.y + ly);
                    float4* dst = out_frame + static_cast<size_t>(row) * cfg.width + (tg.x + lx);
                    *dst = make_float4(acc.x * inv_spp, acc.y * inv_spp, acc.z * inv_spp, acc.w * inv_spp);
                } else {
                    s_carry[parity] = acc;
                }
            }
            __syncthreads();
            STAMP(9);  // A: accumulate + store
            // advance the cursor
            const unsigned adv = cs + static_cast<unsigned>(n);
            cp += adv / static_cast<unsigned>(spp);
            cs = adv % static_cast<unsigned>(spp);
            parity ^= 1;
        }
    }
    STAMP_FLUSH();
}

// ---------------------------------------------------------------------------------------------
// small kernels
// ---------------------------------------------------------------------------------------------
__global__ void unpack_rows_kernel(mcrt_config cfg, Shard sh, const float4* packed, float4* frame) {
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

__global__ void quantize_kernel(const float4* rgba, uchar4* out, size_t n) {  // image_writer.cpp:18-22
    size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 c = rgba[i];
    uchar4 q;
    q.x = static_cast<unsigned char>(sclamp(c.x, 0.0f, 1.0f) * 255.0f + 0.5f);
    q.y = static_cast<unsigned char>(sclamp(c.y, 0.0f, 1.0f) * 255.0f + 0.5f);
    q.z = static_cast<unsigned char>(sclamp(c.z, 0.0f, 1.0f) * 255.0f + 0.5f);
    q.w = static_cast<unsigned char>(sclamp(c.w, 0.0f, 1.0f) * 255.0f + 0.5f);
    out[i] = q;
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

// ---------------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------------
hipError_t read_phase_stamps(unsigned long long out[16], bool reset) {
#ifdef MCRT_STAMPS
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase_cycles), 16 * sizeof(unsigned long long));
    if (e == hipSuccess && reset) {
        unsigned long long z[16] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), z, sizeof z);
    }
    return e;
#else
    for (int i = 0; i < 16; ++i) out[i] = 0;
    (void)reset;
    return hipErrorNotSupported;
#endif
}

Shard make_shard(const mcrt_config& cfg, int first, int step) {
    Shard s{};
    s.first = first;
    s.step = step < 1 ? 1 : step;
    if (cfg.width <= 0 || cfg.height <= 0 || cfg.tile_size <= 0) return s;
    s.tiles_x = (cfg.width + cfg.tile_size - 1) / cfg.tile_size;
    s.tiles_y = (cfg.height + cfg.tile_size - 1) / cfg.tile_size;
    s.owned_rows = (first < s.tiles_y && first >= 0) ? (s.tiles_y - first + s.step - 1) / s.step : 0;
    return s;
}

static int owned_tiles(const RenderParams& p) { return p.shard.owned_rows * p.shard.tiles_x; }

// rare features that need the general kernel variant
static bool needs_general_variant(const RenderParams& p) {
    const bool soft = p.cfg.soft_shadows && p.cfg.shadow_samples > 1;
    return p.cfg.ao_enabled || (soft && 2 * p.cfg.shadow_samples > kMtShortMax) || p.cfg.max_bounces > kMaxStack ||
           (soft && 2 * p.cfg.shadow_samples > p.lds_draw_floats) || p.scene_in_lds == 0;
}

// Work units per tile: enough units to keep ~8 workgroups per CU in the queue even when only a
// few tiles carry the character, but never finer than one 256-item chunk.  Units of one tile
// share its RNG stream; a later unit catches up by twisting from the seeded state, which costs
// (tile draws / 624) twists at worst — small next to a chunk that contains hits.
int choose_parts_per_tile(const mcrt_config& cfg, int n_tiles, int target_units) {
    if (n_tiles <= 0) return 1;
    const long long spp = cfg.samples_per_pixel > 1 ? cfg.samples_per_pixel : 1;
    const long long tile_items = static_cast<long long>(cfg.tile_size) * cfg.tile_size * spp;
    const long long chunks = (tile_items + kChunk - 1) / kChunk;
    long long parts = (target_units + n_tiles - 1) / n_tiles;
    if (parts > chunks) parts = chunks;
    if (parts > 4096) parts = 4096;
    if (parts < 1) parts = 1;
    return static_cast<int>(parts);
}

void fill_launch_geometry(RenderParams& p, int target_units) {
    const int n = owned_tiles(p);
    p.parts_per_tile = choose_parts_per_tile(p.cfg, n, target_units);
    const bool soft = p.cfg.soft_shadows && p.cfg.shadow_samples > 1;
    const long long want = soft ? static_cast<long long>(kChunk) * 2 * p.cfg.shadow_samples : 0;
    p.lds_draw_floats = static_cast<int>(want < kDrawCap ? want : kDrawCap);
    p.grid_blocks = render_grid_blocks(p);
}

int render_grid_blocks(const RenderParams& p) {
    int n = owned_tiles(p) * (p.parts_per_tile > 0 ? p.parts_per_tile : 1);
    // long per-hit RNG streams / very deep recursion need per-thread HBM slices: bound the grid
    if (2 * p.cfg.shadow_samples > kMtShortMax || (p.cfg.ao_enabled && 2 * p.cfg.ao_samples > kMtShortMax) ||
        p.cfg.max_bounces > kMaxStack)
        return n < 1024 ? n : 1024;
    return n;
}
size_t tile_rng_bytes(const RenderParams& p) {
    return p.draws_per_sample > 0 ? static_cast<size_t>(owned_tiles(p)) * 624 * 4 : 0;
}
size_t hit_rng_bytes(const RenderParams& p) {
    bool need = (p.cfg.soft_shadows && 2 * p.cfg.shadow_samples > kMtShortMax) ||
                (p.cfg.ao_enabled && 2 * p.cfg.ao_samples > kMtShortMax);
    return need ? static_cast<size_t>(render_grid_blocks(p)) * kBlock * 624 * 4 : 0;
}
size_t deep_stack_bytes(const RenderParams& p) {
    return p.cfg.max_bounces > kMaxStack
               ? static_cast<size_t>(render_grid_blocks(p)) * kBlock * p.cfg.max_bounces * 16
               : 0;
}

hipError_t launch_render(const RenderParams& p, hipStream_t stream, hipEvent_t ev_k0, hipEvent_t ev_k1) {
    int n = owned_tiles(p);
    if (n <= 0) return hipSuccess;
    if (p.draws_per_sample > 0) {
        hipLaunchKernelGGL(seed_tiles_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, p, n);
    }
    if (ev_k0) (void)hipEventRecord(ev_k0, stream);
    const size_t dyn = static_cast<size_t>(p.lds_draw_floats) * 4 + static_cast<size_t>(p.lds_face_entries) * 16 +
                       static_cast<size_t>(p.lds_alpha_words) * 4;
    if (needs_general_variant(p))
        hipLaunchKernelGGL(render_units_kernel<true>, dim3(p.grid_blocks), dim3(kBlock), dyn, stream, p.scene, p.tile_rng,
                           reinterpret_cast<float4*>(p.out), p, n * p.parts_per_tile);
    else
        hipLaunchKernelGGL(render_units_kernel<false>, dim3(p.grid_blocks), dim3(kBlock), dyn, stream, p.scene, p.tile_rng,
                           reinterpret_cast<float4*>(p.out), p, n * p.parts_per_tile);
    if (ev_k1) (void)hipEventRecord(ev_k1, stream);
    return hipGetLastError();
}

hipError_t launch_unpack_rows(const mcrt_config& cfg, const Shard& sh, const float* packed, float* frame,
                              hipStream_t stream) {
    size_t n = static_cast<size_t>(sh.owned_rows) * cfg.tile_size * cfg.width;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(unpack_rows_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, stream, cfg, sh,
                       reinterpret_cast<const float4*>(packed), reinterpret_cast<float4*>(frame));
    return hipGetLastError();
}

hipError_t launch_quantize(const float* rgba, uint8_t* out, size_t n_pixels, hipStream_t stream) {
    if (n_pixels == 0) return hipSuccess;
    hipLaunchKernelGGL(quantize_kernel, dim3(static_cast<unsigned>((n_pixels + 255) / 256)), dim3(256), 0, stream,
                       reinterpret_cast<const float4*>(rgba), reinterpret_cast<uchar4*>(out), n_pixels);
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
