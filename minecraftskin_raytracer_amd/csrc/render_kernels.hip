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
constexpr int kChunk = 1024;       // work items per chunk (4 per thread)
constexpr int kMaxDrawsPerItem = 4;

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

// Fill jit[0 .. d1-d0) with stream draws d0 .. d1-1 as uniform floats.  Collective.
__device__ __forceinline__ void stream_fill(TileStream& ts, float* jit, long long d0, long long d1) {
    if (d1 <= d0) return;
    int b0 = static_cast<int>(d0 / 624), b1 = static_cast<int>((d1 - 1) / 624);
    for (int b = b0; b <= b1; ++b) {
        while (ts.block < b) stream_twist(ts);
        const uint32_t* s = ts.st + ts.cur * 624;
        for (int e = threadIdx.x; e < 624; e += kBlock) {
            long long d = static_cast<long long>(b) * 624 + e;
            if (d >= d0 && d < d1) jit[d - d0] = mt_to_unit(mt_temper(s[e]));
        }
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
// the trace kernel
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void render_tiles_kernel(RenderParams p, int n_tiles) {
    __shared__ uint32_t s_mt[2 * 624];
    __shared__ float s_jit[kChunk * kMaxDrawsPerItem];
    __shared__ float4 s_col[kChunk];
    __shared__ float4 s_carry[2];
    __shared__ unsigned long long s_mask;

    const SceneView sc = view_of(p.scene);
    const mcrt_config& cfg = p.cfg;
    const int tid = threadIdx.x;
    const int spp = cfg.samples_per_pixel > 1 ? cfg.samples_per_pixel : 1;
    const int dd = p.draws_per_sample;
    const bool dof = cfg.dof_enabled && cfg.aperture > 1e-6f;
    const float aspect = static_cast<float>(cfg.width) / static_cast<float>(cfg.height);
    const float fW = static_cast<float>(cfg.width), fH = static_cast<float>(cfg.height);
    float focusDist = cfg.focus_distance;
    if (focusDist <= 0.0f) focusDist = sc.hdr->cam_focus_auto;
    const float inv_spp = 1.0f / static_cast<float>(spp);

    uint32_t* my_hit_rng =
        p.hit_rng ? p.hit_rng + (static_cast<size_t>(blockIdx.x) * kBlock + tid) * 624 : nullptr;
    C4 local_stack[kMaxStack];
    C4* stack = p.deep_stack
                    ? reinterpret_cast<C4*>(p.deep_stack) +
                          (static_cast<size_t>(blockIdx.x) * kBlock + tid) * static_cast<size_t>(max(cfg.max_bounces, 1))
                    : local_stack;

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const TileGeom tg = tile_of(p, tile);
        const long long npix = static_cast<long long>(tg.w) * tg.h;
        const long long total = npix * spp;

        // ---- per-tile setup: RNG state into LDS, culling mask ----
        TileStream ts;
        ts.st = s_mt;
        ts.cur = 0;
        ts.block = -1;
        if (dd > 0) {
            const uint32_t* src = p.tile_rng + static_cast<size_t>(tile) * 624;
            for (int e = tid; e < 624; e += kBlock) s_mt[e] = src[e];
        }
        if (tid < 64) {
            bool touch = true;
            const bool cull = sc.hdr->cull_ok != 0 && !dof;
            if (tid < sc.n_meshes && cull) touch = mesh_touches_tile(sc.meshes[tid], tg, cfg, aspect);
            unsigned long long m = __ballot(touch && tid < sc.n_meshes);
            if (tid == 0) s_mask = (sc.n_meshes < 64) ? m : ~0ull;
            if (tid == 0 && !cull) s_mask = ~0ull;
        }
        __syncthreads();
        const unsigned long long mesh_mask = s_mask;

        for (long long c0 = 0; c0 < total; c0 += kChunk) {
            const int n = static_cast<int>(min(static_cast<long long>(kChunk), total - c0));
            if (dd > 0) stream_fill(ts, s_jit, c0 * dd, (c0 + n) * dd);

            // ---- trace the chunk's work items ----
            for (int item = tid; item < n; item += kBlock) {
                const long long w = c0 + item;
                const long long pix = w / spp;
                const int ly = static_cast<int>(pix / tg.w);
                const int lx = static_cast<int>(pix - static_cast<long long>(ly) * tg.w);
                const int px = tg.x + lx, py = tg.y + ly;
                const float* jd = s_jit + item * dd;
                float jx = 0.5f, jy = 0.5f;
                int dpos = 0;
                if (spp > 1) {
                    jx = jd[0];
                    jy = jd[1];
                    dpos = 2;
                }
                const float u = (static_cast<float>(px) + jx) / fW;
                const float v = (static_cast<float>(py) + jy) / fH;
                Ray ray = dof ? lens_ray(sc, u, v, aspect, cfg.aperture, focusDist, jd[dpos], jd[dpos + 1])
                              : camera_ray(sc, u, v, aspect);
                Hit hit = hit_scene(sc, ray, mesh_mask);
                C4 c;
                if (!hit.hit) {
                    c = background(sc, cfg, u, v);  // tile_renderer.cpp:111-114
                } else if (cfg.max_bounces < 0) {
                    c = background(sc, cfg, 0.5f, 0.5f);  // raytracer.cpp:86-90 (depth 0 > maxBounces)
                } else {
                    c = trace_from_hit(sc, cfg, ray, hit, 0, stack, my_hit_rng);
                }
                s_col[item] = make_float4(c.r, c.g, c.b, c.a);
            }
            __syncthreads();

            // ---- ordered per-pixel accumulation (tile_renderer.cpp:116-124) ----
            const long long p_first = c0 / spp;
            const long long p_last = (c0 + n - 1) / spp;
            const int parity = static_cast<int>((c0 / kChunk) & 1);
            for (long long pix = p_first + tid; pix <= p_last; pix += kBlock) {
                const long long wb = pix * spp, we = wb + spp;
                const long long sb = wb > c0 ? wb : c0;
                const long long se = we < c0 + n ? we : c0 + n;
                float4 acc = (sb == wb) ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : s_carry[parity ^ 1];
                for (long long q = sb; q < se; ++q) {
                    float4 s = s_col[q - c0];
                    acc.x += s.x;
                    acc.y += s.y;
                    acc.z += s.z;
                    acc.w += s.w;
                }
                if (se == we) {
                    const int ly = static_cast<int>(pix / tg.w);
                    const int lx = static_cast<int>(pix - static_cast<long long>(ly) * tg.w);
                    const int row = (p.layout == MCRT_LAYOUT_PACKED) ? (tg.owned_row * cfg.tile_size + ly) : (tg.y + ly);
                    float4* dst = reinterpret_cast<float4*>(p.out) + static_cast<size_t>(row) * cfg.width + (tg.x + lx);
                    *dst = make_float4(acc.x * inv_spp, acc.y * inv_spp, acc.z * inv_spp, acc.w * inv_spp);
                } else {
                    s_carry[parity] = acc;
                }
            }
            __syncthreads();
        }
    }
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

int render_grid_blocks(const RenderParams& p) {
    int n = owned_tiles(p);
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
    hipLaunchKernelGGL(render_tiles_kernel, dim3(p.grid_blocks), dim3(kBlock), 0, stream, p, n);
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
