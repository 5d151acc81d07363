// api.cpp — C-ABI implementation of include/mcrt.h (host side; kernels in render_kernels.hip).
//
// There is deliberately no CPU fallback anywhere in this file: every render / probe entry point
// needs a HIP device and fails with MCRT_ERR_NO_DEVICE / MCRT_ERR_HIP otherwise.
#include "flatten.h"
#include "kernels.h"
#include "mcrt.h"
#include "mcrt_detmath.h"

#include <hip/hip_runtime.h>


#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <utility>
#include <string>
#include <thread>
#include <vector>

using namespace mcrt;

namespace {

thread_local std::string g_err;
thread_local mcrt_timings g_timings = {0, 0, 0, 0, 0};

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
}  // namespace
// error hook for the library's other translation units (png_writer.cpp)
__attribute__((visibility("hidden"))) int mcrt_detail_fail(int code, const char* msg) { return fail(code, msg ? msg : ""); }
namespace {
int hip_fail(hipError_t e, const char* what) {
    return fail(e == hipErrorNoDevice || e == hipErrorInvalidDevice ? MCRT_ERR_NO_DEVICE : MCRT_ERR_HIP,
                std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(call)                                    \
    do {                                                 \
        hipError_t e_ = (call);                          \
        if (e_ != hipSuccess) return hip_fail(e_, #call); \
    } while (0)

double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

struct DeviceBuffer {
    void* ptr = nullptr;
    size_t bytes = 0;
    hipError_t reserve(size_t need) {
        if (need <= bytes) return hipSuccess;
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        bytes = 0;
        hipError_t e = hipMalloc(&ptr, need);
        if (e == hipSuccess) bytes = need;
        return e;
    }
    void release() {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        bytes = 0;
    }
    DeviceBuffer() = default;
    DeviceBuffer(const DeviceBuffer&) = delete;
    DeviceBuffer& operator=(const DeviceBuffer&) = delete;
    ~DeviceBuffer() { release(); }
};

bool valid_frame(const mcrt_config* c) { return c->width > 0 && c->height > 0 && c->tile_size > 0; }

int draws_per_sample(const mcrt_config& c) {
    int spp = c.samples_per_pixel > 1 ? c.samples_per_pixel : 1;
    return (spp > 1 ? 2 : 0) + ((c.dof_enabled && c.aperture > 1e-6f) ? 2 : 0);
}

}  // namespace

// A lane renders every n-th tile row of a shard with its own workspace on its own stream.  The
// pipeline of one lane is a chain of dependent kernels whose tails and sparse deeper levels leave
// most of the chip idle; two or three lanes in flight fill those gaps (measured: 1080p 0.53 -> 0.43
// ms, 4K/8 bounces/16 spp 9.6 -> 5.3 ms with three lanes).  Lane 0 runs on the caller's stream,
// the others fork from it and join it through events, so the caller sees ordinary stream order.
constexpr int kMaxLanes = 4;
// what the seeded per-tile mt19937 states in Lane::tile_rng are a function of (tile_renderer.cpp:78: the
// seed is tile.y * width + tile.x) — scene and every other setting do not enter
struct RngKey {
    const void* ptr = nullptr;
    int width = 0, tile_size = 0, first = 0, step = 0, tiles_x = 0, owned_rows = 0;
    int rect[4] = {0, 0, 0, 0};
    int parts = 0, part_twists = 0;  // the engine states at the starts of the streams' parts depend on these too
    bool operator==(const RngKey& o) const {
        return parts == o.parts && part_twists == o.part_twists && ptr == o.ptr && width == o.width && tile_size == o.tile_size && first == o.first && step == o.step && tiles_x == o.tiles_x &&
               owned_rows == o.owned_rows && rect[0] == o.rect[0] && rect[1] == o.rect[1] && rect[2] == o.rect[2] && rect[3] == o.rect[3];
    }
};
struct Lane {
    hipStream_t stream = nullptr;  // owned; unused for lane 0
    hipEvent_t done = nullptr;
    // wavefront workspace, grown on demand (never shrinks; no allocation in the steady state)
    DeviceBuffer tile_rng, tile_draws, scol, end, units, unit_hits, tile_mask, queues[5], texel_refs, targets, cand, lit[2], stack, counters, hit_rng;
    RngKey rng_key;               // which tile seeds tile_rng holds (ptr == nullptr: none)
    bool counters_dirty = false;  // a render's launches failed half way: counters and their base no longer fit (cleared before the next render)
};

struct mcrt_scene {
    int device = 0;
    uint32_t alpha_words = 0;
    uint32_t n_meshes = 0;
    bool posed = false;  // any mesh with MESH_ROTATED
    std::vector<uint8_t> host_meshes;  // host copy of FlatHeader + FlatMesh[] (screen bounds for workspace planning)
    DeviceBuffer blob;
    Lane lanes[kMaxLanes];
    int forced_lanes = 0;  // mcrt_scene_set_lanes: 0 = automatic
    size_t budget = 0;     // current workspace budget (0 = workspace_budget()); halved when the device is short of memory
    // recorded launch sequences of recent renders (hipGraph), replayed when the parameters repeat
    struct Recorded {
        int n_lanes = 0;
        RenderParams p[kMaxLanes];
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        unsigned long long last_use = 0;
        int sightings = 0;
    };
    static constexpr int kRecorded = 4;
    Recorded recorded[kRecorded];
    unsigned long long use_clock = 0;
    hipStream_t capture_stream = nullptr;
    hipEvent_t fork = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    // One handle = one frame in flight: all renders of a handle share its workspace.  `last_done` is recorded
    // at the end of every render; a render enqueued on a different stream than the previous one waits for it.
    hipEvent_t last_done = nullptr;
    std::atomic<hipEvent_t> busy_probe{nullptr};  // = last_done once it exists: what OTHER handles' renders query (device_shared)
    hipStream_t last_stream = nullptr;
    bool have_last = false;
    bool flags_checked = true;  // no render since mcrt_scene_check last read (and cleared) the lanes' overflow words
    // the one-shot host path (mcrt_render & co): frame buffer, streams and events kept with the pooled workspace
    DeviceBuffer frame;                // float4 frame / packed rows / RGBA8 plane of a host-buffer render
    hipStream_t main_stream = nullptr;  // the render
    hipStream_t copy_stream = nullptr;  // downloads of finished tile rows, overlapping the render
    std::vector<hipEvent_t> marks;      // event pool of the row-group downloads
    size_t marks_used = 0;
    // pinned host staging for the small transfers of every call (the scene blob up, the lanes' flag words
    // back): no pin / unpin of a few KB of pageable memory per call
    void* staging = nullptr;
    size_t staging_bytes = 0;
    const uint32_t* seed_table = nullptr;  // the device's table of mt19937 seeding results (kernels.h), or NULL
    bool holds_seed_table = false;
    const uint32_t* seed_table_full = nullptr;  // the device's table for every 32-bit seed (ambient occlusion), or NULL
    bool holds_full_table = false, full_table_tried = false;
};

namespace {

// workspace budget of a render (bytes, all lanes together); MCRT_WORKSPACE_MB overrides (tests use a
// small value to force multi-batch renders).  By default a third of the scene's device's memory (96 GB of the
// MI355X's 288): the workspace is sized for the worst case of every
// sample hitting (~300 B per sample), buffers only ever grow to what a frame needs, and a frame cut into few large
// batches is much faster than many small ones (4K / 8 bounces / 16 spp: 9.7 ms with 4 GiB, 6.1 ms in one batch).
// total memory of a device, asked once per device and process (hipMemGetInfo costs ~10 ms per call on this runtime: asked per
// scene it made every one-shot render 12 ms longer)
size_t device_total_memory(int device) {
    static std::mutex mu;
    static std::vector<size_t> totals;
    std::lock_guard<std::mutex> lock(mu);
    if (device < 0) return 0;
    if (totals.size() <= static_cast<size_t>(device)) totals.resize(static_cast<size_t>(device) + 1, 0);
    size_t& t = totals[static_cast<size_t>(device)];
    if (t == 0) {
        size_t total_b = 0;
        if (hipDeviceTotalMem(&total_b, device) != hipSuccess) {
            (void)hipGetLastError();
            total_b = static_cast<size_t>(24) << 30;
        }
        t = total_b ? total_b : 1;
    }
    return t;
}
size_t workspace_budget(int device) {  // read per scene: tests switch MCRT_WORKSPACE_MB between renders
    const char* e = std::getenv("MCRT_WORKSPACE_MB");
    const long long mb = e ? std::atoll(e) : 0;
    if (mb > 0) return static_cast<size_t>(mb) << 20;
    // (when the device cannot give that much right now — other allocations, a shared GPU — prepare() halves the budget
    // and re-plans instead of failing)
    return std::max<size_t>(static_cast<size_t>(256) << 20, device_total_memory(device) / 3);
}

// lanes for a shard: enough work per lane that the extra launches pay (MCRT_LANES forces a count)
int lane_count(const mcrt_scene* s, const mcrt_config& cfg, const Shard& sh) {
    static const int forced = [] {
        const char* e = std::getenv("MCRT_LANES");
        return e ? std::atoi(e) : 0;
    }();
    int lanes;
    if (s->forced_lanes > 0) {
        lanes = s->forced_lanes;
    } else if (forced > 0) {
        lanes = forced;
    } else {
        const int spp = cfg.samples_per_pixel > 1 ? cfg.samples_per_pixel : 1;
        const double samples = static_cast<double>(sh.owned_rows) * cfg.tile_size * cfg.width * spp;
        // One lane up to 2.8e7 samples, three above.  With the launch shapes of a frame that is alone on ONE stream (four
        // waves per tile stream, `lit` at 4 096 workgroups: choose_grids) a single lane beats two or three for every frame
        // up to 2560x1440 / 6 spp (1080p / 4 spp alone: 0.189 / 0.224 / 0.215 ms with 1 / 2 / 3 lanes; 1440p / 6 spp: 0.416 /
        // 0.439 / 0.423); 3840x2160 / 4 spp, 3.3e7 samples: 0.515 / 0.511 / 0.496, and the gap widens from there (GUI defaults,
        // 1.3e8: 5.07 / 4.30 / 4.13).  Two lanes never came out first.
        lanes = samples >= 2.8e7 ? 3 : 1;
    }
    lanes = std::min(lanes, kMaxLanes);
    return std::max(1, std::min(lanes, sh.owned_rows));
}

// Upper bound, per owned tile row of `sh`, of the tiles that plan_tiles can find touched by a mesh.
// It repeats the device's test (mesh_touches_tile + the thin-lens dilation) in double precision
// with several pixels of extra margin, so it can only over-count; anything unusual → every tile.
void touched_tiles_per_row(const mcrt_scene* sc, const mcrt_config& cfg, const Shard& sh, std::vector<int>& out) {
    out.assign(static_cast<size_t>(sh.owned_rows > 0 ? sh.owned_rows : 0), sh.tiles_x);
    if (sh.owned_rows <= 0 || sc->host_meshes.size() < sizeof(FlatHeader)) return;
    const FlatHeader* h = reinterpret_cast<const FlatHeader*>(sc->host_meshes.data());
    const FlatMesh* fm = reinterpret_cast<const FlatMesh*>(sc->host_meshes.data() + h->mesh_offset);
    const int n = static_cast<int>(h->n_meshes);
    if (n == 0) {
        std::fill(out.begin(), out.end(), 0);
        return;
    }
    if (!h->cull_ok || n >= 64) return;
    const double W = cfg.width, H = cfg.height, T = cfg.tile_size;
    const double aspect = static_cast<double>(static_cast<float>(cfg.width) / static_cast<float>(cfg.height));
    const bool dof = cfg.dof_enabled && cfg.aperture > 1e-6f;
    const double half_h = h->cam_half_h, half_w = half_h * aspect;
    std::vector<uint8_t> grid(static_cast<size_t>(sh.tiles_x) * sh.tiles_y, 0);
    for (int i = 0; i < n; ++i) {
        const FlatMesh& m = fm[i];
        double u0 = m.screen[0], v0 = m.screen[1], u1 = m.screen[2], v1 = m.screen[3];
        if (!(u0 <= u1) || !std::isfinite(u0 + u1 + v0 + v1)) return;  // no bound: touches every tile
        if (dof) {
            const double focus = cfg.focus_distance > 0.0f ? cfg.focus_distance : h->cam_focus_auto;
            if (!(m.depth[0] > 0.0f) || !(focus > 0.0)) return;
            const double f_lo = 1.0 / focus, f_hi = std::sqrt(1.0 + half_w * half_w + half_h * half_h) / focus;
            const double z_hi = 1.0 / m.depth[0], z_lo = 1.0 / m.depth[1];
            const double d = std::max(std::max(std::fabs(z_hi - f_lo), std::fabs(z_hi - f_hi)),
                                      std::max(std::fabs(z_lo - f_lo), std::fabs(z_lo - f_hi)));
            const double pad = (cfg.aperture * d / half_h * 1.02 + 1e-3) * 1.01 + 1e-4;
            if (!(pad < 1e6)) return;
            u0 -= pad, v0 -= pad, u1 += pad, v1 += pad;
        }
        // bound units → pixels: u = (2x/W - 1) * aspect, v = 1 - 2y/H; the device pads tiles by 2 px + 1e-3 (u) / 2e-3 (v)
        const double pad_x = 6.0 + (1e-3 * aspect + 1e-3) * W / (2.0 * aspect) + 1e-3 * W;
        const double pad_y = 6.0 + 2e-3 * H / 2.0 + 1e-3 * H;
        const double xa = (u0 / aspect + 1.0) * W * 0.5 - pad_x, xb = (u1 / aspect + 1.0) * W * 0.5 + pad_x;
        const double ya = (1.0 - v1) * H * 0.5 - pad_y, yb = (1.0 - v0) * H * 0.5 + pad_y;
        if (!std::isfinite(xa + xb + ya + yb)) return;
        if (xb < 0.0 || yb < 0.0 || xa >= W || ya >= H) continue;
        const int tx0 = static_cast<int>(std::max(0.0, std::floor(xa / T))), tx1 = static_cast<int>(std::min<double>(sh.tiles_x - 1, std::floor(xb / T)));
        const int ty0 = static_cast<int>(std::max(0.0, std::floor(ya / T))), ty1 = static_cast<int>(std::min<double>(sh.tiles_y - 1, std::floor(yb / T)));
        for (int ty = ty0; ty <= ty1; ++ty)
            for (int tx = tx0; tx <= tx1; ++tx) grid[static_cast<size_t>(ty) * sh.tiles_x + tx] = 1;
    }
    for (int j = 0; j < sh.owned_rows; ++j) {
        const int ty = sh.first + j * sh.step;
        int c = 0;
        for (int tx = 0; tx < sh.tiles_x; ++tx) c += grid[static_cast<size_t>(ty) * sh.tiles_x + tx];
        out[static_cast<size_t>(j)] = c;
    }
}

const uint32_t* acquire_full_seed_table(int device);  // below, with the per-device tables

// fill RenderParams for lane `li` of `n_lanes` over the shard (first, step) + make sure its
// workspace exists (allocation only when it has to grow)
int prepare(mcrt_scene* sc, int li, int n_lanes, const mcrt_config* cfg, int first, int step, int layout, float* d_out,
            uint8_t* d_out8, RenderParams& p, std::vector<int>* row_touched_out = nullptr, const mcrt_tile* rect = nullptr) {
    Lane* s = &sc->lanes[li];
    std::memset(&p, 0, sizeof p);
    p.scene = static_cast<const uint8_t*>(sc->blob.ptr);
    p.seed_table = sc->seed_table;
    p.seed_table_full = sc->seed_table_full;
    {
        static const bool decisions = [] {  // development knob: MCRT_BUNDLE_DECISIONS=0 traces every hit's shadow rays
            const char* e = std::getenv("MCRT_BUNDLE_DECISIONS");
            return !(e && e[0] == '0');
        }();
        p.bundle_decisions = decisions ? 1 : 0;
        static const bool inside_fast = [] {  // development knob: MCRT_INSIDE_FAST=0 sends every candidate through the general routine
            const char* e = std::getenv("MCRT_INSIDE_FAST");
            return !(e && e[0] == '0');
        }();
        p.inside_fast = inside_fast ? 1 : 0;
    }
    p.cfg = *cfg;
    if (cfg->width > 0 && cfg->height > 0) {
        p.inv_width = 1.0f / static_cast<float>(cfg->width);
        p.inv_height = 1.0f / static_cast<float>(cfg->height);
        static const bool fast_div = [] {  // development knob: MCRT_DIV_FRAME=0 takes the general division everywhere
            const char* e = std::getenv("MCRT_DIV_FRAME");
            return !(e && e[0] == '0');
        }();
        p.div_frame = (fast_div && cfg->width <= kDivFrameMax && cfg->height <= kDivFrameMax) ? 1 : 0;
    }
    p.shard = make_shard(*cfg, first + li * step, step * n_lanes);
    p.shard.pack_first = li;
    p.shard.pack_step = n_lanes;
    if (rect) {  // one tile: the rectangle (renderTile); the output holds its pixel rows, packed
        p.rect_x = rect->x, p.rect_y = rect->y, p.rect_w = rect->width, p.rect_h = rect->height;
        p.shard.first = 0, p.shard.step = 1, p.shard.tiles_x = 1, p.shard.tiles_y = 1, p.shard.owned_rows = 1;
        p.shard.pack_first = 0, p.shard.pack_step = 1;
    }
    p.layout = layout;
    p.out = d_out;
    p.out8 = d_out8;
    p.draws_per_sample = draws_per_sample(*cfg);
    const bool fits = sc->alpha_words <= static_cast<uint32_t>(kAlphaLdsWordsMax) && sc->n_meshes * 6 <= static_cast<uint32_t>(kFaceLdsEntriesMax);
    p.scene_in_lds = fits ? 1 : 0;
    p.scene_posed = sc->posed ? 1 : 0;
    p.lds_alpha_words = fits ? static_cast<int>(sc->alpha_words) : 0;
    p.lds_face_entries = fits ? static_cast<int>(sc->n_meshes * 6) : 0;
    // The budget bounds the batch size; when the device cannot give that much right now (other
    // allocations, a shared GPU) the budget is halved — down to one tile row per batch — and the
    // lane's buffers are re-planned, instead of failing the render.
    WorkspaceBytes w{};
    std::vector<int> row_touched;
    if (rect)
        row_touched.assign(1, 1);  // the rectangle counts as touched (plan_tiles decides on the device)
    else
        touched_tiles_per_row(sc, *cfg, p.shard, row_touched);
    for (;;) {
        if (!sc->budget) sc->budget = workspace_budget(sc->device);
        w = plan_workspace(p, sc->budget / static_cast<size_t>(n_lanes), row_touched.empty() ? nullptr : row_touched.data());
        if (p.rows_per_batch <= 0 && p.shard.owned_rows > 0)
            return fail(MCRT_ERR_INVALID, "one tile row holds more than 2^31 samples (width x tile size x samples per pixel)");
        hipError_t e = hipSuccess;
        auto want = [&](DeviceBuffer& b, size_t bytes) {
            if (e == hipSuccess) e = b.reserve(bytes);
        };
        want(s->tile_rng, w.tile_rng);
        want(s->tile_draws, w.tile_draws);
        want(s->scol, w.scol);
        want(s->end, w.end);
        want(s->units, w.units);
        want(s->tile_mask, w.tile_mask);
        want(s->unit_hits, w.unit_hits);
        for (auto& q : s->queues) want(q, w.queue_each);
        want(s->texel_refs, w.texel_refs);
        want(s->targets, w.targets);
        want(s->cand, w.cand);
        want(s->lit[0], w.lit0);
        want(s->lit[1], w.lit1);
        want(s->stack, w.stack);
        {
            const void* before = s->counters.ptr;
            want(s->counters, w.counters);
            if (e == hipSuccess && s->counters.ptr != before) e = hipMemset(s->counters.ptr, 0, w.counters);  // incl. the sticky overflow word
        }
        want(s->hit_rng, w.hit_rng);
        if (e == hipSuccess) break;
        (void)hipGetLastError();
        if (e != hipErrorOutOfMemory || p.rows_per_batch <= 1 || sc->budget < (static_cast<size_t>(64) << 20))
            return hip_fail(e, "workspace allocation");
        // make room: this lane's partially grown buffers go, then try again with half the budget
        (void)hipDeviceSynchronize();
        s->tile_rng.release(), s->tile_draws.release(), s->scol.release(), s->end.release(), s->units.release(), s->tile_mask.release();
        s->unit_hits.release();
        for (auto& q : s->queues) q.release();
        s->texel_refs.release();
        s->targets.release(), s->cand.release(), s->lit[0].release(), s->lit[1].release(), s->stack.release();
        s->counters.release(), s->hit_rng.release();
        sc->budget /= 2;
    }
    if (row_touched_out) *row_touched_out = row_touched;
    p.tile_rng = w.tile_rng ? static_cast<uint32_t*>(s->tile_rng.ptr) : nullptr;
    WaveSpace& ws = p.ws;
    ws.tile_draws = static_cast<float*>(s->tile_draws.ptr);
    ws.scol = static_cast<float4*>(s->scol.ptr);
    ws.end = static_cast<uint32_t*>(s->end.ptr);
    ws.units = static_cast<uint4*>(s->units.ptr);
    ws.tile_mask = static_cast<unsigned long long*>(s->tile_mask.ptr);
    for (int k = 0; k < 2; ++k) {  // [1] = [0] + cap: the second ping-pong queue (general variants) = the deep records (flat pipeline)
        ws.q_o[k] = static_cast<float4*>(s->queues[0].ptr) + static_cast<size_t>(k) * ws.cap;
        ws.q_d[k] = static_cast<float4*>(s->queues[1].ptr) + static_cast<size_t>(k) * ws.cap;
        ws.q_p[k] = static_cast<float4*>(s->queues[2].ptr) + static_cast<size_t>(k) * ws.cap;
        ws.q_n[k] = static_cast<float4*>(s->queues[3].ptr) + static_cast<size_t>(k) * ws.cap;
        ws.q_t[k] = static_cast<float4*>(s->queues[4].ptr) + static_cast<size_t>(k) * ws.cap;
    }
    ws.q_x = static_cast<int32_t*>(s->texel_refs.ptr);
    ws.targets = static_cast<float*>(s->targets.ptr);
    ws.cand = static_cast<unsigned long long*>(s->cand.ptr);
    ws.lit[0] = static_cast<uint32_t*>(s->lit[0].ptr);
    ws.lit[1] = static_cast<uint32_t*>(s->lit[1].ptr);
    ws.unit_hits = static_cast<uint32_t*>(s->unit_hits.ptr);
    ws.stack = static_cast<float4*>(s->stack.ptr);
    ws.counters = static_cast<uint32_t*>(s->counters.ptr);
    ws.counter_base = ws.counters + kCounterWords;
    ws.frame_info = ws.counters + 2 * kCounterWords;
    ws.hit_rng = w.hit_rng ? static_cast<uint32_t*>(s->hit_rng.ptr) : nullptr;
    return MCRT_OK;
}

// the launches of one render on `stream`: lanes fork from and join the stream
// marks: per lane, or nullptr
int launch_lanes(mcrt_scene* s, const RenderParams* p, int n_lanes, hipStream_t stream, const LaunchMarks* marks = nullptr) {
    if (n_lanes > 1) {
        if (!s->fork) HIP_TRY(hipEventCreateWithFlags(&s->fork, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(s->fork, stream));
        for (int li = 1; li < n_lanes; ++li) {
            Lane& ln = s->lanes[li];
            HIP_TRY(hipStreamWaitEvent(ln.stream, s->fork, 0));
            HIP_TRY(launch_render(p[li], ln.stream, marks ? &marks[li] : nullptr));
            HIP_TRY(hipEventRecord(ln.done, ln.stream));
        }
    }
    HIP_TRY(launch_render(p[0], stream, marks ? &marks[0] : nullptr));
    for (int li = 1; li < n_lanes; ++li) HIP_TRY(hipStreamWaitEvent(stream, s->lanes[li].done, 0));
    return MCRT_OK;
}

// Every shell of the process (pooled ones included).  A render sizes its launches by whether its device is busy with
// another handle's frame at the moment it is enqueued (choose_grids): it asks the other shells' last-render events.
std::mutex g_live_mutex;
std::vector<mcrt_scene*> g_live;
void register_live(mcrt_scene* s) {
    std::lock_guard<std::mutex> lock(g_live_mutex);
    g_live.push_back(s);
}
void unregister_live(mcrt_scene* s) {
    std::lock_guard<std::mutex> lock(g_live_mutex);
    g_live.erase(std::remove(g_live.begin(), g_live.end(), s), g_live.end());
}
bool device_shared(const mcrt_scene* s) {
    static const int forced = [] {  // development knob: MCRT_SHARED_GRIDS=0 / 1 fixes the answer
        const char* e = std::getenv("MCRT_SHARED_GRIDS");
        return e ? (std::atoi(e) != 0 ? 1 : 0) : -1;
    }();
    if (forced >= 0) return forced != 0;
    bool shared = false;
    {
        std::lock_guard<std::mutex> lock(g_live_mutex);
        for (const mcrt_scene* q : g_live) {
            if (q == s || q->device != s->device) continue;
            hipEvent_t e = q->busy_probe.load(std::memory_order_acquire);
            if (e && hipEventQuery(e) == hipErrorNotReady) {
                shared = true;
                break;
            }
        }
    }
    (void)hipGetLastError();  // hipErrorNotReady is an answer, not a failure of this render
    return shared;
}

RngKey rng_key_of(const RenderParams& p) {
    RngKey k;
    k.ptr = p.tile_rng;
    k.width = p.cfg.width, k.tile_size = p.cfg.tile_size;
    k.first = p.shard.first, k.step = p.shard.step, k.tiles_x = p.shard.tiles_x, k.owned_rows = p.shard.owned_rows;
    k.rect[0] = p.rect_x, k.rect[1] = p.rect_y, k.rect[2] = p.rect_w, k.rect[3] = p.rect_h;
    k.parts = p.stream_parts, k.part_twists = p.stream_part_twists;
    return k;
}

// MCRT_GRAPH=0 turns launch recording off (every render then issues its four launches per lane and pass itself)
bool graphs_enabled() {
    static const bool v = [] {
        const char* e = std::getenv("MCRT_GRAPH");
        return !e || std::atoi(e) != 0;
    }();
    return v;
}

// deepest recursion the workspace is laid out for (one stack slot per level and sample; the general
// variants also keep one queue counter per level)
constexpr int kMaxBounces = 4000;

// The launches of one render, directly or — when the same parameters keep coming — as one replayed
// hipGraph.  The launch sequence of a render is a pure function of its RenderParams (all control flow
// that depends on data lives on the device), so it is recorded once through stream capture on a private
// stream, lanes included, and replayed with a single hipGraphLaunch: ~75 us of launch calls per render
// become one.
int launch_or_replay(mcrt_scene* s, const RenderParams* p, int n_lanes, hipStream_t stream, bool may_record) {
    if (!graphs_enabled() || !may_record) return launch_lanes(s, p, n_lanes, stream);
    ++s->use_clock;
    mcrt_scene::Recorded* slot = nullptr;
    for (auto& r : s->recorded)
        if (r.exec && r.n_lanes == n_lanes && std::memcmp(r.p, p, sizeof(RenderParams) * n_lanes) == 0) slot = &r;
    if (!slot) {
        // Recording costs tens of milliseconds (capture + instantiation): a parameter set is recorded
        // at its kRecordAt-th sighting, earlier renders launch directly.
        constexpr int kRecordAt = 4;
        mcrt_scene::Recorded* victim = &s->recorded[0];
        for (auto& r : s->recorded) {
            if (!r.exec && r.n_lanes == n_lanes && std::memcmp(r.p, p, sizeof(RenderParams) * n_lanes) == 0) {
                slot = &r;
                break;
            }
            if (r.last_use < victim->last_use) victim = &r;
        }
        if (!slot) {  // first sighting: remember the parameters
            if (victim->exec) {  // evicting a recorded sequence: an earlier launch of it may still be running
                (void)hipDeviceSynchronize();
                (void)hipGraphExecDestroy(victim->exec);
            }
            if (victim->graph) (void)hipGraphDestroy(victim->graph);
            victim->exec = nullptr;
            victim->graph = nullptr;
            victim->n_lanes = n_lanes;
            victim->sightings = 0;
            std::memcpy(victim->p, p, sizeof(RenderParams) * kMaxLanes);
            slot = victim;
        }
        slot->last_use = s->use_clock;
        if (++slot->sightings < kRecordAt) return launch_lanes(s, p, n_lanes, stream);
        if (!s->capture_stream) HIP_TRY(hipStreamCreateWithFlags(&s->capture_stream, hipStreamNonBlocking));
        hipError_t e = hipStreamBeginCapture(s->capture_stream, hipStreamCaptureModeThreadLocal);
        if (e == hipSuccess) {
            const int rc = launch_lanes(s, p, n_lanes, s->capture_stream);
            hipGraph_t g = nullptr;
            e = hipStreamEndCapture(s->capture_stream, &g);
            if (rc == MCRT_OK && e == hipSuccess && g) {
                hipGraphExec_t ex = nullptr;
                e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
                if (e == hipSuccess && ex) {
                    slot->graph = g;
                    slot->exec = ex;
                } else {
                    (void)hipGraphDestroy(g);
                }
            } else if (g) {
                (void)hipGraphDestroy(g);
            }
        }
        if (!slot->exec) {  // recording failed: forget it and launch directly
            (void)hipGetLastError();
            slot->n_lanes = 0;
            return launch_lanes(s, p, n_lanes, stream);
        }
    }
    slot->last_use = s->use_clock;
    HIP_TRY(hipGraphLaunch(slot->exec, stream));
    return MCRT_OK;
}

// Tile rows that become final together, for a caller that downloads rows while the rest still renders.
struct RowGroup {
    std::vector<hipEvent_t> wait;  // recorded events after which the rows are complete in device memory
    std::vector<int> rows;         // tile-row indices in the frame
};
hipEvent_t next_mark(mcrt_scene* s) {  // pooled per scene shell
    if (s->marks_used == s->marks.size()) {
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
        s->marks.push_back(e);
    }
    return s->marks[s->marks_used++];
}

// enqueue one render of the shard (first, step) on `stream`.  groups != nullptr (one-shot host path): the
// launches also record events that tell when which tile rows are final, *groups lists them in
// completion order (direct launches, no graph replay: the events are this call's own).
int enqueue_render(mcrt_scene* s, const mcrt_config* cfg, int first, int step, int layout, float* d_out, uint8_t* d_out8,
                   hipStream_t stream, bool may_record = true, std::vector<RowGroup>* groups = nullptr, const mcrt_tile* rect = nullptr) {
    Shard whole = make_shard(*cfg, first, step);
    if (rect) whole.owned_rows = 1;
    if (whole.owned_rows <= 0) return MCRT_OK;
    if (cfg->max_bounces > kMaxBounces) return fail(MCRT_ERR_INVALID, "max_bounces above 4000 is not supported (one stack slot per level and sample)");
    const int n_lanes = lane_count(s, *cfg, whole);
    if (cfg->ao_enabled && cfg->ao_samples > 0 && !s->full_table_tried) {  // the first ambient-occlusion render of this shell
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (!(hipStreamIsCapturing(stream, &st) == hipSuccess && st != hipStreamCaptureStatusNone)) {  // (building it launches and waits)
            s->full_table_tried = true;
            s->seed_table_full = acquire_full_seed_table(s->device);
            s->holds_full_table = s->seed_table_full != nullptr;
        }
    }
    RenderParams p[kMaxLanes];
    std::memset(p, 0, sizeof p);
    std::vector<int> row_touched[kMaxLanes];
    for (int li = 0; li < n_lanes; ++li) {
        int rc = prepare(s, li, n_lanes, cfg, first, step, layout, d_out, d_out8, p[li], groups ? &row_touched[li] : nullptr, rect);
        if (rc != MCRT_OK) return rc;
        Lane& ln = s->lanes[li];
        if (li > 0 && !ln.stream) {
            HIP_TRY(hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&ln.done, hipEventDisableTiming));
        }
    }
    if (n_lanes > 1 && !s->fork) HIP_TRY(hipEventCreateWithFlags(&s->fork, hipEventDisableTiming));
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;  // the caller records a graph of its own
    {  // Lanes share the device among themselves; a caller's graph is replayed in circumstances unknown now (no event queries
       // while it records).  Frames of 6.4e7 samples and more keep the large grids: their kernels run for milliseconds,
       // balance counts for more than room for the neighbours (GUI defaults and 4K / 16 spp, 1.3e8 samples: 3.14 / 3.20 ms,
       // 1.49 / 1.48; 8K 17.9 / 18.0; but 4K / 4 spp, 3.3e7 samples: 0.351 / 0.340 ms).
        const int spp = cfg->samples_per_pixel > 1 ? cfg->samples_per_pixel : 1;
        const double samples = static_cast<double>(whole.owned_rows) * cfg->tile_size * cfg->width * spp;
        const bool company = n_lanes > 1 || (!capturing && device_shared(s));
        const bool shared = company && samples < 6.4e7;
        for (int li = 0; li < n_lanes; ++li) choose_grids(p[li], shared, company);
    }
    if (capturing && groups) return fail(MCRT_ERR_INVALID, "row-group events cannot be recorded into a caller's graph");
    // all renders of a handle share its workspace: they run one after the other whatever streams they are given
    if (!capturing && s->have_last && s->last_stream != stream) HIP_TRY(hipStreamWaitEvent(stream, s->last_done, 0));
    s->flags_checked = false;
    // the tiles' seeded mt19937 states: kept across renders, re-made (on the caller's stream, ahead of
    // the lanes' fork) only when the frame width, the tile size or the shard changed
    for (int li = 0; li < n_lanes; ++li) {
        if (!p[li].tile_rng) continue;
        Lane& ln = s->lanes[li];
        const RngKey k = rng_key_of(p[li]);
        if (!capturing && k == ln.rng_key) continue;
        HIP_TRY(launch_seed_tiles(p[li], stream));
        ln.rng_key = capturing ? RngKey{} : k;  // a captured seeding pass runs when the caller's graph does, not now
    }
    if (!s->last_done) {
        HIP_TRY(hipEventCreateWithFlags(&s->last_done, hipEventDisableTiming));
        s->busy_probe.store(s->last_done, std::memory_order_release);
    }
    // The pass counters run on from render to render (`resolve` leaves their values as the next pass's base): no memset per
    // pass.  Only after a render whose launches failed half way are they put back to zero, base and all but the sticky flags.
    for (int li = 0; li < n_lanes; ++li) {
        Lane& ln = s->lanes[li];
        if (!ln.counters_dirty || !ln.counters.ptr) continue;
        uint32_t* c = static_cast<uint32_t*>(ln.counters.ptr);
        HIP_TRY(hipMemsetAsync(c, 0, static_cast<size_t>(kCounterWords - 4) * 4, stream));
        HIP_TRY(hipMemsetAsync(c + kCounterWords, 0, (static_cast<size_t>(kCounterWords) + 4) * 4, stream));
        ln.counters_dirty = false;
    }
    auto launches_failed = [&]() {
        for (int li = 0; li < n_lanes; ++li) s->lanes[li].counters_dirty = true;
    };
    int rc;
    if (groups) {
        // which rows are final when: a row that holds no touched tile is complete behind plan_tiles (which
        // renders background tiles itself); the rows of a pass behind its resolve; everything at the end
        LaunchMarks marks[kMaxLanes];
        std::vector<hipEvent_t> batch_events[kMaxLanes];
        RowGroup early, late;
        std::vector<RowGroup> per_batch;
        for (int li = 0; li < n_lanes; ++li) {
            const RenderParams& q = p[li];
            const int batches = (q.shard.owned_rows + q.rows_per_batch - 1) / q.rows_per_batch;
            auto row_of = [&](int j) { return q.shard.first + j * q.shard.step; };
            if (batches <= 1) {
                hipEvent_t planned = nullptr;
                if (q.bg_in_plan) {
                    planned = next_mark(s);
                    if (!planned) return fail(MCRT_ERR_HIP, "event creation failed");
                    marks[li].after_plan = planned;
                    early.wait.push_back(planned);
                }
                for (int j = 0; j < q.shard.owned_rows; ++j) {
                    const bool background_only = planned && static_cast<size_t>(j) < row_touched[li].size() && row_touched[li][static_cast<size_t>(j)] == 0;
                    (background_only ? early : late).rows.push_back(row_of(j));
                }
            } else {
                batch_events[li].resize(static_cast<size_t>(batches));
                for (int b = 0; b < batches; ++b) {
                    hipEvent_t e = next_mark(s);
                    if (!e) return fail(MCRT_ERR_HIP, "event creation failed");
                    batch_events[li][static_cast<size_t>(b)] = e;
                    RowGroup g;
                    g.wait.push_back(e);
                    for (int j = b * q.rows_per_batch; j < q.shard.owned_rows && j < (b + 1) * q.rows_per_batch; ++j) g.rows.push_back(row_of(j));
                    per_batch.push_back(std::move(g));
                }
                marks[li].batch_done = batch_events[li].data();
                marks[li].n_batch_done = batches;
            }
        }
        rc = launch_lanes(s, p, n_lanes, stream, marks);
        if (rc != MCRT_OK) launches_failed();
        if (rc == MCRT_OK) {
            HIP_TRY(hipEventRecord(s->last_done, stream));
            late.wait.push_back(s->last_done);
            if (!early.rows.empty()) groups->push_back(std::move(early));
            for (auto& g : per_batch) groups->push_back(std::move(g));
            if (!late.rows.empty()) groups->push_back(std::move(late));
            s->last_stream = stream;
            s->have_last = true;
        }
        return rc;
    }
    rc = capturing ? launch_lanes(s, p, n_lanes, stream) : launch_or_replay(s, p, n_lanes, stream, may_record);
    if (rc != MCRT_OK) launches_failed();
    if (rc == MCRT_OK && !capturing) {
        HIP_TRY(hipEventRecord(s->last_done, stream));
        s->last_stream = stream;
        s->have_last = true;
    }
    return rc;
}

}  // namespace

namespace {
void destroy_scene_now(mcrt_scene* s);

size_t workspace_bytes(const mcrt_scene* s) {
    size_t n = s->blob.bytes + s->frame.bytes;
    for (const Lane& ln : s->lanes) {
        n += ln.tile_rng.bytes + ln.tile_draws.bytes + ln.scol.bytes + ln.end.bytes + ln.units.bytes + ln.unit_hits.bytes + ln.tile_mask.bytes;
        for (const auto& q : ln.queues) n += q.bytes;
        n += ln.texel_refs.bytes;
        n += ln.targets.bytes + ln.cand.bytes + ln.lit[0].bytes + ln.lit[1].bytes + ln.stack.bytes + ln.counters.bytes + ln.hit_rng.bytes;
    }
    return n;
}

std::mutex g_pool_mutex;
std::vector<mcrt_scene*> g_pool;  // idle scene shells, at most one per device

// ---- per-device seed tables (kernels.h: kSeedWindow words of mt[397] by seed; MCRT_SEED_TABLE=0 turns them off)
struct SeedTable {
    uint32_t* ptr = nullptr;
    int users = 0;  // scene shells (live or pooled) that hold the pointer
};
std::mutex g_seed_mutex;
std::vector<SeedTable> g_seed_tables;  // by device

// the table of `device` (built on first use: one allocation, one ~2 ms kernel), or nullptr when turned off /
// not available — the kernels then seed every hit by the recurrence
const uint32_t* acquire_seed_table(int device) {
    static const bool enabled = [] {
        const char* e = std::getenv("MCRT_SEED_TABLE");
        return !e || std::atoi(e) != 0;
    }();
    if (!enabled) return nullptr;
    std::lock_guard<std::mutex> lock(g_seed_mutex);
    if (g_seed_tables.size() <= static_cast<size_t>(device)) g_seed_tables.resize(static_cast<size_t>(device) + 1);
    SeedTable& t = g_seed_tables[static_cast<size_t>(device)];
    if (!t.ptr) {
        uint32_t* p = nullptr;
        if (hipMalloc(&p, static_cast<size_t>(kSeedWindow) * 4) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        if (launch_build_seed_table(p, nullptr) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipFree(p);
            return nullptr;
        }
        t.ptr = p;
    }
    ++t.users;
    return t.ptr;
}
// ---- the table for EVERY seed: mt[397] of all 2^32 seeds, 16 GiB of the device's HBM.  The ambient-occlusion seeds,
// (unsigned)(P.x * 73856093 + P.y * 19349663 + P.z * 83492791) (raytracer.cpp:122-123), cover the whole 32-bit range, and the
// 397-step recurrence is over a quarter of the AO stage's cycles (its multiply issues at a quarter of the rate).  Built on
// a device's first AO render (0.2 s), when the device has the room; MCRT_AO_SEED_TABLE=0 turns it off, mcrt_trim() frees it.
std::vector<SeedTable> g_full_tables;  // by device (g_seed_mutex)
const uint32_t* acquire_full_seed_table(int device) {
    static const bool enabled = [] {
        const char* e = std::getenv("MCRT_AO_SEED_TABLE");
        return !e || std::atoi(e) != 0;
    }();
    if (!enabled) return nullptr;
    std::lock_guard<std::mutex> lock(g_seed_mutex);
    if (g_full_tables.size() <= static_cast<size_t>(device)) g_full_tables.resize(static_cast<size_t>(device) + 1);
    SeedTable& t = g_full_tables[static_cast<size_t>(device)];
    if (!t.ptr) {
        const size_t bytes = static_cast<size_t>(1) << 34;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < bytes * 3) {  // only where it is a small part of what is free
            (void)hipGetLastError();
            return nullptr;
        }
        uint32_t* p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        hipError_t e = hipSuccess;
        for (uint32_t part = 0; part < 16u && e == hipSuccess; ++part) e = launch_build_seed_table_range(p, part << 28, 1u << 28, nullptr);
        if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            (void)hipFree(p);
            return nullptr;
        }
        t.ptr = p;
    }
    ++t.users;
    return t.ptr;
}
void release_full_seed_table(int device) {
    std::lock_guard<std::mutex> lock(g_seed_mutex);
    if (static_cast<size_t>(device) < g_full_tables.size() && g_full_tables[static_cast<size_t>(device)].users > 0)
        --g_full_tables[static_cast<size_t>(device)].users;
}
void release_seed_table(int device) {
    std::lock_guard<std::mutex> lock(g_seed_mutex);
    if (static_cast<size_t>(device) < g_seed_tables.size() && g_seed_tables[static_cast<size_t>(device)].users > 0)
        --g_seed_tables[static_cast<size_t>(device)].users;
}
void free_unused_seed_tables() {  // mcrt_trim
    std::lock_guard<std::mutex> lock(g_seed_mutex);
    for (size_t d = 0; d < g_full_tables.size(); ++d) {
        SeedTable& t = g_full_tables[d];
        if (t.ptr && t.users == 0) {
            (void)hipSetDevice(static_cast<int>(d));
            (void)hipFree(t.ptr);
            t.ptr = nullptr;
        }
    }
    for (size_t d = 0; d < g_seed_tables.size(); ++d) {
        SeedTable& t = g_seed_tables[d];
        if (t.ptr && t.users == 0) {
            (void)hipSetDevice(static_cast<int>(d));
            (void)hipFree(t.ptr);
            t.ptr = nullptr;
        }
    }
}

// keeps `s` for reuse unless it is large — MCRT_POOL_MB, by default a twelfth of the device's memory (24 GB of the
// MI355X's 288: the 1080p and 4K frames of BASELINE.json stay pooled, and a host application that never calls
// mcrt_trim() does not sit on a fifth of the card; the one-shot entry points plan their workspace to stay below it, see
// render_to_host) — or the device already has one
size_t pool_limit(int device) {
    static const long long forced_mb = [] {
        const char* e = std::getenv("MCRT_POOL_MB");
        return e ? std::atoll(e) : -1ll;
    }();
    return forced_mb >= 0 ? static_cast<size_t>(forced_mb) << 20 : device_total_memory(device) / 12;
}
bool pool_scene(mcrt_scene* s) {
    const size_t limit = pool_limit(s->device);
    if (workspace_bytes(s) > limit) return false;
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    for (mcrt_scene* q : g_pool)
        if (q->device == s->device) return false;
    g_pool.push_back(s);
    return true;
}
// device < 0: any
mcrt_scene* take_pooled_scene(int device) {
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    for (size_t i = 0; i < g_pool.size(); ++i)
        if (device < 0 || g_pool[i]->device == device) {
            mcrt_scene* s = g_pool[i];
            g_pool.erase(g_pool.begin() + static_cast<long>(i));
            return s;
        }
    return nullptr;
}
}  // namespace

extern "C" {

int mcrt_abi_version(void) { return MCRT_ABI_VERSION; }

int mcrt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* mcrt_last_error(void) { return g_err.c_str(); }

void mcrt_config_init(mcrt_config* c) {  // raytracer.h:10-38
    if (!c) return;
    c->width = 256;
    c->height = 256;
    c->max_bounces = 3;
    c->samples_per_pixel = 1;
    c->tile_size = 32;
    c->thread_count = 0;
    c->soft_shadows = 1;
    c->shadow_samples = 8;
    c->ao_enabled = 0;
    c->ao_samples = 8;
    c->ao_radius = 3.0f;
    c->ao_intensity = 0.5f;
    c->dof_enabled = 0;
    c->aperture = 0.5f;
    c->focus_distance = 0.0f;
    c->gradient_bg = 1;
    c->gradient_scale = 1.0f;
    const float center[4] = {0.91f, 0.89f, 0.86f, 1.0f}, edge[4] = {0.56f, 0.63f, 0.71f, 1.0f};
    std::memcpy(c->bg_center, center, 16);
    std::memcpy(c->bg_edge, edge, 16);
}

int mcrt_generate_tiles(int w, int h, int ts, mcrt_tile* tiles, int capacity) {  // tile_renderer.cpp:18-39
    if (w <= 0 || h <= 0 || ts <= 0) return 0;
    int cols = (w + ts - 1) / ts, rows = (h + ts - 1) / ts;
    int n = 0;
    for (int ty = 0; ty < rows; ++ty)
        for (int tx = 0; tx < cols; ++tx, ++n) {
            if (!tiles || n >= capacity) continue;
            mcrt_tile& t = tiles[n];
            t.x = tx * ts;
            t.y = ty * ts;
            t.width = ts < w - t.x ? ts : w - t.x;
            t.height = ts < h - t.y ? ts : h - t.y;
        }
    return n;
}

size_t mcrt_scene_flatten(const mcrt_scene_desc* desc, void* blob, size_t capacity) {
    std::vector<uint8_t> b;
    std::string err;
    if (!flatten_scene(desc, b, err)) {
        g_err = err;
        return 0;
    }
    if (blob && capacity) std::memcpy(blob, b.data(), b.size() < capacity ? b.size() : capacity);
    return b.size();
}

}  // extern "C"
namespace {
// uploads an already flattened scene to `device` (a pooled shell with its workspace when one is idle)
int create_scene_from_blob(const std::vector<uint8_t>& b, int device, mcrt_scene** out) {
    *out = nullptr;
    int n = mcrt_device_count();
    if (n <= 0) return fail(MCRT_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
    if (device < 0 || device >= n) return fail(MCRT_ERR_NO_DEVICE, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    mcrt_scene* s = take_pooled_scene(device);  // an idle shell with its workspace, or nullptr
    if (!s) {
        s = new mcrt_scene();
        s->device = device;
        register_live(s);
    }
    s->forced_lanes = 0;
    s->budget = 0;  // a budget halved under memory pressure is not inherited
    s->have_last = false;  // a pooled shell was synchronised when its previous owner let go of it
    s->last_stream = nullptr;
    s->marks_used = 0;
    if (!s->holds_seed_table) {
        s->seed_table = acquire_seed_table(device);
        s->holds_seed_table = s->seed_table != nullptr;
    }
    const FlatHeader* fh = reinterpret_cast<const FlatHeader*>(b.data());
    const FlatMesh* fm = reinterpret_cast<const FlatMesh*>(b.data() + fh->mesh_offset);
    s->alpha_words = fh->alpha_words;
    s->n_meshes = fh->n_meshes;
    s->posed = false;
    for (uint32_t i = 0; i < fh->n_meshes; ++i) s->posed = s->posed || (fm[i].flags & MESH_ROTATED) != 0;
    s->host_meshes.assign(b.begin(), b.begin() + fh->mesh_offset + sizeof(FlatMesh) * fh->n_meshes);
    hipError_t e = s->blob.reserve(b.size());
    const size_t staging_need = b.size() + 64;  // the blob, then the lanes' flag words
    if (e == hipSuccess && s->staging_bytes < staging_need) {
        if (s->staging) (void)hipHostFree(s->staging);
        s->staging = nullptr;
        s->staging_bytes = 0;
        e = hipHostMalloc(&s->staging, staging_need + (staging_need >> 2), hipHostMallocDefault);
        if (e == hipSuccess) s->staging_bytes = staging_need + (staging_need >> 2);
    }
    if (e == hipSuccess) {
        std::memcpy(s->staging, b.data(), b.size());
        e = hipMemcpy(s->blob.ptr, s->staging, b.size(), hipMemcpyHostToDevice);
    }
    for (int i = 0; i < 4 && e == hipSuccess; ++i)
        if (!s->ev[i]) e = hipEventCreate(&s->ev[i]);
    if (e != hipSuccess) {
        destroy_scene_now(s);
        return hip_fail(e, "scene upload");
    }
    *out = s;
    return MCRT_OK;
}
}  // namespace
extern "C" {

int mcrt_scene_create(const mcrt_scene_desc* desc, int device, mcrt_scene** out) {
    if (!out) return fail(MCRT_ERR_INVALID, "out is NULL");
    *out = nullptr;
    std::vector<uint8_t> b;
    std::string err;
    if (!flatten_scene(desc, b, err)) return fail(MCRT_ERR_INVALID, err);
    return create_scene_from_blob(b, device, out);
}

void mcrt_scene_destroy(mcrt_scene* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    (void)hipDeviceSynchronize();  // renders of this scene may still be running on the caller's streams
    if (!s->flags_checked) (void)mcrt_scene_check(s);  // reads and clears the lanes' sticky overflow words: the next owner of the workspace starts clean
    // A modest workspace is kept for the next scene on this device (one idle shell per device): a fresh
    // hipMalloc of the lanes' buffers costs milliseconds per render call of the one-shot API
    // (TileRenderer::render), tens of GB for large frames take far longer.  mcrt_trim() lets go of it.
    if (pool_scene(s)) return;
    destroy_scene_now(s);
}

void mcrt_trim(void) {
    for (;;) {
        mcrt_scene* s = take_pooled_scene(-1);
        if (!s) break;
        (void)hipSetDevice(s->device);
        destroy_scene_now(s);
    }
    free_unused_seed_tables();
}

namespace {
void destroy_scene_now(mcrt_scene* s) {
    if (!s) return;
    unregister_live(s);  // before its events go
    if (s->holds_seed_table) release_seed_table(s->device);
    if (s->holds_full_table) release_full_seed_table(s->device);
    s->blob.release();  // the other buffers are released by their destructors below
    for (auto& ln : s->lanes) {
        if (ln.stream) (void)hipStreamSynchronize(ln.stream);
        if (ln.done) (void)hipEventDestroy(ln.done);
        if (ln.stream) (void)hipStreamDestroy(ln.stream);
    }
    for (auto& r : s->recorded) {
        if (r.exec) (void)hipGraphExecDestroy(r.exec);
        if (r.graph) (void)hipGraphDestroy(r.graph);
    }
    if (s->capture_stream) (void)hipStreamDestroy(s->capture_stream);
    if (s->fork) (void)hipEventDestroy(s->fork);
    if (s->last_done) (void)hipEventDestroy(s->last_done);
    if (s->staging) (void)hipHostFree(s->staging);
    if (s->main_stream) (void)hipStreamDestroy(s->main_stream);
    if (s->copy_stream) (void)hipStreamDestroy(s->copy_stream);
    for (hipEvent_t m : s->marks) (void)hipEventDestroy(m);
    for (auto& e : s->ev)
        if (e) (void)hipEventDestroy(e);
    delete s;
}
}  // namespace

int mcrt_scene_check(mcrt_scene* s) {
    if (!s) return fail(MCRT_ERR_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipDeviceSynchronize());
    bool flagged = false;
    uint32_t* flags = reinterpret_cast<uint32_t*>(static_cast<char*>(s->staging) + (s->staging_bytes - 64));  // pinned
    int li = 0;
    for (Lane& ln : s->lanes) {
        ++li;
        if (!ln.counters.ptr) continue;
        uint32_t* word = static_cast<uint32_t*>(ln.counters.ptr) + (kCounterWords - 1);
        uint32_t local = 0;
        uint32_t* dst = s->staging ? &flags[li - 1] : &local;
        HIP_TRY(hipMemcpy(dst, word, 4, hipMemcpyDeviceToHost));
        const uint32_t flag = *dst;
        if (flag) {  // reported once: the word is cleared so that later renders (and the next owner of a pooled workspace) start clean
            flagged = true;
            HIP_TRY(hipMemset(word, 0, 4));
        }
    }
    s->flags_checked = true;
    if (flagged) return fail(MCRT_ERR_HIP, "internal error: more tiles were touched than the workspace was planned for");
    return MCRT_OK;
}

int mcrt_scene_set_lanes(mcrt_scene* s, int lanes) {
    if (!s || lanes < 0) return fail(MCRT_ERR_INVALID, "bad argument");
    s->forced_lanes = lanes;
    return MCRT_OK;
}

int mcrt_owned_pixel_rows(const mcrt_config* cfg, int first, int step) {
    if (!cfg || !valid_frame(cfg)) return 0;
    Shard sh = make_shard(*cfg, first, step);
    return sh.owned_rows * cfg->tile_size;
}

int mcrt_render_device(mcrt_scene* s, const mcrt_config* cfg, int first, int step, int layout, float* d_out,
                       void* stream) {
    if (!s || !cfg || !d_out) return fail(MCRT_ERR_INVALID, "NULL argument");
    if (!valid_frame(cfg)) return MCRT_OK;  // zero tiles
    if (first < 0 || step < 1) return fail(MCRT_ERR_INVALID, "tile_row_first must be >= 0 and tile_row_step >= 1");
    HIP_TRY(hipSetDevice(s->device));
    return enqueue_render(s, cfg, first, step, layout, d_out, nullptr, static_cast<hipStream_t>(stream));
}

int mcrt_render_device_ex(mcrt_scene* s, const mcrt_config* cfg, int first, int step, int layout, float* d_out_f32,
                          uint8_t* d_out_rgba8, void* stream) {
    if (!s || !cfg || (!d_out_f32 && !d_out_rgba8)) return fail(MCRT_ERR_INVALID, "NULL argument");
    if (!valid_frame(cfg)) return MCRT_OK;  // zero tiles
    if (first < 0 || step < 1) return fail(MCRT_ERR_INVALID, "tile_row_first must be >= 0 and tile_row_step >= 1");
    HIP_TRY(hipSetDevice(s->device));
    return enqueue_render(s, cfg, first, step, layout, d_out_f32, d_out_rgba8, static_cast<hipStream_t>(stream));
}

int mcrt_time_render_device(mcrt_scene* s, const mcrt_config* cfg, int first, int step, int layout, float* d_out,
                            void* stream, int iters, float* avg_render_ms) {
    if (!s || !cfg || !d_out || iters < 1) return fail(MCRT_ERR_INVALID, "bad argument");
    if (!valid_frame(cfg)) return fail(MCRT_ERR_INVALID, "empty frame");
    HIP_TRY(hipSetDevice(s->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    double sum = 0.0;
    for (int i = 0; i < iters; ++i) {
        // the events bracket the whole pipeline of the frame (fork, every lane, join) on `stream`
        HIP_TRY(hipEventRecord(s->ev[0], st));
        int rc = enqueue_render(s, cfg, first, step, layout, d_out, nullptr, st);
        if (rc != MCRT_OK) return rc;
        HIP_TRY(hipEventRecord(s->ev[3], st));
        HIP_TRY(hipEventSynchronize(s->ev[3]));
        float a = 0;
        HIP_TRY(hipEventElapsedTime(&a, s->ev[0], s->ev[3]));
        sum += a;
    }
    if (avg_render_ms) *avg_render_ms = static_cast<float>(sum / iters);
    return MCRT_OK;
}

int mcrt_unpack_rows_device(const mcrt_config* cfg, int first, int step, const float* d_packed, float* d_frame,
                            void* stream) {
    if (!cfg || !d_packed || !d_frame) return fail(MCRT_ERR_INVALID, "NULL argument");
    if (!valid_frame(cfg)) return MCRT_OK;
    Shard sh = make_shard(*cfg, first, step);
    HIP_TRY(launch_unpack_rows(*cfg, sh, d_packed, d_frame, static_cast<hipStream_t>(stream)));
    return MCRT_OK;
}

int mcrt_assemble_frame_device(const mcrt_config* cfg, int world, const float* d_gathered, size_t rank_stride_pixels,
                               float* d_frame, void* stream) {
    if (!cfg || !d_gathered || !d_frame || world < 1) return fail(MCRT_ERR_INVALID, "bad argument");
    if (!valid_frame(cfg)) return MCRT_OK;
    const int tiles_y = (cfg->height + cfg->tile_size - 1) / cfg->tile_size;
    const size_t need = static_cast<size_t>((tiles_y + world - 1) / world) * cfg->tile_size * cfg->width;
    if (world > 1 && rank_stride_pixels < need) return fail(MCRT_ERR_INVALID, "rank stride smaller than a rank's packed rows");
    HIP_TRY(launch_assemble_frame(*cfg, world, d_gathered, rank_stride_pixels, d_frame, static_cast<hipStream_t>(stream)));
    return MCRT_OK;
}

int mcrt_quantize_rgba8_device(const float* d_rgba, uint8_t* d_out, size_t n_pixels, void* stream) {
    if (!d_rgba || !d_out) return fail(MCRT_ERR_INVALID, "NULL argument");
    HIP_TRY(launch_quantize(d_rgba, d_out, n_pixels, static_cast<hipStream_t>(stream)));
    return MCRT_OK;
}

}  // extern "C"
namespace {

int validate_config(const mcrt_config* cfg) {
    if (cfg->max_bounces > kMaxBounces) return fail(MCRT_ERR_INVALID, "max_bounces above 4000 is not supported (one stack slot per level and sample)");
    return MCRT_OK;
}

struct CopySpan {  // a contiguous run: device bytes → host bytes
    const char* src;
    char* dst;
    size_t bytes;
};

int one_shot_streams(mcrt_scene* s) {
    if (!s->main_stream) HIP_TRY(hipStreamCreateWithFlags(&s->main_stream, hipStreamNonBlocking));
    if (!s->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking));
    return MCRT_OK;
}

// pixel rows of tile row r of the frame
int tile_row_height(const mcrt_config& c, int r) { return std::min(c.tile_size, c.height - r * c.tile_size); }

// One rank of a host-buffer render: a scene shell on its device with the shard (rank, n_ranks) enqueued.
struct HostRank {
    mcrt_scene* scene = nullptr;
    int device = 0;
    std::vector<RowGroup> groups;
};

// Peer access between the gather root and a rank's device, both directions, once per pair and process.  false: the
// pair cannot reach each other directly (the caller falls back to per-device downloads).
bool peer_access(int root, int other) {
    if (root == other) return true;
    static std::mutex mu;
    static std::vector<std::pair<int, int>> enabled;
    std::lock_guard<std::mutex> lock(mu);
    for (const auto& pr : enabled)
        if (pr.first == root && pr.second == other) return true;
    int a = 0, b = 0;
    if (hipDeviceCanAccessPeer(&a, root, other) != hipSuccess || hipDeviceCanAccessPeer(&b, other, root) != hipSuccess || !a || !b) {
        (void)hipGetLastError();
        return false;
    }
    auto enable = [](int dev, int peer) {
        if (hipSetDevice(dev) != hipSuccess) return false;
        const hipError_t e = hipDeviceEnablePeerAccess(peer, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return false;
        (void)hipGetLastError();
        return true;
    };
    if (!enable(root, other) || !enable(other, root)) {
        (void)hipGetLastError();
        return false;
    }
    enabled.emplace_back(root, other);
    return true;
}

// Renders the frame on the given devices — rank r of N takes tile rows r, r+N, ... (cyclic: the figure sits in
// the middle rows) on devices[r] — and assembles it in `out`: float4 pixels (px_bytes = 16) or the RGBA8 plane
// quantised in the kernels' epilogue (px_bytes = 4: a quarter of the bytes on every link).
//   gather = 0: every device downloads its own rows straight into the host frame (N PCIe links in parallel,
//               no device-to-device traffic); rows that are final early travel while the rest still renders.
//   gather = 1: the ranks' packed rows go to devices[0] by peer copies (xGMI; peer access enabled once per pair — a
//               pair without it falls back to gather = 0); as each rank's rows arrive they are un-permuted into the
//               frame on the root and downloaded, so progress follows the ranks as they land.
// Progress callbacks (exactly totalTiles, done = 1..total) fire on the calling thread as rows land in `out`.
int render_to_host(const mcrt_scene_desc* desc, const mcrt_config* cfg, void* out, int px_bytes, mcrt_progress_fn progress, void* user,
                   const int* devices, int n_ranks, int gather) {
    const double t0 = now_ms();
    std::vector<uint8_t> blob;
    std::string err;
    if (!flatten_scene(desc, blob, err)) return fail(MCRT_ERR_INVALID, err);
    const Shard all = make_shard(*cfg, 0, 1);
    if (n_ranks > all.tiles_y) n_ranks = all.tiles_y;  // no more ranks than tile rows
    if (n_ranks <= 1) gather = 0;
    for (int r = 1; r < n_ranks && gather; ++r)
        if (!peer_access(devices[0], devices[r])) gather = 0;
    const int W = cfg->width, T = cfg->tile_size;
    const size_t px = static_cast<size_t>(px_bytes);
    const size_t row_bytes = static_cast<size_t>(T) * W * px;
    const size_t whole_bytes = static_cast<size_t>(W) * cfg->height * px;
    const int total_tiles = all.tiles_x * all.tiles_y;
    char* out_bytes = static_cast<char*>(out);
    std::vector<HostRank> ranks(static_cast<size_t>(n_ranks));
    int rc = MCRT_OK;
    auto cleanup = [&](int code) {
        for (HostRank& r : ranks)
            if (r.scene) mcrt_scene_destroy(r.scene);  // synchronises, checks, pools the workspace
        return code;
    };
    const double t1 = now_ms();
    // ---- every rank: upload, enqueue its shard, note when which rows are final
    for (int r = 0; r < n_ranks && rc == MCRT_OK; ++r) {
        HostRank& hr = ranks[static_cast<size_t>(r)];
        hr.device = devices[r];
        rc = create_scene_from_blob(blob, hr.device, &hr.scene);
        if (rc != MCRT_OK) break;
        mcrt_scene* s = hr.scene;
        // One lane: after a render that forked internal streams the runtime serves pageable copies from its staged
        // path (1 MB chunks, ~15 GB/s: 2.3 ms per 1080p call instead of 0.8, tools/micro/hostpath.cpp), and
        // for a host-buffer render the download, not the chain of kernels, is the longer part.
        s->forced_lanes = 1;
        rc = one_shot_streams(s);
        if (rc != MCRT_OK) break;
        const Shard mine = make_shard(*cfg, r, n_ranks);
        const bool packed = n_ranks > 1;
        const size_t frame_bytes = packed ? static_cast<size_t>(mine.owned_rows) * row_bytes : whole_bytes;
        size_t want = frame_bytes;
        if (gather && r == 0)  // the root also holds every rank's packed rows and the assembled frame
            want = static_cast<size_t>(n_ranks) * (static_cast<size_t>((all.tiles_y + n_ranks - 1) / n_ranks) * row_bytes) + whole_bytes;
        hipError_t e = s->frame.reserve(want);
        if (e != hipSuccess) {
            rc = hip_fail(e, "frame allocation");
            break;
        }
        {   // A one-shot render plans its workspace to stay poolable: with the default budget (a third of the device) an
            // 8K / 64 spp frame would take tens of GB that no pool keeps — allocated and freed on EVERY call, and a hipMalloc
            // of that size now and then stalls for seconds (3.6-4.5 s observed, tools/gpu_hostpath_8k.py).  Below the pool's
            // limit the frame is cut into a few more passes (8K: +2 % device time) and the next call finds its buffers.
            const size_t limit = pool_limit(s->device);
            const size_t fixed = s->blob.bytes + s->frame.bytes + (static_cast<size_t>(512) << 20);  // + tile states, tables' slack
            if (limit > 2 * fixed) {
                const size_t room = (limit - fixed) / 10 * 9;
                const size_t budget = workspace_budget(s->device);
                s->budget = budget < room ? budget : room;
            }
        }
        if (r == 0) (void)hipEventRecord(s->ev[0], s->main_stream);
        rc = enqueue_render(s, cfg, r, n_ranks, packed ? MCRT_LAYOUT_PACKED : MCRT_LAYOUT_FRAME, px_bytes == 16 ? static_cast<float*>(s->frame.ptr) : nullptr,
                            px_bytes == 4 ? static_cast<uint8_t*>(s->frame.ptr) : nullptr, s->main_stream, /*may_record=*/false, &hr.groups);
        if (rc == MCRT_OK && r == 0) (void)hipEventRecord(s->ev[3], s->main_stream);
    }
    if (rc != MCRT_OK) return cleanup(rc);
    const double t2 = now_ms();
    int done_tiles = 0;
    auto report_rows = [&](const std::vector<int>& rows) {
        if (!progress) return;
        for (size_t i = 0; i < rows.size(); ++i)
            for (int x = 0; x < all.tiles_x; ++x) progress(++done_tiles, total_tiles, user);
    };
    // device bytes → host frame for a list of tile rows; `packed_rank` >= 0: the source holds only that rank's rows, packed
    auto spans_of = [&](const char* src0, const std::vector<int>& rows, int packed_rank) {
        std::vector<CopySpan> spans;
        for (size_t i = 0; i < rows.size();) {
            size_t j = i + 1;
            if (packed_rank < 0)  // consecutive tile rows of a frame-shaped source are one run
                while (j < rows.size() && rows[j] == rows[j - 1] + 1) ++j;
            size_t bytes = 0;
            for (size_t k = i; k < j; ++k) bytes += static_cast<size_t>(tile_row_height(*cfg, rows[k])) * W * px;
            const int row = rows[i];
            const size_t src_off = (packed_rank < 0 ? static_cast<size_t>(row) : static_cast<size_t>((row - packed_rank) / n_ranks)) * row_bytes;
            spans.push_back(CopySpan{src0 + src_off, out_bytes + static_cast<size_t>(row) * row_bytes, bytes});
            i = j;
        }
        return spans;
    };
    // the spans on `s`'s copy stream, straight into the caller's pages.  (Measured and rejected, profiles/r03_experiments:
    // landing the rows in a pinned ring and copying them out with several threads — 1.08 ms instead of 0.77 for resident
    // pages, 9.8 ms instead of 4.0 for pages that were never touched: eight threads faulting pages of one address space
    // serialise on its lock, while the runtime's own pinning faults them in bulk.)
    auto download = [&](mcrt_scene* s, const std::vector<CopySpan>& spans) -> hipError_t {
        hipError_t ce = hipSuccess;
        for (const CopySpan& sp : spans)
            if (ce == hipSuccess) ce = hipMemcpyAsync(sp.dst, sp.src, sp.bytes, hipMemcpyDeviceToHost, s->copy_stream);
        if (ce == hipSuccess) ce = hipStreamSynchronize(s->copy_stream);
        return ce;
    };
    hipError_t e = hipSuccess;
    if (!gather) {
        // ---- downloads: a rank's row groups in completion order.  The host waits for a group's events, then
        // copies on the (idle) copy stream; early groups travel while the GPU still renders the rest.
        auto download_group = [&](int r, size_t g) -> hipError_t {
            HostRank& hr = ranks[static_cast<size_t>(r)];
            mcrt_scene* s = hr.scene;
            hipError_t ce = hipSetDevice(hr.device);
            const RowGroup& grp = hr.groups[g];
            for (size_t i = 0; i < grp.wait.size() && ce == hipSuccess; ++i) ce = hipEventSynchronize(grp.wait[i]);
            if (ce != hipSuccess) return ce;
            return download(s, spans_of(static_cast<const char*>(s->frame.ptr), grp.rows, n_ranks == 1 ? -1 : r));
        };
        if (n_ranks == 1) {
            for (size_t g = 0; g < ranks[0].groups.size() && e == hipSuccess; ++g) {
                e = download_group(0, g);
                if (e == hipSuccess) report_rows(ranks[0].groups[g].rows);
            }
        } else {
            // one copier thread per rank (the PCIe links work in parallel); the calling thread delivers the progress calls
            std::mutex mu;
            std::condition_variable cv;
            std::vector<std::pair<int, size_t>> landed;  // (rank, group) in arrival order
            int copiers_left = n_ranks;
            hipError_t first_error = hipSuccess;
            std::vector<std::thread> copiers;
            for (int r = 0; r < n_ranks; ++r)
                copiers.emplace_back([&, r] {
                    hipError_t ce = hipSuccess;
                    for (size_t g = 0; g < ranks[static_cast<size_t>(r)].groups.size() && ce == hipSuccess; ++g) {
                        ce = download_group(r, g);
                        if (ce == hipSuccess) {
                            std::lock_guard<std::mutex> lock(mu);
                            landed.emplace_back(r, g);
                            cv.notify_one();
                        }
                    }
                    std::lock_guard<std::mutex> lock(mu);
                    if (ce != hipSuccess && first_error == hipSuccess) first_error = ce;
                    --copiers_left;
                    cv.notify_one();
                });
            size_t reported = 0;
            for (;;) {
                std::unique_lock<std::mutex> lock(mu);
                cv.wait(lock, [&] { return reported < landed.size() || copiers_left == 0; });
                if (reported < landed.size()) {
                    const std::pair<int, size_t> it = landed[reported++];
                    lock.unlock();
                    report_rows(ranks[static_cast<size_t>(it.first)].groups[it.second].rows);
                } else {
                    break;
                }
            }
            for (std::thread& t : copiers) t.join();
            e = first_error;
        }
    } else {
        // ---- peer gather to the root device; rank by rank: its packed rows arrive (xGMI), one launch un-permutes them
        // into the frame on the root, its rows travel to the host, its tiles are reported
        mcrt_scene* root = ranks[0].scene;
        const size_t rank_stride = static_cast<size_t>((all.tiles_y + n_ranks - 1) / n_ranks) * row_bytes;
        char* gathered = static_cast<char*>(root->frame.ptr);  // rank 0 rendered into slot 0 already
        char* assembled = gathered + static_cast<size_t>(n_ranks) * rank_stride;
        std::vector<hipEvent_t> unpacked(static_cast<size_t>(n_ranks), nullptr);
        for (int r = 0; r < n_ranks && e == hipSuccess; ++r) {
            HostRank& hr = ranks[static_cast<size_t>(r)];
            const Shard mine = make_shard(*cfg, r, n_ranks);
            if (r > 0) {  // behind the rank's render, on the rank's stream: its packed rows to the root's slot r
                e = hipSetDevice(hr.device);
                if (e == hipSuccess)
                    e = hipMemcpyPeerAsync(gathered + static_cast<size_t>(r) * rank_stride, ranks[0].device, hr.scene->frame.ptr, hr.device,
                                           static_cast<size_t>(mine.owned_rows) * row_bytes, hr.scene->main_stream);
                hipEvent_t sent = next_mark(hr.scene);
                if (e == hipSuccess && !sent) e = hipErrorOutOfMemory;
                if (e == hipSuccess) e = hipEventRecord(sent, hr.scene->main_stream);
                if (e == hipSuccess) e = hipSetDevice(ranks[0].device);
                if (e == hipSuccess) e = hipStreamWaitEvent(root->main_stream, sent, 0);
            }
            if (e == hipSuccess) e = hipSetDevice(ranks[0].device);
            if (e == hipSuccess)
                e = px_bytes == 16 ? launch_unpack_rows(*cfg, mine, reinterpret_cast<const float*>(gathered + static_cast<size_t>(r) * rank_stride),
                                                        reinterpret_cast<float*>(assembled), root->main_stream)
                                   : launch_unpack_rows8(*cfg, mine, reinterpret_cast<const uint8_t*>(gathered + static_cast<size_t>(r) * rank_stride),
                                                         reinterpret_cast<uint8_t*>(assembled), root->main_stream);
            unpacked[static_cast<size_t>(r)] = next_mark(root);
            if (e == hipSuccess && !unpacked[static_cast<size_t>(r)]) e = hipErrorOutOfMemory;
            if (e == hipSuccess) e = hipEventRecord(unpacked[static_cast<size_t>(r)], root->main_stream);
        }
        for (int r = 0; r < n_ranks && e == hipSuccess; ++r) {
            e = hipEventSynchronize(unpacked[static_cast<size_t>(r)]);
            std::vector<int> rows;
            for (int row = r; row < all.tiles_y; row += n_ranks) rows.push_back(row);
            if (e == hipSuccess) e = download(root, spans_of(assembled, rows, -1));
            if (e == hipSuccess) report_rows(rows);
        }
    }
    if (e != hipSuccess) return cleanup(hip_fail(e, "frame download"));
    const double t3 = now_ms();
    float kernel_ms = 0.0f;
    (void)hipSetDevice(ranks[0].device);
    (void)hipEventElapsedTime(&kernel_ms, ranks[0].scene->ev[0], ranks[0].scene->ev[3]);
    for (HostRank& hr : ranks) {
        const int c = mcrt_scene_check(hr.scene);
        if (c != MCRT_OK) rc = c;
    }
    cleanup(rc);
    if (rc != MCRT_OK) return rc;
    g_timings.flatten_ms = static_cast<float>(t1 - t0);
    g_timings.h2d_ms = static_cast<float>(t2 - t1);  // uploads, workspace checks and every launch call
    g_timings.kernel_ms = kernel_ms;                  // rank 0's pipeline on the device (hipEvents)
    g_timings.d2h_ms = static_cast<float>(t3 - t2);   // from the last launch call until the last row has landed (overlaps the render)
    g_timings.total_ms = static_cast<float>(now_ms() - t0);
    return MCRT_OK;
}

}  // namespace
extern "C" {

int mcrt_render_multi(const mcrt_scene_desc* desc, const mcrt_config* cfg, float* out_rgba, mcrt_progress_fn progress,
                      void* user, const int* devices, int n_devices, int gather) {
    if (!desc || !cfg) return fail(MCRT_ERR_INVALID, "NULL argument");
    if (!valid_frame(cfg)) return MCRT_OK;  // generateTiles → empty → untouched Image (tile_renderer.cpp:144-146)
    if (!out_rgba) return fail(MCRT_ERR_INVALID, "out_rgba is NULL");
    if (validate_config(cfg) != MCRT_OK) return MCRT_ERR_INVALID;
    const int visible = mcrt_device_count();
    if (visible <= 0) return fail(MCRT_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
    std::vector<int> all;
    if (!devices || n_devices <= 0) {  // every visible device
        for (int d = 0; d < visible; ++d) all.push_back(d);
        devices = all.data();
        n_devices = visible;
    }
    for (int i = 0; i < n_devices; ++i)
        if (devices[i] < 0 || devices[i] >= visible) return fail(MCRT_ERR_NO_DEVICE, "device index out of range");
    return render_to_host(desc, cfg, out_rgba, 16, progress, user, devices, n_devices, gather ? 1 : 0);
}

int mcrt_render(const mcrt_scene_desc* desc, const mcrt_config* cfg, float* out_rgba, mcrt_progress_fn progress,
                void* user, int device) {
    if (device == MCRT_DEVICE_ALL) return mcrt_render_multi(desc, cfg, out_rgba, progress, user, nullptr, 0, 0);
    if (!desc || !cfg) return fail(MCRT_ERR_INVALID, "NULL argument");
    if (!valid_frame(cfg)) return MCRT_OK;  // generateTiles → empty → untouched Image (tile_renderer.cpp:144-146)
    if (!out_rgba) return fail(MCRT_ERR_INVALID, "out_rgba is NULL");
    if (validate_config(cfg) != MCRT_OK) return MCRT_ERR_INVALID;
    if (mcrt_device_count() <= 0) return fail(MCRT_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
    return render_to_host(desc, cfg, out_rgba, 16, progress, user, &device, 1, 0);
}

int mcrt_render_rgba8(const mcrt_scene_desc* desc, const mcrt_config* cfg, uint8_t* out_rgba8, mcrt_progress_fn progress, void* user,
                      const int* devices, int n_devices, int gather) {
    if (!desc || !cfg) return fail(MCRT_ERR_INVALID, "NULL argument");
    if (!valid_frame(cfg)) return MCRT_OK;
    if (!out_rgba8) return fail(MCRT_ERR_INVALID, "out_rgba8 is NULL");
    if (validate_config(cfg) != MCRT_OK) return MCRT_ERR_INVALID;
    const int visible = mcrt_device_count();
    if (visible <= 0) return fail(MCRT_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
    std::vector<int> all;
    if (!devices || n_devices <= 0) {  // every visible device
        for (int d = 0; d < visible; ++d) all.push_back(d);
        devices = all.data();
        n_devices = visible;
    }
    for (int i = 0; i < n_devices; ++i)
        if (devices[i] < 0 || devices[i] >= visible) return fail(MCRT_ERR_NO_DEVICE, "device index out of range");
    return render_to_host(desc, cfg, out_rgba8, 4, progress, user, devices, n_devices, gather ? 1 : 0);
}

int mcrt_render_rect(const mcrt_scene_desc* desc, const mcrt_config* cfg, const mcrt_tile* tile, float* frame_rgba, int device) {
    if (!desc || !cfg || !tile || !frame_rgba) return fail(MCRT_ERR_INVALID, "NULL argument");
    if (cfg->width <= 0 || cfg->height <= 0) return MCRT_OK;
    if (validate_config(cfg) != MCRT_OK) return MCRT_ERR_INVALID;
    if (tile->width <= 0 || tile->height <= 0) return MCRT_OK;  // renderTile's loops do not run
    if (tile->x < 0 || tile->y < 0 || tile->x > cfg->width - tile->width || tile->y > cfg->height - tile->height)
        return fail(MCRT_ERR_INVALID, "tile rectangle reaches outside the frame");
    mcrt_scene* s = nullptr;
    int rc = mcrt_scene_create(desc, device, &s);
    if (rc != MCRT_OK) return rc;
    s->forced_lanes = 1;  // one stream, like every host-buffer entry point (see render_to_host)
    const size_t row_floats = static_cast<size_t>(cfg->width) * 4;
    std::vector<float> host(row_floats * static_cast<size_t>(tile->height));
    hipError_t e = s->frame.reserve(host.size() * 4);
    if (e == hipSuccess) {
        // the rectangle is the launch's one tile; its pixel rows come back packed (full frame width)
        mcrt_config one = *cfg;
        if (one.tile_size <= 0) one.tile_size = 1;  // renderTile itself never reads tileSize
        rc = enqueue_render(s, &one, 0, 1, MCRT_LAYOUT_PACKED, static_cast<float*>(s->frame.ptr), nullptr, nullptr, /*may_record=*/false, nullptr, tile);
        if (rc == MCRT_OK) rc = mcrt_scene_check(s);
        if (rc == MCRT_OK) e = hipMemcpy(host.data(), s->frame.ptr, host.size() * 4, hipMemcpyDeviceToHost);
    }
    mcrt_scene_destroy(s);
    if (e != hipSuccess) return hip_fail(e, "render_rect");
    if (rc != MCRT_OK) return rc;
    for (int ly = 0; ly < tile->height; ++ly)
        std::memcpy(frame_rgba + 4 * (static_cast<size_t>(tile->y + ly) * cfg->width + tile->x),
                    host.data() + row_floats * static_cast<size_t>(ly) + 4 * static_cast<size_t>(tile->x), static_cast<size_t>(tile->width) * 16);
    return MCRT_OK;
}

int mcrt_render_tile(const mcrt_scene_desc* desc, const mcrt_config* cfg, int tile_index, float* frame_rgba, int device) {
    if (!desc || !cfg || !frame_rgba) return fail(MCRT_ERR_INVALID, "NULL argument");
    if (!valid_frame(cfg)) return MCRT_OK;
    Shard all = make_shard(*cfg, 0, 1);
    if (tile_index < 0 || tile_index >= all.tiles_x * all.tiles_y) return fail(MCRT_ERR_INVALID, "tile index out of range");
    const int row = tile_index / all.tiles_x, col = tile_index % all.tiles_x;
    mcrt_tile t;
    t.x = col * cfg->tile_size, t.y = row * cfg->tile_size;
    t.width = std::min(cfg->tile_size, cfg->width - t.x), t.height = std::min(cfg->tile_size, cfg->height - t.y);
    return mcrt_render_rect(desc, cfg, &t, frame_rgba, device);
}

// ---- PNG hand-off: the encoder and the file writers live in png_writer.cpp -------------------------
int mcrt_render_png(const mcrt_scene_desc* desc, const mcrt_config* cfg, const char* path, int device) {
    if (!desc || !cfg || !path) return fail(MCRT_ERR_INVALID, "NULL argument");
    if (!valid_frame(cfg)) return fail(MCRT_ERR_INVALID, "empty image");
    const size_t npix = static_cast<size_t>(cfg->width) * cfg->height;
    std::vector<uint8_t> host(npix * 4);
    // the RGBA8 plane straight from the kernels' epilogue: 4 B per pixel over PCIe (and xGMI) instead of 16
    const int rc = device == MCRT_DEVICE_ALL ? mcrt_render_rgba8(desc, cfg, host.data(), nullptr, nullptr, nullptr, 0, 0)
                                             : mcrt_render_rgba8(desc, cfg, host.data(), nullptr, nullptr, &device, 1, 0);
    if (rc != MCRT_OK) return rc;
    return mcrt_write_png_rgba8(path, host.data(), cfg->width, cfg->height);
}

int mcrt_last_timings(mcrt_timings* out) {
    if (!out) return MCRT_ERR_INVALID;
    *out = g_timings;
    return MCRT_OK;
}

// ---- probes -----------------------------------------------------------------------------------
int mcrt_probe_intersect(mcrt_scene* s, const float* rays, int n, mcrt_hit* out) {
    if (!s || !rays || !out || n < 0) return fail(MCRT_ERR_INVALID, "bad argument");
    if (n == 0) return MCRT_OK;
    HIP_TRY(hipSetDevice(s->device));
    DeviceBuffer d_rays, d_out;
    HIP_TRY(d_rays.reserve(static_cast<size_t>(n) * 24));
    HIP_TRY(d_out.reserve(static_cast<size_t>(n) * sizeof(mcrt_hit)));
    HIP_TRY(hipMemcpy(d_rays.ptr, rays, static_cast<size_t>(n) * 24, hipMemcpyHostToDevice));
    hipError_t e = launch_probe_intersect(static_cast<const uint8_t*>(s->blob.ptr), static_cast<float*>(d_rays.ptr), n,
                                          static_cast<mcrt_hit*>(d_out.ptr), nullptr);
    if (e == hipSuccess) e = hipMemcpy(out, d_out.ptr, static_cast<size_t>(n) * sizeof(mcrt_hit), hipMemcpyDeviceToHost);
    d_rays.release();
    d_out.release();
    if (e != hipSuccess) return hip_fail(e, "probe_intersect");
    return MCRT_OK;
}

int mcrt_probe_trace(mcrt_scene* s, const mcrt_config* cfg, const float* rays, int n, int depth, float* out_rgba) {
    if (!s || !cfg || !rays || !out_rgba || n < 0) return fail(MCRT_ERR_INVALID, "bad argument");
    if (n == 0) return MCRT_OK;
    HIP_TRY(hipSetDevice(s->device));
    DeviceBuffer d_rays, d_out, d_rng, d_stack;
    HIP_TRY(d_rays.reserve(static_cast<size_t>(n) * 24));
    HIP_TRY(d_out.reserve(static_cast<size_t>(n) * 16));
    bool long_rng = (cfg->soft_shadows && 2 * cfg->shadow_samples > 227) || (cfg->ao_enabled && 2 * cfg->ao_samples > 227);
    if (long_rng) HIP_TRY(d_rng.reserve(static_cast<size_t>(n) * 624 * 4));
    if (cfg->max_bounces > 16) HIP_TRY(d_stack.reserve(static_cast<size_t>(n) * cfg->max_bounces * 16));
    HIP_TRY(hipMemcpy(d_rays.ptr, rays, static_cast<size_t>(n) * 24, hipMemcpyHostToDevice));
    hipError_t e = launch_probe_trace(static_cast<const uint8_t*>(s->blob.ptr), *cfg, static_cast<float*>(d_rays.ptr), n,
                                      depth, static_cast<float*>(d_out.ptr), static_cast<uint32_t*>(d_rng.ptr),
                                      static_cast<float*>(d_stack.ptr), nullptr);
    if (e == hipSuccess) e = hipMemcpy(out_rgba, d_out.ptr, static_cast<size_t>(n) * 16, hipMemcpyDeviceToHost);
    d_rays.release();
    d_out.release();
    d_rng.release();
    d_stack.release();
    if (e != hipSuccess) return hip_fail(e, "probe_trace");
    return MCRT_OK;
}

int mcrt_probe_mt_uniform(int device, const uint32_t* seeds, int n_seeds, int n_draws, float* out) {
    if (!seeds || !out || n_seeds < 0 || n_draws < 0) return fail(MCRT_ERR_INVALID, "bad argument");
    if (n_seeds == 0 || n_draws == 0) return MCRT_OK;
    if (mcrt_device_count() <= 0) return fail(MCRT_ERR_NO_DEVICE, "no HIP device");
    HIP_TRY(hipSetDevice(device));
    DeviceBuffer d_seeds, d_out, d_store;
    HIP_TRY(d_seeds.reserve(static_cast<size_t>(n_seeds) * 4));
    HIP_TRY(d_out.reserve(static_cast<size_t>(n_seeds) * n_draws * 4));
    if (n_draws > 227) HIP_TRY(d_store.reserve(static_cast<size_t>(n_seeds) * 624 * 4));
    HIP_TRY(hipMemcpy(d_seeds.ptr, seeds, static_cast<size_t>(n_seeds) * 4, hipMemcpyHostToDevice));
    hipError_t e = launch_probe_mt(static_cast<uint32_t*>(d_seeds.ptr), n_seeds, n_draws, static_cast<float*>(d_out.ptr),
                                   static_cast<uint32_t*>(d_store.ptr), nullptr);
    if (e == hipSuccess)
        e = hipMemcpy(out, d_out.ptr, static_cast<size_t>(n_seeds) * n_draws * 4, hipMemcpyDeviceToHost);
    d_seeds.release();
    d_out.release();
    d_store.release();
    if (e != hipSuccess) return hip_fail(e, "probe_mt");
    return MCRT_OK;
}

int mcrt_probe_detmath(int device, int op, const float* x, const float* y, size_t n, float* out) {
    if (!x || !out || op < 0 || op > 5 || (op == 2 && !y)) return fail(MCRT_ERR_INVALID, "bad argument");
    if (n == 0) return MCRT_OK;
    if (mcrt_device_count() <= 0) return fail(MCRT_ERR_NO_DEVICE, "no HIP device");
    HIP_TRY(hipSetDevice(device));
    DeviceBuffer dx, dy, dout;
    HIP_TRY(dx.reserve(n * 4));
    HIP_TRY(dout.reserve(n * 4));
    HIP_TRY(hipMemcpy(dx.ptr, x, n * 4, hipMemcpyHostToDevice));
    if (y) {
        HIP_TRY(dy.reserve(n * 4));
        HIP_TRY(hipMemcpy(dy.ptr, y, n * 4, hipMemcpyHostToDevice));
    }
    hipError_t e = launch_probe_detmath(op, static_cast<float*>(dx.ptr), static_cast<float*>(dy.ptr), n,
                                        static_cast<float*>(dout.ptr), nullptr);
    if (e == hipSuccess) e = hipMemcpy(out, dout.ptr, n * 4, hipMemcpyDeviceToHost);
    dx.release();
    dy.release();
    dout.release();
    if (e != hipSuccess) return hip_fail(e, "probe_detmath");
    return MCRT_OK;
}

int mcrt_probe_div_const(int device, uint32_t d_first, uint32_t d_count, int mode, uint64_t* mismatches, uint32_t* a_failing_divisor) {
    if (!mismatches || d_first == 0 || d_count == 0 || d_count > 65535u) return fail(MCRT_ERR_INVALID, "bad argument");
    if (mcrt_device_count() <= 0) return fail(MCRT_ERR_NO_DEVICE, "no HIP device");
    HIP_TRY(hipSetDevice(device));
    DeviceBuffer counts, rds;
    HIP_TRY(counts.reserve(16));
    HIP_TRY(rds.reserve(static_cast<size_t>(d_count) * 4));
    {  // the reciprocals as the host forms them for the render kernels (prepare(): 1.0f / float(width))
        std::vector<float> host(d_count);
        for (uint32_t i = 0; i < d_count; ++i) host[i] = 1.0f / static_cast<float>(d_first + i);
        HIP_TRY(hipMemcpy(rds.ptr, host.data(), host.size() * 4, hipMemcpyHostToDevice));
    }
    hipError_t e = hipMemset(counts.ptr, 0, 16);
    for (uint32_t off = 0; e == hipSuccess && off < d_count; off += 32) {  // ~12 G quotients per launch
        e = launch_probe_div_const(d_first + off, d_count - off < 32u ? d_count - off : 32u, mode, static_cast<const float*>(rds.ptr) + off,
                                   static_cast<unsigned long long*>(counts.ptr), nullptr);
        if (e == hipSuccess) e = hipDeviceSynchronize();
    }
    rds.release();
    unsigned long long host[2] = {0, 0};
    if (e == hipSuccess) e = hipMemcpy(host, counts.ptr, 16, hipMemcpyDeviceToHost);
    counts.release();
    if (e != hipSuccess) return hip_fail(e, "probe_div_const");
    *mismatches = host[0];
    if (a_failing_divisor) *a_failing_divisor = static_cast<uint32_t>(host[1]);
    return MCRT_OK;
}

int mcrt_probe_detmath_range(int device, int op, uint32_t lo_bits, uint32_t hi_bits, float y0, uint64_t* mismatches) {
    if (!mismatches || op < 0 || op > 5 || hi_bits < lo_bits) return fail(MCRT_ERR_INVALID, "bad argument");
    if (mcrt_device_count() <= 0) return fail(MCRT_ERR_NO_DEVICE, "no HIP device");
    HIP_TRY(hipSetDevice(device));
    const uint64_t total = static_cast<uint64_t>(hi_bits) - lo_bits + 1;
    const uint64_t chunk = 1ull << 26;  // 64 Mi values = 256 MiB per pass
    DeviceBuffer dout;
    HIP_TRY(dout.reserve(chunk * 4));
    std::vector<float> host(chunk);
    unsigned nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 4;
    if (nt > 32) nt = 32;
    uint64_t bad = 0;
    for (uint64_t off = 0; off < total; off += chunk) {
        uint64_t cnt = total - off < chunk ? total - off : chunk;
        uint32_t base = lo_bits + static_cast<uint32_t>(off);
        hipError_t e = launch_probe_detmath_range(op, base, cnt, y0, static_cast<float*>(dout.ptr), nullptr);
        if (e == hipSuccess) e = hipMemcpy(host.data(), dout.ptr, cnt * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) {
            dout.release();
            return hip_fail(e, "probe_detmath_range");
        }
        std::atomic<uint64_t> part{0};
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < nt; ++t)
            pool.emplace_back([&, t] {
                uint64_t b = 0;
                for (uint64_t i = t; i < cnt; i += nt) {
                    float x = mcrt_u2f(base + static_cast<uint32_t>(i));
                    // ops 3/4 (device: the fused mcrt_sincosf) are held against the separate functions
                    float ref = op == 5 ? 1.0f / x
                                        : ((op == 0 || op == 3) ? mcrt_sinf(x) : ((op == 1 || op == 4) ? mcrt_cosf(x) : mcrt_powf(x, y0)));
                    uint32_t a = mcrt_f2u(ref), d = mcrt_f2u(host[i]);
                    if (a != d && !(std::isnan(ref) && std::isnan(host[i]))) ++b;
                }
                part += b;
            });
        for (auto& th : pool) th.join();
        bad += part.load();
    }
    dout.release();
    *mismatches = bad;
    return MCRT_OK;
}

}  // extern "C"
