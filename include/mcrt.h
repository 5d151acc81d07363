/* mcrt.h — C ABI of the MI355X-native tile-render hot path.
 *
 * This is the drop-in boundary for the reference's
 *     TileRenderer::render(const Scene&, const RayTracer::Config&, std::function<void(int,int)>)
 *     (/root/reference/src/raytracer/tile_renderer.h:26-28, tile_renderer.cpp:129-189)
 * and everything below it (renderTile → RayTracer::traceRay → intersectScene / shade).
 * Plain C types only: pointers + sizes, caller-owned buffers, no STL, no torch types.
 * The reference-side binding (how `Scene` is turned into `mcrt_scene_desc`) is shown in
 * INTEGRATION.md and shipped as minecraftskin_raytracer_amd/csrc/host/tile_renderer_hip.cpp.
 *
 * Three libraries implement (parts of) this ABI with different prefixes:
 *   libmcrt.so        (product, HIP/gfx950)      mcrt_*        — this header
 *   libmcrt_oracle.so (tests only, CPU)          mcrt_oracle_* — oracle/mcrt_oracle.h
 *   libmcref.so       (tests only, the compiled reference itself, built from /root/reference
 *                      where it lies)            mcref_*       — oracle/ref_shim.cpp
 */
#ifndef MCRT_H
#define MCRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCRT_ABI_VERSION 3 /* 2: mcrt_render_multi, MCRT_DEVICE_ALL; mcrt_time_render_device lost its second output
                            * 3: mcrt_render_rgba8, mcrt_render_rect */

/* error codes (0 = ok) */
#define MCRT_OK 0
#define MCRT_ERR_INVALID 1   /* bad argument / malformed scene description */
#define MCRT_ERR_NO_DEVICE 2 /* no HIP device, or HIP runtime failure */
#define MCRT_ERR_HIP 3       /* a HIP call failed; see mcrt_last_error() */
#define MCRT_ERR_NOMEM 4

/* ---- RayTracer::Config  (/root/reference/src/raytracer/raytracer.h:10-38) ---------------- */
typedef struct mcrt_config {
    int32_t width;             /* 256 */
    int32_t height;            /* 256 */
    int32_t max_bounces;       /* 3 */
    int32_t samples_per_pixel; /* 1 */
    int32_t tile_size;         /* 32; semantically significant: seeds the per-tile RNG */
    int32_t thread_count;      /* 0 = auto; accepted and ignored by the GPU path */
    int32_t soft_shadows;      /* bool, true */
    int32_t shadow_samples;    /* 8 */
    int32_t ao_enabled;        /* bool, false */
    int32_t ao_samples;        /* 8 */
    float ao_radius;           /* 3 */
    float ao_intensity;        /* .5 */
    int32_t dof_enabled;       /* bool, false */
    float aperture;            /* .5 */
    float focus_distance;      /* 0 = auto */
    int32_t gradient_bg;       /* bool, true */
    float gradient_scale;      /* 1 */
    float bg_center[4];        /* .91 .89 .86 1 */
    float bg_edge[4];          /* .56 .63 .71 1 */
} mcrt_config;

/* fills the struct with the in-class defaults of RayTracer::Config */
void mcrt_config_init(mcrt_config* cfg);

/* ---- Scene description: the reference's data model as POD -------------------------------
 * Scene/Light/Camera  /root/reference/src/scene/scene.h:10-34
 * Mesh                /root/reference/src/scene/mesh.h:12-27
 * Triangle            /root/reference/src/scene/triangle.h:9-16  (only v0..v2 and `texture`
 *                     are read by the ray tracer)
 * TextureRegion       /root/reference/src/skin/texture_region.h:8-27 */
typedef struct mcrt_texture {
    int32_t width;
    int32_t height;
    int64_t n_pixels;  /* pixels.size(); 0 ⇒ sample() returns Color() = (0,0,0,1) */
    const float* rgba; /* n_pixels * 4, row-major */
} mcrt_texture;

typedef struct mcrt_mesh {
    int32_t n_triangles;             /* mesh.triangles.size() (12 for a box) */
    const float* tri_vertices;       /* n_triangles * 9: v0.xyz v1.xyz v2.xyz */
    const int32_t* tri_texture;      /* n_triangles: index into scene textures, -1 = nullptr */
    int32_t n_local_triangles;       /* mesh.localTriangles.size() */
    const float* local_tri_vertices; /* n_local_triangles * 9 (unrotated box) */
    int32_t is_outer_layer;
    int32_t has_rotation;
    float pivot[3];
    float rot_x; /* degrees */
    float rot_z; /* degrees */
} mcrt_mesh;

typedef struct mcrt_scene_desc {
    int32_t n_meshes;
    const mcrt_mesh* meshes;
    int32_t n_textures;
    const mcrt_texture* textures;
    float light_position[3];
    float light_color[4];
    float light_intensity; /* unused by shade(), carried for completeness */
    float light_radius;
    float camera_position[3];
    float camera_target[3];
    float camera_up[3];
    float camera_fov; /* degrees */
    float background_color[4];
} mcrt_scene_desc;

/* ---- Tile  (/root/reference/src/raytracer/tile_renderer.h:11-14) -------------------------- */
typedef struct mcrt_tile {
    int32_t x, y, width, height;
} mcrt_tile;

/* TileRenderer::generateTiles (tile_renderer.cpp:18-39): row-major grid, edge tiles clipped,
 * zero tiles if any argument <= 0.  Returns the tile count; writes min(count, capacity) tiles
 * (tiles may be NULL to query the count). */
int mcrt_generate_tiles(int image_width, int image_height, int tile_size, mcrt_tile* tiles,
                        int capacity);

/* ---- library state ------------------------------------------------------------------------ */
int mcrt_abi_version(void);
/* number of HIP devices visible (0 when there is none; never fails) */
int mcrt_device_count(void);
/* message of the last failing call on this thread ("" if none) */
const char* mcrt_last_error(void);

/* ---- render: host buffers (the TileRenderer::render drop-in) ------------------------------
 * Renders the whole frame on `device` (>= 0; MCRT_DEVICE_ALL = every visible device, see
 * mcrt_render_multi) and copies it into out_rgba (width*height*4 floats, row-major, caller-owned).
 * Tile rows travel to the host as soon as they are final, while the rest of the frame still renders:
 * rows that hold only background right behind the first kernel, the rows of a pass behind that pass
 * when a large frame takes several.  `progress`, if non-NULL, is invoked exactly totalTiles times
 * with done = 1..total on the calling thread (tile_renderer.cpp:168-172 contract), for each row group
 * as it has landed in out_rgba.
 * Invalid sizes (any of width/height/tile_size <= 0) → MCRT_OK with nothing written, like the
 * reference returning an untouched Image (tile_renderer.cpp:144-146).  max_bounces above 4000 →
 * MCRT_ERR_INVALID (the workspace holds one colour per level and sample).
 * There is no CPU fallback: without a usable HIP device this returns MCRT_ERR_NO_DEVICE. */
typedef void (*mcrt_progress_fn)(int done, int total, void* user);
#define MCRT_DEVICE_ALL (-1)
int mcrt_render(const mcrt_scene_desc* scene, const mcrt_config* cfg, float* out_rgba,
                mcrt_progress_fn progress, void* user, int device);

/* The same frame spread over several devices of the node, one process: rank r of
 * n_devices renders tile rows r, r+n, r+2n, ... (cyclic — the figure occupies the middle rows) on
 * devices[r] with a replica of the scene; no collective inside the render.  devices = NULL or
 * n_devices <= 0: every visible device.  A device may be listed more than once (each entry is a rank
 * with its own workspace), which is how the path is tested on a one-GPU box.
 *   gather = 0  every device downloads its own rows straight into out_rgba: n PCIe links in parallel,
 *               no device-to-device traffic; the frame is assembled by the copies themselves
 *   gather = 1  the ranks' packed rows travel to devices[0] by peer copies (xGMI), one launch
 *               un-permutes them there (mcrt_assemble_frame_device), one download brings the frame back
 * Results are bit-identical to mcrt_render on one device.  Progress and errors as for mcrt_render. */
int mcrt_render_multi(const mcrt_scene_desc* scene, const mcrt_config* cfg, float* out_rgba,
                      mcrt_progress_fn progress, void* user, const int* devices, int n_devices, int gather);

/* The same render delivering the RGBA8 plane — `(uint8_t)(clamp(c,0,1)*255.0f+0.5f)` per channel, quantised in the
 * kernels' epilogue exactly like ImageWriter::writePNG / Image::toRGBA8 (image_writer.cpp:18-22, image.cpp:31-36) —
 * into out_rgba8 (width*height*4 bytes): 4 B per pixel on every link (PCIe, and xGMI when gather = 1) instead of 16.
 * devices / n_devices / gather as for mcrt_render_multi (one device: pass its index, n_devices = 1). */
int mcrt_render_rgba8(const mcrt_scene_desc* scene, const mcrt_config* cfg, uint8_t* out_rgba8, mcrt_progress_fn progress,
                      void* user, const int* devices, int n_devices, int gather);

/* TileRenderer::renderTile (tile_renderer.cpp:71-127): renders the one tile with row-major index
 * tile_index (generateTiles order) and writes its pixels into frame_rgba, a full width*height
 * float4 frame owned by the caller; all other pixels are left untouched. */
int mcrt_render_tile(const mcrt_scene_desc* scene, const mcrt_config* cfg, int tile_index,
                     float* frame_rgba, int device);
/* The same for an ARBITRARY Tile {x, y, width, height} of the frame, as the reference's renderTile accepts one: the
 * rectangle's own mt19937(tile.y * width + tile.x), its pixels in the rectangle's row-major order, cfg->tile_size not
 * read (tile_renderer.cpp:71-127).  An empty rectangle renders nothing; one that reaches outside the frame (the
 * reference would write out of bounds) → MCRT_ERR_INVALID.  A rectangle is one tile on the device — one RNG stream —
 * so a very large one renders correctly but far slower than mcrt_render of the same pixels. */
int mcrt_render_rect(const mcrt_scene_desc* scene, const mcrt_config* cfg, const mcrt_tile* tile, float* frame_rgba, int device);

/* ---- render: resident scene, device buffers (bench / multi-GPU path) --------------------- */
typedef struct mcrt_scene mcrt_scene; /* flattened scene resident in HBM on one device */

int mcrt_scene_create(const mcrt_scene_desc* desc, int device, mcrt_scene** out);
void mcrt_scene_destroy(mcrt_scene* scene);

/* out_layout */
#define MCRT_LAYOUT_FRAME 0  /* d_out is the full W*H float4 frame; only owned rows are written */
#define MCRT_LAYOUT_PACKED 1 /* d_out holds only the owned tile rows, packed in order */

/* Renders tile rows first, first+step, first+2*step, ... (a tile row = tile_size pixel rows) of
 * the frame into device memory on `stream` (a hipStream_t, NULL = default stream).  Asynchronous:
 * returns after enqueueing.  tile_row_first=0, tile_row_step=1 renders everything.
 * Sharding for N GPUs: rank r uses (first=r, step=N) — disjoint tile rows, no data-path
 * collective inside the render. */
int mcrt_render_device(mcrt_scene* scene, const mcrt_config* cfg, int tile_row_first,
                       int tile_row_step, int out_layout, float* d_out_rgba, void* stream);

/* mcrt_scene_destroy keeps the scene's device workspace (up to MCRT_POOL_MB, default 49152 MiB; one
 * idle set per device) for the next mcrt_scene_create / one-shot render on that device, because
 * allocating it dominates a single small render.  mcrt_trim() frees what is being kept. */
void mcrt_trim(void);
/* The library also keeps, per device that has rendered, a 128 MiB table of std::mt19937 seeding results
 * (state word 397 for the seeds -2^24 .. 2^24-1, which is where the per-hit shadow seeds of a scene at the
 * reference's scale lie): built once in ~2 ms, it replaces a 397-step recurrence per hit by one load, with
 * identical results.  MCRT_SEED_TABLE=0 turns it off; mcrt_trim() frees it when no scene handle is left. */

/* Waits for the scene's device work and reports an internal inconsistency of the last renders (the
 * workspace is sized for the tiles the host expects meshes to touch; the device flags a tile beyond
 * that bound instead of writing past it).  MCRT_OK in every correct run; the one-shot entry points
 * call it themselves. */
int mcrt_scene_check(mcrt_scene* scene);

/* A render is spread over internal *lanes* (streams with their own workspace, every n-th tile row of
 * the shard each, forked from / joined to the caller's stream) when the shard is large enough.
 * lanes = 0 restores that automatic choice, lanes >= 1 forces a count (at most 4).
 * One scene handle = one frame in flight: every render of a handle uses the handle's workspace, so
 * renders of one handle run one after the other on the device whatever streams they are given (the
 * library chains them with an event).  A caller that wants several frames in flight creates one
 * handle per frame in flight — and should then use lanes = 1, its frames already fill the chip.
 * Streams of the HIP runtime share its hardware queues (4 by default): more than 4 streams that
 * should run concurrently need GPU_MAX_HW_QUEUES set before the runtime initialises (bench.py: 8). */
int mcrt_scene_set_lanes(mcrt_scene* scene, int lanes);

/* Same render with the quantisation fused into the epilogue: writes the float4 frame to d_out_f32
 * and/or `(uint8_t)(clamp(c,0,1)*255.0f+0.5f)` per channel to d_out_rgba8 (either may be NULL, not
 * both) — the RGBA8 plane is what ImageWriter::writePNG hands to the PNG encoder
 * (/root/reference/src/output/image_writer.cpp:16-26), 4 B/pixel to copy back instead of 16. */
int mcrt_render_device_ex(mcrt_scene* scene, const mcrt_config* cfg, int tile_row_first, int tile_row_step,
                          int layout, float* d_out_f32, uint8_t* d_out_rgba8, void* stream);

/* number of pixel rows owned by (first, step) and therefore the packed buffer height */
int mcrt_owned_pixel_rows(const mcrt_config* cfg, int tile_row_first, int tile_row_step);

/* Scatter one rank's packed rows (as produced with MCRT_LAYOUT_PACKED) into a full frame.
 * Used by the gather root after the RCCL gather. */
int mcrt_unpack_rows_device(const mcrt_config* cfg, int tile_row_first, int tile_row_step,
                            const float* d_packed, float* d_frame, void* stream);

/* The same for all ranks of a gather in one launch: d_gathered holds `world` packed buffers (rank r
 * rendered with first=r, step=world), rank_stride_pixels float4 apart (>= the padded packed size
 * ceil(tile_rows/world) * tile_size * width). */
int mcrt_assemble_frame_device(const mcrt_config* cfg, int world, const float* d_gathered, size_t rank_stride_pixels,
                               float* d_frame, void* stream);

/* float RGBA → RGBA8, `(uint8_t)(clamp(c,0,1)*255.0f+0.5f)` per channel
 * (/root/reference/src/output/image_writer.cpp:18-22 ≡ src/skin/image.cpp:31-36). */
int mcrt_quantize_rgba8_device(const float* d_rgba, uint8_t* d_out, size_t n_pixels, void* stream);
void mcrt_quantize_rgba8(const float* rgba, uint8_t* out, size_t n_pixels);

/* ---- PNG hand-off (the step after the path: ImageWriter::writePNG, image_writer.cpp:6-28) -------- */
/* Writes an 8-bit RGBA PNG (colour type 6, no interlace, filter 0, zlib *stored* blocks: no
 * compression, bounded by memory bandwidth).  Any PNG reader decodes it to the same pixels the
 * reference's stbi_write_png output decodes to.  Returns MCRT_OK or MCRT_ERR_INVALID (bad
 * arguments, or the file cannot be created / written — writePNG's `false`). */
int mcrt_write_png_rgba8(const char* path, const uint8_t* rgba, int width, int height);
/* In-memory form: returns the PNG size; writes it when `capacity` suffices. */
size_t mcrt_encode_png_rgba8(const uint8_t* rgba, int width, int height, uint8_t* out, size_t capacity);
/* Quantise a float RGBA image exactly like ImageWriter::writePNG and write it. */
int mcrt_write_png_f32(const char* path, const float* rgba, int width, int height);
/* TileRenderer::render + ImageWriter::writePNG in one call: render on `device` (an index, or MCRT_DEVICE_ALL), quantise
 * in the kernel epilogue, copy 4 B/pixel back (mcrt_render_rgba8), write the file.  Invalid frame sizes write nothing and
 * return MCRT_ERR_INVALID (writePNG rejects empty images, image_writer.cpp:7-9). */
int mcrt_render_png(const mcrt_scene_desc* scene, const mcrt_config* cfg, const char* path, int device);

/* timings of the last mcrt_render() / mcrt_render_multi() on this thread, milliseconds: flatten_ms — scene
 * flattening on the host; h2d_ms — uploads, workspace checks and every launch call; kernel_ms — rank 0's
 * pipeline on its device (hipEvents); d2h_ms — from the last launch call until the last row has landed
 * (it overlaps the render); total_ms — the whole call */
typedef struct mcrt_timings {
    float flatten_ms, h2d_ms, kernel_ms, d2h_ms, total_ms;
} mcrt_timings;
int mcrt_last_timings(mcrt_timings* out);

/* Device-only timing helper used by bench.py: enqueues `iters` renders of the given shard on
 * `stream`, each bracketed by hipEvents recorded on that same stream and waited for, and returns
 * the average duration in ms of one render's whole pipeline (every lane and kernel of the frame). */
int mcrt_time_render_device(mcrt_scene* scene, const mcrt_config* cfg, int tile_row_first,
                            int tile_row_step, int out_layout, float* d_out_rgba, void* stream,
                            int iters, float* avg_render_ms);

/* ---- scene construction helpers (MeshBuilder / SkinParser layout) -------
 * Build the reference's character scene from an RGBA8 skin image (64x64 or 64x32), exactly as
 * SkinParser::parse (skin_parser.cpp:11-132, texel = u8/255.0f per image.cpp:16-21) followed by
 * MeshBuilder::buildScene (mesh_builder.cpp:145-202) would.  pose = 12 floats:
 * {head, body, rightArm, leftArm, rightLeg, leftLeg} x {rotX, rotZ} degrees (pose.h:9-22).
 * The returned description owns its arrays; free with mcrt_scene_desc_free(). */
int mcrt_build_skin_scene(const uint8_t* skin_rgba8, int skin_width, int skin_height,
                          const float pose[12], mcrt_scene_desc** out);
/* MeshBuilder::buildDefaultScene (mesh_builder.cpp:204-223): white 1x1 textures, no outer layer */
int mcrt_build_default_scene(const float pose[12], mcrt_scene_desc** out);
/* one of the 7 built-in poses of pose.h:25-92 (index 0..6) → 12 floats; returns MCRT_ERR_INVALID
 * for other indices */
int mcrt_builtin_pose(int index, float pose_out[12]);
void mcrt_scene_desc_free(mcrt_scene_desc* desc);

/* ---- flattened blob (what actually travels to HBM) — exposed for host-only tests ---------- */
/* Serialises the flattened scene (per-mesh AABB, rotation trig, face texture table, texel pool,
 * camera basis, light) into a byte blob.  Returns the blob size; copies min(size, capacity). */
size_t mcrt_scene_flatten(const mcrt_scene_desc* desc, void* blob, size_t capacity);

/* ---- per-function device probes (GPU parity tests mirror the reference's unit tests) ------ */
typedef struct mcrt_hit {
    int32_t hit;
    float t;
    float point[3];
    float normal[3];
    float texture_color[4];
    int32_t is_outer_layer;
} mcrt_hit;

/* intersectScene (intersection.cpp:408-421) for n rays; rays = n*6 floats (origin, direction) */
int mcrt_probe_intersect(mcrt_scene* scene, const float* rays, int n, mcrt_hit* out);
/* RayTracer::traceRay(ray, scene, depth, maxBounces, ShadingParams{}, &cfg) (raytracer.cpp:82-148)
 * for n rays; out = n*4 floats */
int mcrt_probe_trace(mcrt_scene* scene, const mcrt_config* cfg, const float* rays, int n, int depth,
                     float* out_rgba);
/* first n outputs of uniform_real_distribution<float>(0,1) over std::mt19937(seed) for each seed */
int mcrt_probe_mt_uniform(int device, const uint32_t* seeds, int n_seeds, int n_draws, float* out);
/* device detmath: op 0 = sinf, 1 = cosf, 2 = powf(x, y), 3 / 4 = sin / cos output of the fused
 * mcrt_sincosf, 5 = the kernels' 3-instruction reciprocal (reference: IEEE 1.0f / x), over n inputs */
int mcrt_probe_detmath(int device, int op, const float* x, const float* y, size_t n, float* out);
/* device detmath range check against the host build of the same header:
 * op 0/1: all floats with bit patterns in [lo_bits, hi_bits]; op 2: powf(x, y0).
 * Returns the number of mismatching inputs in *mismatches (host side is multi-threaded). */
int mcrt_probe_detmath_range(int device, int op, uint32_t lo_bits, uint32_t hi_bits, float y0,
                             uint64_t* mismatches);
/* div_frame — the frame-constant division of the sample coordinates (rt_core.h) — against the general division on the
 * device, for the integer divisors d_first .. d_first + d_count - 1 and every float a sample coordinate can take
 * (0 and 2^-33 .. d + 1); mode 1 checks the form with a second correction, mode 2 the uncorrected product (the probe's
 * own check); mode 3: rt::sqrt_pos against sqrtf for 0 and every float from 2^-96 to infinity; mode 4: the device's 1.0f / d
 * against the host's (the reciprocals the probe — like the render kernels — uses are the HOST's).  tools/gpu_verify_div.py */
int mcrt_probe_div_const(int device, uint32_t d_first, uint32_t d_count, int mode, uint64_t* mismatches, uint32_t* a_failing_divisor);

#ifdef __cplusplus
}
#endif
#endif /* MCRT_H */
