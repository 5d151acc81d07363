/* mcrt_oracle.h — C ABI of the CPU oracle (TEST INFRASTRUCTURE, not product code).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * It restates the reference's render hot path on the POD scene description of include/mcrt.h:
 * same algorithmic structure as the reference (per-ray AABB recompute, duplicate primary
 * intersection, full mt19937 seeding per hit, std::thread tile pool over an atomic tile queue),
 * float32 arithmetic in the reference's operation order.  Pinned against the compiled reference
 * (oracle/_ref, built by oracle/Makefile from /root/reference where it lies) and against the
 * committed fixtures in tests/golden/.
 */
#ifndef MCRT_ORACLE_H
#define MCRT_ORACLE_H

#include "mcrt.h"

#ifdef __cplusplus
extern "C" {
#endif

/* TileRenderer::render (tile_renderer.cpp:129-189).  threads <= 0 → hardware_concurrency().
 * progress may be NULL.  Returns 0. */
int mcrt_oracle_render(const mcrt_scene_desc* scene, const mcrt_config* cfg, float* out_rgba,
                       mcrt_progress_fn progress, void* user);
/* TileRenderer::renderTile (tile_renderer.cpp:71-127) for one tile into a full-frame buffer */
int mcrt_oracle_render_tile(const mcrt_scene_desc* scene, const mcrt_config* cfg,
                            const mcrt_tile* tile, float* frame_rgba);
/* tile rows row_first, row_first + row_step, ... of the frame on the same thread pool; only those rows of the
 * full-frame buffer are written.  Returns the number of tiles rendered. */
int mcrt_oracle_render_rows(const mcrt_scene_desc* scene, const mcrt_config* cfg, int row_first, int row_step,
                            float* out_rgba);
/* TileRenderer::generateTiles */
int mcrt_oracle_generate_tiles(int w, int h, int tile, mcrt_tile* tiles, int capacity);
/* intersectScene (intersection.cpp:408-421) */
int mcrt_oracle_intersect(const mcrt_scene_desc* scene, const float* rays, int n, mcrt_hit* out);
/* intersectMesh on one mesh (intersection.cpp:373-406) */
int mcrt_oracle_intersect_mesh(const mcrt_scene_desc* scene, int mesh_index, const float* rays,
                               int n, mcrt_hit* out);
/* RayTracer::traceRay; cfg may be NULL (= config nullptr in the reference) */
int mcrt_oracle_trace(const mcrt_scene_desc* scene, const mcrt_config* cfg, const float* rays,
                      int n, int depth, int max_bounces, float* out_rgba);
/* shade (shading.cpp:62-96) with ShadingParams{kd, ks, ambient, shininess} */
int mcrt_oracle_shade(const mcrt_scene_desc* scene, const mcrt_hit* hit, const float view_dir[3],
                      const float params[4], float shadow_factor, float out_rgba[4]);
/* isInShadow (shading.cpp:14-26) */
int mcrt_oracle_in_shadow(const mcrt_scene_desc* scene, const float point[3], const float normal[3],
                          const float light_pos[3]);
/* computeSoftShadow (shading.cpp:28-60) */
float mcrt_oracle_soft_shadow(const mcrt_scene_desc* scene, const float point[3],
                              const float normal[3], int samples, uint32_t seed);
/* RayTracer::computeAO (raytracer.cpp:38-78) */
float mcrt_oracle_ao(const mcrt_scene_desc* scene, const float point[3], const float normal[3],
                     int samples, float radius, uint32_t seed);
/* RayTracer::backgroundColor (raytracer.cpp:16-34); cfg may be NULL */
void mcrt_oracle_background(const mcrt_scene_desc* scene, const mcrt_config* cfg, float u, float v,
                            float out_rgba[4]);
/* Camera::generateRay (camera.cpp:8-26) → 6 floats */
void mcrt_oracle_camera_ray(const mcrt_scene_desc* scene, float u, float v, float aspect,
                            float out_ray[6]);
/* uniform_real_distribution<float>(0,1)(std::mt19937(seed)), own MT restatement */
void mcrt_oracle_mt_uniform(uint32_t seed, int n, float* out);
/* the same through libstdc++ <random>, to pin the restatement (random.tcc:3348-3380) */
void mcrt_oracle_mt_uniform_std(uint32_t seed, int n, float* out);
/* the reference's float→unsigned seed cast as x86-64 GCC compiles it (cvttss2si 64 → low 32) */
uint32_t mcrt_oracle_seed_cast(float f);
/* ImageWriter quantiser (image_writer.cpp:18-22) */
void mcrt_oracle_quantize(const float* rgba, uint8_t* out, size_t n_pixels);

#ifdef __cplusplus
}
#endif
#endif
