// flat_scene.h — the scene as it lives in HBM: one contiguous blob
//   [FlatHeader][FlatMesh x n_meshes][float4 texel pool][2-bit alpha predicates per texel]
// produced on the host by flatten.cpp from the reference-shaped mcrt_scene_desc and read by the
// kernels through wave-uniform (scalar) loads.  Everything a ray needs that the reference
// recomputes per ray from per-frame constants is evaluated once here, with the reference's own
// float expressions, so the values are bit-identical:
//   - per-mesh AABB            (intersection.cpp:45-64 recomputes it from 36 vertices per ray)
//   - posed-mesh cos/sin       (intersection.cpp:17-33 calls cosf/sinf per ray per mesh)
//   - camera basis and tan(fov) (camera.cpp:10-16 recomputes them per ray)
//   - face → texture table     (intersection.cpp:124-129: triangles[2*face].texture)
#ifndef MCRT_FLAT_SCENE_H
#define MCRT_FLAT_SCENE_H

#include <stdint.h>

#define MCRT_FLAT_MAGIC 0x4d435254u /* "MCRT" */

enum : uint32_t {
    MESH_OUTER = 1u,    // Mesh::isOuterLayer
    MESH_ROTATED = 2u,  // Mesh::hasRotation
    MESH_APPLY_X = 4u,  // |rotX| > 0.01  (rotatePoint gate, intersection.cpp:16)
    MESH_APPLY_Z = 8u,  // |rotZ| > 0.01  (intersection.cpp:26)
    MESH_EMPTY = 16u,   // tris.empty()   (intersection.cpp:205)
    MESH_OPAQUE = 32u,  // no texel of its six faces has alpha == 0.0f: every slab hit is a hit (:311 never takes the
                        // transparent branch) — shadow / AO rays skip the face, UV and texel work for such a mesh
};

// texture slot states for FlatMesh::tex_off
#define MCRT_TEX_NULL (-1)  /* Triangle::texture == nullptr → opaque magenta (intersection.cpp:305) */
#define MCRT_TEX_EMPTY (-2) /* width/height <= 0 or no pixels → Color() (texture_region.h:20-22) */

struct FlatMesh {  // 192 bytes
    float lo[3];
    float hi[3];
    float pivot[3];
    // inverse transform (world → local): undo Z by angle -rotZ, then undo X by angle -rotX
    float inv_z_cos, inv_z_sin;
    float inv_x_cos, inv_x_sin;
    // forward transform (local → world): rotX then rotZ
    float fwd_x_cos, fwd_x_sin;
    float fwd_z_cos, fwd_z_sin;
    uint32_t flags;
    // face slots in determineFace order: 0 back(-Z) 1 front(+Z) 2 left(+X) 3 right(-X) 4 top 5 bottom
    int32_t tex_off[6];  // offset into the texel pool, or MCRT_TEX_NULL / MCRT_TEX_EMPTY
    int32_t tex_w[6];
    int32_t tex_h[6];
    // conservative screen-space bound of the mesh for primary-ray culling, in pixels of a
    // normalised [0,1]x[0,1] image (u0,v0,u1,v1); u0 > u1 means "never cull"
    float screen[4];
    // world-space bounding sphere (centre, padded radius) — a conservative pre-test for posed meshes,
    // whose exact test needs the ray in the mesh's local frame
    float sphere[4];
    // first-pass grouping (meshes 0..63): a mesh whose box lies inside another un-posed mesh's box is
    // that mesh's member — the first pass tests only group roots and takes the members along.
    // group = bit i of every member incl. the root itself (0 for a non-root); see FlatHeader::root_mask
    uint32_t group_lo, group_hi;
    // camera-space depth range (along the camera's forward axis) of the box corners, valid when the
    // screen bound is: thin-lens rays shift a point's image by lens_offset * (1/z - 1/z_focus), so the
    // bound is dilated by that much when depth of field is on
    float depth[2];
};

struct FlatHeader {  // 192 bytes
    uint32_t magic;
    uint32_t n_meshes;
    uint32_t n_texels;
    uint32_t cull_ok;  // 1 when the camera basis is well-conditioned and FlatMesh::screen is valid
    float light_pos[3];
    float light_radius;
    float light_color[4];
    float cam_pos[3];
    float cam_half_h;  // tanf(fov * 0.5f * PI / 180)   (camera.cpp:15)
    float cam_fwd[3];
    float cam_focus_auto;  // |target - position|      (tile_renderer.cpp:82-85)
    float cam_right[3];
    // Slack of the conservative candidate masks (rt_core.h: bundle_classify, ball / hemisphere candidates) and of the
    // bounds that only ever SKIP work (bounding spheres, screen bounds), in world units: kMaskSlack x the largest
    // coordinate magnitude of the scene (box corners, pivots, light, camera) — ~330 ulp of that magnitude, the float
    // error of the per-ray arithmetic being a few ulp of it.  Scale-free: a scene scaled or moved far from the origin
    // keeps the same margin in ulps.  2e-3 for the reference's character scene (camera 50 away); +inf for a scene
    // with non-finite coordinates (every mesh is then a candidate of every query).
    float mask_slack;
    float cam_up[3];  // trueUp = right x forward
    float pad1;
    float background[4];
    uint32_t mesh_offset;   // byte offset of FlatMesh[0] from the blob start
    uint32_t texel_offset;  // byte offset of the float4 texel pool
    uint32_t blob_bytes;
    uint32_t alpha_offset;  // byte offset of the alpha-predicate words: 16 texels per uint32,
                            // bit 2k = (alpha == 0.0f), bit 2k+1 = (alpha > 0.0f)  (intersection.cpp:311,349)
    uint32_t alpha_words;
    uint32_t root_lo, root_hi;  // bit i: mesh i (< 64) is a group root (every mesh is in exactly one group)
    uint32_t pad2[9];
};

constexpr float kMaskSlack = 4e-5f;

static_assert(sizeof(FlatMesh) == 192, "FlatMesh layout");
static_assert(sizeof(FlatHeader) == 192, "FlatHeader layout");

#endif
