// flatten.cpp — mcrt_scene_desc (reference-shaped) → flat HBM blob (flat_scene.h).  Host only.
//
// Each value is produced by the same float32 expression the reference evaluates per ray, so that
// hoisting it here cannot change a bit of the output:
//   AABB                — computeAABB, /root/reference/src/raytracer/intersection.cpp:45-64
//   rotation cos/sin    — rotatePoint, intersection.cpp:12-37 (per-angle gates at :16 and :26)
//   camera basis, halfH — Camera::generateRay, /root/reference/src/scene/camera.cpp:10-16
//   face → texture      — determineFace, intersection.cpp:124-129
// Compile with -ffp-contract=off.
#include "flatten.h"

#include "mcrt_detmath.h"

#include <cfloat>
#include <cmath>
#include <limits>
#include <cstring>

namespace mcrt {
namespace {

struct V3 {
    float x, y, z;
};
inline V3 sub(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 cross(V3 a, V3 b) {
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline float length(V3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
inline V3 normalize(V3 a) {  // vec3.h:46-50
    float l = length(a);
    if (l < 1e-8f) return V3{0.0f, 0.0f, 0.0f};
    float inv = 1.0f / l;
    return V3{a.x * inv, a.y * inv, a.z * inv};
}
inline float selmin(float a, float b) { return (b < a) ? b : a; }  // std::min
inline float selmax(float a, float b) { return (a < b) ? b : a; }  // std::max

const float kPi = static_cast<float>(3.14159265358979323846);

void angle_trig(float deg, float& c, float& s) {
    float rad = deg * kPi / 180.0f;
    c = mcrt_cosf(rad);
    s = mcrt_sinf(rad);
}

// forward rotation of a local-space point (double precision; used only for the culling bound)
void rotate_fwd_d(const mcrt_mesh& m, const double in[3], double out[3]) {
    double p[3] = {in[0] - m.pivot[0], in[1] - m.pivot[1], in[2] - m.pivot[2]};
    const double d2r = 3.14159265358979323846 / 180.0;
    if (std::fabs(m.rot_x) > 0.01f) {
        double c = std::cos(m.rot_x * d2r), s = std::sin(m.rot_x * d2r);
        double ny = p[1] * c - p[2] * s, nz = p[1] * s + p[2] * c;
        p[1] = ny;
        p[2] = nz;
    }
    if (std::fabs(m.rot_z) > 0.01f) {
        double c = std::cos(m.rot_z * d2r), s = std::sin(m.rot_z * d2r);
        double nx = p[0] * c - p[1] * s, ny = p[0] * s + p[1] * c;
        p[0] = nx;
        p[1] = ny;
    }
    out[0] = p[0] + m.pivot[0];
    out[1] = p[1] + m.pivot[1];
    out[2] = p[2] + m.pivot[2];
}

}  // namespace

bool flatten_scene(const mcrt_scene_desc* d, std::vector<uint8_t>& blob, std::string& err) {
    if (!d) {
        err = "scene description is NULL";
        return false;
    }
    if (d->n_meshes < 0 || d->n_textures < 0 || (d->n_meshes > 0 && !d->meshes) ||
        (d->n_textures > 0 && !d->textures)) {
        err = "scene description has negative counts or NULL arrays";
        return false;
    }
    // ---- texel pool -------------------------------------------------------------------------
    std::vector<int64_t> tex_base(d->n_textures, MCRT_TEX_EMPTY);
    int64_t n_texels = 0;
    for (int i = 0; i < d->n_textures; ++i) {
        const mcrt_texture& t = d->textures[i];
        if (t.width <= 0 || t.height <= 0 || t.n_pixels <= 0) continue;  // sample() → Color()
        if (t.n_pixels < static_cast<int64_t>(t.width) * t.height || !t.rgba) {
            err = "texture " + std::to_string(i) + ": fewer pixels than width*height (out-of-bounds read in the reference)";
            return false;
        }
        tex_base[i] = n_texels;
        n_texels += static_cast<int64_t>(t.width) * t.height;
    }
    if (n_texels > 0x7fffffff / 4) {
        err = "texel pool too large";
        return false;
    }

    const size_t mesh_off = sizeof(FlatHeader);
    const size_t texel_off = mesh_off + sizeof(FlatMesh) * static_cast<size_t>(d->n_meshes);
    const size_t alpha_off = texel_off + static_cast<size_t>(n_texels) * 16;
    const size_t alpha_words = static_cast<size_t>((n_texels + 15) / 16);
    const size_t total = alpha_off + alpha_words * 4;
    blob.assign(total, 0);
    FlatHeader* h = reinterpret_cast<FlatHeader*>(blob.data());
    FlatMesh* fm = reinterpret_cast<FlatMesh*>(blob.data() + mesh_off);
    float* pool = reinterpret_cast<float*>(blob.data() + texel_off);

    for (int i = 0; i < d->n_textures; ++i) {
        if (tex_base[i] < 0) continue;
        const mcrt_texture& t = d->textures[i];
        std::memcpy(pool + 4 * tex_base[i], t.rgba, static_cast<size_t>(t.width) * t.height * 16);
    }
    // alpha predicates: the only two ways a texel's value decides hit/miss
    // (texColor.a == 0.0f, intersection.cpp:311; backTexColor.a > 0.0f, :349)
    uint32_t* abits = reinterpret_cast<uint32_t*>(blob.data() + alpha_off);
    for (int64_t i = 0; i < n_texels; ++i) {
        const float a = pool[4 * i + 3];
        uint32_t bits = (a == 0.0f ? 1u : 0u) | (a > 0.0f ? 2u : 0u);
        abits[i >> 4] |= bits << ((i & 15) * 2);
    }

    // ---- camera -----------------------------------------------------------------------------
    V3 pos{d->camera_position[0], d->camera_position[1], d->camera_position[2]};
    V3 tgt{d->camera_target[0], d->camera_target[1], d->camera_target[2]};
    V3 upv{d->camera_up[0], d->camera_up[1], d->camera_up[2]};
    V3 fwd = normalize(sub(tgt, pos));
    V3 right = normalize(cross(fwd, upv));
    V3 up = cross(right, fwd);
    float half_h = std::tan(d->camera_fov * 0.5f * kPi / 180.0f);

    h->magic = MCRT_FLAT_MAGIC;
    h->n_meshes = static_cast<uint32_t>(d->n_meshes);
    h->n_texels = static_cast<uint32_t>(n_texels);
    std::memcpy(h->light_pos, d->light_position, 12);
    h->light_radius = d->light_radius;
    std::memcpy(h->light_color, d->light_color, 16);
    std::memcpy(h->cam_pos, d->camera_position, 12);
    h->cam_half_h = half_h;
    h->cam_fwd[0] = fwd.x, h->cam_fwd[1] = fwd.y, h->cam_fwd[2] = fwd.z;
    h->cam_focus_auto = length(sub(tgt, pos));
    h->cam_right[0] = right.x, h->cam_right[1] = right.y, h->cam_right[2] = right.z;
    h->cam_up[0] = up.x, h->cam_up[1] = up.y, h->cam_up[2] = up.z;
    std::memcpy(h->background, d->background_color, 16);
    h->mesh_offset = static_cast<uint32_t>(mesh_off);
    h->texel_offset = static_cast<uint32_t>(texel_off);
    h->blob_bytes = static_cast<uint32_t>(total);
    h->alpha_offset = static_cast<uint32_t>(alpha_off);
    h->alpha_words = static_cast<uint32_t>(alpha_words);

    // ---- magnitude of the scene's coordinates → slack of everything that is merely conservative (flat_scene.h) ----
    double mag = 0.0;
    {
        auto take = [&](double v) { mag = (std::isfinite(v) ? std::fmax(mag, std::fabs(v)) : std::numeric_limits<double>::infinity()); };
        for (int k = 0; k < 3; ++k) take(d->camera_position[k]), take(d->light_position[k]);
        take(d->light_radius);
        for (int i = 0; i < d->n_meshes && std::isfinite(mag); ++i) {
            const mcrt_mesh& m = d->meshes[i];
            if (m.n_triangles > 0 && m.tri_vertices)
                for (int v = 0; v < m.n_triangles * 9; ++v) take(m.tri_vertices[v]);
            if (m.n_local_triangles > 0 && m.local_tri_vertices)
                for (int v = 0; v < m.n_local_triangles * 9; ++v) take(m.local_tri_vertices[v]);
            if (m.has_rotation)
                for (int k = 0; k < 3; ++k) take(m.pivot[k]);
        }
    }
    const double slack_d = static_cast<double>(kMaskSlack) * mag;
    h->mask_slack = std::isfinite(slack_d) ? static_cast<float>(slack_d) : std::numeric_limits<float>::infinity();
    const double slack = std::isfinite(slack_d) ? slack_d : 0.0;  // (non-finite scenes get no bounds at all below)

    // Culling is only offered for a well-conditioned pinhole camera: orthonormal basis, finite
    // positive tan(fov/2).  Anything else renders every mesh for every primary ray.
    auto finite3 = [](V3 v) { return std::isfinite(v.x) && std::isfinite(v.y) && std::isfinite(v.z); };
    bool cull_ok = finite3(pos) && finite3(fwd) && finite3(right) && finite3(up) &&
                   std::fabs(length(fwd) - 1.0f) < 1e-3f && std::fabs(length(right) - 1.0f) < 1e-3f &&
                   std::fabs(length(up) - 1.0f) < 1e-3f && std::isfinite(half_h) && half_h > 1e-4f &&
                   half_h < 1e4f && std::isfinite(slack_d);
    h->cull_ok = cull_ok ? 1u : 0u;

    // ---- meshes -----------------------------------------------------------------------------
    for (int i = 0; i < d->n_meshes; ++i) {
        const mcrt_mesh& m = d->meshes[i];
        FlatMesh& f = fm[i];
        if (m.n_triangles < 0 || m.n_local_triangles < 0 || (m.n_triangles > 0 && (!m.tri_vertices || !m.tri_texture)) ||
            (m.n_local_triangles > 0 && !m.local_tri_vertices)) {
            err = "mesh " + std::to_string(i) + ": negative counts or NULL arrays";
            return false;
        }
        const bool rotated = m.has_rotation != 0;
        const float* verts = rotated ? m.local_tri_vertices : m.tri_vertices;
        const int ntri = rotated ? m.n_local_triangles : m.n_triangles;
        f.flags = (m.is_outer_layer ? MESH_OUTER : 0u) | (rotated ? MESH_ROTATED : 0u) |
                  (ntri <= 0 ? MESH_EMPTY : 0u);
        float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        for (int v = 0; v < ntri * 3; ++v)
            for (int k = 0; k < 3; ++k) {
                lo[k] = selmin(lo[k], verts[3 * v + k]);
                hi[k] = selmax(hi[k], verts[3 * v + k]);
            }
        std::memcpy(f.lo, lo, 12);
        std::memcpy(f.hi, hi, 12);
        std::memcpy(f.pivot, m.pivot, 12);
        f.inv_z_cos = f.inv_x_cos = f.fwd_x_cos = f.fwd_z_cos = 1.0f;
        f.inv_z_sin = f.inv_x_sin = f.fwd_x_sin = f.fwd_z_sin = 0.0f;
        if (rotated) {
            if (std::fabs(m.rot_x) > 0.01f) {
                f.flags |= MESH_APPLY_X;
                angle_trig(-m.rot_x, f.inv_x_cos, f.inv_x_sin);
                angle_trig(m.rot_x, f.fwd_x_cos, f.fwd_x_sin);
            }
            if (std::fabs(m.rot_z) > 0.01f) {
                f.flags |= MESH_APPLY_Z;
                angle_trig(-m.rot_z, f.inv_z_cos, f.inv_z_sin);
                angle_trig(m.rot_z, f.fwd_z_cos, f.fwd_z_sin);
            }
        }
        for (int face = 0; face < 6; ++face) {
            int tri = face * 2;
            int ti = (tri < m.n_triangles) ? m.tri_texture[tri] : -1;
            f.tex_w[face] = f.tex_h[face] = 0;
            if (ti < 0) {
                f.tex_off[face] = MCRT_TEX_NULL;
            } else if (ti >= d->n_textures) {
                err = "mesh " + std::to_string(i) + ": texture index out of range";
                return false;
            } else if (tex_base[ti] < 0) {
                f.tex_off[face] = MCRT_TEX_EMPTY;
            } else {
                f.tex_off[face] = static_cast<int32_t>(tex_base[ti]);
                f.tex_w[face] = d->textures[ti].width;
                f.tex_h[face] = d->textures[ti].height;
            }
        }
        {  // MESH_OPAQUE: a null texture is opaque magenta, an empty one Color() with alpha 1
            bool opaque = true;
            for (int face = 0; face < 6 && opaque; ++face) {
                if (f.tex_off[face] < 0) continue;
                const int64_t n = static_cast<int64_t>(f.tex_w[face]) * f.tex_h[face];
                for (int64_t k = 0; k < n && opaque; ++k) opaque = !(pool[4 * (f.tex_off[face] + k) + 3] == 0.0f);
            }
            if (opaque) f.flags |= MESH_OPAQUE;
        }

        // Bounding sphere in world space: centre = (posed) box centre, radius = half diagonal, padded.
        {
            double c[3] = {0.5 * (static_cast<double>(lo[0]) + hi[0]), 0.5 * (static_cast<double>(lo[1]) + hi[1]),
                           0.5 * (static_cast<double>(lo[2]) + hi[2])};
            double w[3] = {c[0], c[1], c[2]};
            if (rotated) rotate_fwd_d(m, c, w);
            double dx = 0.5 * (static_cast<double>(hi[0]) - lo[0]), dy = 0.5 * (static_cast<double>(hi[1]) - lo[1]),
                   dz = 0.5 * (static_cast<double>(hi[2]) - lo[2]);
            double r = std::sqrt(dx * dx + dy * dy + dz * dz);
            f.sphere[0] = static_cast<float>(w[0]);
            f.sphere[1] = static_cast<float>(w[1]);
            f.sphere[2] = static_cast<float>(w[2]);
            // padded by 1 % and 25 slacks (0.05 at the reference's scale): the posed corners are formed in float by the kernels
            f.sphere[3] = (ntri > 0 && std::isfinite(r) && std::isfinite(slack_d)) ? static_cast<float>(r * 1.01 + 25.0 * slack) : -1.0f;  // < 0: no bound
        }

        // Conservative screen bound: project the 8 box corners (posed ones rotated forward) and
        // pad generously.  Only ever used to SKIP a mesh whose slab test would have missed, so it
        // must never be too small; any doubt → "never cull" (u0 > u1).
        f.screen[0] = 1.0f, f.screen[1] = 1.0f, f.screen[2] = 0.0f, f.screen[3] = 0.0f;
        bool bound_ok = cull_ok && ntri > 0;
        double u0 = 1e30, v0 = 1e30, u1 = -1e30, v1 = -1e30, z0 = 1e30, z1 = -1e30;
        for (int c = 0; c < 8 && bound_ok; ++c) {
            // the box inflated by one slack: a ray the float slab test lets graze the box still projects inside the bound
            double p[3] = {(c & 1) ? hi[0] + slack : lo[0] - slack, (c & 2) ? hi[1] + slack : lo[1] - slack, (c & 4) ? hi[2] + slack : lo[2] - slack};
            double w[3];
            if (rotated)
                rotate_fwd_d(m, p, w);
            else
                w[0] = p[0], w[1] = p[1], w[2] = p[2];
            double rel[3] = {w[0] - pos.x, w[1] - pos.y, w[2] - pos.z};
            double zc = rel[0] * fwd.x + rel[1] * fwd.y + rel[2] * fwd.z;
            double xc = rel[0] * right.x + rel[1] * right.y + rel[2] * right.z;
            double yc = rel[0] * up.x + rel[1] * up.y + rel[2] * up.z;
            double extent = std::fabs(rel[0]) + std::fabs(rel[1]) + std::fabs(rel[2]);
            if (!(zc > 1e-3 * (1.0 + extent)) || !std::isfinite(zc + xc + yc)) {
                bound_ok = false;  // corner at/behind the camera plane
                break;
            }
            double su = xc / zc, sv = yc / zc;  // image plane at distance 1: x in ±halfW, y in ±halfH
            // u = (su/halfW + 1)/2 needs the aspect ratio, which is a render-time value; store
            // su and sv scaled by 1/halfH only and let the kernel apply aspect.
            double nu = su / half_h, nv = sv / half_h;
            u0 = std::fmin(u0, nu), u1 = std::fmax(u1, nu);
            v0 = std::fmin(v0, nv), v1 = std::fmax(v1, nv);
            z0 = std::fmin(z0, zc), z1 = std::fmax(z1, zc);
        }
        if (bound_ok && std::isfinite(u0 + u1 + v0 + v1)) {
            // stored in "halfH units": x range [u0,u1] (divide by aspect to get [-1,1]), y range
            // [v0,v1] in [-1,1] with +y up.
            f.screen[0] = static_cast<float>(u0);
            f.screen[1] = static_cast<float>(v0);
            f.screen[2] = static_cast<float>(u1);
            f.screen[3] = static_cast<float>(v1);
            f.depth[0] = static_cast<float>(z0 * 0.99);  // z0 > 0: every corner is in front of the camera
            f.depth[1] = static_cast<float>(z1 * 1.01);
        }
    }

    // ---- first-pass groups --------------------------------------------------------------------
    // Mesh j joins root k when both are un-posed and non-empty and box k contains box j (float
    // compares on the stored bounds: the slab arithmetic is monotonic in the bounds, so a ray that
    // overlaps box j overlaps box k).  One level only: a root is never a member.  Larger boxes are
    // considered first so that nested triples attach to the outermost box.
    {
        const int n = d->n_meshes < 64 ? d->n_meshes : 64;
        auto contains = [&](int k, int j) {
            for (int a = 0; a < 3; ++a)
                if (!(fm[k].lo[a] <= fm[j].lo[a] && fm[k].hi[a] >= fm[j].hi[a])) return false;
            return true;
        };
        auto plain = [&](int i) { return (fm[i].flags & (MESH_ROTATED | MESH_EMPTY)) == 0; };
        std::vector<int> parent(n, -1);
        for (int j = 0; j < n; ++j) {
            if (!plain(j)) continue;
            for (int k = 0; k < n; ++k) {
                if (k == j || !plain(k) || !contains(k, j)) continue;
                if (contains(j, k) && k > j) continue;  // identical boxes: the lower index is the root
                if (parent[j] < 0 || contains(k, parent[j])) parent[j] = k;
            }
        }
        // resolve chains to the outermost ancestor (containment is transitive, so that is still valid)
        for (int j = 0; j < n; ++j) {
            int r = parent[j], guard = 0;
            while (r >= 0 && parent[r] >= 0 && guard++ < 64) r = parent[r];
            parent[j] = r;
        }
        uint64_t roots = 0;
        std::vector<uint64_t> group(n, 0);
        for (int j = 0; j < n; ++j) {
            const int r = parent[j] < 0 ? j : parent[j];
            group[r] |= 1ull << j;
            if (parent[j] < 0) roots |= 1ull << j;
        }
        for (int j = 0; j < n; ++j) {
            fm[j].group_lo = static_cast<uint32_t>(group[j]);
            fm[j].group_hi = static_cast<uint32_t>(group[j] >> 32);
        }
        h->root_lo = static_cast<uint32_t>(roots);
        h->root_hi = static_cast<uint32_t>(roots >> 32);
    }
    return true;
}

}  // namespace mcrt
