// scene_builder.cpp — SkinParser layout + MeshBuilder scene construction on the POD description.
// Host only; the "f-2" row of SURVEY.md §8: runs once per skin/pose change, also the generator of
// every synthetic scene used by tests and bench.py.
//
// Follows (paths under /root/reference/src):
//   skin/image.cpp:16-21          texel = u8 / 255.0f
//   skin/image.h:21-33            Image::extractRegion (out-of-range texels stay Color())
//   skin/skin_parser.cpp:11-20    box-unwrap layout of one body part
//   skin/skin_parser.cpp:45-110   64x64 and legacy 64x32 part origins, legacy left = mirrored right
//   scene/mesh_builder.cpp:66-143 buildBox / buildBoxWithPose (vertex order, face → texture)
//   scene/mesh_builder.cpp:145-223 buildScene / buildDefaultScene (part table, light, camera)
//   scene/pose.h:25-92            built-in poses
// Compile with -ffp-contract=off.
#include "mcrt.h"
#include "mcrt_detmath.h"

#include <cmath>
#include <cstring>
#include <memory>
#include <new>
#include <vector>

namespace {

struct Region {  // TextureRegion
    int w = 0, h = 0;
    std::vector<float> px;  // rgba
};
struct PartTex {  // BodyPartTexture
    Region top, bottom, front, back, left, right;
};

struct SkinImage {
    int w, h;
    std::vector<float> px;
};

Region cut(const SkinImage& img, int x, int y, int w, int h) {
    Region r;
    r.w = w;
    r.h = h;
    r.px.assign(static_cast<size_t>(w) * h * 4, 0.0f);
    for (size_t i = 0; i < static_cast<size_t>(w) * h; ++i) r.px[4 * i + 3] = 1.0f;  // Color()
    for (int row = 0; row < h; ++row)
        for (int col = 0; col < w; ++col) {
            int sx = x + col, sy = y + row;
            if (sx < 0 || sx >= img.w || sy < 0 || sy >= img.h) continue;
            std::memcpy(&r.px[4 * (static_cast<size_t>(row) * w + col)],
                        &img.px[4 * (static_cast<size_t>(sy) * img.w + sx)], 16);
        }
    return r;
}

PartTex unwrap(const SkinImage& img, int ox, int oy, int w, int h, int d) {
    PartTex p;
    p.top = cut(img, ox + d, oy, w, d);
    p.bottom = cut(img, ox + d + w, oy, w, d);
    p.left = cut(img, ox, oy + d, d, h);
    p.front = cut(img, ox + d, oy + d, w, h);
    p.right = cut(img, ox + d + w, oy + d, d, h);
    p.back = cut(img, ox + 2 * d + w, oy + d, w, h);
    return p;
}

Region flip_h(const Region& r) {
    Region m;
    m.w = r.w;
    m.h = r.h;
    m.px.resize(r.px.size());
    for (int y = 0; y < r.h; ++y)
        for (int x = 0; x < r.w; ++x)
            std::memcpy(&m.px[4 * (static_cast<size_t>(y) * r.w + x)],
                        &r.px[4 * (static_cast<size_t>(y) * r.w + (r.w - 1 - x))], 16);
    return m;
}

PartTex mirror_part(const PartTex& p) {
    PartTex m;
    m.top = flip_h(p.top);
    m.bottom = flip_h(p.bottom);
    m.front = flip_h(p.front);
    m.back = flip_h(p.back);
    m.left = flip_h(p.right);
    m.right = flip_h(p.left);
    return m;
}

bool region_clear(const Region& r) {
    for (size_t i = 0; i < r.px.size() / 4; ++i)
        if (r.px[4 * i + 3] != 0.0f) return false;
    return true;
}
bool fully_transparent(const PartTex& p) {
    return region_clear(p.top) && region_clear(p.bottom) && region_clear(p.front) &&
           region_clear(p.back) && region_clear(p.left) && region_clear(p.right);
}

struct Skin {
    PartTex inner[6];  // head, body, rightArm, leftArm, rightLeg, leftLeg
    PartTex outer[6];
};

// the description plus the storage it points into
struct OwnedDesc {
    mcrt_scene_desc desc;
    std::vector<mcrt_mesh> meshes;
    std::vector<mcrt_texture> textures;
    std::vector<std::unique_ptr<std::vector<float>>> floats;
    std::vector<std::unique_ptr<std::vector<int32_t>>> ints;
    const float* keep(std::vector<float> v) {
        floats.push_back(std::make_unique<std::vector<float>>(std::move(v)));
        return floats.back()->data();
    }
    const int32_t* keep(std::vector<int32_t> v) {
        ints.push_back(std::make_unique<std::vector<int32_t>>(std::move(v)));
        return ints.back()->data();
    }
};

const float kPi = static_cast<float>(3.14159265358979323846);

void rotate_about(float v[3], const float pivot[3], float degX, float degZ) {  // mesh_builder.cpp:25-52
    float p[3] = {v[0] - pivot[0], v[1] - pivot[1], v[2] - pivot[2]};
    if (std::fabs(degX) > 0.01f) {
        float rad = degX * kPi / 180.0f;
        float c = mcrt_cosf(rad), s = mcrt_sinf(rad);
        float ny = p[1] * c - p[2] * s;
        float nz = p[1] * s + p[2] * c;
        p[1] = ny;
        p[2] = nz;
    }
    if (std::fabs(degZ) > 0.01f) {
        float rad = degZ * kPi / 180.0f;
        float c = mcrt_cosf(rad), s = mcrt_sinf(rad);
        float nx = p[0] * c - p[1] * s;
        float ny = p[0] * s + p[1] * c;
        p[0] = nx;
        p[1] = ny;
    }
    v[0] = p[0] + pivot[0];
    v[1] = p[1] + pivot[1];
    v[2] = p[2] + pivot[2];
}

int add_texture(OwnedDesc& o, const Region& r) {
    mcrt_texture t;
    t.width = r.w;
    t.height = r.h;
    t.n_pixels = static_cast<int64_t>(r.px.size() / 4);
    t.rgba = o.keep(r.px);
    o.textures.push_back(t);
    return static_cast<int>(o.textures.size()) - 1;
}

// One box part (mesh_builder.cpp:66-143).  posed → keeps the unrotated triangles as
// localTriangles and rotates the world-space copy.
void add_box(OwnedDesc& o, const PartTex& tex, const float pos[3], const float size[3], float offset,
             bool posed, const float pivot[3], float rotX, float rotZ) {
    float hw = size[0] / 2.0f + offset;
    float hh = size[1] / 2.0f + offset;
    float hd = size[2] / 2.0f + offset;
    float x0 = pos[0] - hw, x1 = pos[0] + hw;
    float y0 = pos[1] - hh, y1 = pos[1] + hh;
    float z0 = pos[2] - hd, z1 = pos[2] + hd;
    const float c[8][3] = {{x0, y0, z0}, {x1, y0, z0}, {x0, y1, z0}, {x1, y1, z0},
                           {x0, y0, z1}, {x1, y0, z1}, {x0, y1, z1}, {x1, y1, z1}};
    // corner index = x + 2*y + 4*z; quads in addFace order: back, front, left, right, top, bottom
    static const int quad[6][4] = {{2, 3, 1, 0}, {7, 6, 4, 5}, {3, 7, 5, 1},
                                   {6, 2, 0, 4}, {6, 7, 3, 2}, {0, 1, 5, 4}};
    const Region* face_tex[6] = {&tex.back, &tex.front, &tex.left, &tex.right, &tex.top, &tex.bottom};

    std::vector<float> verts;
    std::vector<int32_t> tix;
    verts.reserve(12 * 9);
    for (int f = 0; f < 6; ++f) {
        int ti = add_texture(o, *face_tex[f]);
        const int* q = quad[f];
        const int tri[2][3] = {{q[0], q[1], q[2]}, {q[0], q[2], q[3]}};
        for (auto& t : tri) {
            for (int k = 0; k < 3; ++k) verts.insert(verts.end(), c[t[k]], c[t[k]] + 3);
            tix.push_back(ti);
        }
    }
    mcrt_mesh m;
    std::memset(&m, 0, sizeof m);
    m.n_triangles = 12;
    m.is_outer_layer = offset > 0.0f ? 1 : 0;
    if (posed) {
        m.n_local_triangles = 12;
        m.local_tri_vertices = o.keep(verts);
        m.has_rotation = 1;
        std::memcpy(m.pivot, pivot, 12);
        m.rot_x = rotX;
        m.rot_z = rotZ;
        if (!(std::fabs(rotX) < 0.01f && std::fabs(rotZ) < 0.01f))
            for (size_t v = 0; v < verts.size() / 3; ++v) rotate_about(&verts[3 * v], pivot, rotX, rotZ);
    }
    m.tri_vertices = o.keep(std::move(verts));
    m.tri_texture = o.keep(std::move(tix));
    o.meshes.push_back(m);
}

OwnedDesc* assemble(const Skin& skin, const float pose[12]) {  // mesh_builder.cpp:145-202
    auto* o = new OwnedDesc();
    static const float position[6][3] = {{0, 28, 0}, {0, 18, 0}, {-6, 18, 0}, {6, 18, 0}, {-2, 6, 0}, {2, 6, 0}};
    static const float size[6][3] = {{8, 8, 8}, {8, 12, 4}, {4, 12, 4}, {4, 12, 4}, {4, 12, 4}, {4, 12, 4}};
    static const float pivot[6][3] = {{0, 24, 0}, {0, 18, 0}, {-6, 24, 0}, {6, 24, 0}, {-2, 12, 0}, {2, 12, 0}};
    for (int p = 0; p < 6; ++p) {
        float rx = pose ? pose[2 * p] : 0.0f, rz = pose ? pose[2 * p + 1] : 0.0f;
        bool posed = std::fabs(rx) > 0.01f || std::fabs(rz) > 0.01f;
        add_box(*o, skin.inner[p], position[p], size[p], 0.0f, posed, pivot[p], rx, rz);
        if (!fully_transparent(skin.outer[p]))
            add_box(*o, skin.outer[p], position[p], size[p], 0.5f, posed, pivot[p], rx, rz);
    }
    mcrt_scene_desc& d = o->desc;
    std::memset(&d, 0, sizeof d);
    d.n_meshes = static_cast<int32_t>(o->meshes.size());
    d.meshes = o->meshes.data();
    d.n_textures = static_cast<int32_t>(o->textures.size());
    d.textures = o->textures.data();
    const float lp[3] = {0, 40, 30}, lc[4] = {1, 1, 1, 1};
    const float cp[3] = {0, 18, 50}, ct[3] = {0, 18, 0}, cu[3] = {0, 1, 0};
    const float bg[4] = {0.2f, 0.3f, 0.5f, 1.0f};
    std::memcpy(d.light_position, lp, 12);
    std::memcpy(d.light_color, lc, 16);
    d.light_intensity = 1.0f;
    d.light_radius = 3.0f;  // Light's in-class default (scene.h:14)
    std::memcpy(d.camera_position, cp, 12);
    std::memcpy(d.camera_target, ct, 12);
    std::memcpy(d.camera_up, cu, 12);
    d.camera_fov = 60.0f;
    std::memcpy(d.background_color, bg, 16);
    return o;
}

}  // namespace

extern "C" {

int mcrt_build_skin_scene(const uint8_t* rgba8, int w, int h, const float pose[12], mcrt_scene_desc** out) {
    if (!rgba8 || !out || w != 64 || (h != 64 && h != 32)) return MCRT_ERR_INVALID;  // skin_parser.cpp:122-131
    SkinImage img;
    img.w = w;
    img.h = h;
    img.px.resize(static_cast<size_t>(w) * h * 4);
    for (size_t i = 0; i < img.px.size(); ++i) img.px[i] = rgba8[i] / 255.0f;
    Skin s;
    // part origins: head, body, right arm, left arm, right leg, left leg
    s.inner[0] = unwrap(img, 0, 0, 8, 8, 8);
    s.outer[0] = unwrap(img, 32, 0, 8, 8, 8);
    s.inner[1] = unwrap(img, 16, 16, 8, 12, 4);
    s.inner[2] = unwrap(img, 40, 16, 4, 12, 4);
    s.inner[4] = unwrap(img, 0, 16, 4, 12, 4);
    if (h == 64) {
        s.outer[1] = unwrap(img, 16, 32, 8, 12, 4);
        s.outer[2] = unwrap(img, 40, 32, 4, 12, 4);
        s.inner[3] = unwrap(img, 32, 48, 4, 12, 4);
        s.outer[3] = unwrap(img, 48, 48, 4, 12, 4);
        s.outer[4] = unwrap(img, 0, 32, 4, 12, 4);
        s.inner[5] = unwrap(img, 16, 48, 4, 12, 4);
        s.outer[5] = unwrap(img, 0, 48, 4, 12, 4);
    } else {
        s.inner[3] = mirror_part(s.inner[2]);
        s.inner[5] = mirror_part(s.inner[4]);
    }
    *out = &assemble(s, pose)->desc;
    return MCRT_OK;
}

int mcrt_build_default_scene(const float pose[12], mcrt_scene_desc** out) {
    if (!out) return MCRT_ERR_INVALID;
    Skin s;
    Region white;
    white.w = white.h = 1;
    white.px = {1.0f, 1.0f, 1.0f, 1.0f};
    for (auto& p : s.inner) p.top = p.bottom = p.front = p.back = p.left = p.right = white;
    *out = &assemble(s, pose)->desc;
    return MCRT_OK;
}

int mcrt_builtin_pose(int index, float pose_out[12]) {
    // {head, body, rightArm, leftArm, rightLeg, leftLeg} x {rotX, rotZ}
    static const float poses[7][12] = {
        {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},               // standing
        {0, 0, 0, 0, 30, 0, -30, 0, -25, 0, 25, 0},         // walking
        {-5, 0, 5, 0, 50, 0, -50, 0, -45, 0, 45, 0},        // running
        {5, 0, 0, 0, -140, -20, 0, 0, 0, 0, 0, 0},          // waving
        {0, 0, 0, 0, -10, 0, -10, 0, -90, 0, -90, 0},       // sitting
        {-10, 0, 5, 0, -90, 10, 20, -10, -15, 0, 20, 0},    // fighting
        {30, 15, 0, 5, -45, 30, 150, -10, 0, 0, 0, 0},      // dab
    };
    if (index < 0 || index > 6 || !pose_out) return MCRT_ERR_INVALID;
    std::memcpy(pose_out, poses[index], sizeof poses[index]);
    return MCRT_OK;
}

void mcrt_scene_desc_free(mcrt_scene_desc* d) {
    if (d) delete reinterpret_cast<OwnedDesc*>(d);
}

}  // extern "C"
