// ref_shim.cpp — C-ABI shim around the REAL reference, TEST INFRASTRUCTURE ONLY.
//
// Compiled by oracle/Makefile together with the reference's own sources where they lie under
// /root/reference (never copied into this repo) into oracle/_ref/libmcref.so.  It converts the POD
// scene description of include/mcrt.h into the reference's `Scene` and calls the reference's own
// TileRenderer / RayTracer / intersect* / shade, so the oracle restatement and the HIP path can be
// compared with the genuine article.  Exports mirror oracle/mcrt_oracle.h with the prefix mcref_.

#include "mcrt.h"

#include "raytracer/intersection.h"
#include "raytracer/raytracer.h"
#include "raytracer/shading.h"
#include "raytracer/tile_renderer.h"
#include "scene/mesh_builder.h"
#include "scene/pose.h"
#include "scene/scene.h"
#include "skin/skin_parser.h"
#include "output/image_writer.h"

#include <stb/stb_image_write.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <thread>
#include <atomic>
#include <map>
#include <string>
#include <unistd.h>
#include <vector>

namespace {

// Build a reference Scene from the POD description.  Textures referenced by a mesh are placed in
// that mesh's ownedTextures when there are at most 6 distinct ones (the MeshBuilder layout),
// otherwise they live in a side pool that outlives the Scene (pointer identity is all the
// reference needs: triangle.h:15, intersection.cpp:124-129).
struct BuiltScene {
    Scene scene;
    std::vector<std::unique_ptr<TextureRegion>> pool;
};

TextureRegion to_region(const mcrt_texture& t) {
    std::vector<Color> px(static_cast<size_t>(t.n_pixels > 0 ? t.n_pixels : 0));
    for (size_t i = 0; i < px.size(); ++i)
        px[i] = Color(t.rgba[4 * i + 0], t.rgba[4 * i + 1], t.rgba[4 * i + 2], t.rgba[4 * i + 3]);
    TextureRegion r;
    r.width = t.width;
    r.height = t.height;
    r.pixels = std::move(px);
    return r;
}

void fill_tris(std::vector<Triangle>& dst, const float* v, int n) {
    dst.resize(n);
    for (int i = 0; i < n; ++i) {
        dst[i].v0 = Vec3(v[9 * i + 0], v[9 * i + 1], v[9 * i + 2]);
        dst[i].v1 = Vec3(v[9 * i + 3], v[9 * i + 4], v[9 * i + 5]);
        dst[i].v2 = Vec3(v[9 * i + 6], v[9 * i + 7], v[9 * i + 8]);
    }
}

std::unique_ptr<BuiltScene> build(const mcrt_scene_desc* d) {
    auto out = std::make_unique<BuiltScene>();
    Scene& s = out->scene;
    s.meshes.resize(d->n_meshes);
    for (int m = 0; m < d->n_meshes; ++m) {
        const mcrt_mesh& dm = d->meshes[m];
        Mesh& mesh = s.meshes[m];
        fill_tris(mesh.triangles, dm.tri_vertices, dm.n_triangles);
        fill_tris(mesh.localTriangles, dm.local_tri_vertices, dm.n_local_triangles);
        mesh.isOuterLayer = dm.is_outer_layer != 0;
        mesh.hasRotation = dm.has_rotation != 0;
        mesh.pivot = Vec3(dm.pivot[0], dm.pivot[1], dm.pivot[2]);
        mesh.rotX = dm.rot_x;
        mesh.rotZ = dm.rot_z;
        std::map<int, const TextureRegion*> placed;
        int owned = 0;
        for (int t = 0; t < dm.n_triangles; ++t) {
            int ti = dm.tri_texture[t];
            if (ti < 0) {
                mesh.triangles[t].texture = nullptr;
                continue;
            }
            auto it = placed.find(ti);
            if (it == placed.end()) {
                const TextureRegion* ptr;
                if (owned < 6) {
                    mesh.ownedTextures[owned] = to_region(d->textures[ti]);
                    ptr = &mesh.ownedTextures[owned++];
                } else {
                    out->pool.push_back(std::make_unique<TextureRegion>(to_region(d->textures[ti])));
                    ptr = out->pool.back().get();
                }
                it = placed.emplace(ti, ptr).first;
            }
            mesh.triangles[t].texture = it->second;
        }
    }
    s.light.position = Vec3(d->light_position[0], d->light_position[1], d->light_position[2]);
    s.light.color = Color(d->light_color[0], d->light_color[1], d->light_color[2], d->light_color[3]);
    s.light.intensity = d->light_intensity;
    s.light.radius = d->light_radius;
    s.camera.position = Vec3(d->camera_position[0], d->camera_position[1], d->camera_position[2]);
    s.camera.target = Vec3(d->camera_target[0], d->camera_target[1], d->camera_target[2]);
    s.camera.up = Vec3(d->camera_up[0], d->camera_up[1], d->camera_up[2]);
    s.camera.fov = d->camera_fov;
    s.backgroundColor = Color(d->background_color[0], d->background_color[1], d->background_color[2],
                              d->background_color[3]);
    return out;
}

RayTracer::Config to_config(const mcrt_config* c) {
    RayTracer::Config k;
    k.width = c->width;
    k.height = c->height;
    k.maxBounces = c->max_bounces;
    k.samplesPerPixel = c->samples_per_pixel;
    k.tileSize = c->tile_size;
    k.threadCount = c->thread_count;
    k.softShadows = c->soft_shadows != 0;
    k.shadowSamples = c->shadow_samples;
    k.aoEnabled = c->ao_enabled != 0;
    k.aoSamples = c->ao_samples;
    k.aoRadius = c->ao_radius;
    k.aoIntensity = c->ao_intensity;
    k.dofEnabled = c->dof_enabled != 0;
    k.aperture = c->aperture;
    k.focusDistance = c->focus_distance;
    k.gradientBg = c->gradient_bg != 0;
    k.gradientScale = c->gradient_scale;
    k.bgCenter = Color(c->bg_center[0], c->bg_center[1], c->bg_center[2], c->bg_center[3]);
    k.bgEdge = Color(c->bg_edge[0], c->bg_edge[1], c->bg_edge[2], c->bg_edge[3]);
    return k;
}

void store_hit(const HitResult& h, mcrt_hit* o) {
    o->hit = h.hit ? 1 : 0;
    o->t = h.t;
    o->point[0] = h.point.x;
    o->point[1] = h.point.y;
    o->point[2] = h.point.z;
    o->normal[0] = h.normal.x;
    o->normal[1] = h.normal.y;
    o->normal[2] = h.normal.z;
    o->texture_color[0] = h.textureColor.r;
    o->texture_color[1] = h.textureColor.g;
    o->texture_color[2] = h.textureColor.b;
    o->texture_color[3] = h.textureColor.a;
    o->is_outer_layer = h.isOuterLayer ? 1 : 0;
}

// Scene → owned POD description
struct OwnedDesc {
    mcrt_scene_desc desc;
    std::vector<mcrt_mesh> meshes;
    std::vector<mcrt_texture> textures;
    std::vector<std::vector<float>> floats;
    std::vector<std::vector<int32_t>> ints;
};

const float* keep(OwnedDesc& o, std::vector<float> v) {
    o.floats.push_back(std::move(v));
    return o.floats.back().data();
}

std::vector<float> tri_floats(const std::vector<Triangle>& t) {
    std::vector<float> v;
    v.reserve(t.size() * 9);
    for (const auto& tr : t) {
        const Vec3* p[3] = {&tr.v0, &tr.v1, &tr.v2};
        for (auto q : p) {
            v.push_back(q->x);
            v.push_back(q->y);
            v.push_back(q->z);
        }
    }
    return v;
}

OwnedDesc* to_desc(const Scene& s) {
    auto* o = new OwnedDesc();
    o->floats.reserve(s.meshes.size() * 8 + 8);
    o->ints.reserve(s.meshes.size() + 1);
    o->meshes.resize(s.meshes.size());
    for (size_t m = 0; m < s.meshes.size(); ++m) {
        const Mesh& mesh = s.meshes[m];
        mcrt_mesh& dm = o->meshes[m];
        std::map<const TextureRegion*, int> seen;
        std::vector<int32_t> tix(mesh.triangles.size(), -1);
        for (size_t t = 0; t < mesh.triangles.size(); ++t) {
            const TextureRegion* tr = mesh.triangles[t].texture;
            if (!tr) continue;
            auto it = seen.find(tr);
            if (it == seen.end()) {
                mcrt_texture tx;
                tx.width = tr->width;
                tx.height = tr->height;
                tx.n_pixels = static_cast<int64_t>(tr->pixels.size());
                std::vector<float> px;
                px.reserve(tr->pixels.size() * 4);
                for (const Color& c : tr->pixels) {
                    px.push_back(c.r);
                    px.push_back(c.g);
                    px.push_back(c.b);
                    px.push_back(c.a);
                }
                tx.rgba = keep(*o, std::move(px));
                o->textures.push_back(tx);
                it = seen.emplace(tr, static_cast<int>(o->textures.size()) - 1).first;
            }
            tix[t] = it->second;
        }
        o->ints.push_back(std::move(tix));
        dm.n_triangles = static_cast<int32_t>(mesh.triangles.size());
        dm.tri_vertices = keep(*o, tri_floats(mesh.triangles));
        dm.tri_texture = o->ints.back().data();
        dm.n_local_triangles = static_cast<int32_t>(mesh.localTriangles.size());
        dm.local_tri_vertices = keep(*o, tri_floats(mesh.localTriangles));
        dm.is_outer_layer = mesh.isOuterLayer ? 1 : 0;
        dm.has_rotation = mesh.hasRotation ? 1 : 0;
        dm.pivot[0] = mesh.pivot.x;
        dm.pivot[1] = mesh.pivot.y;
        dm.pivot[2] = mesh.pivot.z;
        dm.rot_x = mesh.rotX;
        dm.rot_z = mesh.rotZ;
    }
    mcrt_scene_desc& d = o->desc;
    d.n_meshes = static_cast<int32_t>(o->meshes.size());
    d.meshes = o->meshes.data();
    d.n_textures = static_cast<int32_t>(o->textures.size());
    d.textures = o->textures.data();
    d.light_position[0] = s.light.position.x;
    d.light_position[1] = s.light.position.y;
    d.light_position[2] = s.light.position.z;
    d.light_color[0] = s.light.color.r;
    d.light_color[1] = s.light.color.g;
    d.light_color[2] = s.light.color.b;
    d.light_color[3] = s.light.color.a;
    d.light_intensity = s.light.intensity;
    d.light_radius = s.light.radius;
    d.camera_position[0] = s.camera.position.x;
    d.camera_position[1] = s.camera.position.y;
    d.camera_position[2] = s.camera.position.z;
    d.camera_target[0] = s.camera.target.x;
    d.camera_target[1] = s.camera.target.y;
    d.camera_target[2] = s.camera.target.z;
    d.camera_up[0] = s.camera.up.x;
    d.camera_up[1] = s.camera.up.y;
    d.camera_up[2] = s.camera.up.z;
    d.camera_fov = s.camera.fov;
    d.background_color[0] = s.backgroundColor.r;
    d.background_color[1] = s.backgroundColor.g;
    d.background_color[2] = s.backgroundColor.b;
    d.background_color[3] = s.backgroundColor.a;
    return o;
}

Pose to_pose(const float p[12]) {
    Pose q;
    PartPose* parts[6] = {&q.head, &q.body, &q.rightArm, &q.leftArm, &q.rightLeg, &q.leftLeg};
    for (int i = 0; i < 6; ++i) {
        parts[i]->rotX = p ? p[2 * i] : 0.0f;
        parts[i]->rotZ = p ? p[2 * i + 1] : 0.0f;
    }
    return q;
}

}  // namespace

extern "C" {

int mcref_generate_tiles(int w, int h, int tile, mcrt_tile* tiles, int capacity) {
    std::vector<Tile> g = TileRenderer::generateTiles(w, h, tile);
    int n = static_cast<int>(g.size());
    for (int i = 0; i < n && i < capacity && tiles; ++i)
        tiles[i] = mcrt_tile{g[i].x, g[i].y, g[i].width, g[i].height};
    return n;
}

int mcref_render(const mcrt_scene_desc* scene, const mcrt_config* cfg, float* out_rgba,
                 mcrt_progress_fn progress, void* user) {
    auto b = build(scene);
    RayTracer::Config k = to_config(cfg);
    std::function<void(int, int)> cb;
    if (progress) cb = [=](int d, int t) { progress(d, t, user); };
    Image img = TileRenderer::render(b->scene, k, cb);
    if (img.width > 0 && img.height > 0)
        std::memcpy(out_rgba, img.pixels.data(), img.pixels.size() * sizeof(Color));
    return static_cast<int>(TileRenderer::lastErrors().size());
}

int mcref_render_tile(const mcrt_scene_desc* scene, const mcrt_config* cfg, const mcrt_tile* tile,
                      float* frame_rgba) {
    auto b = build(scene);
    RayTracer::Config k = to_config(cfg);
    Image img(k.width, k.height);
    std::memcpy(img.pixels.data(), frame_rgba, img.pixels.size() * sizeof(Color));
    Tile t{tile->x, tile->y, tile->width, tile->height};
    TileRenderer::renderTile(t, b->scene, k, img);
    std::memcpy(frame_rgba, img.pixels.data(), img.pixels.size() * sizeof(Color));
    return 0;
}

// Tile rows row_first, row_first + row_step, ... of the frame, rendered by the reference's own renderTile over a
// std::thread pool with an atomic tile queue (the structure of TileRenderer::render, tile_renderer.cpp:129-189;
// threadCount <= 0 -> hardware_concurrency).  Only the pixel rows of those tile rows are written to out_rgba (a full
// width x height frame).  What bench.py times for frames too large to render whole on the CPU: a cyclic sample of the
// frame's tile rows, scaled by rows / sampled rows.  Returns the number of tiles rendered.
int mcref_render_rows(const mcrt_scene_desc* scene, const mcrt_config* cfg, int row_first, int row_step, float* out_rgba) {
    auto b = build(scene);
    RayTracer::Config k = to_config(cfg);
    if (k.width <= 0 || k.height <= 0 || k.tileSize <= 0 || row_first < 0 || row_step < 1) return 0;
    std::vector<Tile> all = TileRenderer::generateTiles(k.width, k.height, k.tileSize);
    std::vector<Tile> tiles;
    for (const Tile& t : all) {
        const int row = t.y / k.tileSize;
        if (row >= row_first && (row - row_first) % row_step == 0) tiles.push_back(t);
    }
    Image img(k.width, k.height);
    int threads = k.threadCount;
    if (threads <= 0) {
        threads = static_cast<int>(std::thread::hardware_concurrency());
        if (threads <= 0) threads = 1;
    }
    threads = std::min<int>(threads, static_cast<int>(tiles.size()));
    std::atomic<int> next{0};
    auto worker = [&]() {
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= static_cast<int>(tiles.size())) break;
            TileRenderer::renderTile(tiles[static_cast<size_t>(i)], b->scene, k, img);
        }
    };
    std::vector<std::thread> pool;
    for (int i = 0; i < threads; ++i) pool.emplace_back(worker);
    for (auto& t : pool) t.join();
    for (const Tile& t : tiles)
        if (t.x == 0)
            std::memcpy(out_rgba + 4 * static_cast<size_t>(t.y) * k.width, img.pixels.data() + static_cast<size_t>(t.y) * k.width,
                        static_cast<size_t>(t.height) * k.width * sizeof(Color));
    return static_cast<int>(tiles.size());
}

int mcref_intersect(const mcrt_scene_desc* scene, const float* rays, int n, mcrt_hit* out) {
    auto b = build(scene);
    for (int i = 0; i < n; ++i) {
        Ray r(Vec3(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]),
              Vec3(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]));
        store_hit(intersectScene(r, b->scene), out + i);
    }
    return 0;
}

int mcref_intersect_mesh(const mcrt_scene_desc* scene, int mesh_index, const float* rays, int n,
                         mcrt_hit* out) {
    auto b = build(scene);
    if (mesh_index < 0 || mesh_index >= static_cast<int>(b->scene.meshes.size())) return 1;
    for (int i = 0; i < n; ++i) {
        Ray r(Vec3(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]),
              Vec3(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]));
        store_hit(intersectMesh(r, b->scene.meshes[mesh_index]), out + i);
    }
    return 0;
}

int mcref_trace(const mcrt_scene_desc* scene, const mcrt_config* cfg, const float* rays, int n,
                int depth, int max_bounces, float* out_rgba) {
    auto b = build(scene);
    RayTracer::Config k;
    if (cfg) k = to_config(cfg);
    for (int i = 0; i < n; ++i) {
        Ray r(Vec3(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]),
              Vec3(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]));
        Color c = RayTracer::traceRay(r, b->scene, depth, max_bounces, ShadingParams{}, cfg ? &k : nullptr);
        out_rgba[4 * i + 0] = c.r;
        out_rgba[4 * i + 1] = c.g;
        out_rgba[4 * i + 2] = c.b;
        out_rgba[4 * i + 3] = c.a;
    }
    return 0;
}

int mcref_shade(const mcrt_scene_desc* scene, const mcrt_hit* hit, const float view_dir[3],
                const float params[4], float shadow_factor, float out_rgba[4]) {
    auto b = build(scene);
    HitResult h;
    h.hit = hit->hit != 0;
    h.t = hit->t;
    h.point = Vec3(hit->point[0], hit->point[1], hit->point[2]);
    h.normal = Vec3(hit->normal[0], hit->normal[1], hit->normal[2]);
    h.textureColor = Color(hit->texture_color[0], hit->texture_color[1], hit->texture_color[2],
                           hit->texture_color[3]);
    h.isOuterLayer = hit->is_outer_layer != 0;
    ShadingParams p;
    if (params) {
        p.kd = params[0];
        p.ks = params[1];
        p.ambient = params[2];
        p.shininess = params[3];
    }
    Color c = shade(h, Vec3(view_dir[0], view_dir[1], view_dir[2]), b->scene.light, b->scene, p,
                    shadow_factor);
    out_rgba[0] = c.r;
    out_rgba[1] = c.g;
    out_rgba[2] = c.b;
    out_rgba[3] = c.a;
    return 0;
}

int mcref_in_shadow(const mcrt_scene_desc* scene, const float point[3], const float normal[3],
                    const float light_pos[3]) {
    auto b = build(scene);
    return isInShadow(Vec3(point[0], point[1], point[2]), Vec3(normal[0], normal[1], normal[2]),
                      Vec3(light_pos[0], light_pos[1], light_pos[2]), b->scene)
               ? 1
               : 0;
}

float mcref_soft_shadow(const mcrt_scene_desc* scene, const float point[3], const float normal[3],
                        int samples, uint32_t seed) {
    auto b = build(scene);
    return computeSoftShadow(Vec3(point[0], point[1], point[2]), Vec3(normal[0], normal[1], normal[2]),
                             b->scene.light, b->scene, samples, seed);
}

float mcref_ao(const mcrt_scene_desc* scene, const float point[3], const float normal[3], int samples,
               float radius, uint32_t seed) {
    auto b = build(scene);
    return RayTracer::computeAO(Vec3(point[0], point[1], point[2]),
                                Vec3(normal[0], normal[1], normal[2]), b->scene, samples, radius, seed);
}

void mcref_background(const mcrt_scene_desc* scene, const mcrt_config* cfg, float u, float v,
                      float out_rgba[4]) {
    auto b = build(scene);
    RayTracer::Config k;
    if (cfg) k = to_config(cfg);
    Color c = RayTracer::backgroundColor(b->scene, u, v, cfg ? &k : nullptr);
    out_rgba[0] = c.r;
    out_rgba[1] = c.g;
    out_rgba[2] = c.b;
    out_rgba[3] = c.a;
}

void mcref_camera_ray(const mcrt_scene_desc* scene, float u, float v, float aspect, float out_ray[6]) {
    auto b = build(scene);
    Ray r = b->scene.camera.generateRay(u, v, aspect);
    out_ray[0] = r.origin.x;
    out_ray[1] = r.origin.y;
    out_ray[2] = r.origin.z;
    out_ray[3] = r.direction.x;
    out_ray[4] = r.direction.y;
    out_ray[5] = r.direction.z;
}

// the reference's seed expression compiled by the same compiler/flags as the reference itself
uint32_t mcref_seed_cast(float f) { return static_cast<unsigned int>(f); }

void mcref_quantize(const float* rgba, uint8_t* out, size_t n_pixels) {
    // goes through the reference's own ImageWriter formula by writing/reading is overkill; restate
    // it here with the reference's Color::clamp so the formula under test is the reference's.
    for (size_t i = 0; i < n_pixels; ++i) {
        Color c = Color(rgba[4 * i], rgba[4 * i + 1], rgba[4 * i + 2], rgba[4 * i + 3]).clamp();
        out[4 * i + 0] = static_cast<uint8_t>(c.r * 255.0f + 0.5f);
        out[4 * i + 1] = static_cast<uint8_t>(c.g * 255.0f + 0.5f);
        out[4 * i + 2] = static_cast<uint8_t>(c.b * 255.0f + 0.5f);
        out[4 * i + 3] = static_cast<uint8_t>(c.a * 255.0f + 0.5f);
    }
}

// The reference's PNG hand-off, for checking the build's store-only PNG writer: ImageWriter::writePNG
// (image_writer.cpp:6-28) and Image::load (image.cpp:8-25, the vendored stb decoder + /255.0f).
int mcref_write_png(const char* path, const float* rgba, int w, int h) {
    Image img(w, h);
    for (size_t i = 0; i < static_cast<size_t>(w) * h; ++i)
        img.pixels[i] = Color(rgba[4 * i], rgba[4 * i + 1], rgba[4 * i + 2], rgba[4 * i + 3]);
    return ImageWriter::writePNG(img, path) ? 1 : 0;
}
int mcref_load_png(const char* path, float* out_rgba, int capacity_pixels, int* w, int* h) {
    std::optional<Image> img = Image::load(path);
    if (!img) return 0;
    *w = img->width;
    *h = img->height;
    if (static_cast<long long>(img->width) * img->height > capacity_pixels) return 0;
    for (size_t i = 0; i < img->pixels.size(); ++i) {
        out_rgba[4 * i + 0] = img->pixels[i].r;
        out_rgba[4 * i + 1] = img->pixels[i].g;
        out_rgba[4 * i + 2] = img->pixels[i].b;
        out_rgba[4 * i + 3] = img->pixels[i].a;
    }
    return 1;
}

// SkinParser::parse + MeshBuilder::buildScene through the reference's own code.  The skin is
// written as a PNG with the reference's vendored stb and parsed back, exactly the app's path.
int mcref_build_skin_scene(const uint8_t* rgba8, int w, int h, const float pose[12],
                           mcrt_scene_desc** out) {
    char path[] = "/tmp/mcref_skin_XXXXXX.png";
    int fd = mkstemps(path, 4);
    if (fd < 0) return 1;
    close(fd);
    int ok = stbi_write_png(path, w, h, 4, rgba8, w * 4);
    if (!ok) {
        unlink(path);
        return 2;
    }
    auto res = SkinParser::parse(path);
    unlink(path);
    if (!res.isOk()) return 3;
    Scene s = MeshBuilder::buildScene(*res.value, to_pose(pose));
    *out = &to_desc(s)->desc;
    return 0;
}

int mcref_build_default_scene(const float pose[12], mcrt_scene_desc** out) {
    Scene s = MeshBuilder::buildDefaultScene(to_pose(pose));
    *out = &to_desc(s)->desc;
    return 0;
}

int mcref_builtin_pose(int index, float pose_out[12]) {
    auto poses = getBuiltinPoses();
    if (index < 0 || index >= static_cast<int>(poses.size())) return 1;
    const Pose& q = poses[index];
    const PartPose* parts[6] = {&q.head, &q.body, &q.rightArm, &q.leftArm, &q.rightLeg, &q.leftLeg};
    for (int i = 0; i < 6; ++i) {
        pose_out[2 * i] = parts[i]->rotX;
        pose_out[2 * i + 1] = parts[i]->rotZ;
    }
    return 0;
}

void mcref_scene_desc_free(mcrt_scene_desc* d) {
    // desc is the first member of OwnedDesc
    delete reinterpret_cast<OwnedDesc*>(d);
}

}  // extern "C"
