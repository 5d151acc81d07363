// mcrt_scene_adapter.hpp — turns a reference-shaped `Scene` / `RayTracer::Config` into the POD
// types of include/mcrt.h.  Header-only and duck-typed: it compiles against the reference's own
// headers (/root/reference/src/scene/scene.h, mesh.h, triangle.h, raytracer/raytracer.h) and
// against the mirror in mcskin_types.hpp alike, because it only names members:
//   scene.meshes[i].{triangles, localTriangles, isOuterLayer, hasRotation, pivot, rotX, rotZ}
//   triangle.{v0, v1, v2, texture}      texture->{width, height, pixels[i].{r,g,b,a}}
//   scene.light.{position, color, intensity, radius}   scene.camera.{position, target, up, fov}
//   scene.backgroundColor
// Texture pointers (Triangle::texture, triangle.h:15) become indices into a per-scene texture
// table; nullptr becomes -1 (→ opaque magenta, intersection.cpp:305).
#ifndef MCRT_SCENE_ADAPTER_HPP
#define MCRT_SCENE_ADAPTER_HPP

#include "mcrt.h"

#include <cstdint>
#include <deque>
#include <map>
#include <vector>

namespace mcrt_adapter {

class SceneDescription {
public:
    template <class SceneT>
    explicit SceneDescription(const SceneT& scene) {
        std::map<const void*, int32_t> tex_index;
        meshes_.resize(scene.meshes.size());
        for (size_t m = 0; m < scene.meshes.size(); ++m) {
            const auto& mesh = scene.meshes[m];
            mcrt_mesh& out = meshes_[m];
            std::vector<int32_t>& tix = ints_.emplace_back();
            tix.reserve(mesh.triangles.size());
            for (const auto& tri : mesh.triangles) {
                if (!tri.texture) {
                    tix.push_back(-1);
                    continue;
                }
                auto it = tex_index.find(tri.texture);
                if (it == tex_index.end()) {
                    it = tex_index.emplace(tri.texture, add_texture(*tri.texture)).first;
                }
                tix.push_back(it->second);
            }
            out.n_triangles = static_cast<int32_t>(mesh.triangles.size());
            out.tri_vertices = vertices(mesh.triangles);
            out.tri_texture = tix.data();
            out.n_local_triangles = static_cast<int32_t>(mesh.localTriangles.size());
            out.local_tri_vertices = vertices(mesh.localTriangles);
            out.is_outer_layer = mesh.isOuterLayer ? 1 : 0;
            out.has_rotation = mesh.hasRotation ? 1 : 0;
            out.pivot[0] = mesh.pivot.x, out.pivot[1] = mesh.pivot.y, out.pivot[2] = mesh.pivot.z;
            out.rot_x = mesh.rotX;
            out.rot_z = mesh.rotZ;
        }
        desc_.n_meshes = static_cast<int32_t>(meshes_.size());
        desc_.meshes = meshes_.data();
        desc_.n_textures = static_cast<int32_t>(textures_.size());
        desc_.textures = textures_.data();
        put3(desc_.light_position, scene.light.position);
        put4(desc_.light_color, scene.light.color);
        desc_.light_intensity = scene.light.intensity;
        desc_.light_radius = scene.light.radius;
        put3(desc_.camera_position, scene.camera.position);
        put3(desc_.camera_target, scene.camera.target);
        put3(desc_.camera_up, scene.camera.up);
        desc_.camera_fov = scene.camera.fov;
        put4(desc_.background_color, scene.backgroundColor);
    }
    SceneDescription(const SceneDescription&) = delete;
    SceneDescription& operator=(const SceneDescription&) = delete;

    const mcrt_scene_desc* get() const { return &desc_; }

private:
    template <class V>
    static void put3(float* d, const V& v) { d[0] = v.x, d[1] = v.y, d[2] = v.z; }
    template <class C>
    static void put4(float* d, const C& c) { d[0] = c.r, d[1] = c.g, d[2] = c.b, d[3] = c.a; }

    template <class Tris>
    const float* vertices(const Tris& tris) {
        std::vector<float>& v = floats_.emplace_back();
        v.reserve(tris.size() * 9);
        for (const auto& t : tris) {
            v.insert(v.end(), {t.v0.x, t.v0.y, t.v0.z, t.v1.x, t.v1.y, t.v1.z, t.v2.x, t.v2.y, t.v2.z});
        }
        return v.data();
    }
    template <class Tex>
    int32_t add_texture(const Tex& tex) {
        std::vector<float>& px = floats_.emplace_back();
        px.reserve(tex.pixels.size() * 4);
        for (const auto& c : tex.pixels) px.insert(px.end(), {c.r, c.g, c.b, c.a});
        mcrt_texture t;
        t.width = tex.width;
        t.height = tex.height;
        t.n_pixels = static_cast<int64_t>(tex.pixels.size());
        t.rgba = px.data();
        textures_.push_back(t);
        return static_cast<int32_t>(textures_.size()) - 1;
    }

    mcrt_scene_desc desc_{};
    std::vector<mcrt_mesh> meshes_;
    std::vector<mcrt_texture> textures_;
    std::deque<std::vector<float>> floats_;  // deque: element addresses stay put
    std::deque<std::vector<int32_t>> ints_;
};

template <class ConfigT>
mcrt_config to_mcrt_config(const ConfigT& c) {  // RayTracer::Config, raytracer.h:10-38
    mcrt_config k;
    k.width = c.width;
    k.height = c.height;
    k.max_bounces = c.maxBounces;
    k.samples_per_pixel = c.samplesPerPixel;
    k.tile_size = c.tileSize;
    k.thread_count = c.threadCount;
    k.soft_shadows = c.softShadows ? 1 : 0;
    k.shadow_samples = c.shadowSamples;
    k.ao_enabled = c.aoEnabled ? 1 : 0;
    k.ao_samples = c.aoSamples;
    k.ao_radius = c.aoRadius;
    k.ao_intensity = c.aoIntensity;
    k.dof_enabled = c.dofEnabled ? 1 : 0;
    k.aperture = c.aperture;
    k.focus_distance = c.focusDistance;
    k.gradient_bg = c.gradientBg ? 1 : 0;
    k.gradient_scale = c.gradientScale;
    k.bg_center[0] = c.bgCenter.r, k.bg_center[1] = c.bgCenter.g, k.bg_center[2] = c.bgCenter.b, k.bg_center[3] = c.bgCenter.a;
    k.bg_edge[0] = c.bgEdge.r, k.bg_edge[1] = c.bgEdge.g, k.bg_edge[2] = c.bgEdge.b, k.bg_edge[3] = c.bgEdge.a;
    return k;
}

}  // namespace mcrt_adapter

#endif
