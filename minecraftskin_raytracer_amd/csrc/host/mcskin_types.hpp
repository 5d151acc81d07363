// mcskin_types.hpp — stand-alone mirror of the reference types that cross the TileRenderer
// boundary, so that callers written against the reference (and tile_renderer_hip.cpp) compile
// without the reference tree.  One header on purpose; same names, members and defaults as
//   math/vec3.h, math/color.h, math/ray.h, skin/texture_region.h, skin/image.h (container part),
//   scene/triangle.h, scene/mesh.h, scene/scene.h, raytracer/raytracer.h (Config only),
//   raytracer/tile_renderer.h
// of /root/reference/src.  Only data + the trivial algebra callers use to BUILD scenes lives here;
// no ray tracing happens on the host.
#ifndef MCSKIN_TYPES_HPP
#define MCSKIN_TYPES_HPP

#include <algorithm>
#include <array>
#include <cmath>
#include <cstddef>
#include <functional>
#include <string>
#include <vector>

struct Vec3 {
    float x = 0.0f, y = 0.0f, z = 0.0f;
    Vec3() = default;
    Vec3(float px, float py, float pz) : x(px), y(py), z(pz) {}
    friend Vec3 operator+(Vec3 a, const Vec3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
    friend Vec3 operator-(Vec3 a, const Vec3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
    friend Vec3 operator*(Vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
    friend Vec3 operator*(float s, Vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
    Vec3 operator-() const { return {-x, -y, -z}; }
    // reciprocal-then-multiply, like the reference (vec3.h:22)
    Vec3 operator/(float s) const {
        const float r = 1.0f / s;
        return {x * r, y * r, z * r};
    }
    bool operator==(const Vec3& o) const { return x == o.x && y == o.y && z == o.z; }
    bool operator!=(const Vec3& o) const { return !(*this == o); }
    float dot(const Vec3& o) const { return x * o.x + y * o.y + z * o.z; }
    Vec3 cross(const Vec3& o) const { return {y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x}; }
    float lengthSquared() const { return dot(*this); }
    float length() const { return std::sqrt(lengthSquared()); }
    Vec3 normalize() const {
        const float l = length();
        return l < 1e-8f ? Vec3{} : *this / l;
    }
};

struct Color {
    float r = 0.0f, g = 0.0f, b = 0.0f, a = 1.0f;  // default: opaque black (color.h:8)
    Color() = default;
    Color(float pr, float pg, float pb, float pa = 1.0f) : r(pr), g(pg), b(pb), a(pa) {}
    friend Color operator+(Color p, const Color& q) { return {p.r + q.r, p.g + q.g, p.b + q.b, p.a + q.a}; }
    friend Color operator*(Color p, float s) { return {p.r * s, p.g * s, p.b * s, p.a * s}; }
    friend Color operator*(Color p, const Color& q) { return {p.r * q.r, p.g * q.g, p.b * q.b, p.a * q.a}; }
    bool operator==(const Color& o) const { return r == o.r && g == o.g && b == o.b && a == o.a; }
    bool operator!=(const Color& o) const { return !(*this == o); }
    Color clamp() const {
        auto c = [](float v) { return std::clamp(v, 0.0f, 1.0f); };
        return {c(r), c(g), c(b), c(a)};
    }
};

struct Ray {
    Vec3 origin, direction;
    Ray() = default;
    Ray(const Vec3& o, const Vec3& d) : origin(o), direction(d) {}
    Vec3 at(float t) const { return origin + direction * t; }
};

struct TextureRegion {  // skin/texture_region.h:8-27
    int width = 0, height = 0;
    std::vector<Color> pixels;  // row-major
    TextureRegion() = default;
    TextureRegion(int w, int h) : width(w), height(h), pixels(static_cast<size_t>(w) * h) {}
    TextureRegion(int w, int h, std::vector<Color> px) : width(w), height(h), pixels(std::move(px)) {}
    Color sample(float u, float v) const {  // nearest texel
        if (width <= 0 || height <= 0 || pixels.empty()) return {};
        const int tx = std::clamp(static_cast<int>(u * width), 0, width - 1);
        const int ty = std::clamp(static_cast<int>(v * height), 0, height - 1);
        return pixels[static_cast<size_t>(ty) * width + tx];
    }
};

struct Triangle {  // scene/triangle.h:9-16
    Vec3 v0, v1, v2, normal;
    float u0 = 0, v0_uv = 0, u1 = 0, v1_uv = 0, u2 = 0, v2_uv = 0;
    const TextureRegion* texture = nullptr;
};

struct HitResult {  // scene/triangle.h:19-26
    bool hit = false;
    float t = 0.0f;
    Vec3 point, normal;
    Color textureColor;
    bool isOuterLayer = false;
};

// Mesh owns its six face textures; Triangle::texture points into ownedTextures, so copies and moves
// re-point those pointers (scene/mesh.h:31-119).
struct Mesh {
    std::vector<Triangle> triangles;
    bool isOuterLayer = false;
    std::array<TextureRegion, 6> ownedTextures;  // front, back, left, right, top, bottom
    bool hasRotation = false;
    Vec3 pivot;
    float rotX = 0.0f, rotZ = 0.0f;  // degrees
    std::vector<Triangle> localTriangles;

    Mesh() = default;
    Mesh(const Mesh& o) { assign(o); }
    Mesh(Mesh&& o) noexcept { assign(std::move(o)); }
    Mesh& operator=(const Mesh& o) {
        if (this != &o) assign(o);
        return *this;
    }
    Mesh& operator=(Mesh&& o) noexcept {
        if (this != &o) assign(std::move(o));
        return *this;
    }

private:
    template <class M>
    void assign(M&& o) {
        const TextureRegion* old_begin = o.ownedTextures.data();
        triangles = std::forward<M>(o).triangles;
        localTriangles = std::forward<M>(o).localTriangles;
        ownedTextures = std::forward<M>(o).ownedTextures;
        isOuterLayer = o.isOuterLayer;
        hasRotation = o.hasRotation;
        pivot = o.pivot;
        rotX = o.rotX;
        rotZ = o.rotZ;
        auto repoint = [&](std::vector<Triangle>& ts) {
            for (Triangle& t : ts)
                if (t.texture >= old_begin && t.texture < old_begin + 6) t.texture = ownedTextures.data() + (t.texture - old_begin);
        };
        repoint(triangles);
        repoint(localTriangles);
    }
};

struct Light {  // scene/scene.h:10-15
    Vec3 position;
    Color color;
    float intensity = 1.0f;
    float radius = 3.0f;
};

struct Camera {  // scene/scene.h:18-26 (generateRay runs on the device: rt_core.h camera_ray)
    Vec3 position, target, up;
    float fov = 60.0f;
};

struct Scene {  // scene/scene.h:29-34
    std::vector<Mesh> meshes;
    Light light;
    Camera camera;
    Color backgroundColor;
};

struct Image {  // skin/image.h:9-15
    int width = 0, height = 0;
    std::vector<Color> pixels;
    Image() = default;
    Image(int w, int h) : width(w), height(h), pixels(static_cast<size_t>(w) * h) {}
};

class RayTracer {
public:
    struct Config {  // raytracer/raytracer.h:10-38
        int width = 256, height = 256;
        int maxBounces = 3;
        int samplesPerPixel = 1;
        int tileSize = 32;
        int threadCount = 0;
        bool softShadows = true;
        int shadowSamples = 8;
        bool aoEnabled = false;
        int aoSamples = 8;
        float aoRadius = 3.0f;
        float aoIntensity = 0.5f;
        bool dofEnabled = false;
        float aperture = 0.5f;
        float focusDistance = 0.0f;
        bool gradientBg = true;
        float gradientScale = 1.0f;
        Color bgCenter{0.91f, 0.89f, 0.86f, 1.0f};
        Color bgEdge{0.56f, 0.63f, 0.71f, 1.0f};
    };
};

struct Tile {  // raytracer/tile_renderer.h:11-14
    int x, y, width, height;
};

class TileRenderer {  // raytracer/tile_renderer.h:16-47
public:
    static std::vector<Tile> generateTiles(int imageWidth, int imageHeight, int tileSize);
    static Image render(const Scene& scene, const RayTracer::Config& config,
                        std::function<void(int, int)> progressCallback = nullptr);
    static void renderTile(const Tile& tile, const Scene& scene, const RayTracer::Config& config, Image& output);
    struct TileError {
        int tileIndex;
        std::string message;
    };
    static const std::vector<TileError>& lastErrors();

private:
    static std::vector<TileError> errors_;
};

class ImageWriter {  // output/image_writer.h:6-12
public:
    static bool writePNG(const Image& image, const std::string& path);
};

#endif
