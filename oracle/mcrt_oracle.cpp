// mcrt_oracle.cpp — CPU oracle for the tile-render hot path.  TEST INFRASTRUCTURE ONLY.
//
// A restatement (not a copy) of the reference algorithm, written against the POD scene
// description of include/mcrt.h.  Every routine names the reference lines it follows
// (paths relative to /root/reference/src).  Arithmetic is IEEE float32 in the reference's
// operation order; compile with -ffp-contract=off and without -ffast-math.
//
// libm: sinf/cosf/powf go through include/mcrt_detmath.h, which is bit-identical to glibc 2.35's
// FMA-variant routines (checked exhaustively by tools/check_detmath.cpp); tanf and sqrtf are
// the host's (sqrtf is correctly rounded; tanf is evaluated once per ray exactly like
// scene/camera.cpp:15 does).
//
// Parity status: pinned.  (a) bit-compared with the compiled reference (oracle/_ref) by
// tests/test_oracle_vs_ref.py in the dev container; (b) bit-compared with the golden fixtures
// under tests/golden/ (generated from the compiled reference by tools/make_golden.py).

#include "mcrt_oracle.h"
#include "mcrt_detmath.h"

#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <mutex>
#include <random>
#include <thread>
#include <vector>

namespace {

// ------------------------------------------------------------------------------------------
// value types — math/vec3.h:6-51, math/color.h:5-42
// ------------------------------------------------------------------------------------------
struct F3 {
    float x, y, z;
};
struct F4 {
    float r, g, b, a;
};

inline F3 mk3(float x, float y, float z) { return F3{x, y, z}; }
inline F3 ld3(const float* p) { return F3{p[0], p[1], p[2]}; }
inline F3 add3(F3 a, F3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline F3 sub3(F3 a, F3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline F3 mul3(F3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
// vec3.h:22 — division is reciprocal-then-multiply
inline F3 div3(F3 a, float s) {
    float inv = 1.0f / s;
    return mk3(a.x * inv, a.y * inv, a.z * inv);
}
inline float dot3(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline F3 cross3(F3 a, F3 b) {
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline float len3(F3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
// vec3.h:46-50
inline F3 unit3(F3 a) {
    float l = len3(a);
    if (l < 1e-8f) return mk3(0.0f, 0.0f, 0.0f);
    return div3(a, l);
}

// std::min / std::max / std::clamp as compare-selects (algorithm header semantics)
inline float smin(float a, float b) { return (b < a) ? b : a; }
inline float smax(float a, float b) { return (a < b) ? b : a; }
inline float sclamp(float v, float lo, float hi) { return (v < lo) ? lo : ((hi < v) ? hi : v); }
inline int iclamp(int v, int lo, int hi) { return (v < lo) ? lo : ((hi < v) ? hi : v); }

inline F4 clamp4(F4 c) {  // color.h:34-41
    return F4{sclamp(c.r, 0.0f, 1.0f), sclamp(c.g, 0.0f, 1.0f), sclamp(c.b, 0.0f, 1.0f),
              sclamp(c.a, 0.0f, 1.0f)};
}

const float kPi = static_cast<float>(3.14159265358979323846);

// ------------------------------------------------------------------------------------------
// std::mt19937 + uniform_real_distribution<float>(0,1)  — SURVEY §8 a20
// ------------------------------------------------------------------------------------------
struct Mt {
    uint32_t s[624];
    int pos;
    explicit Mt(uint32_t seed) {
        s[0] = seed;
        for (int j = 1; j < 624; ++j) s[j] = 1812433253u * (s[j - 1] ^ (s[j - 1] >> 30)) + (uint32_t)j;
        pos = 624;
    }
    void refill() {
        for (int k = 0; k < 624; ++k) {
            uint32_t y = (s[k] & 0x80000000u) | (s[(k + 1) % 624] & 0x7fffffffu);
            uint32_t v = s[(k + 397) % 624] ^ (y >> 1);
            if (y & 1u) v ^= 0x9908b0dfu;
            s[k] = v;
        }
        pos = 0;
    }
    uint32_t next() {
        if (pos >= 624) refill();
        uint32_t y = s[pos++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
    // libstdc++ generate_canonical<float,24> with one 32-bit draw: float(x) / 2^32, and the
    // ">= 1 → nextafter(1,0)" fix-up (bits/random.tcc:3348-3380)
    float uniform() {
        float r = static_cast<float>(next()) * 0x1p-32f;
        if (r >= 1.0f) r = 0x1.fffffep-1f;
        return r;
    }
};

// raytracer.cpp:110-112,122-123: static_cast<unsigned int>(float) as GCC/x86-64 does it
// (cvttss2si to 64 bits, keep the low 32; out-of-range/NaN → 0x8000000000000000 → 0)
inline uint32_t seed_cast(float f) {
    if (!(f > -9.2233720e18f && f < 9.2233720e18f)) return 0u;
    return static_cast<uint32_t>(static_cast<int64_t>(f));
}

// ------------------------------------------------------------------------------------------
// scene access helpers
// ------------------------------------------------------------------------------------------
struct Ray {
    F3 o, d;
};
struct Hit {
    bool hit = false;
    float t = 0.0f;
    F3 p{0, 0, 0};
    F3 n{0, 0, 0};
    F4 tex{0.0f, 0.0f, 0.0f, 1.0f};  // Color() default
    bool outer = false;
};

// skin/texture_region.h:19-26
F4 texel(const mcrt_texture& tx, float u, float v) {
    if (tx.width <= 0 || tx.height <= 0 || tx.n_pixels <= 0) return F4{0.0f, 0.0f, 0.0f, 1.0f};
    int x = iclamp(static_cast<int>(u * tx.width), 0, tx.width - 1);
    int y = iclamp(static_cast<int>(v * tx.height), 0, tx.height - 1);
    const float* p = tx.rgba + 4 * (static_cast<int64_t>(y) * tx.width + x);
    return F4{p[0], p[1], p[2], p[3]};
}

// intersection.cpp:12-37 — rotate about pivot, X then Z, each gated by |deg| > 0.01
F3 spin(F3 point, F3 pivot, float degX, float degZ) {
    F3 p = sub3(point, pivot);
    if (std::fabs(degX) > 0.01f) {
        float rad = degX * kPi / 180.0f;
        float c = mcrt_cosf(rad), s = mcrt_sinf(rad);
        float ny = p.y * c - p.z * s;
        float nz = p.y * s + p.z * c;
        p.y = ny;
        p.z = nz;
    }
    if (std::fabs(degZ) > 0.01f) {
        float rad = degZ * kPi / 180.0f;
        float c = mcrt_cosf(rad), s = mcrt_sinf(rad);
        float nx = p.x * c - p.y * s;
        float ny = p.x * s + p.y * c;
        p.x = nx;
        p.y = ny;
    }
    return add3(p, pivot);
}

// intersection.cpp:45-64 — recomputed for every ray, as the reference does
void bounds(const float* verts, int ntri, F3& lo, F3& hi) {
    lo = mk3(FLT_MAX, FLT_MAX, FLT_MAX);
    hi = mk3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
    for (int i = 0; i < ntri * 3; ++i) {
        const float* v = verts + 3 * i;
        lo.x = smin(lo.x, v[0]);
        lo.y = smin(lo.y, v[1]);
        lo.z = smin(lo.z, v[2]);
        hi.x = smax(hi.x, v[0]);
        hi.y = smax(hi.y, v[1]);
        hi.z = smax(hi.z, v[2]);
    }
}

struct Face {
    F3 n;
    int tex;  // scene texture index, -1 = nullptr
};

// intersection.cpp:86-132 — (axis, min-side?) → face slot → normal + texture of triangle 2*slot
Face face_of(const mcrt_mesh& m, int axis, bool negSide) {
    Face f;
    int slot;
    if (axis == 2) {
        slot = negSide ? 0 : 1;
        f.n = negSide ? mk3(0, 0, -1) : mk3(0, 0, 1);
    } else if (axis == 0) {
        slot = negSide ? 3 : 2;
        f.n = negSide ? mk3(-1, 0, 0) : mk3(1, 0, 0);
    } else {
        slot = negSide ? 5 : 4;
        f.n = negSide ? mk3(0, -1, 0) : mk3(0, 1, 0);
    }
    int tri = slot * 2;
    f.tex = (tri < m.n_triangles) ? m.tri_texture[tri] : -1;
    return f;
}

// intersection.cpp:136-196
void face_uv(F3 hp, F3 lo, F3 hi, int axis, bool negSide, float& u, float& v) {
    F3 ext = sub3(hi, lo);
    float sx = (ext.x > 1e-8f) ? ext.x : 1.0f;
    float sy = (ext.y > 1e-8f) ? ext.y : 1.0f;
    float sz = (ext.z > 1e-8f) ? ext.z : 1.0f;
    if (axis == 2) {
        float lx = (hp.x - lo.x) / sx;
        float ly = (hp.y - lo.y) / sy;
        u = negSide ? (1.0f - lx) : lx;
        v = 1.0f - ly;
    } else if (axis == 0) {
        float lz = (hp.z - lo.z) / sz;
        float ly = (hp.y - lo.y) / sy;
        u = negSide ? lz : (1.0f - lz);
        v = 1.0f - ly;
    } else {
        float lx = (hp.x - lo.x) / sx;
        float lz = (hp.z - lo.z) / sz;
        u = lx;
        v = negSide ? (1.0f - lz) : lz;
    }
    u = sclamp(u, 0.0f, 1.0f);
    v = sclamp(v, 0.0f, 1.0f);
}

// intersection.cpp:268-285 ≡ :323-335 — face through which the ray leaves the box
void exit_face(const float* d, const float* o, const float* lo, const float* hi, int& axis,
               bool& negSide) {
    float best = FLT_MAX;
    axis = 0;
    negSide = false;
    for (int i = 0; i < 3; ++i) {
        if (std::fabs(d[i]) < 1e-8f) continue;
        float inv = 1.0f / d[i];
        float t0 = (lo[i] - o[i]) * inv;
        float t1 = (hi[i] - o[i]) * inv;
        bool neg = false;
        if (t0 > t1) {
            float tmp = t0;
            t0 = t1;
            t1 = tmp;
            neg = true;
        }
        if (t1 < best) {
            best = t1;
            axis = i;
            negSide = neg;
        }
    }
}

F4 face_texel(const mcrt_scene_desc& sc, const Face& f, float u, float v) {
    if (f.tex < 0) return F4{1.0f, 0.0f, 1.0f, 1.0f};  // intersection.cpp:305 magenta
    return texel(sc.textures[f.tex], u, v);
}

// intersection.cpp:200-371 — slab test against the AABB of `verts`
Hit hit_box(const mcrt_scene_desc& sc, const mcrt_mesh& m, const Ray& ray, const float* verts,
            int ntri) {
    Hit res;
    if (ntri <= 0) return res;
    F3 lo3, hi3;
    bounds(verts, ntri, lo3, hi3);

    float tmin = -FLT_MAX, tmax = FLT_MAX;
    int axis = 0;
    bool negSide = false;
    const float d[3] = {ray.d.x, ray.d.y, ray.d.z};
    const float o[3] = {ray.o.x, ray.o.y, ray.o.z};
    const float lo[3] = {lo3.x, lo3.y, lo3.z};
    const float hi[3] = {hi3.x, hi3.y, hi3.z};

    for (int i = 0; i < 3; ++i) {
        if (std::fabs(d[i]) < 1e-8f) {
            if (o[i] < lo[i] || o[i] > hi[i]) return res;
        } else {
            float inv = 1.0f / d[i];
            float t0 = (lo[i] - o[i]) * inv;
            float t1 = (hi[i] - o[i]) * inv;
            bool enterNeg = true;
            if (t0 > t1) {
                float tmp = t0;
                t0 = t1;
                t1 = tmp;
                enterNeg = false;
            }
            if (t0 > tmin) {
                tmin = t0;
                axis = i;
                negSide = enterNeg;
            }
            tmax = smin(tmax, t1);
            if (tmin > tmax || tmax < 0.0f) return res;
        }
    }

    float tHit = tmin;
    if (tHit < 0.0f) {  // origin inside the box: leave through the exit face (:255-288)
        tHit = tmax;
        if (tHit < 0.0f) return res;
        exit_face(d, o, lo, hi, axis, negSide);
    }

    F3 hp = add3(ray.o, mul3(ray.d, tHit));
    Face f = face_of(m, axis, negSide);
    float u, v;
    face_uv(hp, lo3, hi3, axis, negSide, u, v);
    F4 tc = face_texel(sc, f, u, v);

    if (tc.a == 0.0f) {  // :311-361
        if (!m.is_outer_layer) return res;
        if (tmax > tHit) {
            int ax2;
            bool neg2;
            exit_face(d, o, lo, hi, ax2, neg2);
            F3 bp = add3(ray.o, mul3(ray.d, tmax));
            Face bf = face_of(m, ax2, neg2);
            float bu, bv;
            face_uv(bp, lo3, hi3, ax2, neg2, bu, bv);
            F4 bc = face_texel(sc, bf, bu, bv);
            if (bc.a > 0.0f) {
                res.hit = true;
                res.t = tmax;
                res.p = bp;
                res.n = mul3(bf.n, -1.0f);
                res.tex = bc;
                res.outer = true;
                return res;
            }
        }
        return res;
    }

    res.hit = true;
    res.t = tHit;
    res.p = hp;
    res.n = f.n;
    res.tex = tc;
    res.outer = m.is_outer_layer != 0;
    return res;
}

// intersection.cpp:373-406
Hit hit_mesh(const mcrt_scene_desc& sc, const mcrt_mesh& m, const Ray& ray) {
    if (!m.has_rotation) return hit_box(sc, m, ray, m.tri_vertices, m.n_triangles);

    F3 pivot = ld3(m.pivot);
    F3 zero = mk3(0, 0, 0);
    F3 lo = spin(ray.o, pivot, 0.0f, -m.rot_z);
    lo = spin(lo, pivot, -m.rot_x, 0.0f);
    F3 ld = spin(ray.d, zero, 0.0f, -m.rot_z);
    ld = spin(ld, zero, -m.rot_x, 0.0f);

    Ray local{lo, unit3(ld)};
    Hit h = hit_box(sc, m, local, m.local_tri_vertices, m.n_local_triangles);
    if (h.hit) {
        h.p = spin(h.p, pivot, m.rot_x, m.rot_z);
        h.n = unit3(spin(h.n, zero, m.rot_x, m.rot_z));
        h.t = dot3(sub3(h.p, ray.o), ray.d);
    }
    return h;
}

// intersection.cpp:408-421
Hit hit_scene(const mcrt_scene_desc& sc, const Ray& ray) {
    Hit best;
    best.hit = false;
    best.t = FLT_MAX;
    for (int i = 0; i < sc.n_meshes; ++i) {
        Hit h = hit_mesh(sc, sc.meshes[i], ray);
        if (h.hit && h.t < best.t) best = h;
    }
    return best;
}

// ------------------------------------------------------------------------------------------
// shading — shading.cpp
// ------------------------------------------------------------------------------------------
bool in_shadow(const mcrt_scene_desc& sc, F3 point, F3 normal, F3 lightPos) {  // :14-26
    F3 origin = add3(point, mul3(normal, 1e-3f));
    F3 toLight = sub3(lightPos, origin);
    float dist = len3(toLight);
    if (dist < 1e-6f) return false;
    Ray r{origin, div3(toLight, dist)};
    Hit h = hit_scene(sc, r);
    return h.hit && h.t < dist;
}

float soft_shadow(const mcrt_scene_desc& sc, F3 point, F3 normal, int samples, uint32_t seed) {  // :28-60
    F3 lpos = ld3(sc.light_position);
    if (samples <= 1 || sc.light_radius < 1e-4f) return in_shadow(sc, point, normal, lpos) ? 0.0f : 1.0f;

    F3 toPoint = unit3(sub3(point, lpos));
    F3 tangent;
    if (std::fabs(toPoint.x) < 0.9f)
        tangent = unit3(cross3(mk3(1, 0, 0), toPoint));
    else
        tangent = unit3(cross3(mk3(0, 1, 0), toPoint));
    F3 bitangent = cross3(toPoint, tangent);

    Mt rng(seed);
    int lit = 0;
    for (int i = 0; i < samples; ++i) {
        float angle = 2.0f * kPi * rng.uniform();
        float r = sc.light_radius * std::sqrt(rng.uniform());
        F3 off = add3(mul3(tangent, r * mcrt_cosf(angle)), mul3(bitangent, r * mcrt_sinf(angle)));
        F3 sample = add3(lpos, off);
        if (!in_shadow(sc, point, normal, sample)) ++lit;
    }
    return static_cast<float>(lit) / static_cast<float>(samples);
}

struct ShadeParams {
    float kd = 0.75f, ks = 0.15f, ambient = 0.20f, shininess = 16.0f;  // shading.h:9-14
};

F4 shade(const mcrt_scene_desc& sc, const Hit& hit, F3 viewDir, const ShadeParams& pr,
         float shadowFactor) {  // :62-96
    F4 tex = hit.tex;
    float alpha = tex.a;
    F4 amb{tex.r * pr.ambient, tex.g * pr.ambient, tex.b * pr.ambient, tex.a * pr.ambient};

    F3 lpos = ld3(sc.light_position);
    F3 L = unit3(sub3(lpos, hit.p));
    F3 N = unit3(hit.n);
    F3 V = unit3(viewDir);

    float vis = shadowFactor;
    if (vis < 0.0f) vis = in_shadow(sc, hit.p, N, lpos) ? 0.0f : 1.0f;

    const float* lc = sc.light_color;
    float ndl = smax(0.0f, dot3(N, L));
    float kdiff = pr.kd * ndl * vis;
    F4 diff{tex.r * lc[0] * kdiff, tex.g * lc[1] * kdiff, tex.b * lc[2] * kdiff, tex.a * lc[3] * kdiff};

    F3 H = unit3(add3(L, V));
    float ndh = smax(0.0f, dot3(N, H));
    float spec = mcrt_powf(ndh, pr.shininess);
    float kspec = pr.ks * spec * vis;
    F4 sp{lc[0] * kspec, lc[1] * kspec, lc[2] * kspec, lc[3] * kspec};

    F4 out{amb.r + diff.r + sp.r, amb.g + diff.g + sp.g, amb.b + diff.b + sp.b, amb.a + diff.a + sp.a};
    out.a = alpha;
    return clamp4(out);
}

// ------------------------------------------------------------------------------------------
// raytracer.cpp
// ------------------------------------------------------------------------------------------
F4 background(const mcrt_scene_desc& sc, float u, float v, const mcrt_config* cfg) {  // :16-34
    if (cfg && cfg->gradient_bg) {
        float cx = u - 0.5f, cy = v - 0.5f;
        float dist = std::sqrt(cx * cx + cy * cy) * 2.0f * cfg->gradient_scale;
        dist = sclamp(dist, 0.0f, 1.0f);
        float t = dist * dist;
        F4 c;
        c.r = cfg->bg_center[0] * (1.0f - t) + cfg->bg_edge[0] * t;
        c.g = cfg->bg_center[1] * (1.0f - t) + cfg->bg_edge[1] * t;
        c.b = cfg->bg_center[2] * (1.0f - t) + cfg->bg_edge[2] * t;
        c.a = 1.0f;
        return c;
    }
    return F4{sc.background_color[0], sc.background_color[1], sc.background_color[2],
              sc.background_color[3]};
}

float ambient_occlusion(const mcrt_scene_desc& sc, F3 point, F3 normal, int samples, float radius,
                        uint32_t seed) {  // :38-78
    F3 N = unit3(normal);
    F3 T;
    if (std::fabs(N.x) < 0.9f)
        T = unit3(cross3(mk3(1, 0, 0), N));
    else
        T = unit3(cross3(mk3(0, 1, 0), N));
    F3 B = cross3(N, T);

    Mt rng(seed);
    int occluded = 0;
    for (int i = 0; i < samples; ++i) {
        float r1 = rng.uniform();
        float r2 = rng.uniform();
        float sinT = std::sqrt(1.0f - r1);
        float cosT = std::sqrt(r1);
        float phi = 2.0f * kPi * r2;
        F3 local = mk3(sinT * mcrt_cosf(phi), cosT, sinT * mcrt_sinf(phi));
        F3 world = add3(add3(mul3(T, local.x), mul3(N, local.y)), mul3(B, local.z));
        world = unit3(world);
        Ray r{add3(point, mul3(N, 1e-3f)), world};
        Hit h = hit_scene(sc, r);
        if (h.hit && h.t < radius) ++occluded;
    }
    return 1.0f - static_cast<float>(occluded) / static_cast<float>(samples);
}

F4 trace(const mcrt_scene_desc& sc, const Ray& ray, int depth, int maxBounces, const ShadeParams& pr,
         const mcrt_config* cfg) {  // :82-148
    if (depth > maxBounces) return cfg ? background(sc, 0.5f, 0.5f, cfg) : background(sc, 0, 0, nullptr);

    Hit hit = hit_scene(sc, ray);
    if (!hit.hit) {
        if (depth == 0 && cfg) return background(sc, 0.5f, 0.5f, cfg);
        return background(sc, 0, 0, nullptr);
    }

    F3 view = unit3(sub3(ray.o, hit.p));
    float shadowFactor = -1.0f;
    if (cfg && cfg->soft_shadows && cfg->shadow_samples > 1) {
        uint32_t seed = seed_cast(hit.p.x * 12345.0f + hit.p.y * 67890.0f + hit.p.z * 11111.0f +
                                  static_cast<float>(depth) * 99999.0f);
        shadowFactor = soft_shadow(sc, hit.p, hit.n, cfg->shadow_samples, seed);
    }

    F4 c = shade(sc, hit, view, pr, shadowFactor);
    float alpha = c.a;

    if (cfg && cfg->ao_enabled && depth == 0) {
        uint32_t seed = seed_cast(hit.p.x * 73856093.0f + hit.p.y * 19349663.0f + hit.p.z * 83492791.0f);
        float ao = ambient_occlusion(sc, hit.p, hit.n, cfg->ao_samples, cfg->ao_radius, seed);
        float k = 1.0f - cfg->ao_intensity * (1.0f - ao);
        c.r *= k;
        c.g *= k;
        c.b *= k;
    }

    if (depth < maxBounces) {
        F3 N = unit3(hit.n);
        F3 D = unit3(ray.d);
        F3 R = unit3(sub3(D, mul3(N, 2.0f * dot3(D, N))));
        Ray next{add3(hit.p, mul3(N, 1e-3f)), R};
        F4 rc = trace(sc, next, depth + 1, maxBounces, pr, cfg);
        const float keep = 1.0f - 0.1f, refl = 0.1f;  // SKIN_REFLECTIVITY, raytracer.cpp:11
        c = F4{c.r * keep + rc.r * refl, c.g * keep + rc.g * refl, c.b * keep + rc.b * refl,
               c.a * keep + rc.a * refl};
    }
    c.a = alpha;
    return clamp4(c);
}

// ------------------------------------------------------------------------------------------
// camera.cpp:8-26 and tile_renderer.cpp
// ------------------------------------------------------------------------------------------
Ray camera_ray(const mcrt_scene_desc& sc, float u, float v, float aspect) {
    F3 pos = ld3(sc.camera_position);
    F3 fwd = unit3(sub3(ld3(sc.camera_target), pos));
    F3 right = unit3(cross3(fwd, ld3(sc.camera_up)));
    F3 up = cross3(right, fwd);
    float halfH = std::tan(sc.camera_fov * 0.5f * kPi / 180.0f);
    float halfW = halfH * aspect;
    float su = (2.0f * u - 1.0f) * halfW;
    float sv = (2.0f * (1.0f - v) - 1.0f) * halfH;
    F3 dir = unit3(add3(add3(fwd, mul3(right, su)), mul3(up, sv)));
    return Ray{pos, dir};
}

Ray lens_ray(const mcrt_scene_desc& sc, float u, float v, float aspect, float aperture,
             float focusDist, Mt& rng) {  // tile_renderer.cpp:42-69
    Ray pin = camera_ray(sc, u, v, aspect);
    if (aperture < 1e-6f) return pin;
    F3 pos = ld3(sc.camera_position);
    F3 fwd = unit3(sub3(ld3(sc.camera_target), pos));
    F3 right = unit3(cross3(fwd, ld3(sc.camera_up)));
    F3 up = cross3(right, fwd);
    F3 focus = add3(pin.o, mul3(pin.d, focusDist));
    float angle = 2.0f * kPi * rng.uniform();
    float radius = aperture * std::sqrt(rng.uniform());
    float lx = radius * mcrt_cosf(angle);
    float ly = radius * mcrt_sinf(angle);
    F3 origin = add3(pos, add3(mul3(right, lx), mul3(up, ly)));
    return Ray{origin, unit3(sub3(focus, origin))};
}

void render_tile(const mcrt_scene_desc& sc, const mcrt_config& cfg, const mcrt_tile& tile,
                 float* frame) {  // tile_renderer.cpp:71-127
    float aspect = static_cast<float>(cfg.width) / static_cast<float>(cfg.height);
    int spp = cfg.samples_per_pixel > 1 ? cfg.samples_per_pixel : 1;
    Mt rng(static_cast<uint32_t>(tile.y * cfg.width + tile.x));

    float focusDist = cfg.focus_distance;
    if (focusDist <= 0.0f) focusDist = len3(sub3(ld3(sc.camera_target), ld3(sc.camera_position)));
    ShadeParams pr;

    for (int py = tile.y; py < tile.y + tile.height; ++py) {
        for (int px = tile.x; px < tile.x + tile.width; ++px) {
            float ar = 0.0f, ag = 0.0f, ab = 0.0f, aa = 0.0f;
            for (int s = 0; s < spp; ++s) {
                float jx = (spp == 1) ? 0.5f : rng.uniform();
                float jy = (spp == 1) ? 0.5f : rng.uniform();
                float u = (static_cast<float>(px) + jx) / static_cast<float>(cfg.width);
                float v = (static_cast<float>(py) + jy) / static_cast<float>(cfg.height);
                Ray ray = (cfg.dof_enabled && cfg.aperture > 1e-6f)
                              ? lens_ray(sc, u, v, aspect, cfg.aperture, focusDist, rng)
                              : camera_ray(sc, u, v, aspect);
                F4 c = trace(sc, ray, 0, cfg.max_bounces, pr, &cfg);
                Hit again = hit_scene(sc, ray);  // duplicate primary test, :111
                if (!again.hit) c = background(sc, u, v, &cfg);
                ar += c.r;
                ag += c.g;
                ab += c.b;
                aa += c.a;
            }
            float inv = 1.0f / static_cast<float>(spp);
            float* dst = frame + 4 * (static_cast<size_t>(py) * cfg.width + px);
            dst[0] = ar * inv;
            dst[1] = ag * inv;
            dst[2] = ab * inv;
            dst[3] = aa * inv;
        }
    }
}

std::vector<mcrt_tile> tile_grid(int w, int h, int ts) {  // tile_renderer.cpp:18-39
    std::vector<mcrt_tile> out;
    if (w <= 0 || h <= 0 || ts <= 0) return out;
    int cols = (w + ts - 1) / ts, rows = (h + ts - 1) / ts;
    out.reserve(static_cast<size_t>(cols) * rows);
    for (int ty = 0; ty < rows; ++ty)
        for (int tx = 0; tx < cols; ++tx) {
            mcrt_tile t;
            t.x = tx * ts;
            t.y = ty * ts;
            t.width = (ts < w - t.x) ? ts : (w - t.x);
            t.height = (ts < h - t.y) ? ts : (h - t.y);
            out.push_back(t);
        }
    return out;
}

void store_hit(const Hit& h, mcrt_hit* o) {
    o->hit = h.hit ? 1 : 0;
    o->t = h.t;
    o->point[0] = h.p.x;
    o->point[1] = h.p.y;
    o->point[2] = h.p.z;
    o->normal[0] = h.n.x;
    o->normal[1] = h.n.y;
    o->normal[2] = h.n.z;
    o->texture_color[0] = h.tex.r;
    o->texture_color[1] = h.tex.g;
    o->texture_color[2] = h.tex.b;
    o->texture_color[3] = h.tex.a;
    o->is_outer_layer = h.outer ? 1 : 0;
}
Hit load_hit(const mcrt_hit* i) {
    Hit h;
    h.hit = i->hit != 0;
    h.t = i->t;
    h.p = ld3(i->point);
    h.n = ld3(i->normal);
    h.tex = F4{i->texture_color[0], i->texture_color[1], i->texture_color[2], i->texture_color[3]};
    h.outer = i->is_outer_layer != 0;
    return h;
}

}  // namespace

// ==========================================================================================
// exported C ABI
// ==========================================================================================
extern "C" {

int mcrt_oracle_generate_tiles(int w, int h, int tile, mcrt_tile* tiles, int capacity) {
    std::vector<mcrt_tile> g = tile_grid(w, h, tile);
    int n = static_cast<int>(g.size());
    for (int i = 0; i < n && i < capacity && tiles; ++i) tiles[i] = g[i];
    return n;
}

int mcrt_oracle_render_tile(const mcrt_scene_desc* scene, const mcrt_config* cfg,
                            const mcrt_tile* tile, float* frame_rgba) {
    render_tile(*scene, *cfg, *tile, frame_rgba);
    return 0;
}

int mcrt_oracle_render(const mcrt_scene_desc* scene, const mcrt_config* cfg, float* out_rgba,
                       mcrt_progress_fn progress, void* user) {  // tile_renderer.cpp:129-189
    int threads = cfg->thread_count;
    if (threads <= 0) {
        threads = static_cast<int>(std::thread::hardware_concurrency());
        if (threads <= 0) threads = 1;
    }
    std::vector<mcrt_tile> tiles = tile_grid(cfg->width, cfg->height, cfg->tile_size);
    int total = static_cast<int>(tiles.size());
    if (cfg->width > 0 && cfg->height > 0) {
        // Image(w,h): pixels default-constructed to Color() = (0,0,0,1)
        size_t n = static_cast<size_t>(cfg->width) * cfg->height;
        for (size_t i = 0; i < n; ++i) {
            out_rgba[4 * i + 0] = 0.0f;
            out_rgba[4 * i + 1] = 0.0f;
            out_rgba[4 * i + 2] = 0.0f;
            out_rgba[4 * i + 3] = 1.0f;
        }
    }
    if (total == 0) return 0;

    std::atomic<int> next{0}, done{0};
    std::mutex cbMutex;
    auto worker = [&]() {
        for (;;) {
            int idx = next.fetch_add(1);
            if (idx >= total) break;
            render_tile(*scene, *cfg, tiles[idx], out_rgba);
            int d = done.fetch_add(1) + 1;
            if (progress) {
                std::lock_guard<std::mutex> lock(cbMutex);
                progress(d, total, user);
            }
        }
    };
    int n = threads < total ? threads : total;
    std::vector<std::thread> pool;
    pool.reserve(n);
    for (int i = 0; i < n; ++i) pool.emplace_back(worker);
    for (auto& t : pool) t.join();
    return 0;
}

// Tile rows row_first, row_first + row_step, ... of the frame by the same thread pool (tile_renderer.cpp:129-189
// over a subset of generateTiles' tiles); only those rows of out_rgba (a full frame) are written.  Returns the
// number of tiles rendered.  Used where a whole frame is too slow on the CPU: a cyclic sample of its tile rows.
int mcrt_oracle_render_rows(const mcrt_scene_desc* scene, const mcrt_config* cfg, int row_first, int row_step, float* out_rgba) {
    if (cfg->width <= 0 || cfg->height <= 0 || cfg->tile_size <= 0 || row_first < 0 || row_step < 1) return 0;
    int threads = cfg->thread_count;
    if (threads <= 0) {
        threads = static_cast<int>(std::thread::hardware_concurrency());
        if (threads <= 0) threads = 1;
    }
    std::vector<mcrt_tile> tiles;
    for (const mcrt_tile& t : tile_grid(cfg->width, cfg->height, cfg->tile_size)) {
        const int row = t.y / cfg->tile_size;
        if (row >= row_first && (row - row_first) % row_step == 0) tiles.push_back(t);
    }
    const int total = static_cast<int>(tiles.size());
    std::atomic<int> next{0};
    auto worker = [&]() {
        for (;;) {
            const int idx = next.fetch_add(1);
            if (idx >= total) break;
            render_tile(*scene, *cfg, tiles[static_cast<size_t>(idx)], out_rgba);
        }
    };
    const int n = threads < total ? threads : total;
    std::vector<std::thread> pool;
    for (int i = 0; i < n; ++i) pool.emplace_back(worker);
    for (auto& t : pool) t.join();
    return total;
}

int mcrt_oracle_intersect(const mcrt_scene_desc* scene, const float* rays, int n, mcrt_hit* out) {
    for (int i = 0; i < n; ++i) {
        Ray r{ld3(rays + 6 * i), ld3(rays + 6 * i + 3)};
        store_hit(hit_scene(*scene, r), out + i);
    }
    return 0;
}

int mcrt_oracle_intersect_mesh(const mcrt_scene_desc* scene, int mesh_index, const float* rays,
                               int n, mcrt_hit* out) {
    if (mesh_index < 0 || mesh_index >= scene->n_meshes) return 1;
    for (int i = 0; i < n; ++i) {
        Ray r{ld3(rays + 6 * i), ld3(rays + 6 * i + 3)};
        store_hit(hit_mesh(*scene, scene->meshes[mesh_index], r), out + i);
    }
    return 0;
}

int mcrt_oracle_trace(const mcrt_scene_desc* scene, const mcrt_config* cfg, const float* rays,
                      int n, int depth, int max_bounces, float* out_rgba) {
    ShadeParams pr;
    for (int i = 0; i < n; ++i) {
        Ray r{ld3(rays + 6 * i), ld3(rays + 6 * i + 3)};
        F4 c = trace(*scene, r, depth, max_bounces, pr, cfg);
        out_rgba[4 * i + 0] = c.r;
        out_rgba[4 * i + 1] = c.g;
        out_rgba[4 * i + 2] = c.b;
        out_rgba[4 * i + 3] = c.a;
    }
    return 0;
}

int mcrt_oracle_shade(const mcrt_scene_desc* scene, const mcrt_hit* hit, const float view_dir[3],
                      const float params[4], float shadow_factor, float out_rgba[4]) {
    ShadeParams pr;
    if (params) {
        pr.kd = params[0];
        pr.ks = params[1];
        pr.ambient = params[2];
        pr.shininess = params[3];
    }
    F4 c = shade(*scene, load_hit(hit), ld3(view_dir), pr, shadow_factor);
    out_rgba[0] = c.r;
    out_rgba[1] = c.g;
    out_rgba[2] = c.b;
    out_rgba[3] = c.a;
    return 0;
}

int mcrt_oracle_in_shadow(const mcrt_scene_desc* scene, const float point[3], const float normal[3],
                          const float light_pos[3]) {
    return in_shadow(*scene, ld3(point), ld3(normal), ld3(light_pos)) ? 1 : 0;
}

float mcrt_oracle_soft_shadow(const mcrt_scene_desc* scene, const float point[3],
                              const float normal[3], int samples, uint32_t seed) {
    return soft_shadow(*scene, ld3(point), ld3(normal), samples, seed);
}

float mcrt_oracle_ao(const mcrt_scene_desc* scene, const float point[3], const float normal[3],
                     int samples, float radius, uint32_t seed) {
    return ambient_occlusion(*scene, ld3(point), ld3(normal), samples, radius, seed);
}

void mcrt_oracle_background(const mcrt_scene_desc* scene, const mcrt_config* cfg, float u, float v,
                            float out_rgba[4]) {
    F4 c = background(*scene, u, v, cfg);
    out_rgba[0] = c.r;
    out_rgba[1] = c.g;
    out_rgba[2] = c.b;
    out_rgba[3] = c.a;
}

void mcrt_oracle_camera_ray(const mcrt_scene_desc* scene, float u, float v, float aspect,
                            float out_ray[6]) {
    Ray r = camera_ray(*scene, u, v, aspect);
    out_ray[0] = r.o.x;
    out_ray[1] = r.o.y;
    out_ray[2] = r.o.z;
    out_ray[3] = r.d.x;
    out_ray[4] = r.d.y;
    out_ray[5] = r.d.z;
}

void mcrt_oracle_mt_uniform(uint32_t seed, int n, float* out) {
    Mt g(seed);
    for (int i = 0; i < n; ++i) out[i] = g.uniform();
}

void mcrt_oracle_mt_uniform_std(uint32_t seed, int n, float* out) {
    std::mt19937 g(seed);
    std::uniform_real_distribution<float> d(0.0f, 1.0f);
    for (int i = 0; i < n; ++i) out[i] = d(g);
}

uint32_t mcrt_oracle_seed_cast(float f) { return seed_cast(f); }

void mcrt_oracle_quantize(const float* rgba, uint8_t* out, size_t n_pixels) {  // image_writer.cpp:18-22
    for (size_t i = 0; i < n_pixels * 4; ++i) {
        float c = sclamp(rgba[i], 0.0f, 1.0f);
        out[i] = static_cast<uint8_t>(c * 255.0f + 0.5f);
    }
}

}  // extern "C"
