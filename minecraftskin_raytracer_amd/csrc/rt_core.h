// rt_core.h — device-side restatement of the reference's per-ray arithmetic for gfx950.
//
// float32, no contraction (compile with -ffp-contract=off); IEEE division and sqrt (hipcc's
// default -fhip-fp32-correctly-rounded-divide-sqrt); std::min/max/clamp as compare-selects, not
// v_min/v_max (different NaN/±0 rules).  Scene data comes from the flat blob (flat_scene.h)
// through wave-uniform indices, so the compiler keeps it in SGPRs (s_load) — the mesh loop is
// uniform across the wave even when the rays are not.
//
// Reference lines followed (under /root/reference/src) are named at each function.
#ifndef MCRT_RT_CORE_H
#define MCRT_RT_CORE_H

#include <hip/hip_runtime.h>

#include "flat_scene.h"
#include "mcrt.h"
#include "mcrt_detmath.h"

#define DEV __device__ __forceinline__
#define DEVNI __device__ __noinline__

namespace rt {

constexpr float kFltMax = 3.402823466e+38f;
constexpr float kPi = 3.14159265358979323846f;
constexpr float kTwoPi = 2.0f * kPi;  // `2.0f * static_cast<float>(M_PI)` folded in float

struct V3 {
    float x, y, z;
};
struct C4 {
    float r, g, b, a;
};

DEV V3 mk(float x, float y, float z) { return V3{x, y, z}; }
DEV V3 ld3(const float* p) { return V3{p[0], p[1], p[2]}; }
DEV V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
DEV V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
DEV V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
DEV V3 vdiv(V3 a, float s) {  // vec3.h:22
    float inv = 1.0f / s;
    return V3{a.x * inv, a.y * inv, a.z * inv};
}
DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DEV V3 cross(V3 a, V3 b) {
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
DEV float length(V3 a) { return __builtin_sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
DEV V3 normalize(V3 a) {  // vec3.h:46-50
    float l = length(a);
    if (l < 1e-8f) return V3{0.0f, 0.0f, 0.0f};
    return vdiv(a, l);
}
DEV float smin(float a, float b) { return (b < a) ? b : a; }
DEV float smax(float a, float b) { return (a < b) ? b : a; }
DEV float sclamp(float v, float lo, float hi) { return (v < lo) ? lo : ((hi < v) ? hi : v); }
DEV int iclamp(int v, int lo, int hi) { return (v < lo) ? lo : ((hi < v) ? hi : v); }
DEV C4 clamp4(C4 c) {
    return C4{sclamp(c.r, 0.0f, 1.0f), sclamp(c.g, 0.0f, 1.0f), sclamp(c.b, 0.0f, 1.0f),
              sclamp(c.a, 0.0f, 1.0f)};
}
DEV float comp(V3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }

struct Ray {
    V3 o, d;
};
struct Hit {
    bool hit;
    bool outer;
    float t;
    V3 p, n;
    C4 tex;
};

// view of the flat blob
struct SceneView {
    const FlatHeader* hdr;
    const FlatMesh* meshes;
    const float4* texels;
    int n_meshes;
};
DEV SceneView view_of(const uint8_t* blob) {
    SceneView s;
    s.hdr = reinterpret_cast<const FlatHeader*>(blob);
    s.meshes = reinterpret_cast<const FlatMesh*>(blob + s.hdr->mesh_offset);
    s.texels = reinterpret_cast<const float4*>(blob + s.hdr->texel_offset);
    s.n_meshes = static_cast<int>(s.hdr->n_meshes);
    return s;
}

// ---------------------------------------------------------------------------------------------
// std::mt19937 — truncated, array-free form for short streams.
// Draw i of a freshly seeded engine is temper(mt[i+397] ^ twist(mt[i], mt[i+1])) for i < 227, and
// mt[j] = 1812433253 * (mt[j-1] ^ (mt[j-1] >> 30)) + j, so two scalar recurrences — one at index
// i, one at index i+397 — produce the stream with no 624-word state.
// ---------------------------------------------------------------------------------------------
DEV uint32_t mt_step(uint32_t prev, uint32_t j) { return 1812433253u * (prev ^ (prev >> 30)) + j; }
DEV uint32_t mt_temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}
DEV uint32_t mt_twist(uint32_t cur, uint32_t nxt, uint32_t far) {
    uint32_t y = (cur & 0x80000000u) | (nxt & 0x7fffffffu);
    uint32_t v = far ^ (y >> 1);
    return (y & 1u) ? (v ^ 0x9908b0dfu) : v;
}
// uniform_real_distribution<float>(0,1): float(x) * 2^-32, >= 1 → nextafter(1, 0)
DEV float mt_to_unit(uint32_t x) {
    float r = static_cast<float>(x) * 0x1p-32f;
    return (r >= 1.0f) ? 0x1.fffffep-1f : r;
}

constexpr int kMtShortMax = 227;  // draws available from the two-recurrence form

struct MtShort {
    uint32_t lo, hi;  // mt[i], mt[i+397]
    uint32_t i;
    DEV void seed(uint32_t s) {
        lo = s;
        uint32_t x = s;
        for (uint32_t j = 1; j <= 397; ++j) x = mt_step(x, j);
        hi = x;
        i = 0;
    }
    DEV float uniform() {
        uint32_t nxt = mt_step(lo, i + 1);
        uint32_t y = mt_temper(mt_twist(lo, nxt, hi));
        lo = nxt;
        hi = mt_step(hi, i + 398);
        ++i;
        return mt_to_unit(y);
    }
};

// Full engine over caller-provided storage, for streams longer than kMtShortMax draws
// (shadowSamples or aoSamples > 113).  Rare path; state lives in a per-thread slice of HBM.
struct MtFull {
    uint32_t* s;
    int pos;
    DEV void seed(uint32_t* storage, uint32_t seedv) {
        s = storage;
        s[0] = seedv;
        for (int j = 1; j < 624; ++j) s[j] = mt_step(s[j - 1], static_cast<uint32_t>(j));
        pos = 624;
    }
    DEV float uniform() {
        if (pos >= 624) {
            for (int k = 0; k < 624; ++k) s[k] = mt_twist(s[k], s[(k + 1) % 624], s[(k + 397) % 624]);
            pos = 0;
        }
        return mt_to_unit(mt_temper(s[pos++]));
    }
};

// per-hit stream: short form in registers, or the full engine when the stream is long
struct HitRng {
    MtShort sh;
    MtFull fu;
    bool full;
    DEV void seed(uint32_t seedv, int draws, uint32_t* storage) {
        full = draws > kMtShortMax;
        if (full)
            fu.seed(storage, seedv);
        else
            sh.seed(seedv);
    }
    DEV float uniform() { return full ? fu.uniform() : sh.uniform(); }
};

// raytracer.cpp:110-112 / :122-123 — static_cast<unsigned>(float) as x86-64 GCC compiles it:
// cvttss2si to 64 bits, low 32 bits kept (two's-complement wrap); out of range / NaN → 0.
DEV uint32_t seed_cast(float f) {
    if (!(f > -0x1p63f && f < 0x1p63f)) return 0u;
    return static_cast<uint32_t>(static_cast<long long>(f));
}

// ---------------------------------------------------------------------------------------------
// intersection.cpp
// ---------------------------------------------------------------------------------------------
// TextureRegion::sample (texture_region.h:19-26) through the face table of the flat mesh
DEV C4 face_texel(const SceneView& sc, const FlatMesh& m, int face, float u, float v) {
    int off = m.tex_off[face];
    if (off == MCRT_TEX_NULL) return C4{1.0f, 0.0f, 1.0f, 1.0f};  // :305
    if (off == MCRT_TEX_EMPTY) return C4{0.0f, 0.0f, 0.0f, 1.0f};
    int w = m.tex_w[face], h = m.tex_h[face];
    int x = iclamp(static_cast<int>(u * w), 0, w - 1);
    int y = iclamp(static_cast<int>(v * h), 0, h - 1);
    float4 t = sc.texels[off + y * w + x];
    return C4{t.x, t.y, t.z, t.w};
}

// determineFace :86-132 → face slot 0..5 and its normal
DEV int face_slot(int axis, bool neg, V3& n) {
    if (axis == 2) {
        n = neg ? mk(0.0f, 0.0f, -1.0f) : mk(0.0f, 0.0f, 1.0f);
        return neg ? 0 : 1;
    }
    if (axis == 0) {
        n = neg ? mk(-1.0f, 0.0f, 0.0f) : mk(1.0f, 0.0f, 0.0f);
        return neg ? 3 : 2;
    }
    n = neg ? mk(0.0f, -1.0f, 0.0f) : mk(0.0f, 1.0f, 0.0f);
    return neg ? 5 : 4;
}

// computeFaceUV :136-196
DEV void face_uv(V3 hp, V3 lo, V3 hi, int axis, bool neg, float& u, float& v) {
    V3 ext = hi - lo;
    float sx = (ext.x > 1e-8f) ? ext.x : 1.0f;
    float sy = (ext.y > 1e-8f) ? ext.y : 1.0f;
    float sz = (ext.z > 1e-8f) ? ext.z : 1.0f;
    if (axis == 2) {
        float lx = (hp.x - lo.x) / sx;
        float ly = (hp.y - lo.y) / sy;
        u = neg ? (1.0f - lx) : lx;
        v = 1.0f - ly;
    } else if (axis == 0) {
        float lz = (hp.z - lo.z) / sz;
        float ly = (hp.y - lo.y) / sy;
        u = neg ? lz : (1.0f - lz);
        v = 1.0f - ly;
    } else {
        float lx = (hp.x - lo.x) / sx;
        float lz = (hp.z - lo.z) / sz;
        u = lx;
        v = neg ? (1.0f - lz) : lz;
    }
    u = sclamp(u, 0.0f, 1.0f);
    v = sclamp(v, 0.0f, 1.0f);
}

// Slab state of one ray against one box (:221-250 plus the exit-face scan :268-285 ≡ :323-335,
// which is the same function of (ray, box) in both places).
struct Slab {
    bool overlap;  // survived the slab loop
    float tmin, tmax;
    int in_axis, out_axis;
    bool in_neg, out_neg;
};

DEV Slab slab_test(const Ray& r, V3 lo, V3 hi) {
    Slab s;
    s.overlap = true;
    s.tmin = -kFltMax;
    s.tmax = kFltMax;
    s.in_axis = 0;
    s.in_neg = false;
    float best_exit = kFltMax;
    s.out_axis = 0;
    s.out_neg = false;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float d = comp(r.d, i), o = comp(r.o, i), l = comp(lo, i), h = comp(hi, i);
        if (__builtin_fabsf(d) < 1e-8f) {
            if (o < l || o > h) s.overlap = false;
        } else {
            float inv = 1.0f / d;
            float t0 = (l - o) * inv;
            float t1 = (h - o) * inv;
            bool swapped = t0 > t1;
            float tn = swapped ? t1 : t0;  // near
            float tf = swapped ? t0 : t1;  // far
            if (tn > s.tmin) {
                s.tmin = tn;
                s.in_axis = i;
                s.in_neg = !swapped;
            }
            s.tmax = smin(s.tmax, tf);
            // the reference tests this inside the loop; tmin only grows and tmax only shrinks, so
            // a failing test at any iteration is equivalent to failing here and staying failed
            if (s.tmin > s.tmax || s.tmax < 0.0f) s.overlap = false;
            if (tf < best_exit) {
                best_exit = tf;
                s.out_axis = i;
                s.out_neg = swapped;
            }
        }
    }
    return s;
}

// intersectAABB :200-371 in the mesh's own space.  `t_limit`: candidates that cannot beat it are
// dropped before the texture fetch (closest-hit keeps only strictly smaller t, :415).
DEV Hit hit_box(const SceneView& sc, const FlatMesh& m, const Ray& r) {
    Hit res;
    res.hit = false;
    res.outer = false;
    res.t = 0.0f;
    res.p = mk(0, 0, 0);
    res.n = mk(0, 0, 0);
    res.tex = C4{0.0f, 0.0f, 0.0f, 1.0f};
    if (m.flags & MESH_EMPTY) return res;
    V3 lo = ld3(m.lo), hi = ld3(m.hi);
    Slab s = slab_test(r, lo, hi);
    if (!s.overlap) return res;

    float tHit = s.tmin;
    int axis = s.in_axis;
    bool neg = s.in_neg;
    if (tHit < 0.0f) {  // origin inside: exit face
        tHit = s.tmax;
        if (tHit < 0.0f) return res;
        axis = s.out_axis;
        neg = s.out_neg;
    }
    V3 hp = r.o + r.d * tHit;
    V3 n;
    int face = face_slot(axis, neg, n);
    float u, v;
    face_uv(hp, lo, hi, axis, neg, u, v);
    C4 tc = face_texel(sc, m, face, u, v);
    const bool outer = (m.flags & MESH_OUTER) != 0;

    if (tc.a == 0.0f) {  // :311-361
        if (!outer) return res;
        if (s.tmax > tHit) {
            V3 bp = r.o + r.d * s.tmax;
            V3 bn;
            int bface = face_slot(s.out_axis, s.out_neg, bn);
            float bu, bv;
            face_uv(bp, lo, hi, s.out_axis, s.out_neg, bu, bv);
            C4 bc = face_texel(sc, m, bface, bu, bv);
            if (bc.a > 0.0f) {
                res.hit = true;
                res.t = s.tmax;
                res.p = bp;
                res.n = bn * -1.0f;
                res.tex = bc;
                res.outer = true;
            }
        }
        return res;
    }
    res.hit = true;
    res.t = tHit;
    res.p = hp;
    res.n = n;
    res.tex = tc;
    res.outer = outer;
    return res;
}

// rotatePoint :12-37 with the trig hoisted to the flat mesh
DEV V3 spin(V3 p, V3 pivot, bool ax, float cx, float sx, bool az, float cz, float sz) {
    V3 q = p - pivot;
    if (ax) {
        float ny = q.y * cx - q.z * sx;
        float nz = q.y * sx + q.z * cx;
        q.y = ny;
        q.z = nz;
    }
    if (az) {
        float nx = q.x * cz - q.y * sz;
        float ny = q.x * sz + q.y * cz;
        q.x = nx;
        q.y = ny;
    }
    return q + pivot;
}

// intersectMesh :373-406
DEV Hit hit_mesh(const SceneView& sc, const FlatMesh& m, const Ray& r) {
    if (!(m.flags & MESH_ROTATED)) return hit_box(sc, m, r);
    const bool ax = (m.flags & MESH_APPLY_X) != 0, az = (m.flags & MESH_APPLY_Z) != 0;
    V3 pivot = ld3(m.pivot);
    V3 zero = mk(0.0f, 0.0f, 0.0f);
    V3 lo = spin(r.o, pivot, false, 1.0f, 0.0f, az, m.inv_z_cos, m.inv_z_sin);
    lo = spin(lo, pivot, ax, m.inv_x_cos, m.inv_x_sin, false, 1.0f, 0.0f);
    V3 ld = spin(r.d, zero, false, 1.0f, 0.0f, az, m.inv_z_cos, m.inv_z_sin);
    ld = spin(ld, zero, ax, m.inv_x_cos, m.inv_x_sin, false, 1.0f, 0.0f);
    Ray local{lo, normalize(ld)};
    Hit h = hit_box(sc, m, local);
    if (h.hit) {
        h.p = spin(h.p, pivot, ax, m.fwd_x_cos, m.fwd_x_sin, az, m.fwd_z_cos, m.fwd_z_sin);
        h.n = normalize(spin(h.n, zero, ax, m.fwd_x_cos, m.fwd_x_sin, az, m.fwd_z_cos, m.fwd_z_sin));
        h.t = dot(h.p - r.o, r.d);
    }
    return h;
}

// intersectScene :408-421.  mesh_mask: bit i set → mesh i is tested (primary-ray culling; all
// ones for secondary rays).  Meshes beyond bit 63 are always tested.
DEV Hit hit_scene(const SceneView& sc, const Ray& r, uint64_t mesh_mask) {
    Hit best;
    best.hit = false;
    best.outer = false;
    best.t = kFltMax;
    best.p = mk(0, 0, 0);
    best.n = mk(0, 0, 0);
    best.tex = C4{0.0f, 0.0f, 0.0f, 1.0f};
    for (int i = 0; i < sc.n_meshes; ++i) {
        if (i < 64 && !((mesh_mask >> i) & 1ull)) continue;
        Hit h = hit_mesh(sc, sc.meshes[i], r);
        if (h.hit && h.t < best.t) best = h;
    }
    return best;
}

// "hit && t < limit" over the scene without keeping the hit: isInShadow :25 and computeAO :72.
// The reference finds the closest hit first; min t < limit ⇔ some t < limit, so the scan may
// stop at the first mesh that qualifies.
DEV bool any_hit_before(const SceneView& sc, const Ray& r, float limit) {
    for (int i = 0; i < sc.n_meshes; ++i) {
        Hit h = hit_mesh(sc, sc.meshes[i], r);
        if (h.hit && h.t < limit) return true;
    }
    return false;
}

// ---------------------------------------------------------------------------------------------
// shading.cpp
// ---------------------------------------------------------------------------------------------
DEV bool in_shadow(const SceneView& sc, V3 point, V3 normal, V3 light) {  // :14-26
    V3 origin = point + normal * 1e-3f;
    V3 to = light - origin;
    float dist = length(to);
    if (dist < 1e-6f) return false;
    Ray r{origin, vdiv(to, dist)};
    return any_hit_before(sc, r, dist);
}

DEV float soft_shadow(const SceneView& sc, V3 point, V3 normal, int samples, uint32_t seed,
                      uint32_t* mt_storage) {  // :28-60
    V3 lpos = ld3(sc.hdr->light_pos);
    float radius = sc.hdr->light_radius;
    if (samples <= 1 || radius < 1e-4f) return in_shadow(sc, point, normal, lpos) ? 0.0f : 1.0f;
    V3 toPoint = normalize(point - lpos);
    V3 tangent = (__builtin_fabsf(toPoint.x) < 0.9f) ? normalize(cross(mk(1, 0, 0), toPoint))
                                                     : normalize(cross(mk(0, 1, 0), toPoint));
    V3 bitangent = cross(toPoint, tangent);
    HitRng rng;
    rng.seed(seed, 2 * samples, mt_storage);
    int lit = 0;
    for (int i = 0; i < samples; ++i) {
        float angle = kTwoPi * rng.uniform();
        float rr = radius * __builtin_sqrtf(rng.uniform());
        V3 off = tangent * (rr * mcrt_cosf(angle)) + bitangent * (rr * mcrt_sinf(angle));
        if (!in_shadow(sc, point, normal, lpos + off)) ++lit;
    }
    return static_cast<float>(lit) / static_cast<float>(samples);
}

// shade :62-96 with ShadingParams{} (kd .75, ks .15, ambient .20, shininess 16 — shading.h:9-14;
// renderTile always passes the defaults, tile_renderer.cpp:106-107)
DEV C4 shade(const SceneView& sc, const Hit& hit, V3 viewDir, float shadowFactor) {
    const float kd = 0.75f, ks = 0.15f, ambient = 0.20f, shininess = 16.0f;
    C4 tex = hit.tex;
    V3 lpos = ld3(sc.hdr->light_pos);
    const float* lc = sc.hdr->light_color;
    V3 L = normalize(lpos - hit.p);
    V3 N = normalize(hit.n);
    V3 V = normalize(viewDir);
    float vis = shadowFactor;
    if (vis < 0.0f) vis = in_shadow(sc, hit.p, N, lpos) ? 0.0f : 1.0f;
    float ndl = smax(0.0f, dot(N, L));
    float kdiff = kd * ndl * vis;
    V3 H = normalize(L + V);
    float ndh = smax(0.0f, dot(N, H));
    float kspec = ks * mcrt_powf(ndh, shininess) * vis;
    C4 out;
    out.r = tex.r * ambient + tex.r * lc[0] * kdiff + lc[0] * kspec;
    out.g = tex.g * ambient + tex.g * lc[1] * kdiff + lc[1] * kspec;
    out.b = tex.b * ambient + tex.b * lc[2] * kdiff + lc[2] * kspec;
    out.a = tex.a;
    return clamp4(out);
}

// ---------------------------------------------------------------------------------------------
// raytracer.cpp
// ---------------------------------------------------------------------------------------------
DEV C4 background(const SceneView& sc, const mcrt_config& cfg, float u, float v) {  // :16-34
    if (cfg.gradient_bg) {
        float cx = u - 0.5f, cy = v - 0.5f;
        float dist = __builtin_sqrtf(cx * cx + cy * cy) * 2.0f * cfg.gradient_scale;
        dist = sclamp(dist, 0.0f, 1.0f);
        float t = dist * dist;
        C4 c;
        c.r = cfg.bg_center[0] * (1.0f - t) + cfg.bg_edge[0] * t;
        c.g = cfg.bg_center[1] * (1.0f - t) + cfg.bg_edge[1] * t;
        c.b = cfg.bg_center[2] * (1.0f - t) + cfg.bg_edge[2] * t;
        c.a = 1.0f;
        return c;
    }
    const float* b = sc.hdr->background;
    return C4{b[0], b[1], b[2], b[3]};
}

DEV float ambient_occlusion(const SceneView& sc, V3 point, V3 normal, int samples, float radius,
                            uint32_t seed, uint32_t* mt_storage) {  // :38-78
    V3 N = normalize(normal);
    V3 T = (__builtin_fabsf(N.x) < 0.9f) ? normalize(cross(mk(1, 0, 0), N)) : normalize(cross(mk(0, 1, 0), N));
    V3 B = cross(N, T);
    HitRng rng;
    rng.seed(seed, 2 * samples, mt_storage);
    int occluded = 0;
    for (int i = 0; i < samples; ++i) {
        float r1 = rng.uniform();
        float r2 = rng.uniform();
        float sinT = __builtin_sqrtf(1.0f - r1);
        float cosT = __builtin_sqrtf(r1);
        float phi = kTwoPi * r2;
        V3 local = mk(sinT * mcrt_cosf(phi), cosT, sinT * mcrt_sinf(phi));
        V3 world = normalize(T * local.x + N * local.y + B * local.z);
        Ray r{point + N * 1e-3f, world};
        if (any_hit_before(sc, r, radius)) ++occluded;
    }
    return 1.0f - static_cast<float>(occluded) / static_cast<float>(samples);
}

// Shading of one hit level: everything traceRay does at a hit before recursing (:104-131).
DEV C4 shade_level(const SceneView& sc, const mcrt_config& cfg, const Ray& ray, const Hit& hit,
                   int depth, uint32_t* mt_storage) {
    V3 view = normalize(ray.o - hit.p);
    float shadowFactor = -1.0f;
    if (cfg.soft_shadows && cfg.shadow_samples > 1) {
        uint32_t seed = seed_cast(hit.p.x * 12345.0f + hit.p.y * 67890.0f + hit.p.z * 11111.0f +
                                  static_cast<float>(depth) * 99999.0f);
        shadowFactor = soft_shadow(sc, hit.p, hit.n, cfg.shadow_samples, seed, mt_storage);
    }
    C4 c = shade(sc, hit, view, shadowFactor);
    if (cfg.ao_enabled && depth == 0) {
        uint32_t seed = seed_cast(hit.p.x * 73856093.0f + hit.p.y * 19349663.0f + hit.p.z * 83492791.0f);
        float ao = ambient_occlusion(sc, hit.p, hit.n, cfg.ao_samples, cfg.ao_radius, seed, mt_storage);
        float k = 1.0f - cfg.ao_intensity * (1.0f - ao);
        c.r *= k;
        c.g *= k;
        c.b *= k;
    }
    return c;
}

// reflection ray of a hit (:133-139)
DEV Ray reflect_ray(const Ray& ray, const Hit& hit) {
    V3 N = normalize(hit.n);
    V3 D = normalize(ray.d);
    V3 R = normalize(D - N * (2.0f * dot(D, N)));
    return Ray{hit.p + N * 1e-3f, R};
}

// One fold step of the recursion unwinding (:143-147): level colour `c` (alpha = texel alpha)
// combined with the colour returned by the deeper level.
DEV C4 fold_reflection(C4 c, C4 deeper) {
    const float keep = 1.0f - 0.1f, refl = 0.1f;  // SKIN_REFLECTIVITY, raytracer.cpp:11
    C4 o;
    o.r = c.r * keep + deeper.r * refl;
    o.g = c.g * keep + deeper.g * refl;
    o.b = c.b * keep + deeper.b * refl;
    o.a = c.a;  // shadedColor.a = originalAlpha
    return clamp4(o);
}

constexpr int kMaxStack = 16;  // levels kept in registers/LDS-free scratch; deeper → global stack

// RayTracer::traceRay (:82-148) for a ray whose depth-`depth` hit is already known, as a loop:
// walk down while rays keep hitting (level colours pushed), then fold back to front.
// `stack` is caller-provided storage for (max_bounces + 1) C4 entries.
DEV C4 trace_from_hit(const SceneView& sc, const mcrt_config& cfg, Ray ray, Hit hit, int depth,
                      C4* stack, uint32_t* mt_storage) {
    const float* b = sc.hdr->background;
    const C4 flat_bg{b[0], b[1], b[2], b[3]};
    const int max_b = cfg.max_bounces;
    int top = 0;
    C4 tail;  // colour returned by the level below the last pushed one
    for (;;) {
        C4 c = shade_level(sc, cfg, ray, hit, depth, mt_storage);
        if (depth >= max_b) {  // no reflection: `shadedColor.a = originalAlpha; return clamp()`
            tail = clamp4(c);
            break;
        }
        stack[top++] = c;
        ray = reflect_ray(ray, hit);
        ++depth;
        hit = hit_scene(sc, ray, ~0ull);
        if (!hit.hit) {  // bounced ray missed → flat scene.backgroundColor (:94-102, depth > 0)
            tail = flat_bg;
            break;
        }
    }
    while (top > 0) tail = fold_reflection(stack[--top], tail);
    return tail;
}

// ---------------------------------------------------------------------------------------------
// camera.cpp:8-26 and tile_renderer.cpp:42-69
// ---------------------------------------------------------------------------------------------
DEV Ray camera_ray(const SceneView& sc, float u, float v, float aspect) {
    const FlatHeader* h = sc.hdr;
    float halfH = h->cam_half_h;
    float halfW = halfH * aspect;
    float su = (2.0f * u - 1.0f) * halfW;
    float sv = (2.0f * (1.0f - v) - 1.0f) * halfH;
    V3 dir = normalize(ld3(h->cam_fwd) + ld3(h->cam_right) * su + ld3(h->cam_up) * sv);
    return Ray{ld3(h->cam_pos), dir};
}

DEV Ray lens_ray(const SceneView& sc, float u, float v, float aspect, float aperture, float focusDist,
                 float d0, float d1) {
    Ray pin = camera_ray(sc, u, v, aspect);
    if (aperture < 1e-6f) return pin;
    const FlatHeader* h = sc.hdr;
    V3 focus = pin.o + pin.d * focusDist;
    float angle = kTwoPi * d0;
    float radius = aperture * __builtin_sqrtf(d1);
    float lx = radius * mcrt_cosf(angle);
    float ly = radius * mcrt_sinf(angle);
    V3 origin = ld3(h->cam_pos) + (ld3(h->cam_right) * lx + ld3(h->cam_up) * ly);
    return Ray{origin, normalize(focus - origin)};
}

}  // namespace rt

#endif
