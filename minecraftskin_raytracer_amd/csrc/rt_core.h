// rt_core.h — device-side restatement of the reference's per-ray arithmetic for gfx950.
//
// float32, no contraction (compile with -ffp-contract=off); IEEE division and sqrt (hipcc's
// default -fhip-fp32-correctly-rounded-divide-sqrt); std::min/max/clamp as compare-selects, not
// v_min/v_max (different NaN/±0 rules).  Scene data comes from the flat blob (flat_scene.h)
// through wave-uniform indices, so the compiler keeps it in SGPRs (s_load) — the mesh loop is
// uniform across the wave even when the rays are not.
//
// Legal restructurings used here (all leave every produced float bit-identical):
//   * 1/d per axis is a function of the ray only → computed once per ray, not once per mesh;
//   * a candidate whose entry distance cannot beat the current limit (closest-hit keeps strictly
//     smaller t, intersection.cpp:415; shadow/AO rays ask "t < limit") is dropped before any UV or
//     texel work;
//   * hit/miss of a face depends on the texel only through (alpha == 0) and (alpha > 0): both
//     predicates are precomputed per texel (2 bits, flatten.cpp) — the colour is fetched once, for
//     the winning hit;
//   * the hit normal is produced for the winning hit only.
// Reference lines followed (under /root/reference/src) are named at each function.
#ifndef MCRT_RT_CORE_H
#define MCRT_RT_CORE_H

#include <hip/hip_runtime.h>

#include "flat_scene.h"
#include "mcrt.h"
#include "mcrt_detmath.h"

#define DEV __device__ __forceinline__
// Shared (non-inlined) routines: one copy of the mesh loop and of the libm kernels in the code
// object.  A fully inlined single render kernel (round 1's first design) was 85 KB of code — larger than the 64 KB instruction
// cache CUs share — and ran instruction-fetch bound.
#define DEVCALL __device__ __noinline__

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
// 1.0f / x correctly rounded (== the IEEE division) for every float x with 2^-126 <= |x| <= 2^126:
// hardware reciprocal estimate (1 ulp) and one Newton step with fused multiply-adds — 3 instructions
// instead of the ~10 of the general division expansion.  Not a numerical argument: the equality is
// checked for EVERY float of that range on the device (mcrt_probe_detmath_range op 5,
// tests/test_gpu_parity.py).  The callers' operands — ray direction components that matter
// (|d| >= 1e-8), vector lengths in [1e-8, 2e19], light distances >= 1e-6 — lie inside it.
DEV float rcp_exact(float x) {
    const float r = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
// x / d for a divisor that stays the same for a whole frame — its width and height, `(px + jitter) / width`
// (tile_renderer.cpp:88-89), two divisions per sample — given rd = 1.0f / d rounded correctly (formed on the host): the
// product with the reciprocal and ONE Newton correction with fused multiply-adds (the residual x - q·d of a nearly
// correct quotient is exact), 3 instructions instead of the ~11 of the general expansion.  Not a numerical argument:
// for every integer divisor up to kDivFrameMax (kernels.h) the equality with the IEEE quotient has been checked on the device for
// EVERY float x a sample coordinate can take (0 and 2^-33 … d + 1; 6·10¹² quotients: tools/gpu_verify_div.py,
// mcrt_probe_div_const, result under profiles/) — larger frames take the general division.  (div_frame2, with the
// second correction that Markstein's theorem asks for in general, passes the same check and is not needed.)
DEV float div_frame(float x, float d, float rd) {
    const float q0 = x * rd;
    return __builtin_fmaf(__builtin_fmaf(-q0, d, x), rd, q0);
}
// sqrtf for x = 0 and every x >= 2^-96 (infinity included), as the compiler forms the correctly rounded root — hardware
// estimate within one ulp, then the choice among its two neighbours by the signs of the fused residuals — without the
// rescaling of arguments below 2^-96 and the class test that the general expansion spends 7 of its 16 instructions on.
// Checked against it for EVERY float of that range on the device (mcrt_probe_div_const mode 3,
// test_frame_division_is_the_ieee_division).  Below 2^-96 the result is merely some number below 2^-47, which is
// all the callers need there: squared lengths that small fall under normalize's 1e-8 and isInShadow's 1e-6
// thresholds (vec3.h:46-50, shading.cpp:19), and the other arguments — the gradient's cx² + cy², draws k·2^-32 and
// 1 - draw — are 0 or at least 2^-50.
DEV float sqrt_pos(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float lo = __uint_as_float(__float_as_uint(s) - 1u), hi = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_lo = __builtin_fmaf(-lo, s, x), r_hi = __builtin_fmaf(-hi, s, x);
    float r = (0.0f >= r_lo) ? lo : s;
    r = (0.0f < r_hi) ? hi : r;
    return r;
}
DEV float div_frame2(float x, float d, float rd) {  // probe only
    const float q1 = div_frame(x, d, rd);
    return __builtin_fmaf(__builtin_fmaf(-q1, d, x), rd, q1);
}
DEV V3 vdiv(V3 a, float s) {  // vec3.h:22 — callers guarantee 1e-8 <= s (normalize, isInShadow)
    float inv = rcp_exact(s);
    return V3{a.x * inv, a.y * inv, a.z * inv};
}
DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DEV V3 cross(V3 a, V3 b) {
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
DEV float length(V3 a) { return sqrt_pos(a.x * a.x + a.y * a.y + a.z * a.z); }
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
    // how the hit was found, for compact hit records: the texel reference of the face (pool index or
    // MCRT_TEX_NULL / MCRT_TEX_EMPTY) and the face in the mesh's local frame — for an un-posed mesh the normal
    // is face_normal(axis, neg), negated when `back` (the outer layer's exit face, intersection.cpp:349-357)
    int texel;
    int axis;
    bool neg, back;
};

// Views of the flat blob.  SceneView reads everything from HBM (probes, rare sequential paths).
// SceneViewLds is what the ray-tracing kernels use: the two tables that candidates index PER LANE — the
// alpha predicates and the face → texture table — live in LDS and are typed as LDS pointers, so
// the lookups compile to ds_read (a pointer that may be either LDS or global becomes a flat load).
#define MCRT_LDS __attribute__((address_space(3)))
struct SceneView {
    static constexpr bool kLds = false;
    static constexpr bool kPosed = true;
    const FlatHeader* hdr;
    const FlatMesh* meshes;
    const float4* texels;
    const uint32_t* abits;
    unsigned long long roots;  // FlatHeader::root_lo/hi
    int n_meshes;
    DEV SceneView global() const { return *this; }
};
// kPosedT = false: the scene holds no posed (hasRotation) mesh — selected per scene on the host; the
// local-frame ray, the forward rotation of hit point/normal and the bounding-sphere pre-test drop
// out of the code and of the register budget.
template <bool kPosedT>
struct SceneViewLdsT {
    static constexpr bool kLds = true;
    static constexpr bool kPosed = kPosedT;
    const FlatHeader* hdr;
    const FlatMesh* meshes;
    const float4* texels;
    const uint32_t* abits_hbm;
    const MCRT_LDS uint32_t* abits;  // 16 texels per word
    const MCRT_LDS int* faces;       // per (mesh, face slot): {texel offset | MCRT_TEX_*, width, height, 0}
    const MCRT_LDS float* mtab;      // per mesh kMeshTabWords words: lo hi flags . pivot . trig[8]
    unsigned long long roots;
    int n_meshes;
    DEV SceneView global() const { return SceneView{hdr, meshes, texels, abits_hbm, roots, n_meshes}; }
};
using SceneViewLds = SceneViewLdsT<true>;
constexpr int kMeshTabWords = 24;
DEV SceneView view_of(const uint8_t* blob) {
    SceneView s;
    s.hdr = reinterpret_cast<const FlatHeader*>(blob);
    s.meshes = reinterpret_cast<const FlatMesh*>(blob + s.hdr->mesh_offset);
    s.texels = reinterpret_cast<const float4*>(blob + s.hdr->texel_offset);
    s.abits = reinterpret_cast<const uint32_t*>(blob + s.hdr->alpha_offset);
    s.roots = (static_cast<unsigned long long>(s.hdr->root_hi) << 32) | s.hdr->root_lo;
    s.n_meshes = static_cast<int>(s.hdr->n_meshes);
    return s;
}
template <bool kPosedT>
DEV SceneViewLdsT<kPosedT> view_with_lds(const SceneView& g, const MCRT_LDS uint32_t* abits, const MCRT_LDS int* faces,
                                         const MCRT_LDS float* mtab) {
    return SceneViewLdsT<kPosedT>{g.hdr, g.meshes, g.texels, g.abits, abits, faces, mtab, g.roots, g.n_meshes};
}

DEVCALL float dev_sinf(float x) { return mcrt_sinf(x); }
DEVCALL float dev_cosf(float x) { return mcrt_cosf(x); }
struct SinCos {
    float s, c;
};
// sinf and cosf of one angle: shared argument reduction, same bits as the two separate calls
DEVCALL SinCos dev_sincosf(float x) {
    SinCos r;
    mcrt_sincosf(x, &r.s, &r.c);
    return r;
}
DEVCALL float dev_powf(float x, float y) { return mcrt_powf(x, y); }

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
    const float r = static_cast<float>(x) * 0x1p-32f;  // in [0, 1]
    return __builtin_fminf(r, 0x1.fffffep-1f);         // (r >= 1) ? nextafter(1, 0) : r
}

constexpr int kMtShortMax = 227;  // draws available from the two-recurrence form

struct MtShort {
    uint32_t lo, hi;  // mt[i], mt[i+397]
    uint32_t i;
    DEV static uint32_t word397(uint32_t s) {  // mt[397] of a freshly seeded engine: the sequential part
        uint32_t x = s;
        for (uint32_t j = 1; j <= 397; ++j) x = mt_step(x, j);
        return x;
    }
    DEV void seed(uint32_t s) { seed_known(s, word397(s)); }
    DEV void seed_known(uint32_t s, uint32_t mt397) {  // mt397 = word397(s), e.g. from the device's seed table
        lo = s;
        hi = mt397;
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
    // advance past one draw without producing it (a lane that takes draws 2j, 2j+1 of a hit's stream)
    DEV void skip() {
        lo = mt_step(lo, i + 1);
        hi = mt_step(hi, i + 398);
        ++i;
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
DEV uint32_t shadow_seed(V3 p, int depth) {  // raytracer.cpp:110-112
    return seed_cast(p.x * 12345.0f + p.y * 67890.0f + p.z * 11111.0f + static_cast<float>(depth) * 99999.0f);
}
DEV uint32_t ao_seed(V3 p) {  // raytracer.cpp:122-123
    return seed_cast(p.x * 73856093.0f + p.y * 19349663.0f + p.z * 83492791.0f);
}

// ---------------------------------------------------------------------------------------------
// intersection.cpp
// ---------------------------------------------------------------------------------------------
// A ray prepared for slab tests: the reciprocals of intersection.cpp:229 hoisted out of the mesh loop.
struct RayQ {
    V3 o, d, inv;
    bool px, py, pz;  // |d| < 1e-8 on that axis (:222)
};
DEV RayQ prepare(const Ray& r) {
    RayQ q;
    q.o = r.o;
    q.d = r.d;
    q.px = __builtin_fabsf(r.d.x) < 1e-8f;
    q.py = __builtin_fabsf(r.d.y) < 1e-8f;
    q.pz = __builtin_fabsf(r.d.z) < 1e-8f;
    q.inv.x = q.px ? 0.0f : rcp_exact(r.d.x);  // |d| >= 1e-8 where the value is used
    q.inv.y = q.py ? 0.0f : rcp_exact(r.d.y);
    q.inv.z = q.pz ? 0.0f : rcp_exact(r.d.z);
    return q;
}

// texel reference of a face at (u,v): pool index, or MCRT_TEX_NULL / MCRT_TEX_EMPTY
// (TextureRegion::sample, texture_region.h:19-26)
template <class SV>
DEV int face_texel_index(const SV& sc, int mesh_index, int face, float u, float v) {
    int off, w, h;
    if constexpr (SV::kLds) {
        const MCRT_LDS int* f = sc.faces + (mesh_index * 6 + face) * 4;
        off = f[0], w = f[1], h = f[2];
    } else {
        const FlatMesh& m = sc.meshes[mesh_index];
        off = m.tex_off[face], w = m.tex_w[face], h = m.tex_h[face];
    }
    if (off < 0) return off;
    int x = iclamp(static_cast<int>(u * w), 0, w - 1);
    int y = iclamp(static_cast<int>(v * h), 0, h - 1);
    return off + y * w + x;
}
// bit0: alpha == 0.0f, bit1: alpha > 0.0f.  nullptr texture → magenta, empty → Color(): alpha 1.
template <class SV>
DEV uint32_t alpha_bits(const SV& sc, int texel) {
    if (texel < 0) return 2u;
    return (sc.abits[texel >> 4] >> ((texel & 15) * 2)) & 3u;
}
template <class SV>
DEV C4 texel_color(const SV& sc, int texel) {
    if (texel == MCRT_TEX_NULL) return C4{1.0f, 0.0f, 1.0f, 1.0f};  // intersection.cpp:305
    if (texel < 0) return C4{0.0f, 0.0f, 0.0f, 1.0f};
    float4 t = sc.texels[texel];
    return C4{t.x, t.y, t.z, t.w};
}

// determineFace :86-132 → face slot 0..5
DEV int face_slot(int axis, bool neg) {
    if (axis == 2) return neg ? 0 : 1;
    if (axis == 0) return neg ? 3 : 2;
    return neg ? 5 : 4;
}
DEV V3 face_normal(int axis, bool neg) {
    float s = neg ? -1.0f : 1.0f;
    return mk(axis == 0 ? s : 0.0f, axis == 1 ? s : 0.0f, axis == 2 ? s : 0.0f);
}

// the hit normal intersectMesh returns for an un-posed mesh (:297-303 and, for the exit face, :354)
DEV V3 unposed_normal(int axis, bool neg, bool back) {
    V3 n = face_normal(axis, neg);
    if (back) n = n * -1.0f;
    return n;
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
    bool overlap;
    float tmin, tmax;
    int in_axis, out_axis;
    bool in_neg, out_neg;
};

DEV void slab_axis(Slab& s, float& best_exit, int i, bool par, float o, float inv, float l, float h) {
    if (par) {
        if (o < l || o > h) s.overlap = false;
        return;
    }
    float t0 = (l - o) * inv;
    float t1 = (h - o) * inv;
    bool swapped = t0 > t1;
    float tn = swapped ? t1 : t0;
    float tf = swapped ? t0 : t1;
    if (tn > s.tmin) {
        s.tmin = tn;
        s.in_axis = i;
        s.in_neg = !swapped;
    }
    s.tmax = smin(s.tmax, tf);
    // the reference tests this inside the loop; tmin only grows and tmax only shrinks, so a failure
    // at any iteration is equivalent to failing here and staying failed
    if (s.tmin > s.tmax || s.tmax < 0.0f) s.overlap = false;
    if (tf < best_exit) {
        best_exit = tf;
        s.out_axis = i;
        s.out_neg = swapped;
    }
}
DEV Slab slab_test(const RayQ& r, V3 lo, V3 hi) {
    Slab s;
    s.overlap = true;
    s.tmin = -kFltMax;
    s.tmax = kFltMax;
    s.in_axis = 0;
    s.in_neg = false;
    s.out_axis = 0;
    s.out_neg = false;
    float best_exit = kFltMax;
    slab_axis(s, best_exit, 0, r.px, r.o.x, r.inv.x, lo.x, hi.x);
    slab_axis(s, best_exit, 1, r.py, r.o.y, r.inv.y, lo.y, hi.y);
    slab_axis(s, best_exit, 2, r.pz, r.o.z, r.inv.z, lo.z, hi.z);
    return s;
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
// The per-mesh data a ray test reads, in registers
struct MeshData {
    V3 lo, hi, pivot;
    V3 centre;      // world-space bounding sphere
    float radius;   // padded; < 0 → no bound
    uint32_t flags;
    float inv_z_cos, inv_z_sin, inv_x_cos, inv_x_sin, fwd_x_cos, fwd_x_sin, fwd_z_cos, fwd_z_sin;
    unsigned long long group;  // first-pass group of a root (mesh_uniform only)
};
// wave-uniform mesh index → scalar loads from the blob
template <class SV>
DEV MeshData mesh_uniform(const SV& sc, int i) {
    const FlatMesh& m = sc.meshes[i];
    return MeshData{ld3(m.lo),   ld3(m.hi),   ld3(m.pivot), ld3(m.sphere), m.sphere[3], m.flags,     m.inv_z_cos,
                    m.inv_z_sin, m.inv_x_cos, m.inv_x_sin,  m.fwd_x_cos,   m.fwd_x_sin, m.fwd_z_cos, m.fwd_z_sin,
                    (static_cast<unsigned long long>(m.group_hi) << 32) | m.group_lo};
}
// per-lane mesh index → the LDS mesh table (ray-tracing kernels) or vector loads from the blob
template <class SV>
DEV MeshData mesh_lane(const SV& sc, int i) {
    if constexpr (SV::kLds) {
        const MCRT_LDS float* t = sc.mtab + i * kMeshTabWords;
        MeshData d;
        d.lo = mk(t[0], t[1], t[2]);
        d.hi = mk(t[3], t[4], t[5]);
        d.flags = __float_as_uint(t[6]);
        d.pivot = mk(t[8], t[9], t[10]);
        d.inv_z_cos = t[12], d.inv_z_sin = t[13], d.inv_x_cos = t[14], d.inv_x_sin = t[15];
        d.fwd_x_cos = t[16], d.fwd_x_sin = t[17], d.fwd_z_cos = t[18], d.fwd_z_sin = t[19];
        d.centre = mk(t[20], t[21], t[22]);
        d.radius = t[23];
        d.group = 0ull;
        return d;
    } else {
        return mesh_uniform(sc, i);
    }
}

DEV V3 to_world(const MeshData& m, V3 p) {
    return spin(p, m.pivot, (m.flags & MESH_APPLY_X) != 0, m.fwd_x_cos, m.fwd_x_sin, (m.flags & MESH_APPLY_Z) != 0,
                m.fwd_z_cos, m.fwd_z_sin);
}
// the world-space ray in the posed mesh's local frame (intersection.cpp:384-393)
DEV RayQ to_local(const MeshData& m, const RayQ& world) {
    const bool ax = (m.flags & MESH_APPLY_X) != 0, az = (m.flags & MESH_APPLY_Z) != 0;
    const V3 zero = mk(0.0f, 0.0f, 0.0f);
    V3 o = spin(world.o, m.pivot, false, 1.0f, 0.0f, az, m.inv_z_cos, m.inv_z_sin);
    o = spin(o, m.pivot, ax, m.inv_x_cos, m.inv_x_sin, false, 1.0f, 0.0f);
    V3 d = spin(world.d, zero, false, 1.0f, 0.0f, az, m.inv_z_cos, m.inv_z_sin);
    d = spin(d, zero, ax, m.inv_x_cos, m.inv_x_sin, false, 1.0f, 0.0f);
    return prepare(Ray{o, normalize(d)});
}

// Phase 1 of a scene query: can this mesh matter at all?  Same slab arithmetic as slab_test without
// the face bookkeeping: overlap (:221-250), the entry/exit distance (:254-259) and — for un-posed
// meshes, whose local t is the reported t — the distance limit.
DEV void quick_axis(bool& ok, float& tmin, float& tmax, bool par, float o, float inv, float l, float h) {
    if (par) {
        if (o < l || o > h) ok = false;
        return;
    }
    const float t0 = (l - o) * inv, t1 = (h - o) * inv;
    const bool swapped = t0 > t1;
    const float tn = swapped ? t1 : t0, tf = swapped ? t0 : t1;
    if (tn > tmin) tmin = tn;
    tmax = smin(tmax, tf);
    if (tmin > tmax || tmax < 0.0f) ok = false;
}
template <bool kPosed>
DEV bool mesh_may_hit(const MeshData& m, const RayQ& world, float t_limit) {
    if (m.flags & MESH_EMPTY) return false;
    if (kPosed && (m.flags & MESH_ROTATED)) {
        // posed mesh: the exact test needs the ray in the mesh's frame (phase 2).  Here only a
        // conservative bounding-sphere test — every point of the posed box lies within `radius` of
        // `centre`, so a ray that misses the padded sphere, or leaves it behind, cannot hit the box.
        // |oc x d|^2 <= r^2 |d|^2  written without division; generous slack for float rounding.
        if (m.radius < 0.0f) return true;
        const V3 oc = m.centre - world.o;
        const float dd = dot(world.d, world.d), b = dot(oc, world.d), oc2 = dot(oc, oc);
        const float r2 = m.radius * m.radius;
        if (oc2 * dd - b * b > r2 * dd + 1e-4f * oc2 * dd + 1e-6f) return false;
        if (b < 0.0f && oc2 > r2 * 1.001f + 1e-3f) return false;  // sphere entirely behind the origin
        return true;
    }
    bool ok = true;
    float tmin = -kFltMax, tmax = kFltMax;
    quick_axis(ok, tmin, tmax, world.px, world.o.x, world.inv.x, m.lo.x, m.hi.x);
    quick_axis(ok, tmin, tmax, world.py, world.o.y, world.inv.y, m.lo.y, m.hi.y);
    quick_axis(ok, tmin, tmax, world.pz, world.o.z, world.inv.z, m.lo.z, m.hi.z);
    if (!ok) return false;  // includes tmax < 0: the box lies behind the origin
    // un-posed: local t is the reported t, so nothing of this box — or of a box inside it — can be
    // hit before the limit unless the ray enters before the limit (origin inside: tmin < 0)
    return tmin < t_limit;
}

// What a mesh contributes to a ray query: intersectMesh (:373-406) over intersectAABB (:200-371)
// without the texel colour and without the normal vector.
struct Cand {
    float t;    // HitResult::t (world-space for posed meshes, :402)
    V3 p;       // HitResult::point (world space)
    int texel;  // texel reference of the face that was hit
    int axis;   // face: axis / min-side, in the mesh's local frame
    bool neg;
    bool back;  // outer-layer exit face: normal flipped, isOuterLayer forced (:349-357)
};

// Phase 2: full evaluation of one mesh.  Returns true when it yields a hit with t < t_limit.
// kAnyHit: the caller only asks whether (shadow and AO rays) — c is not filled, and a mesh without
// transparent texels (MESH_OPAQUE: the inner layer of a skin) is decided by its slab test alone.
template <bool kAnyHit, class SV>
DEV bool mesh_candidate(const SV& sc, const MeshData& m, int mesh_index, const RayQ& world, float t_limit, Cand& c) {
    if (m.flags & MESH_EMPTY) return false;
    const bool rotated = SV::kPosed && (m.flags & MESH_ROTATED) != 0;
    const V3 lo = m.lo, hi = m.hi;
    RayQ local = world;
    if (rotated) local = to_local(m, world);
    const Slab s = slab_test(local, lo, hi);
    if (!s.overlap) return false;

    float tHit = s.tmin;
    int axis = s.in_axis;
    bool neg = s.in_neg;
    if (tHit < 0.0f) {  // origin inside the box: leave through the exit face (:255-288)
        tHit = s.tmax;
        if (tHit < 0.0f) return false;
        axis = s.out_axis;
        neg = s.out_neg;
    }
    // un-posed meshes: local t IS the reported t, and the exit face is never nearer than the entry
    if (!rotated && !(tHit < t_limit)) return false;
    if (kAnyHit && (m.flags & MESH_OPAQUE)) {  // texColor.a != 0 whatever the face and texel (:311): an ordinary hit at tHit
        if (!rotated) return true;
        const V3 pw = to_world(m, local.o + local.d * tHit);
        return dot(pw - world.o, world.d) < t_limit;  // :402
    }

    V3 hp = local.o + local.d * tHit;
    float t_front = tHit;
    V3 p_front = hp;
    if (rotated) {
        p_front = to_world(m, hp);
        t_front = dot(p_front - world.o, world.d);  // :402
    }
    float u, v;
    face_uv(hp, lo, hi, axis, neg, u, v);
    int texel = face_texel_index(sc, mesh_index, face_slot(axis, neg), u, v);
    if (!(alpha_bits(sc, texel) & 1u)) {  // texColor.a != 0 → ordinary hit
        if (!(t_front < t_limit)) return false;
        c.t = t_front;
        c.p = p_front;
        c.texel = texel;
        c.axis = axis;
        c.neg = neg;
        c.back = false;
        return true;
    }
    // transparent texel (:311-361): inner layer → miss; outer layer → try the exit face
    if (!(m.flags & MESH_OUTER)) return false;
    if (!(s.tmax > tHit)) return false;
    V3 bp = local.o + local.d * s.tmax;
    float bu, bv;
    face_uv(bp, lo, hi, s.out_axis, s.out_neg, bu, bv);
    int btexel = face_texel_index(sc, mesh_index, face_slot(s.out_axis, s.out_neg), bu, bv);
    if (!(alpha_bits(sc, btexel) & 2u)) return false;  // backTexColor.a > 0
    float t_back = s.tmax;
    V3 p_back = bp;
    if (rotated) {
        p_back = to_world(m, bp);
        t_back = dot(p_back - world.o, world.d);
    }
    if (!(t_back < t_limit)) return false;
    c.t = t_back;
    c.p = p_back;
    c.texel = btexel;
    c.axis = s.out_axis;
    c.neg = s.out_neg;
    c.back = true;
    return true;
}

// Phase 1 over the scene: bit i of the result = mesh i (< 64) may be hit.  Only group roots are
// tested (flat_scene.h: a root's box contains its members' boxes); a passing root takes its members
// along — phase 2 is exact, phase 1 only has to be conservative.  The loop index is wave-uniform
// (scalar loads); meshes beyond 63 are handled by the callers' tail loops.
// kLeaving (reflection rays: nine in ten leave the figure): an un-posed root is first asked whether the ray moves away
// from it — its origin beyond a face and its direction not towards it on that axis, which makes both slab distances of
// the axis negative or the axis "parallel" with the origin outside (intersection.cpp:222-249): the slab test's own
// verdict from two comparisons per axis — and the slab test is skipped when that holds for every lane of the wave.
template <bool kLeaving = false, class SV>
DEV unsigned long long scene_candidates(const SV& sc, const RayQ& q, unsigned long long mesh_mask, float t_limit) {
    unsigned long long cand = 0ull;
    const int n = sc.n_meshes < 64 ? sc.n_meshes : 64;
    const unsigned long long roots = sc.roots;
#pragma unroll 1
    for (int i = 0; i < n; ++i) {
        if (!((roots >> i) & 1ull)) continue;
        const MeshData m = mesh_uniform(sc, i);
        if (!(m.group & mesh_mask)) continue;  // uniform: the mask is per tile
        if (kLeaving && !(SV::kPosed && (m.flags & MESH_ROTATED)) && !(m.flags & MESH_EMPTY)) {
            // beyond the max face: every d > -1e-8 (away, or "parallel": |d| < 1e-8, :222); beyond the min face: every d < 1e-8
            const bool away = ((q.o.x > m.hi.x) & (q.d.x > -1e-8f)) | ((q.o.x < m.lo.x) & (q.d.x < 1e-8f)) | ((q.o.y > m.hi.y) & (q.d.y > -1e-8f)) |
                              ((q.o.y < m.lo.y) & (q.d.y < 1e-8f)) | ((q.o.z > m.hi.z) & (q.d.z > -1e-8f)) | ((q.o.z < m.lo.z) & (q.d.z < 1e-8f));
            if (!__ballot(!away)) continue;
        }
        if (mesh_may_hit<SV::kPosed>(m, q, t_limit)) cand |= m.group;
    }
    return cand & mesh_mask;
}

// intersectScene :408-421.  mesh_mask: bit i set → mesh i is tested (primary-ray culling; all
// ones for secondary rays).  Meshes beyond bit 63 are always tested.
template <bool kLeaving = false, class SV>
DEV Hit hit_scene(const SV& sc, const Ray& r, uint64_t mesh_mask) {
    const RayQ q = prepare(r);
    Cand best;
    best.t = kFltMax;
    best.p = mk(0, 0, 0);
    best.texel = MCRT_TEX_EMPTY;
    best.axis = 0;
    best.neg = false;
    best.back = false;
    int best_mesh = -1;
    // per lane: ascending mesh index, strictly smaller t wins → first mesh on ties, like the reference
    unsigned long long cand = scene_candidates<kLeaving>(sc, q, mesh_mask, kFltMax);
    while (cand) {
        const int i = __builtin_ctzll(cand);
        cand &= cand - 1ull;
        Cand c;
        if (mesh_candidate<false>(sc, mesh_lane(sc, i), i, q, best.t, c)) {
            best = c;
            best_mesh = i;
        }
    }
    if constexpr (!SV::kLds) {  // LDS views hold at most 64 meshes
        for (int i = 64; i < sc.n_meshes; ++i) {
            Cand c;
            if (mesh_candidate<false>(sc, mesh_uniform(sc, i), i, q, best.t, c)) {
                best = c;
                best_mesh = i;
            }
        }
    }
    Hit h;
    h.hit = best_mesh >= 0;
    h.outer = false;
    h.t = h.hit ? best.t : kFltMax;
    h.p = best.p;
    h.n = mk(0, 0, 0);
    h.tex = C4{0.0f, 0.0f, 0.0f, 1.0f};
    h.texel = best.texel;
    h.axis = best.axis;
    h.neg = best.neg;
    h.back = best.back;
    if (h.hit) {
        const MeshData m = mesh_lane(sc, best_mesh);
        V3 n = face_normal(best.axis, best.neg);
        if (best.back) n = n * -1.0f;  // :354
        if (SV::kPosed && (m.flags & MESH_ROTATED)) {  // :400
            n = normalize(spin(n, mk(0.0f, 0.0f, 0.0f), (m.flags & MESH_APPLY_X) != 0, m.fwd_x_cos, m.fwd_x_sin,
                               (m.flags & MESH_APPLY_Z) != 0, m.fwd_z_cos, m.fwd_z_sin));
        }
        h.n = n;
        h.tex = texel_color(sc, best.texel);
        h.outer = best.back || (m.flags & MESH_OUTER) != 0;
    }
    return h;
}

// "hit && t < limit" over the scene without keeping the hit: isInShadow :25 and computeAO :72.
// The reference finds the closest hit first; min t < limit ⇔ some t < limit, so a lane stops at
// its first qualifying mesh.
template <class SV>
DEV bool any_hit_inline(const SV& sc, const Ray& r, float limit) {
    const RayQ q = prepare(r);
    unsigned long long cand = scene_candidates(sc, q, ~0ull, limit);
    while (cand) {
        const int i = __builtin_ctzll(cand);
        cand &= cand - 1ull;
        Cand c;
        if (mesh_candidate<true>(sc, mesh_lane(sc, i), i, q, limit, c)) return true;
    }
    if constexpr (!SV::kLds) {
        for (int i = 64; i < sc.n_meshes; ++i) {
            Cand c;
            if (mesh_candidate<true>(sc, mesh_uniform(sc, i), i, q, limit, c)) return true;
        }
    }
    return false;
}
// mesh_candidate<true> for an UN-POSED mesh whose box holds the ray's origin STRICTLY inside (the caller has checked
// lo < o < hi per axis): what intersectAABB does then, and nothing else.  Every non-parallel axis has t_near < 0 < t_far
// (:229-236), so tmin < 0 and the reference takes the exit face — tmax with the axis of the smallest t_far, first axis on
// ties (:255-288); the slab loop's reject (:246) cannot fire (tmin < 0 <= tmax); a transparent exit texel ends the test for
// inner and outer layer alike (:312, :318: tmax > tHit fails when tHit IS tmax).  Most shadow and ambient-occlusion rays
// that have candidates at all start under a transparent texel of their part's outer layer — inside that box — and this is
// less than half the instructions of the general routine (no entry-face bookkeeping, no overlap logic, no second face).
template <class SV>
DEV bool mesh_candidate_inside(const SV& sc, const MeshData& m, int mesh_index, const RayQ& r, float t_limit) {
    float t_exit = kFltMax;
    int axis = 0;
    bool neg = false;
    auto far_side = [&](int i, bool par, float o, float inv, float l, float h) __attribute__((always_inline)) {
        if (par) return;  // |d| < 1e-8: the origin is inside the slab, the axis takes no part (:222-227)
        const float t0 = (l - o) * inv, t1 = (h - o) * inv;
        const bool swapped = t0 > t1;
        const float tf = swapped ? t0 : t1;
        if (tf < t_exit) {  // == `tmax = min(tmax, t1)` and the exit-face scan's strict `<` (:241, :268-285)
            t_exit = tf;
            axis = i;
            neg = swapped;
        }
    };
    far_side(0, r.px, r.o.x, r.inv.x, m.lo.x, m.hi.x);
    far_side(1, r.py, r.o.y, r.inv.y, m.lo.y, m.hi.y);
    far_side(2, r.pz, r.o.z, r.inv.z, m.lo.z, m.hi.z);
    if (!(t_exit < t_limit)) return false;  // local t is the reported t (un-posed)
    if (m.flags & MESH_OPAQUE) return true;
    const V3 hp = r.o + r.d * t_exit;
    float u, v;
    face_uv(hp, m.lo, m.hi, axis, neg, u, v);
    return !(alpha_bits(sc, face_texel_index(sc, mesh_index, face_slot(axis, neg), u, v)) & 1u);  // texColor.a != 0 (:311)
}

// The same with the first pass already done for this ray (a per-hit conservative candidate mask):
// only the exact second pass runs.  Scenes beyond 64 meshes still scan their tail exactly.
// inside: the candidates whose (un-posed) box holds the ray's origin strictly inside (origin_inside_boxes)
template <class SV>
DEV bool any_hit_masked(const SV& sc, const Ray& r, float limit, unsigned long long cand, unsigned long long inside = 0ull) {
    const RayQ q = prepare(r);
    while (cand) {
        const int i = __builtin_ctzll(cand);
        cand &= cand - 1ull;
        if ((inside >> i) & 1ull) {
            if (mesh_candidate_inside(sc, mesh_lane(sc, i), i, q, limit)) return true;
            continue;
        }
        Cand c;
        if (mesh_candidate<true>(sc, mesh_lane(sc, i), i, q, limit, c)) return true;
    }
    if constexpr (!SV::kLds) {
        for (int i = 64; i < sc.n_meshes; ++i) {
            Cand c;
            if (mesh_candidate<true>(sc, mesh_uniform(sc, i), i, q, limit, c)) return true;
        }
    }
    return false;
}

// Conservative first pass for ALL shadow rays of one hit towards a disk light (computeSoftShadow,
// shading.cpp:28-60): origin O, light centre L, sample radius R.  Ray i runs from O to T_i with
// |T_i - L| <= R, so its point at fraction s of the way lies within s·R of the central segment's
// point O + s·(L - O).  If ray i meets a box B (fraction s <= far(B) / |T_i - O|, far = distance from
// O to B's farthest corner), the central segment's point at the SAME fraction s lies in B inflated
// by s·R.  The test below solves "exists s in [0,1] with O + s·(L-O) in B inflated by s·R" exactly
// (linear in s per axis) with generous float slack; it only ever adds candidates — the exact
// per-ray test (any_hit_masked) decides.  Posed meshes: the same test with the segment rotated into
// the mesh's local frame.
template <bool kPosed, class SV>
DEV unsigned long long bundle_candidates(const SV& sc, V3 O, V3 L, float R) {
    const V3 D = L - O;
    // every margin of the test is a multiple of the scene's slack (flat_scene.h: kMaskSlack x the scene's coordinate
    // magnitude, 2e-3 at the reference's scale), so the test keeps its margin in ulps however the scene is scaled or placed
    const float slack = sc.hdr->mask_slack;
    const float Rb = R * 1.001f + 1e-3f * slack;
    const float flat = 1e-3f * slack + 1e-37f, tol = 5e-3f * slack;
    unsigned long long cand = 0ull;
    const int n = sc.n_meshes < 64 ? sc.n_meshes : 64;
    const unsigned long long roots = sc.roots;
#pragma unroll 1
    for (int i = 0; i < n; ++i) {
        if (!((roots >> i) & 1ull)) continue;
        const MeshData m = mesh_uniform(sc, i);
        if (m.flags & MESH_EMPTY) continue;
        // posed mesh: the same test in the mesh's local frame — the rotation is rigid, so the bundle keeps
        // its radius; the float error of rotating two points (a few ulp of the coordinates) is far inside the slack
        V3 o = O, d = D;
        if (kPosed && (m.flags & MESH_ROTATED)) {
            const bool ax = (m.flags & MESH_APPLY_X) != 0, az = (m.flags & MESH_APPLY_Z) != 0;
            auto to_mesh = [&](V3 pnt) __attribute__((always_inline)) {
                V3 q = spin(pnt, m.pivot, false, 1.0f, 0.0f, az, m.inv_z_cos, m.inv_z_sin);
                return spin(q, m.pivot, ax, m.inv_x_cos, m.inv_x_sin, false, 1.0f, 0.0f);
            };
            o = to_mesh(O);
            d = to_mesh(L) - o;
        }
        bool pass;
        {
            // C(s) = o + s·d inside the box inflated by s·Rb (+ slack), for some s in [0, 1]: per axis
            //   (d + Rb)·s >= lo - slack - o      and      (d - Rb)·s <= hi + slack - o
            // — linear in s, so the feasible s form an interval
            float s_in = -1e-4f, s_out = 1.0f + 1e-4f;
            bool ok = true;
            auto axis = [&](float o1, float d1, float l, float h) __attribute__((always_inline)) {
                const float al = d1 + Rb, bl = l - slack - o1;  // al·s >= bl
                const float ah = d1 - Rb, bh = h + slack - o1;  // ah·s <= bh
                const float ql = bl * __builtin_amdgcn_rcpf(al), qh = bh * __builtin_amdgcn_rcpf(ah);
                const bool flat_l = __builtin_fabsf(al) < flat, flat_h = __builtin_fabsf(ah) < flat;
                // flat: the inequality does not depend on s (holds iff 0 >= bl resp. 0 <= bh; tiny margin)
                ok = ok & !(flat_l & (bl > tol)) & !(flat_h & (bh < -tol));
                s_in = (!flat_l & (al > 0.0f)) ? smax(s_in, ql - 1e-5f) : s_in;
                s_out = (!flat_l & (al < 0.0f)) ? smin(s_out, ql + 1e-5f) : s_out;
                s_out = (!flat_h & (ah > 0.0f)) ? smin(s_out, qh + 1e-5f) : s_out;
                s_in = (!flat_h & (ah < 0.0f)) ? smax(s_in, qh - 1e-5f) : s_in;
            };
            axis(o.x, d.x, m.lo.x, m.hi.x);
            axis(o.y, d.y, m.lo.y, m.hi.y);
            axis(o.z, d.z, m.lo.z, m.hi.z);
            pass = ok & !(s_in > s_out);
        }
        if (pass) cand |= m.group;
    }
    return cand;
}

// ---------------------------------------------------------------------------------------------
// Whole-bundle decisions.  Most hits of a frame are lit by ALL of their S light samples or by NONE: the S shadow
// rays leave one origin towards a small disk far away, so against the boxes around the origin they all do the same
// thing — move away from the face they start on, or leave the enclosing outer-layer box through one and the same
// texel.  bundle_decide_mesh proves such an outcome for one mesh and EVERY ray the hit can cast (every target T
// with |T - L| <= R), from interval bounds with margins far above the float error of the per-ray arithmetic
// (intersection.cpp:200-371, shading.cpp:14-26), or says BUNDLE_UNKNOWN.  A hit whose candidates are all decided
// needs no light samples at all: no mt19937 stream, no sin/cos, no shadow rays — its lit count is 0 or S.
// What a decision rests on:
//   moving away   origin beyond a face on one axis and no target on the box's side of it: both slab distances
//                 of that axis are negative (or the axis is parallel and outside) — an exact sign argument;
//   one face      the origin lies inside the box (exit face) or beyond exactly one face with every ray heading
//                 inwards (entry face): the points where the rays cross that face's plane lie in a rectangle
//                 [X_lo, X_hi] (the perspective image of the target box); if it is inside the face with margin, every
//                 ray reports that face, and if the corner texels of the rectangle agree (<= 2 x 2 texels; the float
//                 u,v → texel pipeline is monotone per coordinate, so every ray's texel lies between the corners')
//                 on `alpha == 0`, the face decides: opaque → hit at t < distance (checked), transparent →
//                 miss (an outer-layer ENTRY face would go on to the exit face: left undecided);
//   passing by    beyond one face and the rectangle is outside the face with margin: the ray has left another
//                 slab before it enters this one → miss.
// A posed mesh is examined in its own frame, like the reference examines it (intersection.cpp:384-395), with the
// rotation's float error added to the bounds.  Origins beyond two faces, rectangles that straddle texels or face
// edges, boxes without thickness on the crossed axis: BUNDLE_UNKNOWN — those meshes stay in the hit's mask and
// its rays are traced.
// ---------------------------------------------------------------------------------------------
enum : int { BUNDLE_UNKNOWN = 0, BUNDLE_MISS = 1, BUNDLE_HIT = 2 };
struct BundleGeom {
    V3 O;                    // point + normal * 1e-3f, the origin isInShadow forms (shading.cpp:15)
    V3 nlo, nhi;             // bounds of T - O per axis over every target
    float dist_lo, dist_hi;  // bounds of |T - O|
    float omax;              // max |O| component
    bool ok;
};
DEV float max3abs(V3 v) { return smax(smax(__builtin_fabsf(v.x), __builtin_fabsf(v.y)), __builtin_fabsf(v.z)); }
// rot_slop: 0, or (posed meshes, O and L in the mesh's frame) the relative error of the reference's rotated ray
// direction (rotateDir + normalize, intersection.cpp:388-393) against the rotated difference formed here
DEV BundleGeom bundle_geom(V3 O, V3 L, float R, float rot_slop) {
    BundleGeom g;
    g.O = O;
    g.omax = max3abs(O);
    const V3 D = L - O;
    // |T - L| <= R (1 + 2e-6) componentwise and in length (unit frame vectors, |sin|,|cos| <= 1); slop: the roundings
    // of L + offset, of the differences below and of the reference's own T - O
    const float Rb = R * 1.001f + 2e-6f * (max3abs(L) + g.omax + R) + rot_slop * (max3abs(D) + R) + 1e-30f;
    g.nlo = mk(D.x - Rb, D.y - Rb, D.z - Rb);
    g.nhi = mk(D.x + Rb, D.y + Rb, D.z + Rb);
    const float dc = __builtin_amdgcn_sqrtf(dot(D, D));  // (an estimate will do: the bounds carry 1e-5)
    g.dist_lo = dc * (1.0f - 1e-5f) - Rb;
    g.dist_hi = dc * (1.0f + 1e-5f) + Rb;
    g.ok = g.dist_lo > 1e-3f && g.dist_hi < 1e18f;  // isInShadow's `distToLight < 1e-6` exit is never taken
    return g;
}
template <class SV>
DEV int bundle_decide_mesh(const SV& sc, const MeshData& m, int mesh_index, const BundleGeom& gw, V3 Lw, float R) {
    if (m.flags & MESH_EMPTY) return BUNDLE_MISS;  // intersection.cpp:205
    BundleGeom g = gw;
    if (SV::kPosed && (m.flags & MESH_ROTATED)) {
        // a posed mesh is tested in its own frame (intersection.cpp:384-395).  The origin is rotated exactly as
        // to_local does (the same bits as the reference's localOrigin); the rotation is rigid, so the targets stay
        // within R of the rotated light centre and every distance keeps its value; the reference's local ray
        // direction differs from the rotated difference by a few 1e-7 of its length.
        const bool rx = (m.flags & MESH_APPLY_X) != 0, rz = (m.flags & MESH_APPLY_Z) != 0;
        auto to_mesh = [&](V3 pnt) __attribute__((always_inline)) {
            const V3 q = spin(pnt, m.pivot, false, 1.0f, 0.0f, rz, m.inv_z_cos, m.inv_z_sin);
            return spin(q, m.pivot, rx, m.inv_x_cos, m.inv_x_sin, false, 1.0f, 0.0f);
        };
        g = bundle_geom(to_mesh(gw.O), to_mesh(Lw), R, 8e-6f);
        if (!g.ok) return BUNDLE_UNKNOWN;
    }
    const V3 lo = m.lo, hi = m.hi, O = g.O;
    const bool ax = O.x > hi.x, bx = O.x < lo.x, ay = O.y > hi.y, by = O.y < lo.y, az = O.z > hi.z, bz = O.z < lo.z;
    // moving away (exact): beyond the max face with T - O >= 0 for every target, or beyond the min face with <= 0
    if ((ax & (g.nlo.x > 0.0f)) | (bx & (g.nhi.x < 0.0f)) | (ay & (g.nlo.y > 0.0f)) | (by & (g.nhi.y < 0.0f)) | (az & (g.nlo.z > 0.0f)) |
        (bz & (g.nhi.z < 0.0f)))
        return BUNDLE_MISS;
    // the numbers involved: the origin, the box, and never less than a fiftieth of the scene's coordinate magnitude (500
    // slacks: 1.0 at the reference's scale) — a floor that moves with the scene instead of an absolute one
    const float scale = smax(smax(500.0f * sc.hdr->mask_slack, g.omax), smax(max3abs(lo), max3abs(hi)));
    const float mg = 2e-5f * scale;  // >= 8 x the float error of a crossing point (a few ulp of box-sized numbers)
    const int n_out = static_cast<int>(ax | bx) + static_cast<int>(ay | by) + static_cast<int>(az | bz);
    if (n_out > 1) return BUNDLE_UNKNOWN;
    // strictly inside the slabs of the other axes
    const bool ix = (O.x - lo.x > mg) & (hi.x - O.x > mg), iy = (O.y - lo.y > mg) & (hi.y - O.y > mg), iz = (O.z - lo.z > mg) & (hi.z - O.z > mg);
    int a;
    bool inside;
    if (n_out == 1) {
        a = (ax | bx) ? 0 : ((ay | by) ? 1 : 2);
        if (!((a == 0 || ix) && (a == 1 || iy) && (a == 2 || iz))) return BUNDLE_UNKNOWN;
        // a box without thickness on this axis: both slab distances are equal and the reference's `t0 > t1` swap
        // (:233), hence the face it reports, no longer follows the ray's direction
        if (!(comp(hi, a) - comp(lo, a) > mg)) return BUNDLE_UNKNOWN;
        inside = false;
    } else {
        if (!(ix & iy & iz)) return BUNDLE_UNKNOWN;
        inside = true;
        // exit axis of the central ray: the smallest fraction s = (face - O) / (L - O) over the axes whose sign is
        // the same for every target (the rectangle test below proves it for the whole bundle, or fails)
        float best = kFltMax;
        a = -1;
        auto axis = [&](int c, float o, float l, float h, float nl, float nh) __attribute__((always_inline)) {
            if (!((nl > 0.0f) | (nh < 0.0f))) return;
            const float d = 0.5f * (nl + nh);
            const float s = ((d > 0.0f ? h : l) - o) * __builtin_amdgcn_rcpf(d);
            if (s < best) {
                best = s;
                a = c;
            }
        };
        axis(0, O.x, lo.x, hi.x, g.nlo.x, g.nhi.x);
        axis(1, O.y, lo.y, hi.y, g.nlo.y, g.nhi.y);
        axis(2, O.z, lo.z, hi.z, g.nlo.z, g.nhi.z);
        if (a < 0) return BUNDLE_UNKNOWN;
    }
    const int b1 = a == 0 ? 1 : 0, b2 = a == 2 ? 1 : 2;
    const float nl_a = comp(g.nlo, a), nh_a = comp(g.nhi, a);
    const bool towards_max = nl_a > 0.0f;  // every ray's direction component on axis a is positive
    if (!(towards_max | (nh_a < 0.0f))) return BUNDLE_UNKNOWN;
    // outside: heading inwards on axis a (otherwise the moving-away rule has answered already)
    if (!inside && towards_max != (a == 0 ? bx : (a == 1 ? by : bz))) return BUNDLE_UNKNOWN;
    // the face the rays cross: inside → they leave through it; outside → they enter through it
    const bool face_max = inside ? towards_max : !towards_max;
    const float f_a = face_max ? comp(hi, a) : comp(lo, a);
    const float h = f_a - comp(O, a);                     // signed like T - O on this axis
    const float d_near = towards_max ? nl_a : nh_a;       // closest to 0
    const float d_far = towards_max ? nh_a : nl_a;
    if (!(__builtin_fabsf(d_near) > 1e-6f * g.dist_hi)) return BUNDLE_UNKNOWN;  // |dir_a| >= 1e-8: the axis is never "parallel" (:222)
    // fraction of the way to the target at which a ray crosses the plane: s = h / (T_a - O_a) > 0
    const float s_lo = h * __builtin_amdgcn_rcpf(d_far) * (1.0f - 1e-5f), s_hi = h * __builtin_amdgcn_rcpf(d_near) * (1.0f + 1e-5f);
    if (!(s_lo >= 0.0f)) return BUNDLE_UNKNOWN;
    const float t_hi = s_hi * g.dist_hi;  // the distance along a ray is at most this
    const float mg2 = mg + 8e-6f * t_hi;  // the slab distances being compared carry a relative error, and so do the directions
    if (!(t_hi + mg2 < g.dist_lo * 0.999f)) return BUNDLE_UNKNOWN;  // `hit.t < distToLight` (shading.cpp:25) holds for every ray
    auto cross_lo = [&](float o, float nl) __attribute__((always_inline)) { return o + (nl >= 0.0f ? s_lo * nl : s_hi * nl) - mg2; };
    auto cross_hi = [&](float o, float nh) __attribute__((always_inline)) { return o + (nh >= 0.0f ? s_hi * nh : s_lo * nh) + mg2; };
    const float x1lo = cross_lo(comp(O, b1), comp(g.nlo, b1)), x1hi = cross_hi(comp(O, b1), comp(g.nhi, b1));
    const float x2lo = cross_lo(comp(O, b2), comp(g.nlo, b2)), x2hi = cross_hi(comp(O, b2), comp(g.nhi, b2));
    const float l1 = comp(lo, b1), h1 = comp(hi, b1), l2 = comp(lo, b2), h2 = comp(hi, b2);
    const bool through = (x1lo > l1 + mg2) & (x1hi < h1 - mg2) & (x2lo > l2 + mg2) & (x2hi < h2 - mg2);
    if (!through) {
        // beyond one face and the crossing points lie beside the face: by then the ray has left another slab → miss
        if (!inside && ((x1hi < l1 - mg2) | (x1lo > h1 + mg2) | (x2hi < l2 - mg2) | (x2lo > h2 + mg2))) return BUNDLE_MISS;
        return BUNDLE_UNKNOWN;
    }
    // every ray reports face (a, neg) first, at a distance below its light distance
    if (m.flags & MESH_OPAQUE) return BUNDLE_HIT;
    const bool neg = inside ? !towards_max : towards_max;  // exit: the min side when heading down; entry: the min side when heading up
    const int face = face_slot(a, neg);
    int off, w, hgt;
    if constexpr (SV::kLds) {
        const MCRT_LDS int* f = sc.faces + (mesh_index * 6 + face) * 4;
        off = f[0], w = f[1], hgt = f[2];
    } else {
        const FlatMesh& fm = sc.meshes[mesh_index];
        off = fm.tex_off[face], w = fm.tex_w[face], hgt = fm.tex_h[face];
    }
    if (off < 0) return BUNDLE_HIT;  // null / empty texture: alpha 1
    // The texels under the rectangle.  The reference maps a crossing point to a texel by float u,v (computeFaceUV
    // :136-196, TextureRegion::sample): per coordinate a monotone function of the point.  Here the same map in plain
    // arithmetic with a margin of 1e-3 texel (+ the maps' own float error, which grows with the texel count): the
    // reference's texel of every ray lies between the two ends.
    //   axis z: u ← x (mirrored on the min side), v ← y mirrored;  axis x: u ← z (mirrored on the max side),
    //   v ← y mirrored;  axis y: u ← x, v ← z (mirrored on the min side)
    const bool u_from_b2 = a == 0;
    const float ul = u_from_b2 ? x2lo : x1lo, uh = u_from_b2 ? x2hi : x1hi, ulo = u_from_b2 ? l2 : l1, uhi = u_from_b2 ? h2 : h1;
    const float vl = u_from_b2 ? x1lo : x2lo, vh = u_from_b2 ? x1hi : x2hi, vlo = u_from_b2 ? l1 : l2, vhi = u_from_b2 ? h1 : h2;
    const bool u_flip = a == 2 ? neg : (a == 0 ? !neg : false), v_flip = a == 1 ? neg : true;
    const float iu = __builtin_amdgcn_rcpf(uhi - ulo), iv = __builtin_amdgcn_rcpf(vhi - vlo);  // extents > 2 mg (strictly inside, above)
    float u0 = (ul - ulo) * iu, u1 = (uh - ulo) * iu, v0 = (vl - vlo) * iv, v1 = (vh - vlo) * iv;
    if (u_flip) {
        const float t = 1.0f - u1;
        u1 = 1.0f - u0;
        u0 = t;
    }
    if (v_flip) {
        const float t = 1.0f - v1;
        v1 = 1.0f - v0;
        v0 = t;
    }
    const float fw = static_cast<float>(w), fh = static_cast<float>(hgt);
    const float du = 1e-3f + 4e-6f * fw, dv = 1e-3f + 4e-6f * fh;
    const int txa = iclamp(static_cast<int>(__builtin_floorf(sclamp(u0, 0.0f, 1.0f) * fw - du)), 0, w - 1);
    const int txb = iclamp(static_cast<int>(__builtin_floorf(sclamp(u1, 0.0f, 1.0f) * fw + du)), 0, w - 1);
    const int tya = iclamp(static_cast<int>(__builtin_floorf(sclamp(v0, 0.0f, 1.0f) * fh - dv)), 0, hgt - 1);
    const int tyb = iclamp(static_cast<int>(__builtin_floorf(sclamp(v1, 0.0f, 1.0f) * fh + dv)), 0, hgt - 1);
    if (!(txb - txa <= 1 && tyb - tya <= 1 && txb >= txa && tyb >= tya)) return BUNDLE_UNKNOWN;  // also NaN bounds
    const uint32_t b00 = alpha_bits(sc, off + tya * w + txa), b01 = alpha_bits(sc, off + tya * w + txb);
    const uint32_t b10 = alpha_bits(sc, off + tyb * w + txa), b11 = alpha_bits(sc, off + tyb * w + txb);
    const uint32_t any_clear = (b00 | b01 | b10 | b11) & 1u, all_clear = b00 & b01 & b10 & b11 & 1u;
    if (!any_clear) return BUNDLE_HIT;  // texColor.a != 0 for every ray (:311)
    if (!all_clear) return BUNDLE_UNKNOWN;
    // every ray meets a transparent texel: an exit face ends the test (tmax > tHit fails, :318), an inner-layer
    // entry face too (:312); an outer-layer entry face goes on to the exit face
    if (inside || !(m.flags & MESH_OUTER)) return BUNDLE_MISS;
    return BUNDLE_UNKNOWN;
}
// Candidates and decisions of one hit: a wave-uniform loop over the group roots (scalar mesh data) with the exact
// moving-away rule and bundle_candidates' conservative segment test on the root's box — both hold for the members
// inside it — then bundle_decide_mesh for every member of the groups that are left.
// Returns the hit's lit count when every candidate is decided — 0 (some mesh stops every ray) or S (no mesh stops
// any) — else -1; `cand` = the meshes whose rays have to be traced.  decide = false: candidates only.
// bundle_candidates' segment test with the ray-dependent half prepared once per hit: along one axis the box inflated
// by s·Rb holds the segment point o + s·d iff (d + Rb)·s >= lo - slack - o and (d - Rb)·s <= hi + slack - o, i.e. each
// side bounds s from below or from above depending on the sign of its coefficient.  Every half-constraint becomes a
// pair (c, k) with bound = fma(b, c, k): c = 1/coefficient and k = -+1e-5 where it applies, c = 0 and k = -+1e30
// where it does not.  Plain floats, no per-lane booleans: the loop over the roots keeps them in VGPRs (as booleans
// they were lane masks in SGPR pairs, hoisted out of the loop and spilled: two v_readlane per use).
struct SegAxis {
    float c_in_lo, k_in_lo, c_out_lo, k_out_lo;  // from the min face
    float c_in_hi, k_in_hi, c_out_hi, k_out_hi;  // from the max face
};
DEV SegAxis seg_axis(float d1, float Rb, float nudge) {
    float al = d1 + Rb, ah = d1 - Rb;
    // a vanishing coefficient (a constraint that does not depend on s) is nudged to a thousandth of the test's slack
    al = __builtin_fabsf(al) < nudge ? nudge : al;
    ah = __builtin_fabsf(ah) < nudge ? nudge : ah;
    const float ial = __builtin_amdgcn_rcpf(al), iah = __builtin_amdgcn_rcpf(ah);
    const float big = 1e30f, eps = 1e-5f;
    SegAxis c;
    c.c_in_lo = al > 0.0f ? ial : 0.0f, c.k_in_lo = al > 0.0f ? -eps : -big;  // al·s >= bl, al > 0: s >= bl / al
    c.c_out_lo = al > 0.0f ? 0.0f : ial, c.k_out_lo = al > 0.0f ? big : eps;  //              al < 0: s <= bl / al
    c.c_out_hi = ah > 0.0f ? iah : 0.0f, c.k_out_hi = ah > 0.0f ? eps : big;  // ah·s <= bh, ah > 0: s <= bh / ah
    c.c_in_hi = ah > 0.0f ? 0.0f : iah, c.k_in_hi = ah > 0.0f ? -big : -eps;  //              ah < 0: s >= bh / ah
    return c;
}
DEV void seg_axis_apply(const SegAxis& c, float o1, float l, float h, float slack, float& s_in, float& s_out) {
    const float bl = (l - o1) - slack, bh = (h - o1) + slack;
    s_in = __builtin_fmaxf(s_in, __builtin_fmaxf(__builtin_fmaf(bl, c.c_in_lo, c.k_in_lo), __builtin_fmaf(bh, c.c_in_hi, c.k_in_hi)));
    s_out = __builtin_fminf(s_out, __builtin_fminf(__builtin_fmaf(bl, c.c_out_lo, c.k_out_lo), __builtin_fmaf(bh, c.c_out_hi, c.k_out_hi)));
}

template <bool kPosed, class SV>
DEV int bundle_classify(const SceneView& scg, const SV& sc, V3 O, V3 L, float R, int S, bool decide, unsigned long long& cand) {
    const BundleGeom g = bundle_geom(O, L, R, 0.0f);
    decide = decide && g.ok && scg.n_meshes <= 64;  // meshes beyond the mask are tested per ray
    const V3 D = L - O;
    const float slack = scg.hdr->mask_slack;  // kMaskSlack x the scene's coordinate magnitude (flat_scene.h): scale-free margins
    const float Rb = R * 1.001f + 1e-3f * slack, nudge = 1e-3f * slack + 1e-37f;
    const SegAxis cx = seg_axis(D.x, Rb, nudge), cy = seg_axis(D.y, Rb, nudge), cz = seg_axis(D.z, Rb, nudge);
    // moving away from a box (exact, see bundle_decide_mesh), as floats: up = -1 where every target lies above the
    // origin on that axis, dn = 1 where below; the rule holds iff up·(hi - O) > 0 or dn·(lo - O) > 0 (float
    // subtraction keeps the sign of the comparison)
    const float upx = (decide & (g.nlo.x > 0.0f)) ? -1.0f : 0.0f, dnx = (decide & (g.nhi.x < 0.0f)) ? 1.0f : 0.0f;
    const float upy = (decide & (g.nlo.y > 0.0f)) ? -1.0f : 0.0f, dny = (decide & (g.nhi.y < 0.0f)) ? 1.0f : 0.0f;
    const float upz = (decide & (g.nlo.z > 0.0f)) ? -1.0f : 0.0f, dnz = (decide & (g.nhi.z < 0.0f)) ? 1.0f : 0.0f;
    unsigned long long keep = 0ull;
    bool dark = false;
    const int n = scg.n_meshes < 64 ? scg.n_meshes : 64;
    const unsigned long long roots = scg.roots;
#pragma unroll 1
    for (int i = 0; i < n; ++i) {
        if (!((roots >> i) & 1ull)) continue;
        const MeshData m = mesh_uniform(scg, i);
        if (m.flags & MESH_EMPTY) continue;
        const bool rotated = kPosed && (m.flags & MESH_ROTATED) != 0;
        float s_in = -1e-4f, s_out = 1.0f + 1e-4f, away = 0.0f;
        if (rotated) {  // the segment in the mesh's frame
            const bool ax = (m.flags & MESH_APPLY_X) != 0, az = (m.flags & MESH_APPLY_Z) != 0;
            auto to_mesh = [&](V3 pnt) __attribute__((always_inline)) {
                V3 q = spin(pnt, m.pivot, false, 1.0f, 0.0f, az, m.inv_z_cos, m.inv_z_sin);
                return spin(q, m.pivot, ax, m.inv_x_cos, m.inv_x_sin, false, 1.0f, 0.0f);
            };
            const V3 o = to_mesh(O);
            const V3 d = to_mesh(L) - o;
            seg_axis_apply(seg_axis(d.x, Rb, nudge), o.x, m.lo.x, m.hi.x, slack, s_in, s_out);
            seg_axis_apply(seg_axis(d.y, Rb, nudge), o.y, m.lo.y, m.hi.y, slack, s_in, s_out);
            seg_axis_apply(seg_axis(d.z, Rb, nudge), o.z, m.lo.z, m.hi.z, slack, s_in, s_out);
        } else {  // its members lie inside the root's box: moving away from it is moving away from them
            away = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(upx * (m.hi.x - O.x), dnx * (m.lo.x - O.x)),
                                                   __builtin_fmaxf(upy * (m.hi.y - O.y), dny * (m.lo.y - O.y))),
                                   __builtin_fmaxf(upz * (m.hi.z - O.z), dnz * (m.lo.z - O.z)));
            seg_axis_apply(cx, O.x, m.lo.x, m.hi.x, slack, s_in, s_out);
            seg_axis_apply(cy, O.y, m.lo.y, m.hi.y, slack, s_in, s_out);
            seg_axis_apply(cz, O.z, m.lo.z, m.hi.z, slack, s_in, s_out);
        }
        const bool pass = !(away > 0.0f) & !(s_in > s_out);
        if (pass) keep |= m.group;
    }
    // the decisions, per lane over its own candidates (mesh data from the LDS table: as a wave-uniform loop over the
    // members with scalar mesh data this part ran fewer instructions and 2 % slower — every lane then waits for the
    // scalar loads of every mesh some lane needs)
    if (decide) {
        unsigned long long rest = keep;
        keep = 0ull;
        while (rest) {
            const int j = __builtin_ctzll(rest);
            rest &= rest - 1ull;
            const int r = bundle_decide_mesh(sc, mesh_lane(sc, j), j, g, L, R);
            if (r == BUNDLE_HIT) {
                dark = true;
                break;
            }
            if (r == BUNDLE_UNKNOWN) keep |= 1ull << j;
        }
    }
    cand = keep;
    if (!decide) return -1;
    return dark ? 0 : (keep ? -1 : S);
}

// Conservative first pass for ALL ambient-occlusion rays of one hit (computeAO, raytracer.cpp:38-78):
// they start at O and only count hits closer than `radius`, so only meshes whose box (bounding
// sphere when posed) comes within `radius` of O can matter.
template <bool kPosed, class SV>
DEV unsigned long long ball_candidates(const SV& sc, V3 O, float radius) {
    const float reach = radius * 1.001f + sc.hdr->mask_slack;  // scale-free: flat_scene.h
    unsigned long long cand = 0ull;
    const int n = sc.n_meshes < 64 ? sc.n_meshes : 64;
    const unsigned long long roots = sc.roots;
#pragma unroll 1
    for (int i = 0; i < n; ++i) {
        if (!((roots >> i) & 1ull)) continue;
        const MeshData m = mesh_uniform(sc, i);
        if (m.flags & MESH_EMPTY) continue;
        bool pass;
        if (kPosed && (m.flags & MESH_ROTATED)) {
            const V3 oc = m.centre - O;
            const float rr = m.radius + reach;
            pass = (m.radius < 0.0f) | !(dot(oc, oc) > rr * rr * 1.001f);
        } else {
            // distance from O to the box, per axis
            const float dx = smax(smax(m.lo.x - O.x, O.x - m.hi.x), 0.0f);
            const float dy = smax(smax(m.lo.y - O.y, O.y - m.hi.y), 0.0f);
            const float dz = smax(smax(m.lo.z - O.z, O.z - m.hi.z), 0.0f);
            pass = !(dx * dx + dy * dy + dz * dz > reach * reach * 1.001f);
        }
        if (pass) cand |= m.group;
    }
    return cand;
}

// ball_candidates for the rays of computeAO, which leave into the hemisphere around the hit's normal N (unit length):
// direction = normalize(T·x + N·y + B·z) with y = sqrt(r1) >= 0 (raytracer.cpp:44-68).  When N is an axis vector
// ±e_a (every hit on an un-posed mesh), T and B come out of the cross products as exact axis vectors, so the
// direction's component along a is y / |.| with N's sign: never towards the other side.  An un-posed box that ends
// before the origin on that side — O_a > hi_a for N = +e_a, O_a < lo_a for N = -e_a — then has both slab distances of
// axis a negative (or the axis is "parallel" with the origin outside): no AO ray of this hit can report it
// (intersection.cpp:222-249), whatever its direction.  Exact; posed meshes and other normals are left alone.
template <bool kPosed, class SV>
DEV unsigned long long hemisphere_candidates(const SV& sc, V3 O, V3 N, float radius) {
    const float reach = radius * 1.001f + sc.hdr->mask_slack;  // scale-free: flat_scene.h
    // the normal as an axis: n_c = ±1 on one axis and ±0 on the others, else no pruning (all limits stay open)
    const bool axial = (__builtin_fabsf(N.x) == 1.0f & N.y == 0.0f & N.z == 0.0f) | (N.x == 0.0f & __builtin_fabsf(N.y) == 1.0f & N.z == 0.0f) |
                       (N.x == 0.0f & N.y == 0.0f & __builtin_fabsf(N.z) == 1.0f);
    // pruned iff up_c·(O_c - hi_c) > 0 or dn_c·(lo_c - O_c) > 0 for some axis (float subtraction keeps the comparison's sign)
    const float upx = (axial & (N.x > 0.0f)) ? 1.0f : 0.0f, dnx = (axial & (N.x < 0.0f)) ? 1.0f : 0.0f;
    const float upy = (axial & (N.y > 0.0f)) ? 1.0f : 0.0f, dny = (axial & (N.y < 0.0f)) ? 1.0f : 0.0f;
    const float upz = (axial & (N.z > 0.0f)) ? 1.0f : 0.0f, dnz = (axial & (N.z < 0.0f)) ? 1.0f : 0.0f;
    unsigned long long cand = 0ull;
    const int n = sc.n_meshes < 64 ? sc.n_meshes : 64;
    const unsigned long long roots = sc.roots;
#pragma unroll 1
    for (int i = 0; i < n; ++i) {
        if (!((roots >> i) & 1ull)) continue;
        const MeshData m = mesh_uniform(sc, i);
        if (m.flags & MESH_EMPTY) continue;
        bool pass;
        if (kPosed && (m.flags & MESH_ROTATED)) {
            const V3 oc = m.centre - O;
            const float rr = m.radius + reach;
            pass = (m.radius < 0.0f) | !(dot(oc, oc) > rr * rr * 1.001f);
            if (pass) cand |= m.group;
            continue;
        }
        const float dx = smax(smax(m.lo.x - O.x, O.x - m.hi.x), 0.0f);
        const float dy = smax(smax(m.lo.y - O.y, O.y - m.hi.y), 0.0f);
        const float dz = smax(smax(m.lo.z - O.z, O.z - m.hi.z), 0.0f);
        pass = !(dx * dx + dy * dy + dz * dz > reach * reach * 1.001f);
        if (!__ballot(pass)) continue;
        unsigned long long grp = m.group;  // uniform; the members are un-posed boxes inside the root's
        while (grp) {
            const int j = __builtin_ctzll(grp);
            grp &= grp - 1ull;
            const FlatMesh& fm = sc.meshes[j];
            const bool rotated = kPosed && (fm.flags & MESH_ROTATED) != 0;  // (a posed mesh is its own root)
            const float behind = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(upx * (O.x - fm.hi[0]), dnx * (fm.lo[0] - O.x)),
                                                                 __builtin_fmaxf(upy * (O.y - fm.hi[1]), dny * (fm.lo[1] - O.y))),
                                                 __builtin_fmaxf(upz * (O.z - fm.hi[2]), dnz * (fm.lo[2] - O.z)));
            if (pass & (rotated | !(behind > 0.0f))) cand |= 1ull << j;
        }
    }
    return cand;
}

// The candidates (bits < 64) whose box holds O strictly inside — un-posed, non-empty meshes only: for those
// any_hit_masked takes mesh_candidate_inside.  Per lane, from the per-lane mesh table.
template <class SV>
DEV unsigned long long origin_inside_boxes(const SV& sc, V3 O, unsigned long long cand) {
    unsigned long long inside = 0ull;
    while (cand) {
        const int j = __builtin_ctzll(cand);
        cand &= cand - 1ull;
        const MeshData m = mesh_lane(sc, j);
        const bool plain = !(m.flags & MESH_EMPTY) && !(SV::kPosed && (m.flags & MESH_ROTATED));
        if (plain & (O.x > m.lo.x) & (O.x < m.hi.x) & (O.y > m.lo.y) & (O.y < m.hi.y) & (O.z > m.lo.z) & (O.z < m.hi.z)) inside |= 1ull << j;
    }
    return inside;
}

// shared copy for the rare sequential paths (AO, very long shadow streams, probes)
DEVCALL bool any_hit_call(SceneView sc, Ray r, float limit) { return any_hit_inline(sc, r, limit); }
template <class SV>
DEV bool any_hit_before(const SV& sc, const Ray& r, float limit) { return any_hit_call(sc.global(), r, limit); }

// ---------------------------------------------------------------------------------------------
// shading.cpp
// ---------------------------------------------------------------------------------------------
template <class SV>
DEV bool in_shadow(const SV& sc, V3 point, V3 normal, V3 light) {  // :14-26
    V3 origin = point + normal * 1e-3f;
    V3 to = light - origin;
    float dist = length(to);
    if (dist < 1e-6f) return false;
    Ray r{origin, vdiv(to, dist)};
    return any_hit_before(sc, r, dist);
}

// How the shadow term of a hit is obtained (raytracer.cpp:107-115 + shading.cpp:28-41,:76-80):
enum ShadowMode : int {
    SHADOW_SOFT = 0,       // softShadows && shadowSamples > 1 && radius >= 1e-4: disk samples, raw normal
    SHADOW_SOFT_POINT = 1, // softShadows && shadowSamples > 1 but radius < 1e-4: one ray, raw normal
    SHADOW_HARD = 2,       // otherwise: shade()'s own test, normalised normal
};
template <class SV>
DEV int shadow_mode(const SV& sc, const mcrt_config& cfg) {
    if (cfg.soft_shadows && cfg.shadow_samples > 1) return (sc.hdr->light_radius < 1e-4f) ? SHADOW_SOFT_POINT : SHADOW_SOFT;
    return SHADOW_HARD;
}

// the disk frame at the light, facing the shaded point (shading.cpp:35-41) — a function of the hit only
struct LightFrame {
    V3 tangent, bitangent;
};
template <class SV>
DEV LightFrame light_frame(const SV& sc, V3 point) {
    V3 lpos = ld3(sc.hdr->light_pos);
    V3 toPoint = normalize(point - lpos);
    V3 tangent = (__builtin_fabsf(toPoint.x) < 0.9f) ? normalize(cross(mk(1, 0, 0), toPoint))
                                                     : normalize(cross(mk(0, 1, 0), toPoint));
    return LightFrame{tangent, cross(toPoint, tangent)};
}
// one stratified-disk light sample position (shading.cpp:46-53)
template <class SV>
DEV V3 light_sample_on_frame(const SV& sc, const LightFrame& f, float d0, float d1) {
    float angle = kTwoPi * d0;
    float rr = sc.hdr->light_radius * sqrt_pos(d1);  // a draw: 0 or at least 2^-32
    // inlined libm kernels: calls here would serialise the independent samples of a hit
    float sn, cs;
    mcrt_sincosf(angle, &sn, &cs);
    V3 off = f.tangent * (rr * cs) + f.bitangent * (rr * sn);
    return ld3(sc.hdr->light_pos) + off;
}
template <class SV>
DEV V3 light_sample_position(const SV& sc, V3 point, float d0, float d1) {
    return light_sample_on_frame(sc, light_frame(sc, point), d0, d1);
}
template <class SV>
DEV bool light_sample_visible(const SV& sc, V3 point, V3 normal, float d0, float d1) {
    return !in_shadow(sc, point, normal, light_sample_position(sc, point, d0, d1));
}
// isInShadow (:14-26) with the scene scan inlined at the call site (hard-shadow rays of `lit` / `shadow`)
template <class SV>
DEV bool in_shadow_inline(const SV& sc, V3 point, V3 normal, V3 light) {
    V3 origin = point + normal * 1e-3f;
    V3 to = light - origin;
    float dist = length(to);
    if (dist < 1e-6f) return false;
    Ray r{origin, vdiv(to, dist)};
    return any_hit_inline(sc, r, dist);
}

// isInShadow for one of the S rays of a hit whose bundle mask is known
template <class SV>
DEV bool in_shadow_masked(const SV& sc, V3 point, V3 normal, V3 light, unsigned long long cand, unsigned long long inside = 0ull) {
    V3 origin = point + normal * 1e-3f;
    V3 to = light - origin;
    float dist = length(to);
    if (dist < 1e-6f) return false;
    Ray r{origin, vdiv(to, dist)};
    return any_hit_masked(sc, r, dist, cand, inside);
}

// computeSoftShadow :28-60, sequential form (probes; `lit` spreads the samples over lanes)
template <class SV>
DEV float soft_shadow(const SV& sc, V3 point, V3 normal, int samples, uint32_t seed, uint32_t* mt_storage) {
    V3 lpos = ld3(sc.hdr->light_pos);
    if (samples <= 1 || sc.hdr->light_radius < 1e-4f) return in_shadow(sc, point, normal, lpos) ? 0.0f : 1.0f;
    HitRng rng;
    rng.seed(seed, 2 * samples, mt_storage);
    int lit = 0;
    for (int i = 0; i < samples; ++i) {
        float d0 = rng.uniform();
        float d1 = rng.uniform();
        if (light_sample_visible(sc, point, normal, d0, d1)) ++lit;
    }
    return static_cast<float>(lit) / static_cast<float>(samples);
}

// shade :62-96 with ShadingParams{} (kd .75, ks .15, ambient .20, shininess 16 — shading.h:9-14;
// renderTile always passes the defaults, tile_renderer.cpp:106-107) and the visibility term given.
template <class SV>
DEV C4 shade(const SV& sc, const Hit& hit, V3 viewDir, float vis) {
    const float kd = 0.75f, ks = 0.15f, ambient = 0.20f, shininess = 16.0f;
    C4 tex = hit.tex;
    V3 lpos = ld3(sc.hdr->light_pos);
    const float* lc = sc.hdr->light_color;
    V3 L = normalize(lpos - hit.p);
    V3 N = normalize(hit.n);
    V3 V = normalize(viewDir);
    float ndl = smax(0.0f, dot(N, L));
    float kdiff = kd * ndl * vis;
    V3 H = normalize(L + V);
    float ndh = smax(0.0f, dot(N, H));
    float kspec = ks * dev_powf(ndh, shininess) * vis;
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
template <class SV>
DEV C4 background(const SV& sc, const mcrt_config& cfg, float u, float v) {  // :16-34
    if (cfg.gradient_bg) {
        float cx = u - 0.5f, cy = v - 0.5f;
        float dist = sqrt_pos(cx * cx + cy * cy) * 2.0f * cfg.gradient_scale;  // u, v in [0, 1]: the argument is 0 or in [2^-50, 0.5]
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

template <class SV>
DEV float ambient_occlusion(const SV& sc, V3 point, V3 normal, int samples, float radius,
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
        float sinT = sqrt_pos(1.0f - r1);
        float cosT = sqrt_pos(r1);
        float phi = kTwoPi * r2;
        const SinCos sc_phi = dev_sincosf(phi);
        V3 local = mk(sinT * sc_phi.c, cosT, sinT * sc_phi.s);
        V3 world = normalize(T * local.x + N * local.y + B * local.z);
        Ray r{point + N * 1e-3f, world};
        if (any_hit_before(sc, r, radius)) ++occluded;
    }
    return 1.0f - static_cast<float>(occluded) / static_cast<float>(samples);
}

// colour of a hit level once its visibility term is known: shade + AO (raytracer.cpp:117-131)
template <class SV>
DEV C4 level_color(const SV& sc, const mcrt_config& cfg, V3 ray_origin, const Hit& hit, int depth, float vis,
                   uint32_t* mt_storage) {
    V3 view = normalize(ray_origin - hit.p);
    C4 c = shade(sc, hit, view, vis);
    if (cfg.ao_enabled && depth == 0) {
        float ao = ambient_occlusion(sc, hit.p, hit.n, cfg.ao_samples, cfg.ao_radius, ao_seed(hit.p), mt_storage);
        float k = 1.0f - cfg.ao_intensity * (1.0f - ao);
        c.r *= k;
        c.g *= k;
        c.b *= k;
    }
    return c;
}

// visibility term of a hit, sequential form
template <class SV>
DEV float hit_visibility(const SV& sc, const mcrt_config& cfg, const Hit& hit, int depth, uint32_t* mt_storage) {
    const int mode = shadow_mode(sc, cfg);
    V3 lpos = ld3(sc.hdr->light_pos);
    if (mode == SHADOW_SOFT) return soft_shadow(sc, hit.p, hit.n, cfg.shadow_samples, shadow_seed(hit.p, depth), mt_storage);
    if (mode == SHADOW_SOFT_POINT) return in_shadow(sc, hit.p, hit.n, lpos) ? 0.0f : 1.0f;
    return in_shadow(sc, hit.p, normalize(hit.n), lpos) ? 0.0f : 1.0f;
}

// reflection ray of a hit (:133-139): from the incoming direction, the hit point and its normal
DEV Ray reflect_ray(V3 dir, V3 point, V3 normal) {
    V3 N = normalize(normal);
    V3 D = normalize(dir);
    V3 R = normalize(D - N * (2.0f * dot(D, N)));
    return Ray{point + N * 1e-3f, R};
}
DEV Ray reflect_ray(const Ray& ray, const Hit& hit) { return reflect_ray(ray.d, hit.p, hit.n); }

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

constexpr int kMaxStack = 16;  // levels kept in per-lane scratch; deeper → per-thread HBM slice

// RayTracer::traceRay (:82-148) for a ray whose depth-`depth` hit is already known, as a loop:
// walk down while rays keep hitting (level colours pushed), then fold back to front.  Sequential
// per-lane form (probes); the render pipeline runs the same steps as kernels over hit records.
template <class SV>
DEV C4 trace_from_hit(const SV& sc, const mcrt_config& cfg, Ray ray, Hit hit, int depth,
                      C4* stack, uint32_t* mt_storage) {
    const float* b = sc.hdr->background;
    const C4 flat_bg{b[0], b[1], b[2], b[3]};
    const int max_b = cfg.max_bounces;
    int top = 0;
    C4 tail;
    for (;;) {
        float vis = hit_visibility(sc, cfg, hit, depth, mt_storage);
        C4 c = level_color(sc, cfg, ray.o, hit, depth, vis, mt_storage);
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
template <class SV>
DEV Ray camera_ray(const SV& sc, float u, float v, float aspect) {
    const FlatHeader* h = sc.hdr;
    float halfH = h->cam_half_h;
    float halfW = halfH * aspect;
    float su = (2.0f * u - 1.0f) * halfW;
    float sv = (2.0f * (1.0f - v) - 1.0f) * halfH;
    V3 dir = normalize(ld3(h->cam_fwd) + ld3(h->cam_right) * su + ld3(h->cam_up) * sv);
    return Ray{ld3(h->cam_pos), dir};
}

template <class SV>
DEV Ray lens_ray(const SV& sc, float u, float v, float aspect, float aperture, float focusDist,
                 float d0, float d1) {
    Ray pin = camera_ray(sc, u, v, aspect);
    if (aperture < 1e-6f) return pin;
    const FlatHeader* h = sc.hdr;
    V3 focus = pin.o + pin.d * focusDist;
    float angle = kTwoPi * d0;
    float radius = aperture * sqrt_pos(d1);
    const SinCos sc_lens = dev_sincosf(angle);
    float lx = radius * sc_lens.c;
    float ly = radius * sc_lens.s;
    V3 origin = ld3(h->cam_pos) + (ld3(h->cam_right) * lx + ld3(h->cam_up) * ly);
    return Ray{origin, normalize(focus - origin)};
}

}  // namespace rt

#endif
