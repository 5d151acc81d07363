/* mcrt_detmath.h — deterministic float32 sinf / cosf / powf shared by the HIP kernels,
 * the host-side scene flattener and the CPU oracle.
 *
 * Why this exists (SURVEY.md §7 hard part 1): the reference calls libm `std::cos/std::sin/std::pow`
 * on floats at
 *   - /root/reference/src/raytracer/shading.cpp:51,54   (soft-shadow disk sample, per shadow ray)
 *   - /root/reference/src/raytracer/shading.cpp:90      (Blinn-Phong specular, powf(x, 16))
 *   - /root/reference/src/raytracer/raytracer.cpp:62-64 (AO hemisphere sample)
 *   - /root/reference/src/raytracer/tile_renderer.cpp:61-62 (DOF lens sample)
 *   - /root/reference/src/raytracer/intersection.cpp:17-33  (posed-mesh rotation, per mesh)
 * glibc's sinf/cosf/powf are not correctly rounded, so a GPU kernel can only be bit-exact with
 * the reference if it evaluates the SAME algorithm.  glibc 2.35 (the libm of this image) uses the
 * ARM "optimized-routines" algorithms: double-precision polynomial kernels with a final
 * double→float rounding.  On x86-64 hosts with FMA+AVX2 (this container and every modern GPU
 * host) the IFUNC resolver selects the FMA-compiled bodies, in which every `a*b+c` of the
 * polynomial/reduction is a fused multiply-add.  The routines below restate that algorithm with
 * explicit fma() so CPU (gcc) and GPU (hipcc, v_fma_f64) produce identical bits; the constants are
 * the published table values (verified against the .rodata of this image's libm.so.6).
 * tools/check_detmath.cpp compares them with the system libm over every float (sinf/cosf) and
 * over x in [0,2], y=16 plus random (x,y) (powf); tests/test_detmath.py runs a sampled version.
 *
 * Domain notes: mcrt_sinf/mcrt_cosf cover all finite floats (NaN/Inf → NaN).  mcrt_powf covers
 * x >= 0 (incl. 0, subnormal, +Inf) with finite y > 0 — the only way the render path calls it
 * (x = max(0, N·H), y = shininess > 0); other arguments return NaN.
 */
#ifndef MCRT_DETMATH_H
#define MCRT_DETMATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MCRT_HD __host__ __device__ inline
#else
#define MCRT_HD static inline
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define MCRT_CONST_TAB __constant__ static const
#else
#define MCRT_CONST_TAB static const
#endif

/* ---- bit casts -------------------------------------------------------------------------- */
MCRT_HD uint32_t mcrt_f2u(float f) { union { float f; uint32_t u; } c; c.f = f; return c.u; }
MCRT_HD float mcrt_u2f(uint32_t u) { union { float f; uint32_t u; } c; c.u = u; return c.f; }
MCRT_HD uint64_t mcrt_d2u(double d) { union { double d; uint64_t u; } c; c.d = d; return c.u; }
MCRT_HD double mcrt_u2d(uint64_t u) { union { double d; uint64_t u; } c; c.u = u; return c.d; }

/* fused multiply-add in double, single rounding on both host and device */
MCRT_HD double mcrt_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

/* ---- sinf / cosf ------------------------------------------------------------------------ */
/* polynomial coefficients: [0] for cos/sin, [1] for -cos/sin (quadrants 2,3) */
#define MCRT_SC_HPI_INV 0x1.45f306dc9c883p+23 /* 2/pi * 2^24 */
#define MCRT_SC_HPI 0x1.921fb54442d18p+0      /* pi/2 */
#define MCRT_SC_C0 0x1.0000000000000p+0
#define MCRT_SC_C1 -0x1.ffffffd0c621cp-2
#define MCRT_SC_C2 0x1.55553e1068f19p-5
#define MCRT_SC_C3 -0x1.6c087e89a359dp-10
#define MCRT_SC_C4 0x1.99343027bf8c3p-16
#define MCRT_SC_S1 -0x1.555545995a603p-3
#define MCRT_SC_S2 0x1.1107605230bc4p-7
#define MCRT_SC_S3 -0x1.994eb3774cf24p-13
#define MCRT_SC_PI63 0x1.921fb54442d18p-62 /* pi/2 * 2^-62 */

/* sin kernel: xs = signed reduced argument, x2 = its square */
MCRT_HD float mcrt_sc_sinpoly(double xs, double x2) {
    double x3 = xs * x2;
    double s1 = mcrt_fma(x2, MCRT_SC_S3, MCRT_SC_S2);
    double x7 = x3 * x2;
    double s = mcrt_fma(x3, MCRT_SC_S1, xs);
    return (float)mcrt_fma(x7, s1, s);
}
/* cos kernel; neg selects the negated-cosine coefficient set */
MCRT_HD float mcrt_sc_cospoly(double x2, int neg) {
    double sg = neg ? -1.0 : 1.0;
    double x4 = x2 * x2;
    double c2 = mcrt_fma(x2, sg * MCRT_SC_C4, sg * MCRT_SC_C3);
    double c1 = mcrt_fma(x2, sg * MCRT_SC_C1, sg * MCRT_SC_C0);
    double x6 = x4 * x2;
    double c = mcrt_fma(x4, sg * MCRT_SC_C2, c1);
    return (float)mcrt_fma(x6, c2, c);
}

MCRT_CONST_TAB uint32_t mcrt_inv_pio4[24] = {
    0xa2u,       0xa2f9u,     0xa2f983u,   0xa2f9836eu, 0xf9836e4eu, 0x836e4e44u,
    0x6e4e4415u, 0x4e441529u, 0x441529fcu, 0x1529fc27u, 0x29fc2757u, 0xfc2757d1u,
    0x2757d1f5u, 0x57d1f534u, 0xd1f534ddu, 0xf534ddc0u, 0x34ddc0dbu, 0xddc0db62u,
    0xc0db6295u, 0xdb629599u, 0x6295993cu, 0x95993c43u, 0x993c4390u, 0x3c439041u};

/* |x| < 120: n = round(x * 2/pi), returns x - n*pi/2 */
MCRT_HD double mcrt_sc_reduce_fast(double x, int* np) {
    double r = x * MCRT_SC_HPI_INV;
    int n = ((int32_t)r + 0x800000) >> 24;
    *np = n;
    return mcrt_fma(-(double)n, MCRT_SC_HPI, x);
}
/* |x| >= 120 (finite): 4/pi table multiply */
MCRT_HD double mcrt_sc_reduce_large(uint32_t xi, int* np) {
    const uint32_t* arr = &mcrt_inv_pio4[(xi >> 26) & 15];
    int shift = (int)((xi >> 23) & 7);
    uint64_t n, res0, res1, res2;
    xi = (xi & 0xffffffu) | 0x800000u;
    xi <<= shift;
    res0 = (uint64_t)(uint32_t)(xi * arr[0]);
    res1 = (uint64_t)xi * arr[4];
    res2 = (uint64_t)xi * arr[8];
    res0 = (res2 >> 32) | (res0 << 32);
    res0 += res1;
    n = (res0 + (1ULL << 61)) >> 62;
    res0 -= n << 62;
    *np = (int)n;
    return (double)(int64_t)res0 * MCRT_SC_PI63;
}

MCRT_HD float mcrt_sc_eval(float y, int want_cos) {
    double x = (double)y;
    uint32_t top = (mcrt_f2u(y) >> 20) & 0x7ffu;
    int n_poly, n_sign;
    if (top < 0x3f4u) { /* |y| < pi/4 */
        double x2 = x * x;
        if (top < 0x398u) /* |y| < 2^-12 */
            return want_cos ? 1.0f : y;
        return want_cos ? mcrt_sc_cospoly(x2, 0) : mcrt_sc_sinpoly(x, x2);
    }
    if (top < 0x42fu) { /* |y| < 120 */
        x = mcrt_sc_reduce_fast(x, &n_poly);
        n_sign = n_poly;
    } else if (top < 0x7f8u) { /* finite: the sign of y joins the quadrant, not the polynomial pick */
        uint32_t xi = mcrt_f2u(y);
        x = mcrt_sc_reduce_large(xi, &n_poly);
        n_sign = n_poly + (int)(xi >> 31);
    } else {
        return mcrt_u2f(0x7fc00000u); /* Inf/NaN */
    }
    {
        int q = n_sign & 3; /* quadrant sign table {+,-,-,+} */
        double sgn = (q == 1 || q == 2) ? -1.0 : 1.0;
        int odd = (want_cos ? (n_poly ^ 1) : n_poly) & 1;
        if (odd == 0)
            return mcrt_sc_sinpoly(x * sgn, x * x);
        return mcrt_sc_cospoly(x * x, (n_sign & 2) != 0);
    }
}
MCRT_HD float mcrt_sinf(float y) { return mcrt_sc_eval(y, 0); }
MCRT_HD float mcrt_cosf(float y) { return mcrt_sc_eval(y, 1); }

/* sinf and cosf of the same argument with ONE argument reduction and one evaluation of each
 * polynomial: for a given y both functions share n, the reduced x and x*x; sinf takes the sin-type
 * polynomial when n is even and the cos-type one when n is odd, cosf the other one (mcrt_sc_eval).
 * The results are the values mcrt_sinf(y) / mcrt_cosf(y) return, bit for bit (tools/check_detmath.cpp
 * sweeps all 2^32 arguments).  |y| >= 120 and non-finite y take the separate functions. */
MCRT_HD void mcrt_sincosf(float y, float* sin_out, float* cos_out) {
    uint32_t top = (mcrt_f2u(y) >> 20) & 0x7ffu;
    if (top >= 0x42fu) {
        *sin_out = mcrt_sc_eval(y, 0);
        *cos_out = mcrt_sc_eval(y, 1);
        return;
    }
    {
        double x = (double)y;
        int n = 0;
        double xr = x;
        if (top >= 0x3f4u) xr = mcrt_sc_reduce_fast(x, &n);
        {
            int q = n & 3;
            double sgn = (q == 1 || q == 2) ? -1.0 : 1.0;
            double x2 = xr * xr;
            float sp = mcrt_sc_sinpoly(xr * sgn, x2);
            float cp = mcrt_sc_cospoly(x2, (n & 2) != 0);
            int odd = n & 1;
            float sv = odd ? cp : sp;
            float cv = odd ? sp : cp;
            if (top < 0x398u) { /* |y| < 2^-12 */
                sv = y;
                cv = 1.0f;
            }
            *sin_out = sv;
            *cos_out = cv;
        }
    }
}

/* ---- powf -------------------------------------------------------------------------------- */
MCRT_CONST_TAB double mcrt_pow_invc[16] = {
    0x1.661ec79f8f3bep+0, 0x1.571ed4aaf883dp+0, 0x1.49539f0f010b0p+0, 0x1.3c995b0b80385p+0,
    0x1.30d190c8864a5p+0, 0x1.25e227b0b8ea0p+0, 0x1.1bb4a4a1a343fp+0, 0x1.12358f08ae5bap+0,
    0x1.0953f419900a7p+0, 0x1.0000000000000p+0, 0x1.e608cfd9a47acp-1, 0x1.ca4b31f026aa0p-1,
    0x1.b2036576afce6p-1, 0x1.9c2d163a1aa2dp-1, 0x1.886e6037841edp-1, 0x1.767dcf5534862p-1};
MCRT_CONST_TAB double mcrt_pow_logc[16] = {
    -0x1.efec65b963019p-2, -0x1.b0b6832d4fca4p-2, -0x1.7418b0a1fb77bp-2, -0x1.39de91a6dcf7bp-2,
    -0x1.01d9bf3f2b631p-2, -0x1.97c1d1b3b7af0p-3, -0x1.2f9e393af3c9fp-3, -0x1.960cbbf788d5cp-4,
    -0x1.a6f9db6475fcep-5, 0x0.0p+0,              0x1.338ca9f24f53dp-4,  0x1.476a9543891bap-3,
    0x1.e840b4ac4e4d2p-3,  0x1.40645f0c6651cp-2,  0x1.88e9c2c1b9ff8p-2,  0x1.ce0a44eb17bccp-2};
MCRT_CONST_TAB uint64_t mcrt_exp2_tab[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};

#define MCRT_POW_A0 0x1.27616c9496e0bp-2
#define MCRT_POW_A1 -0x1.71969a075c67ap-2
#define MCRT_POW_A2 0x1.ec70a6ca7baddp-2
#define MCRT_POW_A3 -0x1.7154748bef6c8p-1
#define MCRT_POW_A4 0x1.71547652ab82bp+0
#define MCRT_EXP2_SHIFT 0x1.8p+47 /* 0x1.8p52 / 32 */
#define MCRT_EXP2_C0 0x1.c6af84b912394p-5
#define MCRT_EXP2_C1 0x1.ebfce50fac4f3p-3
#define MCRT_EXP2_C2 0x1.62e42ff0c52d6p-1

MCRT_HD float mcrt_powf(float x, float y) {
    uint32_t ix = mcrt_f2u(x);
    uint32_t iy = mcrt_f2u(y);
    /* supported domain: x >= 0, finite y > 0 */
    if ((iy >> 31) || (iy & 0x7fffffffu) == 0 || (iy & 0x7f800000u) == 0x7f800000u)
        return mcrt_u2f(0x7fc00000u);
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
        if (ix == 0) return 0.0f;                    /* +0^y = +0 */
        if (ix == 0x7f800000u) return x;             /* +Inf^y = +Inf */
        if (ix > 0x7f800000u) return mcrt_u2f(0x7fc00000u); /* NaN or negative */
        /* subnormal x: normalise so the exponent goes negative */
        ix = mcrt_f2u(x * 0x1p23f);
        ix &= 0x7fffffffu;
        ix -= 23u << 23;
    }
    /* log2(x) in double: x = 2^k z, z in [OFF, 2 OFF), 16 sub-intervals */
    uint32_t tmp = ix - 0x3f330000u;
    int i = (int)((tmp >> 19) & 15u);
    uint32_t top = tmp & 0xff800000u;
    uint32_t iz = ix - top;
    int k = (int32_t)top >> 23;
    double z = (double)mcrt_u2f(iz);
    double r = mcrt_fma(z, mcrt_pow_invc[i], -1.0);
    double y0 = mcrt_pow_logc[i] + (double)k;
    double r2 = r * r;
    double p0 = mcrt_fma(MCRT_POW_A0, r, MCRT_POW_A1);
    double p1 = mcrt_fma(MCRT_POW_A2, r, MCRT_POW_A3);
    double r4 = r2 * r2;
    double q = mcrt_fma(MCRT_POW_A4, r, y0);
    q = mcrt_fma(p1, r2, q);
    double logx = mcrt_fma(p0, r4, q);
    double ylogx = (double)y * logx;
    if (((mcrt_d2u(ylogx) >> 47) & 0xffffu) >= (0x405f800000000000ull >> 47)) { /* |ylogx| >= 126 */
        if (ylogx > 0x1.fffffffd1d571p+6) return mcrt_u2f(0x7f800000u); /* overflow */
        if (ylogx <= -150.0) return 0.0f;                               /* underflow */
        if (ylogx < -149.0) return mcrt_u2f(1u);                        /* 0x1.4p-75f squared → 2^-149 */
    }
    /* exp2(ylogx): ylogx = kk/32 + rr */
    double kd = ylogx + MCRT_EXP2_SHIFT;
    uint64_t ki = mcrt_d2u(kd);
    kd -= MCRT_EXP2_SHIFT;
    double rr = ylogx - kd;
    uint64_t t = mcrt_exp2_tab[ki & 31u];
    t += ki << 47;
    double s = mcrt_u2d(t);
    double zz = mcrt_fma(MCRT_EXP2_C0, rr, MCRT_EXP2_C1);
    double rr2 = rr * rr;
    double yy = mcrt_fma(MCRT_EXP2_C2, rr, 1.0);
    yy = mcrt_fma(zz, rr2, yy);
    return (float)(yy * s);
}

#endif /* MCRT_DETMATH_H */
