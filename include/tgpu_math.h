/*
 * tgpu_math.h -- the arithmetic contract of the hot path.
 *
 * The reference calls libm (std::sin/cos/exp/log/atan2/acos) from glibc
 * (SURVEY.md 8c "third-party arithmetic"); ROCm's device libm rounds
 * differently in the last ulp, and a one-ulp change in a sampled direction is
 * enough to flip a discrete outcome a few bounces later.  To make "GPU ==
 * CPU oracle" a bit-exact statement, both sides evaluate the transcendental
 * functions below: branch-free-ish single-precision polynomial kernels written
 * only in terms of IEEE add/mul/fma/div/sqrt and integer bit operations, which
 * are correctly rounded and therefore identical on x86-64 and gfx950.
 *
 * Compile both sides with -ffp-contract=off: the only fused operations are the
 * explicit TGPU_FMA calls.  Accuracy (tests/test_math.py): <= 2 ulp vs float64
 * libm over the ranges the path uses.
 *
 * Algorithms: classic Cody-Waite range reduction + minimax polynomials in the
 * style of Cephes' single-precision library (S. Moshier), re-derived here.
 */
#ifndef TGPU_MATH_H_
#define TGPU_MATH_H_

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define TGPU_HD __host__ __device__ __forceinline__
#else
#define TGPU_HD static inline
#endif

#define TGPU_FMA(a, b, c) __builtin_fmaf((a), (b), (c))

TGPU_HD float tgpu_bits_to_float(uint32_t u) { float f; __builtin_memcpy(&f, &u, 4); return f; }
TGPU_HD uint32_t tgpu_float_to_bits(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); return u; }

/* sin and cos of x for |x| < ~1e4 (the path uses [-2pi, 2pi]). */
TGPU_HD void tgpu_sincosf(float x, float *s, float *c)
{
    const float TWO_OVER_PI = 0.636619772367581343f;
    /* pi/2 split in three parts (Cody-Waite), first two have trailing zero bits */
    const float P1 = 1.5703125f;
    const float P2 = 4.837512969970703125e-4f;
    const float P3 = 7.54978995489188e-8f;

    float q = __builtin_rintf(x*TWO_OVER_PI);
    float r = TGPU_FMA(q, -P1, x);
    r = TGPU_FMA(q, -P2, r);
    r = TGPU_FMA(q, -P3, r);
    int n = (int)q;

    float r2 = r*r;
    /* sin(r) ~ r + r^3 * S(r^2) on [-pi/4, pi/4] */
    float sp = TGPU_FMA(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    sp = TGPU_FMA(sp, r2, -1.6666654611e-1f);
    float sr = TGPU_FMA(sp*r2, r, r);
    /* cos(r) ~ 1 - r^2/2 + r^4 * C(r^2) */
    float cp = TGPU_FMA(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    cp = TGPU_FMA(cp, r2, 4.166664568298827e-2f);
    float cr = TGPU_FMA(cp*r2, r2, TGPU_FMA(r2, -0.5f, 1.0f));

    float ss = (n & 1) ? cr : sr;
    float cc = (n & 1) ? sr : cr;
    if (n & 2) ss = -ss;
    if ((n + 1) & 2) cc = -cc;
    *s = ss;
    *c = cc;
}

TGPU_HD float tgpu_sinf(float x) { float s, c; tgpu_sincosf(x, &s, &c); return s; }
TGPU_HD float tgpu_cosf(float x) { float s, c; tgpu_sincosf(x, &s, &c); return c; }

/* exp(x); returns 0 below -87.3 (no subnormal results), +inf above 88.72. */
TGPU_HD float tgpu_expf(float x)
{
    const float LOG2E = 1.44269504088896341f;
    const float LN2_HI = 0.693359375f;
    const float LN2_LO = -2.12194440e-4f;
    if (!(x > -87.3f)) return x != x ? x : 0.0f;
    if (x > 88.72f) return tgpu_bits_to_float(0x7F800000u);

    float n = __builtin_rintf(x*LOG2E);
    float r = TGPU_FMA(n, -LN2_HI, x);
    r = TGPU_FMA(n, -LN2_LO, r);
    float r2 = r*r;
    float p = TGPU_FMA(r, 1.9875691500e-4f, 1.3981999507e-3f);
    p = TGPU_FMA(p, r, 8.3334519073e-3f);
    p = TGPU_FMA(p, r, 4.1665795894e-2f);
    p = TGPU_FMA(p, r, 1.6666665459e-1f);
    p = TGPU_FMA(p, r, 5.0000001201e-1f);
    float e = TGPU_FMA(p, r2, r) + 1.0f;
    /* scale by 2^n in two steps so n = 128 / n = -126 stay representable */
    int ni = (int)n;
    int n1 = ni/2, n2 = ni - n1;
    e *= tgpu_bits_to_float((uint32_t)(n1 + 127) << 23);
    e *= tgpu_bits_to_float((uint32_t)(n2 + 127) << 23);
    return e;
}

/* natural log for x > 0 (normal numbers); log(0) = -inf, log(<0) = NaN. */
TGPU_HD float tgpu_logf(float x)
{
    const float SQRTH = 0.70710678118654752440f;
    if (x == 0.0f) return tgpu_bits_to_float(0xFF800000u);
    if (!(x > 0.0f)) return tgpu_bits_to_float(0x7FC00000u);
    if (x == tgpu_bits_to_float(0x7F800000u)) return x;
    uint32_t ix = tgpu_float_to_bits(x);
    int e = 0;
    if (ix < 0x00800000u) { /* subnormal: scale up */
        x *= 8388608.0f;
        ix = tgpu_float_to_bits(x);
        e = -23;
    }
    e += (int)(ix >> 23) - 126;
    float m = tgpu_bits_to_float((ix & 0x007FFFFFu) | 0x3F000000u); /* [0.5, 1) */
    if (m < SQRTH) { e -= 1; m = m + m - 1.0f; } else { m = m - 1.0f; }
    float z = m*m;
    float p = TGPU_FMA(m, 7.0376836292e-2f, -1.1514610310e-1f);
    p = TGPU_FMA(p, m, 1.1676998740e-1f);
    p = TGPU_FMA(p, m, -1.2420140846e-1f);
    p = TGPU_FMA(p, m, 1.4249322787e-1f);
    p = TGPU_FMA(p, m, -1.6668057665e-1f);
    p = TGPU_FMA(p, m, 2.0000714765e-1f);
    p = TGPU_FMA(p, m, -2.4999993993e-1f);
    p = TGPU_FMA(p, m, 3.3333331174e-1f);
    float y = p*m*z;
    float fe = (float)e;
    y = TGPU_FMA(fe, -2.12194440e-4f, y);
    y = TGPU_FMA(z, -0.5f, y);
    float r = m + y;
    r = TGPU_FMA(fe, 0.693359375f, r);
    return r;
}

/* tan(x) = sin/cos of the same reduction (<= 4 ulp away from the poles); pow(x, y) = exp(y*log(x)) for x > 0 (x == 0: 0 for y > 0,
 * 1 for y == 0): relative error ~ |y log x| ulp -- both are used by the Oren-Nayar and Phong BSDFs only, identically on CPU and GPU. */
TGPU_HD float tgpu_tanf(float x) { float s, c; tgpu_sincosf(x, &s, &c); return s/c; }
TGPU_HD float tgpu_powf(float x, float y)
{
    if (y == 0.0f) return 1.0f;
    if (x == 0.0f) return y > 0.0f ? 0.0f : tgpu_bits_to_float(0x7F800000u);
    return tgpu_expf(y*tgpu_logf(x));
}

/* atan(x) for any finite x */
TGPU_HD float tgpu_atanf(float x)
{
    const float PIO2 = 1.5707963267948966192f;
    const float PIO4 = 0.7853981633974483096f;
    float ax = x < 0.0f ? -x : x;
    float y, t;
    if (ax > 2.414213562373095f) {        /* tan(3pi/8) */
        y = PIO2; t = -1.0f/ax;
    } else if (ax > 0.4142135623730950f) { /* tan(pi/8) */
        y = PIO4; t = (ax - 1.0f)/(ax + 1.0f);
    } else {
        y = 0.0f; t = ax;
    }
    float z = t*t;
    float p = TGPU_FMA(z, 8.05374449538e-2f, -1.38776856032e-1f);
    p = TGPU_FMA(p, z, 1.99777106478e-1f);
    p = TGPU_FMA(p, z, -3.33329491539e-1f);
    y += TGPU_FMA(p*z, t, t);
    return x < 0.0f ? -y : y;
}

/* atan2(y, x) with the usual quadrant conventions; atan2(0,0) = 0. */
TGPU_HD float tgpu_atan2f(float y, float x)
{
    const float PI = 3.14159265358979323846f;
    const float PIO2 = 1.5707963267948966192f;
    if (x == 0.0f) {
        if (y > 0.0f) return PIO2;
        if (y < 0.0f) return -PIO2;
        return 0.0f;
    }
    float a = tgpu_atanf(y/x);
    if (x < 0.0f) a = (y < 0.0f) ? a - PI : a + PI;
    return a;
}

/* asin / acos on [-1, 1]; arguments outside are clamped. */
TGPU_HD float tgpu_asinf(float x)
{
    const float PIO2 = 1.5707963267948966192f;
    float a = x < 0.0f ? -x : x;
    if (a > 1.0f) a = 1.0f;
    int big = a > 0.5f;
    float z, t;
    if (big) { z = 0.5f*(1.0f - a); t = __builtin_sqrtf(z); } else { t = a; z = a*a; }
    float p = TGPU_FMA(z, 4.2163199048e-2f, 2.4181311049e-2f);
    p = TGPU_FMA(p, z, 4.5470025998e-2f);
    p = TGPU_FMA(p, z, 7.4953002686e-2f);
    p = TGPU_FMA(p, z, 1.6666752422e-1f);
    float r = TGPU_FMA(p*z, t, t);
    if (big) r = PIO2 - (r + r);
    return x < 0.0f ? -r : r;
}

TGPU_HD float tgpu_acosf(float x)
{
    const float PI = 3.14159265358979323846f;
    const float PIO2 = 1.5707963267948966192f;
    if (x > 0.5f) {
        if (x > 1.0f) x = 1.0f;
        return 2.0f*tgpu_asinf(__builtin_sqrtf(0.5f*(1.0f - x)));
    }
    if (x < -0.5f) {
        if (x < -1.0f) x = -1.0f;
        return PI - 2.0f*tgpu_asinf(__builtin_sqrtf(0.5f*(1.0f + x)));
    }
    return PIO2 - tgpu_asinf(x);
}

#endif /* TGPU_MATH_H_ */
