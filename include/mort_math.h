/*
 * mort_math.h -- compiler- and libm-independent elementary functions.
 *
 * The reference calls CUDA's libm (sqrtf, sinf, cosf, acosf, atan2f, logf,
 * f64 sin, pow(float,int)) from device code (vec3.cuh:181-193,208-212;
 * objects.cuh:103-104,141-142,420; textures.cuh:201).  CUDA's, glibc's and
 * ROCm OCML's versions of these differ in the last ULPs, and a path tracer
 * amplifies a 1-ULP direction change into a different hit.  So that the host
 * oracle (gcc) and the gfx950 kernels (hipcc) produce bit-identical images,
 * both compute every non-IEEE-exact function from the definitions in this
 * file, which use only IEEE-754 correctly rounded operations (+ - * / sqrt,
 * conversions) in a fixed order.  Both sides are compiled with
 * -ffp-contract=off.  Each fp32 function evaluates in fp64 and rounds once,
 * which puts it within 1 ULP of the correctly rounded result
 * (tests/test_math.py measures this against glibc).
 *
 * Domains: sin/cos |x| < 1e5 (callers pass phi in [0, 2*pi] and Perlin phases
 * of a few hundred); others full range with IEEE NaN/inf behaviour.
 */
#ifndef MORT_MATH_H
#define MORT_MATH_H

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define MORT_HD __host__ __device__ __forceinline__
#else
#define MORT_HD static inline __attribute__((always_inline))
#endif

#if defined(__clang__)
#pragma STDC FP_CONTRACT OFF
#endif

/* ---- bit casts ---- */
MORT_HD uint64_t mort_d2u(double x) { union { double d; uint64_t u; } c; c.d = x; return c.u; }
MORT_HD double mort_u2d(uint64_t x) { union { double d; uint64_t u; } c; c.u = x; return c.d; }

/* ---- exact helpers (IEEE-exact on both compilers) ---- */
MORT_HD float mort_sqrtf(float x) { return __builtin_sqrtf(x); }
MORT_HD double mort_sqrt(double x) { return __builtin_sqrt(x); }
MORT_HD float mort_fabsf(float x) { return __builtin_fabsf(x); }
MORT_HD double mort_fabs(double x) { return __builtin_fabs(x); }
MORT_HD float mort_floorf(float x) { return __builtin_floorf(x); }

/* CUDA min(double,double)/max and fmaxf: IEEE fmin/fmax (a NaN loses). */
MORT_HD double mort_fmin(double a, double b) { return (a != a) ? b : ((b != b) ? a : (a < b ? a : b)); }
MORT_HD float mort_fmaxf(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a > b ? a : b)); }

/* float/double -> int as CUDA's cvt.rzi.s32 does it: truncate, saturate, NaN -> 0.
 * (C leaves out-of-range conversions undefined; x86 and gfx950 differ there.) */
MORT_HD int mort_f2i(float x) {
    if (x != x) return 0;
    if (x >= 2147483648.0f) return 2147483647;
    if (x <= -2147483648.0f) return (-2147483647 - 1);
    return (int)x;
}
MORT_HD int mort_d2i(double x) {
    if (x != x) return 0;
    if (x >= 2147483648.0) return 2147483647;
    if (x <= -2147483648.0) return (-2147483647 - 1);
    return (int)x;
}

/* CUDA pow(float,int) (vec3.cuh:211) is exponentiation by squaring:
 * for n = 5 the products are a*(a^2)^2. */
MORT_HD float mort_powi5f(float a) {
    float r = a;
    float a2 = a * a;
    float a4 = a2 * a2;
    return r * a4;
}

/* ---- sin / cos core on [-pi/4, pi/4], fp64 Taylor sums ---- */
MORT_HD double mort_ksin(double r) {
    const double z = r * r;
    double p = 1.0 / 1307674368000.0;            /* 1/15! */
    p = p * z - 1.0 / 6227020800.0;              /* 1/13! */
    p = p * z + 1.0 / 39916800.0;                /* 1/11! */
    p = p * z - 1.0 / 362880.0;                  /* 1/9!  */
    p = p * z + 1.0 / 5040.0;                    /* 1/7!  */
    p = p * z - 1.0 / 120.0;                     /* 1/5!  */
    p = p * z + 1.0 / 6.0;                       /* 1/3!  */
    return r - (r * z) * p;
}
MORT_HD double mort_kcos(double r) {
    const double z = r * r;
    double p = 1.0 / 20922789888000.0;           /* 1/16! */
    p = p * z - 1.0 / 87178291200.0;             /* 1/14! */
    p = p * z + 1.0 / 479001600.0;               /* 1/12! */
    p = p * z - 1.0 / 3628800.0;                 /* 1/10! */
    p = p * z + 1.0 / 40320.0;                   /* 1/8!  */
    p = p * z - 1.0 / 720.0;                     /* 1/6!  */
    p = p * z + 1.0 / 24.0;                      /* 1/4!  */
    return (1.0 - 0.5 * z) + (z * z) * p;
}

/* Reduce x to r in [-pi/4, pi/4] and the quadrant q = k mod 4.
 * pi/2 is split hi (33 bits, k*hi exact for |k| < 2^20) + lo. */
MORT_HD double mort_rem_pio2(double x, int *q) {
    const double two_over_pi = 0.6366197723675814;
    const double pio2_hi = 1.5707963267341256;
    const double pio2_lo = 6.077100506506192e-11;
    double kd = x * two_over_pi;
    /* round to nearest integer, half away from zero (exact for |kd| < 2^31) */
    int k = (int)(kd < 0.0 ? kd - 0.5 : kd + 0.5);
    double fk = (double)k;
    *q = k & 3;
    return (x - fk * pio2_hi) - fk * pio2_lo;
}

MORT_HD double mort_sin(double x) {
    if (!(mort_fabs(x) < 1.0e5)) return x - x; /* inf/NaN/out of domain -> NaN or 0 */
    int q;
    double r = mort_rem_pio2(x, &q);
    double s = (q & 1) ? mort_kcos(r) : mort_ksin(r);
    return (q & 2) ? -s : s;
}
MORT_HD double mort_cos(double x) {
    if (!(mort_fabs(x) < 1.0e5)) return x - x;
    int q;
    double r = mort_rem_pio2(x, &q);
    double c = (q & 1) ? mort_ksin(r) : mort_kcos(r);
    return ((q + 1) & 2) ? -c : c;
}
MORT_HD float mort_sinf(float x) { return (float)mort_sin((double)x); }
MORT_HD float mort_cosf(float x) { return (float)mort_cos((double)x); }
/* mort_sinf(x) and mort_cosf(x) at once: the same operations on the same values (one reduction, each kernel
 * polynomial once), hence the same bits; tests/test_math.py compares them */
MORT_HD void mort_sincosf(float xf, float *s_out, float *c_out) {
    const double x = (double)xf;
    if (!(mort_fabs(x) < 1.0e5)) { *s_out = (float)(x - x); *c_out = (float)(x - x); return; }
    int q;
    const double r = mort_rem_pio2(x, &q);
    const double ks = mort_ksin(r), kc = mort_kcos(r);
    const double s = (q & 1) ? kc : ks;
    const double c = (q & 1) ? ks : kc;
    *s_out = (float)((q & 2) ? -s : s);
    *c_out = (float)(((q + 1) & 2) ? -c : c);
}

/* ---- atan on fp64: table of 5 break points + 10-term odd series ---- */
MORT_HD double mort_atan_pos01(double x) { /* 0 <= x <= 1 */
    double c, ac;
    if (x < 0.125)      { c = 0.0;  ac = 0.0; }
    else if (x < 0.375) { c = 0.25; ac = 0.24497866312686414; }
    else if (x < 0.625) { c = 0.5;  ac = 0.4636476090008061; }
    else if (x < 0.875) { c = 0.75; ac = 0.6435011087932844; }
    else                { c = 1.0;  ac = 0.7853981633974483; }
    double t = (x - c) / (1.0 + x * c); /* |t| <= 0.125 */
    double z = t * t;
    double p = 1.0 / 19.0;
    p = p * z - 1.0 / 17.0;
    p = p * z + 1.0 / 15.0;
    p = p * z - 1.0 / 13.0;
    p = p * z + 1.0 / 11.0;
    p = p * z - 1.0 / 9.0;
    p = p * z + 1.0 / 7.0;
    p = p * z - 1.0 / 5.0;
    p = p * z + 1.0 / 3.0;
    return ac + (t - (t * z) * p);
}
MORT_HD double mort_atan(double x) {
    const double pio2 = 1.5707963267948966;
    if (x != x) return x;
    double a = mort_fabs(x);
    double r = (a <= 1.0) ? mort_atan_pos01(a) : (pio2 - mort_atan_pos01(1.0 / a));
    return (x < 0.0) ? -r : r;
}

/* atan2f with the usual quadrant rules; signed zeros as IEEE atan2. */
MORT_HD float mort_atan2f(float yf, float xf) {
    const double pi = 3.141592653589793;
    const double pio2 = 1.5707963267948966;
    double y = (double)yf, x = (double)xf;
    if (x != x || y != y) return (float)(x + y);
    const int ysign = (int)(mort_d2u(y) >> 63);
    const int xsign = (int)(mort_d2u(x) >> 63);
    double r;
    if (y == 0.0) {
        r = xsign ? pi : 0.0;
    } else if (x == 0.0) {
        r = pio2;
    } else {
        double ay = mort_fabs(y), ax = mort_fabs(x);
        if (ax >= ay) {
            double a = mort_atan_pos01(ay / ax); /* inf/inf -> NaN handled below */
            if (ax == ay) a = 0.7853981633974483;
            r = xsign ? (pi - a) : a;
        } else {
            double a = mort_atan_pos01(ax / ay);
            r = xsign ? (pio2 + a) : (pio2 - a);
        }
    }
    return (float)(ysign ? -r : r);
}

/* acosf(x) = 2*atan(sqrt((1-x)/(1+x))); |x| > 1 -> NaN, like libm. */
MORT_HD float mort_acosf(float xf) {
    double x = (double)xf;
    if (x != x) return xf;
    if (x > 1.0 || x < -1.0) return (float)((x - x) / (x - x));
    if (x == -1.0) return (float)3.141592653589793;
    double t = mort_sqrt((1.0 - x) / (1.0 + x));
    return (float)(2.0 * mort_atan(t));
}

/* logf on fp64: x = m*2^e, m in [sqrt(1/2), sqrt(2)); log m = 2 atanh(s). */
MORT_HD float mort_logf(float xf) {
    double x = (double)xf;
    if (x != x) return xf;
    if (x < 0.0) return (float)((x - x) / (x - x));
    if (x == 0.0) return (float)(-1.0 / (x * x)); /* -inf */
    uint64_t u = mort_d2u(x);
    if ((u >> 52) == 0x7ffu) return xf; /* +inf */
    int e = (int)((u >> 52) & 0x7ffu) - 1023; /* float inputs are normal in fp64 */
    double m = mort_u2d((u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL); /* [1,2) */
    if (m > 1.4142135623730951) { m = m * 0.5; e = e + 1; }
    double s = (m - 1.0) / (m + 1.0); /* |s| <= 0.1716 */
    double z = s * s;
    double p = 1.0 / 23.0;
    p = p * z + 1.0 / 21.0;
    p = p * z + 1.0 / 19.0;
    p = p * z + 1.0 / 17.0;
    p = p * z + 1.0 / 15.0;
    p = p * z + 1.0 / 13.0;
    p = p * z + 1.0 / 11.0;
    p = p * z + 1.0 / 9.0;
    p = p * z + 1.0 / 7.0;
    p = p * z + 1.0 / 5.0;
    p = p * z + 1.0 / 3.0;
    double lm = 2.0 * (s + (s * z) * p);
    return (float)((double)e * 0.6931471805599453 + lm);
}

#endif /* MORT_MATH_H */
