/*
 * oracle/det_math.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Deterministic exp / log / sincos built from IEEE-754 double +,-,*,/ and fma only, so that
 * the CPU oracle and the HIP kernels (which carry their own copy of the same recipe in
 * time_series_spark_amd/csrc/tsf_detmath.h) produce bit-identical results.  Needed because
 * Stan's L-BFGS on the Prophet posterior is chaotic: a 1-ulp difference in one exp() moves the
 * final forecast by ~1e-3 (measured, see DESIGN.md "Why canonical arithmetic").
 *
 * Accuracy targets (checked against libm in tests/test_oracle.py): exp, sin, cos <= 2 ulp on
 * the ranges used (|x| < 700 for exp, |x| < 1e5 for sincos); log <= 4 ulp.
 */
#ifndef ORACLE_DET_MATH_H
#define ORACLE_DET_MATH_H
#include <math.h>
#include <stdint.h>
#include <string.h>

static inline double det_pow2i(int n)      /* 2^n for -1022 <= n <= 1023 */
{
    uint64_t bits = (uint64_t)(n + 1023) << 52;
    double d;
    memcpy(&d, &bits, 8);
    return d;
}

static inline double det_exp(double x)
{
    if (x != x) return x;
    if (x > 709.782712893384) return INFINITY;
    if (x < -745.2) return 0.0;
    const double n = rint(x * 1.4426950408889634);
    double r = fma(-n, 6.93147180369123816490e-01, x);
    r = fma(-n, 1.90821492927058770002e-10, r);
    /* Taylor degree 13 on |r| <= 0.3466, Horner with fma */
    double p = 1.6059043836821613e-10;             /* 1/13! */
    p = fma(p, r, 2.08767569878681e-09);           /* 1/12! */
    p = fma(p, r, 2.505210838544172e-08);          /* 1/11! */
    p = fma(p, r, 2.755731922398589e-07);          /* 1/10! */
    p = fma(p, r, 2.7557319223985893e-06);         /* 1/9!  */
    p = fma(p, r, 2.48015873015873e-05);           /* 1/8!  */
    p = fma(p, r, 1.984126984126984e-04);          /* 1/7!  */
    p = fma(p, r, 1.388888888888889e-03);          /* 1/6!  */
    p = fma(p, r, 8.333333333333333e-03);          /* 1/5!  */
    p = fma(p, r, 4.1666666666666664e-02);         /* 1/4!  */
    p = fma(p, r, 1.6666666666666666e-01);         /* 1/3!  */
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    const int ni = (int)n;
    const int n1 = ni / 2, n2 = ni - n1;
    return (p * det_pow2i(n1)) * det_pow2i(n2);
}

static inline double det_log(double x)     /* x > 0, finite, normal */
{
    uint64_t bits;
    memcpy(&bits, &x, 8);
    int e = (int)((bits >> 52) & 0x7ff) - 1023;
    bits = (bits & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m;
    memcpy(&m, &bits, 8);                      /* m in [1,2) */
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s;
    double R = 1.479819860511658591e-01;
    R = fma(R, z, 1.531383769920937332e-01);
    R = fma(R, z, 1.818357216161805012e-01);
    R = fma(R, z, 2.222219843214978396e-01);
    R = fma(R, z, 2.857142874366239149e-01);
    R = fma(R, z, 3.999999999940941908e-01);
    R = fma(R, z, 6.666666666666735130e-01);
    R = R * z;
    /* log(1+f) = 2s + s*R ; result = e*ln2 + that */
    const double l1p = fma(s, R, 2.0 * s);
    const double de = (double)e;
    return fma(de, 6.93147180369123816490e-01, fma(de, 1.90821492927058770002e-10, l1p));
}

/* sin and cos of x, |x| < ~1e5 (2-term Cody-Waite reduction with fma). */
static inline void det_sincos(double x, double *s_out, double *c_out)
{
    const double n = rint(x * 6.36619772367581382433e-01);        /* x * 2/pi */
    double r = fma(-n, 1.5707963267948966, x);
    r = fma(-n, 6.123233995736766e-17, r);
    const double z = r * r;
    double ps = 1.58969099521155010221e-10;
    ps = fma(ps, z, -2.50507602534068634195e-08);
    ps = fma(ps, z, 2.75573137070700676789e-06);
    ps = fma(ps, z, -1.98412698298579493134e-04);
    ps = fma(ps, z, 8.33333333332248946124e-03);
    ps = fma(ps, z, -1.66666666666666324348e-01);
    const double sn = fma(r * z, ps, r);
    double pc = -1.13596475577881948265e-11;
    pc = fma(pc, z, 2.08757232129817482790e-09);
    pc = fma(pc, z, -2.75573143513906633035e-07);
    pc = fma(pc, z, 2.48015872894767294178e-05);
    pc = fma(pc, z, -1.38888888888741095749e-03);
    pc = fma(pc, z, 4.16666666666666019037e-02);
    const double cs = fma(z * z, pc, fma(-0.5, z, 1.0));
    const long long q = (long long)n;
    switch ((int)(q & 3)) {
    case 0: *s_out = sn;  *c_out = cs;  break;
    case 1: *s_out = cs;  *c_out = -sn; break;
    case 2: *s_out = -sn; *c_out = -cs; break;
    default: *s_out = -cs; *c_out = sn; break;
    }
}

#endif
