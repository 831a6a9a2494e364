// tsf_detmath.h -- deterministic exp / log / sincos for the gfx950 kernels.
//
// The canonical arithmetic of this library (DESIGN.md "Why canonical arithmetic") needs
// transcendental functions whose result depends only on IEEE-754 double +,-,*,/ and fma, so
// that the device path is reproducible to the bit (Stan's L-BFGS on the Prophet posterior
// amplifies a 1-ulp difference to ~1e-3 in the forecast).  ocml's exp/log/sin/cos are not
// specified to that level, so the kernels use these fixed recipes instead:
//   exp    : n = rint(x*log2e); r = x - n*ln2 (2-term, fma); Taylor degree 13 in Horner/fma
//            form; scale by 2^(n/2) * 2^(n - n/2).
//   log    : x = m*2^e, m in [sqrt(1/2), sqrt(2)); s = f/(2+f), f = m-1; 7-term odd series
//            (classic fdlibm-style coefficients); e*ln2 added in two fma steps.
//   sincos : n = rint(x*2/pi); 2-term Cody-Waite reduction with fma; degree-13/14 kernels;
//            quadrant select.  Valid for |x| < ~1e5 (Fourier arguments here are < 1e4).
// Everything compiles with -ffp-contract=off: the only fused operations are the explicit
// __builtin_fma calls.
#pragma once
#include <hip/hip_runtime.h>

namespace tsf {

__device__ __forceinline__ double dm_pow2i(int n)
{
    return __longlong_as_double((long long)(n + 1023) << 52);
}

__device__ __forceinline__ double dm_exp(double x)
{
    if (x != x) return x;
    if (x > 709.782712893384) return __builtin_huge_val();
    if (x < -745.2) return 0.0;
    const double n = __builtin_rint(x * 1.4426950408889634);
    double r = __builtin_fma(-n, 6.93147180369123816490e-01, x);
    r = __builtin_fma(-n, 1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;
    p = __builtin_fma(p, r, 2.08767569878681e-09);
    p = __builtin_fma(p, r, 2.505210838544172e-08);
    p = __builtin_fma(p, r, 2.755731922398589e-07);
    p = __builtin_fma(p, r, 2.7557319223985893e-06);
    p = __builtin_fma(p, r, 2.48015873015873e-05);
    p = __builtin_fma(p, r, 1.984126984126984e-04);
    p = __builtin_fma(p, r, 1.388888888888889e-03);
    p = __builtin_fma(p, r, 8.333333333333333e-03);
    p = __builtin_fma(p, r, 4.1666666666666664e-02);
    p = __builtin_fma(p, r, 1.6666666666666666e-01);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    const int ni = (int)n;
    const int n1 = ni / 2, n2 = ni - n1;
    return (p * dm_pow2i(n1)) * dm_pow2i(n2);
}

// dm_exp without branches (the three special cases are selects on the finished result): the
// same value for every input, and a straight-line chain the scheduler can place under other work
__device__ __forceinline__ double dm_exp_sel(double x)
{
    const double n = __builtin_rint(x * 1.4426950408889634);
    double r = __builtin_fma(-n, 6.93147180369123816490e-01, x);
    r = __builtin_fma(-n, 1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;
    p = __builtin_fma(p, r, 2.08767569878681e-09);
    p = __builtin_fma(p, r, 2.505210838544172e-08);
    p = __builtin_fma(p, r, 2.755731922398589e-07);
    p = __builtin_fma(p, r, 2.7557319223985893e-06);
    p = __builtin_fma(p, r, 2.48015873015873e-05);
    p = __builtin_fma(p, r, 1.984126984126984e-04);
    p = __builtin_fma(p, r, 1.388888888888889e-03);
    p = __builtin_fma(p, r, 8.333333333333333e-03);
    p = __builtin_fma(p, r, 4.1666666666666664e-02);
    p = __builtin_fma(p, r, 1.6666666666666666e-01);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    const bool in_range = (x <= 709.782712893384) && (x >= -745.2);
    const int ni = in_range ? (int)n : 0;
    const int n1 = ni / 2, n2 = ni - n1;
    double e = (p * dm_pow2i(n1)) * dm_pow2i(n2);
    if (x > 709.782712893384) e = __builtin_huge_val();
    if (x < -745.2) e = 0.0;
    if (x != x) e = x;
    return e;
}

// fp64 arithmetic with a LITERAL operand held in a scalar register pair (a VOP3 instruction of gfx950 cannot encode a 64-bit
// literal; left to itself the compiler materialises each one in a VGPR pair, two v_mov_b32).  Same instruction, same
// rounding: only the operand's register file differs.  Used where vector issue is what a wave waits for and registers are
// not (the cooperative kernel's row waves); the quadratic-form kernels measured it and do not (tsf_quad_kernels.h).
__device__ __forceinline__ double fma_vvs(double a, double b, double c_scalar)
{
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c_scalar));
    return r;
}
__device__ __forceinline__ double fma_vsv(double a, double b_scalar, double c)
{
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b_scalar), "v"(c));
    return r;
}
__device__ __forceinline__ double mul_vs(double a, double b_scalar)
{
    double r;
    asm("v_mul_f64 %0, %1, %2" : "=v"(r) : "v"(a), "s"(b_scalar));
    return r;
}

// dm_exp_sel (tsf_detmath.h) with its literals as scalar operands: the same operations on the same values
__device__ __forceinline__ double dm_exp_sel_sc(double x)
{
    const double n = __builtin_rint(mul_vs(x, 1.4426950408889634));
    double r = fma_vsv(-n, 6.93147180369123816490e-01, x);
    r = fma_vsv(-n, 1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;
    p = fma_vvs(p, r, 2.08767569878681e-09);
    p = fma_vvs(p, r, 2.505210838544172e-08);
    p = fma_vvs(p, r, 2.755731922398589e-07);
    p = fma_vvs(p, r, 2.7557319223985893e-06);
    p = fma_vvs(p, r, 2.48015873015873e-05);
    p = fma_vvs(p, r, 1.984126984126984e-04);
    p = fma_vvs(p, r, 1.388888888888889e-03);
    p = fma_vvs(p, r, 8.333333333333333e-03);
    p = fma_vvs(p, r, 4.1666666666666664e-02);
    p = fma_vvs(p, r, 1.6666666666666666e-01);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    const bool in_range = (x <= 709.782712893384) && (x >= -745.2);
    const int ni = in_range ? (int)n : 0;
    const int n1 = ni / 2, n2 = ni - n1;
    double e = (p * dm_pow2i(n1)) * dm_pow2i(n2);
    if (x > 709.782712893384) e = __builtin_huge_val();
    if (x < -745.2) e = 0.0;
    if (x != x) e = x;
    return e;
}


__device__ __forceinline__ double dm_log(double x)
{
    unsigned long long bits = (unsigned long long)__double_as_longlong(x);
    int e = (int)((bits >> 52) & 0x7ff) - 1023;
    bits = (bits & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m = __longlong_as_double((long long)bits);
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s;
    double R = 1.479819860511658591e-01;
    R = __builtin_fma(R, z, 1.531383769920937332e-01);
    R = __builtin_fma(R, z, 1.818357216161805012e-01);
    R = __builtin_fma(R, z, 2.222219843214978396e-01);
    R = __builtin_fma(R, z, 2.857142874366239149e-01);
    R = __builtin_fma(R, z, 3.999999999940941908e-01);
    R = __builtin_fma(R, z, 6.666666666666735130e-01);
    R = R * z;
    const double l1p = __builtin_fma(s, R, 2.0 * s);
    const double de = (double)e;
    return __builtin_fma(de, 6.93147180369123816490e-01,
                         __builtin_fma(de, 1.90821492927058770002e-10, l1p));
}

__device__ __forceinline__ void dm_sincos(double x, double &s_out, double &c_out)
{
    const double n = __builtin_rint(x * 6.36619772367581382433e-01);
    double r = __builtin_fma(-n, 1.5707963267948966, x);
    r = __builtin_fma(-n, 6.123233995736766e-17, r);
    const double z = r * r;
    double ps = 1.58969099521155010221e-10;
    ps = __builtin_fma(ps, z, -2.50507602534068634195e-08);
    ps = __builtin_fma(ps, z, 2.75573137070700676789e-06);
    ps = __builtin_fma(ps, z, -1.98412698298579493134e-04);
    ps = __builtin_fma(ps, z, 8.33333333332248946124e-03);
    ps = __builtin_fma(ps, z, -1.66666666666666324348e-01);
    const double sn = __builtin_fma(r * z, ps, r);
    double pc = -1.13596475577881948265e-11;
    pc = __builtin_fma(pc, z, 2.08757232129817482790e-09);
    pc = __builtin_fma(pc, z, -2.75573143513906633035e-07);
    pc = __builtin_fma(pc, z, 2.48015872894767294178e-05);
    pc = __builtin_fma(pc, z, -1.38888888888741095749e-03);
    pc = __builtin_fma(pc, z, 4.16666666666666019037e-02);
    const double cs = __builtin_fma(z * z, pc, __builtin_fma(-0.5, z, 1.0));
    const long long q = (long long)n;
    switch ((int)(q & 3)) {
    case 0: s_out = sn;  c_out = cs;  break;
    case 1: s_out = cs;  c_out = -sn; break;
    case 2: s_out = -sn; c_out = -cs; break;
    default: s_out = -cs; c_out = sn; break;
    }
}

}  // namespace tsf
