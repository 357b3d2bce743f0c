// pxf_fastmath.h -- fp64 reciprocal / rsqrt based divides and square roots for the TOLERANCE-parity code
// paths (pxf_program_fast.hip, the Zernike kernels).  Never included by a bit-exact Op: those use IEEE
// `/` and `sqrt` so that their results equal the x86 reference bit for bit.
//
// v_rcp_f64 / v_rsq_f64 are single-precision-grade seeds; two Newton-Raphson steps (quadratic) bring them
// to ~1 ulp, and f_div adds the residual correction LLVM itself emits for an `afn` fp64 divide.  About
// 8 VALU instructions (one of them quarter rate) instead of the ~14 of the IEEE expansion
// (div_scale x2, rcp, 5 fma, mul, div_fmas, div_fixup), and reciprocals can be shared.
#pragma once
#include <hip/hip_runtime.h>

namespace pxf {

__device__ __forceinline__ double f_rcp(double a)
{
    double r = __builtin_amdgcn_rcp(a);
    double e = fma(-a, r, 1.);
    r = fma(e, r, r);
    e = fma(-a, r, 1.);
    return fma(e, r, r);
}
__device__ __forceinline__ double f_div(double x, double y)
{
    const double r = f_rcp(y);
    const double q = x * r;
    return fma(fma(-y, q, x), r, q);
}
// f_div with IEEE results for the operands the refinement cannot handle (zero / huge / tiny / non-finite
// divisors: x/0 = inf, 0/0 = NaN ...): the in-band failure values of the reference must survive.
__device__ __forceinline__ double f_div_ieee_edges(double x, double y)
{
    double q = f_div(x, y);
    if (__builtin_expect(!(fabs(y) > 1e-290 && fabs(y) < 1e290 && fabs(x) < 1e290), 0)) q = x / y;
    return q;
}
__device__ __forceinline__ double f_rsq(double a)  // 1/sqrt(a), a > 0 finite
{
    double y = __builtin_amdgcn_rsq(a);
    double e = fma(-a * y, y, 1.);
    y = fma(.5 * y, e, y);
    e = fma(-a * y, y, 1.);
    return fma(.5 * y, e, y);
}
__device__ __forceinline__ double f_sqrt(double a)  // sqrt(a); 0 -> 0, negative -> NaN
{
    const double s = a * f_rsq(a);
    return a == 0. ? 0. : s;
}

}  // namespace pxf
