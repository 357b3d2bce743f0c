// pxf_exactdiv.h -- IEEE-754 fp64 quotients that share their reciprocal, bit for bit equal to `/`.
//
// Why.  The bit-exact kernels may not replace a division by a multiplication with a reciprocal: the reference
// divides, and n * (1/d) is not the correctly rounded n/d.  But what the compiler emits for `n / d` on gfx950 IS
// a reciprocal-based sequence, one that is arranged to deliver the correctly rounded quotient:
//       ds = v_div_scale(d)   ns = v_div_scale(n)          operands pre-scaled if the exponents are extreme
//       r  = v_rcp_f64(ds)    r = fma(r, fma(-ds, r, 1), r)   twice      reciprocal to < 1 ulp
//       q  = ns * r           e = fma(-ds, q, ns)          q = v_div_fmas(e, r, q)    one correction step (+ unscale)
//       q  = v_div_fixup(q, d, n)                          zeros, infinities, NaNs, overflow / underflow by hand
// eleven instructions, one of them (v_rcp_f64) quarter-rate: 56 issue cycles (tools/dev/instr_rate.hip).  With both
// operands ORDINARY numbers -- normal, exponents well inside the range, |exponent difference| < 768 -- v_div_scale
// returns its operand unchanged, v_div_fmas is a plain fma and v_div_fixup returns its first operand: the sequence
// is then rcp + 4 fma for the reciprocal of d (which does not depend on n) and mul + 2 fma per numerator.  Quotients
// that share a denominator -- the three components of a surface normal over its length (woltsurf.f95:48-50 and the
// like), y/rho and x/rho of a grating (transformationsf.f95:218) -- need the reciprocal ONCE:
//       three quotients: 3 x 56 = 168 issue cycles  ->  32 + 3 x 12 + 20 (the window test) = 88
// and every bit is what `/` gives, because inside the window it is the same sequence of operations on the same
// operands.  Outside the window (zeros, denormals, infinities, NaNs, huge or tiny magnitudes) the code falls back to
// `/` itself.  The window: biased exponent of every operand in [1023-300, 1023+300] (2^-300 <= |v| < 2^301), so
// exponent differences stay below 601 < 768, no operand is denormal or within 53 binades of it, no reciprocal or
// quotient leaves the normal range: none of v_div_scale's conditions can hold (CDNA3 ISA, V_DIV_SCALE_F64), and
// v_div_fixup's special cases need a zero, an infinity, a NaN or an exponent difference beyond 1024.
// Pinned by pxf_selftest_exact_division (every path against `/` on 2^31 random operand tuples, exponents drawn across
// and around the window) in the GPU suite, and by the parity tests of every Op that uses it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pxf {

constexpr unsigned kDivExpLo = (1023u - 300u) << 20, kDivExpHi = (1023u + 300u) << 20;

__device__ __forceinline__ unsigned div_exp_field(double v)
{
    return (unsigned)((unsigned long long)__double_as_longlong(v) >> 32) & 0x7ff00000u;
}
__device__ __forceinline__ unsigned umin3(unsigned a, unsigned b, unsigned c) { return min(min(a, b), c); }
__device__ __forceinline__ unsigned umax3(unsigned a, unsigned b, unsigned c) { return max(max(a, b), c); }

// the reciprocal the division sequence forms for an in-window d
__device__ __forceinline__ double div_refined_rcp(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    r = __builtin_fma(r, __builtin_fma(-d, r, 1.), r);
    r = __builtin_fma(r, __builtin_fma(-d, r, 1.), r);
    return r;
}
// ... and the quotient from it (n, d in the window, r = div_refined_rcp(d)):  == n / d
__device__ __forceinline__ double div_with_rcp(double n, double d, double r)
{
    const double q = n * r;
    return __builtin_fma(__builtin_fma(-d, q, n), r, q);
}

// q_i = n_i / d, i = 0..2  (bit for bit)
__device__ __forceinline__ void div3_exact(double n0, double n1, double n2, double d, double &q0, double &q1, double &q2)
{
    const unsigned e0 = div_exp_field(n0), e1 = div_exp_field(n1), e2 = div_exp_field(n2), ed = div_exp_field(d);
    if (umin3(umin3(e0, e1, e2), ed, ed) >= kDivExpLo && umax3(umax3(e0, e1, e2), ed, ed) <= kDivExpHi) {
        const double r = div_refined_rcp(d);
        q0 = div_with_rcp(n0, d, r);
        q1 = div_with_rcp(n1, d, r);
        q2 = div_with_rcp(n2, d, r);
    } else {
        q0 = n0 / d;
        q1 = n1 / d;
        q2 = n2 / d;
    }
}

// q_i = n_i / d, i = 0..1
__device__ __forceinline__ void div2_exact(double n0, double n1, double d, double &q0, double &q1)
{
    const unsigned e0 = div_exp_field(n0), e1 = div_exp_field(n1), ed = div_exp_field(d);
    if (umin3(e0, e1, ed) >= kDivExpLo && umax3(e0, e1, ed) <= kDivExpHi) {
        const double r = div_refined_rcp(d);
        q0 = div_with_rcp(n0, d, r);
        q1 = div_with_rcp(n1, d, r);
    } else {
        q0 = n0 / d;
        q1 = n1 / d;
    }
}

// v / |v|  (transformationsf.f95:216 `n/abs(n)`): +-1 for every finite non-zero v -- the division of a number by its
// own magnitude is exact -- and NaN for a zero, an infinity or a NaN, as `/` gives.
__device__ __forceinline__ double unit_sign_exact(double v)
{
    // -normal | -subnormal | +subnormal | +normal
    if (__builtin_amdgcn_class(v, 0x008 | 0x010 | 0x080 | 0x100)) return __builtin_copysign(1., v);
    return v / __builtin_fabs(v);
}

}  // namespace pxf
