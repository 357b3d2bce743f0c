// pxf_libm_match.h -- cos(atan2(r, z)) and tan(beta/2) with the x86-64 C library's OWN rounding, for the one
// place on the path where a single ulp of a libm result is amplified beyond the parity tolerance: the
// `(1 - cos beta)` of the Wolter-Schwarzschild secondary (woltsurf.f95:518-519, 810-811) quantises its F in
// steps of 2.6e-9 mm, so the converged point equals the reference's only if the whole Newton path does
// (DESIGN.md section 4).
//
// The reference calls `cos(beta)`, `beta = atan2(sqrt(x**2+y**2), z)`, through glibc (2.35 in this image, FMA
// build selected at load time on every AVX2 host).  glibc's cosine for 2^-27 < |x| < 0.855 is a PUBLISHED
// algorithm (IBM Accurate Mathematical Library, "do_cos": x = xi + t with xi = k/128 from adding 1.5*2^45,
// sin/cos(xi) from a head+tail table, short Taylor polynomials in t, one compensated combination) and is NOT
// correctly rounded: ~5e-4 of its results are the other neighbour.  It is restated here operation by
// operation -- including which products the library's FMA build fuses -- with the table regenerated from
// 113-bit arithmetic (pxf_trig_tables.h); tests/test_host.py holds it to libm's `cos` bit for bit over the
// whole range on the machine the tests run on.  `beta` itself is formed as the correctly rounded arctangent
// (double-double reduction against tan(k/128)), which is what glibc's atan2 returns in 99.8 % of the cases.
//
// Host + device: the same source runs inside WSOp (device tables in constant memory) and behind the host
// utility pxf_host_cos_atan2 (CPU test).  Explicit fma() calls fuse regardless of -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "pxf_trig_tables.h"

namespace pxf {

// glibc s_sin.c: do_cos(x, 0) for 2^-27 < |x| < 0.85546875 (the caller guarantees the range).
__host__ __device__ __forceinline__ double libm_cos_small(double x, const double *__restrict__ sincos_tab)
{
    const double big = 0x1.8p45;
    const double sn3 = -1.66666666666664880952546298448555E-01, sn5 = 8.33333214285722277379541354343671E-03;
    const double cs2 = 4.99999999999999999999950396842453E-01, cs4 = -4.16666666666664434524222570944589E-02,
                 cs6 = 1.38888874007937613028114285595617E-03;
    const double ax = fabs(x);
    const double u = big + ax;          // rounds |x| to a multiple of 1/128; the multiple sits in the low word
    uint64_t ub;
    memcpy(&ub, &u, 8);
    const int k = (int)(uint32_t)(ub & 0xffffffffu) * 4;
    const double t = ax - (u - big);    // exact
    const double sn = sincos_tab[k], ssn = sincos_tab[k + 1], cs = sincos_tab[k + 2], ccs = sincos_tab[k + 3];
    const double tt = t * t;
    const double s = fma(t * tt, fma(tt, sn5, sn3), t);
    const double c = tt * fma(tt, fma(tt, cs6, cs4), cs2);
    const double cor = fma(-sn, s, fma(-cs, c, fma(-s, ssn, ccs)));
    return cs + cor;
}

// glibc s_tan.c, the polynomial branch 1.259e-8 < |x| <= 0.0608 (FMA build): x + x^3 (d3 + x^2 (d5 + ...)).
// The W-S secondary's kterm = tan(beta/2)**2 / tan(betas/2)**2 - 1 vanishes at the edge beta = betas of the
// mirror, where rays that start above the mirror land after their first Newton steps (woltsurf.f95:511-523):
// there ONE ulp of tan(beta/2) moves the next step by ~4e-9 mm.  The caller guarantees 2e-8 < x < 0.0607.
__host__ __device__ __forceinline__ double libm_tan_small(double x)
{
    const double d3 = 0x1.5555555555555p-2, d5 = 0x1.11111111107c6p-3, d7 = 0x1.ba1ba1cdb8745p-5,
                 d9 = 0x1.664ed49cfc666p-6, d11 = 0x1.2385a3cf2e4eap-7;
    const double x2 = x * x;
    double t2 = fma(x2, d11, d9);
    t2 = fma(x2, t2, d7);
    t2 = fma(x2, t2, d5);
    t2 = fma(x2, t2, d3);
    return fma(t2, x * x2, x);
}

// beta = atan2(r, z) and cos(beta) as the reference's libm returns them, for r >= 0, z > 0 with
// r/z < tan(0.855) (beta inside do_cos' range).  Returns false outside that range (caller falls back).
__host__ __device__ __forceinline__ bool libm_cos_atan2(double r, double z, const double *__restrict__ sincos_tab,
                                                        const double *__restrict__ tan_tab, double &beta, double &cb)
{
    const double th = r / z;
    if (!(z > 0. && th >= 0x1p-26 && th < 1.15 && z < 1e150 && r > 1e-150)) return false;
    const double tl = fma(-th, z, r) / z;                       // r/z = th + tl
    int k0 = (int)rintf(atanf((float)th) * 128.f);
    k0 = k0 > PXF_TRIG_TAB_N - 1 ? PXF_TRIG_TAB_N - 1 : k0;
    const double xi = k0 * 0.0078125, Th = tan_tab[2 * k0], Tl = tan_tab[2 * k0 + 1];
    // w = (t - T) / (1 + t T) in double-double, |w| <~ 1/200
    const double s = th - Th, bb = s - th;
    const double e = ((th - (s - bb)) + (-Th - bb)) + (tl - Tl);
    const double ph = th * Th, pl = fma(th, Th, -ph) + (th * Tl + tl * Th);
    const double dh = 1. + ph, dl = ((1. - dh) + ph) + pl;
    const double wh = s / dh;
    const double wl = (fma(-wh, dh, s) + e - wh * dl) / dh;
    const double w2 = wh * wh;
    const double p = wh * w2 * (-1. / 3 + w2 * (0.2 + w2 * (-1. / 7 + w2 * (1. / 9 - w2 * (1. / 11)))));
    const double lo = wl + p;
    const double s1 = xi + wh, b2 = s1 - xi;
    const double e1 = (xi - (s1 - b2)) + (wh - b2);
    beta = s1 + (e1 + lo);
    if (!(beta < 0.85546875)) return false;
    cb = libm_cos_small(beta, sincos_tab);
    return true;
}

}  // namespace pxf
