// pxf_libm_match.h -- cos(atan2(r, z)) and tan(beta/2) with the x86-64 C library's OWN rounding, for the one
// place on the path where a single ulp of a libm result is amplified beyond the parity tolerance: the
// `(1 - cos beta)` of the Wolter-Schwarzschild secondary (woltsurf.f95:518-519, 810-811) quantises its F in
// steps of 2.6e-9 mm, so the converged point equals the reference's only if the whole Newton path does
// (DESIGN.md section 4).
//
// The reference's `sin(beta)`, `cos(beta)` with `beta = atan2(sqrt(x**2+y**2), z)` compile (flang and gcc alike) to ONE
// call of glibc's `sincos`, `tan(beta/2)` to `tan`.  Both are PUBLISHED algorithms (IBM Accurate Mathematical Library):
// the cosine splits x = k/128 + t by adding 1.5*2^45, takes sin/cos(k/128) from a head+tail table, two short Taylor
// polynomials in t and one compensated combination; it is NOT correctly rounded (~5e-4 of its results are the other
// neighbour), and the `sincos` build (plain multiplies and adds) differs again from the `cos()` build (FMA) in 2.5e-5
// of the arguments of the W-S range -- round 2 first restated `cos()` and sat at 2.1e-11 mm RMS with 1 ray in 1e4
// off, then the `sincos` build: 1.0e-12, none off.  They are restated here operation by operation with the table
// regenerated from 113-bit arithmetic (pxf_trig_tables.h); tests/test_host.py holds them to libm bit for bit over the
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

__host__ __device__ __forceinline__ double libm_sc_cos(double x, double dx, const double *__restrict__ tab);

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
    // the reference binary takes sin(beta), cos(beta) from ONE `sincos` call (woltsurf.f95:516-520 compiled): the
    // non-FMA build of the cosine, which differs from `cos()` in 2.5e-5 of the arguments of this range
    cb = libm_sc_cos(beta, 0., sincos_tab);
    return true;
}

// ---- sincos(atan2(y, x)) at an arbitrary azimuth -------------------------------------------------------------
// The back-surface variants (woltsurf.f95:760-763, 862-865) shrink the ray's radius by the substrate thickness
// through `theta = atan2(y, x); x' = (r - thick) cos(theta); y' = (r - thick) sin(theta)` -- compiled (flang and
// gcc alike) into ONE call of the C library's `sincos`, which in glibc 2.35 is the same IBM algorithm as sin / cos
// but built WITHOUT fused multiply-adds (no multiarch variant), so it is restated here with plain operations
// (the translation units that include this header are compiled -ffp-contract=off): Taylor polynomial below 0.126,
// table + polynomials below 0.855, reflection about pi/2 up to 2.426, Cody-Waite reduction by pi/2 beyond.
// tests/test_host.py holds it to libm's sincos bit for bit over (-pi, pi].
__host__ __device__ __forceinline__ double libm_sc_cos(double x, double dx, const double *__restrict__ tab)
{
    const double big = 0x1.8p45;
    const double sn3 = -1.66666666666664880952546298448555E-01, sn5 = 8.33333214285722277379541354343671E-03;
    const double cs2 = 4.99999999999999999999950396842453E-01, cs4 = -4.16666666666664434524222570944589E-02,
                 cs6 = 1.38888874007937613028114285595617E-03;
    if (x < 0) dx = -dx;
    const double u = big + fabs(x);
    uint64_t ub;
    memcpy(&ub, &u, 8);
    const int k = (int)(uint32_t)(ub & 0xffffffffu) * 4;
    x = fabs(x) - (u - big) + dx;
    const double sn = tab[k], ssn = tab[k + 1], cs = tab[k + 2], ccs = tab[k + 3];
    const double xx = x * x;
    const double s = x + x * xx * (sn3 + xx * sn5);
    const double c = xx * (cs2 + xx * (cs4 + xx * cs6));
    const double cor = (ccs - s * ssn - cs * c) - sn * s;
    return cs + cor;
}
__host__ __device__ __forceinline__ double libm_sc_sin(double x, double dx, const double *__restrict__ tab)
{
    const double big = 0x1.8p45;
    const double sn3 = -1.66666666666664880952546298448555E-01, sn5 = 8.33333214285722277379541354343671E-03;
    const double cs2 = 4.99999999999999999999950396842453E-01, cs4 = -4.16666666666664434524222570944589E-02,
                 cs6 = 1.38888874007937613028114285595617E-03;
    const double s1 = -0x1.5555555555555p-3, s2 = 0x1.1111111110ECEp-7, s3 = -0x1.A01A019DB08B8p-13,
                 s4 = 0x1.71DE27B9A7ED9p-19, s5 = -0x1.ADDFFC2FCDF59p-26;
    const double xold = x;
    if (fabs(x) < 0.126) {
        const double xx = x * x;
        const double t = (((((s5 * xx + s4) * xx + s3) * xx + s2) * xx + s1) * x - 0.5 * dx) * xx + dx;
        return x + t;
    }
    if (x <= 0) dx = -dx;
    const double u = big + fabs(x);
    uint64_t ub;
    memcpy(&ub, &u, 8);
    const int k = (int)(uint32_t)(ub & 0xffffffffu) * 4;
    x = fabs(x) - (u - big);
    const double sn = tab[k], ssn = tab[k + 1], cs = tab[k + 2], ccs = tab[k + 3];
    const double xx = x * x;
    const double s = x + (dx + x * xx * (sn3 + xx * sn5));
    const double c = x * dx + xx * (cs2 + xx * (cs4 + xx * cs6));
    const double cor = (ssn + s * ccs - sn * c) + cs * s;
    return copysign(sn + cor, xold);
}
// glibc s_sincos.c for 2^-27 <= |x| < 105414350 (here: |x| <= pi)
__host__ __device__ __forceinline__ void libm_sincos(double x, const double *__restrict__ tab, double &sinx, double &cosx)
{
    const double hp0 = 0x1.921FB54442D18p0, hp1 = 0x1.1A62633145C07p-54;
    const double mp1 = 0x1.921FB58000000p0, mp2 = -0x1.DDE973C000000p-27, pp3 = -0x1.CB3B398000000p-55,
                 pp4 = -0x1.d747f23e32ed7p-83, hpinv = 0x1.45F306DC9C883p-1, toint = 0x1.8p52;
    const double ax = fabs(x);
    if (ax < 0x1p-27) {
        sinx = x;
        cosx = 1.;
    } else if (ax < 0.85546875) {
        sinx = libm_sc_sin(x, 0., tab);
        cosx = libm_sc_cos(x, 0., tab);
    } else if (ax < 0x1.368fdp+1) {   // high word < 0x400368fd: 2.426265
        const double y = hp0 - ax, a = y + hp1, da = (y - a) + hp1;
        sinx = copysign(libm_sc_cos(a, da, tab), x);
        cosx = libm_sc_sin(a, da, tab);
    } else {
        const double t = x * hpinv + toint, xn = t - toint;
        uint64_t tb;
        memcpy(&tb, &t, 8);
        int n = (int)(tb & 3);
        const double y = (x - xn * mp1) - xn * mp2;
        double t1 = xn * pp3;
        const double t2 = y - t1;
        double db = (y - t2) - t1;
        t1 = xn * pp4;
        const double b = t2 - t1;
        db += (t2 - b) - t1;
        const double r0 = (n & 1) ? libm_sc_cos(b, db, tab) : libm_sc_sin(b, db, tab);
        sinx = (n & 2) ? -r0 : r0;
        n++;
        const double r1 = (n & 1) ? libm_sc_cos(b, db, tab) : libm_sc_sin(b, db, tab);
        cosx = (n & 2) ? -r1 : r1;
    }
}

// theta = atan2(y, x), correctly rounded (double-double: octant reduction, reduction against tan(k/128), short
// series; pi/2 and pi as head + tail) -- what glibc's atan2 returns in all but ~2e-3 of the cases -- then
// cos(theta), sin(theta) by libm_sincos.  false: a zero / non-finite / extreme operand (the caller uses ocml).
__host__ __device__ __forceinline__ bool libm_sincos_atan2(double y, double x, const double *__restrict__ sincos_tab,
                                                           const double *__restrict__ tan_tab, double &theta,
                                                           double &sinth, double &costh)
{
    const double ax = fabs(x), ay = fabs(y);
    if (!(ax > 1e-150 && ay > 1e-150 && ax < 1e150 && ay < 1e150)) return false;
    const bool swap = ay > ax;
    const double num = swap ? ax : ay, den = swap ? ay : ax;    // t = num / den in (0, 1]
    const double th = num / den;
    if (!(th > 1e-100)) return false;
    const double tl = fma(-th, den, num) / den;
    int k0 = (int)rintf(atanf((float)th) * 128.f);
    k0 = k0 > PXF_TRIG_TAB_N - 1 ? PXF_TRIG_TAB_N - 1 : k0;
    const double xi = k0 * 0.0078125, Th = tan_tab[2 * k0], Tl = tan_tab[2 * k0 + 1];
    const double s = th - Th, bb = s - th;
    const double e = ((th - (s - bb)) + (-Th - bb)) + (tl - Tl);
    const double ph = th * Th, pl = fma(th, Th, -ph) + (th * Tl + tl * Th);
    const double dh = 1. + ph, dl = ((1. - dh) + ph) + pl;
    const double wh = s / dh;
    const double wl = (fma(-wh, dh, s) + e - wh * dl) / dh;
    const double w2 = wh * wh;
    const double p = wh * w2 * (-1. / 3 + w2 * (0.2 + w2 * (-1. / 7 + w2 * (1. / 9 - w2 * (1. / 11)))));
    // atan(t) = rh + rl
    double rh = xi + wh;
    const double b2 = rh - xi;
    double rl = ((xi - (rh - b2)) + (wh - b2)) + (wl + p);
    const double hp0 = 0x1.921FB54442D18p0, hp1 = 0x1.1A62633145C07p-54;
    if (swap) {            // pi/2 - r
        const double q = hp0 - rh, c2 = q - hp0;
        rl = ((hp0 - (q - c2)) + (-rh - c2)) + (hp1 - rl);
        rh = q;
    }
    if (x < 0) {           // pi - r
        const double q = 2 * hp0 - rh, c2 = q - 2 * hp0;
        rl = ((2 * hp0 - (q - c2)) + (-rh - c2)) + (2 * hp1 - rl);
        rh = q;
    }
    const double r = rh + rl;
    theta = y < 0 ? -r : r;
    libm_sincos(theta, sincos_tab, sinth, costh);
    return true;
}

}  // namespace pxf
