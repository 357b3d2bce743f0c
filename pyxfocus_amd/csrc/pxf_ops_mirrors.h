// pxf_ops_mirrors.h -- the tolerance-parity mirror Ops of SURVEY 8f row f-2: woltersine, the Legendre-Legendre
// deformed mirrors (wolterprimLL, woltersecLL, ellipsoidWoltLL) and the Wolter-Schwarzschild primary /
// secondary (+ back surfaces).  Reference: woltsurf.f95:167-215, 220-379, 643-718, 387-588, 726-933 and
// legendre/legendrep, specialFunctions.f95:337-388.  Shared by the stand-alone entries (pxf_wolter2.hip) and
// the fused programs that hold such a mirror (pxf_program_mirror.hip) -- the SAME device code in both.
//
// These surfaces need per-ray transcendentals (atan2, sin/cos, asin, tan, pow), which come from
// ocml here and glibc in the reference: parity is to a stated tolerance (tests/util.py), not
// bitwise; the Newton iterations contract the differences.  Structure follows the reference's
// loops exactly, including tolerances (1.e-10 / 1.e-7 / 1.e-8 as fp32 literals), the
// stale-gradient normals, and the W-S 26-iteration cap that restores the start position.
//
// Legendre-Legendre: the reference evaluates every P_n by an explicit factorial sum, twice per
// coefficient per Newton step.  Here the (coeff, axial, az) list is folded on the host into a
// dense matrix C[axial][az], passed by value in the kernel arguments; P_n and P'_n come from the
// three-term recurrences, and per-iteration reciprocals are shared, so a Newton step
// costs 6 IEEE divides instead of ~30 (parity for these surfaces is to 1e-10, see tests/util.py).
#pragma once
#include <math.h>
#include <stdlib.h>

#include "pxf_kernel.h"
#include "pxf_libm_match.h"

namespace pxf {

// sin/cos/tan(k/128) head+tail tables of pxf_libm_match.h: constant memory for the kernels (one copy per
// translation unit that includes this header), a host copy for the pxf_host_cos_atan2 utility
static __constant__ double c_sincos_tab[4 * PXF_TRIG_TAB_N] = PXF_SINCOS_TAB_INIT;
static __constant__ double c_tan_tab[2 * PXF_TRIG_TAB_N] = PXF_TAN_TAB_INIT;
static const double h_sincos_tab[4 * PXF_TRIG_TAB_N] = PXF_SINCOS_TAB_INIT;
static const double h_tan_tab[2 * PXF_TRIG_TAB_N] = PXF_TAN_TAB_INIT;

// ---- woltersine : woltsurf.f95:167-215 --------------------------------------------------------
struct WolterSineOp {
    PXF_OP_DEFAULTS
    static constexpr unsigned RD = POS | DIR;
    static constexpr unsigned WR = POS | NRM;
    double twop, pp, cst, amp, freq, pi32, twopi32;
    template <int V>
    __device__ __forceinline__ void apply(Ray (&r)[V], const bool (&sel)[V], unsigned (&wm)[V]) const
    {
#pragma unroll
        for (int k = 0; k < V; k++) {
            double *c = r[k].c;
            double Fx = 0., Fy = 0., Fz = 0.;
            bool act = sel[k];
            for (int it = 0; act && it < kNewtonCap; it++) {
                const double x = c[X], y = c[Y], z = c[Z];
                double sn, cs;
                sincos(twopi32 * freq * z, &sn, &cs);
                const double rad = sqrt(x * x + y * y) + amp * sn;
                const double F = twop * z + pp + cst - rad * rad;
                Fx = -2. * x;
                Fy = -2. * y;
                Fz = twop - 2 * rad * amp * 2 * pi32 * freq * cs;
                const double Fp = Fx * c[L] + Fy * c[M] + Fz * c[N];
                const double delt = -F / Fp;
                c[X] = x + c[L] * delt;
                c[Y] = y + c[M] * delt;
                c[Z] = z + c[N] * delt;
                act = fabs(delt) > PXF_TOL_1EM10;
            }
            const double Fp = sqrt(Fx * Fx + Fy * Fy + Fz * Fz);
            c[UX] = Fx / Fp;
            c[UY] = Fy / Fp;
            c[UZ] = Fz / Fp;
            wm[k] = WR;
        }
    }
};

// ---- Legendre-Legendre deformed mirrors --------------------------------------------------------
constexpr int kLLMax = 256;  // (max axial + 1) * (max az + 1)
constexpr int kLLRows = 32;  // register form: at most this many (padded) axial rows

struct LLParams {
    double twop, pp, cst;    // primary
    double e2, twoe2, d;     // secondary
    double aa2, bb2, zfoc;   // ellipsoid: aa**2, bb**2
    double tol, zmid, zhalf;  // Newton tolerance, (zmax+zmin)/2, (zmax-zmin)/2
    double dphi, zspan;      // dphi, zmax-zmin
    int ka, kt;              // highest axial / azimuthal order present
    int rows, stride;        // layout of the matrix: C[i*stride + j]; rows = ka+1 (generic) or padded to even (register form)
};
struct LLData : LLParams {
    double cmat[kLLMax];     // by value in the kernel-argument segment
    double za[kLLRows], zb[kLLRows];  // (2n+3)/(n+2), (n+1)/(n+2): P_{n+2} = za_n x P_{n+1} - zb_n P_n
};

// R consecutive axial orders of the any-order form (ll_trace, KTMAX < 0): s0[r] = sum_j C[i+r][j] P_j(t),
// s1[r] = sum_j C[i+r][j] P'_j(t) from one run of the t-recurrence, then folded into the three sums with
// P_{i+r}(z), P'_{i+r}(z); the z-recurrence state (pz0 = P_i, pz1 = P_{i+1}, dz0, dz1) advances by R orders.
// `crow` points into the kernel-argument segment (wave-uniform: scalar loads), ra / rb into the LDS table.
template <int R>
__device__ __forceinline__ void ll_rows(const double *__restrict__ crow, int nt, double tc, double zc,
                                        const double *__restrict__ ra, const double *__restrict__ rb, int i,
                                        double &pz0, double &pz1, double &dz0, double &dz1, double &add, double &addt,
                                        double &addzz)
{
    double s0[R], s1[R];
#pragma unroll
    for (int r = 0; r < R; r++) s0[r] = s1[r] = 0.;
    double pt0 = 1., pt1 = tc, dt0 = 0., dt1 = 1.;
    for (int j = 0; j < nt; j++) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            const double cij = crow[r * nt + j];
            s0[r] = __builtin_fma(cij, pt0, s0[r]);   // tolerance-parity surface: FMA allowed here
            s1[r] = __builtin_fma(cij, dt0, s1[r]);
        }
        const double pn = __builtin_fma(ra[j] * tc, pt1, -(rb[j] * pt0));   // P_{j+2}
        const double dn = __builtin_fma((double)(2 * j + 3), pt1, dt0);     // P'_{j+2}
        pt0 = pt1;
        pt1 = pn;
        dt0 = dt1;
        dt1 = dn;
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        add = __builtin_fma(pz0, s0[r], add);
        addt = __builtin_fma(pz0, s1[r], addt);
        addzz = __builtin_fma(dz0, s0[r], addzz);
        const double pn = __builtin_fma(ra[i + r] * zc, pz1, -(rb[i + r] * pz0));
        const double dn = __builtin_fma((double)(2 * (i + r) + 3), pz1, dz0);
        pz0 = pz1;
        pz1 = pn;
        dz0 = dz1;
        dz1 = dn;
    }
}

// KIND: 0 primary (:220-288), 1 secondary (:293-379), 2 ellipsoid (:643-718).
// KTMAX >= 0 ("register form"): the azimuthal orders are known to be <= KTMAX.  The host pads the matrix to
// KTMAX+1 columns and an even number of rows with zeros, so the double sum has NO data-dependent control
// flow: P_j(t), P'_j(t) are generated once per Newton step into registers (fully unrolled, compile-time
// recurrence coefficients), the coefficients come straight from the kernel-argument segment through
// wave-uniform SCALAR loads (an SGPR pair is one operand of v_fma_f64) -- no LDS traffic at all -- and two
// axial orders run per trip as four independent FMA chains.  (Round 1-2 staged the matrix in LDS: four SIMDs
// reading one ds_read_b64 per two FMAs made the LDS pipe, not the fp64 pipe, the bound: 54 % VALU issue.)
// KTMAX < 0: any order; four axial orders share one run of the t-recurrence (ll_rows), coefficients by scalar loads.
template <int KIND, int KTMAX>
__device__ __forceinline__ void ll_trace(const LLParams &P, const double *__restrict__ cmat,
                                         const double *__restrict__ za, const double *__restrict__ zb, double *c,
                                         bool act)
{
    double Fx = 0., Fy = 0., Fz = 0.;
    const double izh = 1. / P.zhalf, tdp = 2 / P.dphi;   // wave-uniform, hoisted
    for (int it = 0; act && it < kNewtonCap; it++) {
        const double x = c[X], y = c[Y], z = c[Z];
        const double ang = atan2(y, x);
        const double zarg = (z - P.zmid) * izh;
        const double targ = ang * tdp;
        // legendre() clamps |x|>1 to sign(x); legendrep() returns 0 there (specialFunctions.f95:345-349,384-386)
        const double zc = fabs(zarg) > 1. ? copysign(1., zarg) : zarg;
        const double tc = fabs(targ) > 1. ? copysign(1., targ) : targ;
        const bool zin = !(fabs(zarg) > 1.), tin = !(fabs(targ) > 1.);
        double add = 0., addt = 0., addzz = 0.;
        double pz0 = 1., pz1 = zc, dz0 = 0., dz1 = 1.;  // P_i, P_{i+1}, P'_i, P'_{i+1} in zarg
        if constexpr (KTMAX >= 0) {
            constexpr int NT_ = KTMAX + 1;
            double Pt[NT_], Dt[NT_];
            Pt[0] = 1.;
            Dt[0] = 0.;
            if constexpr (KTMAX >= 1) {
                Pt[1] = tc;
                Dt[1] = 1.;
            }
#pragma unroll
            for (int j = 0; j + 2 <= KTMAX; j++) {
                Pt[j + 2] = ((double)(2 * j + 3) / (double)(j + 2)) * tc * Pt[j + 1] -
                            ((double)(j + 1) / (double)(j + 2)) * Pt[j];
                Dt[j + 2] = Dt[j] + (2 * j + 3) * Pt[j + 1];
            }
            // two axial orders per trip: four independent accumulation chains; P'(t) = 0 outside
            // [-1, 1] and P'(z) = 0 outside [-1, 1] enter the sums linearly -> applied once after the loop
            for (int i = 0; i < P.rows; i += 2) {
                double s0 = 0., s1 = 0., u0 = 0., u1 = 0.;
                const double *row0 = cmat + i * NT_, *row1 = row0 + NT_;   // kernel-argument segment: s_load
#pragma unroll
                for (int j = 0; j <= KTMAX; j++) {
                    const double c0 = row0[j], c1 = row1[j];
                    s0 = __builtin_fma(c0, Pt[j], s0);   // tolerance-parity surface: FMA allowed here
                    s1 = __builtin_fma(c0, Dt[j], s1);
                    u0 = __builtin_fma(c1, Pt[j], u0);
                    u1 = __builtin_fma(c1, Dt[j], u1);
                }
                add += pz0 * s0;
                addt += pz0 * s1;
                addzz += dz0 * s0;
                add += pz1 * u0;
                addt += pz1 * u1;
                addzz += dz1 * u0;
                // advance the z recurrences by two orders
                const double pn = za[i] * zc * pz1 - zb[i] * pz0;
                const double dn = dz0 + (2 * i + 3) * pz1;
                const double pn2 = za[i + 1] * zc * pn - zb[i + 1] * pz1;
                const double dn2 = dz1 + (2 * i + 5) * pn;
                pz0 = pn;
                pz1 = pn2;
                dz0 = dn;
                dz1 = dn2;
            }
            addt = tin ? addt : 0.;
            addzz = zin ? addzz : 0.;
        } else {
            // Any order.  R axial orders at a time (4, then 2, then 1 for what is left) share ONE run of the
            // t-recurrence: per azimuthal order 2 FMAs per coefficient plus the recurrence step (3 FMAs and a multiply,
            // its two ratios from the LDS table), instead of regenerating P_j(t), P'_j(t) for every coefficient.  The
            // coefficients come from the kernel-argument segment through scalar loads (wave-uniform addresses), as in
            // the register form; P'(t) = 0 outside [-1, 1] enters linearly and is applied once after the loops.
            const double *__restrict__ ra = za, *__restrict__ rb = zb;   // LDS: (2n+3)/(n+2), (n+1)/(n+2)
            const int nt = P.kt + 1;
            int i = 0;
            while (i <= P.ka) {
                const int left = P.ka + 1 - i;
                if (left >= 4) {
                    ll_rows<4>(cmat + i * nt, nt, tc, zc, ra, rb, i, pz0, pz1, dz0, dz1, add, addt, addzz);
                    i += 4;
                } else if (left >= 2) {
                    ll_rows<2>(cmat + i * nt, nt, tc, zc, ra, rb, i, pz0, pz1, dz0, dz1, add, addt, addzz);
                    i += 2;
                } else {
                    ll_rows<1>(cmat + i * nt, nt, tc, zc, ra, rb, i, pz0, pz1, dz0, dz1, add, addt, addzz);
                    i += 1;
                }
            }
            addt = tin ? addt : 0.;
            addzz = zin ? addzz : 0.;
        }
        const double r2 = y * y + x * x;
        const double ir2 = 1. / r2;
        const double addx = -(addt * tdp * (y * ir2));
        const double addy = addt * tdp * (x * ir2);
        const double addz = addzz * (2 / P.zspan);
        const double rr = sqrt(r2);
        // x/rr, y/rr stay IEEE divides: they set the Newton trajectory at first order, and the
        // secondary's F cancels to ~1e-8 (e^2(d+z)^2 - z^2), so a 1-ulp change of the path moves
        // the converged z by ~2e-9 -- more than the 1e-10 parity these surfaces are held to
        const double xr = x / rr, yr = y / rr;
        const double G = rr + add;
        double F;
        if (KIND == 0) {
            F = -(G * G - P.pp - P.twop * z - P.cst);
            Fx = -(2 * G * (xr + addx));
            Fy = -(2 * G * (yr + addy));
            Fz = P.twop - 2 * G * addz;
        } else if (KIND == 1) {
            const double dz = P.d + z;
            F = -(G * G - P.e2 * (dz * dz) + z * z);
            Fx = -(2 * G * (xr + addx));
            Fy = -(2 * G * (yr + addy));
            Fz = P.twoe2 * dz - 2 * z - 2 * G * addz;
        } else {
            const double zz = z - P.zfoc;
            F = zz * zz / P.aa2 + G * G / P.bb2 - 1.;
            Fx = 2 * G / P.bb2 * (xr + addx);
            Fy = 2 * G / P.bb2 * (yr + addy);
            Fz = 2 * zz / P.aa2 + (2 * G / P.bb2) * addz;
        }
        const double Fp = Fx * c[L] + Fy * c[M] + Fz * c[N];
        const double delt = -F / Fp;
        c[X] = x + c[L] * delt;
        c[Y] = y + c[M] * delt;
        c[Z] = z + c[N] * delt;
        act = fabs(delt) > P.tol;
    }
    const double Fp = sqrt(Fx * Fx + Fy * Fy + Fz * Fz);
    c[UX] = Fx / Fp;
    c[UY] = Fy / Fp;
    c[UZ] = Fz / Fp;
}

template <int KIND, int KTMAX = -1>
struct LLOp : LLData {
    static constexpr int MAXV = 1;
    static constexpr unsigned RD = POS | DIR;
    static constexpr unsigned WR = POS | NRM;

    // LDS (any-order form only): the recurrence ratios a_n = (2n+3)/(n+2), b_n = (n+1)/(n+2) for n < kLLMax
    __device__ __forceinline__ double *lds() const
    {
        __shared__ double s_c[2 * kLLMax];
        return s_c;
    }
    __device__ __forceinline__ void prologue() const
    {
        if constexpr (KTMAX < 0) {
            double *s = lds();
            for (int i = threadIdx.x; i < kLLMax; i += kBlock) {
                s[i] = (double)(2 * i + 3) / (double)(i + 2);
                s[kLLMax + i] = (double)(i + 1) / (double)(i + 2);
            }
            __syncthreads();
        }
    }

    template <int V>
    __device__ __forceinline__ void apply(Ray (&r)[V], const bool (&sel)[V], unsigned (&wm)[V]) const
    {
        if constexpr (KTMAX < 0) {
            const double *s = lds();
            ll_trace<KIND, KTMAX>(*this, cmat, s, s + kLLMax, r[0].c, sel[0]);
        } else {
            ll_trace<KIND, KTMAX>(*this, cmat, za, zb, r[0].c, sel[0]);
        }
        wm[0] = WR;
    }
};

// ---- Wolter-Schwarzschild : woltsurf.f95:387-476 (primary), 484-588 (secondary) ----------------
// BACK = wsprimaryBack / wssecondaryBack (:726-933): the surface function is evaluated at the ray's
// radius shrunk by `thick` (the back side of a substrate of thickness `thick`).
//
// The reference evaluates, per Newton step, asin / atan2, tan, four sin / cos and two pow.  All of
// the angle functions are functions of ONE angle beta that is itself defined by a ratio of
// lengths (primary: sin beta = r/f; secondary: tan beta = r/z), so they are obtained here
// algebraically -- cos beta from a square root, the half-angle values from
// cos^2(b/2) = (1+cos b)/2, sin(b/2) = sin b / (2 cos(b/2)), tan(b/2) = sin b / (1+cos b), none of
// which cancels -- and pow(kterm,-k) = pow(kterm,1-k)/kterm: one pow, a few divides and square
// roots per step instead of eight libm calls (6.9 -> 3.7 ms per 1e8 rays).  The functions
// of the constant betas come from the host (glibc: the reference's own values).  These are exact
// identities; results move by rounding only, far inside the 2e-8 mm these surfaces are held to (the
// reference's own 1.e-8 stopping step).  Branch tests use the same monotone images
// (beta <= betas  <=>  sin beta <= sin betas on [0, pi/2];  <=>  cos beta >= cos betas on [0, pi]).
// RN(z / sqrt(r*r + z*z)): the quotient formed in double-double arithmetic (error ~1e-31) and rounded
// to double once.  Explicit fma() calls fuse regardless of -ffp-contract=off.
__device__ __forceinline__ double cos_of_ratio_rn(double r, double z)
{
    const double h1 = r * r, l1 = fma(r, r, -h1);
    const double h2 = z * z, l2 = fma(z, z, -h2);
    double sh = h1 + h2;
    const double bb = sh - h1;
    double sl = ((h1 - (sh - bb)) + (h2 - bb)) + (l1 + l2);     // two_sum error + the product tails
    const double t = sh + sl;
    sl = sl - (t - sh);
    sh = t;
    const double ph = sqrt(sh);
    const double pe = fma(-ph, ph, sh) + sl;                    // s - ph^2
    const double pl = pe / (2 * ph);
    const double qh = z / ph;
    const double ql = (fma(-qh, ph, z) - qh * pl) / ph;
    return qh + ql;
}

template <bool SECONDARY, bool BACK = false>
struct WSOp {
    PXF_OP_DEFAULTS
    static constexpr unsigned RD = POS | DIR;
    static constexpr unsigned WR = POS | NRM;
    double betas, ff, g, k, thick;
    double sbs, cbs, sb2, cb2s, tbs, tb2s;  // sin, cos of betas; sin, cos of betas/2; tan betas, tan betas/2
    template <int V>
    __device__ __forceinline__ void apply(Ray (&r)[V], const bool (&sel)[V], unsigned (&wm)[V]) const
    {
#pragma unroll
        for (int q = 0; q < V; q++) {
            double *c = r[q].c;
            const double xi = c[X], yi = c[Y], zi = c[Z];
            double delt = 100., Fx = 0., Fy = 0., Fz = 0.;
            int cnt = 0;
            if (!sel[q]) {
                wm[q] = 0;
                continue;
            }
            while (fabs(delt) > PXF_TOL_1EM8) {
                const double xa = c[X], ya = c[Y], z = c[Z];
                double x = xa, y = ya;
                if (BACK) {  // (rb - thick) * (cos, sin)(atan2(ya, xa)), woltsurf.f95:760-763 / 862-865
                    const double rb = sqrt(xa * xa + ya * ya);
                    // theta and its sincos with the C library's own rounding (pxf_libm_match.h): the secondary's F
                    // is quantised by (1 - cos beta), so an ulp of x, y here decides which quantum a ray whose
                    // first steps land on the mirror's edge ends in
                    // (the primary has no such amplifier: there the offset is the plain scaling, 1.5 - 4e-12 mm RMS
                    // either way, at half the instructions)
                    if (SECONDARY) {
                        double th, sn, cs;
                        if (!libm_sincos_atan2(ya, xa, c_sincos_tab, c_tan_tab, th, sn, cs)) sincos(atan2(ya, xa), &sn, &cs);
                        x = (rb - thick) * cs;
                        y = (rb - thick) * sn;
                    } else {
                        const double sc = (rb - thick) / rb;
                        x = sc * xa;
                        y = sc * ya;
                    }
                }
                const double r2 = x * x + y * y;
                const double rr = sqrt(r2);
                double F, Fb;
                if (!SECONDARY) {
                    const double s = rr / ff;              // sin(beta), beta = asin(rr/ff)
                    const double den = sqrt(1 - r2 / (ff * ff));  // cos(beta) of the ray's own radius (:447)
                    const bool flag = s <= sbs;            // beta <= betas
                    double sb, cb, cb2, sh, pk, pkm;
                    if (flag) {
                        sb = sbs;
                        cb = cbs;
                        cb2 = cb2s;
                        sh = sb2;
                        pk = 0.;   // kterm = 0: 0**(1-k)
                        pkm = 0.;  // unused: Fb is replaced below
                    } else {
                        sb = s;
                        cb = den;
                        cb2 = sqrt((1 + cb) / 2);
                        sh = s / (2 * cb2);
                        const double t = s / (1 + cb);
                        const double kterm = (1 / k) * (t * t) - 1;
                        pk = pow(kterm, 1 - k);
                        pkm = pk / kterm;
                    }
                    const double cb2_2 = cb2 * cb2;
                    F = -z - ff * (sb2 * sb2) + ff * ff * (sb * sb) / (4 * ff * (sb2 * sb2)) + g * (cb2_2 * cb2_2) * pk;
                    Fz = -1.;
                    if (flag) {
                        Fb = ff * ff * sbs * cbs / (2 * ff * (sb2 * sb2)) + g * (1 - k) * cb2s * sb2 * (1 / k);
                        const double q2 = r2 + z * z;
                        F = F + (rr - ff * sbs) * z / q2 * Fb;
                        Fz = Fz + (rr - ff * sbs) * (r2 - z * z) / (q2 * q2) * Fb;
                    } else {
                        Fb = ff * ff * sb * cb / (2 * ff * (sb2 * sb2)) - 2 * g * (cb2_2 * cb2) * sh * pk +
                             g * (1 - k) * cb2 * sh * pkm * (1 / k);
                    }
                    Fx = Fb * (x / den / ff / rr);
                    Fy = Fb * (y / den / ff / rr);
                } else {
                    const double r3 = r2 + z * z;
                    const double qq = sqrt(r3);
                    const double sb = rr / qq;                 // sin of beta = atan2(rr, z)
                    // cos(beta) must carry the reference's OWN rounding: its `a` holds (1 - cos beta) with
                    // cos beta ~ 1 - 3e-4, so one ulp of cos beta moves the converged z by ~2.6e-9 mm (1e-16 /
                    // 3e-4 x 8400 mm).  glibc's cos is not correctly rounded (~5e-4 of its results are the other
                    // neighbour), so it is restated operation by operation (pxf_libm_match.h: beta as the
                    // correctly rounded arctangent, then the library's table + polynomial cosine).  Outside that
                    // routine's range the correctly rounded quotient z / sqrt(rr^2 + z^2) stands in.
                    double beta = 0., cb;
                    bool flag;
                    const bool matched = libm_cos_atan2(rr, z, c_sincos_tab, c_tan_tab, beta, cb);
                    if (matched) {
                        flag = beta <= betas;
                    } else {
                        cb = cos_of_ratio_rn(rr, z);
                        flag = cb >= cbs;                     // beta <= betas
                    }
                    if (flag) {
                        F = -z + cbs / (1 / ff);  // cos(betas) / a with a = 1/ff, in the reference's two roundings (:513,521)
                        const double dadbs = sbs / ff / (1 - cbs) + (k + 1) * (cbs + 1) * tb2s / (cb2s * cb2s) / 2 / g / k;
                        const double dbdzs = -(sbs * sbs) / rr;
                        const double gam = (-ff * sbs - ff * ff * cbs * dadbs) * dbdzs;
                        F = F + gam * (z - rr / tbs);
                        Fx = -2. / tbs * x / rr;
                        Fy = -2. / tbs * y / rr;
                        Fz = gam - 1.;
                    } else {
                        // tan(beta/2): the library's own polynomial where it applies (kterm cancels to ~1e-10 at the
                        // mirror's edge, see pxf_libm_match.h), else the half-angle identity
                        const double hb = .5 * beta;
                        const double t = (matched && hb > 2e-8 && hb < 0.0607) ? libm_tan_small(hb) : sb / (1 + cb);
                        const double kterm = (1 / k) * (t * t) - 1;
                        const double p1 = pow(kterm, 1 + k), pkk = p1 / kterm;  // kterm**(1+k), kterm**k
                        const double a = (1 - cb) / (1 - cbs) / ff + (1 + cb) / (2 * g) * p1;
                        F = -z + cb / a;
                        const double c2sq = (1 + cb) / 2;         // cos(beta/2)**2
                        const double dadb = sb / ff / (1 - cbs) - sb / (2 * g) * p1 +
                                            (k + 1) * (cb + 1) * t * pkk / 2 / g / k / c2sq;
                        Fb = -sb / a - cb / (a * a) * dadb;
                        Fx = Fb * (x * z / r3 / rr);
                        Fy = Fb * (y * z / r3 / rr);
                        Fz = -1. + Fb * (-rr / r3);
                    }
                }
                const double Fp = Fx * c[L] + Fy * c[M] + Fz * c[N];
                delt = -F / Fp;
                c[X] = xa + c[L] * delt;
                c[Y] = ya + c[M] * delt;
                c[Z] = z + c[N] * delt;
                if (cnt > 25 || delt != delt) {  // :454-460 / :566-572: give up, restore the start point
                    delt = 0.;
                    c[X] = xi;
                    c[Y] = yi;
                    c[Z] = zi;
                    cnt = 1000;
                }
                cnt = cnt + 1;
            }
            if (cnt < 26) {
                const double Fp = sqrt(Fx * Fx + Fy * Fy + Fz * Fz);
                const double sgn = SECONDARY ? 1. : -1.;  // the primary stores the inward normal (:470-472)
                c[UX] = sgn * Fx / Fp;
                c[UY] = sgn * Fy / Fp;
                c[UZ] = sgn * Fz / Fp;
                wm[q] = WR;
            } else {
                wm[q] = POS;  // position restored, normal left as it was
            }
        }
    }
};

// ---- host side shared by the stand-alone entries and the program launcher --------------------------
static inline int fold_ll(LLData &op, double zmax, double zmin, double dphi, const double *coeff,
                   const int32_t *axial, const int32_t *az, int32_t cnum)
{
    if (cnum < 0 || (cnum > 0 && (!coeff || !axial || !az))) return PXF_EINVAL;
    int ka = 0, kt = 0;
    for (int a = 0; a < cnum; a++) {
        if (axial[a] < 0 || az[a] < 0) return PXF_EINVAL;
        if (axial[a] > ka) ka = axial[a];
        if (az[a] > kt) kt = az[a];
    }
    if ((ka + 1) * (kt + 1) > kLLMax) return PXF_ERANGE;
    for (int i = 0; i < kLLMax; i++) op.cmat[i] = 0.;
    for (int a = 0; a < cnum; a++) op.cmat[axial[a] * (kt + 1) + az[a]] += coeff[a];
    op.ka = ka;
    op.kt = kt;
    op.rows = ka + 1;
    op.stride = kt + 1;
    for (int i = 0; i < kLLRows; i++) {
        op.za[i] = (double)(2 * i + 3) / (double)(i + 2);
        op.zb[i] = (double)(i + 1) / (double)(i + 2);
    }
    op.zmid = (zmax + zmin) / 2.;
    op.zhalf = (zmax - zmin) / 2.;
    op.dphi = dphi;
    op.zspan = zmax - zmin;
    return PXF_OK;
}


// Surface constants of the three Legendre-Legendre kinds (host libm: the reference's own values)
inline void ll_setup_primary(LLParams &op, double r0, double z0)   // woltsurf.f95:234-239 (psi = 1)
{
    double p, d, e;
    pxf_wolter_params(r0, z0, 1., &p, &d, &e);
    op.twop = 2 * p;
    op.pp = p * p;
    op.cst = 4 * (e * e) * p * d / (e * e - 1);
    op.tol = PXF_TOL_1EM10;
}
inline void ll_setup_secondary(LLParams &op, double r0, double z0, double psi)   // :306-312
{
    double p, d, e;
    pxf_wolter_params(r0, z0, psi, &p, &d, &e);
    op.e2 = e * e;
    op.twoe2 = 2 * (e * e);
    op.d = d;
    op.tol = (double)1.e-7f;  // :312
}
inline void ll_setup_ellipsoid(LLParams &op, double r0, double z0, double psi, double S)   // :656-664
{
    const double P = r0 / sin((psi * asin(r0 / z0) - asin(r0 / S)) / (1 + psi));
    const double ff = (S + P) / 2.;
    const double bq = -(r0 * r0 + (ff - P) * (ff - P) + ff * ff);
    const double cq = ff * ff * ((ff - P) * (ff - P));
    const double aa = sqrt((-bq + sqrt(bq * bq - 4 * cq)) / 2.);
    const double bb = sqrt(aa * aa - ff * ff);
    op.aa2 = aa * aa;
    op.bb2 = bb * bb;
    op.zfoc = ff - P + z0;
    op.tol = PXF_TOL_1EM10;
}
inline WolterSineOp make_woltersine(double r0, double z0, double amp, double freq)   // :177-178: psi = 1
{
    double p, d, e;
    pxf_wolter_params(r0, z0, 1., &p, &d, &e);
    return WolterSineOp{2 * p, p * p, 4 * (e * e) * p * d / (e * e - 1), amp, freq, (double)acosf(-1.f),
                        (double)(2.f * acosf(-1.f))};
}

// Register form: repack the matrix to `cols` columns and an even number of rows (zero padding).
// Returns false when it does not fit (then the generic LDS form runs).
static inline bool ll_repack(LLData &d, int cols)
{
    const int rows = (d.ka + 2) & ~1;
    if (d.kt + 1 > cols || rows > kLLRows || rows * cols > kLLMax) return false;
    double m[kLLMax];
    for (int i = 0; i < kLLMax; i++) m[i] = 0.;
    for (int i = 0; i <= d.ka; i++)
        for (int j = 0; j <= d.kt; j++) m[i * cols + j] = d.cmat[i * (d.kt + 1) + j];
    for (int i = 0; i < kLLMax; i++) d.cmat[i] = m[i];
    d.rows = rows;
    d.stride = cols;
    return true;
}

// launch with the smallest compile-time bound that covers the azimuthal orders present
template <int KIND>
static int ll_launch(const LLData &d0, const Cols &c, int64_t num, const uint8_t *mask, pxf_stream_t stream)
{
    static const bool force_generic = getenv("PXF_LL_GENERIC") != nullptr;   // dev knob (tools/bench_ops.py A/B)
    LLData d = d0;
#define PXF_LL_TRY(KT)                                       \
    if (!force_generic && d.kt <= KT && ll_repack(d, KT + 1)) { \
        LLOp<KIND, KT> op{};                                 \
        static_cast<LLData &>(op) = d;                       \
        return launch_rays(c, num, mask, stream, op);        \
    }
    PXF_LL_TRY(3)
    PXF_LL_TRY(7)
    PXF_LL_TRY(11)
    PXF_LL_TRY(15)
#undef PXF_LL_TRY
    LLOp<KIND, -1> op{};
    static_cast<LLData &>(op) = d;
    return launch_rays(c, num, mask, stream, op);
}

// Chase parameters of the equivalent Wolter-I plus the trig of the constant betas (host, glibc)
template <class OP>
static void chase(double alpha, double z0, double psi, OP &op)
{
    op.betas = 4 * alpha;  // woltsurf.f95:398-401
    op.ff = z0 / cos(op.betas);
    op.g = op.ff / psi;
    const double t = tan(op.betas / 2);
    op.k = t * t;
    op.sbs = sin(op.betas);
    op.cbs = cos(op.betas);
    op.sb2 = sin(op.betas / 2);
    op.cb2s = cos(op.betas / 2);
    op.tbs = tan(op.betas);
    op.tb2s = t;
}

}  // namespace pxf
