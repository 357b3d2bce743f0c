// pxf_ops_zernike.h -- the Zernike evaluation and Newton Op (zernsurf.f95:8-203 with zernset,
// specialFunctions.f95:142-232), shared by the stand-alone kernels (pxf_zernike.hip) and the fused
// program kernel (pxf_program_zern.hip).  See pxf_zernike.hip for the design notes.
//
// Floating-point contraction: this code is tolerance-parity and fp64-VALU-bound, so its two hot
// function bodies ask for FMA contraction THEMSELVES (`#pragma clang fp contract(fast)`); a
// translation unit compiled -ffp-contract=off (as every bit-exact Op needs) can therefore include
// this header and still run the Zernike arithmetic fused.
#pragma once
#include <math.h>

#include "pxf_kernel.h"

namespace pxf {

constexpr int kZernMaxN = 15;
constexpr int zern_npairs(int nmax) { return ((nmax + 2) / 2) * ((nmax + 3) / 2); }  // #(n,m): n<=nmax
// index of pair (n,m) (m same parity as n, 0<=m<=n) in the table: pairs of all smaller n, then m/2
constexpr int zern_pair(int n, int m) { return zern_npairs(n - 1) + m / 2; }

struct ZernTab {
    double2 a[zern_npairs(kZernMaxN)];  // .x = Ac (cos-type, and m=0), .y = As (sin-type)
};

// q-recursion coefficients, specialFunctions.f95:195-197
constexpr double zh3(double n, double m) { return -4 * (m + 2) * (m + 1) / (n + m + 2) / (n - m); }
constexpr double zh2(double n, double m) { return zh3(n, m) * (n + m + 4) * (n - m - 2) / 4. / (m + 3) + (m + 2); }
constexpr double zh1(double n, double m)
{
    return .5 * (m + 4) * (m + 3) - (m + 4) * zh2(n, m) + zh3(n, m) * (n + m + 6) * (n - m - 4) / 8.;
}

// The (h1,h2,h3) of every (n,m) with m <= n-4, for the instantiations that read them from LDS instead of
// carrying them as immediates.  Why: a fully unrolled evaluation for n >= 8 has hundreds of distinct
// 64-bit constants, the compiler keeps them in SGPRs, runs out (106), and spills SGPRs into VGPR lanes
// inside the Newton loop; several of those instantiations then returned WRONG results (not rounding:
// z off by 1 mm for orders 11, 12, 14; 1/rad in the x column for 15) whichever way the VGPR overflow was
// handled (AGPRs or scratch).  Read from LDS (a broadcast ds_read) the constants occupy no scalar registers.
struct ZernH {
    double h[zern_npairs(kZernMaxN)][3];
};
constexpr ZernH make_zern_h()
{
    ZernH t{};
    for (int n = 0; n <= kZernMaxN; n++)
        for (int m = n & 1; m <= n - 4; m += 2) {
            t.h[zern_pair(n, m)][0] = zh1(n, m);
            t.h[zern_pair(n, m)][1] = zh2(n, m);
            t.h[zern_pair(n, m)][2] = zh3(n, m);
        }
    return t;
}
static __device__ __constant__ const ZernH kZernH = make_zern_h();
#ifndef PXF_ZERN_HLDS_FROM
#define PXF_ZERN_HLDS_FROM 8  // (dev builds of tools/zern_repro.sh override this and PXF_ZERN_LOOP_FROM)
#endif
constexpr bool zern_h_in_lds(int nmax) { return nmax >= PXF_ZERN_HLDS_FROM; }

template <int NMAX, bool HLDS>
__device__ __forceinline__ double (*zern_h_lds())[3]
{
    __shared__ double s_h[HLDS ? zern_npairs(NMAX) : 1][3];
    return s_h;
}
template <int NMAX, bool HLDS = zern_h_in_lds(NMAX)>
__device__ __forceinline__ void zern_h_stage()  // call before the block-wide barrier of the Op's prologue
{
    if constexpr (HLDS) {
        double(*s)[3] = zern_h_lds<NMAX, HLDS>();
        for (int i = threadIdx.x; i < zern_npairs(NMAX); i += kBlock) {
            s[i][0] = kZernH.h[i][0];
            s[i][1] = kZernH.h[i][1];
            s[i][2] = kZernH.h[i][2];
        }
    }
}

// The same evaluation with RUN-TIME loops and no per-thread arrays, for the high orders (n >= kZernLoopFrom)
// whose fully unrolled form needs more registers than a lane has (heavy spills; several spilled
// instantiations were found to return wrong results).  cos/sin(n theta) ascend with n by the
// angle-addition recurrence (three consecutive multiples are kept), and within a chain of fixed n
// they DEscend by two with  cos((m-2)t) = 2 cos(2t) cos(m t) - cos((m+2)t)  (same for sin); the
// powers rho'^n .. rho'^(n-3) are a sliding window.  Same radial q-recursion, same accumulation order.
#ifndef PXF_ZERN_LOOP_FROM
#define PXF_ZERN_LOOP_FROM 9
#endif
constexpr int kZernLoopFrom = PXF_ZERN_LOOP_FROM;

__device__ __forceinline__ void zern_eval_loop(const double2 *__restrict__ tab, const double (*__restrict__ hq)[3],
                                               int nmax, double x, double y, double rad, double inv_rad, double &S,
                                               double &Sx, double &Sy)
{
#pragma clang fp contract(fast)
    const double rho = sqrt(x * x + y * y);
    const double inv_rho = 1. / rho;
    const double ct = x * inv_rho, st = y * inv_rho;
    const double rp = rho * inv_rad;
    const double inv_rp = rad * inv_rho;
    const double irp2 = inv_rp * inv_rp, irp3 = irp2 * inv_rp;
    const double c2t = ct * ct - st * st;  // cos(2 theta)
    // (cos, sin) of n*theta, (n+1)*theta, (n+2)*theta for the current n
    double cn = 1., sn = 0., cn1 = ct, sn1 = st, cn2 = c2t, sn2 = 2. * st * ct;
    // rho'^n, rho'^(n-1), rho'^(n-2), rho'^(n-3) (entries below the zeroth power are never used)
    double p0 = 1., p1 = 0., p2 = 0., p3 = 0.;
    double Sv = 0., Sr = 0., St = 0.;
    int pair = 0;  // index of (n, m = n & 1) in the table; pairs of one n are consecutive, m ascending
#pragma unroll 1
    for (int nn = 0; nn <= nmax; nn++) {
        const int npn = nn / 2 + 1;  // pairs of this n: m = n, n-2, ... down to n & 1
        double R2 = 0., Rp2 = 0., R4 = 0., Rp4 = 0.;
        double cm = cn, sm = sn, cmp = cn2, smp = sn2;  // at m and at m+2
#pragma unroll 1
        for (int k = 0; k < npn; k++) {
            const int mm = nn - 2 * k;
            double R, Rp;
            if (k == 0) {
                R = p0;
                Rp = nn == 0 ? 0. : (double)nn * p1;
            } else if (k == 1) {
                R = (double)nn * p0 - (double)(nn - 1) * p2;
                const double dnn = (double)nn * p1;
                const double dn2 = nn - 2 == 0 ? 0. : (double)(nn - 2) * p3;
                Rp = (double)nn * dnn - (double)(nn - 1) * dn2;
            } else {
                const double *h = hq[pair + mm / 2];
                const double h1 = h[0], h2 = h[1], h3 = h[2];
                const double g = h2 + h3 * irp2;
                R = h1 * R4 + g * R2;
                Rp = h1 * Rp4 + g * Rp2 - 2 * h3 * irp3 * R2;
            }
            R4 = R2;
            Rp4 = Rp2;
            R2 = R;
            Rp2 = Rp;
            const double2 a = tab[pair + mm / 2];
            const double t1 = a.x * cm + a.y * sm;
            Sv += R * t1;
            Sr += Rp * t1;
            if (mm > 0) St += R * ((double)mm * (a.y * cm - a.x * sm));
            // step m -> m-2
            const double cd = 2. * c2t * cm - cmp, sd = 2. * c2t * sm - smp;
            cmp = cm;
            smp = sm;
            cm = cd;
            sm = sd;
        }
        pair += npn;
        // n -> n+1
        cn = cn1;
        sn = sn1;
        cn1 = cn2;
        sn1 = sn2;
        cn2 = cn1 * ct - sn1 * st;
        sn2 = sn1 * ct + cn1 * st;
        p3 = p2;
        p2 = p1;
        p1 = p0;
        p0 = p0 * rp;
    }
    const double Srho = Sr * inv_rad;
    const double Sth = St * inv_rho;
    S = Sv;
    Sx = ct * Srho - st * Sth;
    Sy = st * Srho + ct * Sth;
}

// Value and gradient of  S(x,y) = sum_i coeff_i Z_i(rho/rad, theta)  from the folded (Ac,As) table:
// returns S, and the cartesian gradient (dS/dx, dS/dy).  rho == 0 gives NaN, as the reference's
// division by rho does (zernsurf.f95:63-66).
template <int NMAX, bool HLDS = zern_h_in_lds(NMAX)>
__device__ __forceinline__ void zern_eval(const double2 *__restrict__ tab, double x, double y, double rad,
                                          double inv_rad, double &S, double &Sx, double &Sy)
{
#pragma clang fp contract(fast)
    if constexpr (NMAX >= kZernLoopFrom) {
        zern_eval_loop(tab, zern_h_lds<NMAX, HLDS>(), NMAX, x, y, rad, inv_rad, S, Sx, Sy);
        return;
    }
    // one reciprocal per evaluation instead of nine divides (an IEEE fp64 divide is ~11 VALU
    // instructions; this kernel is VALU-bound and tolerance-parity, so 1/rho is shared)
    const double rho = sqrt(x * x + y * y);
    const double inv_rho = 1. / rho;
    const double ct = x * inv_rho, st = y * inv_rho;  // cos/sin(atan2(y,x))
    const double rp = rho * inv_rad;
    const double inv_rp = rad * inv_rho;
    const double irp2 = inv_rp * inv_rp, irp3 = irp2 * inv_rp;
    double cm[NMAX + 1], sm[NMAX + 1], pw[NMAX + 1];
    cm[0] = 1.;
    sm[0] = 0.;
    pw[0] = 1.;
#pragma unroll
    for (int k = 1; k <= NMAX; k++) {
        cm[k] = cm[k - 1] * ct - sm[k - 1] * st;
        sm[k] = sm[k - 1] * ct + cm[k - 1] * st;
        pw[k] = pw[k - 1] * rp;
    }
    double Sv = 0., Sr = 0., St = 0.;
#pragma unroll
    for (int nn = 0; nn <= NMAX; nn++) {
        double R2 = 0., Rp2 = 0., R4 = 0., Rp4 = 0.;  // R_n^{m+2}, R_n^{m+4} of the chain
#pragma unroll
        for (int mm = nn; mm >= 0; mm -= 2) {
            double R, Rp;
            if (mm == nn) {
                R = pw[nn];
                Rp = nn == 0 ? 0. : (double)nn * pw[nn > 0 ? nn - 1 : 0];
            } else if (mm == nn - 2) {
                R = (double)nn * pw[nn] - (double)(nn - 1) * pw[nn - 2];
                const double dnn = (double)nn * pw[nn - 1];
                const double dn2 = nn - 2 == 0 ? 0. : (double)(nn - 2) * pw[nn > 2 ? nn - 3 : 0];
                Rp = (double)nn * dnn - (double)(nn - 1) * dn2;
            } else {
                double h1, h2, h3;
                if constexpr (HLDS) {
                    const double *hq = zern_h_lds<NMAX, HLDS>()[zern_pair(nn, mm)];
                    h1 = hq[0];
                    h2 = hq[1];
                    h3 = hq[2];
                } else {
                    h3 = zh3(nn, mm);
                    h2 = zh2(nn, mm);
                    h1 = zh1(nn, mm);
                }
                const double g = h2 + h3 * irp2;
                R = h1 * R4 + g * R2;
                Rp = h1 * Rp4 + g * Rp2 - 2 * h3 * irp3 * R2;
            }
            R4 = R2;
            Rp4 = Rp2;
            R2 = R;
            Rp2 = Rp;
            const double2 a = tab[zern_pair(nn, mm)];
            const double t1 = a.x * cm[mm] + a.y * sm[mm];
            Sv += R * t1;
            Sr += Rp * t1;
            if (mm > 0) St += R * ((double)mm * (a.y * cm[mm] - a.x * sm[mm]));
        }
    }
    const double Srho = Sr * inv_rad;   // dS/drho
    const double Sth = St * inv_rho;    // (1/rho) dS/dtheta
    S = Sv;
    Sx = ct * Srho - st * Sth;
    Sy = st * Srho + ct * Sth;
}

// CONIC: EXTENSION without a reference counterpart (BASELINE config 4 names a "Zernike-perturbed
// conic"; the reference's zernsurf is a pure Zernike sag): the sag of a conic of radius
// R and constant K is added under the Zernike sum,  F = z - c rho^2/(1+sqrt(1-(K+1) c^2 rho^2)) - S.
// Parity unpinned; checked against the conic kernel (coeff = 0) and an independent NumPy Newton.
#ifdef PXF_ZERN_MIN_WAVES_ALL   // dev builds (tools/zern_repro.sh): one occupancy target for every order
#define PXF_ZERN_MIN_WAVES(NMAX) PXF_ZERN_MIN_WAVES_ALL
#endif
#ifndef PXF_ZERN_MIN_WAVES
// >= 2 waves per SIMD keeps a kernel within 256 registers per lane.  Beyond that the compiler parks
// values in AGPRs, and the NMAX = 15 program kernel built that way returned 1/rad in the x column
// (every other column right): wrong code, not rounding.  Scratch spills are slower but correct.
#define PXF_ZERN_MIN_WAVES(NMAX) ((NMAX) <= 7 ? 3 : 2)
#endif
// HLDS: read the q-recursion constants from LDS instead of carrying them as immediates (n >= 8, see above)
template <int NMAX, bool CONIC = false, bool HLDS = zern_h_in_lds(NMAX)>
struct ZernOp {
    static constexpr int MAXV = 1;
    static constexpr int MIN_WAVES = PXF_ZERN_MIN_WAVES(NMAX);
    static constexpr unsigned RD = POS | DIR | B(OPD);  // opd column only touched when has_opd
    static constexpr unsigned WR = POS | NRM | B(OPD);
    static constexpr int NP = zern_npairs(NMAX);
    double2 tab_arg[zern_npairs(NMAX)];  // folded table, by value in the kernel-argument segment
    double rad, nr;
    int has_opd;
    double cc = 0., K1 = 0.;  // conic: curvature 1/R and K+1 (CONIC only)

    __device__ __forceinline__ double2 *lds() const
    {
        __shared__ double2 s_tab[zern_npairs(NMAX)];
        return s_tab;
    }

    __device__ __forceinline__ void prologue() const
    {
        double2 *s = lds();
        for (int i = threadIdx.x; i < NP; i += kBlock) s[i] = tab_arg[i];
        zern_h_stage<NMAX, HLDS>();
        __syncthreads();
    }

    template <int V>
    __device__ __forceinline__ void apply(Ray (&r)[V], const bool (&sel)[V], unsigned (&wm)[V]) const
    {
#pragma clang fp contract(fast)
        static_assert(V == 1, "Zernike kernel is one ray per thread");
        const double2 *__restrict__ tab = lds();
        double *c = r[0].c;
        const double l = c[L], m = c[M], n = c[N];
        double x = c[X], y = c[Y], z = c[Z];
        double t = 0., Fx = 0., Fy = 0., Fz = 0.;
        const double inv_rad = 1. / rad;
        bool act = sel[0];
        for (int it = 0; act && it < kNewtonCap; it++) {
            double S, Sx, Sy;
            zern_eval<NMAX, HLDS>(tab, x, y, rad, inv_rad, S, Sx, Sy);
            double F = z - S;   // F = z - sum c_i Z_i ; grad F = (-dS/dx, -dS/dy, 1)
            Fx = -Sx;
            Fy = -Sy;
            Fz = 1.;
            if (CONIC) {
                const double r2 = x * x + y * y;
                const double sq = sqrt(1. - K1 * cc * cc * r2);
                F -= cc * r2 / (1. + sq);
                const double g = cc / sq;  // (1/rho) d(sag)/d(rho)
                Fx -= g * x;
                Fy -= g * y;
            }
            const double Fp = Fx * l + Fy * m + Fz * n;
            const double delta = -F / Fp;
            x = x + l * delta;
            y = y + m * delta;
            z = z + n * delta;
            t = t + delta;
            act = fabs(delta) > PXF_TOL_1EM10;
        }
        const double iFp = 1. / sqrt(Fx * Fx + Fy * Fy + Fz * Fz);
        c[X] = x;
        c[Y] = y;
        c[Z] = z;
        c[UX] = Fx * iFp;
        c[UY] = Fy * iFp;
        c[UZ] = Fz * iFp;
        c[OPD] = c[OPD] + t * nr;  // zernsurf.f95:199 -- once, from the accumulated path length
        wm[0] = POS | NRM | (has_opd ? B(OPD) : 0u);
    }
};

// The opd column is optional.  The generic kernel loads every column an Op declares in RD, so the
// OPD and plain variants are distinct types that differ only in their RD/WR sets (the plain one
// never touches the opd column: 96 B/ray instead of 112).
template <int NMAX, bool WOPD, bool CONIC = false>
struct ZernOpSel : ZernOp<NMAX, CONIC> {
    static constexpr unsigned RD = POS | DIR | (WOPD ? B(OPD) : 0u);
    static constexpr unsigned WR = POS | NRM | (WOPD ? B(OPD) : 0u);
    template <int V>
    __device__ __forceinline__ void apply(Ray (&r)[V], const bool (&sel)[V], unsigned (&wm)[V]) const
    {
        if (!WOPD) r[0].c[OPD] = 0.;
        ZernOp<NMAX, CONIC>::template apply<V>(r, sel, wm);
    }
};

// ---- host: fold (coeff, rorder, aorder) into the (n,|m|) table ----------------------------
// `rot`: the set is evaluated at theta+rot (tracezernrot's second set, zernsurf.f95:302): by the
// angle-addition identities  Ac cos(m(t+rot)) + As sin(m(t+rot))  =
//   [Ac cos(m rot) + As sin(m rot)] cos(m t) + [As cos(m rot) - Ac sin(m rot)] sin(m t),
// so a rotated set folds into the SAME table and costs the kernel nothing.  `accumulate` adds to
// an already folded table.
inline int fold_table(const double *coeff, const int32_t *rorder, const int32_t *aorder, int32_t arrsize,
                      ZernTab &tab, int &nmax, double rot = 0., bool accumulate = false)
{
    if (arrsize < 0 || (arrsize > 0 && (!coeff || !rorder || !aorder))) return PXF_EINVAL;
    if (!accumulate) {
        for (auto &a : tab.a) a = make_double2(0., 0.);
        nmax = 0;
    }
    const double sqrt_half_f32 = (double)sqrtf(0.5f);  // specialFunctions.f95:225: REAL*4 sqrt(0.5)
    for (int i = 0; i < arrsize; i++) {
        const int n = rorder[i], mm = aorder[i], m = mm < 0 ? -mm : mm;
        if (n < 0 || m > n || ((n - m) & 1)) return PXF_EINVAL;  // the reference reads garbage here
        if (n > kZernMaxN) return PXF_ERANGE;
        if (n > nmax) nmax = n;
        const double norm = sqrt(2. * (n + 1));
        double2 &a = tab.a[zern_pair(n, m)];
        double ac = 0., as = 0.;
        if (mm < 0)
            as = coeff[i] * norm;
        else if (mm > 0)
            ac = coeff[i] * norm;
        else
            ac = coeff[i] * (norm * sqrt_half_f32);
        if (rot != 0. && m > 0) {
            const double cr = cos(m * rot), sr = sin(m * rot);
            const double ac2 = ac * cr + as * sr, as2 = as * cr - ac * sr;
            ac = ac2;
            as = as2;
        }
        a.x += ac;
        a.y += as;
    }
    return PXF_OK;
}

}  // namespace pxf
