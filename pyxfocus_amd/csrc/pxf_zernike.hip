// pxf_zernike.hip -- Newton trace to a Zernike sag surface: tracezern / tracezernOPD
// (zernsurf.f95:8-101 / 108-203) with zernset (specialFunctions.f95:142-232) inlined.
//
// What the reference does per Newton step and per ray: two heap allocations, pow(rho,n),
// atan2, and 3 x arrsize sin/cos(m*theta) calls (~3.1 us/ray for 36 terms on one CPU core).
// What this kernel does instead (MI355X-first, fp64 VALU-bound, no transcendentals at all):
//   * theta never materialises: cos(theta)=x/rho, sin(theta)=y/rho, and cos/sin(m*theta) come
//     from the angle-addition recurrence (m <= 15);
//   * rho'^n is a running product; the radial polynomials R_n^m, dR_n^m/drho' follow the
//     REFERENCE's own downward q-recursion (specialFunctions.f95:187-199), whose coefficients
//     h1,h2,h3 depend only on (n,m) and are compile-time constants of the fully unrolled body;
//   * the user's (coeff, rorder, aorder) list is folded ON THE HOST into one (Ac, As) pair per
//     (n,|m|) -- cos-type and sin-type amplitude including the sqrt(2(n+1)) normalisation (and
//     the reference's single-precision sqrt(0.5) for m=0, specialFunctions.f95:225) -- and that
//     table is staged in LDS once per workgroup; every lane reads the same address (broadcast);
//   * one ray per thread (VALU-bound: wide loads buy nothing, fewer VGPRs do).
// Parity is to a stated tolerance, not bitwise: summation order over terms, recurrences instead of
// libm, and FMA contraction (this TU only) all differ from the reference at the 1e-16 level, and
// the Newton iteration contracts those differences quadratically.  See DESIGN.md.
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

// Value and gradient of  S(x,y) = sum_i coeff_i Z_i(rho/rad, theta)  from the folded (Ac,As) table:
// returns S, and the cartesian gradient (dS/dx, dS/dy).  rho == 0 gives NaN, as the reference's
// division by rho does (zernsurf.f95:63-66).
template <int NMAX>
__device__ __forceinline__ void zern_eval(const double2 *__restrict__ tab, double x, double y, double rad,
                                          double inv_rad, double &S, double &Sx, double &Sy)
{
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
                const double h3 = zh3(nn, mm), h2 = zh2(nn, mm), h1 = zh1(nn, mm);
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
#ifndef PXF_ZERN_MIN_WAVES
#define PXF_ZERN_MIN_WAVES(NMAX) ((NMAX) <= 7 ? 3 : 1)
#endif
template <int NMAX, bool CONIC = false>
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
        __syncthreads();
    }

    template <int V>
    __device__ __forceinline__ void apply(Ray (&r)[V], const bool (&sel)[V], unsigned (&wm)[V]) const
    {
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
            zern_eval<NMAX>(tab, x, y, rad, inv_rad, S, Sx, Sy);
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

// zernphase : zernsurf.f95:206-250.  A thin Zernike PHASE plate in the z=0 plane: one evaluation
// (no Newton); l += dS/dx*wave, m += dS/dy*wave, n renormalised with its sign kept, opd += S*wave.
// R opd x y l m n   W opd l m n   = 80 B/ray
template <int NMAX>
struct ZernPhaseOp {
    static constexpr int MAXV = 1;
    static constexpr unsigned RD = B(OPD) | B(X) | B(Y) | DIR;
    static constexpr unsigned WR = B(OPD) | DIR;
    double2 tab_arg[zern_npairs(NMAX)];
    double rad, wave;
    __device__ __forceinline__ double2 *lds() const
    {
        __shared__ double2 s_tab[zern_npairs(NMAX)];
        return s_tab;
    }
    __device__ __forceinline__ void prologue() const
    {
        double2 *s = lds();
        for (int i = threadIdx.x; i < zern_npairs(NMAX); i += kBlock) s[i] = tab_arg[i];
        __syncthreads();
    }
    template <int V>
    __device__ __forceinline__ void apply(Ray (&r)[V], const bool (&sel)[V], unsigned (&wm)[V]) const
    {
        double *c = r[0].c;
        double S, Sx, Sy;
        zern_eval<NMAX>(lds(), c[X], c[Y], rad, 1. / rad, S, Sx, Sy);
        c[L] = c[L] + Sx * wave;
        c[M] = c[M] + Sy * wave;
        c[N] = copysign(sqrt(1. - c[L] * c[L] - c[M] * c[M]), c[N]);
        c[OPD] = c[OPD] + S * wave;
        wm[0] = WR;
    }
};

// ---- host: fold (coeff, rorder, aorder) into the (n,|m|) table ----------------------------
// `rot`: the set is evaluated at theta+rot (tracezernrot's second set, zernsurf.f95:302): by the
// angle-addition identities  Ac cos(m(t+rot)) + As sin(m(t+rot))  =
//   [Ac cos(m rot) + As sin(m rot)] cos(m t) + [As cos(m rot) - Ac sin(m rot)] sin(m t),
// so a rotated set folds into the SAME table and costs the kernel nothing.  `accumulate` adds to
// an already folded table.
static int fold_table(const double *coeff, const int32_t *rorder, const int32_t *aorder, int32_t arrsize,
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

template <int NMAX, bool WOPD, bool CONIC = false>
static int launch_nmax(const Cols &c, int64_t num, const uint8_t *mask, pxf_stream_t stream,
                       const ZernTab &tab, double rad, double nr, double cc = 0., double K1 = 0.)
{
    ZernOpSel<NMAX, WOPD, CONIC> op;
    op.cc = cc;
    op.K1 = K1;
    for (int i = 0; i < zern_npairs(NMAX); i++) op.tab_arg[i] = tab.a[i];
    op.rad = rad;
    op.nr = nr;
    op.has_opd = WOPD;
    return launch_rays(c, num, mask, stream, op);
}

template <bool WOPD, int NMAX = kZernMaxN>
static int dispatch_nmax(int nmax, const Cols &c, int64_t num, const uint8_t *mask, pxf_stream_t stream,
                         const ZernTab &tab, double rad, double nr)
{
    if constexpr (NMAX < 1) {
        return launch_nmax<1, WOPD>(c, num, mask, stream, tab, rad, nr);
    } else {
        if (nmax >= NMAX) return launch_nmax<NMAX, WOPD>(c, num, mask, stream, tab, rad, nr);
        return dispatch_nmax<WOPD, NMAX - 1>(nmax, c, num, mask, stream, tab, rad, nr);
    }
}

// the conic extension is instantiated for the even orders up to 10 only (compile time)
template <bool WOPD, int NMAX = 10>
static int dispatch_conic(int nmax, const Cols &c, int64_t num, const uint8_t *mask, pxf_stream_t stream,
                          const ZernTab &tab, double rad, double nr, double cc, double K1)
{
    if constexpr (NMAX < 2) {
        return launch_nmax<2, WOPD, true>(c, num, mask, stream, tab, rad, nr, cc, K1);
    } else {
        if (nmax > NMAX - 2) return launch_nmax<NMAX, WOPD, true>(c, num, mask, stream, tab, rad, nr, cc, K1);
        return dispatch_conic<WOPD, NMAX - 2>(nmax, c, num, mask, stream, tab, rad, nr, cc, K1);
    }
}

template <bool WOPD>
static int run_table(const ZernTab &tab, int nmax, double *opd, double *x, double *y, double *z, const double *l,
                     const double *m, const double *n, double *ux, double *uy, double *uz, int64_t num,
                     double rad, double nr, const uint8_t *mask, pxf_stream_t stream);

template <bool WOPD>
static int run(double *opd, double *x, double *y, double *z, const double *l, const double *m,
               const double *n, double *ux, double *uy, double *uz, int64_t num, const double *coeff,
               const int32_t *rorder, const int32_t *aorder, int32_t arrsize, double rad, double nr,
               const uint8_t *mask, pxf_stream_t stream)
{
    ZernTab tab;
    int nmax = 0;
    int rc = fold_table(coeff, rorder, aorder, arrsize, tab, nmax);
    if (rc != PXF_OK) return rc;
    return run_table<WOPD>(tab, nmax, opd, x, y, z, l, m, n, ux, uy, uz, num, rad, nr, mask, stream);
}

template <bool WOPD>
static int run_table(const ZernTab &tab, int nmax, double *opd, double *x, double *y, double *z, const double *l,
                     const double *m, const double *n, double *ux, double *uy, double *uz, int64_t num,
                     double rad, double nr, const uint8_t *mask, pxf_stream_t stream)
{
    if (num < 0) return PXF_EINVAL;
    if (num == 0) return PXF_OK;
    Cols c{};
    c.p[OPD] = opd;
    c.p[X] = x;
    c.p[Y] = y;
    c.p[Z] = z;
    c.p[L] = const_cast<double *>(l);
    c.p[M] = const_cast<double *>(m);
    c.p[N] = const_cast<double *>(n);
    c.p[UX] = ux;
    c.p[UY] = uy;
    c.p[UZ] = uz;
    // the folded table travels by value in the kernel arguments (<= 1152 B) and is staged into LDS
    // by each workgroup's prologue: no allocation, no copy, nothing to keep alive, graph-capturable
    return dispatch_nmax<WOPD>(nmax, c, num, mask, stream, tab, rad, nr);
}

template <int NMAX = kZernMaxN>
static int dispatch_phase(int nmax, const Cols &c, int64_t num, const uint8_t *mask, pxf_stream_t stream,
                          const ZernTab &tab, double rad, double wave)
{
    if constexpr (NMAX < 1) {
        return dispatch_phase<1>(nmax, c, num, mask, stream, tab, rad, wave);
    } else {
        if (nmax >= NMAX || NMAX == 1) {
            ZernPhaseOp<NMAX> op;
            for (int i = 0; i < zern_npairs(NMAX); i++) op.tab_arg[i] = tab.a[i];
            op.rad = rad;
            op.wave = wave;
            return launch_rays(c, num, mask, stream, op);
        }
        return dispatch_phase<NMAX - 1>(nmax, c, num, mask, stream, tab, rad, wave);
    }
}

}  // namespace pxf

using namespace pxf;

extern "C" {

int pxf_tracezern(double *x, double *y, double *z, const double *l, const double *m, const double *n,
                  double *ux, double *uy, double *uz, int64_t num, const double *coeff,
                  const int32_t *rorder, const int32_t *aorder, int32_t arrsize, double rad,
                  const uint8_t *mask, pxf_stream_t stream)
{
    return run<false>(nullptr, x, y, z, l, m, n, ux, uy, uz, num, coeff, rorder, aorder, arrsize, rad, 0.,
                      mask, stream);
}

int pxf_tracezernopd(double *opd, double *x, double *y, double *z, const double *l, const double *m,
                     const double *n, double *ux, double *uy, double *uz, int64_t num,
                     const double *coeff, const int32_t *rorder, const int32_t *aorder,
                     int32_t arrsize, double rad, double nr, const uint8_t *mask, pxf_stream_t stream)
{
    return run<true>(opd, x, y, z, l, m, n, ux, uy, uz, num, coeff, rorder, aorder, arrsize, rad, nr, mask,
                     stream);
}

int pxf_tracezernconic(double *opd, double *x, double *y, double *z, const double *l, const double *m,
                       const double *n, double *ux, double *uy, double *uz, int64_t num, double R, double K,
                       const double *coeff, const int32_t *rorder, const int32_t *aorder, int32_t arrsize,
                       double rad, double nr, const uint8_t *mask, pxf_stream_t stream)
{
    ZernTab tab;
    int nmax = 0;
    int rc = fold_table(coeff, rorder, aorder, arrsize, tab, nmax);
    if (rc != PXF_OK) return rc;
    if (nmax > 10) return PXF_ERANGE;
    if (num < 0 || R == 0.) return PXF_EINVAL;
    if (num == 0) return PXF_OK;
    Cols c{};
    c.p[OPD] = opd;
    c.p[X] = x;
    c.p[Y] = y;
    c.p[Z] = z;
    c.p[L] = const_cast<double *>(l);
    c.p[M] = const_cast<double *>(m);
    c.p[N] = const_cast<double *>(n);
    c.p[UX] = ux;
    c.p[UY] = uy;
    c.p[UZ] = uz;
    if (opd) return dispatch_conic<true>(nmax, c, num, mask, stream, tab, rad, nr, 1. / R, K + 1.);
    return dispatch_conic<false>(nmax, c, num, mask, stream, tab, rad, 0., 1. / R, K + 1.);
}

int pxf_tracezernrot(double *x, double *y, double *z, const double *l, const double *m, const double *n,
                     double *ux, double *uy, double *uz, int64_t num, const double *coeff1,
                     const int32_t *rorder1, const int32_t *aorder1, int32_t arrsize1, const double *coeff2,
                     const int32_t *rorder2, const int32_t *aorder2, int32_t arrsize2, double rad, double rot,
                     const uint8_t *mask, pxf_stream_t stream)
{
    ZernTab tab;
    int nmax = 0;
    int rc = fold_table(coeff1, rorder1, aorder1, arrsize1, tab, nmax);
    if (rc != PXF_OK) return rc;
    rc = fold_table(coeff2, rorder2, aorder2, arrsize2, tab, nmax, rot, true);
    if (rc != PXF_OK) return rc;
    return run_table<false>(tab, nmax, nullptr, x, y, z, l, m, n, ux, uy, uz, num, rad, 0., mask, stream);
}

int pxf_zernphase(double *opd, const double *x, const double *y, double *l, double *m, double *n, int64_t num,
                  const double *coeff, const int32_t *rorder, const int32_t *aorder, int32_t arrsize,
                  double rad, double wave, const uint8_t *mask, pxf_stream_t stream)
{
    ZernTab tab;
    int nmax = 0;
    int rc = fold_table(coeff, rorder, aorder, arrsize, tab, nmax);
    if (rc != PXF_OK) return rc;
    Cols c{};
    c.p[OPD] = opd;
    c.p[X] = const_cast<double *>(x);
    c.p[Y] = const_cast<double *>(y);
    c.p[L] = l;
    c.p[M] = m;
    c.p[N] = n;
    return dispatch_phase(nmax, c, num, mask, stream, tab, rad, wave);
}

}  // extern "C"
