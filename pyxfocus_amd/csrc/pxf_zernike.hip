// pxf_zernike.hip -- Newton trace to a Zernike sag surface: tracezern / tracezernOPD
// (zernsurf.f95:8-101 / 108-203) with zernset (specialFunctions.f95:142-232) replaced by a polynomial.
//
// What the reference does per Newton step and per ray: two heap allocations, pow(rho,n), atan2, and
// 3 x arrsize sin/cos(m*theta) calls (~3.1 us/ray for 36 terms on one CPU core).
// What these kernels do instead (pxf_ops_zernike.h): the whole Zernike sum is one bivariate polynomial in
// (x/rad, y/rad); the host expands the caller's (coeff, rorder, aorder) into its monomial coefficients,
// the table is staged in LDS once per workgroup, and every evaluation of S, dS/dx, dS/dy is a nested
// Horner scheme -- 80 FMAs for 36 terms, no transcendental, no divide.  Two rays per thread with 16-byte
// accesses like every other streaming kernel; HBM-bound up to radial order 8.
// Parity is to a stated tolerance, not bitwise (different arithmetic altogether; the Newton iteration
// contracts the differences quadratically): 2e-11 mm asserted, ~1e-14 observed.  See DESIGN.md.
#include <math.h>

#include "pxf_kernel.h"
#include "pxf_ops_zernike.h"

namespace pxf {

// zernphase : zernsurf.f95:206-250.  A thin Zernike PHASE plate in the z=0 plane: one evaluation
// (no Newton); l += dS/dx*wave, m += dS/dy*wave, n renormalised with its sign kept, opd += S*wave.
// R opd x y l m n   W opd l m n   = 80 B/ray
template <int NMAX>
struct ZernPhaseOp {
    static constexpr int MAXV = 1;
    static constexpr int MIN_WAVES = 2;  // stay within 256 registers: see PXF_ZERN_MIN_WAVES
    static constexpr unsigned RD = B(OPD) | B(X) | B(Y) | DIR;
    static constexpr unsigned WR = B(OPD) | DIR;
    double tab_arg[zern_nmono(NMAX)];
    double rad, wave;
    __device__ __forceinline__ double *lds() const
    {
        __shared__ double s_tab[zern_nmono(NMAX)];
        return s_tab;
    }
    __device__ __forceinline__ void prologue() const
    {
        double *s = lds();
        for (int i = threadIdx.x; i < zern_nmono(NMAX); i += kBlock) s[i] = tab_arg[i];
        __syncthreads();
    }
    template <int V>
    __device__ __forceinline__ void apply(Ray (&r)[V], const bool (&sel)[V], unsigned (&wm)[V]) const
    {
        double *c = r[0].c;
        double S, Sx, Sy;
        zern_eval<NMAX>(lds(), c[X], c[Y], 1. / rad, S, Sx, Sy);
        c[L] = c[L] + Sx * wave;
        c[M] = c[M] + Sy * wave;
        c[N] = copysign(sqrt(1. - c[L] * c[L] - c[M] * c[M]), c[N]);
        c[OPD] = c[OPD] + S * wave;
        wm[0] = WR;
    }
};

template <int NMAX, bool WOPD, bool CONIC = false>
static int launch_nmax(const Cols &c, int64_t num, const uint8_t *mask, pxf_stream_t stream,
                       const ZernTab &tab, double rad, double nr, double cc = 0., double K1 = 0.)
{
    ZernOpSel<NMAX, WOPD, CONIC> op;
    op.cc = cc;
    op.K1 = K1;
    zern_fill_table<NMAX>(op.tab_arg, tab);
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
            zern_fill_table<NMAX>(op.tab_arg, tab);
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

int pxf_zernike_monomials(const double *coeff, const int32_t *rorder, const int32_t *aorder, int32_t arrsize,
                          int32_t degree, double *out)
{
    ZernTab tab;
    int nmax = 0;
    int rc = fold_table(coeff, rorder, aorder, arrsize, tab, nmax);
    if (rc != PXF_OK) return rc;
    if (!out || degree < nmax || degree > kZernMaxN) return PXF_EINVAL;
    zern_monomials(tab, degree, out);
    return PXF_OK;
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
