// pxf_wolter2.hip -- widening batches 2-3: the stand-alone entries of woltersine,
// the Legendre-Legendre deformed mirrors and the Wolter-Schwarzschild mirrors (Ops in pxf_ops_mirrors.h),
// plus torus, conicplus[opd] and legSurf (surfacesf.f95:468-668).
#include <math.h>
#include <stdlib.h>

#include "pxf_ops_mirrors.h"

namespace pxf {

// x**k with a RUN-TIME integer exponent: binary exponentiation, the sequence compiler-rt's
// __powidf2 (called by the flang-built reference) performs -- bit-identical products.
__device__ __forceinline__ double powi_ref(double a, int b)
{
    const bool recip = b < 0;
    double r = 1;
    while (true) {
        if (b & 1) r *= a;
        b /= 2;
        if (b == 0) break;
        a *= a;
    }
    return recip ? 1 / r : r;
}

// ---- torus : surfacesf.f95:468-507 (tangent-plane frame; Newton; |grad| WITHOUT Fz, :501) -------
struct TorusOp {
    PXF_OP_DEFAULTS
    static constexpr unsigned RD = POS | DIR;
    static constexpr unsigned WR = POS | NRM;
    double rin, rout;
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
                const double zr = z + rin + rout;
                const double t = zr * zr + y * y + x * x + rout * rout - rin * rin;
                const double F = t * t - (4 * (rout * rout) * (y * y + zr * zr));
                const double zr2 = rin + rout + z;
                Fx = 4 * x * (-(rin * rin) + zr2 * zr2 + rout * rout + x * x + y * y);
                const double q = 2 * rin * (rout + z) + 2 * rout * z + z * z + y * y + x * x;
                Fy = 4 * y * q;
                Fz = 4 * (rout + rin + z) * q;
                const double Fp = Fx * c[L] + Fy * c[M] + Fz * c[N];
                const double delt = -F / Fp;
                c[X] = x + c[L] * delt;
                c[Y] = y + c[M] * delt;
                c[Z] = z + c[N] * delt;
                act = fabs(delt) > PXF_TOL_1EM10;
            }
            const double Fp = sqrt(Fx * Fx + Fy * Fy);
            c[UX] = Fx / Fp;
            c[UY] = Fy / Fp;
            c[UZ] = Fz / Fp;
            wm[k] = WR;
        }
    }
};

// ---- conicplus / conicplusopd : surfacesf.f95:514-638 (conic + even polynomial, Newton) -----------
constexpr int kConicPlusMaxTerms = 16;
template <bool WOPD>
struct ConicPlusOp {
    PXF_OP_DEFAULTS
    static constexpr unsigned RD = POS | DIR | (WOPD ? B(OPD) : 0u);
    static constexpr unsigned WR = POS | NRM | (WOPD ? B(OPD) : 0u);
    double cc, K1, nr;  // c = 1/R, K+1
    int np;
    double p[kConicPlusMaxTerms];
    template <int V>
    __device__ __forceinline__ void apply(Ray (&r)[V], const bool (&sel)[V], unsigned (&wm)[V]) const
    {
#pragma unroll
        for (int k = 0; k < V; k++) {
            double *c = r[k].c;
            double Fx = 0., Fy = 0.;
            const double Fz = 1.;
            bool act = sel[k];
            for (int it = 0; act && it < kNewtonCap; it++) {
                const double x = c[X], y = c[Y], z = c[Z];
                const double rad = sqrt(x * x + y * y);
                double a0 = 0., a1 = 0.;
                for (int j = 1; j <= np; j++) {
                    a0 = a0 + p[j - 1] * powi_ref(rad, 2 * j);
                    a1 = a1 + p[j - 1] * (2 * j) * powi_ref(rad, 2 * j - 1);
                }
                const double sq = sqrt(1 - K1 * (cc * cc) * (rad * rad));
                const double denom = sq + 1;
                const double F = z - cc * (rad * rad) / denom + a0;
                const double Fr = -(2 * cc * rad / denom + (K1 * ((rad * rad) * rad) * ((cc * cc) * cc)) / ((denom * denom) * sq)) + a1;
                Fx = Fr * x / rad;
                Fy = Fr * y / rad;
                const double Fp = Fx * c[L] + Fy * c[M] + Fz * c[N];
                const double delt = -F / Fp;
                c[X] = x + c[L] * delt;
                c[Y] = y + c[M] * delt;
                c[Z] = z + c[N] * delt;
                if (WOPD) c[OPD] = c[OPD] + delt * nr;
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

// ---- legSurf : surfacesf.f95:642-668 (2-D Legendre PHASE surface; no propagation) -----------------
// legendre()/legendrep() (specialFunctions.f95:337-388) are explicit sums
//   sum_i K(n,i) x^(n-2i),  K(n,i) = (-1)^i (2n-2i)!/i!/(n-i)!/(n-2i)!/2^n ;
// K(n,i) is evaluated on the HOST in the reference's operation order and the sum on the device in
// the same order with the same integer-power routine, so the result is bit-identical.
constexpr int kLegMaxOrder = 12, kLegMaxTerms = 48;
struct LegSurfOp {
    static constexpr int MAXV = 1;
    static constexpr unsigned RD = B(X) | B(Y) | DIR;
    static constexpr unsigned WR = DIR;
    double xwidth, ywidth, order;
    int nc;
    double K[kLegMaxOrder + 1][kLegMaxOrder / 2 + 1];
    double coeff[kLegMaxTerms];
    signed char xo[kLegMaxTerms], yo[kLegMaxTerms];
    __device__ __forceinline__ void prologue() const {}
    __device__ __forceinline__ double leg(double x, int n) const
    {
        const double x2 = fabs(x) > 1. ? x / fabs(x) : x;
        if (n == 0) return 1.;
        double s = 0.;
        for (int i = 0; i <= n / 2; i++) s = s + K[n][i] * powi_ref(x2, n - 2 * i);
        return s;
    }
    __device__ __forceinline__ double legp(double x, int n) const
    {
        double s = 0.;
        if (n == 0)
            s = 0.;
        else if (n == 1)
            s = 1.;
        else if (x == 0. && n % 2 == 0)
            s = 0.;
        else
            for (int i = 0; i <= n / 2; i++) s = s + K[n][i] * (n - 2 * i) * powi_ref(x, n - 2 * i - 1);
        if (fabs(x) > 1.) s = 0.;
        return s;
    }
    template <int V>
    __device__ __forceinline__ void apply(Ray (&r)[V], const bool (&sel)[V], unsigned (&wm)[V]) const
    {
        double *c = r[0].c;
        const double xa = c[X] / xwidth, ya = c[Y] / ywidth;
        double dphidx = 0, dphidy = 0;
        for (int j = 0; j < nc; j++) {
            dphidx = dphidx + coeff[j] * leg(ya, yo[j]) * legp(xa, xo[j]);
            dphidy = dphidy + coeff[j] * legp(ya, yo[j]) * leg(xa, xo[j]);
        }
        c[L] = c[L] + dphidx * order / xwidth;
        c[M] = c[M] + dphidy * order / ywidth;
        c[N] = c[N] / fabs(c[N]) * sqrt(1. - c[L] * c[L] - c[M] * c[M]);
        wm[0] = WR;
    }
};

static Cols cols9(double *x, double *y, double *z, const double *l, const double *m, const double *n,
                  double *ux, double *uy, double *uz)
{
    Cols c{};
    c.p[X] = x;
    c.p[Y] = y;
    c.p[Z] = z;
    c.p[L] = const_cast<double *>(l);
    c.p[M] = const_cast<double *>(m);
    c.p[N] = const_cast<double *>(n);
    c.p[UX] = ux;
    c.p[UY] = uy;
    c.p[UZ] = uz;
    return c;
}

template <bool WOPD>
static int conicplus_run(double *opd, double *x, double *y, double *z, const double *l, const double *m,
                         const double *n, double *ux, double *uy, double *uz, int64_t num, double R, double K,
                         const double *p, int32_t Np, double nr, const uint8_t *mask, pxf_stream_t stream)
{
    if (Np < 0 || (Np > 0 && !p)) return PXF_EINVAL;
    if (Np > kConicPlusMaxTerms) return PXF_ERANGE;
    ConicPlusOp<WOPD> op{};
    op.cc = 1 / R;
    op.K1 = K + 1;
    op.nr = nr;
    op.np = Np;
    for (int j = 0; j < Np; j++) op.p[j] = p[j];
    Cols c = cols9(x, y, z, l, m, n, ux, uy, uz);
    c.p[OPD] = opd;
    return launch_rays(c, num, mask, stream, op);
}

}  // namespace pxf

using namespace pxf;

extern "C" {

int pxf_woltersine(double *x, double *y, double *z, const double *l, const double *m, const double *n,
                   double *ux, double *uy, double *uz, int64_t num, double r0, double z0, double amp,
                   double freq, const uint8_t *mask, pxf_stream_t stream)
{
    const WolterSineOp op = make_woltersine(r0, z0, amp, freq);
    return launch_rays(cols9(x, y, z, l, m, n, ux, uy, uz), num, mask, stream, op);
}

int pxf_wolterprimll(double *x, double *y, double *z, const double *l, const double *m, const double *n,
                     double *ux, double *uy, double *uz, int64_t num, double r0, double z0, double zmax,
                     double zmin, double dphi, const double *coeff, const int32_t *axial, const int32_t *az,
                     int32_t cnum, const uint8_t *mask, pxf_stream_t stream)
{
    LLData op{};
    int rc = fold_ll(op, zmax, zmin, dphi, coeff, axial, az, cnum);
    if (rc != PXF_OK) return rc;
    ll_setup_primary(op, r0, z0);
    return ll_launch<0>(op, cols9(x, y, z, l, m, n, ux, uy, uz), num, mask, stream);
}

int pxf_woltersecll(double *x, double *y, double *z, const double *l, const double *m, const double *n,
                    double *ux, double *uy, double *uz, int64_t num, double r0, double z0, double psi,
                    double zmax, double zmin, double dphi, const double *coeff, const int32_t *axial,
                    const int32_t *az, int32_t cnum, const uint8_t *mask, pxf_stream_t stream)
{
    LLData op{};
    int rc = fold_ll(op, zmax, zmin, dphi, coeff, axial, az, cnum);
    if (rc != PXF_OK) return rc;
    ll_setup_secondary(op, r0, z0, psi);
    return ll_launch<1>(op, cols9(x, y, z, l, m, n, ux, uy, uz), num, mask, stream);
}

int pxf_ellipsoidwoltll(double *x, double *y, double *z, const double *l, const double *m, const double *n,
                        double *ux, double *uy, double *uz, int64_t num, double r0, double z0, double psi,
                        double S, double zmax, double zmin, double dphi, const double *coeff,
                        const int32_t *axial, const int32_t *az, int32_t cnum, const uint8_t *mask,
                        pxf_stream_t stream)
{
    LLData op{};
    int rc = fold_ll(op, zmax, zmin, dphi, coeff, axial, az, cnum);
    if (rc != PXF_OK) return rc;
    ll_setup_ellipsoid(op, r0, z0, psi, S);
    return ll_launch<2>(op, cols9(x, y, z, l, m, n, ux, uy, uz), num, mask, stream);
}

int pxf_wsprimaryback(double *x, double *y, double *z, const double *l, const double *m, const double *n,
                      double *ux, double *uy, double *uz, int64_t num, double alpha, double z0, double psi,
                      double thick, const uint8_t *mask, pxf_stream_t stream)
{
    WSOp<false, true> op{};
    chase(alpha, z0, psi, op);
    op.thick = thick;
    return launch_rays(cols9(x, y, z, l, m, n, ux, uy, uz), num, mask, stream, op);
}

int pxf_wssecondaryback(double *x, double *y, double *z, const double *l, const double *m, const double *n,
                        double *ux, double *uy, double *uz, int64_t num, double alpha, double z0, double psi,
                        double thick, const uint8_t *mask, pxf_stream_t stream)
{
    WSOp<true, true> op{};
    chase(alpha, z0, psi, op);
    op.thick = thick;
    return launch_rays(cols9(x, y, z, l, m, n, ux, uy, uz), num, mask, stream, op);
}

int pxf_torus(double *x, double *y, double *z, const double *l, const double *m, const double *n, double *ux,
              double *uy, double *uz, int64_t num, double rin, double rout, const uint8_t *mask,
              pxf_stream_t stream)
{
    return launch_rays(cols9(x, y, z, l, m, n, ux, uy, uz), num, mask, stream, TorusOp{rin, rout});
}

int pxf_conicplus(double *x, double *y, double *z, const double *l, const double *m, const double *n,
                  double *ux, double *uy, double *uz, int64_t num, double R, double K, const double *p,
                  int32_t Np, const uint8_t *mask, pxf_stream_t stream)
{
    return conicplus_run<false>(nullptr, x, y, z, l, m, n, ux, uy, uz, num, R, K, p, Np, 0., mask, stream);
}

int pxf_conicplusopd(double *opd, double *x, double *y, double *z, const double *l, const double *m,
                     const double *n, double *ux, double *uy, double *uz, int64_t num, double R, double K,
                     const double *p, int32_t Np, double nr, const uint8_t *mask, pxf_stream_t stream)
{
    return conicplus_run<true>(opd, x, y, z, l, m, n, ux, uy, uz, num, R, K, p, Np, nr, mask, stream);
}

int pxf_legsurf(const double *x, const double *y, double *l, double *m, double *n, int64_t num, double xwidth,
                double ywidth, double order, const double *coeff, const int32_t *xo, const int32_t *yo,
                int32_t Nc, const uint8_t *mask, pxf_stream_t stream)
{
    if (Nc < 0 || (Nc > 0 && (!coeff || !xo || !yo))) return PXF_EINVAL;
    if (Nc > kLegMaxTerms) return PXF_ERANGE;
    LegSurfOp op{};
    op.xwidth = xwidth;
    op.ywidth = ywidth;
    op.order = order;
    op.nc = Nc;
    for (int j = 0; j < Nc; j++) {
        if (xo[j] < 0 || yo[j] < 0) return PXF_EINVAL;
        if (xo[j] > kLegMaxOrder || yo[j] > kLegMaxOrder) return PXF_ERANGE;
        op.coeff[j] = coeff[j];
        op.xo[j] = (signed char)xo[j];
        op.yo[j] = (signed char)yo[j];
    }
    // K(n,i) in the reference's order: (-1)**i * f(2n-2i) / f(i) / f(n-i) / f(n-2i) / 2**n
    auto fact = [](int q) {
        double a = 1;
        for (int i = 1; i <= q; i++) a = a * (double)i;
        return a;
    };
    for (int nn = 0; nn <= kLegMaxOrder; nn++)
        for (int i = 0; i <= nn / 2; i++)
            op.K[nn][i] = (double)((i & 1) ? -1 : 1) * fact(2 * nn - 2 * i) / fact(i) / fact(nn - i) / fact(nn - 2 * i) /
                          (double)(1 << nn);
    Cols c{};
    c.p[X] = const_cast<double *>(x);
    c.p[Y] = const_cast<double *>(y);
    c.p[L] = l;
    c.p[M] = m;
    c.p[N] = n;
    return launch_rays(c, num, mask, stream, op);
}

int pxf_host_cos_atan2(const double *r, const double *z, double *beta, double *cosbeta, double *tanhalf,
                       int64_t num)
{
    if (num < 0 || (num > 0 && (!r || !z || !beta || !cosbeta || !tanhalf))) return PXF_EINVAL;
    for (int64_t i = 0; i < num; i++) {
        if (!libm_cos_atan2(r[i], z[i], h_sincos_tab, h_tan_tab, beta[i], cosbeta[i])) beta[i] = cosbeta[i] = NAN;
        const double hb = .5 * beta[i];
        tanhalf[i] = (hb > 2e-8 && hb < 0.0607) ? libm_tan_small(hb) : NAN;
    }
    return PXF_OK;
}

int pxf_host_sincos_atan2(const double *y, const double *x, double *theta, double *sintheta, double *costheta,
                          int64_t num)
{
    if (num < 0 || (num > 0 && (!y || !x || !theta || !sintheta || !costheta))) return PXF_EINVAL;
    for (int64_t i = 0; i < num; i++)
        if (!libm_sincos_atan2(y[i], x[i], h_sincos_tab, h_tan_tab, theta[i], sintheta[i], costheta[i]))
            theta[i] = sintheta[i] = costheta[i] = NAN;
    return PXF_OK;
}

int pxf_wsprimary(double *x, double *y, double *z, const double *l, const double *m, const double *n,
                  double *ux, double *uy, double *uz, int64_t num, double alpha, double z0, double psi,
                  const uint8_t *mask, pxf_stream_t stream)
{
    WSOp<false> op{};
    chase(alpha, z0, psi, op);
    return launch_rays(cols9(x, y, z, l, m, n, ux, uy, uz), num, mask, stream, op);
}

int pxf_wssecondary(double *x, double *y, double *z, const double *l, const double *m, const double *n,
                    double *ux, double *uy, double *uz, int64_t num, double alpha, double z0, double psi,
                    const uint8_t *mask, pxf_stream_t stream)
{
    WSOp<true> op{};
    chase(alpha, z0, psi, op);
    return launch_rays(cols9(x, y, z, l, m, n, ux, uy, uz), num, mask, stream, op);
}

}  // extern "C"
