// pxf_surfaces.hip -- flat, conic and Wolter-I primary/secondary intersect kernels.
//
// Each Op is the body of one reference Fortran loop, written in the reference's floating-point
// operation order (no FMA contraction) so the results are bit-identical to it; citations are
// file:line in the reference tree.  Per-surface scalars (Van Speybroeck p,d,e) are evaluated
// on the HOST with glibc libm, as the reference does once per call, and passed by value.
#include "pxf_ops_surfaces.h"

namespace pxf {

static Cols cols9(double *opd, double *x, double *y, double *z, const double *l, const double *m,
                  const double *n, double *ux, double *uy, double *uz)
{
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
    return c;
}

}  // namespace pxf

using namespace pxf;

extern "C" {

void pxf_wolter_params(double r0, double z0, double psi, double *p, double *d, double *e)
{
    // woltsurf.f95:18-23, host libm, reference operation order
    const double alpha = .25 * atan(r0 / z0);
    const double thetah = 2 * (1 + 2 * psi) / (1 + psi) * alpha;
    const double thetap = 2 * psi / (1 + psi) * alpha;
    *p = z0 * tan(4 * alpha) * tan(thetap);
    *d = z0 * tan(4 * alpha) * tan(4 * alpha - thetah);
    *e = cos(4 * alpha) * (1 + tan(4 * alpha) * tan(thetah));
}

int pxf_flat(double *x, double *y, double *z, const double *l, const double *m, const double *n,
             double *ux, double *uy, double *uz, int64_t num, const uint8_t *mask, pxf_stream_t stream)
{
    return launch_rays(cols9(nullptr, x, y, z, l, m, n, ux, uy, uz), num, mask, stream, FlatOp<false>{0.});
}

int pxf_flatopd(double *x, double *y, double *z, const double *l, const double *m, const double *n,
                double *ux, double *uy, double *uz, double *opd, int64_t num, double nr,
                const uint8_t *mask, pxf_stream_t stream)
{
    return launch_rays(cols9(opd, x, y, z, l, m, n, ux, uy, uz), num, mask, stream, FlatOp<true>{nr});
}

int pxf_conic(double *x, double *y, double *z, double *l, double *m, double *n, double *ux,
              double *uy, double *uz, int64_t num, double R, double K, const uint8_t *mask,
              pxf_stream_t stream)
{
    return launch_rays(cols9(nullptr, x, y, z, l, m, n, ux, uy, uz), num, mask, stream,
                       ConicOp<false>{R, K, 0.});
}

int pxf_conicopd(double *opd, double *x, double *y, double *z, double *l, double *m, double *n,
                 double *ux, double *uy, double *uz, int64_t num, double R, double K, double nr,
                 const uint8_t *mask, pxf_stream_t stream)
{
    return launch_rays(cols9(opd, x, y, z, l, m, n, ux, uy, uz), num, mask, stream, ConicOp<true>{R, K, nr});
}

int pxf_wolterprimary(double *x, double *y, double *z, const double *l, const double *m,
                      const double *n, double *ux, double *uy, double *uz, int64_t num, double r0,
                      double z0, double psi, const uint8_t *mask, pxf_stream_t stream)
{
    const WolterPrimaryOp<false> op = make_wolter_primary<false>(r0, z0, psi, 0.);
    return launch_rays(cols9(nullptr, x, y, z, l, m, n, ux, uy, uz), num, mask, stream, op);
}

int pxf_wolterprimaryopd(double *opd, double *x, double *y, double *z, const double *l,
                         const double *m, const double *n, double *ux, double *uy, double *uz,
                         int64_t num, double r0, double z0, double psi, double nr,
                         const uint8_t *mask, pxf_stream_t stream)
{
    const WolterPrimaryOp<true> op = make_wolter_primary<true>(r0, z0, psi, nr);
    return launch_rays(cols9(opd, x, y, z, l, m, n, ux, uy, uz), num, mask, stream, op);
}

int pxf_woltersecondary(double *x, double *y, double *z, const double *l, const double *m,
                        const double *n, double *ux, double *uy, double *uz, int64_t num, double r0,
                        double z0, double psi, const uint8_t *mask, pxf_stream_t stream)
{
    const WolterSecondaryOp op = make_wolter_secondary(r0, z0, psi);
    return launch_rays(cols9(nullptr, x, y, z, l, m, n, ux, uy, uz), num, mask, stream, op);
}

int pxf_spocone(double *x, double *y, double *z, double *l, double *m, double *n, double *ux, double *uy,
                double *uz, int64_t num, double R0, double tg, const uint8_t *mask, pxf_stream_t stream)
{
    double ctg, stg;
    host_sincos(tg, ctg, stg);   // woltsurf.f95:591-638 compiles to one `sincos` + one `tan` (pxf_ops_transforms.h)
    return launch_rays(cols9(nullptr, x, y, z, l, m, n, ux, uy, uz), num, mask, stream,
                       SpoConeOp{R0, tan(tg), ctg, stg});
}

int pxf_tracesphere(double *x, double *y, double *z, double *l, double *m, double *n, double *ux,
                    double *uy, double *uz, int64_t num, double rad, const uint8_t *mask, pxf_stream_t stream)
{
    return launch_rays(cols9(nullptr, x, y, z, l, m, n, ux, uy, uz), num, mask, stream,
                       SphereCylOp<false, false>{rad, 0.});
}

int pxf_tracesphereopd(double *opd, double *x, double *y, double *z, double *l, double *m, double *n,
                       double *ux, double *uy, double *uz, int64_t num, double rad, double nr,
                       const uint8_t *mask, pxf_stream_t stream)
{
    return launch_rays(cols9(opd, x, y, z, l, m, n, ux, uy, uz), num, mask, stream,
                       SphereCylOp<true, false>{rad, nr});
}

int pxf_tracecyl(double *x, double *y, double *z, double *l, double *m, double *n, double *ux, double *uy,
                 double *uz, int64_t num, double rad, const uint8_t *mask, pxf_stream_t stream)
{
    return launch_rays(cols9(nullptr, x, y, z, l, m, n, ux, uy, uz), num, mask, stream,
                       SphereCylOp<false, true>{rad, 0.});
}

int pxf_tracecylopd(double *opd, double *x, double *y, double *z, double *l, double *m, double *n,
                    double *ux, double *uy, double *uz, int64_t num, double rad, double nr,
                    const uint8_t *mask, pxf_stream_t stream)
{
    return launch_rays(cols9(opd, x, y, z, l, m, n, ux, uy, uz), num, mask, stream,
                       SphereCylOp<true, true>{rad, nr});
}

int pxf_cylconic(double *x, double *y, double *z, const double *l, const double *m, const double *n,
                 double *ux, double *uy, double *uz, int64_t num, double rad, double k,
                 const uint8_t *mask, pxf_stream_t stream)
{
    return launch_rays(cols9(nullptr, x, y, z, l, m, n, ux, uy, uz), num, mask, stream,
                       CylConicOp{rad, (1 + k) * (rad * rad)});
}

int pxf_paraxial(const double *x, const double *y, double *l, double *m, int64_t num, double F,
                 const uint8_t *mask, pxf_stream_t stream)
{
    Cols c{};
    c.p[X] = const_cast<double *>(x);
    c.p[Y] = const_cast<double *>(y);
    c.p[L] = l;
    c.p[M] = m;
    return launch_rays(c, num, mask, stream, ParaxialOp<false>{F});
}

int pxf_paraxialy(const double *y, double *m, int64_t num, double F, const uint8_t *mask, pxf_stream_t stream)
{
    Cols c{};
    c.p[Y] = const_cast<double *>(y);
    c.p[M] = m;
    return launch_rays(c, num, mask, stream, ParaxialOp<true>{F});
}

}  // extern "C"
