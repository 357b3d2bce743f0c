// pxf_transforms.hip -- transform / itransform / reflect / radgrat[W] / applyT / pointTo kernels.
//
// Reference: transformationsf.f95 (native loops) and transformations.py (applyT, pointTo).
// sin/cos of the six frame parameters are evaluated ONCE on the host with glibc libm (the
// reference re-evaluates the same pure function 18 times per ray, transformationsf.f95:3-28)
// and passed by value; the per-ray work is then add/mul only, in the reference's order, so the
// result is bit-identical.
#include "pxf_ops_transforms.h"

namespace pxf {

// applyT / applyTPos : transformations.py:257-289.  out = M[:3,:4] . [v,1].  Out-of-place.
struct Mat34 {
    double a[12];
};

template <int V, int NVEC>
__global__ __launch_bounds__(kBlock, 2) void applyt_kernel(Cols in, Cols out, int64_t first, int64_t nvec,
                                                        Mat34 mpos, Mat34 mdir)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < nvec; v += stride) {
        const int64_t vv = first + v;
#pragma unroll
        for (int g = 0; g < NVEC; g++) {  // g = 0 position, 1 direction, 2 normal
            const Mat34 &mm = g == 0 ? mpos : mdir;
            double a[V], b[V], c[V], o0[V], o1[V], o2[V];
            bool w[V];
            load_col<V>(in.p[X + 3 * g], vv, a);
            load_col<V>(in.p[Y + 3 * g], vv, b);
            load_col<V>(in.p[Z + 3 * g], vv, c);
#pragma unroll
            for (int k = 0; k < V; k++) {
                o0[k] = mm.a[0] * a[k] + mm.a[1] * b[k] + mm.a[2] * c[k] + mm.a[3];
                o1[k] = mm.a[4] * a[k] + mm.a[5] * b[k] + mm.a[6] * c[k] + mm.a[7];
                o2[k] = mm.a[8] * a[k] + mm.a[9] * b[k] + mm.a[10] * c[k] + mm.a[11];
                w[k] = true;
            }
            store_col<V>(out.p[X + 3 * g], vv, o0, w);
            store_col<V>(out.p[Y + 3 * g], vv, o1, w);
            store_col<V>(out.p[Z + 3 * g], vv, o2, w);
        }
    }
}

static int launch_applyt(const Cols &in, const Cols &out, int ngroups, int64_t num, const double *mpos,
                         const double *mdir, pxf_stream_t stream)
{
    if (num < 0 || !mpos || (ngroups == 3 && !mdir)) return PXF_EINVAL;
    if (num == 0) return PXF_OK;
    bool aligned = true;
    for (int g = 0; g < 3 * ngroups; g++) {
        if (!in.p[X + g] || !out.p[X + g]) return PXF_EINVAL;
        if (((uintptr_t)in.p[X + g] | (uintptr_t)out.p[X + g]) & 15) aligned = false;
    }
    if (num == 0) return PXF_OK;
    Mat34 mp, md;
    for (int i = 0; i < 12; i++) {
        mp.a[i] = mpos[i];
        md.a[i] = mdir ? mdir[i] : 0.;
    }
    // transformations.py:268-271 stacks a row of ones under the direction vectors too, so the
    // (zero) translation column of the rotation-only matrix is added exactly as np.dot does.
    hipStream_t s = (hipStream_t)stream;
    const int64_t nvec2 = aligned ? num / 2 : 0;
    if (nvec2 > 0) {
        if (ngroups == 3)
            hipLaunchKernelGGL((applyt_kernel<2, 3>), dim3(grid_for(nvec2)), dim3(kBlock), 0, s, in, out, (int64_t)0, nvec2, mp, md);
        else
            hipLaunchKernelGGL((applyt_kernel<2, 1>), dim3(grid_for(nvec2)), dim3(kBlock), 0, s, in, out, (int64_t)0, nvec2, mp, md);
    }
    const int64_t done = nvec2 * 2;
    if (done < num) {
        if (ngroups == 3)
            hipLaunchKernelGGL((applyt_kernel<1, 3>), dim3(grid_for(num - done)), dim3(kBlock), 0, s, in, out, done, num - done, mp, md);
        else
            hipLaunchKernelGGL((applyt_kernel<1, 1>), dim3(grid_for(num - done)), dim3(kBlock), 0, s, in, out, done, num - done, mp, md);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PXF_OK : hip_fail(e);
}

static Cols cols_xyzlmnu(const double *x, const double *y, const double *z, const double *l,
                         const double *m, const double *n, const double *ux, const double *uy,
                         const double *uz)
{
    Cols c{};
    const double *src[9] = {x, y, z, l, m, n, ux, uy, uz};
    for (int i = 0; i < 9; i++) c.p[X + i] = const_cast<double *>(src[i]);
    return c;
}

}  // namespace pxf

using namespace pxf;

extern "C" {

int pxf_transform(double *x, double *y, double *z, double *l, double *m, double *n, double *ux,
                  double *uy, double *uz, int64_t num, double tx, double ty, double tz, double rx,
                  double ry, double rz, const uint8_t *mask, pxf_stream_t stream)
{
    double cs[6];
    trig6(rx, ry, rz, false, cs);
    TransformOp<false> op{tx, ty, tz, cs[0], cs[1], cs[2], cs[3], cs[4], cs[5], 7};
    return launch_rays(cols_xyzlmnu(x, y, z, l, m, n, ux, uy, uz), num, mask, stream, op);
}

int pxf_itransform(double *x, double *y, double *z, double *l, double *m, double *n, double *ux,
                   double *uy, double *uz, int64_t num, double tx, double ty, double tz, double rx,
                   double ry, double rz, const uint8_t *mask, pxf_stream_t stream)
{
    double cs[6];
    trig6(rx, ry, rz, true, cs);
    TransformOp<true> op{tx, ty, tz, cs[0], cs[1], cs[2], cs[3], cs[4], cs[5], 7};
    return launch_rays(cols_xyzlmnu(x, y, z, l, m, n, ux, uy, uz), num, mask, stream, op);
}

int pxf_reflect(double *l, double *m, double *n, const double *ux, const double *uy,
                const double *uz, int64_t num, const uint8_t *mask, pxf_stream_t stream)
{
    Cols c{};
    c.p[L] = l;
    c.p[M] = m;
    c.p[N] = n;
    c.p[UX] = const_cast<double *>(ux);
    c.p[UY] = const_cast<double *>(uy);
    c.p[UZ] = const_cast<double *>(uz);
    return launch_rays(c, num, mask, stream, ReflectOp{});
}

int pxf_radgrat(const double *x, const double *y, double *l, double *m, double *n, double wave,
                int64_t num, double dpermm, double order, const uint8_t *mask, pxf_stream_t stream)
{
    Cols c{};
    c.p[X] = const_cast<double *>(x);
    c.p[Y] = const_cast<double *>(y);
    c.p[L] = l;
    c.p[M] = m;
    c.p[N] = n;
    RadgratOp<false> op{dpermm, order, wave, 0., 0.};
    radgrat_consts(op.S, op.C);
    return launch_rays(c, num, mask, stream, op);
}

int pxf_radgratw(const double *x, const double *y, double *l, double *m, double *n,
                 const double *wave, int64_t num, double dpermm, double order, const uint8_t *mask,
                 pxf_stream_t stream)
{
    Cols c{};
    c.p[X] = const_cast<double *>(x);
    c.p[Y] = const_cast<double *>(y);
    c.p[L] = l;
    c.p[M] = m;
    c.p[N] = n;
    c.p[AUX] = const_cast<double *>(wave);
    RadgratOp<true> op{dpermm, order, 0., 0., 0.};
    radgrat_consts(op.S, op.C);
    return launch_rays(c, num, mask, stream, op);
}

int pxf_applyt(const double *x, const double *y, const double *z, const double *l, const double *m,
               const double *n, const double *ux, const double *uy, const double *uz, double *ox,
               double *oy, double *oz, double *ol, double *om, double *on, double *oux, double *ouy,
               double *ouz, int64_t num, const double *mpos, const double *mdir, pxf_stream_t stream)
{
    return launch_applyt(cols_xyzlmnu(x, y, z, l, m, n, ux, uy, uz),
                         cols_xyzlmnu(ox, oy, oz, ol, om, on, oux, ouy, ouz), 3, num, mpos, mdir, stream);
}

int pxf_applytpos(const double *x, const double *y, const double *z, double *ox, double *oy,
                  double *oz, int64_t num, const double *mpos, pxf_stream_t stream)
{
    Cols in{}, out{};
    in.p[X] = const_cast<double *>(x);
    in.p[Y] = const_cast<double *>(y);
    in.p[Z] = const_cast<double *>(z);
    out.p[X] = ox;
    out.p[Y] = oy;
    out.p[Z] = oz;
    return launch_applyt(in, out, 1, num, mpos, nullptr, stream);
}

int pxf_pointto(const double *x, const double *y, const double *z, double *l, double *m, double *n,
                int64_t num, double x0, double y0, double z0, double reverse, pxf_stream_t stream)
{
    Cols c{};
    c.p[X] = const_cast<double *>(x);
    c.p[Y] = const_cast<double *>(y);
    c.p[Z] = const_cast<double *>(z);
    c.p[L] = l;
    c.p[M] = m;
    c.p[N] = n;
    return launch_rays(c, num, nullptr, stream, PointToOp{x0, y0, z0, reverse});
}

int pxf_grat(double *l, double *m, double *n, int64_t num, double d, double order_times_wave,
             const double *order_times_wave_col, const uint8_t *mask, pxf_stream_t stream)
{
    Cols c{};
    c.p[L] = l;
    c.p[M] = m;
    c.p[N] = n;
    if (order_times_wave_col) {
        c.p[AUX] = const_cast<double *>(order_times_wave_col);
        return launch_rays(c, num, mask, stream, GratOp<true>{d, 0.});
    }
    return launch_rays(c, num, mask, stream, GratOp<false>{d, order_times_wave});
}

int pxf_refract(double *l, double *m, double *n, double *ux, double *uy, double *uz, int64_t num, double n1,
                double n2, const uint8_t *mask, pxf_stream_t stream)
{
    Cols c{};
    c.p[L] = l;
    c.p[M] = m;
    c.p[N] = n;
    c.p[UX] = ux;
    c.p[UY] = uy;
    c.p[UZ] = uz;
    return launch_rays(c, num, mask, stream, RefractOp{n1, n2});
}

}  // extern "C"
