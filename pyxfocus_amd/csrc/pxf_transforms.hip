// pxf_transforms.hip -- transform / itransform / reflect / radgrat[W] / applyT / pointTo kernels.
//
// Reference: transformationsf.f95 (native loops) and transformations.py (applyT, pointTo).
// sin/cos of the six frame parameters are evaluated ONCE on the host with glibc libm (the
// reference re-evaluates the same pure function 18 times per ray, transformationsf.f95:3-28)
// and passed by value; the per-ray work is then add/mul only, in the reference's order, so the
// result is bit-identical.
#include <math.h>

#include "pxf_kernel.h"

namespace pxf {

// rotatevector, transformationsf.f95:3-28 (axis 1,2,3 = x,y,z), with c=cos(theta), s=sin(theta)
template <int AXIS>
__device__ __forceinline__ void rotv(double &x, double &y, double &z, double c, double s)
{
    if (AXIS == 1) {
        const double o2 = c * y - s * z, o3 = s * y + c * z;
        y = o2;
        z = o3;
    } else if (AXIS == 2) {
        const double o1 = c * x + s * z, o3 = -s * x + c * z;
        x = o1;
        z = o3;
    } else {
        const double o1 = c * x - s * y, o2 = s * x + c * y;
        x = o1;
        y = o2;
    }
}

// transform : transformationsf.f95:134-163 ; itransform : :168-201.   R 9 + W 9 = 144 B/ray
template <bool INVERSE>
struct TransformOp {
    PXF_OP_DEFAULTS
    static constexpr unsigned RD = POS | DIR | NRM;
    static constexpr unsigned WR = POS | DIR | NRM;
    double tx, ty, tz;
    double cx, sx, cy, sy, cz, sz;  // forward: cos/sin(rx..rz); inverse: cos/sin(-rx..-rz)
    template <int V>
    __device__ __forceinline__ void apply(Ray (&r)[V], const bool (&sel)[V], unsigned (&wm)[V]) const
    {
#pragma unroll
        for (int k = 0; k < V; k++) {
            double *c = r[k].c;
            if (!INVERSE) {
                c[X] = c[X] + tx;
                c[Y] = c[Y] + ty;
                c[Z] = c[Z] + tz;
                rotv<1>(c[X], c[Y], c[Z], cx, sx);
                rotv<1>(c[L], c[M], c[N], cx, sx);
                rotv<1>(c[UX], c[UY], c[UZ], cx, sx);
                rotv<2>(c[X], c[Y], c[Z], cy, sy);
                rotv<2>(c[L], c[M], c[N], cy, sy);
                rotv<2>(c[UX], c[UY], c[UZ], cy, sy);
                rotv<3>(c[X], c[Y], c[Z], cz, sz);
                rotv<3>(c[L], c[M], c[N], cz, sz);
                rotv<3>(c[UX], c[UY], c[UZ], cz, sz);
            } else {
                rotv<3>(c[X], c[Y], c[Z], cz, sz);
                rotv<3>(c[L], c[M], c[N], cz, sz);
                rotv<3>(c[UX], c[UY], c[UZ], cz, sz);
                rotv<2>(c[X], c[Y], c[Z], cy, sy);
                rotv<2>(c[L], c[M], c[N], cy, sy);
                rotv<2>(c[UX], c[UY], c[UZ], cy, sy);
                rotv<1>(c[X], c[Y], c[Z], cx, sx);
                rotv<1>(c[L], c[M], c[N], cx, sx);
                rotv<1>(c[UX], c[UY], c[UZ], cx, sx);
                c[X] = c[X] - tx;
                c[Y] = c[Y] - ty;
                c[Z] = c[Z] - tz;
            }
            wm[k] = WR;
        }
    }
};

// reflect : transformationsf.f95:60-79.   R 6 + W 3 = 72 B/ray
struct ReflectOp {
    PXF_OP_DEFAULTS
    static constexpr unsigned RD = DIR | NRM;
    static constexpr unsigned WR = DIR;
    template <int V>
    __device__ __forceinline__ void apply(Ray (&r)[V], const bool (&sel)[V], unsigned (&wm)[V]) const
    {
#pragma unroll
        for (int k = 0; k < V; k++) {
            double *c = r[k].c;
            const double dot = c[UX] * c[L] + c[UY] * c[M] + c[UZ] * c[N];
            c[L] = c[L] - 2 * dot * c[UX];
            c[M] = c[M] - 2 * dot * c[UY];
            c[N] = c[N] - 2 * dot * c[UZ];
            wm[k] = WR;
        }
    }
};

// radgrat / radgratW : transformationsf.f95:205-238 / 242-272.   R x y l m n (+wave) W l m n
// = 64 B/ray (72 with a per-ray wavelength column, carried in the AUX slot).
//
// The reference computes  yaw = -pi32/2 - atan2(x,y)  (pi32 = acos(-1.) in SINGLE precision)
// and then sin(yaw), cos(yaw).  With phi = atan2(x,y): sin(phi) = x/rho, cos(phi) = y/rho, so
//     sin(yaw) = -( S*cos(phi) + C*sin(phi) ),   cos(yaw) = C*cos(phi) - S*sin(phi)
// with S = sin(pi32/2), C = cos(pi32/2) (host libm; C = -4.371e-8 carries the fp32-pi quirk).
// Three transcendentals per ray become two divides; the result agrees with the libm evaluation
// to a few ulp of sin/cos (<= ~5e-16 absolute), i.e. ~1e-17 in the output cosines.
template <bool WAVECOL>
struct RadgratOp {
    PXF_OP_DEFAULTS
    static constexpr unsigned RD = B(X) | B(Y) | DIR | (WAVECOL ? B(AUX) : 0u);
    static constexpr unsigned WR = DIR;
    double dpermm, order, wave, S, C;
    template <int V>
    __device__ __forceinline__ void apply(Ray (&r)[V], const bool (&sel)[V], unsigned (&wm)[V]) const
    {
#pragma unroll
        for (int k = 0; k < V; k++) {
            double *c = r[k].c;
            const double x = c[X], y = c[Y];
            const double sn = WAVECOL ? y / fabs(y) : c[N] / fabs(c[N]);
            const double w = WAVECOL ? c[AUX] : wave;
            const double rho = sqrt(y * y + x * x);
            const double d = dpermm * rho;
            const double cphi = y / rho, sphi = x / rho;
            const double syaw = -(S * cphi + C * sphi);
            const double cyaw = C * cphi - S * sphi;
            c[L] = c[L] + syaw * order * w / d;
            c[M] = c[M] - cyaw * order * w / d;
            c[N] = sn * sqrt(1. - c[L] * c[L] - c[M] * c[M]);
            wm[k] = WR;
        }
    }
};

// applyT / applyTPos : transformations.py:257-289.  out = M[:3,:4] . [v,1].  Out-of-place.
struct Mat34 {
    double a[12];
};

template <int V, int NVEC>
__global__ __launch_bounds__(kBlock) void applyt_kernel(Cols in, Cols out, int64_t first, int64_t nvec,
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

// pointTo : transformations.py:91-100.   R x y z  W l m n
struct PointToOp {
    PXF_OP_DEFAULTS
    static constexpr unsigned RD = POS;
    static constexpr unsigned WR = DIR;
    double x0, y0, z0, reverse;
    template <int V>
    __device__ __forceinline__ void apply(Ray (&r)[V], const bool (&sel)[V], unsigned (&wm)[V]) const
    {
#pragma unroll
        for (int k = 0; k < V; k++) {
            double *c = r[k].c;
            const double dx = c[X] - x0, dy = c[Y] - y0, dz = c[Z] - z0;
            const double R = sqrt(dx * dx + dy * dy + dz * dz);
            c[L] = reverse * dx / R;
            c[M] = reverse * dy / R;
            c[N] = reverse * dz / R;
            wm[k] = WR;
        }
    }
};

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

static void trig6(double rx, double ry, double rz, bool inverse, double (&cs)[6])
{
    if (inverse) {  // itransform rotates by tmp = -rz, -ry, -rx (transformationsf.f95:180-191)
        rx = -rx;
        ry = -ry;
        rz = -rz;
    }
    cs[0] = cos(rx);
    cs[1] = sin(rx);
    cs[2] = cos(ry);
    cs[3] = sin(ry);
    cs[4] = cos(rz);
    cs[5] = sin(rz);
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
    TransformOp<false> op{tx, ty, tz, cs[0], cs[1], cs[2], cs[3], cs[4], cs[5]};
    return launch_rays(cols_xyzlmnu(x, y, z, l, m, n, ux, uy, uz), num, mask, stream, op);
}

int pxf_itransform(double *x, double *y, double *z, double *l, double *m, double *n, double *ux,
                   double *uy, double *uz, int64_t num, double tx, double ty, double tz, double rx,
                   double ry, double rz, const uint8_t *mask, pxf_stream_t stream)
{
    double cs[6];
    trig6(rx, ry, rz, true, cs);
    TransformOp<true> op{tx, ty, tz, cs[0], cs[1], cs[2], cs[3], cs[4], cs[5]};
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

static void radgrat_consts(double &S, double &C)
{
    const double pi32 = (double)acosf(-1.f);  // pi = acos(-1.) in REAL*4, transformationsf.f95:214,251
    S = sin(pi32 / 2);
    C = cos(pi32 / 2);
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

}  // extern "C"
