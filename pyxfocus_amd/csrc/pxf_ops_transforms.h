// pxf_ops_transforms.h -- per-ray bodies (Ops) of transform / itransform / reflect / radgrat[W] /
// pointTo, shared by pxf_transforms.hip and the fused ray-program kernel (pxf_program.hip).
//
// Reference: transformationsf.f95 (native loops) and transformations.py (pointTo).
// sin/cos of the six frame parameters are evaluated ONCE on the host with glibc libm (the
// reference re-evaluates the same pure function 18 times per ray, transformationsf.f95:3-28)
// and passed by value; the per-ray work is then add/mul only, in the reference's order, so the
// result is bit-identical.
#pragma once
#include <math.h>

#include "pxf_exactdiv.h"
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

// A rotation by an angle of zero -- (cos, sin) = (1, +-0) -- is the identity for every component that is not -0,
// not infinite and not NaN: 1*x - 0*y = x - (+-0) = x, and +0 stays +0.  (A -0 can come back +0 and a NaN or infinity
// reaches its neighbours: those vectors take the full arithmetic.)  `v_cmp_class_f64` tests all of that in one
// instruction, so a vector whose three rotations are all by zero costs three class tests instead of 18 multiplies
// and adds -- it matters in the VALU-bound fused programs, whose leading frame change is usually a pure translation.
__device__ __forceinline__ bool special_value(double v)
{
    return __builtin_amdgcn_class(v, 0x001 | 0x002 | 0x004 | 0x020 | 0x200);   // sNaN | qNaN | -inf | -0 | +inf
}
__device__ __forceinline__ bool plain3(double a, double b, double c)
{
    return !((int)special_value(a) | (int)special_value(b) | (int)special_value(c));
}
__device__ __forceinline__ bool zero_angle(double c, double s) { return c == 1. && s == 0.; }

// transform : transformationsf.f95:134-163 ; itransform : :168-201.   R 9 + W 9 = 144 B/ray
template <bool INVERSE>
struct TransformOp {
    PXF_OP_DEFAULTS
    static constexpr unsigned RD = POS | DIR | NRM;
    static constexpr unsigned WR = POS | DIR | NRM;
    double tx, ty, tz;
    double cx, sx, cy, sy, cz, sz;  // forward: cos/sin(rx..rz); inverse: cos/sin(-rx..-rz)
    // which of the three independent vector groups to move: bit0 position, bit1 direction, bit2
    // normal.  Always 7 for a stand-alone launch; the fused program drops groups whose result is
    // overwritten before anything reads it (e.g. the normal ahead of a surface intersection).
    int groups = 7;
    template <int V>
    __device__ __forceinline__ void apply(Ray (&r)[V], const bool (&sel)[V], unsigned (&wm)[V]) const
    {
#pragma unroll
        for (int k = 0; k < V; k++) {
            double *c = r[k].c;
            const bool gp = groups & 1, gd = groups & 2, gn = groups & 4;
            double before[9];
#pragma unroll
            for (int j = 0; j < 9; j++) before[j] = c[X + j];
            const bool norot = zero_angle(cx, sx) && zero_angle(cy, sy) && zero_angle(cz, sz);   // wave-uniform
            if (!INVERSE) {
                if (gp) {
                    c[X] = c[X] + tx;
                    c[Y] = c[Y] + ty;
                    c[Z] = c[Z] + tz;
                    if (!(norot && plain3(c[X], c[Y], c[Z]))) {
                        rotv<1>(c[X], c[Y], c[Z], cx, sx);
                        rotv<2>(c[X], c[Y], c[Z], cy, sy);
                        rotv<3>(c[X], c[Y], c[Z], cz, sz);
                    }
                }
                if (gd && !(norot && plain3(c[L], c[M], c[N]))) {
                    rotv<1>(c[L], c[M], c[N], cx, sx);
                    rotv<2>(c[L], c[M], c[N], cy, sy);
                    rotv<3>(c[L], c[M], c[N], cz, sz);
                }
                if (gn && !(norot && plain3(c[UX], c[UY], c[UZ]))) {
                    rotv<1>(c[UX], c[UY], c[UZ], cx, sx);
                    rotv<2>(c[UX], c[UY], c[UZ], cy, sy);
                    rotv<3>(c[UX], c[UY], c[UZ], cz, sz);
                }
            } else {
                if (gp) {
                    if (!(norot && plain3(c[X], c[Y], c[Z]))) {
                        rotv<3>(c[X], c[Y], c[Z], cz, sz);
                        rotv<2>(c[X], c[Y], c[Z], cy, sy);
                        rotv<1>(c[X], c[Y], c[Z], cx, sx);
                    }
                    c[X] = c[X] - tx;
                    c[Y] = c[Y] - ty;
                    c[Z] = c[Z] - tz;
                }
                if (gd && !(norot && plain3(c[L], c[M], c[N]))) {
                    rotv<3>(c[L], c[M], c[N], cz, sz);
                    rotv<2>(c[L], c[M], c[N], cy, sy);
                    rotv<1>(c[L], c[M], c[N], cx, sx);
                }
                if (gn && !(norot && plain3(c[UX], c[UY], c[UZ]))) {
                    rotv<3>(c[UX], c[UY], c[UZ], cz, sz);
                    rotv<2>(c[UX], c[UY], c[UZ], cy, sy);
                    rotv<1>(c[UX], c[UY], c[UZ], cx, sx);
                }
            }
            // Store only what CHANGED (compared as bit patterns, so -0 -> +0 and NaN results count exactly as the
            // arithmetic produced them).  The most common frame change in PyXFocus scripts is a pure translation,
            // transform(rays, 0, 0, dz, 0, 0, 0): all nine numbers still go through the reference's arithmetic
            // (1*x - 0*y ... : a NaN or infinity in one component reaches its neighbours, a -0 may become +0), but
            // for ordinary rays eight of them come back unchanged and their columns are not rewritten:
            // 72 B read + 8 B written per ray instead of 144.  A general rotation changes everything and stores
            // everything, as before.
            unsigned w = 0;
#pragma unroll
            for (int j = 0; j < 9; j++)
                if (__double_as_longlong(c[X + j]) != __double_as_longlong(before[j])) w |= B(X + j);
            wm[k] = w;
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
            const double sn = unit_sign_exact(WAVECOL ? y : c[N]);   // == y/|y|, n/|n|
            const double w = WAVECOL ? c[AUX] : wave;
            const double rho = sqrt(y * y + x * x);
            const double d = dpermm * rho;
            double cphi, sphi, dl, dm;
            div2_exact(y, x, rho, cphi, sphi);                       // == y/rho, x/rho
            const double syaw = -(S * cphi + C * sphi);
            const double cyaw = C * cphi - S * sphi;
            div2_exact(syaw * order * w, cyaw * order * w, d, dl, dm);   // == syaw*order*w/d, cyaw*order*w/d
            c[L] = c[L] + dl;
            c[M] = c[M] - dm;
            c[N] = sn * sqrt(1. - c[L] * c[L] - c[M] * c[M]);
            wm[k] = WR;
        }
    }
};

// grat : transformationsf.f95:277-305 (linear grating, grooves along y).  The Fortran takes
// per-ray `order` and `wave` arrays; only their product enters (l - order*wave/d), so the entry
// takes either a scalar product or a per-ray product column (AUX).  Evanescent -> zeroed cosines.
// R l m n (+ow)  W l m n  = 48 B/ray (56)
template <bool OWCOL>
struct GratOp {
    PXF_OP_DEFAULTS
    static constexpr unsigned RD = DIR | (OWCOL ? B(AUX) : 0u);
    static constexpr unsigned WR = DIR;
    double d, ow;
    template <int V>
    __device__ __forceinline__ void apply(Ray (&r)[V], const bool (&sel)[V], unsigned (&wm)[V]) const
    {
#pragma unroll
        for (int k = 0; k < V; k++) {
            double *c = r[k].c;
            const double sn = c[N] / fabs(c[N]);
            const double prod = OWCOL ? c[AUX] : ow;
            const double l = c[L] - prod / d, m = c[M];
            double n = sn * sqrt(1 - l * l - m * m);
            const bool evan = (l * l + m * m) > 1;
            c[L] = evan ? 0. : l;
            c[M] = evan ? 0. : m;
            c[N] = evan ? 0. : n;
            wm[k] = WR;
        }
    }
};

// refract : transformationsf.f95:82-130 (+ rotateaxis :32-55).  Snell rotation of the direction
// about (u x d); the normal is flipped to the far side first.  acos/asin/sin/cos come from ocml,
// so parity is to ~1e-15, not bitwise.  Total internal reflection -> NaN, as in the reference.
// R lmn u  W lmn u = 96 B/ray
struct RefractOp {
    PXF_OP_DEFAULTS
    static constexpr unsigned RD = DIR | NRM;
    static constexpr unsigned WR = DIR | NRM;
    double n1, n2;
    template <int V>
    __device__ __forceinline__ void apply(Ray (&r)[V], const bool (&sel)[V], unsigned (&wm)[V]) const
    {
#pragma unroll
        for (int k = 0; k < V; k++) {
            double *c = r[k].c;
            double dot = c[L] * c[UX] + c[M] * c[UY] + c[N] * c[UZ];
            if (dot < 0) {
                c[UX] = -c[UX];
                c[UY] = -c[UY];
                c[UZ] = -c[UZ];
                dot = -dot;
            }
            wm[k] = WR;
            if (dot == 1) continue;
            const double t1 = acos(dot);
            const double t2 = asin((n1 / n2) * sin(t1));
            double ax = c[UY] * c[N] - c[M] * c[UZ];
            double ay = c[L] * c[UZ] - c[UX] * c[N];
            double az = c[UX] * c[M] - c[L] * c[UY];
            const double dt = t2 - t1;
            const double mag = sqrt(ax * ax + ay * ay + az * az);
            ax = ax / mag;
            ay = ay / mag;
            az = az / mag;
            const double cs = cos(dt), sn = sin(dt);
            const double x = c[L], y = c[M], z = c[N];
            const double o1 = (cs + ax * ax * (1 - cs)) * x + (ax * ay * (1 - cs) - az * sn) * y + (ax * az * (1 - cs) + ay * sn) * z;
            const double o2 = (ay * ax * (1 - cs) + az * sn) * x + (cs + ay * ay * (1 - cs)) * y + (ay * az * (1 - cs) - ax * sn) * z;
            const double o3 = (az * ax * (1 - cs) - ay * sn) * x + (az * ay * (1 - cs) + ax * sn) * y + (cs + az * az * (1 - cs)) * z;
            const double alpha = sqrt(o1 * o1 + o2 * o2 + o3 * o3);
            c[L] = o1 / alpha;
            c[M] = o2 / alpha;
            c[N] = o3 / alpha;
        }
    }
};

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

inline void trig6(double rx, double ry, double rz, bool inverse, double (&cs)[6])
{
    if (inverse) {  // itransform rotates by tmp = -rz, -ry, -rx (transformationsf.f95:180-191)
        rx = -rx;
        ry = -ry;
        rz = -rz;
    }
    host_sincos(rx, cs[0], cs[1]);
    host_sincos(ry, cs[2], cs[3]);
    host_sincos(rz, cs[4], cs[5]);
}

inline void radgrat_consts(double &S, double &C)
{
    const double pi32 = (double)acosf(-1.f);  // pi = acos(-1.) in REAL*4, transformationsf.f95:214,251
    S = sin(pi32 / 2);
    C = cos(pi32 / 2);
}

// indAngle / grazeAngle : analyses.py:164-187.  R l m n ux uy uz  W out  = 56 B/ray.  Shared by pxf_indangle
// and the program operation PXF_OP_INDANGLE (same arithmetic, same bits).
struct IndAngleOp {
    PXF_OP_DEFAULTS
    static constexpr unsigned RD = DIR | NRM;
    static constexpr unsigned WR = B(AUX);
    int graze;
    template <int V>
    __device__ __forceinline__ void apply(Ray (&r)[V], const bool (&sel)[V], unsigned (&wm)[V]) const
    {
#pragma unroll
        for (int k = 0; k < V; k++) {
            const double *c = r[k].c;
            const double a = acos(c[L] * c[UX] + c[M] * c[UY] + c[N] * c[UZ]);
            r[k].c[AUX] = graze ? 3.141592653589793 / 2 - a : a;
            wm[k] = WR;
        }
    }
};

}  // namespace pxf
