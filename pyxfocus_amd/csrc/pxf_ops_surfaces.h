// pxf_ops_surfaces.h -- per-ray bodies (Ops) of flat, conic and Wolter-I primary/secondary.
// Shared by the one-launch-per-call kernels (pxf_surfaces.hip) and the fused ray-program kernel
// (pxf_program.hip): ONE source of arithmetic, so both paths are bit-identical by construction.
//
// Each Op is the body of one reference Fortran loop, written in the reference's floating-point
// operation order (no FMA contraction) so the results are bit-identical to it; citations are
// file:line in the reference tree.  Contract with the program kernel: `apply` modifies ONLY the
// registers of columns it reports in wm[k] (a failed conic ray leaves position/normal alone).
#pragma once
#include <math.h>

#include "pxf_exactdiv.h"
#include "pxf_kernel.h"

namespace pxf {

// ---- flat / flatOPD : surfacesf.f95:4-29 / 32-53 ------------------------------------------
// R x y z l m n (+opd)   W x y z ux uy uz (+opd)   = 96 B/ray (+16)
template <bool WOPD>
struct FlatOp {
    PXF_OP_DEFAULTS
    static constexpr unsigned RD = POS | DIR | (WOPD ? B(OPD) : 0u);
    static constexpr unsigned WR = POS | NRM | (WOPD ? B(OPD) : 0u);
    double nr;
    // the propagation itself (position only); returns the distance.  Shared with the fanned-grating kernels
    // (pxf_fan.hip), which keep the normal implicit while a ray walks the fan.
    static __device__ __forceinline__ double distance(double z, double n)
    {
        // `delta` is implicitly REAL*4 in the reference: round the distance to fp32
        const float delta = (float)(-z / n);
        return (double)delta;
    }
    static __device__ __forceinline__ void move_xy(double *c, double dd)
    {
        c[X] = c[X] + dd * c[L];
        c[Y] = c[Y] + dd * c[M];
    }
    static __device__ __forceinline__ double advance(double *c)
    {
        const double dd = distance(c[Z], c[N]);
        c[Z] = 0.;
        move_xy(c, dd);
        return dd;
    }
    template <int V>
    __device__ __forceinline__ void apply(Ray (&r)[V], const bool (&sel)[V], unsigned (&wm)[V]) const
    {
#pragma unroll
        for (int k = 0; k < V; k++) {
            const double dd = advance(r[k].c);
            r[k].c[UX] = 0.;
            r[k].c[UY] = 0.;
            r[k].c[UZ] = 1.;
            if (WOPD) r[k].c[OPD] = r[k].c[OPD] + dd * nr;
            wm[k] = WR;
        }
    }
};

// ---- conic / conicopd : surfacesf.f95:302-360 / 366-420 -------------------------------------
// R x y z l m n (+opd)   W x y z ux uy uz (+opd) on a hit, l m n on a miss  = 96 B/ray (+16)
template <bool WOPD>
struct ConicOp {
    PXF_OP_DEFAULTS
    static constexpr unsigned RD = POS | DIR | (WOPD ? B(OPD) : 0u);
    static constexpr unsigned WR = POS | DIR | NRM | (WOPD ? B(OPD) : 0u);
    double R, K, nr;
    template <int V>
    __device__ __forceinline__ void apply(Ray (&r)[V], const bool (&sel)[V], unsigned (&wm)[V]) const
    {
        const double K1 = K + 1, twoR = 2 * R;
#pragma unroll
        for (int k = 0; k < V; k++) {
            double x = r[k].c[X], y = r[k].c[Y], z = r[k].c[Z];
            const double l = r[k].c[L], m = r[k].c[M], n = r[k].c[N];
            double s = 0.;
            if (K == -1 && fabs(n) == 1.) {  // :316-317 axial ray onto a paraboloid
                s = (x * x + y * y - twoR * z) / (twoR * n);
            } else {
                double denom = l * l + m * m + K1 * (n * n);
                double b = x * l + y * m + (K1 * z - R) * n;
                b = b / denom;
                double c;
                if (WOPD)  // :386 -- term order differs from :322
                    c = x * x + y * y + K1 * (z * z) - twoR * z;
                else
                    c = x * x + y * y - twoR * z + K1 * (z * z);
                c = c / denom;
                const double disc = b * b - c;
                if (disc >= 0.) {
                    const double sq = sqrt(disc);
                    const double s1 = -b + sq, s2 = -b - sq;
                    s = (fabs(s1) <= fabs(s2)) ? s1 : s2;  // :332-336, ties pick s1
                }
            }
            if (s == 0.) {  // :342-345 miss OR already on the surface: kill the ray in band
                r[k].c[L] = 0.;
                r[k].c[M] = 0.;
                r[k].c[N] = 0.;
                wm[k] = DIR;
            } else {
                x = x + l * s;
                y = y + m * s;
                z = z + n * s;
                const double r2 = x * x + y * y;
                const double denom = sqrt(R * R - K * r2);
                double uz = -(R / fabs(R) * sqrt(R * R - K1 * r2));
                r[k].c[X] = x;
                r[k].c[Y] = y;
                r[k].c[Z] = z;
                r[k].c[UX] = -x / denom;
                r[k].c[UY] = -y / denom;
                r[k].c[UZ] = -uz / denom;
                if (WOPD) r[k].c[OPD] = r[k].c[OPD] + s * nr;
                wm[k] = POS | NRM | (WOPD ? B(OPD) : 0u);
            }
        }
    }
};

// ---- Newton driver shared by the Wolter kernels -------------------------------------------
// All V rays of a thread iterate in ONE loop (two independent dependency chains interleave and
// hide the fp64 divide latency); a ray stops updating the moment its own |delt| <= tol, so each
// ray sees exactly the reference's iteration sequence.  The surface normal is built from the
// gradient evaluated BEFORE the last step (woltsurf.f95:44-47,151-154) -- reproduced, not fixed.

// wolterprimary / wolterprimaryopd : woltsurf.f95:7-54 / 60-108
// R x y z l m n (+opd)   W x y z ux uy uz (+opd)   = 96 B/ray (+16)
template <bool WOPD>
struct WolterPrimaryOp {
    PXF_OP_DEFAULTS
    static constexpr unsigned RD = POS | DIR | (WOPD ? B(OPD) : 0u);
    static constexpr unsigned WR = POS | NRM | (WOPD ? B(OPD) : 0u);
    double twop;  // 2*p  (= Fz)
    double pp;    // p**2
    double cst;   // 4*e**2*p*d/(e**2-1)
    double nr;
    template <int V>
    __device__ __forceinline__ void apply(Ray (&r)[V], const bool (&sel)[V], unsigned (&wm)[V]) const
    {
        const double tol = WOPD ? PXF_TOL_1EM10 : PXF_TOL_1EM8;
        const double Fz = twop;
        bool act[V];
        double Fx[V], Fy[V];
        bool any = false;
#pragma unroll
        for (int k = 0; k < V; k++) {
            act[k] = sel[k];
            any |= act[k];
            Fx[k] = 0.;
            Fy[k] = 0.;
        }
        for (int it = 0; any && it < kNewtonCap; it++) {
            any = false;
#pragma unroll
            for (int k = 0; k < V; k++) {
                if (act[k]) {
                    const double x = r[k].c[X], y = r[k].c[Y], z = r[k].c[Z];
                    const double F = twop * z + pp + cst - x * x - y * y;
                    Fx[k] = -2. * x;
                    Fy[k] = -2. * y;
                    const double Fp = Fx[k] * r[k].c[L] + Fy[k] * r[k].c[M] + Fz * r[k].c[N];
                    const double delt = -F / Fp;
                    r[k].c[X] = x + r[k].c[L] * delt;
                    r[k].c[Y] = y + r[k].c[M] * delt;
                    r[k].c[Z] = z + r[k].c[N] * delt;
                    if (WOPD) r[k].c[OPD] = r[k].c[OPD] + nr * delt;
                    act[k] = fabs(delt) > tol;
                    any |= act[k];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < V; k++) {
            const double Fp = sqrt(Fx[k] * Fx[k] + Fy[k] * Fy[k] + Fz * Fz);
            div3_exact(Fx[k], Fy[k], Fz, Fp, r[k].c[UX], r[k].c[UY], r[k].c[UZ]);   // == Fx/Fp, Fy/Fp, Fz/Fp
            wm[k] = WR;
        }
    }
};

// woltersecondary : woltsurf.f95:114-161      R x y z l m n   W x y z ux uy uz  = 96 B/ray
struct WolterSecondaryOp {
    PXF_OP_DEFAULTS
    static constexpr unsigned RD = POS | DIR;
    static constexpr unsigned WR = POS | NRM;
    double e2;     // e**2
    double twoe2;  // 2*e**2
    double d;
    template <int V>
    __device__ __forceinline__ void apply(Ray (&r)[V], const bool (&sel)[V], unsigned (&wm)[V]) const
    {
        bool act[V];
        double Fx[V], Fy[V], Fz[V];
        bool any = false;
#pragma unroll
        for (int k = 0; k < V; k++) {
            act[k] = sel[k];
            any |= act[k];
            Fx[k] = Fy[k] = Fz[k] = 0.;
        }
        for (int it = 0; any && it < kNewtonCap; it++) {
            any = false;
#pragma unroll
            for (int k = 0; k < V; k++) {
                if (act[k]) {
                    const double x = r[k].c[X], y = r[k].c[Y], z = r[k].c[Z];
                    const double dz = d + z;
                    const double F = e2 * (dz * dz) - z * z - x * x - y * y;
                    Fx[k] = -2. * x;
                    Fy[k] = -2. * y;
                    Fz[k] = twoe2 * dz - 2 * z;
                    const double Fp = Fx[k] * r[k].c[L] + Fy[k] * r[k].c[M] + Fz[k] * r[k].c[N];
                    const double delt = -F / Fp;
                    r[k].c[X] = x + r[k].c[L] * delt;
                    r[k].c[Y] = y + r[k].c[M] * delt;
                    r[k].c[Z] = z + r[k].c[N] * delt;
                    act[k] = fabs(delt) > PXF_TOL_1EM8;
                    any |= act[k];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < V; k++) {
            const double Fp = sqrt(Fx[k] * Fx[k] + Fy[k] * Fy[k] + Fz[k] * Fz[k]);
            div3_exact(Fx[k], Fy[k], Fz[k], Fp, r[k].c[UX], r[k].c[UY], r[k].c[UZ]);   // == Fx/Fp, Fy/Fp, Fz/Fp
            wm[k] = WR;
        }
    }
};


// ---- SPO cone : woltsurf.f95:591-638 --------------------------------------------------------
// Closed-form quadratic; a missed ray (det < 0) gets zeroed direction cosines, position and
// normal untouched.  R xyz lmn  W xyz u / lmn   = 96 B/ray
struct SpoConeOp {
    PXF_OP_DEFAULTS
    static constexpr unsigned RD = POS | DIR;
    static constexpr unsigned WR = POS | DIR | NRM;
    double R0, sl, ctg, stg;  // sl = tan(tg), cos(tg), sin(tg): host libm
    template <int V>
    __device__ __forceinline__ void apply(Ray (&r)[V], const bool (&sel)[V], unsigned (&wm)[V]) const
    {
#pragma unroll
        for (int k = 0; k < V; k++) {
            double *c = r[k].c;
            const double x = c[X], y = c[Y], z = c[Z], l = c[L], m = c[M], n = c[N];
            const double sl2 = sl * sl;
            const double A = n * n * sl2 - m * m - l * l;
            const double Bq = 2 * n * sl * R0 + 2 * z * sl2 * n - 2 * x * l - 2 * y * m;
            const double C = R0 * R0 + 2 * sl * R0 * z + z * z * sl2 - x * x - y * y;
            const double det = Bq * Bq - 4 * A * C;
            if (det >= 0) {
                const double sq = sqrt(det);
                double t1, t2;
                div2_exact(-Bq + sq, -Bq - sq, 2 * A, t1, t2);   // == (-Bq + sq) / (2 A), (-Bq - sq) / (2 A)
                if (fabs(t2) < fabs(t1)) t1 = t2;
                const double xn = x + t1 * l, yn = y + t1 * m;
                c[X] = xn;
                c[Y] = yn;
                c[Z] = z + t1 * n;
                const double rr = sqrt(xn * xn + yn * yn);
                double ax, ay;
                div2_exact(-xn, -yn, rr, ax, ay);                // == -xn / rr, -yn / rr
                c[UX] = ax * ctg;
                c[UY] = ay * ctg;
                c[UZ] = stg;
                wm[k] = POS | NRM;
            } else {
                c[L] = 0.;
                c[M] = 0.;
                c[N] = 0.;
                wm[k] = DIR;
            }
        }
    }
};

// ---- sphere / cylinder about the origin : surfacesf.f95:57-148 / 153-246 ------------------------
// A miss zeroes position AND direction; the normal is then 0/0 = NaN, as in the reference.
// R xyz lmn (+opd)  W xyz lmn u (+opd)  = 120 B/ray (+16)
template <bool WOPD, bool CYL>
struct SphereCylOp {
    PXF_OP_DEFAULTS
    static constexpr unsigned RD = POS | DIR | (WOPD ? B(OPD) : 0u);
    static constexpr unsigned WR = POS | DIR | NRM | (WOPD ? B(OPD) : 0u);
    double rad, nr;
    template <int V>
    __device__ __forceinline__ void apply(Ray (&r)[V], const bool (&sel)[V], unsigned (&wm)[V]) const
    {
#pragma unroll
        for (int k = 0; k < V; k++) {
            double *c = r[k].c;
            double x = c[X], y = c[Y], z = c[Z];
            const double l = c[L], m = c[M], n = c[N];
            bool hit;
            double d1 = 0.;
            if (!CYL) {
                const double dotol = l * x + m * y + n * z;
                const double mago = x * x + y * y + z * z;
                const double det = dotol * dotol - mago + rad * rad;
                hit = !(det < 0);
                if (hit) {
                    const double sq = sqrt(det);
                    d1 = -dotol + sq;
                    const double d2 = -dotol - sq;
                    if (fabs(d2) < fabs(d1)) d1 = d2;
                }
            } else {
                const double a = l * l + n * n;
                const double b = 2 * (x * l + z * n);
                const double cc = x * x + z * z - rad * rad;
                const double det = b * b - 4 * a * cc;
                hit = !(det < 0);
                if (hit) {
                    const double sq = sqrt(det);
                    d1 = (-b + sq) / 2 / a;
                    const double d2 = (-b - sq) / 2 / a;
                    if (fabs(d2) < fabs(d1)) d1 = d2;
                }
            }
            if (hit) {
                if (!CYL) {
                    x = x + d1 * l;
                    y = y + d1 * m;
                    z = z + d1 * n;
                } else {
                    x = x + l * d1;
                    y = y + m * d1;
                    z = z + n * d1;
                }
                if (WOPD) c[OPD] = c[OPD] + d1 * nr;
            } else {
                x = y = z = 0.;
                c[L] = c[M] = c[N] = 0.;
            }
            c[X] = x;
            c[Y] = y;
            c[Z] = z;
            if (!CYL) {
                const double mago = sqrt(x * x + y * y + z * z);
                c[UX] = x / mago;
                c[UY] = y / mago;
                c[UZ] = z / mago;
            } else {
                const double mag = sqrt(x * x + z * z);
                c[UX] = x / mag;
                c[UZ] = z / mag;
                c[UY] = 0.;
            }
            wm[k] = hit ? (POS | NRM | (WOPD ? B(OPD) : 0u)) : (POS | DIR | NRM);
        }
    }
};

// ---- cylindrical conic : surfacesf.f95:251-296 (Newton, tol 1.e-10, stale-gradient normal) -----
struct CylConicOp {
    PXF_OP_DEFAULTS
    static constexpr unsigned RD = POS | DIR;
    static constexpr unsigned WR = POS | NRM;
    double rad, k1r2;  // k1r2 = (1+k)*rad**2
    template <int V>
    __device__ __forceinline__ void apply(Ray (&r)[V], const bool (&sel)[V], unsigned (&wm)[V]) const
    {
        bool act[V];
        double Fx[V], Fy[V];
        bool any = false;
#pragma unroll
        for (int k = 0; k < V; k++) {
            act[k] = sel[k];
            any |= act[k];
            Fx[k] = Fy[k] = 0.;
        }
        for (int it = 0; any && it < kNewtonCap; it++) {
            any = false;
#pragma unroll
            for (int k = 0; k < V; k++) {
                if (act[k]) {
                    double *c = r[k].c;
                    const double x = c[X];
                    const double q = 1 - k1r2 * (x * x);
                    const double sq = sqrt(q);
                    const double low = 1 + sq;
                    const double high = rad * (x * x);
                    const double dL = -(k1r2 * x) / sq;
                    const double dH = 2 * rad * x;
                    const double F = c[Y] - high / low;
                    Fx[k] = (high * dL - low * dH) / (low * low);
                    Fy[k] = 1.;
                    const double Fp = Fx[k] * c[L] + Fy[k] * c[M];
                    const double delt = -F / Fp;
                    c[X] = x + c[L] * delt;
                    c[Y] = c[Y] + c[M] * delt;
                    c[Z] = c[Z] + c[N] * delt;
                    act[k] = fabs(delt) > PXF_TOL_1EM10;
                    any |= act[k];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < V; k++) {
            const double Fp = sqrt(Fx[k] * Fx[k] + Fy[k] * Fy[k]);
            r[k].c[UX] = Fx[k] / Fp;
            r[k].c[UY] = Fy[k] / Fp;
            r[k].c[UZ] = 0.;
            wm[k] = WR;
        }
    }
};

// ---- ideal paraxial lens : surfacesf.f95:423-439 / 443-459 --------------------------------------
template <bool YONLY>
struct ParaxialOp {
    PXF_OP_DEFAULTS
    static constexpr unsigned RD = (YONLY ? 0u : (B(X) | B(L))) | B(Y) | B(M);
    static constexpr unsigned WR = (YONLY ? 0u : B(L)) | B(M);
    double F;
    template <int V>
    __device__ __forceinline__ void apply(Ray (&r)[V], const bool (&sel)[V], unsigned (&wm)[V]) const
    {
#pragma unroll
        for (int k = 0; k < V; k++) {
            double *c = r[k].c;
            if (!YONLY) c[L] = c[L] - c[X] / F;
            c[M] = c[M] - c[Y] / F;
            wm[k] = WR;
        }
    }
};

// host: Van Speybroeck constants -> the Ops' precomputed scalars (woltsurf.f95:18-23,31,137-140)
template <bool WOPD>
inline WolterPrimaryOp<WOPD> make_wolter_primary(double r0, double z0, double psi, double nr)
{
    double p, d, e;
    pxf_wolter_params(r0, z0, psi, &p, &d, &e);
    return WolterPrimaryOp<WOPD>{2 * p, p * p, 4 * (e * e) * p * d / (e * e - 1), nr};
}

inline WolterSecondaryOp make_wolter_secondary(double r0, double z0, double psi)
{
    double p, d, e;
    pxf_wolter_params(r0, z0, psi, &p, &d, &e);
    return WolterSecondaryOp{e * e, 2 * (e * e), d};
}

}  // namespace pxf
