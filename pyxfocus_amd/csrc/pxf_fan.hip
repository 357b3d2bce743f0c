// pxf_fan.hip -- a fanned array of radial gratings, per ray (EXTENSION: no reference function; it is the
// `while` loop of the reference's examples/arcus/sector.py:663-704, `gratArray`).
//
// The reference walks a stack of gratings fanned about a common hub with a host loop over the WHOLE bundle:
// every round it rotates the frame of all rays by one grating pitch (`transform(0,0,0,ang,0,0)`), brings the rays
// that have not met a grating yet to the new grating plane (`flat(ind=...)`), tests which of them land inside
// the grating's radial extent, and diffracts those at the start of the next round (`reflect(ind)`,
// `radgrat(ind)`) -- ~226 rounds for the Arcus petal, each a dozen passes over HBM plus a host sync
// (2.4 s per 1e8 rays through the drop-in API).
//
// Here each ray walks the fan by itself, in registers, with the SAME device code the API calls run
// (TransformOp, FlatOp, ReflectOp, RadgratOp::apply) in the same order, so the result is bit-identical to
// the host loop:
//   pxf_fan_search  every ray iterates {rotate, flat if still free and not grazing, range test} until it meets
//                   a grating; the round index goes to `it_hit`, the largest one to `*imax` (the host loop's
//                   round count), rays that exhaust `cap` rounds are counted in `*nmiss`.
//   pxf_fan_finish  every ray replays the rounds after its own: diffract once (reflect + radgrat) at the
//                   start of round it_hit+1, then keep rotating with the bundle up to round imax (the host
//                   loop rotates ALL rays every round and undoes `imax` rotations at the end).
// Two launches, 2 x 148 B/ray of HBM traffic; fp64-VALU-bound: ~38 vector instructions per round in the search (eleven
// of them the IEEE divide of the fp32-rounded propagation; no square root: the range test compares x^2 + y^2 with
// host-found images of the bounds; the asin of the grazing test is only evaluated within 1e-9 of its threshold) and
// the 18 multiplies and adds of the rotation per round in the finish.  1e9 rays x 226 rows: 173 + 68 ms.
#include <math.h>

#include "pxf_kernel.h"
#include "pxf_ops_surfaces.h"
#include "pxf_ops_transforms.h"

namespace pxf {

struct FanGeom {
    double lo, hi, thr;  // grating radial extent (hubdist, hubdist + length) and the grazing threshold [rad]
    double s_lo, s_hi;   // sin(thr) (1 -+ 1e-9): outside this band |asin n| > thr is decided without the asin
    // exact images of lo / hi under the correctly rounded square root (host, fan_sqrt_bounds): for a >= 0,
    //   sqrt(a) > lo  <=>  a > a_lo      and      sqrt(a) < hi  <=>  a < a_hi ;   sq_ok = lo > 0 and both found
    double a_lo, a_hi;
    TransformOp<false> rot;
    int cap;
    int sq_ok;
};

// abs(arcsin(n)) > thr  (sector.py:669,693).  asin is monotone and ocml's is accurate to a few ulp, so for
// |n| outside [s_lo, s_hi] the comparison is decided by |n| alone; only a ray within 1e-9 (relative) of the
// threshold pays for the asin -- and gets exactly the driver's answer.
__device__ __forceinline__ bool fan_not_grazing(double n, const FanGeom &g)
{
    const double a = fabs(n);
    if (a > g.s_hi) return true;
    if (a < g.s_lo) return false;
    return fabs(asin(n)) > g.thr;  // (NaN lands here: false, as in the driver)
}

// the range test of sector.py:672-673,698-699:  rho = -sqrt(x^2+y^2)*sign(y);  lo < rho < hi
// With lo > 0 only rays with y < 0 can pass (rho = +sqrt(a)), and the IEEE square root is monotone, so the two
// comparisons of the rounded root are comparisons of a = x*x + y*y (the same rounded sum the root would take)
// with the host-found thresholds a_lo, a_hi -- the same booleans without a square root per round.
__device__ __forceinline__ bool fan_in_range(const double *c, const FanGeom &g)
{
    const double y = c[Y];
    const double a = c[X] * c[X] + c[Y] * c[Y];
    if (g.sq_ok) return y < 0. && a > g.a_lo && a < g.a_hi;   // (NaN: false, as below)
    const double sgn = y > 0. ? 1. : (y < 0. ? -1. : y);  // sign(0) = 0, sign(NaN) = NaN
    const double rho = -sqrt(a) * sgn;
    return rho > g.lo && rho < g.hi;
}

// A frame change whose translation is +0 and whose y and z angles are +0 (`transform(rays, 0, 0, 0, ang, 0, 0)`,
// the fan's pitch) sends every vector through the rotation about x and then two rotations by zero: 1*x + 0*z ... --
// the identity for every component that is not -0, not infinite and not NaN (a -0 can come back +0, a NaN or infinity
// contaminates its neighbours: those rays take the full arithmetic).  `v_cmp_class_f64` tests all three conditions in
// one instruction.  Only the values that ENTER the zero-angle rotations matter, i.e. the nine components AFTER the
// rotation about x (which is always done in full): a special input y or z either makes its rotated images special
// too (inf, NaN) or does not matter (-0 in, ordinary number out).  And the x-components (x, l, ux) pass the rotation
// about x untouched, so inside a loop that only rotates they are tested once, not every round.
__device__ __forceinline__ bool fan_special(double v) { return special_value(v); }   // pxf_ops_transforms.h
// ... the same, or a zero of either sign (where "c*y" was written for "c*y - s*0", below)
__device__ __forceinline__ bool fan_special_or_zero(double v)
{
    return __builtin_amdgcn_class(v, 0x001 | 0x002 | 0x004 | 0x020 | 0x040 | 0x200);
}
// +0 translation, (cos, sin) = (1, +0) about y and z -- compared as bit patterns (a -0 sine multiplies differently)
__device__ __forceinline__ bool fan_x_only(const TransformOp<false> &rot)
{
    const long long one = __double_as_longlong(1.);
    return __double_as_longlong(rot.tx) == 0 && __double_as_longlong(rot.ty) == 0 && __double_as_longlong(rot.tz) == 0 &&
           __double_as_longlong(rot.cy) == one && __double_as_longlong(rot.sy) == 0 &&
           __double_as_longlong(rot.cz) == one && __double_as_longlong(rot.sz) == 0;
}
// A vector that ONLY rotates about x, round after round, needs no test per round at all.  With (c, s) = (cos, sin) of
// the pitch both non-zero (fan_rot_window) and a pair (y, z) that holds no -0 / inf / NaN and whose larger magnitude
// lies in [2^-400, 2^400] (fan_pair_window):
//   * at least one of the products in  y' = c*y - s*z,  z' = s*y + c*z  is an ordinary non-zero number, so a result is
//     either non-zero or the +0 of an exact cancellation -- never -0 (that takes two zero terms), never inf / NaN;
//   * a rotation keeps max(|y|, |z|) to a few ulp, so the pair is inside the window again (1e5 rounds move it by 2^-35).
// By induction every later round's rotated pair is plain, and plain values pass the two zero-angle rotations that
// follow unchanged: the round is the rotation about x and nothing else.  The x-component never changes: tested once.
__device__ __forceinline__ bool fan_pair_window(double y, double z)
{
    const double m = fmax(fabs(y), fabs(z));   // (NaN operands are caught by the class tests)
    return !((int)special_value(y) | (int)special_value(z)) && m >= 0x1p-400 && m <= 0x1p400;
}
__device__ __forceinline__ bool fan_rot_window(const TransformOp<false> &rot)   // wave-uniform
{
    const double c = fabs(rot.cx), s = fabs(rot.sx);
    return c >= 0x1p-100 && c <= 1. && s >= 0x1p-100 && s <= 1.;
}
// the whole frame change on the groups asked for (1 position, 2 direction, 4 normal), full arithmetic
__device__ __forceinline__ void fan_rotate_full(TransformOp<false> rot, int groups, Ray (&r)[1])
{
    const bool sel[1] = {true};
    unsigned wm[1];
    rot.groups = groups;
    rot.apply<1>(r, sel, wm);
}
// what a flat leaves behind besides x, y (surfacesf.f95:12-15): z = +0, normal (0, 0, 1)
__device__ __forceinline__ void fan_materialise(double *q)
{
    q[Z] = 0.;
    q[UX] = 0.;
    q[UY] = 0.;
    q[UZ] = 1.;
}
__device__ __forceinline__ void load9(const Cols &c, int64_t i, Ray &r)
{
#pragma unroll
    for (int col = X; col <= UZ; col++) r.c[col] = c.p[col][i];
}
__device__ __forceinline__ void store9(const Cols &c, int64_t i, const Ray &r)
{
#pragma unroll
    for (int col = X; col <= UZ; col++) c.p[col][i] = r.c[col];
}

// The search.  Per round the host loop does: frame change of all nine numbers, `flat` unless the ray is grazing,
// range test.  What a ray that has just been through a `flat` carries is known: z = +0 and normal (0, 0, 1) -- so in
// that state (every round but the grazing ones, which are rare) a round is
//     y1 = c*y          for  c*y - s*(+0)      z1 = s*y       for  s*y + c*(+0)
//     m1 = c*m - s*n    n1 = s*m + c*n                                         (l, x: untouched by a rotation about x)
//     three class tests: x, y1, z1 (zeros count as special there: "- s*0" can turn a -0 into +0); the direction
//                       only ever rotates, so ONE window test before the loop covers it for every round (above)
//     the normal: NOT rotated -- the next flat overwrites it; only a round that turns out grazing needs the rotated
//                 (0, 0, 1), and gets it from the full arithmetic then
//     flat: the shared FlatOp::distance / move_xy on (x, y, z1); z = +0 and the normal stay implicit
// 10 multiplies / adds and 3 class tests where the generic round has 39 and 15.  Every other state (first round, after
// a grazing round, any special value anywhere) takes the generic round, which is the full reference arithmetic.
__global__ __launch_bounds__(kBlock, 2) void fan_search_kernel(Cols c, int64_t num, FanGeom g, int32_t *__restrict__ it_hit,
                                                            int32_t *__restrict__ imax, unsigned long long *__restrict__ nmiss)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    const bool rot_window = fan_x_only(g.rot) && fan_rot_window(g.rot);   // wave-uniform
    const double cx = g.rot.cx, sx = g.rot.sx;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < num; i += stride) {
        Ray r[1];
        load9(c, i, r[0]);
        double *q = r[0].c;
        // round 0: the bundle as it arrives
        // flat_prev: "the ray has just been through a flat": z = +0 and normal = (0, 0, 1) -- IMPLICITLY: the registers
        // of z and of the normal are not kept up to date in that state (fan_materialise writes them when they are needed)
        bool flat_prev = fan_not_grazing(q[N], g);
        if (flat_prev) FlatOp<false>::move_xy(q, FlatOp<false>::distance(q[Z], q[N]));
        bool hit = fan_in_range(q, g);
        int it = 0;
        // l never changes in the search and (m, n) only rotates: one test before the loop covers every round
        const bool dir_plain = rot_window && !fan_special(q[L]) && fan_pair_window(q[M], q[N]);
        while (!hit && it < g.cap) {
            it++;
            // The rays of a wave are neighbours in the bundle and walk the fan alike (97.7 % lane utilisation on the Arcus
            // petal, tools/dev/fan_divergence.py), and nearly every round is the plain one for all of them.  The plain
            // round is therefore tried for the WAVE: if every active lane qualifies (one ballot), it runs as straight-line
            // code with no per-lane branching -- the per-lane form below spends as many scalar instructions on exec-mask
            // bookkeeping as it spends vector instructions on arithmetic (PMC: 44 SALU + 48 VALU per round, 66 % VALU issue).
            {
                const double y1 = cx * q[Y], zz1 = sx * q[Y];
                double l1 = q[L], m1 = q[M], n1 = q[N];
                rotv<1>(l1, m1, n1, cx, sx);
                const bool plain = flat_prev && dir_plain &&
                                   !((int)fan_special(q[X]) | (int)fan_special_or_zero(y1) | (int)fan_special_or_zero(zz1)) &&
                                   fabs(n1) > g.s_hi;                              // not grazing, decided without the asin
                if (__builtin_amdgcn_ballot_w64(plain) == __builtin_amdgcn_ballot_w64(true)) {
                    q[Y] = y1;
                    q[M] = m1;
                    q[N] = n1;
                    FlatOp<false>::move_xy(q, FlatOp<false>::distance(zz1, n1));
                    hit = fan_in_range(q, g);
                    continue;
                }
            }
            bool hot = false;
            double z1 = 0.;
            if (flat_prev && dir_plain) {
                const double y1 = cx * q[Y];
                z1 = sx * q[Y];
                const int bad = (int)fan_special(q[X]) | (int)fan_special_or_zero(y1) | (int)fan_special_or_zero(z1);
                if (!bad) {
                    q[Y] = y1;
                    rotv<1>(q[L], q[M], q[N], cx, sx);
                    hot = true;
                }
            }
            if (!hot) {
                if (flat_prev) fan_materialise(q);
                fan_rotate_full(g.rot, 7, r);
                z1 = q[Z];
            }
            if (fan_not_grazing(q[N], g)) {
                FlatOp<false>::move_xy(q, FlatOp<false>::distance(z1, q[N]));
                flat_prev = true;
            } else {
                if (hot) {   // a grazing round after a flat: z as rotated, and (0,0,1) through this round's frame change
                    fan_materialise(q);
                    fan_rotate_full(g.rot, 4, r);
                    q[Z] = z1;
                }
                flat_prev = false;
            }
            hit = fan_in_range(q, g);
        }
        if (flat_prev) fan_materialise(q);
        store9(c, i, r[0]);
        it_hit[i] = hit ? it : -1;
        // the bundle's round count: every wave would send one atomic to the SAME address (1.6e7 of them at 1e9 rays);
        // a plain read first -- the value only ever grows, so a stale one merely lets a few redundant atomics through
        if (hit) {
            if (it > __atomic_load_n(imax, __ATOMIC_RELAXED)) atomicMax(imax, it);
        } else {
            atomicAdd(nmiss, 1ull);
        }
    }
}

// The finish.  After its one diffraction a ray only rotates, so the window argument above applies to all three of its
// vectors: three x-components and three (y, z) pairs tested ONCE, then every round is the 18 multiplies and adds of the
// rotation about x, in place.  A ray that fails the test takes the full frame change every round.
__global__ __launch_bounds__(kBlock, 2) void fan_finish_kernel(Cols c, int64_t num, TransformOp<false> rot, RadgratOp<false> grat,
                                                               const int32_t *__restrict__ it_hit, int imax)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    const bool rot_window = fan_x_only(rot) && fan_rot_window(rot);   // wave-uniform
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < num; i += stride) {
        const int it = it_hit[i];
        if (it < 0) continue;  // never met a grating (reported by the search): left as it stands
        Ray r[1];
        load9(c, i, r[0]);
        double *q = r[0].c;
        const bool sel[1] = {true};
        unsigned wm[1];
        // diffract once: at the start of round it+1, or after the loop for a ray that met its grating in the last round
        // (sector.py:703-704) -- the same two operations either way
        ReflectOp{}.apply<1>(r, sel, wm);
        grat.apply<1>(r, sel, wm);
        const bool plain = rot_window && !((int)fan_special(q[X]) | (int)fan_special(q[L]) | (int)fan_special(q[UX])) &&
                           fan_pair_window(q[Y], q[Z]) && fan_pair_window(q[M], q[N]) && fan_pair_window(q[UY], q[UZ]);
        if (plain) {
#pragma unroll 2
            for (int j = it + 1; j <= imax; j++) {
                rotv<1>(q[X], q[Y], q[Z], rot.cx, rot.sx);
                rotv<1>(q[L], q[M], q[N], rot.cx, rot.sx);
                rotv<1>(q[UX], q[UY], q[UZ], rot.cx, rot.sx);
            }
        } else {
            for (int j = it + 1; j <= imax; j++) fan_rotate_full(rot, 7, r);
        }
        store9(c, i, r[0]);
    }
}

static int fan_cols(double *const *cols, Cols &c)
{
    if (!cols) return PXF_EINVAL;
    for (int col = X; col <= UZ; col++) {
        if (!cols[col]) return PXF_EINVAL;
        c.p[col] = cols[col];
    }
    return PXF_OK;
}

// Largest double a with sqrt(a) <= v (round to nearest): start at v*v and walk the neighbours (the square of a
// double is within a few ulps of the boundary).  Then  sqrt(a) > v <=> a > that.  Likewise the smallest a with
// sqrt(a) >= v.  Returns false if the walk does not settle (never for finite positive v).
static bool fan_sqrt_bounds(double lo, double hi, double &a_lo, double &a_hi)
{
    if (!(lo > 0. && hi > lo && hi < 1e150)) return false;
    double a = lo * lo;
    int guard = 0;
    while (sqrt(a) > lo && guard++ < 64) a = nextafter(a, 0.);
    while (sqrt(nextafter(a, INFINITY)) <= lo && guard++ < 64) a = nextafter(a, INFINITY);
    if (!(sqrt(a) <= lo && sqrt(nextafter(a, INFINITY)) > lo)) return false;
    a_lo = a;
    a = hi * hi;
    guard = 0;
    while (sqrt(a) < hi && guard++ < 64) a = nextafter(a, INFINITY);
    while (sqrt(nextafter(a, 0.)) >= hi && guard++ < 64) a = nextafter(a, 0.);
    if (!(sqrt(a) >= hi && sqrt(nextafter(a, 0.)) < hi)) return false;
    a_hi = a;
    return true;
}

static TransformOp<false> fan_rotation(double ang)
{
    double cs[6];
    // tran.transform(rays, 0, 0, 0, ang, 0, 0): the wrapper negates its arguments (transformations.py:29); the
    // integer zeros of that call stay +0.0
    trig6(-ang, 0., 0., false, cs);
    return TransformOp<false>{0., 0., 0., cs[0], cs[1], cs[2], cs[3], cs[4], cs[5], 7};
}

}  // namespace pxf

using namespace pxf;

extern "C" {

int pxf_fan_search(double *const *cols, int64_t num, double hubdist, double length, double ang, double thr, int32_t cap,
                   int32_t *it_hit, int32_t *imax, uint64_t *nmiss, pxf_stream_t stream)
{
    Cols c{};
    int rc = fan_cols(cols, c);
    if (rc != PXF_OK) return rc;
    if (num < 0 || cap < 0 || !it_hit || !imax || !nmiss) return PXF_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(imax, 0, sizeof(int32_t), s);
    if (e == hipSuccess) e = hipMemsetAsync(nmiss, 0, sizeof(uint64_t), s);
    if (e != hipSuccess) return hip_fail(e);
    if (num == 0) return PXF_OK;
    const double st = sin(thr);
    FanGeom g{hubdist, length + hubdist, thr, st * (1 - 1e-9), st * (1 + 1e-9), 0., 0., fan_rotation(ang), cap, 0};
    g.sq_ok = fan_sqrt_bounds(g.lo, g.hi, g.a_lo, g.a_hi) ? 1 : 0;
    hipLaunchKernelGGL(fan_search_kernel, dim3(grid_for(num)), dim3(kBlock), 0, s, c, num, g, it_hit, imax,
                       reinterpret_cast<unsigned long long *>(nmiss));
    e = hipGetLastError();
    return e == hipSuccess ? PXF_OK : hip_fail(e);
}

int pxf_fan_finish(double *const *cols, int64_t num, double ang, double dpermm, double order, double wave,
                   const int32_t *it_hit, int32_t imax, pxf_stream_t stream)
{
    Cols c{};
    int rc = fan_cols(cols, c);
    if (rc != PXF_OK) return rc;
    if (num < 0 || imax < 0 || !it_hit) return PXF_EINVAL;
    if (num == 0) return PXF_OK;
    RadgratOp<false> grat{dpermm, order, wave, 0., 0.};
    radgrat_consts(grat.S, grat.C);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(fan_finish_kernel, dim3(grid_for(num)), dim3(kBlock), 0, s, c, num, fan_rotation(ang), grat, it_hit,
                       (int)imax);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PXF_OK : hip_fail(e);
}

}  // extern "C"
