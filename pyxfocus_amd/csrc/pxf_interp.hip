// pxf_interp.hip -- scattered rays onto a grid, on the device: the `griddata` of analyses.interpolateVec
// (analyses.py:189-230), methods 'nearest' and 'linear'.
//
// The reference hands (x, y, value) of every ray to scipy.interpolate.griddata on the host: a Delaunay triangulation
// of all rays (Qhull) and, per grid node, the barycentric interpolation inside the triangle that contains it.  The
// wrapper keeps that call (bit-identical to the reference by construction); this is the on-device alternative for
// focal planes that should not leave HBM (interpolateVec(..., on_device=True)).
//
// No triangulation is built.  The Delaunay triangle that contains a query point q is the optimum of a three-variable
// linear program -- among all triples of points whose triangle contains q, the one that minimises the barycentric
// mean of the squared distances to q (the lower convex hull of the points lifted to the paraboloid, evaluated at q) --
// and a triple is optimal exactly when no other point lies inside its circumcircle.  So every query runs its own small
// simplex walk over a uniform cell grid of the points (built by the caller: points sorted by cell + cell offsets):
//   A. a first triple around q: the nearest point in each of six 60-degree sectors, found by scanning square rings of
//      cells outward; sectors (0,2,4) or (1,3,5) all occupied -> their three points enclose q.  If the rings run out
//      (q at the rim of the cloud or outside it) one exact O(n) pass decides: with a = any point, b / c = the points
//      that open the widest angle to either side of the ray q->a; q is inside the convex hull iff b and c open more
//      than 180 degrees, and then (a, b, c) encloses it.  A cheap reject comes first: the caller passes the support
//      values max_i <p_i, u_k> in 16 directions, and a query beyond any of them is outside the hull.
//   B. pivot: look for a point strictly inside the circumcircle of the current triple (the cells under the circle's
//      bounding box; the violator nearest to q is taken); none -> done.  Otherwise it replaces the vertex that keeps q
//      enclosed -- of the (up to two) such triangles the one with the lower objective -- and the walk continues.  The
//      objective falls strictly, so the walk ends; a cap (kMaxPivots) guards against an inconsistent floating-point
//      predicate on nearly co-circular points, where either triangulation is a valid Delaunay one anyway.
//   C. value = barycentric combination of the three vertex values; NaN outside the hull (griddata's fill value).
// Points in general position have ONE Delaunay triangulation, so the result equals scipy's up to the rounding of the
// barycentric weights (1e-13 relative asserted in tests/test_gpu_analyses.py); exactly co-circular points (rays on a
// regular grid) admit several, Qhull picks one of them arbitrarily, and so may this.
#include <math.h>

#include "pxf_kernel.h"

namespace pxf {

constexpr int kMaxPivots = 256;
constexpr int kSupportDirs = 16;

struct CellGrid {
    const double *xs, *ys, *vs;   // points sorted by cell
    const int32_t *cell_start;    // [gx*gy + 1]
    int64_t n;
    int gx, gy;
    double x0, y0, icx, icy;      // cell (i, j) = [x0 + i/icx, ...)
    double support[kSupportDirs];
    int has_support;
};

__device__ __forceinline__ double orient2(double ax, double ay, double bx, double by, double cx, double cy)
{
    return (bx - ax) * (cy - ay) - (by - ay) * (cx - ax);
}

// > 0: p strictly inside the circumcircle of the counter-clockwise triangle a b c (coordinates relative to p)
__device__ __forceinline__ double incircle(double ax, double ay, double bx, double by, double cx, double cy, double px,
                                           double py)
{
    const double adx = ax - px, ady = ay - py, bdx = bx - px, bdy = by - py, cdx = cx - px, cdy = cy - py;
    const double ad = adx * adx + ady * ady, bd = bdx * bdx + bdy * bdy, cd = cdx * cdx + cdy * cdy;
    return adx * (bdy * cd - bd * cdy) - ady * (bdx * cd - bd * cdx) + ad * (bdx * cdy - bdy * cdx);
}

__device__ __forceinline__ int sector6(double dx, double dy)
{
    const double h = 0.8660254037844386;   // sin 60
    int s;
    double ux = dx, uy = dy;
    const bool lower = dy < 0. || (dy == 0. && dx < 0.);
    if (lower) {
        ux = -dx;
        uy = -dy;
    }
    // upper half plane: [0,60) / [60,120) / [120,180]
    if (.5 * uy - h * ux < 0.) s = 0;           // cross(u60, d) < 0
    else if (-.5 * uy - h * ux < 0.) s = 1;     // cross(u120, d) < 0
    else s = 2;
    return lower ? s + 3 : s;
}

struct Tri {
    int64_t i[3];
    double x[3], y[3];
};

// barycentric weights of q in the triangle (not normalised): w[k] >= 0 for all k <=> q inside (ccw triangle)
__device__ __forceinline__ void bary(const Tri &t, double qx, double qy, double (&w)[3])
{
    w[0] = orient2(qx, qy, t.x[1], t.y[1], t.x[2], t.y[2]);
    w[1] = orient2(t.x[0], t.y[0], qx, qy, t.x[2], t.y[2]);
    w[2] = orient2(t.x[0], t.y[0], t.x[1], t.y[1], qx, qy);
}
__device__ __forceinline__ void make_ccw(Tri &t)
{
    if (orient2(t.x[0], t.y[0], t.x[1], t.y[1], t.x[2], t.y[2]) < 0.) {
        const int64_t i = t.i[1];
        const double x = t.x[1], y = t.y[1];
        t.i[1] = t.i[2];
        t.x[1] = t.x[2];
        t.y[1] = t.y[2];
        t.i[2] = i;
        t.x[2] = x;
        t.y[2] = y;
    }
}
__device__ __forceinline__ bool encloses(const Tri &t, double qx, double qy, double &objective)
{
    double w[3];
    bary(t, qx, qy, w);
    const double area = w[0] + w[1] + w[2];
    if (!(area > 0.) || w[0] < 0. || w[1] < 0. || w[2] < 0.) return false;
    double o = 0.;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const double dx = t.x[k] - qx, dy = t.y[k] - qy;
        o += w[k] * (dx * dx + dy * dy);
    }
    objective = o / area;
    return true;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

template <int METHOD>   // 0 nearest, 1 linear
__global__ __launch_bounds__(kBlock, 2) void griddata_kernel(CellGrid g, const double *__restrict__ qxs,
                                                             const double *__restrict__ qys, int64_t m,
                                                             double *__restrict__ out)
{
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t iq = (int64_t)blockIdx.x * kBlock + threadIdx.x; iq < m; iq += stride) {
        const double qx = qxs[iq], qy = qys[iq];
        if (!(qx == qx) || !(qy == qy) || g.n == 0) {
            out[iq] = nan;
            continue;
        }
        const int cqx = clampi((int)floor((qx - g.x0) * g.icx), 0, g.gx - 1);
        const int cqy = clampi((int)floor((qy - g.y0) * g.icy), 0, g.gy - 1);
        // ---- A. ring search: nearest point overall, nearest point per sector ---------------------------------------
        int64_t sidx[6];
        double sd2[6], sx[6], sy[6];
#pragma unroll
        for (int s = 0; s < 6; s++) {
            sidx[s] = -1;
            sd2[s] = INFINITY;
            sx[s] = sy[s] = 0.;
        }
        int64_t near = -1;
        double near_d2 = INFINITY;
        bool hit_exact = false;
        const int rmax = g.gx > g.gy ? g.gx : g.gy;
        bool have = false;
        int after = 0;    // one more ring is scanned after the sectors first fill up (nearer candidates may follow)
        for (int r = 0; r <= rmax && !hit_exact; r++) {
            const int j0 = cqy - r, j1 = cqy + r, i0 = cqx - r, i1 = cqx + r;
            for (int j = j0; j <= j1; j++) {
                if (j < 0 || j >= g.gy) continue;
                const bool edge_row = (j == j0 || j == j1);
                const int step = edge_row ? 1 : (i1 - i0 > 0 ? i1 - i0 : 1);
                for (int i = i0; i <= i1; i += step) {
                    if (i < 0 || i >= g.gx) continue;
                    const int c = j * g.gx + i;
                    for (int64_t p = g.cell_start[c]; p < g.cell_start[c + 1]; p++) {
                        const double px = g.xs[p], py = g.ys[p];
                        const double dx = px - qx, dy = py - qy, d2 = dx * dx + dy * dy;
                        if (d2 < near_d2) {
                            near_d2 = d2;
                            near = p;
                        }
                        if (d2 == 0.) {
                            hit_exact = true;
                            continue;
                        }
                        const int s = sector6(dx, dy);
#pragma unroll
                        for (int k = 0; k < 6; k++) {
                            if (k == s && d2 < sd2[k]) {
                                sd2[k] = d2;
                                sidx[k] = p;
                                sx[k] = px;
                                sy[k] = py;
                            }
                        }
                    }
                }
            }
            if (METHOD == 0) {
                // the nearest point is final once the ring distance exceeds it: ring r covers every point within r cells
                if (near >= 0) {
                    const double reach = (double)r / (g.icx > g.icy ? g.icx : g.icy);
                    if (reach * reach >= near_d2) break;
                }
                continue;
            }
            have = (sidx[0] >= 0 && sidx[2] >= 0 && sidx[4] >= 0) || (sidx[1] >= 0 && sidx[3] >= 0 && sidx[5] >= 0);
            if (have && after++ >= 1) break;
        }
        if (METHOD == 0) {
            out[iq] = near >= 0 ? g.vs[near] : nan;
            continue;
        }
        if (hit_exact) {
            out[iq] = g.vs[near];
            continue;
        }
        Tri t;
        bool ok = false;
        double obj = INFINITY;
        if (have) {
#pragma unroll
            for (int par = 0; par < 2; par++) {
                if (sidx[par] >= 0 && sidx[par + 2] >= 0 && sidx[par + 4] >= 0) {
                    Tri u;
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        u.i[k] = sidx[par + 2 * k];
                        u.x[k] = sx[par + 2 * k];
                        u.y[k] = sy[par + 2 * k];
                    }
                    make_ccw(u);
                    double o;
                    if (encloses(u, qx, qy, o) && o < obj) {
                        t = u;
                        obj = o;
                        ok = true;
                    }
                }
            }
        }
        if (!ok) {
            // quick reject: beyond a supporting line of the cloud
            bool outside = false;
            if (g.has_support) {
#pragma unroll
                for (int k = 0; k < kSupportDirs; k++) {
                    double sn, cs;
                    sincos(6.283185307179586 * k / kSupportDirs, &sn, &cs);
                    // (directions and maxima come from the host's libm / torch: a margin keeps the reject conservative)
                    outside |= (qx * cs + qy * sn) > g.support[k] + 1e-9 * (fabs(g.support[k]) + 1.);
                }
            }
            if (outside) {
                out[iq] = nan;
                continue;
            }
            // exact: a = the nearest point; b / c = widest opening to the left / right of the ray q -> a
            if (near < 0) {
                out[iq] = nan;
                continue;
            }
            const double ax = g.xs[near] - qx, ay = g.ys[near] - qy;
            int64_t bi = -1, ci = -1;
            double bx = 0., by = 0., cx = 0., cy = 0.;
            for (int64_t p = 0; p < g.n; p++) {
                const double px = g.xs[p] - qx, py = g.ys[p] - qy;
                const double cr = ax * py - ay * px;
                if (cr > 0.) {   // to the left: keep the one furthest counter-clockwise
                    if (bi < 0 || bx * py - by * px > 0.) {
                        bi = p;
                        bx = px;
                        by = py;
                    }
                } else if (cr < 0.) {   // to the right: furthest clockwise
                    if (ci < 0 || cx * py - cy * px < 0.) {
                        ci = p;
                        cx = px;
                        cy = py;
                    }
                } else if (ax * px + ay * py < 0.) {   // exactly opposite a: q lies on the segment a-p
                    bi = ci = p;
                    bx = cx = px;
                    by = cy = py;
                    break;
                }
            }
            if (bi < 0 || ci < 0) {
                out[iq] = nan;
                continue;
            }
            if (bi == ci) {   // q on the segment between two points: interpolate along it
                const double la = sqrt(ax * ax + ay * ay), lb = sqrt(bx * bx + by * by);
                out[iq] = (g.vs[near] * lb + g.vs[bi] * la) / (la + lb);
                continue;
            }
            t.i[0] = near;
            t.x[0] = ax + qx;
            t.y[0] = ay + qy;
            t.i[1] = bi;
            t.x[1] = bx + qx;
            t.y[1] = by + qy;
            t.i[2] = ci;
            t.x[2] = cx + qx;
            t.y[2] = cy + qy;
            make_ccw(t);
            if (!encloses(t, qx, qy, obj)) {   // b and c open less than 180 degrees: outside the hull
                out[iq] = nan;
                continue;
            }
        }
        // ---- B. pivots ----------------------------------------------------------------------------------------------
        for (int piv = 0; piv < kMaxPivots; piv++) {
            // circumcircle (relative to vertex 0) -> cells to visit
            const double bx = t.x[1] - t.x[0], by = t.y[1] - t.y[0], cx = t.x[2] - t.x[0], cy = t.y[2] - t.y[0];
            const double dd = 2. * (bx * cy - by * cx);
            const double b2 = bx * bx + by * by, c2 = cx * cx + cy * cy;
            const double ux = (cy * b2 - by * c2) / dd, uy = (bx * c2 - cx * b2) / dd;
            const double rad = sqrt(ux * ux + uy * uy), ccx = t.x[0] + ux, ccy = t.y[0] + uy;
            int i0 = 0, i1 = g.gx - 1, j0 = 0, j1 = g.gy - 1;
            if (rad == rad && rad < INFINITY) {
                i0 = clampi((int)fmax(fmin(floor((ccx - rad - g.x0) * g.icx), 2e9), -2e9), 0, g.gx - 1);
                i1 = clampi((int)fmax(fmin(floor((ccx + rad - g.x0) * g.icx), 2e9), -2e9), 0, g.gx - 1);
                j0 = clampi((int)fmax(fmin(floor((ccy - rad - g.y0) * g.icy), 2e9), -2e9), 0, g.gy - 1);
                j1 = clampi((int)fmax(fmin(floor((ccy + rad - g.y0) * g.icy), 2e9), -2e9), 0, g.gy - 1);
            }
            int64_t vi = -1;
            double vx = 0., vy = 0., vd2 = INFINITY;
            for (int j = j0; j <= j1; j++) {
                const int64_t p0 = g.cell_start[j * g.gx + i0], p1 = g.cell_start[j * g.gx + i1 + 1];   // a row is contiguous
                for (int64_t p = p0; p < p1; p++) {
                    if (p == t.i[0] || p == t.i[1] || p == t.i[2]) continue;
                    const double px = g.xs[p], py = g.ys[p];
                    if (incircle(t.x[0], t.y[0], t.x[1], t.y[1], t.x[2], t.y[2], px, py) > 0.) {
                        const double dx = px - qx, dy = py - qy, d2 = dx * dx + dy * dy;
                        if (d2 < vd2) {
                            vd2 = d2;
                            vi = p;
                            vx = px;
                            vy = py;
                        }
                    }
                }
            }
            if (vi < 0) break;   // empty circumcircle: the Delaunay triangle
            // the violator replaces the vertex that keeps q enclosed; of several such triangles the lowest objective
            Tri best;
            double bo = INFINITY;
            bool any = false;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                Tri u = t;
                u.i[k] = vi;
                u.x[k] = vx;
                u.y[k] = vy;
                make_ccw(u);
                double o;
                if (encloses(u, qx, qy, o) && o < bo) {
                    best = u;
                    bo = o;
                    any = true;
                }
            }
            if (!any || !(bo < obj)) break;   // (rounding on a nearly co-circular set: the current triangle stands)
            t = best;
            obj = bo;
        }
        // ---- C. barycentric value -------------------------------------------------------------------------------------
        double w[3];
        bary(t, qx, qy, w);
        const double area = w[0] + w[1] + w[2];
        out[iq] = (w[0] * g.vs[t.i[0]] + w[1] * g.vs[t.i[1]] + w[2] * g.vs[t.i[2]]) / area;
    }
}

}  // namespace pxf

using namespace pxf;

extern "C" {

int pxf_griddata(const double *xs, const double *ys, const double *vs, int64_t n, const int32_t *cell_start, int32_t gx,
                 int32_t gy, double x0, double y0, double inv_cell_x, double inv_cell_y, const double *support16,
                 const double *qx, const double *qy, int64_t m, int32_t method, double *out, pxf_stream_t stream)
{
    if (n < 0 || n > 0x7fffffffll || m < 0 || gx <= 0 || gy <= 0 || (int64_t)gx * gy > (1ll << 30) ||
        (method != 0 && method != 1))
        return PXF_EINVAL;
    if (m == 0) return PXF_OK;
    if (!out || !qx || !qy || !cell_start || (n > 0 && (!xs || !ys || !vs))) return PXF_EINVAL;
    if (!(inv_cell_x > 0.) || !(inv_cell_y > 0.)) return PXF_EINVAL;
    if (!device_usable()) return PXF_ENODEV;
    CellGrid g{xs, ys, vs, cell_start, n, gx, gy, x0, y0, inv_cell_x, inv_cell_y, {}, support16 ? 1 : 0};
    for (int k = 0; k < kSupportDirs; k++) g.support[k] = support16 ? support16[k] : 0.;
    hipStream_t s = (hipStream_t)stream;
    const int grid = (int)((m + kBlock - 1) / kBlock < (int64_t)device_cus() * 32 ? (m + kBlock - 1) / kBlock
                                                                               : (int64_t)device_cus() * 32);
    if (method == 0)
        hipLaunchKernelGGL(griddata_kernel<0>, dim3(grid), dim3(kBlock), 0, s, g, qx, qy, m, out);
    else
        hipLaunchKernelGGL(griddata_kernel<1>, dim3(grid), dim3(kBlock), 0, s, g, qx, qy, m, out);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PXF_OK : hip_fail(e);
}

}  // extern "C"
