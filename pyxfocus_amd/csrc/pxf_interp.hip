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
//      than 180 degrees, and then (a, b, c) encloses it.  A cheap reject comes before everything: the caller passes the
//      support values max_i <p_i, u_k> in 64 directions, and a query beyond any of them is outside the hull.  The rings
//      are capped (kMaxRings): interior nodes fill their sectors within two or three.
//   B. pivot: look for a point strictly inside the circumcircle of the current triple (the cells under the circle's
//      bounding box; the violator nearest to q is taken); none -> done.  Otherwise it replaces the vertex that keeps q
//      enclosed -- of the (up to two) such triangles the one with the lower objective -- and the walk continues.  The
//      objective falls strictly, so the walk ends; a cap (kMaxPivots) guards against an inconsistent floating-point
//      predicate on nearly co-circular points, where either triangulation is a valid Delaunay one anyway.
//   C. value = barycentric combination of the three vertex values; NaN outside the hull (griddata's fill value).
// Two kernels.  `griddata_kernel`: one thread per node, bounded work (capped rings, circumcircles that cover at most
// kEasyScan points); the few nodes that need more -- at the rim of the cloud, where the exact pass over all points and
// circumcircles as large as the cloud live -- are queued and handled by `griddata_hard_kernel`, one WORKGROUP per node:
// the same steps with every scan spread over the 256 threads (a lone thread walking 1e7 points is latency, not work).
// Points in general position have ONE Delaunay triangulation, so the result equals scipy's up to the rounding of the
// barycentric weights (1e-13 relative asserted in tests/test_gpu_analyses.py); exactly co-circular points (rays on a
// regular grid) admit several, Qhull picks one of them arbitrarily, and so may this.
#include <math.h>

#include "pxf_kernel.h"

namespace pxf {

constexpr int kMaxPivots = 256;
constexpr int kSupportDirs = 64;
constexpr int kMaxRings = 24;
constexpr int64_t kEasyScan = 2048; // points under a circumcircle's bounding box one thread will still scan by itself      // a node whose sectors are not filled this far out is settled by the exact pass

struct CellGrid {
    const double *xs, *ys, *vs;   // points sorted by cell
    const int32_t *cell_start;    // [gx*gy + 1]
    int64_t n;
    int gx, gy;
    double x0, y0, icx, icy;      // cell (i, j) = [x0 + i/icx, ...)
    double support[kSupportDirs], dirc[kSupportDirs], dirs[kSupportDirs];   // max_i <p_i, u_k>, u_k = (dirc, dirs)
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

// cells under the circumcircle of t (all of them if the circle is degenerate) and the number of points in their rows
__device__ __forceinline__ int64_t circle_rows(const Tri &t, const CellGrid &g, int &i0, int &i1, int &j0, int &j1)
{
    const double bx = t.x[1] - t.x[0], by = t.y[1] - t.y[0], cx = t.x[2] - t.x[0], cy = t.y[2] - t.y[0];
    const double dd = 2. * (bx * cy - by * cx);
    const double b2 = bx * bx + by * by, c2 = cx * cx + cy * cy;
    const double ux = (cy * b2 - by * c2) / dd, uy = (bx * c2 - cx * b2) / dd;
    const double rad = sqrt(ux * ux + uy * uy), ccx = t.x[0] + ux, ccy = t.y[0] + uy;
    i0 = 0, i1 = g.gx - 1, j0 = 0, j1 = g.gy - 1;
    if (rad == rad && rad < INFINITY) {
        i0 = clampi((int)fmax(fmin(floor((ccx - rad - g.x0) * g.icx), 2e9), -2e9), 0, g.gx - 1);
        i1 = clampi((int)fmax(fmin(floor((ccx + rad - g.x0) * g.icx), 2e9), -2e9), 0, g.gx - 1);
        j0 = clampi((int)fmax(fmin(floor((ccy - rad - g.y0) * g.icy), 2e9), -2e9), 0, g.gy - 1);
        j1 = clampi((int)fmax(fmin(floor((ccy + rad - g.y0) * g.icy), 2e9), -2e9), 0, g.gy - 1);
    }
    int64_t cnt = 0;
    for (int j = j0; j <= j1 && cnt <= kEasyScan; j++)
        cnt += g.cell_start[j * g.gx + i1 + 1] - g.cell_start[j * g.gx + i0];   // a row of cells is contiguous
    return cnt;
}

// the violator (vi, vx, vy) replaces the vertex that keeps q enclosed; of several such triangles the lowest objective.
// false: no progress (rounding on a nearly co-circular set): the current triangle stands
__device__ __forceinline__ bool pivot(Tri &t, double &obj, int64_t vi, double vx, double vy, double qx, double qy)
{
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
    if (!any || !(bo < obj)) return false;
    t = best;
    obj = bo;
    return true;
}

__device__ __forceinline__ double tri_value(const Tri &t, const CellGrid &g, double qx, double qy)
{
    double w[3];
    bary(t, qx, qy, w);
    return (w[0] * g.vs[t.i[0]] + w[1] * g.vs[t.i[1]] + w[2] * g.vs[t.i[2]]) / (w[0] + w[1] + w[2]);
}

// queue[0] = number of queued nodes, queue[1..] = their indices
__device__ __forceinline__ void defer(int64_t *queue, int64_t iq)
{
    const unsigned long long k = atomicAdd(reinterpret_cast<unsigned long long *>(queue), 1ull);
    queue[1 + k] = iq;
}

template <int METHOD>   // 0 nearest, 1 linear
__global__ __launch_bounds__(kBlock, 2) void griddata_kernel(CellGrid g, const double *__restrict__ qxs,
                                                             const double *__restrict__ qys, int64_t m,
                                                             double *__restrict__ out, int64_t *__restrict__ queue)
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
        if (METHOD == 1 && g.has_support) {
            // beyond a supporting line of the cloud: outside its convex hull, nothing to search for
            // (directions and maxima come from the host's libm / torch: a margin keeps the reject conservative)
            bool outside = false;
#pragma unroll 4
            for (int k = 0; k < kSupportDirs; k++)
                outside |= (qx * g.dirc[k] + qy * g.dirs[k]) > g.support[k] + 1e-9 * (fabs(g.support[k]) + 1.);
            if (outside) {
                out[iq] = nan;
                continue;
            }
        }
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
        const int rall = g.gx > g.gy ? g.gx : g.gy;
        const int rmax = (METHOD == 1 && rall > kMaxRings) ? kMaxRings : rall;
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
        if (!ok) {   // at the rim of the cloud (or outside it): the exact pass, by a whole workgroup
            defer(queue, iq);
            continue;
        }
        // ---- B. pivots ----------------------------------------------------------------------------------------------
        bool deferred = false;
        for (int piv = 0; piv < kMaxPivots; piv++) {
            int i0, i1, j0, j1;
            if (circle_rows(t, g, i0, i1, j0, j1) > kEasyScan) {   // a circle as large as the cloud: not for one thread
                defer(queue, iq);
                deferred = true;
                break;
            }
            int64_t vi = -1;
            double vx = 0., vy = 0., vd2 = INFINITY;
            for (int j = j0; j <= j1; j++) {
                const int64_t p0 = g.cell_start[j * g.gx + i0], p1 = g.cell_start[j * g.gx + i1 + 1];
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
            if (!pivot(t, obj, vi, vx, vy, qx, qy)) break;
        }
        if (!deferred) out[iq] = tri_value(t, g, qx, qy);   // ---- C.
    }
}

// ---- the deferred nodes: one workgroup each, every scan spread over its threads -------------------------------------
struct Cand {   // a candidate point of a block-wide search; idx < 0: none
    double key, x, y;
    int64_t idx;
};
// block-wide reduction of one candidate per thread; `better(a, b)`: a is to be preferred to b (both valid).
// Ties go to the lower index, so the result does not depend on the thread layout.  All threads get the winner.
template <class Better>
__device__ __forceinline__ Cand block_best(Cand c, Better better)
{
    __shared__ Cand s_c[kBlock];
    s_c[threadIdx.x] = c;
    __syncthreads();
    for (int d = kBlock / 2; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) {
            const Cand a = s_c[threadIdx.x], b = s_c[threadIdx.x + d];
            const bool take_b = b.idx >= 0 && (a.idx < 0 || better(b, a) || (!better(a, b) && b.idx < a.idx));
            if (take_b) s_c[threadIdx.x] = b;
        }
        __syncthreads();
    }
    const Cand r = s_c[0];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(kBlock, 2) void griddata_hard_kernel(CellGrid g, const double *__restrict__ qxs,
                                                                  const double *__restrict__ qys,
                                                                  double *__restrict__ out,
                                                                  const int64_t *__restrict__ queue)
{
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    const int64_t count = queue[0];
    for (int64_t h = blockIdx.x; h < count; h += gridDim.x) {   // (block-uniform: every thread takes every branch alike)
        const int64_t iq = queue[1 + h];
        const double qx = qxs[iq], qy = qys[iq];
        // a = the point nearest to q
        Cand mine{INFINITY, 0., 0., -1};
        for (int64_t p = threadIdx.x; p < g.n; p += kBlock) {
            const double px = g.xs[p], py = g.ys[p], dx = px - qx, dy = py - qy, d2 = dx * dx + dy * dy;
            if (d2 < mine.key) mine = Cand{d2, px, py, p};
        }
        const Cand a = block_best(mine, [](const Cand &u, const Cand &v) { return u.key < v.key; });
        if (a.key == 0.) {
            if (threadIdx.x == 0) out[iq] = g.vs[a.idx];
            continue;
        }
        const double ax = a.x - qx, ay = a.y - qy;
        // b / c = widest opening to the left / right of the ray q -> a; a point exactly opposite a puts q on a segment
        Cand lb{0., 0., 0., -1}, rc{0., 0., 0., -1}, opp{0., 0., 0., -1};
        for (int64_t p = threadIdx.x; p < g.n; p += kBlock) {
            const double px = g.xs[p] - qx, py = g.ys[p] - qy;
            const double cr = ax * py - ay * px;
            if (cr > 0.) {
                if (lb.idx < 0 || lb.x * py - lb.y * px > 0.) lb = Cand{0., px, py, p};
            } else if (cr < 0.) {
                if (rc.idx < 0 || rc.x * py - rc.y * px < 0.) rc = Cand{0., px, py, p};
            } else if (ax * px + ay * py < 0. && opp.idx < 0) {
                opp = Cand{0., px, py, p};
            }
        }
        const Cand o = block_best(opp, [](const Cand &, const Cand &) { return false; });
        if (o.idx >= 0) {   // q on the segment between a and o: interpolate along it
            if (threadIdx.x == 0) {
                const double la = sqrt(ax * ax + ay * ay), lo = sqrt(o.x * o.x + o.y * o.y);
                out[iq] = (g.vs[a.idx] * lo + g.vs[o.idx] * la) / (la + lo);
            }
            continue;
        }
        const Cand b = block_best(lb, [](const Cand &u, const Cand &v) { return v.x * u.y - v.y * u.x > 0.; });   // u further ccw than v
        const Cand c = block_best(rc, [](const Cand &u, const Cand &v) { return v.x * u.y - v.y * u.x < 0.; });   // u further cw than v
        if (b.idx < 0 || c.idx < 0) {
            if (threadIdx.x == 0) out[iq] = nan;
            continue;
        }
        Tri t;
        t.i[0] = a.idx;
        t.x[0] = a.x;
        t.y[0] = a.y;
        t.i[1] = b.idx;
        t.x[1] = b.x + qx;
        t.y[1] = b.y + qy;
        t.i[2] = c.idx;
        t.x[2] = c.x + qx;
        t.y[2] = c.y + qy;
        // (the vertices are re-read so that their coordinates are the stored ones, not a difference added back)
#pragma unroll
        for (int k = 1; k < 3; k++) {
            t.x[k] = g.xs[t.i[k]];
            t.y[k] = g.ys[t.i[k]];
        }
        make_ccw(t);
        double obj;
        if (!encloses(t, qx, qy, obj)) {   // b and c open less than 180 degrees: outside the hull
            if (threadIdx.x == 0) out[iq] = nan;
            continue;
        }
        for (int piv = 0; piv < kMaxPivots; piv++) {
            int i0, i1, j0, j1;
            circle_rows(t, g, i0, i1, j0, j1);
            Cand v{INFINITY, 0., 0., -1};
            for (int j = j0; j <= j1; j++) {
                const int64_t p0 = g.cell_start[j * g.gx + i0], p1 = g.cell_start[j * g.gx + i1 + 1];
                for (int64_t p = p0 + threadIdx.x; p < p1; p += kBlock) {
                    if (p == t.i[0] || p == t.i[1] || p == t.i[2]) continue;
                    const double px = g.xs[p], py = g.ys[p];
                    if (incircle(t.x[0], t.y[0], t.x[1], t.y[1], t.x[2], t.y[2], px, py) > 0.) {
                        const double dx = px - qx, dy = py - qy, d2 = dx * dx + dy * dy;
                        if (d2 < v.key) v = Cand{d2, px, py, p};
                    }
                }
            }
            const Cand w = block_best(v, [](const Cand &u, const Cand &z) { return u.key < z.key; });
            if (w.idx < 0) break;
            if (!pivot(t, obj, w.idx, w.x, w.y, qx, qy)) break;
        }
        if (threadIdx.x == 0) out[iq] = tri_value(t, g, qx, qy);
    }
}

}  // namespace pxf

using namespace pxf;

extern "C" {

size_t pxf_griddata_workspace_bytes(int64_t m) { return (size_t)(m > 0 ? m + 1 : 1) * sizeof(int64_t); }

int pxf_griddata(const double *xs, const double *ys, const double *vs, int64_t n, const int32_t *cell_start, int32_t gx,
                 int32_t gy, double x0, double y0, double inv_cell_x, double inv_cell_y, const double *support,
                 const double *qx, const double *qy, int64_t m, int32_t method, double *out, void *workspace,
                 pxf_stream_t stream)
{
    if (n < 0 || n > 0x7fffffffll || m < 0 || gx <= 0 || gy <= 0 || (int64_t)gx * gy > (1ll << 30) ||
        (method != 0 && method != 1))
        return PXF_EINVAL;
    if (m == 0) return PXF_OK;
    if (!out || !qx || !qy || !cell_start || !workspace || (n > 0 && (!xs || !ys || !vs))) return PXF_EINVAL;
    if (!(inv_cell_x > 0.) || !(inv_cell_y > 0.)) return PXF_EINVAL;
    if (!device_usable()) return PXF_ENODEV;
    CellGrid g{xs, ys, vs, cell_start, n, gx, gy, x0, y0, inv_cell_x, inv_cell_y, {}, {}, {}, support ? 1 : 0};
    for (int k = 0; k < kSupportDirs; k++) {
        g.support[k] = support ? support[k] : 0.;
        g.dirc[k] = cos(6.283185307179586 * k / kSupportDirs);
        g.dirs[k] = sin(6.283185307179586 * k / kSupportDirs);
    }
    hipStream_t s = (hipStream_t)stream;
    int64_t *queue = reinterpret_cast<int64_t *>(workspace);
    hipError_t e = hipMemsetAsync(queue, 0, sizeof(int64_t), s);
    if (e != hipSuccess) return hip_fail(e);
    const int grid = (int)((m + kBlock - 1) / kBlock < (int64_t)device_cus() * 32 ? (m + kBlock - 1) / kBlock
                                                                               : (int64_t)device_cus() * 32);
    if (method == 0) {
        hipLaunchKernelGGL(griddata_kernel<0>, dim3(grid), dim3(kBlock), 0, s, g, qx, qy, m, out, queue);
    } else {
        hipLaunchKernelGGL(griddata_kernel<1>, dim3(grid), dim3(kBlock), 0, s, g, qx, qy, m, out, queue);
        // the deferred nodes (their number is on the device: the workgroups read it); enough workgroups to fill the chip
        const int hard = (int)(m < (int64_t)device_cus() * 8 ? m : (int64_t)device_cus() * 8);
        hipLaunchKernelGGL(griddata_hard_kernel, dim3(hard), dim3(kBlock), 0, s, g, qx, qy, out, queue);
    }
    e = hipGetLastError();
    return e == hipSuccess ? PXF_OK : hip_fail(e);
}

}  // extern "C"
