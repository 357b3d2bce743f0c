// pxf_reduce.hip -- focal-plane reductions behind analyses.centroid / rmsCentroid / rmsX / rmsY /
// analyticImagePlane / analyticXPlane / analyticYPlane (analyses.py:16-58, 118-155) and
// transformations.steerX/Y (transformations.py:79-89).
//
// The reference evaluates each np.average as its own pass over the columns (up to 9 passes for
// analyticImagePlane).  Here ONE pass over x,y,l,m,n(,w) yields all twelve weighted sums; the
// scalar algebra that combines them stays on the host, in the reference's formula order.
// Two-level deterministic tree: per-thread serial sums over a grid-stride range -> 64-lane
// shuffle tree -> LDS across the 4 waves -> one partial row per workgroup -> a single workgroup
// folds the rows in fixed order.  No atomics: bit-reproducible for a given (num, CU count).
// (np.average uses pairwise summation; agreement is to rounding, ~1e-15 relative per sum.)
#include "pxf_kernel.h"
#include "pxf_ops_surfaces.h"
#include "pxf_ops_transforms.h"

namespace pxf {

constexpr int kRedMaxBlocks = 256 * kMaxBlocksPerCU;

template <int K>
__device__ __forceinline__ void block_fold(double (&acc)[K], double *__restrict__ row)
{
    __shared__ double s_part[kBlock / 64][K];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; k++) {
        double v = acc[k];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d, 64);
        if (lane == 0) s_part[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < K) {
        double v = 0.;
#pragma unroll
        for (int w = 0; w < kBlock / 64; w++) v += s_part[w][threadIdx.x];
        row[threadIdx.x] = v;
    }
}

// the twelve sums of include/pxf.h for one ray (shared by moments_kernel and focus_step_kernel: one order of operations)
template <bool HASDIR>
__device__ __forceinline__ void mom_accum(double (&acc)[PXF_NMOM], double ww, double x, double y, double l, double m,
                                          double n)
{
    acc[0] += ww;
    acc[1] += ww * x;
    acc[2] += ww * y;
    if (HASDIR) {
        const double ln = l / n, mn = m / n;
        acc[3] += ww * ln;
        acc[4] += ww * mn;
        acc[5] += ww * (x * l / n);  // x*l/n as written in analyses.py:123
        acc[6] += ww * (y * m / n);
        acc[7] += ww * (ln * ln);
        acc[8] += ww * (mn * mn);
        acc[9] += ww * l;
        acc[10] += ww * m;
        acc[11] += ww * n;
    }
}

template <int V, bool HASDIR, bool HASW>
__global__ __launch_bounds__(kBlock, 2) void moments_kernel(const double *__restrict__ x, const double *__restrict__ y,
                                                         const double *__restrict__ l, const double *__restrict__ m,
                                                         const double *__restrict__ n, const double *__restrict__ w,
                                                         int64_t first, int64_t nvec, double *__restrict__ partial)
{
    double acc[PXF_NMOM];
#pragma unroll
    for (int k = 0; k < PXF_NMOM; k++) acc[k] = 0.;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < nvec; v += stride) {
        const int64_t vv = first + v;
        double X[V], Y[V], Lc[V], Mc[V], Nc[V], W[V];
        load_col<V>(x, vv, X);
        load_col<V>(y, vv, Y);
        if (HASDIR) {
            load_col<V>(l, vv, Lc);
            load_col<V>(m, vv, Mc);
            load_col<V>(n, vv, Nc);
        }
        if (HASW) load_col<V>(w, vv, W);
#pragma unroll
        for (int k = 0; k < V; k++) mom_accum<HASDIR>(acc, HASW ? W[k] : 1., X[k], Y[k], Lc[k], Mc[k], Nc[k]);
    }
    block_fold<PXF_NMOM>(acc, partial + (int64_t)blockIdx.x * PXF_NMOM);
}

// ---- one step of surfaces.focus (surfaces.py:502-510) as ONE pass ---------------------------------------------------
// focus() is twice {fn(rays) -> transform(rays,0,0,dz,0,0,0) -> flat(rays)} with fn one of the analytic best-focus
// formulas, i.e. per ray: a moments pass (48 B), a frame change (80 B: nine numbers read, z rewritten), a flat (96 B),
// and the same again: 448 B/ray in six launches.  The frame change and the flat touch the same six numbers and the
// second fn() only needs the twelve sums of what the flat has just produced, so a step is ONE kernel: load x y z l m n
// (+w, +opd), run TransformOp and FlatOp -- the SAME apply() bodies the stand-alone launches run, in the same order,
// so every ray comes out with the same bits -- store x y z ux uy uz (and l m n only where the rotation arithmetic
// changed them: never for ordinary rays), and fold the moments of the result in the epilogue: 104 B/ray with the sums,
// 96 without.  The normal's trip through the frame change is dead (flat overwrites it) and is skipped.  The sums are
// accumulated by the same threads over the same rays in the same order as moments_kernel (same capped grid, same
// V = 2 / V = 1 split, same fold), so they equal pxf_moments() of the result bit for bit.
template <int V, bool HASW, bool WOPD, bool MOM, bool NT>
__global__ __launch_bounds__(kBlock, 2) void focus_step_kernel(Cols c, const double *__restrict__ w, int64_t first,
                                                               int64_t nvec, TransformOp<false> top, FlatOp<WOPD> fop,
                                                               double *__restrict__ partial)
{
    double acc[PXF_NMOM];
#pragma unroll
    for (int k = 0; k < PXF_NMOM; k++) acc[k] = 0.;
    constexpr unsigned RD = POS | DIR | (WOPD ? B(OPD) : 0u);
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < nvec; v += stride) {
        const int64_t vv = first + v;
        Ray r[V];
        double W[V];
        bool sel[V];
        unsigned wt[V], wf[V];
#pragma unroll
        for (int col = 0; col < NCOL; col++) {
            if (RD & B(col)) {
                double t[V];
                load_col<V, NT>(c.p[col], vv, t);
#pragma unroll
                for (int k = 0; k < V; k++) r[k].c[col] = t[k];
            }
        }
        if (HASW && MOM) load_col<V, NT>(w, vv, W);
#pragma unroll
        for (int k = 0; k < V; k++) {
            sel[k] = true;
            wt[k] = wf[k] = 0;
            r[k].c[UX] = r[k].c[UY] = r[k].c[UZ] = 0.;   // (never read: the frame change skips the normal group)
        }
        top.template apply<V>(r, sel, wt);
        fop.template apply<V>(r, sel, wf);
#pragma unroll
        for (int col = 0; col < NCOL; col++) {
            if ((FlatOp<WOPD>::WR | DIR) & B(col)) {
                double t[V];
                bool wr[V];
#pragma unroll
                for (int k = 0; k < V; k++) {
                    t[k] = r[k].c[col];
                    wr[k] = ((wt[k] | wf[k]) & B(col)) != 0;
                }
                store_col<V, NT>(c.p[col], vv, t, wr);
            }
        }
        if (MOM) {
#pragma unroll
            for (int k = 0; k < V; k++)
                mom_accum<true>(acc, HASW ? W[k] : 1., r[k].c[X], r[k].c[Y], r[k].c[L], r[k].c[M], r[k].c[N]);
        }
    }
    if (MOM) block_fold<PXF_NMOM>(acc, partial + (int64_t)blockIdx.x * PXF_NMOM);
}

template <int V, bool HASW>
__global__ __launch_bounds__(kBlock, 2) void central_kernel(const double *__restrict__ x, const double *__restrict__ y,
                                                         const double *__restrict__ w, int64_t first, int64_t nvec,
                                                         double cx, double cy, double *__restrict__ partial)
{
    double acc[4] = {0., 0., 0., 0.};
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < nvec; v += stride) {
        const int64_t vv = first + v;
        double X[V], Y[V], W[V];
        load_col<V>(x, vv, X);
        load_col<V>(y, vv, Y);
        if (HASW) load_col<V>(w, vv, W);
#pragma unroll
        for (int k = 0; k < V; k++) {
            const double ww = HASW ? W[k] : 1.;
            const double dx = X[k] - cx, dy = Y[k] - cy;
            acc[0] += ww;
            acc[1] += ww * (dx * dx + dy * dy);
            acc[2] += ww * (dx * dx);
            acc[3] += ww * (dy * dy);
        }
    }
    block_fold<4>(acc, partial + (int64_t)blockIdx.x * 4);
}

// rows[0..nrows) of K doubles -> out[K]; `accumulate` adds to what is already there (tail launch)
template <int K>
__global__ __launch_bounds__(kBlock, 2) void fold_rows_kernel(const double *__restrict__ rows, int nrows,
                                                           double *__restrict__ out, int accumulate)
{
    double acc[K];
#pragma unroll
    for (int k = 0; k < K; k++) acc[k] = 0.;
    for (int r = threadIdx.x; r < nrows; r += kBlock) {
#pragma unroll
        for (int k = 0; k < K; k++) acc[k] += rows[(int64_t)r * K + k];
    }
    __shared__ double s_out[K];
    block_fold<K>(acc, s_out);
    __syncthreads();
    if (threadIdx.x < K) out[threadIdx.x] = (accumulate ? out[threadIdx.x] : 0.) + s_out[threadIdx.x];
}

template <int K, class LaunchMain>
static int run_reduce(int64_t num, bool aligned, double *out_dev, void *workspace, hipStream_t s,
                      LaunchMain launch)
{
    double *partial = (double *)workspace;
    const int64_t nvec2 = aligned ? num / 2 : 0;
    bool wrote = false;
    if (nvec2 > 0) {
        const int g = grid_capped(nvec2);
        launch(2, g, (int64_t)0, nvec2, partial);
        hipLaunchKernelGGL((fold_rows_kernel<K>), dim3(1), dim3(kBlock), 0, s, partial, g, out_dev, 0);
        wrote = true;
    }
    const int64_t done = nvec2 * 2;
    if (done < num || !wrote) {
        const int64_t rest = num - done;
        const int g = rest > 0 ? grid_capped(rest) : 0;
        if (g > 0) launch(1, g, done, rest, partial);
        hipLaunchKernelGGL((fold_rows_kernel<K>), dim3(1), dim3(kBlock), 0, s, partial, g, out_dev, wrote ? 1 : 0);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PXF_OK : hip_fail(e);
}

}  // namespace pxf

using namespace pxf;

extern "C" {

size_t pxf_reduce_workspace_bytes(void) { return (size_t)kRedMaxBlocks * PXF_NMOM * sizeof(double); }

int pxf_moments(const double *x, const double *y, const double *l, const double *m, const double *n,
                const double *w, int64_t num, double *out_dev, void *workspace, pxf_stream_t stream)
{
    if (num < 0 || !out_dev || !workspace || (num > 0 && (!x || !y))) return PXF_EINVAL;
    const bool hasdir = l && m && n;
    if (!hasdir && (l || m || n)) return PXF_EINVAL;
    if (device_cus() * kMaxBlocksPerCU > kRedMaxBlocks) return PXF_ERANGE;
    uintptr_t al = (uintptr_t)x | (uintptr_t)y | (uintptr_t)w;
    if (hasdir) al |= (uintptr_t)l | (uintptr_t)m | (uintptr_t)n;
    hipStream_t s = (hipStream_t)stream;
    auto launch = [&](int V, int g, int64_t first, int64_t nvec, double *partial) {
#define PXF_MOM(VV, D, W)                                                                                  \
    hipLaunchKernelGGL((moments_kernel<VV, D, W>), dim3(g), dim3(kBlock), 0, s, x, y, l, m, n, w, first, \
                       nvec, partial)
        if (V == 2) {
            if (hasdir && w) PXF_MOM(2, true, true);
            else if (hasdir) PXF_MOM(2, true, false);
            else if (w) PXF_MOM(2, false, true);
            else PXF_MOM(2, false, false);
        } else {
            if (hasdir && w) PXF_MOM(1, true, true);
            else if (hasdir) PXF_MOM(1, true, false);
            else if (w) PXF_MOM(1, false, true);
            else PXF_MOM(1, false, false);
        }
#undef PXF_MOM
    };
    return run_reduce<PXF_NMOM>(num, (al & 15) == 0, out_dev, workspace, s, launch);
}

int pxf_transform_flat_moments(double *opd, double *x, double *y, double *z, double *l, double *m, double *n,
                               double *ux, double *uy, double *uz, const double *w, int64_t num, double tx, double ty,
                               double tz, double rx, double ry, double rz, double nr, double *moments_dev,
                               void *workspace, pxf_stream_t stream)
{
    if (num < 0 || (moments_dev && !workspace)) return PXF_EINVAL;
    if (num > 0 && (!x || !y || !z || !l || !m || !n || !ux || !uy || !uz)) return PXF_EINVAL;
    if (device_cus() * kMaxBlocksPerCU > kRedMaxBlocks) return PXF_ERANGE;
    Cols c{};
    double *src[10] = {opd, x, y, z, l, m, n, ux, uy, uz};
    uintptr_t al = (uintptr_t)w;
    for (int i = 0; i < 10; i++) {
        c.p[i] = src[i];
        al |= (uintptr_t)src[i];
    }
    double cs[6];
    trig6(rx, ry, rz, false, cs);
    const TransformOp<false> top{tx, ty, tz, cs[0], cs[1], cs[2], cs[3], cs[4], cs[5], 3};   // position + direction
    hipStream_t s = (hipStream_t)stream;
    const bool nt = use_nt(num), hasw = w != nullptr, wopd = opd != nullptr, mom = moments_dev != nullptr;
    auto launch = [&](int V, int g, int64_t first, int64_t nvec, double *partial) {
#define PXF_FS5(VV, HW, WO, MO, NTT)                                                                                  \
    hipLaunchKernelGGL((focus_step_kernel<VV, HW, WO, MO, NTT>), dim3(g), dim3(kBlock), 0, s, c, w, first, nvec, top, \
                       FlatOp<WO>{nr}, partial)
#define PXF_FS4(VV, HW, WO, MO) do { if (nt) PXF_FS5(VV, HW, WO, MO, true); else PXF_FS5(VV, HW, WO, MO, false); } while (0)
#define PXF_FS3(VV, HW, WO) do { if (mom) PXF_FS4(VV, HW, WO, true); else PXF_FS4(VV, false, WO, false); } while (0)
#define PXF_FS2(VV, HW) do { if (wopd) PXF_FS3(VV, HW, true); else PXF_FS3(VV, HW, false); } while (0)
        if (V == 2) {
            if (hasw) PXF_FS2(2, true); else PXF_FS2(2, false);
        } else {
            if (hasw) PXF_FS2(1, true); else PXF_FS2(1, false);
        }
#undef PXF_FS2
#undef PXF_FS3
#undef PXF_FS4
#undef PXF_FS5
    };
    const bool aligned = (al & 15) == 0;
    if (mom) return run_reduce<PXF_NMOM>(num, aligned, moments_dev, workspace, s, launch);
    if (num == 0) return PXF_OK;
    const int64_t nvec2 = aligned ? num / 2 : 0;      // no sums wanted: an uncapped streaming launch
    if (nvec2 > 0) launch(2, grid_for(nvec2), (int64_t)0, nvec2, nullptr);
    if (nvec2 * 2 < num) launch(1, grid_for(num - nvec2 * 2), nvec2 * 2, num - nvec2 * 2, nullptr);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PXF_OK : hip_fail(e);
}

int pxf_central_moments(const double *x, const double *y, const double *w, int64_t num, double cx, double cy,
                        double *out_dev, void *workspace, pxf_stream_t stream)
{
    if (num < 0 || !out_dev || !workspace || (num > 0 && (!x || !y))) return PXF_EINVAL;
    if (device_cus() * kMaxBlocksPerCU > kRedMaxBlocks) return PXF_ERANGE;
    const uintptr_t al = (uintptr_t)x | (uintptr_t)y | (uintptr_t)w;
    hipStream_t s = (hipStream_t)stream;
    auto launch = [&](int V, int g, int64_t first, int64_t nvec, double *partial) {
        if (V == 2) {
            if (w) hipLaunchKernelGGL((central_kernel<2, true>), dim3(g), dim3(kBlock), 0, s, x, y, w, first, nvec, cx, cy, partial);
            else hipLaunchKernelGGL((central_kernel<2, false>), dim3(g), dim3(kBlock), 0, s, x, y, w, first, nvec, cx, cy, partial);
        } else {
            if (w) hipLaunchKernelGGL((central_kernel<1, true>), dim3(g), dim3(kBlock), 0, s, x, y, w, first, nvec, cx, cy, partial);
            else hipLaunchKernelGGL((central_kernel<1, false>), dim3(g), dim3(kBlock), 0, s, x, y, w, first, nvec, cx, cy, partial);
        }
    };
    return run_reduce<4>(num, (al & 15) == 0, out_dev, workspace, s, launch);
}

}  // extern "C"
