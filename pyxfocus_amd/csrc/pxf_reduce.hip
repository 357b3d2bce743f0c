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

template <int V, bool HASDIR, bool HASW>
__global__ __launch_bounds__(kBlock) void moments_kernel(const double *__restrict__ x, const double *__restrict__ y,
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
        for (int k = 0; k < V; k++) {
            const double ww = HASW ? W[k] : 1.;
            acc[0] += ww;
            acc[1] += ww * X[k];
            acc[2] += ww * Y[k];
            if (HASDIR) {
                const double ln = Lc[k] / Nc[k], mn = Mc[k] / Nc[k];
                acc[3] += ww * ln;
                acc[4] += ww * mn;
                acc[5] += ww * (X[k] * Lc[k] / Nc[k]);  // x*l/n as written in analyses.py:123
                acc[6] += ww * (Y[k] * Mc[k] / Nc[k]);
                acc[7] += ww * (ln * ln);
                acc[8] += ww * (mn * mn);
                acc[9] += ww * Lc[k];
                acc[10] += ww * Mc[k];
                acc[11] += ww * Nc[k];
            }
        }
    }
    block_fold<PXF_NMOM>(acc, partial + (int64_t)blockIdx.x * PXF_NMOM);
}

template <int V, bool HASW>
__global__ __launch_bounds__(kBlock) void central_kernel(const double *__restrict__ x, const double *__restrict__ y,
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
__global__ __launch_bounds__(kBlock) void fold_rows_kernel(const double *__restrict__ rows, int nrows,
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
