// pxf_select.hip -- per-ray analysis helpers that are not plain sums:
//   pxf_rho        analyses.rho      (analyses.py:60-71)   sqrt((x-cx)^2+(y-cy)^2)
//   pxf_indangle   analyses.indAngle / grazeAngle (analyses.py:164-187)  acos(d.u), pi/2-acos(d.u)
//   pxf_kth_values order statistics for analyses.hpd (analyses.py:88-97: 2*median(rho)) without
//                  sorting: MSD radix SELECT on the order-preserving integer image of fp64.
//
// Radix select: six passes over the column (digits of 11,11,11,11,11,9 bits).  Each pass builds
// a 2048-bin histogram of the current digit among the elements that match the prefix chosen so
// far -- per workgroup in LDS (ds_add), then one global atomic per non-empty bin -- and a
// single-wave kernel walks the histogram to the bin that holds rank k, appends the digit to the
// prefix and re-bases k.  After the last pass the prefix IS the k-th smallest key.  Up to two
// ranks are tracked at once (np.median of an even count averages the two middle values).
// No host round trip between passes; 6 x 8 B/ray of reads instead of a full sort.  NaN anywhere
// -> NaN out, as np.median does.
#include <math.h>
#include <stddef.h>

#include "pxf_kernel.h"

namespace pxf {

constexpr int kSelBins = 2048;
constexpr int kSelMaxRanks = 2;
constexpr int kSelPasses = 6;
__host__ __device__ constexpr int sel_width(int p) { return p < 5 ? 11 : 9; }
__host__ __device__ constexpr int sel_shift(int p) { return p < 5 ? 64 - 11 * (p + 1) : 0; }

struct SelState {
    unsigned long long hist[kSelMaxRanks][kSelBins];
    unsigned long long prefix[kSelMaxRanks];
    long long k[kSelMaxRanks];
    unsigned long long nan_count;
};

__device__ __forceinline__ unsigned long long key_of(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);  // monotone: smaller double <-> smaller key
}

__device__ __forceinline__ double value_of(unsigned long long key)
{
    const unsigned long long b = (key >> 63) ? (key & 0x7fffffffffffffffull) : ~key;
    return __longlong_as_double((long long)b);
}

template <int NK>
__global__ __launch_bounds__(kBlock) void sel_hist_kernel(const double *__restrict__ v, int64_t num,
                                                          SelState *__restrict__ st, int pass)
{
    __shared__ unsigned s_hist[NK][kSelBins];
    for (int i = threadIdx.x; i < NK * kSelBins; i += kBlock) (&s_hist[0][0])[i] = 0u;
    __syncthreads();
    const int shift = sel_shift(pass), width = sel_width(pass);
    const unsigned dmask = (1u << width) - 1u;
    unsigned long long prefix[NK];
#pragma unroll
    for (int s = 0; s < NK; s++) prefix[s] = st->prefix[s];
    unsigned nan_local = 0;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < num; i += stride) {
        const double x = v[i];
        if (pass == 0 && x != x) nan_local++;
        const unsigned long long key = key_of(x);
        const unsigned digit = (unsigned)(key >> shift) & dmask;
        const unsigned long long hi = pass == 0 ? 0ull : (key >> (shift + width));
#pragma unroll
        for (int s = 0; s < NK; s++)
            if (pass == 0 || hi == prefix[s]) atomicAdd(&s_hist[s][digit], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NK * kSelBins; i += kBlock) {
        const unsigned c = (&s_hist[0][0])[i];
        if (c) atomicAdd(&st->hist[i / kSelBins][i % kSelBins], (unsigned long long)c);
    }
    if (pass == 0 && nan_local) atomicAdd(&st->nan_count, (unsigned long long)nan_local);
}

// one wave: locate the bin holding rank k, extend the prefix, clear the histogram for the next pass
template <int NK>
__global__ __launch_bounds__(64) void sel_pick_kernel(SelState *__restrict__ st, int pass, double *__restrict__ out)
{
    const int lane = threadIdx.x;
    const int width = sel_width(pass), nb = 1 << width;
    for (int s = 0; s < NK; s++) {
        // each lane owns a contiguous run of bins; exclusive scan of the run totals across lanes
        const int per = nb / 64;
        unsigned long long run = 0;
        for (int j = 0; j < per; j++) run += st->hist[s][lane * per + j];
        unsigned long long incl = run;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long y = __shfl_up(incl, d, 64);
            if (lane >= d) incl += y;
        }
        const unsigned long long excl = incl - run;
        const unsigned long long k = (unsigned long long)st->k[s];
        const bool mine = k >= excl && k < incl;  // exactly one lane (k < total by construction)
        if (mine) {
            unsigned long long cum = excl;
            int b = lane * per;
            for (; b < lane * per + per; b++) {
                const unsigned long long c = st->hist[s][b];
                if (k < cum + c) break;
                cum += c;
            }
            st->prefix[s] = (st->prefix[s] << width) | (unsigned long long)b;
            st->k[s] = (long long)(k - cum);
        }
        __syncthreads();
        for (int j = lane; j < kSelBins; j += 64) st->hist[s][j] = 0ull;
    }
    if (pass == kSelPasses - 1 && lane == 0) {
        for (int s = 0; s < NK; s++) out[s] = st->nan_count ? __longlong_as_double(0x7ff8000000000000ll) : value_of(st->prefix[s]);
    }
}

__global__ void sel_init_kernel(SelState *__restrict__ st, long long k0, long long k1)
{
    // the histogram/prefix/nan fields were cleared by the preceding memset on the same stream
    st->k[0] = k0;
    st->k[1] = k1;
}

struct RhoOp {
    PXF_OP_DEFAULTS
    static constexpr unsigned RD = B(X) | B(Y);
    static constexpr unsigned WR = B(AUX);
    double cx, cy;
    template <int V>
    __device__ __forceinline__ void apply(Ray (&r)[V], const bool (&sel)[V], unsigned (&wm)[V]) const
    {
#pragma unroll
        for (int k = 0; k < V; k++) {
            const double dx = r[k].c[X] - cx, dy = r[k].c[Y] - cy;
            r[k].c[AUX] = sqrt(dx * dx + dy * dy);
            wm[k] = WR;
        }
    }
};

// analyses.measureOPD (analyses.py:231-244): distance of every ray from a point
struct Dist3Op {
    PXF_OP_DEFAULTS
    static constexpr unsigned RD = POS;
    static constexpr unsigned WR = B(AUX);
    double px, py, pz;
    template <int V>
    __device__ __forceinline__ void apply(Ray (&r)[V], const bool (&sel)[V], unsigned (&wm)[V]) const
    {
#pragma unroll
        for (int k = 0; k < V; k++) {
            const double dx = r[k].c[X] - px, dy = r[k].c[Y] - py, dz = r[k].c[Z] - pz;
            r[k].c[AUX] = sqrt(dx * dx + dy * dy + dz * dz);
            wm[k] = WR;
        }
    }
};

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

using namespace pxf;

extern "C" {

int pxf_rho(const double *x, const double *y, double cx, double cy, double *out, int64_t num,
            pxf_stream_t stream)
{
    Cols c{};
    c.p[X] = const_cast<double *>(x);
    c.p[Y] = const_cast<double *>(y);
    c.p[AUX] = out;
    return launch_rays(c, num, nullptr, stream, RhoOp{cx, cy});
}

int pxf_indangle(const double *l, const double *m, const double *n, const double *ux, const double *uy,
                 const double *uz, double *out, int64_t num, int32_t graze, pxf_stream_t stream)
{
    Cols c{};
    const double *src[6] = {l, m, n, ux, uy, uz};
    for (int i = 0; i < 6; i++) c.p[L + i] = const_cast<double *>(src[i]);
    c.p[AUX] = out;
    return launch_rays(c, num, nullptr, stream, IndAngleOp{graze});
}

int pxf_dist3(const double *x, const double *y, const double *z, double px, double py, double pz, double *out,
              int64_t num, pxf_stream_t stream)
{
    Cols c{};
    c.p[X] = const_cast<double *>(x);
    c.p[Y] = const_cast<double *>(y);
    c.p[Z] = const_cast<double *>(z);
    c.p[AUX] = out;
    return launch_rays(c, num, nullptr, stream, Dist3Op{px, py, pz});
}

size_t pxf_select_workspace_bytes(void) { return sizeof(SelState); }

// ---- the same selection, step by step, for order statistics over a bundle that is SHARDED across
// GPUs: every rank builds the digit histogram of its shard, the histograms (and the NaN count) are
// summed over ranks by the caller (an all-reduce of 32 KB), every rank picks the same bin.  Six
// rounds give the exact global k-th value with no ray leaving its GPU.
int32_t pxf_select_passes(void) { return kSelPasses; }

void pxf_select_state_layout(int64_t *hist_offset, int64_t *hist_bytes, int64_t *nan_offset)
{
    if (hist_offset) *hist_offset = (int64_t)offsetof(SelState, hist);
    if (hist_bytes) *hist_bytes = (int64_t)sizeof(((SelState *)nullptr)->hist);
    if (nan_offset) *nan_offset = (int64_t)offsetof(SelState, nan_count);
}

int pxf_select_begin(const int64_t *k, int32_t nk, void *workspace, pxf_stream_t stream)
{
    if (!k || nk < 1 || nk > kSelMaxRanks || !workspace) return PXF_EINVAL;
    for (int s = 0; s < nk; s++)
        if (k[s] < 0) return PXF_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(workspace, 0, sizeof(SelState), s);
    if (e != hipSuccess) return hip_fail(e);
    hipLaunchKernelGGL(sel_init_kernel, dim3(1), dim3(1), 0, s, (SelState *)workspace, (long long)k[0],
                       (long long)(nk > 1 ? k[1] : k[0]));
    e = hipGetLastError();
    return e == hipSuccess ? PXF_OK : hip_fail(e);
}

int pxf_select_hist(const double *v, int64_t num, int32_t nk, void *workspace, int32_t pass, pxf_stream_t stream)
{
    if (num < 0 || (num > 0 && !v) || nk < 1 || nk > kSelMaxRanks || !workspace || pass < 0 || pass >= kSelPasses)
        return PXF_EINVAL;
    if (num == 0) return PXF_OK;  // an empty shard contributes nothing
    hipStream_t s = (hipStream_t)stream;
    const int g = grid_capped(num);
    if (nk == 1) hipLaunchKernelGGL((sel_hist_kernel<1>), dim3(g), dim3(kBlock), 0, s, v, num, (SelState *)workspace, (int)pass);
    else hipLaunchKernelGGL((sel_hist_kernel<2>), dim3(g), dim3(kBlock), 0, s, v, num, (SelState *)workspace, (int)pass);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PXF_OK : hip_fail(e);
}

int pxf_select_pick(int32_t nk, void *workspace, int32_t pass, double *out_dev, pxf_stream_t stream)
{
    if (nk < 1 || nk > kSelMaxRanks || !workspace || !out_dev || pass < 0 || pass >= kSelPasses) return PXF_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (nk == 1) hipLaunchKernelGGL((sel_pick_kernel<1>), dim3(1), dim3(64), 0, s, (SelState *)workspace, (int)pass, out_dev);
    else hipLaunchKernelGGL((sel_pick_kernel<2>), dim3(1), dim3(64), 0, s, (SelState *)workspace, (int)pass, out_dev);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PXF_OK : hip_fail(e);
}

int pxf_kth_values(const double *v, int64_t num, const int64_t *k, int32_t nk, double *out_dev,
                   void *workspace, pxf_stream_t stream)
{
    if (num <= 0 || !v || !k || nk < 1 || nk > kSelMaxRanks || !out_dev || !workspace) return PXF_EINVAL;
    for (int s = 0; s < nk; s++)
        if (k[s] < 0 || k[s] >= num) return PXF_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    SelState *st = (SelState *)workspace;
    hipError_t e = hipMemsetAsync(st, 0, sizeof(SelState), s);
    if (e != hipSuccess) return hip_fail(e);
    hipLaunchKernelGGL(sel_init_kernel, dim3(1), dim3(1), 0, s, st, (long long)k[0], (long long)(nk > 1 ? k[1] : k[0]));
    const int g = grid_capped(num);
    for (int p = 0; p < kSelPasses; p++) {
        if (nk == 1) {
            hipLaunchKernelGGL((sel_hist_kernel<1>), dim3(g), dim3(kBlock), 0, s, v, num, st, p);
            hipLaunchKernelGGL((sel_pick_kernel<1>), dim3(1), dim3(64), 0, s, st, p, out_dev);
        } else {
            hipLaunchKernelGGL((sel_hist_kernel<2>), dim3(g), dim3(kBlock), 0, s, v, num, st, p);
            hipLaunchKernelGGL((sel_pick_kernel<2>), dim3(1), dim3(64), 0, s, st, p, out_dev);
        }
    }
    e = hipGetLastError();
    return e == hipSuccess ? PXF_OK : hip_fail(e);
}

}  // extern "C"
