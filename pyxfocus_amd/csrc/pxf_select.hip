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
#include "pxf_ops_transforms.h"

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
__global__ __launch_bounds__(kBlock, 2) void sel_hist_kernel(const double *__restrict__ v, int64_t num,
                                                          SelState *__restrict__ st, int pass,
                                                          const unsigned long long *__restrict__ count_dev = nullptr)
{
    if (count_dev) {  // the element count was produced on the device (candidate buffer of pxf_radial_kth)
        const int64_t c = (int64_t)*count_dev;
        num = c < num ? c : num;
    }
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



// ---- order statistics of the RADII straight from x, y in ONE pass over the rays ------------------
// analyses.hpd is 2*median(sqrt((x-cx)^2+(y-cy)^2)).  Materialising the radii and running the six
// select passes above moves 72 B/ray; the algorithmic minimum is the 16 B/ray of x and y.  This
// path reaches it for the bulk of the data:
//   1. a strided SAMPLE of kFsSample rays gives, by radix select on the sample, a bracket [lo, hi]
//      around the wanted ranks (+-8 sigma of the sample rank);
//   2. ONE pass over all rays recomputes every radius (same arithmetic as RhoOp), counts those below
//      lo and appends those inside [lo, hi] to a candidate buffer (wave-aggregated atomics);
//   3. if the wanted ranks fall inside the bracket and the candidates fit the buffer, the exact
//      values are selected among the candidates (ranks re-based by the count below lo).
// The result is the same element np.median would use; if the bracket misses (heavy ties, adversarial
// order) `ok` stays 0 and the caller falls back to the full path.  mode 1: radii are |y - cy| (hpdY).
constexpr int kFsSample = 1 << 20;

struct FastSelState {
    double lohi[2];
    unsigned long long below, inside, nan_count;
    int ok;
};

__device__ __forceinline__ double fs_radius(double x, double y, double cx, double cy, int mode)
{
    if (mode == 1) return fabs(y - cy);
    const double dx = x - cx, dy = y - cy;
    return sqrt(dx * dx + dy * dy);
}

__global__ __launch_bounds__(kBlock, 2) void fs_sample_kernel(const double *__restrict__ x, const double *__restrict__ y,
                                                           double cx, double cy, int mode, int64_t num, int nsamp,
                                                           double *__restrict__ out)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= nsamp) return;
    // evenly strided: index floor(i * num / nsamp) (i < 2^20, num < 2^43: the product fits 64 bits)
    const int64_t j = (int64_t)(((unsigned long long)i * (unsigned long long)num) / (unsigned long long)nsamp);
    out[i] = fs_radius(mode == 1 ? 0. : x[j], y[j], cx, cy, mode);
}

constexpr int kFsStage = 2048;  // candidates staged per workgroup in LDS between flushes

__global__ __launch_bounds__(kBlock, 2) void fs_collect_kernel(const double *__restrict__ x, const double *__restrict__ y,
                                                            double cx, double cy, int mode, int64_t num,
                                                            FastSelState *__restrict__ fs, double *__restrict__ cand,
                                                            long long cap)
{
    // Candidates are staged in LDS and appended to the global buffer a block at a time: ONE global
    // atomic per flush (a single counter hit once per candidate serialises at ~10 ns each -- 8 ms per
    // 1e8 rays, measured).
    __shared__ double s_buf[kFsStage];
    __shared__ unsigned s_n;
    __shared__ unsigned long long s_base;
    if (threadIdx.x == 0) s_n = 0u;
    __syncthreads();
    const double lo = fs->lohi[0], hi = fs->lohi[1];
    unsigned below = 0, nans = 0;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    const int64_t first = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t iters = (num + stride - 1) / stride;  // the same for every thread: barriers inside the loop
    for (int64_t it = 0; it <= iters; it++) {
        // flush when the next iteration could overflow the stage, and once after the last iteration
        const unsigned staged = s_n;
        const bool last = it == iters;
        if (staged > (unsigned)(kFsStage - kBlock) || (last && staged)) {
            __syncthreads();  // everyone has read s_n
            if (threadIdx.x == 0) s_base = atomicAdd(&fs->inside, (unsigned long long)staged);
            __syncthreads();
            const unsigned long long base = s_base;
            for (unsigned j = threadIdx.x; j < staged; j += kBlock)
                if ((long long)(base + j) < cap) cand[base + j] = s_buf[j];
            __syncthreads();
            if (threadIdx.x == 0) s_n = 0u;
        }
        __syncthreads();
        if (last) break;
        const int64_t i = first + it * stride;
        if (i < num) {
            const double r = fs_radius(mode == 1 ? 0. : __builtin_nontemporal_load(x + i),
                                       __builtin_nontemporal_load(y + i), cx, cy, mode);
            nans += (r != r) ? 1u : 0u;
            below += (r < lo) ? 1u : 0u;
            if (r >= lo && r <= hi) s_buf[atomicAdd(&s_n, 1u)] = r;
        }
        __syncthreads();
    }
    for (int off = 32; off > 0; off >>= 1) {
        below += __shfl_down(below, off);
        nans += __shfl_down(nans, off);
    }
    if ((threadIdx.x & 63) == 0) {
        if (below) atomicAdd(&fs->below, (unsigned long long)below);
        if (nans) atomicAdd(&fs->nan_count, (unsigned long long)nans);
    }
}

// bracket from the sample select (out2 = the two sample order statistics)
__global__ void fs_bracket_kernel(FastSelState *__restrict__ fs, const double *__restrict__ out2, int lo_open,
                                  int hi_open)
{
    // a bracket that reaches the end of the sample must reach the end of the data
    fs->lohi[0] = lo_open ? -INFINITY : out2[0];
    fs->lohi[1] = hi_open ? INFINITY : out2[1];
}

// ranks re-based into the candidate buffer; decides whether the fast path is valid
__global__ void fs_setup_kernel(FastSelState *__restrict__ fs, SelState *__restrict__ st, long long k0, long long k1,
                                long long cap)
{
    const long long below = (long long)fs->below, inside = (long long)fs->inside;
    const bool ok = inside <= cap && k0 >= below && k1 < below + inside && fs->lohi[0] == fs->lohi[0] &&
                    fs->lohi[1] == fs->lohi[1];
    fs->ok = ok ? 1 : 0;
    st->k[0] = ok ? k0 - below : 0;
    st->k[1] = ok ? k1 - below : 0;
    st->nan_count = fs->nan_count;   // NaN anywhere -> NaN out (np.median)
    if (!ok) fs->inside = 0;         // the select passes below then touch nothing
}

// ---- weighted quantiles (analyses.hpd with weights, analyses.py:73-97) without a sort ------------
// The reference sorts the radii, forms cdf = cumsum(w)/sum(w) and takes r[argmin |cdf - q|] for
// q = .75 and .25.  With positive weights that element is either the first one whose cdf reaches q
// (index k) or its predecessor (argmin returns the first of two equal distances).  The same MSD
// radix descent as above finds r_k when the per-digit histograms hold WEIGHT instead of counts
// (fp64 ds_add / global atomic add): after six passes the state has the key of r_k, the weight B
// of all keys below it and the weight G at it; one more pass finds the largest key below r_k and
// the number of elements equal to r_k; a last single-thread kernel makes the reference's choice:
//   predecessor  if r_k is unique, a predecessor exists and |B/W - q| <= |(B+G)/W - q|,  else r_k
// (several elements tied at r_k: whichever of them crosses, its predecessor in sorted order has the
// same value unless it is the first of the tie, which the arbitrary order of an unstable argsort
// does not define either).  Sums are formed in a different order than np.cumsum: the choice can
// differ from the reference's only when the two distances agree to ~1e-13 relative.
struct WSelState {
    double hist[kSelMaxRanks][kSelBins];
    unsigned long long prefix[kSelMaxRanks];
    double below[kSelMaxRanks];   // weight of all keys strictly below the range selected so far
    double target[kSelMaxRanks];  // q * W
    double q[kSelMaxRanks];
    double at[kSelMaxRanks];      // weight at the final key (G)
    double total;                 // W
    long long prev[kSelMaxRanks];             // largest key below r_k, as a signed-monotone image
    unsigned long long count_at[kSelMaxRanks];  // elements equal to r_k
};
constexpr long long kNoPrev = (long long)0x8000000000000000ull;

__device__ __forceinline__ long long signed_key(unsigned long long key) { return (long long)(key ^ 0x8000000000000000ull); }
__device__ __forceinline__ unsigned long long unsigned_key(long long sk) { return (unsigned long long)sk ^ 0x8000000000000000ull; }

template <int NK>
__global__ __launch_bounds__(kBlock, 2) void wsel_hist_kernel(const double *__restrict__ v, const double *__restrict__ w,
                                                           int64_t num, WSelState *__restrict__ st, int pass)
{
    __shared__ double s_hist[NK][kSelBins];
    for (int i = threadIdx.x; i < NK * kSelBins; i += kBlock) (&s_hist[0][0])[i] = 0.;
    __syncthreads();
    const int shift = sel_shift(pass), width = sel_width(pass);
    const unsigned dmask = (1u << width) - 1u;
    unsigned long long prefix[NK];
#pragma unroll
    for (int s = 0; s < NK; s++) prefix[s] = st->prefix[s];
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < num; i += stride) {
        const unsigned long long key = key_of(v[i]);
        const unsigned digit = (unsigned)(key >> shift) & dmask;
        const unsigned long long hi = pass == 0 ? 0ull : (key >> (shift + width));
        bool any = pass == 0;
#pragma unroll
        for (int s = 0; s < NK; s++) any |= hi == prefix[s];
        if (!any) continue;
        const double wi = w[i];
#pragma unroll
        for (int s = 0; s < NK; s++)
            if (pass == 0 || hi == prefix[s]) unsafeAtomicAdd(&s_hist[s][digit], wi);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NK * kSelBins; i += kBlock) {
        const double c = (&s_hist[0][0])[i];
        if (c != 0.) unsafeAtomicAdd(&st->hist[i / kSelBins][i % kSelBins], c);
    }
}

// one wave: W (pass 0), then the bin where the running weight first reaches the target
template <int NK>
__global__ __launch_bounds__(64) void wsel_pick_kernel(WSelState *__restrict__ st, int pass)
{
    const int lane = threadIdx.x;
    const int width = sel_width(pass), nb = 1 << width, per = nb / 64;
    for (int s = 0; s < NK; s++) {
        double run = 0.;
        for (int j = 0; j < per; j++) run += st->hist[s][lane * per + j];
        double incl = run;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const double y = __shfl_up(incl, d, 64);
            if (lane >= d) incl += y;
        }
        const double tot = __shfl(incl, 63, 64);
        if (pass == 0) {
            if (lane == 0) {
                if (s == 0) st->total = tot;
                st->target[s] = st->q[s] * tot;
            }
            __syncthreads();
        }
        const double target = pass == 0 ? st->q[s] * tot : st->target[s];
        const double below = st->below[s];
        const double excl = incl - run;
        // first lane whose inclusive running weight reaches the target (the last non-empty lane if
        // rounding leaves the grand total a hair short of it)
        const bool reach = below + incl >= target;
        const unsigned long long m = __ballot(reach);
        const unsigned long long nz = __ballot(run != 0.);
        int pick = m ? __ffsll((long long)m) - 1 : (nz ? 63 - __clzll((long long)nz) : 0);
        if (lane == pick) {
            double cum = below + excl;
            int b = lane * per, last_nz = -1;
            double cum_last = cum;
            for (; b < lane * per + per; b++) {
                const double c = st->hist[s][b];
                if (c != 0.) {
                    last_nz = b;
                    cum_last = cum;
                }
                if (c != 0. && cum + c >= target) break;
                cum += c;
            }
            if (b == lane * per + per) {  // fell off the end (rounding): take the last non-empty bin
                b = last_nz < 0 ? lane * per : last_nz;
                cum = cum_last;
            }
            st->prefix[s] = (st->prefix[s] << width) | (unsigned long long)b;
            st->below[s] = cum;
            st->at[s] = st->hist[s][b];
        }
        __syncthreads();
        for (int j = lane; j < kSelBins; j += 64) st->hist[s][j] = 0.;
        __syncthreads();
    }
}

template <int NK>
__global__ __launch_bounds__(kBlock, 2) void wsel_neighbors_kernel(const double *__restrict__ v, int64_t num,
                                                                WSelState *__restrict__ st)
{
    unsigned long long keyk[NK];
    long long best[NK];
    unsigned cnt[NK];
#pragma unroll
    for (int s = 0; s < NK; s++) {
        keyk[s] = st->prefix[s];
        best[s] = kNoPrev;
        cnt[s] = 0u;
    }
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < num; i += stride) {
        const unsigned long long key = key_of(v[i]);
#pragma unroll
        for (int s = 0; s < NK; s++) {
            if (key == keyk[s]) cnt[s]++;
            else if (key < keyk[s]) {
                const long long sk = signed_key(key);
                best[s] = sk > best[s] ? sk : best[s];
            }
        }
    }
#pragma unroll
    for (int s = 0; s < NK; s++) {
        for (int off = 32; off > 0; off >>= 1) {
            const long long o = __shfl_down(best[s], off);
            best[s] = o > best[s] ? o : best[s];
            cnt[s] += __shfl_down(cnt[s], off);
        }
        if ((threadIdx.x & 63) == 0) {
            if (best[s] != kNoPrev) atomicMax(&st->prev[s], best[s]);
            if (cnt[s]) atomicAdd(&st->count_at[s], (unsigned long long)cnt[s]);
        }
    }
}

__global__ void wsel_init_kernel(WSelState *__restrict__ st, double q0, double q1)
{
    st->q[0] = q0;
    st->q[1] = q1;
    st->prev[0] = kNoPrev;
    st->prev[1] = kNoPrev;
}

__global__ void wsel_finish_kernel(const WSelState *__restrict__ st, int nk, double *__restrict__ out)
{
    for (int s = 0; s < nk; s++) {
        const double rk = value_of(st->prefix[s]);
        double r = rk;
        if (st->count_at[s] == 1ull && st->prev[s] != kNoPrev) {
            const double W = st->total, q = st->q[s];
            const double dprev = fabs(st->below[s] / W - q), dk = fabs((st->below[s] + st->at[s]) / W - q);
            if (dprev <= dk) r = value_of(unsigned_key(st->prev[s]));
        }
        out[s] = r;
    }
}

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

// ---- radii order statistics in one pass over x, y ---------------------------------------------------
static inline int64_t fs_cap(int64_t num) { const int64_t c = num / 32; return c < (1 << 20) ? (1 << 20) : c; }

size_t pxf_radial_kth_workspace_bytes(int64_t num)
{
    if (num < 0) return 0;
    return 256 + 2 * sizeof(SelState) + 64 + (size_t)kFsSample * sizeof(double) + (size_t)fs_cap(num) * sizeof(double);
}

int pxf_radial_kth(const double *x, const double *y, double cx, double cy, int32_t mode, int64_t num, const int64_t *k,
                   int32_t nk, double *out_dev, int32_t *ok_dev, void *workspace, pxf_stream_t stream)
{
    if (num <= 0 || !y || (mode == 0 && !x) || (mode != 0 && mode != 1) || !k || nk < 1 || nk > kSelMaxRanks ||
        !out_dev || !ok_dev || !workspace)
        return PXF_EINVAL;
    for (int s = 0; s < nk; s++)
        if (k[s] < 0 || k[s] >= num) return PXF_EINVAL;
    const int64_t k0 = k[0] < k[nk - 1] ? k[0] : k[nk - 1], k1 = k[0] < k[nk - 1] ? k[nk - 1] : k[0];
    hipStream_t s = (hipStream_t)stream;
    char *w = (char *)workspace;
    FastSelState *fs = (FastSelState *)w;
    SelState *sa = (SelState *)(w + 256), *sb = sa + 1;
    double *tmp2 = (double *)(w + 256 + 2 * sizeof(SelState));
    double *samp = tmp2 + 8, *cand = samp + kFsSample;
    const long long cap = (long long)fs_cap(num);
    hipError_t e = hipMemsetAsync(w, 0, 256 + 2 * sizeof(SelState) + 64, s);
    if (e != hipSuccess) return hip_fail(e);
    // 1. sample -> bracket.  Sample ranks of the wanted quantiles +- 8 sigma (sigma = sqrt(p(1-p)/S) <= 1/(2 sqrt S))
    const int nsamp = num < kFsSample ? (int)num : kFsSample;
    hipLaunchKernelGGL(fs_sample_kernel, dim3((nsamp + kBlock - 1) / kBlock), dim3(kBlock), 0, s, x, y, cx, cy, (int)mode, num,
                       nsamp, samp);
    const double margin = 4.0 * sqrt((double)nsamp) + 2.;
    long long slo = (long long)floor((double)k0 / (double)num * nsamp - margin);
    long long shi = (long long)ceil((double)k1 / (double)num * nsamp + margin);
    if (slo < 0) slo = 0;
    if (shi > nsamp - 1) shi = nsamp - 1;
    hipLaunchKernelGGL(sel_init_kernel, dim3(1), dim3(1), 0, s, sa, slo, shi);
    for (int p = 0; p < kSelPasses; p++) {
        hipLaunchKernelGGL((sel_hist_kernel<2>), dim3(grid_capped(nsamp)), dim3(kBlock), 0, s, samp, (int64_t)nsamp, sa, p, nullptr);
        hipLaunchKernelGGL((sel_pick_kernel<2>), dim3(1), dim3(64), 0, s, sa, p, tmp2);
    }
    hipLaunchKernelGGL(fs_bracket_kernel, dim3(1), dim3(1), 0, s, fs, tmp2, (int)(slo == 0), (int)(shi == nsamp - 1));
    // 2. one pass over the rays
    hipLaunchKernelGGL(fs_collect_kernel, dim3(grid_capped(num)), dim3(kBlock), 0, s, x, y, cx, cy, (int)mode, num, fs, cand, cap);
    // 3. exact select among the candidates
    hipLaunchKernelGGL(fs_setup_kernel, dim3(1), dim3(1), 0, s, fs, sb, (long long)k0, (long long)k1, cap);
    for (int p = 0; p < kSelPasses; p++) {
        hipLaunchKernelGGL((sel_hist_kernel<2>), dim3(grid_capped(cap)), dim3(kBlock), 0, s, cand, (int64_t)cap, sb, p, &fs->inside);
        hipLaunchKernelGGL((sel_pick_kernel<2>), dim3(1), dim3(64), 0, s, sb, p, tmp2 + 2);
    }
    // results in the caller's rank order, and the validity flag
    e = hipMemcpyAsync(out_dev, tmp2 + 2 + (k[0] <= k[nk - 1] ? 0 : 1), sizeof(double), hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess && nk > 1)
        e = hipMemcpyAsync(out_dev + 1, tmp2 + 2 + (k[0] <= k[nk - 1] ? 1 : 0), sizeof(double), hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(ok_dev, &fs->ok, sizeof(int), hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return hip_fail(e);
    e = hipGetLastError();
    return e == hipSuccess ? PXF_OK : hip_fail(e);
}

// ---- weighted quantiles ---------------------------------------------------------------------------
size_t pxf_wselect_workspace_bytes(void) { return sizeof(WSelState); }

void pxf_wselect_state_layout(int64_t *hist_offset, int64_t *hist_bytes, int64_t *prev_offset, int64_t *count_offset)
{
    if (hist_offset) *hist_offset = (int64_t)offsetof(WSelState, hist);
    if (hist_bytes) *hist_bytes = (int64_t)sizeof(((WSelState *)nullptr)->hist);
    if (prev_offset) *prev_offset = (int64_t)offsetof(WSelState, prev);
    if (count_offset) *count_offset = (int64_t)offsetof(WSelState, count_at);
}

int pxf_wselect_begin(const double *q, int32_t nq, void *workspace, pxf_stream_t stream)
{
    if (!q || nq < 1 || nq > kSelMaxRanks || !workspace) return PXF_EINVAL;
    for (int s = 0; s < nq; s++)
        if (!(q[s] > 0. && q[s] <= 1.)) return PXF_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(workspace, 0, sizeof(WSelState), s);
    if (e != hipSuccess) return hip_fail(e);
    hipLaunchKernelGGL(wsel_init_kernel, dim3(1), dim3(1), 0, s, (WSelState *)workspace, q[0], nq > 1 ? q[1] : q[0]);
    e = hipGetLastError();
    return e == hipSuccess ? PXF_OK : hip_fail(e);
}

int pxf_wselect_hist(const double *v, const double *w, int64_t num, int32_t nq, void *workspace, int32_t pass,
                     pxf_stream_t stream)
{
    if (num < 0 || (num > 0 && (!v || !w)) || nq < 1 || nq > kSelMaxRanks || !workspace || pass < 0 ||
        pass >= kSelPasses)
        return PXF_EINVAL;
    if (num == 0) return PXF_OK;
    hipStream_t s = (hipStream_t)stream;
    const int g = grid_capped(num);
    if (nq == 1) hipLaunchKernelGGL((wsel_hist_kernel<1>), dim3(g), dim3(kBlock), 0, s, v, w, num, (WSelState *)workspace, (int)pass);
    else hipLaunchKernelGGL((wsel_hist_kernel<2>), dim3(g), dim3(kBlock), 0, s, v, w, num, (WSelState *)workspace, (int)pass);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PXF_OK : hip_fail(e);
}

int pxf_wselect_pick(int32_t nq, void *workspace, int32_t pass, pxf_stream_t stream)
{
    if (nq < 1 || nq > kSelMaxRanks || !workspace || pass < 0 || pass >= kSelPasses) return PXF_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (nq == 1) hipLaunchKernelGGL((wsel_pick_kernel<1>), dim3(1), dim3(64), 0, s, (WSelState *)workspace, (int)pass);
    else hipLaunchKernelGGL((wsel_pick_kernel<2>), dim3(1), dim3(64), 0, s, (WSelState *)workspace, (int)pass);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PXF_OK : hip_fail(e);
}

int pxf_wselect_neighbors(const double *v, int64_t num, int32_t nq, void *workspace, pxf_stream_t stream)
{
    if (num < 0 || (num > 0 && !v) || nq < 1 || nq > kSelMaxRanks || !workspace) return PXF_EINVAL;
    if (num == 0) return PXF_OK;
    hipStream_t s = (hipStream_t)stream;
    const int g = grid_capped(num);
    if (nq == 1) hipLaunchKernelGGL((wsel_neighbors_kernel<1>), dim3(g), dim3(kBlock), 0, s, v, num, (WSelState *)workspace);
    else hipLaunchKernelGGL((wsel_neighbors_kernel<2>), dim3(g), dim3(kBlock), 0, s, v, num, (WSelState *)workspace);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PXF_OK : hip_fail(e);
}

int pxf_wselect_finish(int32_t nq, void *workspace, double *out_dev, pxf_stream_t stream)
{
    if (nq < 1 || nq > kSelMaxRanks || !workspace || !out_dev) return PXF_EINVAL;
    hipLaunchKernelGGL(wsel_finish_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (const WSelState *)workspace, (int)nq,
                       out_dev);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PXF_OK : hip_fail(e);
}

int pxf_wquantiles(const double *v, const double *w, int64_t num, const double *q, int32_t nq, double *out_dev,
                   void *workspace, pxf_stream_t stream)
{
    if (num <= 0) return PXF_EINVAL;
    int rc = pxf_wselect_begin(q, nq, workspace, stream);
    for (int p = 0; rc == PXF_OK && p < kSelPasses; p++) {
        rc = pxf_wselect_hist(v, w, num, nq, workspace, p, stream);
        if (rc == PXF_OK) rc = pxf_wselect_pick(nq, workspace, p, stream);
    }
    if (rc == PXF_OK) rc = pxf_wselect_neighbors(v, num, nq, workspace, stream);
    if (rc == PXF_OK) rc = pxf_wselect_finish(nq, workspace, out_dev, stream);
    return rc;
}

}  // extern "C"
