// pxf_compact.hip -- vignette(): stable stream compaction of the ray bundle
// (transformations.py:214-225: `[rays[i][ind] for i in range(10)]`, default
//  ind = where(l**2+m**2+n**2 > .1)), plus the index-array gather and ind->mask helpers.
//
// Order-preserving compaction in three launches:
//   1. pxf_vignette_mask     (only for the default predicate) 24 B read, 1 B written per ray
//   2. pxf_compact_scan      per-TILE popcount of the mask via 64-lane ballots, then an exclusive
//                            scan of the tile counts (one workgroup; <= 0.5M tiles at 1e9 rays)
//   3. pxf_compact_scatter   every wave ballots its 64 flags, the lane's rank inside the wave is
//                            mbcnt(ballot) (popcount of lower lanes), wave bases come from a
//                            32-entry LDS prefix, the tile base from step 2; all columns move in
//                            the same launch: out[c][base+rank] = in[c][i].
// Reads are fully coalesced; writes of kept rays are contiguous per wave.  No atomics, no
// inter-workgroup communication inside a launch -> deterministic and placement-independent.
#include "pxf_kernel.h"

namespace pxf {

constexpr int kTileRows = 8;                      // rows of 256 rays per tile
constexpr int kTile = kBlock * kTileRows;         // 2048 rays per workgroup
constexpr int kMaxCols = 16;
constexpr int kWavesPerBlock = kBlock / 64;

struct ColPtrs {
    const double *in[kMaxCols];
    double *out[kMaxCols];
};

__device__ __forceinline__ unsigned lane_rank(unsigned long long ballot)
{
    // number of set bits below this lane
    return __builtin_amdgcn_mbcnt_hi((unsigned)(ballot >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ballot, 0u));
}

// default predicate, transformations.py:221-223.  Two rays per thread, one 2-byte store.
__global__ __launch_bounds__(kBlock, 2) void vignette_mask_kernel(const double *__restrict__ l,
                                                               const double *__restrict__ m,
                                                               const double *__restrict__ n,
                                                               int64_t num, uint8_t *__restrict__ mask,
                                                               int vec2)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    const int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (vec2) {
        const int64_t nvec = num / 2;
        for (int64_t v = tid; v < nvec; v += stride) {
            double a[2], b[2], c[2];
            load_col<2>(l, v, a);
            load_col<2>(m, v, b);
            load_col<2>(n, v, c);
            const unsigned f0 = (a[0] * a[0] + b[0] * b[0] + c[0] * c[0]) > .1;
            const unsigned f1 = (a[1] * a[1] + b[1] * b[1] + c[1] * c[1]) > .1;
            *reinterpret_cast<uint16_t *>(mask + 2 * v) = (uint16_t)(f0 | (f1 << 8));
        }
        if (tid == 0 && (num & 1)) {
            const int64_t i = num - 1;
            mask[i] = (l[i] * l[i] + m[i] * m[i] + n[i] * n[i]) > .1;
        }
    } else {
        for (int64_t i = tid; i < num; i += stride) mask[i] = (l[i] * l[i] + m[i] * m[i] + n[i] * n[i]) > .1;
    }
}

// step 2a: per-tile counts
__global__ __launch_bounds__(kBlock, 2) void tile_count_kernel(const uint8_t *__restrict__ mask, int64_t num,
                                                            int64_t *__restrict__ counts)
{
    __shared__ unsigned s_w[kWavesPerBlock];
    const int64_t base = (int64_t)blockIdx.x * kTile;
    unsigned cnt = 0;
#pragma unroll
    for (int r = 0; r < kTileRows; r++) {
        const int64_t i = base + r * kBlock + threadIdx.x;
        const bool f = i < num && mask[i] != 0;
        cnt += (unsigned)__popcll(__ballot(f));  // same value in every lane of the wave
    }
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned t = 0;
#pragma unroll
        for (int w = 0; w < kWavesPerBlock; w++) t += s_w[w];
        counts[blockIdx.x] = t;
    }
}

// step 2b: exclusive scan of the tile counts, in place, by ONE workgroup of 1024 threads;
// offsets[ntiles] and *total receive the grand total.
__global__ __launch_bounds__(1024) void tile_scan_kernel(int64_t *__restrict__ offs, int64_t ntiles,
                                                         int64_t *__restrict__ total)
{
    __shared__ int64_t s_wave[16];
    __shared__ int64_t s_carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int64_t b = 0; b < ntiles; b += 1024) {
        const int64_t i = b + threadIdx.x;
        const int64_t v = i < ntiles ? offs[i] : 0;
        int64_t x = v;  // inclusive scan inside the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int64_t y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        if (lane == 63) s_wave[wave] = x;
        __syncthreads();
        int64_t wbase = 0;
        for (int w = 0; w < wave; w++) wbase += s_wave[w];
        const int64_t carry = s_carry;
        if (i < ntiles) offs[i] = carry + wbase + x - v;  // exclusive
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = carry + wbase + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        offs[ntiles] = s_carry;
        if (total) *total = s_carry;
    }
}

// step 3: scatter all columns
__global__ __launch_bounds__(kBlock, 2) void compact_scatter_kernel(ColPtrs p, int ncols, int64_t num,
                                                                 const uint8_t *__restrict__ mask,
                                                                 const int64_t *__restrict__ offs)
{
    __shared__ unsigned s_seg[kTileRows * kWavesPerBlock];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * kTile;
    bool f[kTileRows];
    unsigned rank[kTileRows];
#pragma unroll
    for (int r = 0; r < kTileRows; r++) {
        const int64_t i = base + r * kBlock + threadIdx.x;
        f[r] = i < num && mask[i] != 0;
        const unsigned long long bal = __ballot(f[r]);
        rank[r] = lane_rank(bal);
        if (lane == 0) s_seg[r * kWavesPerBlock + wave] = (unsigned)__popcll(bal);
    }
    __syncthreads();
    const int64_t tile_off = offs[blockIdx.x];
    // exclusive prefix over the 32 (row, wave) segments, row-major = ray order
    unsigned run = 0;
    unsigned segbase[kTileRows];
#pragma unroll
    for (int r = 0; r < kTileRows; r++) {
#pragma unroll
        for (int w = 0; w < kWavesPerBlock; w++) {
            if (w == wave) segbase[r] = run;
            run += s_seg[r * kWavesPerBlock + w];
        }
    }
    for (int c = 0; c < ncols; c++) {
        const double *__restrict__ in = p.in[c];
        double *__restrict__ out = p.out[c];
        double v[kTileRows];
#pragma unroll
        for (int r = 0; r < kTileRows; r++) {
            const int64_t i = base + r * kBlock + threadIdx.x;
            if (f[r]) v[r] = in[i];
        }
#pragma unroll
        for (int r = 0; r < kTileRows; r++)
            if (f[r]) out[tile_off + segbase[r] + rank[r]] = v[r];
    }
}

__global__ __launch_bounds__(kBlock, 2) void gather_kernel(ColPtrs p, int ncols, const int64_t *__restrict__ idx,
                                                        int64_t nidx, int64_t num, int *__restrict__ oob)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < nidx; j += stride) {
        int64_t i = idx[j];
        if (i < 0) i += num;  // numpy negative-index wrap
        if (i < 0 || i >= num) {
            if (oob) *oob = 1;  // numpy raises IndexError; the Python layer checks this flag
            continue;
        }
        for (int c = 0; c < ncols; c++) p.out[c][j] = p.in[c][i];
    }
}

__global__ __launch_bounds__(kBlock, 2) void mask_from_index_kernel(const int64_t *__restrict__ idx, int64_t nidx,
                                                                 int64_t num, uint8_t *__restrict__ mask)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < nidx; j += stride) {
        int64_t i = idx[j];
        if (i < 0) i += num;
        if (i >= 0 && i < num) mask[i] = 1;
    }
}

static int64_t ntiles_for(int64_t num) { return (num + kTile - 1) / kTile; }

static int fill_ptrs(ColPtrs &p, const double *const *in_cols, double *const *out_cols, int32_t ncols)
{
    if (ncols < 1 || ncols > kMaxCols || !in_cols || !out_cols) return PXF_EINVAL;
    for (int c = 0; c < ncols; c++) {
        if (!in_cols[c] || !out_cols[c]) return PXF_EINVAL;
        p.in[c] = in_cols[c];
        p.out[c] = out_cols[c];
    }
    return PXF_OK;
}

}  // namespace pxf

using namespace pxf;

extern "C" {

size_t pxf_compact_workspace_bytes(int64_t num)
{
    if (num < 0) return 0;
    return (size_t)(ntiles_for(num) + 1) * sizeof(int64_t);
}

int pxf_vignette_mask(const double *l, const double *m, const double *n, int64_t num, uint8_t *mask_out,
                      pxf_stream_t stream)
{
    if (num < 0) return PXF_EINVAL;
    if (num == 0) return PXF_OK;
    if (!l || !m || !n || !mask_out) return PXF_EINVAL;
    const int vec2 = ((((uintptr_t)l | (uintptr_t)m | (uintptr_t)n) & 15) == 0) && (((uintptr_t)mask_out & 1) == 0);
    hipLaunchKernelGGL(vignette_mask_kernel, dim3(grid_for(vec2 ? (num + 1) / 2 : num)), dim3(kBlock), 0,
                       (hipStream_t)stream, l, m, n, num, mask_out, vec2);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PXF_OK : hip_fail(e);
}

int pxf_compact_scan(const uint8_t *mask, int64_t num, void *workspace, int64_t *total_dev,
                     pxf_stream_t stream)
{
    if (num < 0 || !workspace || (num > 0 && !mask)) return PXF_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    int64_t *offs = (int64_t *)workspace;
    const int64_t nt = ntiles_for(num);
    if (nt > 0x7fffffffLL) return PXF_ERANGE;
    if (nt > 0) hipLaunchKernelGGL(tile_count_kernel, dim3((unsigned)nt), dim3(kBlock), 0, s, mask, num, offs);
    hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, s, offs, nt, total_dev);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PXF_OK : hip_fail(e);
}

int pxf_compact_scatter(const double *const *in_cols, double *const *out_cols, int32_t ncols, int64_t num,
                        const uint8_t *mask, const void *workspace, pxf_stream_t stream)
{
    if (num == 0) return PXF_OK;
    ColPtrs p{};
    int rc = fill_ptrs(p, in_cols, out_cols, ncols);
    if (rc != PXF_OK) return rc;
    if (num < 0 || !workspace || !mask) return PXF_EINVAL;
    const int64_t nt = ntiles_for(num);
    if (nt > 0x7fffffffLL) return PXF_ERANGE;
    hipLaunchKernelGGL(compact_scatter_kernel, dim3((unsigned)nt), dim3(kBlock), 0, (hipStream_t)stream, p,
                       (int)ncols, num, mask, (const int64_t *)workspace);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PXF_OK : hip_fail(e);
}

int pxf_gather(const double *const *in_cols, double *const *out_cols, int32_t ncols, const int64_t *index,
               int64_t nidx, int64_t num, pxf_stream_t stream)
{
    if (nidx == 0) return PXF_OK;
    ColPtrs p{};
    int rc = fill_ptrs(p, in_cols, out_cols, ncols);
    if (rc != PXF_OK) return rc;
    if (nidx < 0 || num < 0 || !index) return PXF_EINVAL;
    hipLaunchKernelGGL(gather_kernel, dim3(grid_for(nidx)), dim3(kBlock), 0, (hipStream_t)stream, p, (int)ncols,
                       index, nidx, num, (int *)nullptr);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PXF_OK : hip_fail(e);
}

int pxf_mask_from_index(const int64_t *index, int64_t nidx, int64_t num, uint8_t *mask_out,
                        pxf_stream_t stream)
{
    if (nidx < 0 || num < 0 || !mask_out || (nidx > 0 && !index)) return PXF_EINVAL;
    if (num == 0) return PXF_OK;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(mask_out, 0, (size_t)num, s);
    if (e != hipSuccess) return hip_fail(e);
    if (nidx > 0)
        hipLaunchKernelGGL(mask_from_index_kernel, dim3(grid_for(nidx)), dim3(kBlock), 0, s, index, nidx, num,
                           mask_out);
    e = hipGetLastError();
    return e == hipSuccess ? PXF_OK : hip_fail(e);
}

}  // extern "C"
