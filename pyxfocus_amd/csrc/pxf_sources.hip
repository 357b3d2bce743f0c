// pxf_sources.hip -- sources.pointsource / circularbeam / annulus / subannulus / rectbeam on the device
// (sources.py:20-53, 56-88, 91-127, 130-170, 348-377).
//
// The reference draws `rho` then `theta` as two np.random.rand(num) vectors from NumPy's global
// MT19937 state -- inherently serial and host-only; 1e9 rays would take minutes and 80 GB of PCIe
// traffic.  Here each ray draws from a COUNTER-BASED generator, Philox4x32-10 (Salmon et al.,
// SC'11), keyed on (seed) with counter = GLOBAL ray index: no state, no ordering, and a bundle
// sharded over N GPUs is bit-identical to the single-GPU bundle.  The distributions are the
// reference's (same formulas, same 53-bit uniforms in [0,1)); the random STREAM differs, so
// parity with a seeded NumPy run is distributional, not per-ray (the Python layer keeps the
// NumPy path for per-ray reproducibility at host-sized bundles).
#include <math.h>

#include "pxf_kernel.h"

namespace pxf {

__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1)
{
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
        const uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
        const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0;
        c[1] = lo1;
        c[2] = n2;
        c[3] = lo0;
        k0 += W0;
        k1 += W1;
    }
}

// two 53-bit uniforms in [0,1) per global ray index (same construction as NumPy's random_sample);
// `draw` selects further independent pairs for the same ray (third counter word)
__device__ __forceinline__ void uniform2(uint64_t seed, uint64_t index, double &u1, double &u2, uint32_t draw = 0u)
{
    uint32_t c[4] = {(uint32_t)index, (uint32_t)(index >> 32), draw, 0u};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const uint64_t a = ((uint64_t)c[0] << 32) | c[1], b = ((uint64_t)c[2] << 32) | c[3];
    u1 = (double)(a >> 11) * (1.0 / 9007199254740992.0);
    u2 = (double)(b >> 11) * (1.0 / 9007199254740992.0);
}

struct SrcParams {
    double p0, p1, p2, p3;
    uint64_t seed, first;
    double p4 = 0., p5 = 0., p6 = 0.;  // extended kinds only
    int64_t total = 0;                  // bundle length (grids need it: linspace(.., num))
};

// np.linspace(start, stop, n)[i]: i*step + start with step = (stop-start)/(n-1), last element = stop
__device__ __forceinline__ double linspace_at(double start, double stop, int64_t n, int64_t i)
{
    if (n <= 1) return start;
    if (i == n - 1) return stop;
    const double step = (stop - start) / (double)(n - 1);
    return (double)i * step + start;
}

template <int KIND>
__device__ __forceinline__ void make_ray(const SrcParams &q, uint64_t gidx, double (&c)[10])
{
    double u1, u2;
    uniform2(q.seed, gidx, u1, u2);
#pragma unroll
    for (int k = 0; k < 10; k++) c[k] = 0.;
    const double twopi = 2 * 3.141592653589793;
    if (KIND == PXF_SRC_POINTSOURCE) {  // sources.py:39-44
        const double rho = sqrt(u1) * sin(q.p0);
        const double th = u2 * twopi;
        c[L] = rho * cos(th);
        c[M] = rho * sin(th);
        c[N] = sqrt(1. - c[L] * c[L] - c[M] * c[M]);
    } else if (KIND == PXF_SRC_CIRCULARBEAM) {  // sources.py:74-81
        const double rho = sqrt(u1) * q.p0;
        const double th = u2 * twopi;
        c[X] = rho * cos(th);
        c[Y] = rho * sin(th);
        c[N] = 1.;
    } else if (KIND == PXF_SRC_ANNULUS) {  // sources.py:114-121
        const double rho = sqrt(q.p0 * q.p0 + u1 * (q.p1 * q.p1 - q.p0 * q.p0));
        const double th = u2 * twopi;
        c[X] = rho * cos(th);
        c[Y] = rho * sin(th);
        c[N] = q.p3;
    } else if (KIND == PXF_SRC_SUBANNULUS) {  // sources.py:157-164
        const double rho = sqrt(q.p0 * q.p0 + u1 * (q.p1 * q.p1 - q.p0 * q.p0));
        const double th = u2 * q.p2 - q.p2 / 2.;
        c[X] = rho * cos(th);
        c[Y] = rho * sin(th);
        c[N] = q.p3;
    } else if (KIND == PXF_SRC_RECTBEAM) {  // sources.py:364-367
        c[X] = (u1 - .5) * 2 * q.p0;
        c[Y] = (u2 - .5) * 2 * q.p1;
        c[N] = 1.;
    } else if (KIND == PXF_SRC_XSLIT) {  // sources.py:195-204 (deterministic)
        c[X] = linspace_at(q.p0, q.p1, q.total, (int64_t)gidx);
        c[N] = q.p3;
    } else if (KIND == PXF_SRC_RECTARRAY) {  // sources.py:230-245: meshgrid(linspace x, linspace y).flatten()
        const int64_t nside = (int64_t)q.p2;
        const int64_t ix = (int64_t)gidx % nside, iy = (int64_t)gidx / nside;
        c[X] = linspace_at(-q.p0, q.p0, nside, ix);
        c[Y] = linspace_at(-q.p1, q.p1, nside, iy);
        c[N] = 1.;
    } else if (KIND == PXF_SRC_CONVERGINGBEAM || KIND == PXF_SRC_CONVERGINGBEAM2) {
        // sources.py:279-295 / 327-343: p0 zset; beam: p1 rin, p2 rout, p3 tmin, p4 tmax | beam2: p1 xmin,
        // p2 xmax, p3 ymin, p4 ymax; p5 lscat [arcsec], Lorentzian angular scatter
        double u3, u4;
        uniform2(q.seed, gidx, u3, u4, 1u);
        double x, y, rho, th;
        if (KIND == PXF_SRC_CONVERGINGBEAM) {
            rho = sqrt(q.p1 * q.p1 + u1 * (q.p2 * q.p2 - q.p1 * q.p1));
            th = q.p3 + u2 * (q.p4 - q.p3);
            x = rho * cos(th);
            y = rho * sin(th);
        } else {
            x = q.p1 + u1 * (q.p2 - q.p1);
            y = q.p3 + u2 * (q.p4 - q.p3);
            rho = sqrt(x * x + y * y);
            th = atan2(y, x);
        }
        double ls = q.p5 * tan((u3 - .5) * 3.141592653589793);
        ls = ls / (60. * 60.) * 3.141592653589793 / 180.;
        const double nn = -cos(atan(rho / q.p0) + ls);
        const double s = sqrt(1 - nn * nn);
        c[X] = x;
        c[Y] = y;
        c[Z] = q.p0;
        c[L] = -s * cos(th);
        c[M] = -s * sin(th);
        c[N] = nn;
    } else if (KIND == PXF_SRC_GAUSSIANBEAM) {  // sources.py:404-406, two normals by Box-Muller
        const double rr = sqrt(-2. * log(1. - u1));  // 1-u1 in (0,1]
        const double sg = sin(q.p0) / sqrt(2.);
        c[L] = rr * cos(twopi * u2) * sg;
        c[M] = rr * sin(twopi * u2) * sg;
        c[N] = sqrt(1. - c[L] * c[L] - c[M] * c[M]);
    } else if (KIND == PXF_SRC_FANBEAM) {  // sources.py:420-431: p0 xang, p1 yang, p2 n per side
        const int64_t nside = (int64_t)q.p2;
        const int64_t ix = (int64_t)gidx % nside, iy = (int64_t)gidx / nside;
        c[L] = sin(linspace_at(-q.p0, q.p0, nside, ix));
        c[M] = sin(linspace_at(-q.p1, q.p1, nside, iy));
        c[N] = sqrt(1. - c[L] * c[L] - c[M] * c[M]);
    } else {  // PXF_SRC_CIRCFAN, sources.py:445-461: p0 halfang, p1 rings, p2 arms; index = arm*rings + ring
        const int64_t rings = (int64_t)q.p1, arms = (int64_t)q.p2;
        const int64_t ir = (int64_t)gidx % rings, ia = (int64_t)gidx / rings;
        const double rad = linspace_at(0., q.p0, rings, ir);
        const double az = linspace_at(0., twopi, arms + 1, ia);
        c[L] = sin(sin(rad) * cos(az));
        c[M] = sin(sin(rad) * sin(az));
        c[N] = sqrt(1. - c[L] * c[L] - c[M] * c[M]);
    }
}

template <int KIND, int V, bool NT>
__global__ __launch_bounds__(kBlock, 2) void source_kernel(Cols cols, int64_t first, int64_t nvec, SrcParams q)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < nvec; v += stride) {
        const int64_t vv = first + v;
        double r[V][10];
#pragma unroll
        for (int k = 0; k < V; k++) make_ray<KIND>(q, q.first + (uint64_t)(vv * V + k), r[k]);
#pragma unroll
        for (int col = 0; col < 10; col++) {
            double t[V];
            bool w[V];
#pragma unroll
            for (int k = 0; k < V; k++) {
                t[k] = r[k][col];
                w[k] = true;
            }
            store_col<V, NT>(cols.p[col], vv, t, w);
        }
    }
}

// segmented source: ray i belongs to segment i / rps and takes its four parameters from a device
// table [nseg][4] (consecutive rays read the same 32 bytes: one L1 line per wave)
template <int KIND, bool NT>
__global__ __launch_bounds__(kBlock, 2) void source_seg_kernel(Cols cols, int64_t nvec, SrcParams q,
                                                            const double *__restrict__ table, int64_t rps)
{
    constexpr int V = 2;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < nvec; v += stride) {
        const double *t4 = table + ((v * V) / rps) * 4;
        SrcParams qq = q;
        qq.p0 = t4[0];
        qq.p1 = t4[1];
        qq.p2 = t4[2];
        qq.p3 = t4[3];
        double r[V][10];
#pragma unroll
        for (int k = 0; k < V; k++) make_ray<KIND>(qq, q.first + (uint64_t)(v * V + k), r[k]);
#pragma unroll
        for (int col = 0; col < 10; col++) {
            double t[V];
            bool w[V];
#pragma unroll
            for (int k = 0; k < V; k++) {
                t[k] = r[k][col];
                w[k] = true;
            }
            store_col<V, NT>(cols.p[col], v, t, w);
        }
    }
}

template <int KIND>
static void launch_seg_kind(const Cols &c, int64_t num, const SrcParams &q, const double *table, int64_t rps,
                            hipStream_t s)
{
    const int64_t nvec = num / 2;
    if (use_nt(num))
        hipLaunchKernelGGL((source_seg_kernel<KIND, true>), dim3(grid_for(nvec)), dim3(kBlock), 0, s, c, nvec, q, table, rps);
    else
        hipLaunchKernelGGL((source_seg_kernel<KIND, false>), dim3(grid_for(nvec)), dim3(kBlock), 0, s, c, nvec, q, table, rps);
}

template <int KIND>
static void launch_kind(const Cols &c, bool aligned, int64_t num, const SrcParams &q, hipStream_t s)
{
    const int64_t nvec2 = aligned ? num / 2 : 0;
    const bool nt = use_nt(num);
    if (nvec2 > 0) {
        if (nt)
            hipLaunchKernelGGL((source_kernel<KIND, 2, true>), dim3(grid_for(nvec2)), dim3(kBlock), 0, s, c,
                               (int64_t)0, nvec2, q);
        else
            hipLaunchKernelGGL((source_kernel<KIND, 2, false>), dim3(grid_for(nvec2)), dim3(kBlock), 0, s, c,
                               (int64_t)0, nvec2, q);
    }
    const int64_t done = nvec2 * 2;
    if (done < num)
        hipLaunchKernelGGL((source_kernel<KIND, 1, false>), dim3(grid_for(num - done)), dim3(kBlock), 0, s, c,
                           done, num - done, q);
}

}  // namespace pxf

using namespace pxf;

extern "C" int pxf_source(int32_t kind, double *const *cols, int64_t num, double p0, double p1, double p2,
                          double p3, uint64_t seed, uint64_t first_index, pxf_stream_t stream)
{
    if (num < 0 || !cols) return PXF_EINVAL;
    if (num == 0) return PXF_OK;
    Cols c{};
    bool aligned = true;
    for (int k = 0; k < 10; k++) {
        if (!cols[k]) return PXF_EINVAL;
        c.p[k] = cols[k];
        if ((uintptr_t)cols[k] & 15) aligned = false;
    }
    if (num == 0) return PXF_OK;
    SrcParams q{p0, p1, p2, p3, seed, first_index};
    hipStream_t s = (hipStream_t)stream;
    switch (kind) {
        case PXF_SRC_POINTSOURCE: launch_kind<PXF_SRC_POINTSOURCE>(c, aligned, num, q, s); break;
        case PXF_SRC_CIRCULARBEAM: launch_kind<PXF_SRC_CIRCULARBEAM>(c, aligned, num, q, s); break;
        case PXF_SRC_ANNULUS: launch_kind<PXF_SRC_ANNULUS>(c, aligned, num, q, s); break;
        case PXF_SRC_SUBANNULUS: launch_kind<PXF_SRC_SUBANNULUS>(c, aligned, num, q, s); break;
        case PXF_SRC_RECTBEAM: launch_kind<PXF_SRC_RECTBEAM>(c, aligned, num, q, s); break;
        default: return PXF_EINVAL;
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PXF_OK : hip_fail(e);
}

extern "C" int pxf_source_ex(int32_t kind, double *const *cols, int64_t num, int64_t total, const double *params,
                             uint64_t seed, uint64_t first_index, pxf_stream_t stream)
{
    if (num < 0 || !cols || !params || total < num) return PXF_EINVAL;
    if (num == 0) return PXF_OK;
    Cols c{};
    bool aligned = true;
    for (int k = 0; k < 10; k++) {
        if (!cols[k]) return PXF_EINVAL;
        c.p[k] = cols[k];
        if ((uintptr_t)cols[k] & 15) aligned = false;
    }
    SrcParams q{params[0], params[1], params[2], params[3], seed, first_index, params[4], params[5], params[6], total};
    hipStream_t s = (hipStream_t)stream;
    switch (kind) {
        case PXF_SRC_XSLIT: launch_kind<PXF_SRC_XSLIT>(c, aligned, num, q, s); break;
        case PXF_SRC_RECTARRAY: launch_kind<PXF_SRC_RECTARRAY>(c, aligned, num, q, s); break;
        case PXF_SRC_CONVERGINGBEAM: launch_kind<PXF_SRC_CONVERGINGBEAM>(c, aligned, num, q, s); break;
        case PXF_SRC_CONVERGINGBEAM2: launch_kind<PXF_SRC_CONVERGINGBEAM2>(c, aligned, num, q, s); break;
        case PXF_SRC_GAUSSIANBEAM: launch_kind<PXF_SRC_GAUSSIANBEAM>(c, aligned, num, q, s); break;
        case PXF_SRC_FANBEAM: launch_kind<PXF_SRC_FANBEAM>(c, aligned, num, q, s); break;
        case PXF_SRC_CIRCFAN: launch_kind<PXF_SRC_CIRCFAN>(c, aligned, num, q, s); break;
        default: return PXF_EINVAL;
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PXF_OK : hip_fail(e);
}

extern "C" int pxf_source_segments(int32_t kind, double *const *cols, int64_t num, const double *seg_params_dev,
                                   int32_t nseg, int64_t rays_per_seg, uint64_t seed, uint64_t first_index,
                                   pxf_stream_t stream)
{
    if (num < 0 || !cols || !seg_params_dev || nseg < 1) return PXF_EINVAL;
    if (rays_per_seg < 2 || (rays_per_seg & 1) || num != (int64_t)nseg * rays_per_seg) return PXF_EINVAL;
    Cols c{};
    for (int k = 0; k < 10; k++) {
        if (!cols[k] || ((uintptr_t)cols[k] & 15)) return PXF_EINVAL;
        c.p[k] = cols[k];
    }
    SrcParams q{0., 0., 0., 0., seed, first_index};
    hipStream_t s = (hipStream_t)stream;
    switch (kind) {
        case PXF_SRC_POINTSOURCE: launch_seg_kind<PXF_SRC_POINTSOURCE>(c, num, q, seg_params_dev, rays_per_seg, s); break;
        case PXF_SRC_CIRCULARBEAM: launch_seg_kind<PXF_SRC_CIRCULARBEAM>(c, num, q, seg_params_dev, rays_per_seg, s); break;
        case PXF_SRC_ANNULUS: launch_seg_kind<PXF_SRC_ANNULUS>(c, num, q, seg_params_dev, rays_per_seg, s); break;
        case PXF_SRC_SUBANNULUS: launch_seg_kind<PXF_SRC_SUBANNULUS>(c, num, q, seg_params_dev, rays_per_seg, s); break;
        case PXF_SRC_RECTBEAM: launch_seg_kind<PXF_SRC_RECTBEAM>(c, num, q, seg_params_dev, rays_per_seg, s); break;
        default: return PXF_EINVAL;
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PXF_OK : hip_fail(e);
}
