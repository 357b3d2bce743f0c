// pxf_wavefront.hip -- SURVEY 8f row f-4: the wavefront-sensor pair of reconstruct.f95.
//
//   pxf_southwellbin   southwellbin (reconstruct.f95:139-187): bin the ray direction cosines l, m on
//                      a lenslet grid by ray position, mean per bin -> slope tan(asin(.)), empty
//                      bins get the sentinel 100.  HBM-bound: 32 B/ray read, nothing written per ray.
//   pxf_reconstruct    reconstruct (reconstruct.f95:1-131): successive over-relaxation of the
//                      Southwell geometry in Gauss-Seidel order.
//
// All grids are Fortran-ordered like the reference's arrays: a(xi,yi) = a[xi + xdim*yi].
//
// southwellbin: bins are private per workgroup in LDS (ds_add_f64 / ds_add_u32), flushed with one
// global atomic per touched bin; grids larger than one LDS tile (8000 bins) are done tile by tile,
// one pass over the rays' x, y per tile (device-scope fp64 atomics straight to memory measured 11x
// slower than that).  The per-bin sums are order-nondeterministic (the reference's own
// OpenMP loop is a data race on the same sums): parity of the means is to ~1e-15 relative x
// sqrt(rays per bin), asserted at 1e-12 in the tests.  Rays outside the grid, which index out of
// bounds in the reference, are dropped.
//
// reconstruct: a Gauss-Seidel sweep in the reference's order (xi outer, yi inner) only ever makes
// cell (xi,yi) depend on the CURRENT-sweep values of (xi-1,yi), (xi,yi-1) and the PREVIOUS-sweep
// values of (xi+1,yi), (xi,yi+1): all cells of one anti-diagonal xi+yi = d are independent, and
// processing the diagonals in increasing d reproduces the sequential sweep exactly -- including
// the in-sweep invalidation of isolated lenslets.  One workgroup walks the diagonals with a
// barrier between them, the phase grid resident in LDS (up to 160 KB = 141 x 141 lenslets; global
// memory beyond that).  The updates are bit-identical to the reference; only the convergence
// norm is summed in a different order (tree instead of sequential), which can move the stopping
// sweep by one when sqrt(mean square change) lands within rounding of `criteria`.
#include <math.h>

#include "pxf_kernel.h"

namespace pxf {

constexpr int kSwbTileSmall = 4096;  // x (8+8+4) B = 80 KB : two 512-thread workgroups per CU
constexpr int kSwbTileLarge = 8000;  // 160 KB of LDS: one 1024-thread workgroup per CU
constexpr int kSwbBlock = 256;

__device__ __forceinline__ bool swb_bin(double v, double binsize, int dim, int &b)
{
    // reconstruct.f95:158-167 (1-based there): even: floor(v/binsize) + dim/2 + 2;
    // odd: floor((v+binsize/2)/binsize) + (dim-1)/2 + 1
    const double f = (dim % 2 == 0) ? floor(v / binsize) + (double)(dim / 2 + 1)
                                    : floor((v + binsize / 2) / binsize) + (double)((dim - 1) / 2);
    if (!(f >= 0. && f < (double)dim)) return false;  // also drops NaN
    b = (int)f;
    return true;
}

// Bins [tile0, tile0+tilen) are private to the workgroup in LDS (dynamic: tilen x (8+8+4) B); rays
// that fall into other tiles are skipped (their l, m are not even read).  One launch per tile.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void swb_accum_kernel(const double *__restrict__ x, const double *__restrict__ y,
                                                         const double *__restrict__ l, const double *__restrict__ m,
                                                         int64_t num, double binsize, double *__restrict__ xang,
                                                         double *__restrict__ yang, unsigned *__restrict__ accum,
                                                         int xdim, int ydim, int tile0, int tilen)
{
    extern __shared__ double s_swb[];
    double *s_l = s_swb, *s_m = s_swb + tilen;
    unsigned *s_c = reinterpret_cast<unsigned *>(s_swb + 2 * tilen);
    for (int i = threadIdx.x; i < tilen; i += BLOCK) {
        s_l[i] = 0.;
        s_m[i] = 0.;
        s_c[i] = 0u;
    }
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < num; i += stride) {
        const double xv = __builtin_nontemporal_load(x + i), yv = __builtin_nontemporal_load(y + i);
        int xb, yb;
        if (!swb_bin(xv, binsize, xdim, xb) || !swb_bin(yv, binsize, ydim, yb)) continue;
        const int o = xb + xdim * yb - tile0;
        if (o < 0 || o >= tilen) continue;
        const double lv = __builtin_nontemporal_load(l + i), mv = __builtin_nontemporal_load(m + i);
        unsafeAtomicAdd(&s_l[o], lv);
        unsafeAtomicAdd(&s_m[o], mv);
        atomicAdd(&s_c[o], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < tilen; i += BLOCK) {
        const unsigned c = s_c[i];
        if (c) {
            unsafeAtomicAdd(&xang[tile0 + i], s_l[i]);
            unsafeAtomicAdd(&yang[tile0 + i], s_m[i]);
            atomicAdd(&accum[tile0 + i], c);
        }
    }
}

__global__ __launch_bounds__(kSwbBlock) void swb_final_kernel(double *__restrict__ xang, double *__restrict__ yang,
                                                             double *__restrict__ phase,
                                                             const unsigned *__restrict__ accum, int nb)
{
    const int i = blockIdx.x * kSwbBlock + threadIdx.x;
    if (i >= nb) return;
    const unsigned c = accum[i];
    if (c == 0u) {  // :175-178
        phase[i] = 100.;
        xang[i] = 100.;
        yang[i] = 100.;
    } else {  // :180-181 (accum is a default INTEGER there: the quotient is formed in double)
        phase[i] = 0.;
        xang[i] = tan(asin(xang[i] / (double)(int)c));
        yang[i] = tan(asin(yang[i] / (double)(int)c));
    }
}

// ---- reconstruct ---------------------------------------------------------------------------------
constexpr int kRecMaxBlock = 1024;

template <bool PLDS>
__global__ __launch_bounds__(kRecMaxBlock) void reconstruct_kernel(double *xang, double *yang, int xdim, int ydim,
                                                                   double criteria, double h, double w, double *phase,
                                                                   double *phasec, int maxiter, int *sweeps_out)
{
    extern __shared__ double s_dyn[];
    __shared__ double s_red[kRecMaxBlock / 64];
    __shared__ int s_cnt[kRecMaxBlock / 64];
    __shared__ double s_rms;
    const int tid = threadIdx.x, nthr = blockDim.x, nb = xdim * ydim;
    const int lane = tid & 63, wave = tid >> 6, nwave = (nthr + 63) >> 6;
    double *pc = PLDS ? s_dyn : phasec;
    for (int b = tid; b < nb; b += nthr) pc[b] = phase[b];  // phasec = phase (:16)
    __syncthreads();
    const int sx = 1, sy = xdim;
    int counter = 0, sweeps = 0;
    for (;;) {
        double acc = 0.;
        int cnt = 0;
        // interior cells 1 <= xi <= xdim-2, 1 <= yi <= ydim-2 (0-based), by anti-diagonal d = xi + yi
        for (int d = 2; d <= xdim + ydim - 4; d++) {
            const int xlo = max(1, d - (ydim - 2)), xhi = min(xdim - 2, d - 1);
            for (int xi = xlo + tid; xi <= xhi; xi += nthr) {
                const int o = xi * sx + (d - xi) * sy;
                const double pco = pc[o];
                if (pco == 100.) continue;  // :30-32
                double yplus = yang[o + sy], yneg = yang[o - sy], xplus = xang[o + sx], xneg = xang[o - sx];
                if (yplus == 100.) yplus = -yang[o];
                if (yneg == 100.) yneg = -yang[o];
                if (xplus == 100.) xplus = -xang[o];
                if (xneg == 100.) xneg = -xang[o];
                const double bk = .5 * (yplus - yneg + xplus - xneg) * h;  // :56
                double goodpix = 4.;
                double pyplus = pc[o + sy], pyneg = pc[o - sy], pxplus = pc[o + sx], pxneg = pc[o - sx];
                if (pyplus == 100.) { pyplus = 0.; goodpix -= 1; }
                if (pyneg == 100.) { pyneg = 0.; goodpix -= 1; }
                if (pxplus == 100.) { pxplus = 0.; goodpix -= 1; }
                if (pxneg == 100.) { pxneg = 0.; goodpix -= 1; }
                const double psum = pyplus + pyneg + pxplus + pxneg;
                if (goodpix == 0.) {  // :87-93 isolated lenslet: discard
                    pc[o] = 100.;
                    phase[o] = 100.;
                    xang[o] = 100.;
                    yang[o] = 100.;
                    continue;
                }
                const double pn = pco + w * ((psum + bk) / goodpix - pco);  // :98
                pc[o] = pn;
                cnt += 1;
                // :102 (phasec - phase)**2: `phase` holds the previous sweep's value of this cell, which is what
                // pc[o] held until this very update (phase = phasec after every sweep, :114) -- so the old value
                // is already in a register and `phase` itself is only written once, when the loop ends
                const double df = pn - pco;
                acc += df * df;
            }
            __syncthreads();
        }
        // ||phasec - phase||^2 and the number of updated cells over the workgroup
        for (int off = 32; off > 0; off >>= 1) {
            acc += __shfl_down(acc, off);
            cnt += __shfl_down(cnt, off);
        }
        if (lane == 0) {
            s_red[wave] = acc;
            s_cnt[wave] = cnt;
        }
        __syncthreads();
        if (tid == 0) {
            double t = 0.;
            int c = 0;
            for (int k = 0; k < nwave; k++) {
                t += s_red[k];
                c += s_cnt[k];
            }
            s_rms = sqrt(t / (double)c);  // :112 (0/0 -> NaN -> never "< criteria", as in the reference)
        }
        __syncthreads();
        sweeps++;
        if (s_rms < criteria) break;
        counter++;
        if (counter > maxiter) break;
        __syncthreads();  // s_rms is rewritten next sweep only after every thread has read it
    }
    for (int b = tid; b < nb; b += nthr) {  // :114 phase = phasec, once: nothing reads `phase` inside the loop
        const double v = pc[b];
        phase[b] = v;
        if (PLDS) phasec[b] = v;
    }
    if (tid == 0 && sweeps_out) *sweeps_out = sweeps;
}

}  // namespace pxf

using namespace pxf;

extern "C" {

size_t pxf_southwellbin_workspace_bytes(int32_t xdim, int32_t ydim)
{
    if (xdim < 1 || ydim < 1) return 0;
    return (size_t)xdim * (size_t)ydim * sizeof(unsigned);
}

int pxf_southwellbin(const double *x, const double *y, const double *l, const double *m, int64_t num,
                     double binsize, double *xang, double *yang, double *phase, int32_t xdim, int32_t ydim,
                     void *workspace, pxf_stream_t stream)
{
    if (xdim < 1 || ydim < 1 || (int64_t)xdim * ydim > (int64_t)1 << 30 || num < 0) return PXF_EINVAL;
    if (!xang || !yang || !phase || !workspace || !(binsize > 0.)) return PXF_EINVAL;
    if (num > 0 && (!x || !y || !l || !m)) return PXF_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int nb = xdim * ydim;
    unsigned *accum = (unsigned *)workspace;
    hipError_t e = hipMemsetAsync(xang, 0, (size_t)nb * sizeof(double), s);
    if (e == hipSuccess) e = hipMemsetAsync(yang, 0, (size_t)nb * sizeof(double), s);
    if (e == hipSuccess) e = hipMemsetAsync(accum, 0, (size_t)nb * sizeof(unsigned), s);
    if (e != hipSuccess) return hip_fail(e);
    if (num > 0) {
        const int tile = nb <= kSwbTileSmall ? nb : (nb < kSwbTileLarge ? nb : kSwbTileLarge);
        const bool big = tile > kSwbTileSmall;
        const int block = big ? 1024 : 512;
        const size_t lds = (size_t)tile * (2 * sizeof(double) + sizeof(unsigned));
        if (lds > 48 * 1024) {
            e = big ? hipFuncSetAttribute(reinterpret_cast<const void *>(&swb_accum_kernel<1024>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                    : hipFuncSetAttribute(reinterpret_cast<const void *>(&swb_accum_kernel<512>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return hip_fail(e);
        }
        const int64_t want = (num + block - 1) / block;
        const int64_t cap = (int64_t)device_cus() * (big ? 1 : 2);  // resident workgroups: bounded flush traffic
        const int g = (int)(want < cap ? want : cap);
        for (int t0 = 0; t0 < nb; t0 += tile) {
            const int tn = nb - t0 < tile ? nb - t0 : tile;
            if (big)
                hipLaunchKernelGGL((swb_accum_kernel<1024>), dim3(g), dim3(1024), lds, s, x, y, l, m, num, binsize, xang,
                                   yang, accum, (int)xdim, (int)ydim, t0, tn);
            else
                hipLaunchKernelGGL((swb_accum_kernel<512>), dim3(g), dim3(512), lds, s, x, y, l, m, num, binsize, xang,
                                   yang, accum, (int)xdim, (int)ydim, t0, tn);
        }
    }
    hipLaunchKernelGGL(swb_final_kernel, dim3((nb + kSwbBlock - 1) / kSwbBlock), dim3(kSwbBlock), 0, s, xang, yang, phase,
                       accum, nb);
    e = hipGetLastError();
    return e == hipSuccess ? PXF_OK : hip_fail(e);
}

int pxf_reconstruct(double *xang, double *yang, int32_t xdim, int32_t ydim, double criteria, double h, double *phase,
                    double *phasec, int32_t maxiter, int32_t *sweeps_dev, pxf_stream_t stream)
{
    if (xdim < 1 || ydim < 1 || (int64_t)xdim * ydim > (int64_t)1 << 28) return PXF_EINVAL;
    if (!xang || !yang || !phase || !phasec) return PXF_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    // :13-14 -- pi is a default-REAL literal in the reference; sin() on the host is glibc's, as there
    const double pi = (double)3.1415926535897931f;
    const double w = 2 / (1 + sin(pi / (sqrt((double)xdim * (double)ydim) + 1)));
    int diag = (xdim < ydim ? xdim : ydim) - 2;
    int threads = diag <= 64 ? 64 : ((diag + 63) / 64) * 64;
    if (threads > kRecMaxBlock) threads = kRecMaxBlock;
    const size_t lds = (size_t)xdim * (size_t)ydim * sizeof(double);
    static const size_t lds_cap = [] {  // what one workgroup may take on this device, minus the static arrays
        int v = 0, dev = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) v = 65536;
        return (size_t)(v > 1024 ? v - 1024 : 0);
    }();
    if (lds <= lds_cap) {
        hipError_t e = hipSuccess;
        if (lds > 48 * 1024)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(&reconstruct_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return hip_fail(e);
        hipLaunchKernelGGL((reconstruct_kernel<true>), dim3(1), dim3(threads), lds, s, xang, yang, (int)xdim, (int)ydim,
                           criteria, h, w, phase, phasec, (int)maxiter, (int *)sweeps_dev);
    } else {
        hipLaunchKernelGGL((reconstruct_kernel<false>), dim3(1), dim3(threads), 0, s, xang, yang, (int)xdim, (int)ydim,
                           criteria, h, w, phase, phasec, (int)maxiter, (int *)sweeps_dev);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PXF_OK : hip_fail(e);
}

}  // extern "C"
