// pxf_kernel.h -- the one-thread-per-ray streaming skeleton every per-ray kernel is built on.
//
// gfx950 design notes:
//  * A ray bundle is ten separate fp64 columns (SoA).  A thread owns V=2 CONSECUTIVE rays, so
//    every column access is one 16-byte `global_load_dwordx4` / `global_store_dwordx4` per lane
//    = 1 KiB per wave-instruction, the widest coalesced access the hardware has.  V=1 is the
//    fallback for columns that are only 8-byte aligned (views into a larger tensor).
//  * All loads of an iteration are issued back-to-back before any arithmetic (6-10 independent
//    16-B loads in flight per lane), stores at the end.  The grid is UNCAPPED (one workgroup per
//    512 rays) and accesses are non-temporal once the bundle outgrows the 256 MiB Infinity Cache:
//    both measured (tools/ab_kernels.py), together +10 % over "8 workgroups/CU + grid-stride".
//  * An Op declares at compile time which columns it READS and which it WRITES; nothing else is
//    touched, which is what makes the HBM traffic equal the algorithmic byte count.
//  * `apply` returns the set of columns to store for that ray (conic must leave position and
//    normal untouched on a failed ray, surfacesf.f95:342-345); masked-out rays store nothing.
//  * No MFMA: nothing here is a contraction.  No LDS either except in the Zernike kernel.
//  * Compiled with -ffp-contract=off: the x86 reference has no FMA, and with IEEE fp64
//    add/mul/div/sqrt the results are then bit-identical to it.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/pxf.h"

// build-time tunable (A/B'd with tools/ab_kernels.py)
#ifndef PXF_BLOCK
#define PXF_BLOCK 256
#endif

namespace pxf {

enum Col : int { OPD = 0, X, Y, Z, L, M, N, UX, UY, UZ, AUX, NCOL };
constexpr unsigned B(int c) { return 1u << c; }
constexpr unsigned POS = B(X) | B(Y) | B(Z);
constexpr unsigned DIR = B(L) | B(M) | B(N);
constexpr unsigned NRM = B(UX) | B(UY) | B(UZ);

struct Cols {
    double *p[NCOL];
};

constexpr int kBlock = PXF_BLOCK;
constexpr int kMaxBlocksPerCU = 8;      // reductions: partial rows per CU
// streaming kernels: grid cap per CU; 0 = uncapped, one workgroup per 512 rays (measured +6 % over
// the usual "8 per CU + grid-stride": in-order dispatch keeps all CUs on neighbouring DRAM pages)
constexpr int kDefaultBlocksPerCU = 0;
constexpr int kNewtonCap = 1000;  // the reference has none (it would hang); never reached on
                                  // inputs the reference terminates on

// Fortran single-precision literals widened to fp64 (woltsurf.f95:30,83,136; zernsurf.f95:31)
#define PXF_TOL_1EM8 ((double)1.e-8f)
#define PXF_TOL_1EM10 ((double)1.e-10f)

typedef double d2_t __attribute__((ext_vector_type(2)));

// NT = non-temporal access.  Measured on MI355X (tools/ab_kernels.py, interleaved rounds): for a
// bundle that does not fit the 256 MiB Infinity Cache, nt loads + nt stores are worth +4 % over
// plain accesses (6.2 vs 5.9 TB/s); for a bundle that DOES fit, plain accesses keep the columns
// on die between the launches of a chain and win by 20 % (6.6 vs 5.5 TB/s at 3e6 rays).  So NT is
// a runtime choice per launch (use_nt() below), compiled as a template parameter.
template <int V, bool NT = false>
__device__ __forceinline__ void load_col(const double *p, int64_t v, double (&r)[V])
{
    if constexpr (V == 1) {
        r[0] = NT ? __builtin_nontemporal_load(p + v) : p[v];
    } else {  // V = 2, 4: V/2 consecutive 16-byte accesses
        const d2_t *q = reinterpret_cast<const d2_t *>(p + V * v);
#pragma unroll
        for (int h = 0; h < V / 2; h++) {
            const d2_t t = NT ? __builtin_nontemporal_load(q + h) : q[h];
            r[2 * h] = t.x;
            r[2 * h + 1] = t.y;
        }
    }
}

template <int V, bool NT = false>
__device__ __forceinline__ void store_col(double *p, int64_t v, const double (&r)[V], const bool (&w)[V])
{
    if constexpr (V == 1) {
        if (w[0]) {
            if (NT)
                __builtin_nontemporal_store(r[0], p + v);
            else
                p[v] = r[0];
        }
    } else {
#pragma unroll
        for (int h = 0; h < V / 2; h++) {
            double *q = p + V * v + 2 * h;
            if (w[2 * h] && w[2 * h + 1]) {
                d2_t t;
                t.x = r[2 * h];
                t.y = r[2 * h + 1];
                if (NT)
                    __builtin_nontemporal_store(t, reinterpret_cast<d2_t *>(q));
                else
                    *reinterpret_cast<d2_t *>(q) = t;
            } else {
                if (w[2 * h]) q[0] = r[2 * h];
                if (w[2 * h + 1]) q[1] = r[2 * h + 1];
            }
        }
    }
}

// One ray = one column-indexed register file.  Static indexing only (everything unrolls).
struct Ray {
    double c[NCOL];
};

// Default hooks of an Op (kept a macro so Ops stay plain aggregates):
//   MAXV      rays per thread when every column is 16-B aligned
//   prologue  block-wide setup before the ray loop (LDS staging); empty by default
#define PXF_OP_DEFAULTS                  \
    static constexpr int MAXV = 2;       \
    __device__ __forceinline__ void prologue() const {}

// Generic kernel: Op provides
//   static constexpr unsigned RD, WR;            column bitmasks
//   template<int V> __device__ void apply(Ray (&r)[V], const bool (&sel)[V], unsigned (&wm)[V]) const;
// `sel[v]` says whether ray v takes part (mask / tail), `wm[v]` receives the columns to store.
// One workgroup per 256 ray-vectors, NO grid-stride loop.  Round 1 measured the uncapped grid 6 % ahead of "8 workgroups
// per CU + grid-stride" for the streaming kernels, and an XCD-contiguous tile mapping 4 % behind the identity one
// (profiles/r01_ab_grid_nt_variants.txt; both knobs are gone since round 3).  Round 3 found what the loop cost the
// compute-bound kernels: every wave-uniform quantity of the Op is loop-invariant, is hoisted in front of the loop and,
// for want of SGPRs, spilled to VGPR lanes (the W-S and Legendre-Legendre kernels carried 50-100 v_writelane /
// v_readlane each; without the loop: none).
// An Op may ask for a minimum occupancy (waves per SIMD) with `static constexpr int MIN_WAVES`: the
// second __launch_bounds__ argument caps its VGPR budget accordingly (512 / MIN_WAVES per lane).
// The default is 2, never 1: a 512-register budget lets the compiler park VGPRs in AGPRs, and its reload path for
// those has been seen to drop half of a 64-bit copy (profiles/r02_zernike_miscompile.md).  Every kernel of the
// library -- ray kernels through this default, the others through their own `__launch_bounds__(kBlock, 2)` -- is
// therefore compiled for at most 256 registers; tests/test_kernel_resources.py audits the built code objects.
template <class T, class = void>
struct min_waves {
    static constexpr int value = 2;
};
template <class T>
struct min_waves<T, decltype((void)T::MIN_WAVES)> {
    static constexpr int value = T::MIN_WAVES;
};

template <class Op, int V, bool MASKED, bool NT>
__global__ __launch_bounds__(kBlock, min_waves<Op>::value) void ray_kernel(Cols c, int64_t first, int64_t nvec,
                                                     const uint8_t *__restrict__ mask, Op op)
{
    op.prologue();  // block-wide setup (LDS staging); no-op for most Ops
    const int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (v < nvec) {
        Ray r[V];
        bool sel[V];
        unsigned wm[V];
        const int64_t vv = first + v;  // vector index (units of V rays)
#pragma unroll
        for (int col = 0; col < NCOL; col++) {
            if (Op::RD & B(col)) {
                double t[V];
                load_col<V, NT>(c.p[col], vv, t);
#pragma unroll
                for (int k = 0; k < V; k++) r[k].c[col] = t[k];
            }
        }
#pragma unroll
        for (int k = 0; k < V; k++) {
            sel[k] = MASKED ? (mask[vv * V + k] != 0) : true;
            wm[k] = 0;
        }
        op.template apply<V>(r, sel, wm);
#pragma unroll
        for (int col = 0; col < NCOL; col++) {
            if (Op::WR & B(col)) {
                double t[V];
                bool w[V];
#pragma unroll
                for (int k = 0; k < V; k++) {
                    t[k] = r[k].c[col];
                    w[k] = sel[k] && (wm[k] & B(col));
                }
                store_col<V, NT>(c.p[col], vv, t, w);
            }
        }
    }
}

// ---- host side ---------------------------------------------------------------------------
// cos / sin of a host-side angle exactly as the reference BINARY forms them: both compilers (flang for the reference,
// gcc for the oracle) merge the `cos(theta)`, `sin(theta)` of rotatevector (transformationsf.f95:10-22) and of spoCone
// (woltsurf.f95:605-606) into ONE call of the C library's `sincos` -- which in glibc is built without FMA and differs
// from `sin()` / `cos()` (the FMA builds) in the last bit for about one argument in a thousand.  Calling sin() and
// cos() made a frame change with such an angle differ from the reference by an ulp in a fifth of the rays (found by a
// 600-seed soak of tests/test_gpu_fuzz.py; angle -0.1610095633704616).  Hence sincos().
inline void host_sincos(double a, double &c, double &s) { ::sincos(a, &s, &c); }

extern thread_local int g_last_hip_error;
int device_cus();
int blocks_per_cu_cap();
bool use_nt(int64_t num);  // bundle (80 B/ray) larger than the Infinity Cache -> non-temporal accesses  // 0 = no cap: one workgroup per 256 ray-vectors (env PXF_BPC overrides)

bool device_usable();  // the current device is a gfx950 (cached per device ordinal)

// A failed HIP call -> the C ABI's code.  "This device cannot run the library's gfx950 code object" (no device, an
// invalid one, no binary for the GPU, kernel symbol not found on it) is PXF_ENODEV; everything else PXF_EHIP.
inline int hip_fail(hipError_t e)
{
    g_last_hip_error = (int)e;
    switch (e) {
        case hipErrorNoDevice:
        case hipErrorInvalidDevice:
        case hipErrorNoBinaryForGpu:
        case hipErrorInvalidDeviceFunction:
        case hipErrorSharedObjectInitFailed:
            return PXF_ENODEV;
        default:
            return device_usable() ? PXF_EHIP : PXF_ENODEV;
    }
}

inline int grid_for(int64_t nvec)
{
    int64_t want = (nvec + kBlock - 1) / kBlock;
    const int bpc = blocks_per_cu_cap();
    int64_t cap = bpc > 0 ? (int64_t)device_cus() * bpc : 0x7fffffffLL;
    return (int)(want < cap ? want : cap);
}

// one workgroup per kBlock vectors, never capped: for the kernels that have no grid-stride loop
inline unsigned grid_exact(int64_t nvec) { return (unsigned)((nvec + kBlock - 1) / kBlock); }

inline int grid_capped(int64_t nvec)  // reductions: one partial row per workgroup, bounded workspace
{
    int64_t want = (nvec + kBlock - 1) / kBlock;
    int64_t cap = (int64_t)device_cus() * kMaxBlocksPerCU;
    return (int)(want < cap ? want : cap);
}

template <class Op>
int launch_rays(const Cols &c, int64_t num, const uint8_t *mask, pxf_stream_t stream, const Op &op)
{
    if (num < 0) return PXF_EINVAL;
    if (num == 0) return PXF_OK;  // an empty bundle is a no-op (its columns may be NULL)
    if (!device_usable()) return PXF_ENODEV;
    bool aligned = true;
    for (int col = 0; col < NCOL; col++) {
        if ((Op::RD | Op::WR) & B(col)) {
            if (c.p[col] == nullptr) return PXF_EINVAL;
            if ((uintptr_t)c.p[col] & 15) aligned = false;
        }
    }
    hipStream_t s = (hipStream_t)stream;
    const bool nt = use_nt(num);
    int64_t nvec2 = 0;
#define PXF_LAUNCH(V, M, T, FIRST, NVEC)                                                         \
    hipLaunchKernelGGL((ray_kernel<Op, V, M, T>), dim3(grid_exact(NVEC)), dim3(kBlock), 0, s, c, FIRST, NVEC, mask, op)
#define PXF_LAUNCH_V(V, FIRST, NVEC)                 \
    do {                                             \
        if (mask) {                                  \
            if (nt) PXF_LAUNCH(V, true, true, FIRST, NVEC);   \
            else PXF_LAUNCH(V, true, false, FIRST, NVEC);     \
        } else {                                     \
            if (nt) PXF_LAUNCH(V, false, true, FIRST, NVEC);  \
            else PXF_LAUNCH(V, false, false, FIRST, NVEC);    \
        }                                            \
    } while (0)
    if constexpr (Op::MAXV >= 2) {
        nvec2 = aligned ? num / 2 : 0;
        if (nvec2 > 0) PXF_LAUNCH_V(2, (int64_t)0, nvec2);
    }
    const int64_t done = nvec2 * 2;
    if (done < num) {  // unaligned columns, or the odd last ray
        const int64_t rest = num - done;
        PXF_LAUNCH_V(1, done, rest);
    }
#undef PXF_LAUNCH_V
#undef PXF_LAUNCH
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PXF_OK : hip_fail(e);
}

}  // namespace pxf
