// pxf_kernel.h -- the one-thread-per-ray streaming skeleton every per-ray kernel is built on.
//
// gfx950 design notes:
//  * A ray bundle is ten separate fp64 columns (SoA).  A thread owns V=2 CONSECUTIVE rays, so
//    every column access is one 16-byte `global_load_dwordx4` / `global_store_dwordx4` per lane
//    = 1 KiB per wave-instruction, the widest coalesced access the hardware has.  V=1 is the
//    fallback for columns that are only 8-byte aligned (views into a larger tensor).
//  * All loads of an iteration are issued back-to-back before any arithmetic (6-10 independent
//    16-B loads in flight per lane), stores at the end; the grid is capped at 8 blocks per CU
//    and grid-strides the rest.
//  * An Op declares at compile time which columns it READS and which it WRITES; nothing else is
//    touched, which is what makes the HBM traffic equal the algorithmic byte count.
//  * `apply` returns the set of columns to store for that ray (conic must leave position and
//    normal untouched on a failed ray, surfacesf.f95:342-345); masked-out rays store nothing.
//  * No MFMA: nothing here is a contraction.  No LDS either except in the Zernike kernel.
//  * Compiled with -ffp-contract=off: the x86 reference has no FMA, and with IEEE fp64
//    add/mul/div/sqrt the results are then bit-identical to it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pxf.h"

namespace pxf {

enum Col : int { OPD = 0, X, Y, Z, L, M, N, UX, UY, UZ, AUX, NCOL };
constexpr unsigned B(int c) { return 1u << c; }
constexpr unsigned POS = B(X) | B(Y) | B(Z);
constexpr unsigned DIR = B(L) | B(M) | B(N);
constexpr unsigned NRM = B(UX) | B(UY) | B(UZ);

struct Cols {
    double *p[NCOL];
};

constexpr int kBlock = 256;
constexpr int kMaxBlocksPerCU = 8;
constexpr int kNewtonCap = 1000;  // the reference has none (it would hang); never reached on
                                  // inputs the reference terminates on

// Fortran single-precision literals widened to fp64 (woltsurf.f95:30,83,136; zernsurf.f95:31)
#define PXF_TOL_1EM8 ((double)1.e-8f)
#define PXF_TOL_1EM10 ((double)1.e-10f)

template <int V>
struct VecT;
template <>
struct VecT<1> {
    using type = double;
};
template <>
struct VecT<2> {
    using type = double2;
};

template <int V>
__device__ __forceinline__ void load_col(const double *p, int64_t v, double (&r)[V])
{
    if constexpr (V == 1) {
        r[0] = p[v];
    } else {
        const double2 t = *reinterpret_cast<const double2 *>(p + 2 * v);
        r[0] = t.x;
        r[1] = t.y;
    }
}

template <int V>
__device__ __forceinline__ void store_col(double *p, int64_t v, const double (&r)[V], const bool (&w)[V])
{
    if constexpr (V == 1) {
        if (w[0]) p[v] = r[0];
    } else {
        if (w[0] && w[1]) {
            *reinterpret_cast<double2 *>(p + 2 * v) = make_double2(r[0], r[1]);
        } else {
            if (w[0]) p[2 * v] = r[0];
            if (w[1]) p[2 * v + 1] = r[1];
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
template <class Op, int V, bool MASKED>
__global__ __launch_bounds__(kBlock) void ray_kernel(Cols c, int64_t first, int64_t nvec,
                                                     const uint8_t *__restrict__ mask, Op op)
{
    op.prologue();  // block-wide setup (LDS staging); no-op for most Ops
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < nvec; v += stride) {
        Ray r[V];
        bool sel[V];
        unsigned wm[V];
        const int64_t vv = first + v;  // vector index (units of V rays)
#pragma unroll
        for (int col = 0; col < NCOL; col++) {
            if (Op::RD & B(col)) {
                double t[V];
                load_col<V>(c.p[col], vv, t);
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
                store_col<V>(c.p[col], vv, t, w);
            }
        }
    }
}

// ---- host side ---------------------------------------------------------------------------
extern thread_local int g_last_hip_error;
int device_cus();

inline int hip_fail(hipError_t e)
{
    g_last_hip_error = (int)e;
    return PXF_EHIP;
}

inline int grid_for(int64_t nvec)
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
    bool aligned = true;
    for (int col = 0; col < NCOL; col++) {
        if ((Op::RD | Op::WR) & B(col)) {
            if (c.p[col] == nullptr) return PXF_EINVAL;
            if ((uintptr_t)c.p[col] & 15) aligned = false;
        }
    }
    hipStream_t s = (hipStream_t)stream;
    int64_t nvec2 = 0;
    if constexpr (Op::MAXV >= 2) {
        nvec2 = aligned ? num / 2 : 0;
        if (nvec2 == 0) {
        } else if (mask)
            hipLaunchKernelGGL((ray_kernel<Op, 2, true>), dim3(grid_for(nvec2)), dim3(kBlock), 0, s, c,
                               (int64_t)0, nvec2, mask, op);
        else
            hipLaunchKernelGGL((ray_kernel<Op, 2, false>), dim3(grid_for(nvec2)), dim3(kBlock), 0, s, c,
                               (int64_t)0, nvec2, mask, op);
    }
    const int64_t done = nvec2 * 2;
    if (done < num) {  // unaligned columns, or the odd last ray
        const int64_t rest = num - done;
        if (mask)
            hipLaunchKernelGGL((ray_kernel<Op, 1, true>), dim3(grid_for(rest)), dim3(kBlock), 0, s, c, done,
                               rest, mask, op);
        else
            hipLaunchKernelGGL((ray_kernel<Op, 1, false>), dim3(grid_for(rest)), dim3(kBlock), 0, s, c, done,
                               rest, mask, op);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PXF_OK : hip_fail(e);
}

}  // namespace pxf
