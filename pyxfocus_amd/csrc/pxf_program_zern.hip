// pxf_program_zern.hip -- fused ray program AROUND a Zernike trace (PXF_OP_TRACEZERN[OPD]).
//
// BASELINE config 4 is  source -> zernsurf -> reflect -> flat : as three launches it moves
// 96 + 72 + 96 B/ray; with the ray held in registers it moves 120.  The Zernike operation brings a
// coefficient table (LDS, staged once per workgroup; held in registers across the Newton loop up to
// degree 10), so programs that contain one run in this separate kernel -- two rays per thread, like
// the stand-alone Zernike kernel -- and the common programs keep their small register budget.
//
// This translation unit is compiled -ffp-contract=off like every bit-exact Op needs; the Zernike
// evaluation asks for FMA contraction in its own function bodies (pxf_ops_zernike.h).  The other
// operations of the program therefore produce exactly the bits of their stand-alone kernels.
#include "pxf_ops_zernike.h"
#include "pxf_program.h"

namespace pxf {

#ifndef PXF_ZPROG_WAVES
#define PXF_ZPROG_WAVES 2   // <= 256 registers per lane (no AGPRs: pxf_ops_zernike.h)
#endif
constexpr int kMaxOpsZ = PXF_PROGRAM_MAX_OPS_ZERN;
constexpr int kZernFusedMaxN = PXF_PROGRAM_ZERN_MAX_ORDER;

struct ProgramZ {  // a Program cut to the operation count a Zernike program may have (kernel-argument space)
    int nops;
    unsigned load, store;
    ProgOp op[kMaxOpsZ];
};

static_assert(sizeof(ProgramZ) + sizeof(ZernOp<kZernFusedMaxN, false>) + sizeof(Cols) + 64 <= 4096,
              "program + Zernike table must fit the kernel-argument segment");

// The operations a Zernike program may hold besides the trace itself: the frame changes, reflect, flat.
// (The full interpreter step of pxf_program.h inlines every Op of the library: a register and scalar
// footprint this kernel does not need.)
__host__ __device__ constexpr bool zprog_supported(int code)
{
    return code == PXF_OP_FLAT || code == PXF_OP_FLATOPD || code == PXF_OP_TRANSFORM || code == PXF_OP_ITRANSFORM ||
           code == PXF_OP_REFLECT || code == PXF_OP_TRACEZERN || code == PXF_OP_TRACEZERNOPD;
}

template <int V>
__device__ __forceinline__ void exec_op_lite(int code, int flags, const double *__restrict__ P, Ray (&r)[V],
                                             const bool (&sel)[V])
{
    switch (code) {
        case PXF_OP_FLAT: run_op(FlatOp<false>{0.}, r, sel); break;
        case PXF_OP_FLATOPD: run_op(FlatOp<true>{P[0]}, r, sel); break;
        case PXF_OP_TRANSFORM:
            run_op(TransformOp<false>{P[0], P[1], P[2], P[3], P[4], P[5], P[6], P[7], P[8], flags & 7}, r, sel);
            break;
        case PXF_OP_ITRANSFORM:
            run_op(TransformOp<true>{P[0], P[1], P[2], P[3], P[4], P[5], P[6], P[7], P[8], flags & 7}, r, sel);
            break;
        case PXF_OP_REFLECT: run_op(ReflectOp{}, r, sel); break;
        default: break;
    }
}

template <int NMAX, int V, bool NT>
__global__ __launch_bounds__(kBlock, PXF_ZPROG_WAVES) void program_zern_kernel(Cols c, int64_t first, int64_t nvec,
                                                                                      ProgramZ pg, ZernOp<NMAX, false> zop, int iz)
{
    zop.prologue();  // Zernike table -> LDS (contains the barrier: before any thread may exit)
    const int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x;   // no grid-stride loop (pxf_kernel.h: ray_kernel)
    if (v < nvec) {
        const int64_t vv = first + v;
        Ray r[V];
#pragma unroll
        for (int col = 0; col < 10; col++) {
            if (pg.load & B(col)) {  // wave-uniform
                double t[V];
                load_col<V, NT>(c.p[col], vv, t);
#pragma unroll
                for (int k = 0; k < V; k++) r[k].c[col] = t[k];
            }
        }
        bool sel[V];
#pragma unroll
        for (int k = 0; k < V; k++) sel[k] = true;
        for (int i = 0; i < pg.nops; i++) {
            const ProgOp &o = pg.op[i];
            if (i == iz) {
                unsigned wm[V];
                double opd_in[V];  // plain tracezern does not touch the opd column, which may be live in the
#pragma unroll             // program (ZernOp::apply adds t*nr to it)
                for (int k = 0; k < V; k++) {
                    opd_in[k] = r[k].c[OPD];
                    if (!zop.has_opd) r[k].c[OPD] = 0.;
                }
                zop.template apply<V>(r, sel, wm);
                if (!zop.has_opd) {
#pragma unroll
                    for (int k = 0; k < V; k++) r[k].c[OPD] = opd_in[k];
                }
            } else {
                exec_op_lite<V>(o.code, o.flags, o.p, r, sel);
            }
        }
#pragma unroll
        for (int col = 0; col < 10; col++) {
            if (pg.store & B(col)) {
                double t[V];
                bool w[V];
#pragma unroll
                for (int k = 0; k < V; k++) {
                    t[k] = r[k].c[col];
                    w[k] = true;
                }
                store_col<V, NT>(c.p[col], vv, t, w);
            }
        }
    }
}

template <int NMAX>
static int launch_zern_nmax(const ProgramZ &pz, int iz, const ZernTab &tab, const Cols &c, int64_t num,
                            pxf_stream_t stream)
{
    ZernOp<NMAX, false> zop{};
    zern_fill_table<NMAX>(zop.tab_arg, tab);
    zop.rad = pz.op[iz].p[0];
    zop.nr = pz.op[iz].p[1];
    zop.has_opd = pz.op[iz].code == PXF_OP_TRACEZERNOPD;
    hipStream_t s = (hipStream_t)stream;
    // no `ind=` masks in a Zernike program (the caller runs such a sequence as separate launches)
    for (int i = 0; i < pz.nops; i++)
        if (pz.op[i].mask != nullptr || !zprog_supported(pz.op[i].code)) return PXF_ERANGE;
    bool aligned = true;
    for (int col = 0; col < 10; col++)
        if (((pz.load | pz.store) & B(col)) && ((uintptr_t)c.p[col] & 15)) aligned = false;
    const bool nt = use_nt(num);
    const int64_t nvec2 = aligned ? num / 2 : 0;
    if (nvec2 > 0) {
        if (nt) hipLaunchKernelGGL((program_zern_kernel<NMAX, 2, true>), dim3(grid_exact(nvec2)), dim3(kBlock), 0, s, c, (int64_t)0, nvec2, pz, zop, iz);
        else hipLaunchKernelGGL((program_zern_kernel<NMAX, 2, false>), dim3(grid_exact(nvec2)), dim3(kBlock), 0, s, c, (int64_t)0, nvec2, pz, zop, iz);
    }
    const int64_t done = nvec2 * 2;
    if (done < num) {
        const int64_t rest = num - done;
        if (nt) hipLaunchKernelGGL((program_zern_kernel<NMAX, 1, true>), dim3(grid_exact(rest)), dim3(kBlock), 0, s, c, done, rest, pz, zop, iz);
        else hipLaunchKernelGGL((program_zern_kernel<NMAX, 1, false>), dim3(grid_exact(rest)), dim3(kBlock), 0, s, c, done, rest, pz, zop, iz);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PXF_OK : hip_fail(e);
}

int launch_program_zern(const Program &pg, int iz, const pxf_zern_args_t *za, const Cols &c, int64_t num,
                        pxf_stream_t stream)
{
    if (!za || iz < 0 || iz >= pg.nops || pg.nops > kMaxOpsZ) return PXF_EINVAL;
    ZernTab tab;
    int nmax = 0;
    int rc = fold_table(za->coeff, za->rorder, za->aorder, za->arrsize, tab, nmax);
    if (rc != PXF_OK) return rc;
    ProgramZ pz{};
    pz.nops = pg.nops;
    pz.load = pg.load;
    pz.store = pg.store;
    for (int i = 0; i < pg.nops; i++) pz.op[i] = pg.op[i];
    pz.op[iz].aux = nullptr;  // (a host pointer: never dereferenced on the device)
    // Instantiated for radial orders <= 4, <= 7 (36 terms), <= 10 and <= 15; a table of lower order runs
    // in the next instantiation above it with zero high coefficients.
    if (nmax <= 4) return launch_zern_nmax<4>(pz, iz, tab, c, num, stream);
    if (nmax <= 7) return launch_zern_nmax<7>(pz, iz, tab, c, num, stream);
    if (nmax <= 10) return launch_zern_nmax<10>(pz, iz, tab, c, num, stream);
    if (nmax <= kZernFusedMaxN) return launch_zern_nmax<kZernFusedMaxN>(pz, iz, tab, c, num, stream);
    return PXF_ERANGE;
}

}  // namespace pxf
