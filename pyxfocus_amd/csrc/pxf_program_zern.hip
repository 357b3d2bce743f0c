// pxf_program_zern.hip -- fused ray program AROUND a Zernike trace (PXF_OP_TRACEZERN[OPD]).
//
// BASELINE config 4 is  source -> zernsurf -> reflect -> flat : as three launches it moves
// 96 + 72 + 96 B/ray; with the ray held in registers it is bounded by the Zernike kernel's own fp64
// work.  The Zernike operation brings a coefficient table (LDS, staged once per workgroup) and a
// large register footprint, so programs that contain one run in this separate kernel -- one ray per
// thread, like the stand-alone Zernike kernel -- and the common programs keep their small one.
//
// This translation unit is compiled -ffp-contract=off like every bit-exact Op needs; the Zernike
// evaluation asks for FMA contraction in its own function bodies (pxf_ops_zernike.h).  The other
// operations of the program therefore produce exactly the bits of their stand-alone kernels.
#include "pxf_ops_zernike.h"
#include "pxf_program.h"

namespace pxf {

#ifndef PXF_ZPROG_WAVES
#define PXF_ZPROG_WAVES 3
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

// (three waves per SIMD like the stand-alone order-7 kernel: 168 registers, eight of them spilled.  With
// the library's full interpreter step in the kernel the same request cost 340 bytes of spills and two
// waves were faster; with the lean step it is 5.4 ms for config 4 against 6.8 ms at two waves.)
// The operations a Zernike program may hold besides the trace itself: the frame changes, reflect, flat.
// The full interpreter step (pxf_program.h exec_op) inlines every Op of the library; its register and
// scalar footprint on top of the Zernike evaluation pushed this kernel to two waves per SIMD with spills.
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
            run_op(TransformOp<false>{P[0], P[1], P[2], P[3], P[4], P[5], P[6], P[7], P[8], flags}, r, sel);
            break;
        case PXF_OP_ITRANSFORM:
            run_op(TransformOp<true>{P[0], P[1], P[2], P[3], P[4], P[5], P[6], P[7], P[8], flags}, r, sel);
            break;
        case PXF_OP_REFLECT: run_op(ReflectOp{}, r, sel); break;
        default: break;
    }
}

template <int NMAX, bool NT, bool ANYMASK>
__global__ __launch_bounds__(kBlock, PXF_ZPROG_WAVES) void program_zern_kernel(Cols c, int64_t num, ProgramZ pg,
                                                                                      ZernOp<NMAX, false> zop, int iz)
{
    constexpr int V = 1;
    zop.prologue();  // Zernike table -> LDS (contains the barrier: before any thread may exit)
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t vv = (int64_t)blockIdx.x * kBlock + threadIdx.x; vv < num; vv += stride) {
        Ray r[V];
#pragma unroll
        for (int col = 0; col < 10; col++) {
            if (pg.load & B(col)) {  // wave-uniform
                double t[V];
                load_col<V, NT>(c.p[col], vv, t);
                r[0].c[col] = t[0];
            }
        }
        for (int i = 0; i < pg.nops; i++) {
            const ProgOp &o = pg.op[i];
            bool sel[V] = {true};
            Ray keep[ANYMASK ? V : 1];
            const bool masked = ANYMASK && o.mask != nullptr;
            if (masked) {
                sel[0] = o.mask[vv] != 0;
                keep[0] = r[0];
            }
            if (i == iz) {
                unsigned wm[V];
                const double opd_in = r[0].c[OPD];  // plain tracezern does not touch the opd column, which
                if (!zop.has_opd) r[0].c[OPD] = 0.;   // may be live in the program (ZernOp::apply adds t*nr)
                zop.template apply<V>(r, sel, wm);
                if (!zop.has_opd) r[0].c[OPD] = opd_in;
            } else {
                exec_op_lite<V>(o.code, o.flags, o.p, r, sel);
            }
            if (masked && !sel[0]) r[0] = keep[0];  // unselected rays keep every register
        }
#pragma unroll
        for (int col = 0; col < 10; col++) {
            if (pg.store & B(col)) {
                double t[V] = {r[0].c[col]};
                bool w[V] = {true};
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
    const int g = grid_for(num);
    // no `ind=` masks in a Zernike program: the saved copy of the ray that a masked operation needs would
    // push this kernel deep into register spills (46 vector + 90 scalar registers) -- the caller runs
    // such a sequence as separate launches
    for (int i = 0; i < pz.nops; i++)
        if (pz.op[i].mask != nullptr || !zprog_supported(pz.op[i].code)) return PXF_ERANGE;
    if (use_nt(num))
        hipLaunchKernelGGL((program_zern_kernel<NMAX, true, false>), dim3(g), dim3(kBlock), 0, s, c, num, pz, zop, iz);
    else
        hipLaunchKernelGGL((program_zern_kernel<NMAX, false, false>), dim3(g), dim3(kBlock), 0, s, c, num, pz, zop, iz);
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
    // Instantiated for radial orders <= 4 and <= 7 (36 terms).  Higher orders are not offered as fused
    // programs: with the interpreter's operations in the same kernel they exhaust both register files
    // (SGPRs spilled while no VGPR is free), the regime in which stand-alone order-11..15 kernels were
    // seen to return wrong results before their constants moved to LDS (pxf_ops_zernike.h).  The caller
    // runs such a trace as its own launch (PXF_ERANGE).
    if (nmax <= 4) return launch_zern_nmax<4>(pz, iz, tab, c, num, stream);
    if (nmax <= kZernFusedMaxN) return launch_zern_nmax<kZernFusedMaxN>(pz, iz, tab, c, num, stream);
    return PXF_ERANGE;
}

}  // namespace pxf
