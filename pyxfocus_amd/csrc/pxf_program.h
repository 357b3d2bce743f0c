// pxf_program.h -- what the fused-program translation units share: the planned program passed by
// value to the kernels, the wave-uniform interpreter step `exec_op` (each case is the SAME Op::apply
// the stand-alone kernels instantiate), and the host planner.  See pxf_program.hip.
#pragma once
#include "pxf_ops_surfaces.h"
#include "pxf_ops_transforms.h"

namespace pxf {

constexpr int kMaxOps = PXF_PROGRAM_MAX_OPS;

struct ProgOp {
    int code;
    int flags;  // bits 0-2: transform: vector groups to move; bits 3..: ONE-HOT image of `code` (1 << (2 + code)),
                // filled by the planner for the interpreter that tests one bit per case (pxf_program_fast.hip)
    const uint8_t *mask;
    const double *aux;
    double p[9];
};

struct Program {
    int nops;
    unsigned load, store;
    ProgOp op[kMaxOps];
};
static_assert(sizeof(Program) + sizeof(Cols) + 64 <= 4096, "program must fit the kernel-argument segment");

// One-hot image of an op code (codes 1..28) above the three transform-group bits of ProgOp::flags.
__host__ __device__ constexpr int prog_onehot(int code) { return code <= PXF_OP_INDANGLE ? 1 << (2 + code) : 0; }
static_assert(PXF_OP_INDANGLE + 2 < 31, "op codes must fit the one-hot field of ProgOp::flags");

template <class OP, int V>
__device__ __forceinline__ void run_op(const OP &o, Ray (&r)[V], const bool (&sel)[V])
{
    unsigned wm[V];
    o.template apply<V>(r, sel, wm);
}

// One operation on the V rays of a thread.  `P` points at the op's nine doubles: the kernel
// arguments for an ordinary program (wave-uniform -> scalar loads), or the workgroup's LDS copy of
// the ray's SEGMENT parameters for a segmented program.
// The operations of the common chains (the BASELINE configs: frame changes, Wolter-I pair, reflect, flat, radial
// grating, SPO cone, indAngle).  A program made only of these runs in a kernel instantiation whose switch has
// 9 cases instead of 25: the compiler lowers a switch to a compare tree with a block of whole-ray-state phi
// copies per level, so fewer cases = fewer levels = fewer copies.
constexpr int kHotOps = prog_onehot(PXF_OP_TRANSFORM) | prog_onehot(PXF_OP_ITRANSFORM) | prog_onehot(PXF_OP_WOLTERPRIMARY) |
                        prog_onehot(PXF_OP_WOLTERSECONDARY) | prog_onehot(PXF_OP_REFLECT) | prog_onehot(PXF_OP_FLAT) |
                        prog_onehot(PXF_OP_RADGRAT) | prog_onehot(PXF_OP_SPOCONE) | prog_onehot(PXF_OP_INDANGLE);

template <int V, bool NT, bool HOT = false>
__device__ __forceinline__ void exec_op(int code, int flags, const double *__restrict__ P,
                                        const double *__restrict__ aux, int64_t vv, Ray (&r)[V],
                                        const bool (&sel)[V])
{
    if constexpr (HOT) {
        switch (code) {
            case PXF_OP_FLAT: run_op(FlatOp<false>{0.}, r, sel); break;
            case PXF_OP_WOLTERPRIMARY: run_op(WolterPrimaryOp<false>{P[0], P[1], P[2], 0.}, r, sel); break;
            case PXF_OP_WOLTERSECONDARY: run_op(WolterSecondaryOp{P[0], P[1], P[2]}, r, sel); break;
            case PXF_OP_TRANSFORM:
                run_op(TransformOp<false>{P[0], P[1], P[2], P[3], P[4], P[5], P[6], P[7], P[8], flags & 7}, r, sel);
                break;
            case PXF_OP_ITRANSFORM:
                run_op(TransformOp<true>{P[0], P[1], P[2], P[3], P[4], P[5], P[6], P[7], P[8], flags & 7}, r, sel);
                break;
            case PXF_OP_REFLECT: run_op(ReflectOp{}, r, sel); break;
            case PXF_OP_RADGRAT: run_op(RadgratOp<false>{P[0], P[1], P[2], P[3], P[4]}, r, sel); break;
            case PXF_OP_SPOCONE: run_op(SpoConeOp{P[0], P[1], P[2], P[3]}, r, sel); break;
            case PXF_OP_INDANGLE: {
                run_op(IndAngleOp{(int)P[0]}, r, sel);
                double t[V];
#pragma unroll
                for (int k = 0; k < V; k++) t[k] = r[k].c[AUX];
                store_col<V, NT>(const_cast<double *>(aux), vv, t, sel);
                break;
            }
            default: break;
        }
        return;
    }
    switch (code) {
        case PXF_OP_FLAT: run_op(FlatOp<false>{0.}, r, sel); break;
        case PXF_OP_FLATOPD: run_op(FlatOp<true>{P[0]}, r, sel); break;
        case PXF_OP_CONIC: run_op(ConicOp<false>{P[0], P[1], 0.}, r, sel); break;
        case PXF_OP_CONICOPD: run_op(ConicOp<true>{P[0], P[1], P[2]}, r, sel); break;
        case PXF_OP_WOLTERPRIMARY: run_op(WolterPrimaryOp<false>{P[0], P[1], P[2], 0.}, r, sel); break;
        case PXF_OP_WOLTERPRIMARYOPD:
            run_op(WolterPrimaryOp<true>{P[0], P[1], P[2], P[3]}, r, sel);
            break;
        case PXF_OP_WOLTERSECONDARY: run_op(WolterSecondaryOp{P[0], P[1], P[2]}, r, sel); break;
        case PXF_OP_TRANSFORM:
            run_op(TransformOp<false>{P[0], P[1], P[2], P[3], P[4], P[5], P[6], P[7], P[8], flags & 7}, r, sel);
            break;
        case PXF_OP_ITRANSFORM:
            run_op(TransformOp<true>{P[0], P[1], P[2], P[3], P[4], P[5], P[6], P[7], P[8], flags & 7}, r, sel);
            break;
        case PXF_OP_REFLECT: run_op(ReflectOp{}, r, sel); break;
        case PXF_OP_RADGRAT: run_op(RadgratOp<false>{P[0], P[1], P[2], P[3], P[4]}, r, sel); break;
        case PXF_OP_RADGRATW: {
            double t[V];
            load_col<V, NT>(aux, vv, t);
#pragma unroll
            for (int k = 0; k < V; k++) r[k].c[AUX] = t[k];
            run_op(RadgratOp<true>{P[0], P[1], 0., P[3], P[4]}, r, sel);
            break;
        }
        case PXF_OP_SPOCONE: run_op(SpoConeOp{P[0], P[1], P[2], P[3]}, r, sel); break;
        case PXF_OP_SPHERE: run_op(SphereCylOp<false, false>{P[0], 0.}, r, sel); break;
        case PXF_OP_SPHEREOPD: run_op(SphereCylOp<true, false>{P[0], P[1]}, r, sel); break;
        case PXF_OP_CYL: run_op(SphereCylOp<false, true>{P[0], 0.}, r, sel); break;
        case PXF_OP_CYLOPD: run_op(SphereCylOp<true, true>{P[0], P[1]}, r, sel); break;
        case PXF_OP_CYLCONIC: run_op(CylConicOp{P[0], P[1]}, r, sel); break;
        case PXF_OP_PARAXIAL: run_op(ParaxialOp<false>{P[0]}, r, sel); break;
        case PXF_OP_PARAXIALY: run_op(ParaxialOp<true>{P[0]}, r, sel); break;
        case PXF_OP_GRAT:
            if (aux) {
                double t[V];
                load_col<V, NT>(aux, vv, t);
#pragma unroll
                for (int k = 0; k < V; k++) r[k].c[AUX] = t[k];
                run_op(GratOp<true>{P[0], 0.}, r, sel);
            } else {
                run_op(GratOp<false>{P[0], P[1]}, r, sel);
            }
            break;
        case PXF_OP_REFRACT: run_op(RefractOp{P[0], P[1]}, r, sel); break;
        case PXF_OP_INDANGLE: {  // an OUTPUT column: the angle of each ray as it stands here
            run_op(IndAngleOp{(int)P[0]}, r, sel);
            double t[V];
#pragma unroll
            for (int k = 0; k < V; k++) t[k] = r[k].c[AUX];
            store_col<V, NT>(const_cast<double *>(aux), vv, t, sel);
            break;
        }
        default: break;
    }
}

// host planner (pxf_program.hip): micro-ops, liveness, load / store sets; `zern_op` receives the index
// (after dead-op compaction) of the program's Zernike operation, or -1
// `has_mirror`: the program holds a tolerance-parity mirror (PXF_OP_WSPRIMARY .. PXF_OP_ELLIPSOIDWOLTLL); plan_program
// returns PXF_ERANGE for such a program if the caller did not ask (segmented / tolerance-mode programs cannot hold one)
int plan_program(const pxf_op_t *ops, int32_t nops, Program &pg, int *dropped, int *zern_op = nullptr,
                 bool *has_mirror = nullptr);

// pxf_program_zern.hip: launch a planned program whose operation `iz` is a Zernike trace
// (PXF_OP_TRACEZERN[OPD]); `za` is the host-side table of that operation
int launch_program_zern(const Program &pg, int iz, const pxf_zern_args_t *za, const Cols &c, int64_t num,
                        pxf_stream_t stream);

// pxf_program_mirror.hip: launch a planned program that holds mirror operations (their p[0..5] = the caller's
// scalars, aux = the host pxf_ll_args_t of a Legendre-Legendre kind)
int launch_program_mirror(const Program &pg, const Cols &c, int64_t num, pxf_stream_t stream);

}  // namespace pxf
