// pxf_program.hip -- fused ray program: a sequence of per-ray operations in ONE launch.
//
// Why: every stand-alone kernel is HBM-bound and already moves exactly its algorithmic bytes, so
// the only way to trace faster is to move fewer bytes.  A trace is a straight-line program over
// the ray's ten numbers; running it with the ray held in registers turns the two-bounce chain's
// 576 B/ray (six passes) into 120 B/ray (read x,y,z,l,m,n once; write nine columns once).
//
// How: the host turns the caller's pxf_op_t list into micro-operations with read / write /
// maybe-write column sets, removes the dead ones by a backward liveness pass, derives the load
// and store sets by a forward pass, and passes the plan BY VALUE in the kernel arguments.  The
// kernel is a wave-uniform interpreter: `switch (op.code)` is a scalar branch (no divergence),
// and each case is the SAME Op::apply the stand-alone kernels instantiate, so the two paths are
// bit-identical by construction.  Two rays per thread, 16-byte accesses, NT policy as elsewhere.
#include <stdlib.h>

#include "pxf_program.h"

namespace pxf {

// ANYMASK=false instantiations carry no per-op mask code (no saved copy of the ray -> fewer VGPRs)
// HOT: every operation of the program is one of kHotOps (unmasked programs only): the 9-case interpreter step
template <int V, bool NT, bool ANYMASK, bool HOT = false>
__global__ __launch_bounds__(kBlock, 2) void program_kernel(Cols c, int64_t first, int64_t nvec, Program pg)
{
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
        for (int i = 0; i < pg.nops; i++) {
            const ProgOp &o = pg.op[i];
            bool sel[V];
            Ray keep[ANYMASK ? V : 1];
            const bool masked = ANYMASK && o.mask != nullptr;
            if (masked) {
#pragma unroll
                for (int k = 0; k < V; k++) {
                    sel[k] = o.mask[vv * V + k] != 0;
                    keep[k] = r[k];
                }
            } else {
#pragma unroll
                for (int k = 0; k < V; k++) sel[k] = true;
            }
            exec_op<V, NT, HOT>(o.code, o.flags, o.p, o.aux, vv, r, sel);
            if (masked) {  // unselected rays keep every register
#pragma unroll
                for (int k = 0; k < V; k++)
                    if (!sel[k]) r[k] = keep[k];
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

// ---- straight-line programs ---------------------------------------------------------------------
// The interpreter pays for its generality: a `switch` per operation and, at every join, a block of phi copies of
// the whole ray state -- ~300 of the 1166 VALU wave-instructions of the two-bounce program (PMC), on a kernel
// that is VALU-bound.  The operation SEQUENCES of the BASELINE configs are known at compile time, so each gets a
// kernel in which the same Op::apply bodies follow one another with no dispatch at all (a fold over the code list);
// parameters, load / store sets and the transform groups the planner dropped still come from the planned program,
// so results are the interpreter's bit for bit.  pxf_program_run picks the chain kernel when an unmasked
// program's code sequence equals one of the lists below.
// `P` is the operation's nine doubles: an ordinary pointer (segmented programs: the workgroup's LDS copy) or a
// constant-address-space pointer into the kernel-argument segment (program_chain_kernel: scalar loads, issued where the
// operation starts).
template <int CODE, int V, bool NT, class PP>
__device__ __forceinline__ void exec_fixed(int flags, PP P, const double *__restrict__ aux,
                                           int64_t vv, Ray (&r)[V], const bool (&sel)[V])
{
    if constexpr (CODE == PXF_OP_FLAT) run_op(FlatOp<false>{0.}, r, sel);
    else if constexpr (CODE == PXF_OP_WOLTERPRIMARY) run_op(WolterPrimaryOp<false>{P[0], P[1], P[2], 0.}, r, sel);
    else if constexpr (CODE == PXF_OP_WOLTERSECONDARY) run_op(WolterSecondaryOp{P[0], P[1], P[2]}, r, sel);
    else if constexpr (CODE == PXF_OP_TRANSFORM)
        run_op(TransformOp<false>{P[0], P[1], P[2], P[3], P[4], P[5], P[6], P[7], P[8], flags & 7}, r, sel);
    else if constexpr (CODE == PXF_OP_ITRANSFORM)
        run_op(TransformOp<true>{P[0], P[1], P[2], P[3], P[4], P[5], P[6], P[7], P[8], flags & 7}, r, sel);
    else if constexpr (CODE == PXF_OP_REFLECT) run_op(ReflectOp{}, r, sel);
    else if constexpr (CODE == PXF_OP_RADGRAT) run_op(RadgratOp<false>{P[0], P[1], P[2], P[3], P[4]}, r, sel);
    else if constexpr (CODE == PXF_OP_SPOCONE) run_op(SpoConeOp{P[0], P[1], P[2], P[3]}, r, sel);
    else if constexpr (CODE == PXF_OP_INDANGLE) {  // an OUTPUT column: the angle of each ray as it stands here
        run_op(IndAngleOp{(int)P[0]}, r, sel);
        double t[V];
#pragma unroll
        for (int k = 0; k < V; k++) t[k] = r[k].c[AUX];
        store_col<V, NT>(const_cast<double *>(aux), vv, t, sel);
    } else static_assert(CODE == PXF_OP_FLAT, "operation not available in a straight-line program");
}

// the kernel-argument segment of program_chain_kernel(Cols, int64_t, int64_t, Program) as the ABI lays it out
struct ChainKernarg {
    Cols c;
    int64_t first, nvec;
    Program pg;
};
typedef const __attribute__((address_space(4))) char *kernarg_cptr;
typedef const __attribute__((address_space(4))) double *kernarg_dptr;
__device__ __forceinline__ kernarg_dptr chain_params(kernarg_cptr ka, int i, double &pin)
{
    kernarg_dptr P = (kernarg_dptr)(ka + offsetof(ChainKernarg, pg) + offsetof(Program, op) + (size_t)i * sizeof(ProgOp) +
                                    offsetof(ProgOp, p));
    asm volatile("" : "+s"(P), "+v"(pin));
    return P;
}

// One workgroup per 256 ray-vectors and NO grid-stride loop (the launch below never caps the grid): with a loop around
// the body every wave-uniform quantity of the program (load / store bits, transform-group bits, parameter addresses)
// is loop-invariant, gets hoisted in front of it and is then spilled for want of SGPRs.
template <int V, bool NT, int... CODES>
__global__ __launch_bounds__(kBlock, 2) void program_chain_kernel(Cols c, int64_t first, int64_t nvec, Program pg)
{
    const int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x;
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
        // Parameters: each operation reads its nine doubles from the kernel-argument segment WHEN IT STARTS.  Left to
        // itself the compiler hoists every scalar load of the program to the top of the kernel -- 12 operations x 18
        // SGPRs against a file of ~100 -- and then spills them to VGPR lanes: 240 v_writelane / v_readlane among the
        // ~1500 vector instructions of the config-3 program, on a kernel that is VALU-bound.  The empty asm ties each
        // operation's parameter pointer to the ray state the previous operation produced, so its loads cannot move up.
        const kernarg_cptr ka = (kernarg_cptr)__builtin_amdgcn_kernarg_segment_ptr();
        int i = 0;
        ((exec_fixed<CODES, V, NT>(pg.op[i].flags, chain_params(ka, i, r[0].c[X]), pg.op[i].aux, vv, r, sel), i++), ...);
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

template <int... CODES>
struct Chain {
    static constexpr int n = sizeof...(CODES);
    static bool matches(const Program &pg)
    {
        constexpr int codes[] = {CODES...};
        if (pg.nops != n) return false;
        for (int i = 0; i < n; i++)
            if (pg.op[i].code != codes[i] || pg.op[i].mask != nullptr) return false;
        return true;
    }
    template <int V, bool NT>
    static void launch(hipStream_t s, const Cols &c, int64_t first, int64_t nvec, const Program &pg)
    {
        hipLaunchKernelGGL((program_chain_kernel<V, NT, CODES...>), dim3(grid_exact(nvec)), dim3(kBlock), 0, s, c, first, nvec, pg);
    }
};
// BASELINE config 2: transform, wolterprimary, reflect, woltersecondary, reflect, flat
using ChainTwoBounce = Chain<PXF_OP_TRANSFORM, PXF_OP_WOLTERPRIMARY, PXF_OP_REFLECT, PXF_OP_WOLTERSECONDARY, PXF_OP_REFLECT,
                             PXF_OP_FLAT>;
// the grating stage of configs 3 / 5: transform, flat, reflect, radgrat, itransform, flat
using ChainGrating = Chain<PXF_OP_TRANSFORM, PXF_OP_FLAT, PXF_OP_REFLECT, PXF_OP_RADGRAT, PXF_OP_ITRANSFORM, PXF_OP_FLAT>;
// BASELINE config 3: the two in a row
using ChainTwoBounceGrating = Chain<PXF_OP_TRANSFORM, PXF_OP_WOLTERPRIMARY, PXF_OP_REFLECT, PXF_OP_WOLTERSECONDARY,
                                    PXF_OP_REFLECT, PXF_OP_FLAT, PXF_OP_TRANSFORM, PXF_OP_FLAT, PXF_OP_REFLECT,
                                    PXF_OP_RADGRAT, PXF_OP_ITRANSFORM, PXF_OP_FLAT>;

// the Arcus SPO pair per shell (segmented programs): rotate, spoCone, reflect, graze, spoCone, reflect, graze, move
using ChainSpoPair = Chain<PXF_OP_TRANSFORM, PXF_OP_SPOCONE, PXF_OP_REFLECT, PXF_OP_INDANGLE, PXF_OP_SPOCONE, PXF_OP_REFLECT,
                           PXF_OP_INDANGLE, PXF_OP_TRANSFORM>;

// ---- segmented program --------------------------------------------------------------------
// One launch for a bundle made of contiguous SEGMENTS of `rps` rays (mirror shells, modules ...)
// that run the same operation sequence with different parameters -- the reference does this as
// a Python loop over shells (examples/arcus/sector.py:556).  The per-segment parameter blocks live
// in a device table [nseg][nops][9]; each workgroup stages the blocks of the (at most
// kMaxSegPerBlock) segments its 256*V rays fall into in LDS, and every ray then reads its own
// segment's block (all lanes of a wave normally share one address: LDS broadcast).
constexpr int kMaxSegPerBlock = 4;

struct SegInfo {
    const double *table;  // device [nseg][nops][9]
    int64_t rps;          // rays per segment (even)
    int nseg;
};

// CODES...: a compile-time operation sequence (straight-line form, see above); empty = the interpreter
template <bool NT, bool ANYMASK, bool HOT = false, int... CODES>
__global__ __launch_bounds__(kBlock, 2) void program_seg_kernel(Cols c, int64_t nvec, Program pg, SegInfo sg)
{
    constexpr int V = 2;
    __shared__ double s_par[kMaxSegPerBlock][kMaxOps][9];
    const int64_t ntiles = (nvec + kBlock - 1) / kBlock;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t v0 = tile * kBlock;
        const int64_t vlast = (v0 + kBlock - 1 < nvec ? v0 + kBlock - 1 : nvec - 1);
        const int64_t seg0 = (v0 * V) / sg.rps, seg1 = (vlast * V + V - 1) / sg.rps;
        const int nsb = (int)(seg1 - seg0 + 1);
        const int per = pg.nops * 9;
        __syncthreads();  // previous tile's readers are done
        for (int i = threadIdx.x; i < nsb * per; i += kBlock) {
            const int sb = i / per, q = i - sb * per;
            s_par[sb][q / 9][q % 9] = sg.table[(seg0 + sb) * per + q];
        }
        __syncthreads();
        const int64_t vv = v0 + threadIdx.x;
        if (vv < nvec) {
            const int sb = (int)((vv * V) / sg.rps - seg0);
            Ray r[V];
#pragma unroll
            for (int col = 0; col < 10; col++) {
                if (pg.load & B(col)) {
                    double t[V];
                    load_col<V, NT>(c.p[col], vv, t);
#pragma unroll
                    for (int k = 0; k < V; k++) r[k].c[col] = t[k];
                }
            }
            if constexpr (sizeof...(CODES) > 0) {
                bool sel[V];
#pragma unroll
                for (int k = 0; k < V; k++) sel[k] = true;
                int i = 0;
                ((exec_fixed<CODES, V, NT>(pg.op[i].flags, &s_par[sb][i][0], pg.op[i].aux, vv, r, sel), i++), ...);
            } else
            for (int i = 0; i < pg.nops; i++) {
                const ProgOp &o = pg.op[i];
                bool sel[V];
                Ray keep[ANYMASK ? V : 1];
                const bool masked = ANYMASK && o.mask != nullptr;
                if (masked) {
#pragma unroll
                    for (int k = 0; k < V; k++) {
                        sel[k] = o.mask[vv * V + k] != 0;
                        keep[k] = r[k];
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < V; k++) sel[k] = true;
                }
                exec_op<V, NT, HOT>(o.code, o.flags, &s_par[sb][i][0], o.aux, vv, r, sel);
                if (masked) {
#pragma unroll
                    for (int k = 0; k < V; k++)
                        if (!sel[k]) r[k] = keep[k];
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
}

// ---- host: plan ------------------------------------------------------------------------------
struct Micro {
    unsigned rd, wa, wm;  // columns read, always written, maybe written
    int op;               // index into the op list
    int group;            // transform group bit, 0 otherwise
    bool live;
    bool side;            // writes outside the bundle (an output column): never dead
};

int plan_program(const pxf_op_t *ops, int32_t nops, Program &pg, int *dropped, int *zern_op, bool *has_mirror)
{
    int zern_src = -1;  // index (in the caller's list) of the Zernike operation, if any
    bool mirror = false;
    if (nops < 0 || nops > kMaxOps || (nops > 0 && !ops)) return PXF_EINVAL;
    Micro mi[kMaxOps * 3];
    int nm = 0;
    for (int i = 0; i < nops; i++) {
        const pxf_op_t &u = ops[i];
        if (u.reserved != 0) return PXF_EINVAL;
        ProgOp &o = pg.op[i];
        o = ProgOp{};
        o.code = u.code;
        o.mask = u.mask;
        o.aux = u.aux;
        auto add = [&](unsigned rd, unsigned wa, unsigned wmaybe, int group) {
            if (u.mask) {  // a masked op writes conditionally and must see the old values
                wmaybe |= wa;
                wa = 0;
            }
            mi[nm++] = Micro{rd, wa, wmaybe, i, group, true, false};
        };
        switch (u.code) {
            case PXF_OP_FLAT: add(POS | DIR, POS | NRM, 0, 0); break;
            case PXF_OP_FLATOPD:
                o.p[0] = u.a[0];
                add(POS | DIR | B(OPD), POS | NRM | B(OPD), 0, 0);
                break;
            case PXF_OP_CONIC:
                o.p[0] = u.a[0];
                o.p[1] = u.a[1];
                add(POS | DIR, 0, POS | DIR | NRM, 0);  // miss: lmn zeroed, xyz/u kept
                break;
            case PXF_OP_CONICOPD:
                o.p[0] = u.a[0];
                o.p[1] = u.a[1];
                o.p[2] = u.a[2];
                add(POS | DIR | B(OPD), 0, POS | DIR | NRM | B(OPD), 0);
                break;
            case PXF_OP_WOLTERPRIMARY:
            case PXF_OP_WOLTERPRIMARYOPD: {
                const bool wopd = u.code == PXF_OP_WOLTERPRIMARYOPD;
                const WolterPrimaryOp<true> w = make_wolter_primary<true>(u.a[0], u.a[1], u.a[2], wopd ? u.a[3] : 0.);
                o.p[0] = w.twop;
                o.p[1] = w.pp;
                o.p[2] = w.cst;
                o.p[3] = w.nr;
                add(POS | DIR | (wopd ? B(OPD) : 0u), POS | NRM | (wopd ? B(OPD) : 0u), 0, 0);
                break;
            }
            case PXF_OP_WOLTERSECONDARY: {
                const WolterSecondaryOp w = make_wolter_secondary(u.a[0], u.a[1], u.a[2]);
                o.p[0] = w.e2;
                o.p[1] = w.twoe2;
                o.p[2] = w.d;
                add(POS | DIR, POS | NRM, 0, 0);
                break;
            }
            case PXF_OP_TRANSFORM:
            case PXF_OP_ITRANSFORM: {
                double cs[6];
                trig6(u.a[3], u.a[4], u.a[5], u.code == PXF_OP_ITRANSFORM, cs);
                o.p[0] = u.a[0];
                o.p[1] = u.a[1];
                o.p[2] = u.a[2];
                for (int k = 0; k < 6; k++) o.p[3 + k] = cs[k];
                add(POS, POS, 0, 1);  // three independent micro-ops
                add(DIR, DIR, 0, 2);
                add(NRM, NRM, 0, 4);
                break;
            }
            case PXF_OP_REFLECT: add(DIR | NRM, DIR, 0, 0); break;
            case PXF_OP_RADGRAT:
                o.p[0] = u.a[1];  // dpermm
                o.p[1] = u.a[2];  // order
                o.p[2] = u.a[0];  // wave
                radgrat_consts(o.p[3], o.p[4]);
                add(B(X) | B(Y) | DIR, DIR, 0, 0);
                break;
            case PXF_OP_RADGRATW:
                if (!u.aux) return PXF_EINVAL;
                o.p[0] = u.a[0];  // dpermm
                o.p[1] = u.a[1];  // order
                radgrat_consts(o.p[3], o.p[4]);
                add(B(X) | B(Y) | DIR, DIR, 0, 0);
                break;
            case PXF_OP_SPOCONE:
                o.p[0] = u.a[0];
                o.p[1] = tan(u.a[1]);
                host_sincos(u.a[1], o.p[2], o.p[3]);   // the reference binary: one sincos (pxf_ops_transforms.h)
                add(POS | DIR, 0, POS | DIR | NRM, 0);  // miss: lmn zeroed, xyz/u kept
                break;
            case PXF_OP_SPHERE:
            case PXF_OP_CYL:
                o.p[0] = u.a[0];
                add(POS | DIR, POS | NRM, DIR, 0);  // miss zeroes lmn too
                break;
            case PXF_OP_SPHEREOPD:
            case PXF_OP_CYLOPD:
                o.p[0] = u.a[0];
                o.p[1] = u.a[1];
                add(POS | DIR | B(OPD), POS | NRM, DIR | B(OPD), 0);
                break;
            case PXF_OP_CYLCONIC:
                o.p[0] = u.a[0];
                o.p[1] = (1 + u.a[1]) * (u.a[0] * u.a[0]);
                add(POS | DIR, POS | NRM, 0, 0);
                break;
            case PXF_OP_PARAXIAL:
                o.p[0] = u.a[0];
                add(B(X) | B(Y) | B(L) | B(M), B(L) | B(M), 0, 0);
                break;
            case PXF_OP_PARAXIALY:
                o.p[0] = u.a[0];
                add(B(Y) | B(M), B(M), 0, 0);
                break;
            case PXF_OP_GRAT:
                o.p[0] = u.a[0];
                o.p[1] = u.a[1];
                add(DIR, DIR, 0, 0);
                break;
            case PXF_OP_REFRACT:
                o.p[0] = u.a[0];
                o.p[1] = u.a[1];
                add(DIR | NRM, DIR | NRM, 0, 0);
                break;
            case PXF_OP_INDANGLE:
                if (!u.aux) return PXF_EINVAL;
                o.p[0] = u.a[0] != 0. ? 1. : 0.;
                add(DIR | NRM, 0, 0, 0);
                mi[nm - 1].side = true;
                break;
            case PXF_OP_TRACEZERN:
            case PXF_OP_TRACEZERNOPD: {
                const bool wopd = u.code == PXF_OP_TRACEZERNOPD;
                if (!u.aux || zern_src >= 0) return PXF_EINVAL;  // needs its host table; one per program
                zern_src = i;
                o.p[0] = u.a[0];              // rad
                o.p[1] = wopd ? u.a[1] : 0.;  // nr
                add(POS | DIR | (wopd ? B(OPD) : 0u), POS | NRM | (wopd ? B(OPD) : 0u), 0, 0);
                break;
            }
            case PXF_OP_WOLTERPRIMLL:
            case PXF_OP_WOLTERSECLL:
            case PXF_OP_ELLIPSOIDWOLTLL:
                if (!u.aux) return PXF_EINVAL;  // needs its host coefficient list
                // fallthrough
            case PXF_OP_WOLTERSINE:
                mirror = true;
                for (int k = 0; k < 6; k++) o.p[k] = u.a[k];   // derived constants: launch_program_mirror
                add(POS | DIR, POS | NRM, 0, 0);
                break;
            case PXF_OP_WSPRIMARY:
            case PXF_OP_WSSECONDARY:
            case PXF_OP_WSPRIMARYBACK:
            case PXF_OP_WSSECONDARYBACK:
                mirror = true;
                for (int k = 0; k < 6; k++) o.p[k] = u.a[k];
                add(POS | DIR, POS, NRM, 0);   // 26 steps without convergence: position restored, normal kept
                break;
            default: return PXF_EINVAL;
        }
    }
    if (mirror && (!has_mirror || zern_src >= 0)) return PXF_ERANGE;
    if (has_mirror) *has_mirror = mirror;
    // backward liveness: every column is observable when the program ends
    unsigned live = 0x3ffu;
    int dead = 0;
    for (int j = nm - 1; j >= 0; j--) {
        Micro &m = mi[j];
        if (((m.wa | m.wm) & live) == 0 && !m.side) {
            m.live = false;
            dead++;
            continue;
        }
        live = (live & ~m.wa) | m.rd;
    }
    // forward: loads = read (or maybe-written) before definitely defined; stores = anything written
    unsigned defined = 0, load = 0, store = 0;
    for (int i = 0; i < nops; i++) pg.op[i].flags = 0;
    bool op_live[kMaxOps] = {false};
    for (int j = 0; j < nm; j++) {
        const Micro &m = mi[j];
        if (!m.live) continue;
        load |= (m.rd | m.wm) & ~defined;
        defined |= m.wa;
        store |= m.wa | m.wm;
        op_live[m.op] = true;
        pg.op[m.op].flags |= m.group;
    }
    // compact away fully dead ops
    int k = 0, zk = -1;
    for (int i = 0; i < nops; i++)
        if (op_live[i]) {
            if (i == zern_src) zk = k;
            pg.op[k++] = pg.op[i];
        }
    pg.nops = k;
    for (int i = 0; i < k; i++) pg.op[i].flags |= prog_onehot(pg.op[i].code);
    if (zern_op) *zern_op = zk;
    if (zk >= 0 && k > PXF_PROGRAM_MAX_OPS_ZERN) return PXF_ERANGE;
    if (mirror && k > PXF_PROGRAM_MAX_OPS_MIRROR) return PXF_ERANGE;
    pg.load = load;
    pg.store = store;
    if (dropped) *dropped = dead;
    return PXF_OK;
}

}  // namespace pxf

using namespace pxf;

extern "C" {

int pxf_program_plan(const pxf_op_t *ops, int32_t nops, uint32_t *load_mask, uint32_t *store_mask,
                     int32_t *dropped)
{
    Program pg{};
    int dead = 0;
    bool mirror = false;
    int rc = plan_program(ops, nops, pg, &dead, nullptr, &mirror);
    if (rc != PXF_OK) return rc;
    if (load_mask) *load_mask = pg.load;
    if (store_mask) *store_mask = pg.store;
    if (dropped) *dropped = dead;
    return PXF_OK;
}

int pxf_program_run(const pxf_op_t *ops, int32_t nops, double *const *cols, int64_t num, pxf_stream_t stream)
{
    Program pg{};
    int iz = -1;
    bool mirror = false;
    int rc = plan_program(ops, nops, pg, nullptr, &iz, &mirror);
    if (rc != PXF_OK) return rc;
    if (num < 0 || !cols) return PXF_EINVAL;
    if (num == 0 || pg.nops == 0) return PXF_OK;
    if (mirror) {  // a program that holds W-S / Legendre-Legendre / woltersine mirrors: its own kernel
        Cols cm{};
        for (int col = 0; col < 10; col++) {
            cm.p[col] = cols[col];
            if (((pg.load | pg.store) & B(col)) && !cols[col]) return PXF_EINVAL;
        }
        return launch_program_mirror(pg, cm, num, stream);
    }
    if (iz >= 0) {  // a program around a Zernike trace: its own kernel (one ray per thread, LDS table)
        Cols cz{};
        for (int col = 0; col < 10; col++) {
            cz.p[col] = cols[col];
            if (((pg.load | pg.store) & B(col)) && !cols[col]) return PXF_EINVAL;
        }
        return launch_program_zern(pg, iz, reinterpret_cast<const pxf_zern_args_t *>(pg.op[iz].aux), cz, num, stream);
    }
    Cols c{};
    bool aligned = true;
    for (int col = 0; col < 10; col++) {
        c.p[col] = cols[col];
        if ((pg.load | pg.store) & B(col)) {
            if (!cols[col]) return PXF_EINVAL;
            if ((uintptr_t)cols[col] & 15) aligned = false;
        }
    }
    for (int i = 0; i < pg.nops; i++)
        if (pg.op[i].aux && ((uintptr_t)pg.op[i].aux & 15)) aligned = false;
    hipStream_t s = (hipStream_t)stream;
    const bool nt = use_nt(num);
    bool anymask = false;
    for (int i = 0; i < pg.nops; i++) anymask |= pg.op[i].mask != nullptr;
    static int force_v = -1;
    if (force_v < 0) {
        const char *e = getenv("PXF_PROGRAM_V");
        force_v = e ? atoi(e) : 0;
    }
    const int64_t nvec2 = (aligned && force_v != 1) ? num / 2 : 0;
    bool hot = !anymask;
    for (int i = 0; i < pg.nops; i++) hot &= (pg.op[i].flags & kHotOps) != 0;
#define PXF_PLAUNCH(V, T, A, FIRST, NVEC)                                                                             \
    do {                                                                                                              \
        if (hot && !(A))                                                                                              \
            hipLaunchKernelGGL((program_kernel<V, T, false, true>), dim3(grid_exact(NVEC)), dim3(kBlock), 0, s, c, FIRST, NVEC, pg); \
        else                                                                                                          \
            hipLaunchKernelGGL((program_kernel<V, T, A>), dim3(grid_exact(NVEC)), dim3(kBlock), 0, s, c, FIRST, NVEC, pg);  \
    } while (0)
#define PXF_PLAUNCH_V(V, FIRST, NVEC)                      \
    do {                                                   \
        if (anymask) {                                     \
            if (nt) PXF_PLAUNCH(V, true, true, FIRST, NVEC);    \
            else PXF_PLAUNCH(V, false, true, FIRST, NVEC);      \
        } else {                                           \
            if (nt) PXF_PLAUNCH(V, true, false, FIRST, NVEC);   \
            else PXF_PLAUNCH(V, false, false, FIRST, NVEC);     \
        }                                                  \
    } while (0)
    static const bool no_chain = getenv("PXF_PROGRAM_NO_CHAIN") != nullptr;   // dev knob (A/B against the interpreter)
    bool chained = false;
#define PXF_TRY_CHAIN(CH)                                                                  \
    if (!chained && !no_chain && CH::matches(pg)) {                                        \
        chained = true;                                                                    \
        if (nvec2 > 0) {                                                                   \
            if (nt) CH::launch<2, true>(s, c, (int64_t)0, nvec2, pg);                      \
            else CH::launch<2, false>(s, c, (int64_t)0, nvec2, pg);                        \
        }                                                                                  \
        if (nvec2 * 2 < num) {                                                             \
            if (nt) CH::launch<1, true>(s, c, nvec2 * 2, num - nvec2 * 2, pg);             \
            else CH::launch<1, false>(s, c, nvec2 * 2, num - nvec2 * 2, pg);               \
        }                                                                                  \
    }
    PXF_TRY_CHAIN(ChainTwoBounce)
    PXF_TRY_CHAIN(ChainGrating)
    PXF_TRY_CHAIN(ChainTwoBounceGrating)
#undef PXF_TRY_CHAIN
    if (!chained) {
        if (nvec2 > 0) PXF_PLAUNCH_V(2, (int64_t)0, nvec2);
        const int64_t done = nvec2 * 2;
        if (done < num) PXF_PLAUNCH_V(1, done, num - done);
    }
#undef PXF_PLAUNCH_V
#undef PXF_PLAUNCH
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PXF_OK : hip_fail(e);
}

int pxf_program_params(const pxf_op_t *ops, int32_t nops, double *params_out, int32_t *nlive)
{
    Program pg{};
    int iz = -1;
    int rc = plan_program(ops, nops, pg, nullptr, &iz);
    if (rc != PXF_OK) return rc;
    if (iz >= 0) return PXF_EINVAL;  // a Zernike table is not a per-segment parameter block
    if (!params_out || !nlive) return PXF_EINVAL;
    for (int i = 0; i < pg.nops; i++)
        for (int k = 0; k < 9; k++) params_out[i * 9 + k] = pg.op[i].p[k];
    *nlive = pg.nops;
    return PXF_OK;
}

int pxf_program_run_segments(const pxf_op_t *ops, int32_t nops, const double *seg_params_dev, int32_t nseg,
                             int64_t rays_per_seg, double *const *cols, int64_t num, pxf_stream_t stream)
{
    Program pg{};
    int iz = -1;
    int rc = plan_program(ops, nops, pg, nullptr, &iz);
    if (rc != PXF_OK) return rc;
    if (iz >= 0) return PXF_EINVAL;
    if (num < 0 || !cols || nseg < 1 || !seg_params_dev) return PXF_EINVAL;
    // every workgroup (512 rays) must span at most kMaxSegPerBlock segments; ray pairs stay in one segment
    if (rays_per_seg < 2 || (rays_per_seg & 1) || rays_per_seg * (kMaxSegPerBlock - 1) < 2 * kBlock) return PXF_ERANGE;
    if (num != (int64_t)nseg * rays_per_seg) return PXF_EINVAL;
    if (num == 0 || pg.nops == 0) return PXF_OK;
    Cols c{};
    for (int col = 0; col < 10; col++) {
        c.p[col] = cols[col];
        if ((pg.load | pg.store) & B(col)) {
            if (!cols[col]) return PXF_EINVAL;
            if ((uintptr_t)cols[col] & 15) return PXF_EINVAL;  // segmented bundles are whole, aligned bundles
        }
    }
    bool anymask = false;
    for (int i = 0; i < pg.nops; i++) {
        anymask |= pg.op[i].mask != nullptr;
        if (pg.op[i].aux && ((uintptr_t)pg.op[i].aux & 15)) return PXF_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    const bool nt = use_nt(num);
    const int64_t nvec = num / 2;
    SegInfo sg{seg_params_dev, rays_per_seg, nseg};
    const dim3 grid(grid_for(nvec)), block(kBlock);
    if (anymask) {
        if (nt) hipLaunchKernelGGL((program_seg_kernel<true, true>), grid, block, 0, s, c, nvec, pg, sg);
        else hipLaunchKernelGGL((program_seg_kernel<false, true>), grid, block, 0, s, c, nvec, pg, sg);
    } else if (ChainSpoPair::matches(pg) && getenv("PXF_PROGRAM_NO_CHAIN") == nullptr) {
        // the Arcus SPO stage (sector.py:574-612): rotate, spoPrimary, reflect, graze, spoSecondary, reflect, graze, move
        if (nt)
            hipLaunchKernelGGL((program_seg_kernel<true, false, true, PXF_OP_TRANSFORM, PXF_OP_SPOCONE, PXF_OP_REFLECT,
                                                   PXF_OP_INDANGLE, PXF_OP_SPOCONE, PXF_OP_REFLECT, PXF_OP_INDANGLE,
                                                   PXF_OP_TRANSFORM>), grid, block, 0, s, c, nvec, pg, sg);
        else
            hipLaunchKernelGGL((program_seg_kernel<false, false, true, PXF_OP_TRANSFORM, PXF_OP_SPOCONE, PXF_OP_REFLECT,
                                                   PXF_OP_INDANGLE, PXF_OP_SPOCONE, PXF_OP_REFLECT, PXF_OP_INDANGLE,
                                                   PXF_OP_TRANSFORM>), grid, block, 0, s, c, nvec, pg, sg);
    } else {
        bool hot = true;
        for (int i = 0; i < pg.nops; i++) hot &= (pg.op[i].flags & kHotOps) != 0;
        if (hot) {
            if (nt) hipLaunchKernelGGL((program_seg_kernel<true, false, true>), grid, block, 0, s, c, nvec, pg, sg);
            else hipLaunchKernelGGL((program_seg_kernel<false, false, true>), grid, block, 0, s, c, nvec, pg, sg);
        } else {
            if (nt) hipLaunchKernelGGL((program_seg_kernel<true, false>), grid, block, 0, s, c, nvec, pg, sg);
            else hipLaunchKernelGGL((program_seg_kernel<false, false>), grid, block, 0, s, c, nvec, pg, sg);
        }
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PXF_OK : hip_fail(e);
}

}  // extern "C"
