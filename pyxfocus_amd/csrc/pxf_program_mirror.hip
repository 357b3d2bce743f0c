// pxf_program_mirror.hip -- fused ray programs that hold tolerance-parity MIRRORS: Wolter-Schwarzschild
// primary / secondary (+ back surfaces), the Legendre-Legendre deformed mirrors, woltersine
// (PXF_OP_WSPRIMARY .. PXF_OP_ELLIPSOIDWOLTLL), between frame changes, reflections and flats.
//
// The reference traces a mirror pair as ~8 passes over the bundle (examples/axro/stressCompensation.py:57-101:
// transform, wolterprimLL, reflect, transform, woltersecLL, reflect, ..., flat).  The mirror kernels are
// fp64-VALU-bound, the passes around them HBM-bound; with the ray held in registers the frame changes and
// reflections cost nothing extra and the pair moves 120 B/ray once.  Like the Zernike programs
// (pxf_program_zern.hip) these run in their own kernel so that the common programs keep their register
// budget: one ray per thread (the Legendre-Legendre evaluation holds P_j(t), P'_j(t) of every azimuthal order
// in registers), a lean interpreter step, no `ind=` masks.
//
// Legendre-Legendre coefficient matrices do not fit the nine doubles of a ProgOp: all matrices of a program are
// packed -- zero-padded to KT1 columns (KT1 = 4, 8 or 12, the kernel instantiation; azimuthal orders 12..15 would
// push the kernel into register spills, so such a mirror runs as its own launch) and an even number of
// rows -- into ONE pool passed by value in the kernel arguments, read through wave-uniform scalar loads exactly as
// the stand-alone kernels read theirs.  Every mirror operation runs the SAME device code as its stand-alone
// entry (WSOp::apply, ll_trace, WolterSineOp::apply), so a program equals the separate launches bit for bit.
#include "pxf_ops_mirrors.h"
#include "pxf_program.h"

namespace pxf {

constexpr int kMaxOpsM = PXF_PROGRAM_MAX_OPS_MIRROR;
constexpr int kPoolCells = 288;   // two 11 x 11 matrices padded to 12 x 12
constexpr int kPoolRows = 16;     // padded axial rows of any one matrix

struct ProgramM {  // a Program cut to the operation count a mirror program may have (kernel-argument space)
    int nops;
    unsigned load, store;
    ProgOp op[kMaxOpsM];
};
struct MirrorPool {
    double za[kPoolRows], zb[kPoolRows];   // P_{n+2} = za_n x P_{n+1} - zb_n P_n
    double c[kPoolCells];
};
static_assert(sizeof(ProgramM) + sizeof(MirrorPool) + sizeof(Cols) + 64 <= 4096,
              "program + coefficient pool must fit the kernel-argument segment");

__host__ __device__ constexpr bool mprog_is_ll(int code)
{
    return code == PXF_OP_WOLTERPRIMLL || code == PXF_OP_WOLTERSECLL || code == PXF_OP_ELLIPSOIDWOLTLL;
}
__host__ __device__ constexpr bool mprog_is_mirror(int code)
{
    return mprog_is_ll(code) || code == PXF_OP_WSPRIMARY || code == PXF_OP_WSSECONDARY ||
           code == PXF_OP_WSPRIMARYBACK || code == PXF_OP_WSSECONDARYBACK || code == PXF_OP_WOLTERSINE;
}
__host__ __device__ constexpr bool mprog_supported(int code)
{
    return mprog_is_mirror(code) || code == PXF_OP_FLAT || code == PXF_OP_FLATOPD || code == PXF_OP_TRANSFORM ||
           code == PXF_OP_ITRANSFORM || code == PXF_OP_REFLECT;
}

// W-S parameter block (nine doubles).  primary: ff g k thick sbs cbs sb2 cb2s betas; secondary: ff g thick sbs
// cbs cb2s tbs tb2s betas (k = tb2s * tb2s: the product the host forms, same bits).
template <bool SECONDARY, bool BACK>
__device__ __forceinline__ WSOp<SECONDARY, BACK> ws_unpack(const double *__restrict__ P)
{
    WSOp<SECONDARY, BACK> w{};
    if (!SECONDARY) {
        w.ff = P[0]; w.g = P[1]; w.k = P[2]; w.thick = P[3];
        w.sbs = P[4]; w.cbs = P[5]; w.sb2 = P[6]; w.cb2s = P[7]; w.betas = P[8];
    } else {
        w.ff = P[0]; w.g = P[1]; w.thick = P[2]; w.sbs = P[3]; w.cbs = P[4]; w.cb2s = P[5];
        w.tbs = P[6]; w.tb2s = P[7]; w.betas = P[8];
        w.k = P[7] * P[7];
    }
    return w;
}

// L-L parameter block: three surface constants of the kind, tol, zmid, zhalf, dphi, zspan, then
// rows + 64 * (offset of the matrix in the pool)
template <int KIND, int KT1>
__device__ __forceinline__ void ll_step(const double *__restrict__ P, const MirrorPool &pool, double *c, bool act)
{
    LLParams q{};
    if (KIND == 0) { q.twop = P[0]; q.pp = P[1]; q.cst = P[2]; }
    if (KIND == 1) { q.e2 = P[0]; q.twoe2 = P[1]; q.d = P[2]; }
    if (KIND == 2) { q.aa2 = P[0]; q.bb2 = P[1]; q.zfoc = P[2]; }
    q.tol = P[3];
    q.zmid = P[4];
    q.zhalf = P[5];
    q.dphi = P[6];
    q.zspan = P[7];
    const int packed = (int)P[8];
    q.rows = packed & 63;
    q.stride = KT1;
    ll_trace<KIND, KT1 - 1>(q, pool.c + (packed >> 6), pool.za, pool.zb, c, act);
}

template <int KT1, int V>
__device__ __forceinline__ void exec_op_mirror(const ProgOp &o, const MirrorPool &pool, Ray (&r)[V],
                                               const bool (&sel)[V])
{
    const double *__restrict__ P = o.p;
    switch (o.code) {
        case PXF_OP_FLAT: run_op(FlatOp<false>{0.}, r, sel); break;
        case PXF_OP_FLATOPD: run_op(FlatOp<true>{P[0]}, r, sel); break;
        case PXF_OP_TRANSFORM:
            run_op(TransformOp<false>{P[0], P[1], P[2], P[3], P[4], P[5], P[6], P[7], P[8], o.flags & 7}, r, sel);
            break;
        case PXF_OP_ITRANSFORM:
            run_op(TransformOp<true>{P[0], P[1], P[2], P[3], P[4], P[5], P[6], P[7], P[8], o.flags & 7}, r, sel);
            break;
        case PXF_OP_REFLECT: run_op(ReflectOp{}, r, sel); break;
        case PXF_OP_WOLTERSINE: run_op(WolterSineOp{P[0], P[1], P[2], P[3], P[4], P[5], P[6]}, r, sel); break;
        case PXF_OP_WSPRIMARY: run_op(ws_unpack<false, false>(P), r, sel); break;
        case PXF_OP_WSPRIMARYBACK: run_op(ws_unpack<false, true>(P), r, sel); break;
        case PXF_OP_WSSECONDARY: run_op(ws_unpack<true, false>(P), r, sel); break;
        case PXF_OP_WSSECONDARYBACK: run_op(ws_unpack<true, true>(P), r, sel); break;
        case PXF_OP_WOLTERPRIMLL:
            if constexpr (KT1 > 0) ll_step<0, KT1>(P, pool, r[0].c, sel[0]);
            break;
        case PXF_OP_WOLTERSECLL:
            if constexpr (KT1 > 0) ll_step<1, KT1>(P, pool, r[0].c, sel[0]);
            break;
        case PXF_OP_ELLIPSOIDWOLTLL:
            if constexpr (KT1 > 0) ll_step<2, KT1>(P, pool, r[0].c, sel[0]);
            break;
        default: break;
    }
}

// KT1 = padded azimuthal column count of every L-L matrix of the program (0: the program holds none)
template <int KT1, bool NT>
__global__ __launch_bounds__(kBlock, 2) void program_mirror_kernel(Cols c, int64_t nvec, ProgramM pg, MirrorPool pool)
{
    constexpr int V = 1;
    const int64_t vv = (int64_t)blockIdx.x * kBlock + threadIdx.x;   // no grid-stride loop (pxf_kernel.h: ray_kernel)
    if (vv < nvec) {
        Ray r[V];
#pragma unroll
        for (int col = 0; col < 10; col++) {
            if (pg.load & B(col)) {  // wave-uniform
                double t[V];
                load_col<V, NT>(c.p[col], vv, t);
                r[0].c[col] = t[0];
            }
        }
        const bool sel[V] = {true};
        for (int i = 0; i < pg.nops; i++) exec_op_mirror<KT1, V>(pg.op[i], pool, r, sel);
#pragma unroll
        for (int col = 0; col < 10; col++) {
            if (pg.store & B(col)) {
                const double t[V] = {r[0].c[col]};
                const bool w[V] = {true};
                store_col<V, NT>(c.p[col], vv, t, w);
            }
        }
    }
}

template <int KT1>
static int launch_mirror_kt(const ProgramM &pm, const MirrorPool &pool, const Cols &c, int64_t num, pxf_stream_t stream)
{
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(grid_exact(num)), block(kBlock);
    if (use_nt(num)) hipLaunchKernelGGL((program_mirror_kernel<KT1, true>), grid, block, 0, s, c, num, pm, pool);
    else hipLaunchKernelGGL((program_mirror_kernel<KT1, false>), grid, block, 0, s, c, num, pm, pool);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PXF_OK : hip_fail(e);
}

// `pg`: the planned program (plan_program left the caller's scalars a[0..5] in p[0..5] of every mirror
// operation and, for the L-L kinds, the HOST pxf_ll_args_t in aux); `src` = the caller's op list in plan order.
int launch_program_mirror(const Program &pg, const Cols &c, int64_t num, pxf_stream_t stream)
{
    if (pg.nops > kMaxOpsM) return PXF_ERANGE;
    ProgramM pm{};
    pm.nops = pg.nops;
    pm.load = pg.load;
    pm.store = pg.store;
    MirrorPool pool{};
    for (int i = 0; i < kPoolRows; i++) {
        pool.za[i] = (double)(2 * i + 3) / (double)(i + 2);
        pool.zb[i] = (double)(i + 1) / (double)(i + 2);
    }
    // pass 1: fold every L-L table, find the common column count
    LLData ll[kMaxOpsM];
    int kt1 = 0;
    for (int i = 0; i < pg.nops; i++) {
        const ProgOp &o = pg.op[i];
        if (o.mask != nullptr || !mprog_supported(o.code)) return PXF_ERANGE;  // the caller runs those as separate launches
        pm.op[i] = o;
        if (!mprog_is_ll(o.code)) continue;
        const pxf_ll_args_t *la = reinterpret_cast<const pxf_ll_args_t *>(o.aux);
        if (!la) return PXF_EINVAL;
        const double *a = o.p;
        const bool prim = o.code == PXF_OP_WOLTERPRIMLL;   // a: r0 z0 zmax zmin dphi | r0 z0 psi zmax zmin dphi
        ll[i] = LLData{};
        int rc = fold_ll(ll[i], prim ? a[2] : a[3], prim ? a[3] : a[4], prim ? a[4] : a[5], la->coeff, la->axial,
                         la->az, la->cnum);
        if (rc != PXF_OK) return rc;
        if (prim) ll_setup_primary(ll[i], a[0], a[1]);
        else if (o.code == PXF_OP_WOLTERSECLL) ll_setup_secondary(ll[i], a[0], a[1], a[2]);
        else ll_setup_ellipsoid(ll[i], a[0], a[1], a[2], la->S);
        if (ll[i].kt > 11) return PXF_ERANGE;
        const int need = (ll[i].kt + 4) & ~3;   // 4, 8, 12
        if (need > kt1) kt1 = need;
    }
    // pass 2: parameter blocks; L-L matrices into the pool
    int used = 0;
    for (int i = 0; i < pg.nops; i++) {
        ProgOp &o = pm.op[i];
        const double a[6] = {o.p[0], o.p[1], o.p[2], o.p[3], o.p[4], o.p[5]};
        if (mprog_is_ll(o.code)) {
            const LLData &d = ll[i];
            const int rows = (d.ka + 2) & ~1;
            if (rows > kPoolRows || used + rows * kt1 > kPoolCells) return PXF_ERANGE;
            for (int r = 0; r <= d.ka; r++)
                for (int j = 0; j <= d.kt; j++) pool.c[used + r * kt1 + j] = d.cmat[r * (d.kt + 1) + j];
            if (o.code == PXF_OP_WOLTERPRIMLL) { o.p[0] = d.twop; o.p[1] = d.pp; o.p[2] = d.cst; }
            else if (o.code == PXF_OP_WOLTERSECLL) { o.p[0] = d.e2; o.p[1] = d.twoe2; o.p[2] = d.d; }
            else { o.p[0] = d.aa2; o.p[1] = d.bb2; o.p[2] = d.zfoc; }
            o.p[3] = d.tol;
            o.p[4] = d.zmid;
            o.p[5] = d.zhalf;
            o.p[6] = d.dphi;
            o.p[7] = d.zspan;
            o.p[8] = (double)(rows + 64 * used);
            used += rows * kt1;
            o.aux = nullptr;  // (a host pointer: never dereferenced on the device)
        } else if (o.code == PXF_OP_WOLTERSINE) {
            const WolterSineOp w = make_woltersine(a[0], a[1], a[2], a[3]);
            o.p[0] = w.twop; o.p[1] = w.pp; o.p[2] = w.cst; o.p[3] = w.amp; o.p[4] = w.freq; o.p[5] = w.pi32;
            o.p[6] = w.twopi32;
        } else if (o.code == PXF_OP_WSPRIMARY || o.code == PXF_OP_WSPRIMARYBACK) {
            WSOp<false, true> w{};
            chase(a[0], a[1], a[2], w);
            o.p[0] = w.ff; o.p[1] = w.g; o.p[2] = w.k; o.p[3] = o.code == PXF_OP_WSPRIMARYBACK ? a[3] : 0.;
            o.p[4] = w.sbs; o.p[5] = w.cbs; o.p[6] = w.sb2; o.p[7] = w.cb2s; o.p[8] = w.betas;
        } else if (o.code == PXF_OP_WSSECONDARY || o.code == PXF_OP_WSSECONDARYBACK) {
            WSOp<true, true> w{};
            chase(a[0], a[1], a[2], w);
            o.p[0] = w.ff; o.p[1] = w.g; o.p[2] = o.code == PXF_OP_WSSECONDARYBACK ? a[3] : 0.;
            o.p[3] = w.sbs; o.p[4] = w.cbs; o.p[5] = w.cb2s; o.p[6] = w.tbs; o.p[7] = w.tb2s; o.p[8] = w.betas;
        }
    }
    switch (kt1) {
        case 0: return launch_mirror_kt<0>(pm, pool, c, num, stream);
        case 4: return launch_mirror_kt<4>(pm, pool, c, num, stream);
        case 8: return launch_mirror_kt<8>(pm, pool, c, num, stream);
        case 12: return launch_mirror_kt<12>(pm, pool, c, num, stream);
        default: return PXF_ERANGE;
    }
}

}  // namespace pxf
