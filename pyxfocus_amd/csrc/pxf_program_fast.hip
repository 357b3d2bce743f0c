// pxf_program_fast.hip -- OPT-IN tolerance mode of the fused ray program (pxf_program_run_fast).
//
// The default program kernel (pxf_program.hip) is bit-identical to the reference and, for the
// two-bounce chain, no longer HBM-bound: 1237 VALU wave-instructions per wave (two rays per lane),
// most of them IEEE fp64 divides / square roots and the 3-vector work of the Newton loops, hold it at
// 2.6 ms per 1e8 rays where 120 B/ray allow ~1.95 ms (profiles/r02c_config2_kernels.md).  The
// north-star asks for < 1e-10 mm RMS, not bit identity.  This kernel trades the bits for the roof:
//
//  * Wolter-I primary / secondary (woltsurf.f95:7-54, 114-161).  The surfaces are quadrics, so along a
//    ray F(s) = (A s + B) s + C EXACTLY, and the reference's Newton step  delt = -F/(grad F . d)  is
//    the scalar Newton step of that quadratic.  The loop therefore iterates ONE number per ray
//    (3 FMAs + a divide) instead of a position, a gradient and a dot product -- the same sequence
//    of steps, the same stopping rule (strict `> tol`, fp32-literal tolerance), and the normal is still
//    built from the gradient at the point BEFORE the last step (the reference's stale-gradient quirk,
//    which moves focal positions by up to ~1e-7 mm and must be kept).
//  * divides are rcp + two Newton-Raphson steps + one residual correction (what LLVM emits for an
//    `afn` fp64 divide, <= 1 ulp); normalisations use rsq + two Newton steps and three multiplies
//    instead of sqrt + three IEEE divides; FMA contraction is on for this translation unit.
//  * a frame change whose rotation angles are all zero skips its rotations (wave-uniform test).
//
// What the tolerance is made of (tests/test_gpu_fused.py::test_fast_program_*): the
// reference evaluates the secondary's F = e^2 (d+z)^2 - z^2 - x^2 - y^2 from terms of 7e7 mm^2, i.e. with
// ~1e-8 of rounding noise, so ITS OWN converged point jitters by ~1e-9 mm along the ray with the last
// bit of the input (a 1-ulp change of a source coordinate moves the reference's focal positions by
// 4.7e-11 mm RMS, 3e-10 max -- tests/test_oracle.py::test_reference_conditioning_one_ulp).  Any
// evaluation that is not bit-identical lands inside that jitter and nowhere nearer.
//
// Supported operations: flat, transform, itransform, reflect, wolterprimary, woltersecondary, radgrat,
// spoCone, indAngle (output column) -- unmasked, no OPD variants.  Anything else: PXF_ERANGE (the caller falls back to the exact
// kernel).  Segmented bundles (per-shell parameter blocks) are supported the same way as in the
// exact kernel.
#include <stdlib.h>

#include "pxf_fastmath.h"
#include "pxf_program.h"

namespace pxf {

// ---- Newton on a quadratic along the ray -----------------------------------------------------------
// s_0 = 0;  s_{k+1} = s_k - F(s_k)/F'(s_k)  while |step| > tol (the reference's loop on the 3-D point,
// restricted to the ray).  Returns the final parameter and the parameter BEFORE the last step, where
// the reference's surface normal comes from.  Both rays of a thread iterate in one loop.
template <int V>
__device__ __forceinline__ void newton_quadratic(const double (&A)[V], const double (&Bq)[V], const double (&C)[V],
                                                 double tol, double (&s)[V], double (&sp)[V])
{
    bool act[V];
    bool any = true;
#pragma unroll
    for (int k = 0; k < V; k++) {
        s[k] = 0.;
        sp[k] = 0.;
        act[k] = true;
    }
    for (int it = 0; any && it < kNewtonCap; it++) {
        any = false;
#pragma unroll
        for (int k = 0; k < V; k++) {
            if (act[k]) {
                const double Fs = fma(fma(A[k], s[k], Bq[k]), s[k], C[k]);
                const double dF = fma(2. * A[k], s[k], Bq[k]);
                const double delt = -f_div(Fs, dF);
                sp[k] = s[k];
                s[k] = s[k] + delt;
                act[k] = fabs(delt) > tol;  // NaN ends the loop, as in the reference
                any |= act[k];
            }
        }
    }
}

template <int V>
__device__ __forceinline__ void fast_wolterprimary(const double *__restrict__ P, Ray (&r)[V])
{
    const double twop = P[0], k0 = P[1] + P[2];  // F = 2p z + (p^2 + cst) - x^2 - y^2
    double A[V], Bq[V], C[V], s[V], sp[V];
#pragma unroll
    for (int k = 0; k < V; k++) {
        const double *c = r[k].c;
        A[k] = -(c[L] * c[L] + c[M] * c[M]);
        Bq[k] = twop * c[N] - 2. * (c[X] * c[L] + c[Y] * c[M]);
        C[k] = twop * c[Z] + k0 - c[X] * c[X] - c[Y] * c[Y];
    }
    newton_quadratic<V>(A, Bq, C, PXF_TOL_1EM8, s, sp);
#pragma unroll
    for (int k = 0; k < V; k++) {
        double *c = r[k].c;
        const double Fx = -2. * fma(c[L], sp[k], c[X]), Fy = -2. * fma(c[M], sp[k], c[Y]);
        const double inv = f_rsq(Fx * Fx + Fy * Fy + twop * twop);
        c[X] = fma(c[L], s[k], c[X]);
        c[Y] = fma(c[M], s[k], c[Y]);
        c[Z] = fma(c[N], s[k], c[Z]);
        c[UX] = Fx * inv;
        c[UY] = Fy * inv;
        c[UZ] = twop * inv;
    }
}

template <int V>
__device__ __forceinline__ void fast_woltersecondary(const double *__restrict__ P, Ray (&r)[V])
{
    // F = e^2 (d+z)^2 - z^2 - x^2 - y^2.  With eps = e^2 - 1 (exact in fp64: e^2 in [1,2)) the difference
    // of the two 7e7-sized squares becomes  d (2z + d) + eps (d+z)^2  -- two terms of 2.4e4 -- so C carries
    // ~1e-11 of rounding instead of the reference's ~1e-8.
    const double e2 = P[0], twoe2 = P[1], d = P[2], eps = e2 - 1.;
    double A[V], Bq[V], C[V], s[V], sp[V];
#pragma unroll
    for (int k = 0; k < V; k++) {
        const double *c = r[k].c;
        const double dz = d + c[Z];
        A[k] = eps * (c[N] * c[N]) - c[L] * c[L] - c[M] * c[M];
        Bq[k] = 2. * (d + eps * dz) * c[N] - 2. * (c[X] * c[L] + c[Y] * c[M]);
        C[k] = d * (2. * c[Z] + d) + eps * (dz * dz) - c[X] * c[X] - c[Y] * c[Y];
    }
    newton_quadratic<V>(A, Bq, C, PXF_TOL_1EM8, s, sp);
#pragma unroll
    for (int k = 0; k < V; k++) {
        double *c = r[k].c;
        const double zp = fma(c[N], sp[k], c[Z]);
        const double Fx = -2. * fma(c[L], sp[k], c[X]), Fy = -2. * fma(c[M], sp[k], c[Y]);
        const double Fz = twoe2 * (d + zp) - 2. * zp;
        const double inv = f_rsq(Fx * Fx + Fy * Fy + Fz * Fz);
        c[X] = fma(c[L], s[k], c[X]);
        c[Y] = fma(c[M], s[k], c[Y]);
        c[Z] = fma(c[N], s[k], c[Z]);
        c[UX] = Fx * inv;
        c[UY] = Fy * inv;
        c[UZ] = Fz * inv;
    }
}

template <int V>
__device__ __forceinline__ void fast_flat(Ray (&r)[V])
{
#pragma unroll
    for (int k = 0; k < V; k++) {
        double *c = r[k].c;
        const double dd = (double)(float)(-f_div_ieee_edges(c[Z], c[N]));  // fp32 `delta` of surfacesf.f95:4-15
        c[X] = fma(dd, c[L], c[X]);
        c[Y] = fma(dd, c[M], c[Y]);
        c[Z] = 0.;
        c[UX] = 0.;
        c[UY] = 0.;
        c[UZ] = 1.;
    }
}

template <int V>
__device__ __forceinline__ void fast_reflect(Ray (&r)[V])
{
#pragma unroll
    for (int k = 0; k < V; k++) {
        double *c = r[k].c;
        const double t = 2. * (c[UX] * c[L] + c[UY] * c[M] + c[UZ] * c[N]);
        c[L] = fma(-t, c[UX], c[L]);
        c[M] = fma(-t, c[UY], c[M]);
        c[N] = fma(-t, c[UZ], c[N]);
    }
}

template <bool INVERSE>
__device__ __forceinline__ void fast_rot3(double &x, double &y, double &z, const double *__restrict__ P)
{
    // P[3..8] = cx sx cy sy cz sz; identity rotations (sin == 0, cos == 1) are skipped: wave-uniform
    if (!INVERSE) {
        if (!(P[4] == 0. && P[3] == 1.)) rotv<1>(x, y, z, P[3], P[4]);
        if (!(P[6] == 0. && P[5] == 1.)) rotv<2>(x, y, z, P[5], P[6]);
        if (!(P[8] == 0. && P[7] == 1.)) rotv<3>(x, y, z, P[7], P[8]);
    } else {
        if (!(P[8] == 0. && P[7] == 1.)) rotv<3>(x, y, z, P[7], P[8]);
        if (!(P[6] == 0. && P[5] == 1.)) rotv<2>(x, y, z, P[5], P[6]);
        if (!(P[4] == 0. && P[3] == 1.)) rotv<1>(x, y, z, P[3], P[4]);
    }
}

template <int V, bool INVERSE>
__device__ __forceinline__ void fast_transform(const double *__restrict__ P, int groups, Ray (&r)[V])
{
#pragma unroll
    for (int k = 0; k < V; k++) {
        double *c = r[k].c;
        if (groups & 1) {
            if (!INVERSE) {
                c[X] += P[0];
                c[Y] += P[1];
                c[Z] += P[2];
            }
            fast_rot3<INVERSE>(c[X], c[Y], c[Z], P);
            if (INVERSE) {
                c[X] -= P[0];
                c[Y] -= P[1];
                c[Z] -= P[2];
            }
        }
        if (groups & 2) fast_rot3<INVERSE>(c[L], c[M], c[N], P);
        if (groups & 4) fast_rot3<INVERSE>(c[UX], c[UY], c[UZ], P);
    }
}

// radgrat (transformationsf.f95:205-238): P = dpermm, order, wave, S, C (S, C: sin, cos of pi32/2)
template <int V>
__device__ __forceinline__ void fast_radgrat(const double *__restrict__ P, Ray (&r)[V])
{
    const double kk = f_div(P[1] * P[2], P[0]);  // order*wave/dpermm; / rho below
#pragma unroll
    for (int k = 0; k < V; k++) {
        double *c = r[k].c;
        const double x = c[X], y = c[Y];
        const double inv = f_rsq(x * x + y * y);  // 1/rho
        const double cphi = y * inv, sphi = x * inv;
        const double syaw = -(P[3] * cphi + P[4] * sphi), cyaw = P[4] * cphi - P[3] * sphi;
        const double g = kk * inv;
        const double sn = c[N] == 0. ? __builtin_nan("") : copysign(1., c[N]);  // n/abs(n)
        const double l = fma(syaw, g, c[L]), m = fma(-cyaw, g, c[M]);
        c[L] = l;
        c[M] = m;
        c[N] = sn * f_sqrt(1. - l * l - m * m);  // evanescent -> NaN, as the reference
    }
}

// spoCone (woltsurf.f95:591-638): P = R0, tan tg, cos tg, sin tg
template <int V>
__device__ __forceinline__ void fast_spocone(const double *__restrict__ P, Ray (&r)[V])
{
    const double R0 = P[0], sl = P[1], ctg = P[2], stg = P[3], sl2 = sl * sl;
#pragma unroll
    for (int k = 0; k < V; k++) {
        double *c = r[k].c;
        const double x = c[X], y = c[Y], z = c[Z], l = c[L], m = c[M], n = c[N];
        const double A = n * n * sl2 - m * m - l * l;
        const double Bq = 2 * n * sl * R0 + 2 * z * sl2 * n - 2 * x * l - 2 * y * m;
        const double C = R0 * R0 + 2 * sl * R0 * z + z * z * sl2 - x * x - y * y;
        const double det = Bq * Bq - 4 * A * C;
        if (det >= 0) {
            const double sq = f_sqrt(det);
            double t1 = f_div_ieee_edges(-Bq + sq, 2 * A);
            const double t2 = f_div_ieee_edges(-Bq - sq, 2 * A);
            if (fabs(t2) < fabs(t1)) t1 = t2;
            const double xn = fma(t1, l, x), yn = fma(t1, m, y);
            c[X] = xn;
            c[Y] = yn;
            c[Z] = fma(t1, n, z);
            const double g = -ctg * f_rsq(xn * xn + yn * yn);
            c[UX] = xn * g;
            c[UY] = yn * g;
            c[UZ] = stg;
        } else {
            c[L] = 0.;
            c[M] = 0.;
            c[N] = 0.;
        }
    }
}

// One operation on the V rays of a thread, two dispatch forms (A/B'd on one box, tools/ab_programs.py):
//
// FLAT (ordinary programs): a sequence of independent `if`s, each testing ONE BIT of a one-hot image of the
// op code that the planner stores in the program (opaque to the compiler).  A `switch (code)` -- or ifs on
// `code`, which the compiler turns back into one -- is lowered to a decision tree whose every level ends in a
// block of phi copies of the whole ray state (18+ v_mov_b64 per level, several levels per op): with nine
// cases the flat form is 3-17 % faster (config 2: 2.16 -> 2.10 ms, config 3: 2.63 -> 2.19 ms per 1e8 rays).
// SWITCH (segmented programs, and the 25-case exact interpreter of pxf_program.h): there the flat form
// LOSES (SPO pair 2.9 -> 3.4 ms; exact programs 5-50 % slower: two dozen taken branches per op, and
// per-lane parameters read from LDS ahead of every untaken case).
template <int V, bool NT>
__device__ __forceinline__ void exec_fast_indangle(const double *__restrict__ P, const double *aux, int64_t vv,
                                                   Ray (&r)[V])
{
    bool sel[V];  // diagnostic output column: the library's own IndAngleOp (ocml acos)
    unsigned wm[V];
    double t[V];
#pragma unroll
    for (int k = 0; k < V; k++) sel[k] = true;
    IndAngleOp{(int)P[0]}.template apply<V>(r, sel, wm);
#pragma unroll
    for (int k = 0; k < V; k++) t[k] = r[k].c[AUX];
    store_col<V, NT>(const_cast<double *>(aux), vv, t, sel);
}

template <int V, bool NT>
__device__ __forceinline__ void exec_fast_flat(int flags, const double *__restrict__ P, const double *aux, int64_t vv,
                                               Ray (&r)[V])
{
    if (flags & prog_onehot(PXF_OP_TRANSFORM)) fast_transform<V, false>(P, flags & 7, r);
    if (flags & prog_onehot(PXF_OP_WOLTERPRIMARY)) fast_wolterprimary<V>(P, r);
    if (flags & prog_onehot(PXF_OP_REFLECT)) fast_reflect<V>(r);
    if (flags & prog_onehot(PXF_OP_WOLTERSECONDARY)) fast_woltersecondary<V>(P, r);
    if (flags & prog_onehot(PXF_OP_FLAT)) fast_flat<V>(r);
    if (flags & prog_onehot(PXF_OP_SPOCONE)) fast_spocone<V>(P, r);
    if (flags & prog_onehot(PXF_OP_RADGRAT)) fast_radgrat<V>(P, r);
    if (flags & prog_onehot(PXF_OP_ITRANSFORM)) fast_transform<V, true>(P, flags & 7, r);
    if (flags & prog_onehot(PXF_OP_INDANGLE)) exec_fast_indangle<V, NT>(P, aux, vv, r);
}

template <int V, bool NT>
__device__ __forceinline__ void exec_fast_switch(int code, int flags, const double *__restrict__ P, const double *aux,
                                                 int64_t vv, Ray (&r)[V])
{
    switch (code) {
        case PXF_OP_FLAT: fast_flat<V>(r); break;
        case PXF_OP_WOLTERPRIMARY: fast_wolterprimary<V>(P, r); break;
        case PXF_OP_WOLTERSECONDARY: fast_woltersecondary<V>(P, r); break;
        case PXF_OP_TRANSFORM: fast_transform<V, false>(P, flags & 7, r); break;
        case PXF_OP_ITRANSFORM: fast_transform<V, true>(P, flags & 7, r); break;
        case PXF_OP_REFLECT: fast_reflect<V>(r); break;
        case PXF_OP_RADGRAT: fast_radgrat<V>(P, r); break;
        case PXF_OP_SPOCONE: fast_spocone<V>(P, r); break;
        case PXF_OP_INDANGLE: exec_fast_indangle<V, NT>(P, aux, vv, r); break;
        default: break;
    }
}

// Straight-line form (see pxf_program.hip): the operation sequences of the BASELINE configs as compile-time lists,
// the same fast_* bodies one after another with no dispatch.
template <int CODE, int V, bool NT>
__device__ __forceinline__ void exec_fast_fixed(int flags, const double *__restrict__ P, const double *aux, int64_t vv,
                                                Ray (&r)[V])
{
    if constexpr (CODE == PXF_OP_TRANSFORM) fast_transform<V, false>(P, flags & 7, r);
    else if constexpr (CODE == PXF_OP_ITRANSFORM) fast_transform<V, true>(P, flags & 7, r);
    else if constexpr (CODE == PXF_OP_WOLTERPRIMARY) fast_wolterprimary<V>(P, r);
    else if constexpr (CODE == PXF_OP_WOLTERSECONDARY) fast_woltersecondary<V>(P, r);
    else if constexpr (CODE == PXF_OP_REFLECT) fast_reflect<V>(r);
    else if constexpr (CODE == PXF_OP_FLAT) fast_flat<V>(r);
    else if constexpr (CODE == PXF_OP_SPOCONE) fast_spocone<V>(P, r);
    else if constexpr (CODE == PXF_OP_RADGRAT) fast_radgrat<V>(P, r);
    else if constexpr (CODE == PXF_OP_INDANGLE) exec_fast_indangle<V, NT>(P, aux, vv, r);
    else static_assert(CODE == PXF_OP_FLAT, "operation not available in a straight-line tolerance-mode program");
}

template <int... CODES>
static bool fast_chain_matches(const Program &pg)
{
    constexpr int codes[] = {CODES...};
    if (pg.nops != (int)sizeof...(CODES)) return false;
    for (int i = 0; i < pg.nops; i++)
        if (pg.op[i].code != codes[i] || pg.op[i].mask != nullptr) return false;
    return true;
}

static bool fast_supported(const Program &pg)
{
    for (int i = 0; i < pg.nops; i++) {
        const ProgOp &o = pg.op[i];
        if (o.mask || (o.aux && o.code != PXF_OP_INDANGLE)) return false;
        switch (o.code) {
            case PXF_OP_INDANGLE:
            case PXF_OP_FLAT:
            case PXF_OP_WOLTERPRIMARY:
            case PXF_OP_WOLTERSECONDARY:
            case PXF_OP_TRANSFORM:
            case PXF_OP_ITRANSFORM:
            case PXF_OP_REFLECT:
            case PXF_OP_RADGRAT:
            case PXF_OP_SPOCONE: break;
            default: return false;
        }
    }
    return true;
}

template <int V, bool NT, int... CODES>
__global__ __launch_bounds__(kBlock, 2) void program_fast_kernel(Cols c, int64_t first, int64_t nvec, Program pg)
{
    const int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x;   // no grid-stride loop (pxf_kernel.h: ray_kernel)
    if (v < nvec) {
        const int64_t vv = first + v;
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
            int i = 0;
            ((exec_fast_fixed<CODES, V, NT>(pg.op[i].flags, pg.op[i].p, pg.op[i].aux, vv, r), i++), ...);
        } else {
            for (int i = 0; i < pg.nops; i++)
                exec_fast_flat<V, NT>(pg.op[i].flags, pg.op[i].p, pg.op[i].aux, vv, r);
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

// segmented variant: per-segment parameter blocks staged in LDS, as program_seg_kernel
constexpr int kFastMaxSegPerBlock = 4;
struct FastSegInfo {
    const double *table;
    int64_t rps;
    int nseg;
};

template <bool NT, int... CODES>
__global__ __launch_bounds__(kBlock, 2) void program_fast_seg_kernel(Cols c, int64_t nvec, Program pg, FastSegInfo sg)
{
    constexpr int V = 2;
    __shared__ double s_par[kFastMaxSegPerBlock][kMaxOps][9];
    const int64_t ntiles = (nvec + kBlock - 1) / kBlock;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t v0 = tile * kBlock;
        const int64_t vlast = (v0 + kBlock - 1 < nvec ? v0 + kBlock - 1 : nvec - 1);
        const int64_t seg0 = (v0 * V) / sg.rps, seg1 = (vlast * V + V - 1) / sg.rps;
        const int nsb = (int)(seg1 - seg0 + 1);
        const int per = pg.nops * 9;
        __syncthreads();
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
                int i = 0;
                ((exec_fast_fixed<CODES, V, NT>(pg.op[i].flags, &s_par[sb][i][0], pg.op[i].aux, vv, r), i++), ...);
            } else {
                for (int i = 0; i < pg.nops; i++)
                    exec_fast_switch<V, NT>(pg.op[i].code, pg.op[i].flags, &s_par[sb][i][0], pg.op[i].aux, vv, r);
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

}  // namespace pxf

using namespace pxf;

extern "C" {

int pxf_program_run_fast(const pxf_op_t *ops, int32_t nops, double *const *cols, int64_t num, pxf_stream_t stream)
{
    Program pg{};
    int iz = -1;
    int rc = plan_program(ops, nops, pg, nullptr, &iz);
    if (rc != PXF_OK) return rc;
    if (num < 0 || !cols) return PXF_EINVAL;
    if (iz >= 0 || !fast_supported(pg)) return PXF_ERANGE;
    if (num == 0 || pg.nops == 0) return PXF_OK;
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
    const int64_t nvec2 = aligned ? num / 2 : 0;
    const int64_t done = nvec2 * 2, rest = num - done;
    static const bool no_chain = getenv("PXF_PROGRAM_NO_CHAIN") != nullptr;   // dev knob (A/B against the interpreter)
#define PXF_FAST_LAUNCH(...)                                                                                              \
    do {                                                                                                                  \
        if (nvec2 > 0) {                                                                                                  \
            if (nt) hipLaunchKernelGGL((program_fast_kernel<2, true __VA_ARGS__>), dim3(grid_exact(nvec2)), dim3(kBlock), 0, s, c, (int64_t)0, nvec2, pg); \
            else hipLaunchKernelGGL((program_fast_kernel<2, false __VA_ARGS__>), dim3(grid_exact(nvec2)), dim3(kBlock), 0, s, c, (int64_t)0, nvec2, pg);  \
        }                                                                                                                 \
        if (rest > 0) {                                                                                                   \
            if (nt) hipLaunchKernelGGL((program_fast_kernel<1, true __VA_ARGS__>), dim3(grid_exact(rest)), dim3(kBlock), 0, s, c, done, rest, pg); \
            else hipLaunchKernelGGL((program_fast_kernel<1, false __VA_ARGS__>), dim3(grid_exact(rest)), dim3(kBlock), 0, s, c, done, rest, pg);  \
        }                                                                                                                 \
    } while (0)
#define PXF_C2 PXF_OP_TRANSFORM, PXF_OP_WOLTERPRIMARY, PXF_OP_REFLECT, PXF_OP_WOLTERSECONDARY, PXF_OP_REFLECT, PXF_OP_FLAT
#define PXF_GR PXF_OP_TRANSFORM, PXF_OP_FLAT, PXF_OP_REFLECT, PXF_OP_RADGRAT, PXF_OP_ITRANSFORM, PXF_OP_FLAT
    // (the two-bounce program alone is already HBM-bound under the flat-if interpreter, and measured 2.5 % FASTER
    // there than as a straight line -- 2.11 vs 2.17 ms per 1e8 rays, tools/ab_programs.py -- so it keeps the interpreter)
    if (!no_chain && fast_chain_matches<PXF_GR>(pg)) PXF_FAST_LAUNCH(, PXF_GR);
    else if (!no_chain && fast_chain_matches<PXF_C2, PXF_GR>(pg)) PXF_FAST_LAUNCH(, PXF_C2, PXF_GR);
    else PXF_FAST_LAUNCH();
#undef PXF_FAST_LAUNCH
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PXF_OK : hip_fail(e);
}

int pxf_program_run_segments_fast(const pxf_op_t *ops, int32_t nops, const double *seg_params_dev, int32_t nseg,
                                  int64_t rays_per_seg, double *const *cols, int64_t num, pxf_stream_t stream)
{
    Program pg{};
    int iz = -1;
    int rc = plan_program(ops, nops, pg, nullptr, &iz);
    if (rc != PXF_OK) return rc;
    if (iz >= 0 || !fast_supported(pg)) return PXF_ERANGE;
    if (num < 0 || !cols || nseg < 1 || !seg_params_dev) return PXF_EINVAL;
    if (rays_per_seg < 2 || (rays_per_seg & 1) || rays_per_seg * (kFastMaxSegPerBlock - 1) < 2 * kBlock) return PXF_ERANGE;
    if (num != (int64_t)nseg * rays_per_seg) return PXF_EINVAL;
    if (num == 0 || pg.nops == 0) return PXF_OK;
    Cols c{};
    for (int col = 0; col < 10; col++) {
        c.p[col] = cols[col];
        if ((pg.load | pg.store) & B(col)) {
            if (!cols[col]) return PXF_EINVAL;
            if ((uintptr_t)cols[col] & 15) return PXF_EINVAL;
        }
    }
    for (int i = 0; i < pg.nops; i++)
        if (pg.op[i].aux && ((uintptr_t)pg.op[i].aux & 15)) return PXF_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int64_t nvec = num / 2;
    FastSegInfo sg{seg_params_dev, rays_per_seg, nseg};
    const dim3 grid(grid_for(nvec)), block(kBlock);
#define PXF_SPO PXF_OP_TRANSFORM, PXF_OP_SPOCONE, PXF_OP_REFLECT, PXF_OP_INDANGLE, PXF_OP_SPOCONE, PXF_OP_REFLECT, PXF_OP_INDANGLE, PXF_OP_TRANSFORM
    if (getenv("PXF_PROGRAM_NO_CHAIN") == nullptr && fast_chain_matches<PXF_SPO>(pg)) {   // the Arcus SPO pair per shell
        if (use_nt(num)) hipLaunchKernelGGL((program_fast_seg_kernel<true, PXF_SPO>), grid, block, 0, s, c, nvec, pg, sg);
        else hipLaunchKernelGGL((program_fast_seg_kernel<false, PXF_SPO>), grid, block, 0, s, c, nvec, pg, sg);
    } else {
        if (use_nt(num)) hipLaunchKernelGGL((program_fast_seg_kernel<true>), grid, block, 0, s, c, nvec, pg, sg);
        else hipLaunchKernelGGL((program_fast_seg_kernel<false>), grid, block, 0, s, c, nvec, pg, sg);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PXF_OK : hip_fail(e);
}

}  // extern "C"
