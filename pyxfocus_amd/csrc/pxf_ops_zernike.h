// pxf_ops_zernike.h -- the Zernike evaluation and Newton Op (zernsurf.f95:8-203 with zernset,
// specialFunctions.f95:142-232), shared by the stand-alone kernels (pxf_zernike.hip) and the fused
// program kernel (pxf_program_zern.hip).
//
// How the sum is evaluated.  S(x,y) = sum_i coeff_i Z_i(rho/rad, theta) is ONE bivariate polynomial of
// degree N <= 15 in u = x/rad, v = y/rad.  The host folds the caller's (coeff, rorder, aorder) list into one
// (Ac, As) pair per (n,|m|) -- including the sqrt(2(n+1)) normalisation and the reference's single-precision
// sqrt(0.5) for m = 0 (specialFunctions.f95:225) -- and expands that into the (N+1)(N+2)/2 MONOMIAL
// coefficients (integer Zernike coefficients, accumulated in long double: zern_monomials).  The table
// travels by value in the kernel arguments, each workgroup stages it in LDS, and the kernel evaluates S,
// dS/du, dS/dv by nested Horner schemes: N(N+1) + 3(N+1) FMAs per evaluation -- 80 for the 36-term case.
// No theta, no rho, no square root, no divide, no per-thread arrays.
//
// (Round 1 evaluated the polar form -- cos/sin(m theta) recurrences, rho^n, the reference's q-recursion for
// R_n^m and dR/drho, three sums per (n,m): ~330 instructions per evaluation at 36 terms, fp64-VALU-bound at
// 3.7 ms per 1e8 rays; and its orders 9..15 ran a run-time-loop form at 11-23 ms.  The Horner form is
// HBM-bound up to order 8: 1.6 ms.)
//
// Conditioning.  On the unit disk the monomial basis is the worse-conditioned one: its coefficients reach
// ~20 x the term amplitudes at degree 10 and ~4e3 x at degree 15.  Measured against the reference's zernset
// with coefficient amplitudes of 1e-3 mm (tests/test_host.py): 9e-16 mm in S at degree 10, 5e-14 mm at
// degree 15 -- three to five orders inside the 2e-11 mm these kernels are held to; the errors scale with the
// amplitudes.
//
// Floating-point contraction: this code is tolerance-parity, so its hot function bodies ask for FMA
// contraction THEMSELVES (`#pragma clang fp contract(fast)`); a translation unit compiled
// -ffp-contract=off (as every bit-exact Op needs) can include this header and still run it fused.
#pragma once
#include <math.h>

#include "pxf_fastmath.h"
#include "pxf_kernel.h"

namespace pxf {

constexpr int kZernMaxN = 15;
constexpr int zern_npairs(int nmax) { return ((nmax + 2) / 2) * ((nmax + 3) / 2); }  // #(n,m): n<=nmax
// index of pair (n,m) (m same parity as n, 0<=m<=n) in the folded table: pairs of all smaller n, then m/2
constexpr int zern_pair(int n, int m) { return zern_npairs(n - 1) + m / 2; }
constexpr int zern_nmono(int n) { return (n + 1) * (n + 2) / 2; }  // monomials of degree <= n
// Horner layout: blocks j = 0..N (power of v); block j holds the N-j+1 coefficients of u^(N-j) .. u^0
constexpr int zern_mono_off(int N, int j) { return j * (N + 1) - j * (j - 1) / 2; }
// up to this degree the compiler keeps the whole table in registers across the Newton loop (it hoists the
// loop-invariant LDS reads: 72 VGPRs at degree 7, 132 at 10); above, the reads stay inside the loop
constexpr int kZernHoistMaxN = 10;

struct ZernTab {
    double2 a[zern_npairs(kZernMaxN)];  // .x = Ac (cos-type, and m=0), .y = As (sin-type)
};

// t: the Horner-ordered monomial table in LDS (every lane reads the same address: broadcast).  Value and both
// partial derivatives in one sweep: for each power j of v, P_j(u) and P_j'(u) by Horner in u, then the three
// Horner recurrences in v (S, dS/du, dS/dv).  rho == 0 gives NaN, as the reference's division by rho does
// (zernsurf.f95:63-66).
template <int NMAX>
__device__ __forceinline__ void zern_eval(const double *__restrict__ t, double x, double y, double inv_rad, double &S,
                                          double &Sx, double &Sy)
{
#pragma clang fp contract(fast)
    if constexpr (NMAX > kZernHoistMaxN) asm volatile("" ::: "memory");  // re-read the table from LDS each time
    const double u = x * inv_rad, v = y * inv_rad;
    double s = 0., su = 0., sv = 0.;
#pragma unroll
    for (int j = NMAX; j >= 0; j--) {
        const double *c = t + zern_mono_off(NMAX, j);
        double p = c[0], dp = 0.;
#pragma unroll
        for (int i = 1; i <= NMAX - j; i++) {
            dp = fma(dp, u, p);
            p = fma(p, u, c[i]);
        }
        sv = fma(sv, v, s);
        s = fma(s, v, p);
        su = fma(su, v, dp);
    }
    const double poison = (x == 0. && y == 0.) ? __builtin_nan("") : 0.;  // value AND gradient, like the reference
    S = s + poison;
    Sx = su * inv_rad + poison;
    Sy = sv * inv_rad + poison;
}

// CONIC: EXTENSION without a reference counterpart (BASELINE config 4 names a "Zernike-perturbed
// conic"; the reference's zernsurf is a pure Zernike sag): the sag of a conic of radius
// R and constant K is added under the Zernike sum,  F = z - c rho^2/(1+sqrt(1-(K+1) c^2 rho^2)) - S.
// Parity unpinned; checked against the conic kernel (coeff = 0) and an independent NumPy Newton.
#ifndef PXF_ZERN_MIN_WAVES
// >= 2 waves per SIMD keeps a kernel within 256 registers per lane, so the compiler has no AGPRs to spill
// into: its VGPR->AGPR "Reload Reuse" path can drop half of a 64-bit copy (profiles/r02_zernike_miscompile.md).
#define PXF_ZERN_MIN_WAVES(NMAX) 2
#endif
template <int NMAX, bool CONIC = false>
struct ZernOp {
    static constexpr int MAXV = 2;  // the Horner form is light enough for two rays per lane (16-byte accesses)
    static constexpr int MIN_WAVES = PXF_ZERN_MIN_WAVES(NMAX);
    static constexpr unsigned RD = POS | DIR | B(OPD);  // opd column only touched when has_opd
    static constexpr unsigned WR = POS | NRM | B(OPD);
    static constexpr int NP = zern_npairs(NMAX);
    static constexpr int NM = zern_nmono(NMAX);
    double tab_arg[zern_nmono(NMAX)];  // Horner-ordered monomial table, by value in the kernel-argument segment
    double rad, nr;
    int has_opd;
    double cc = 0., K1 = 0.;  // conic: curvature 1/R and K+1 (CONIC only)

    __device__ __forceinline__ double *lds() const
    {
        __shared__ double s_tab[zern_nmono(NMAX)];
        return s_tab;
    }

    __device__ __forceinline__ void prologue() const
    {
        double *s = lds();
        for (int i = threadIdx.x; i < NM; i += kBlock) s[i] = tab_arg[i];
        __syncthreads();
    }

    template <int V>
    __device__ __forceinline__ void apply(Ray (&r)[V], const bool (&sel)[V], unsigned (&wm)[V]) const
    {
#pragma clang fp contract(fast)
        const double *__restrict__ tab = lds();
        const double inv_rad = f_rcp(rad);
        double t[V], Fx[V], Fy[V], Fz[V];
        bool act[V];
        bool any = false;
#pragma unroll
        for (int k = 0; k < V; k++) {
            act[k] = sel[k];
            any |= act[k];
            t[k] = 0.;
            Fx[k] = Fy[k] = Fz[k] = 0.;
        }
        // all V rays of a thread iterate in one loop (independent chains interleave); a ray stops updating
        // the moment its own |delta| <= tol, so each sees the reference's iteration sequence
        for (int it = 0; any && it < kNewtonCap; it++) {
            any = false;
#pragma unroll
            for (int k = 0; k < V; k++) {
                if (act[k]) {
                    double *c = r[k].c;
                    double S, Sx, Sy;
                    zern_eval<NMAX>(tab, c[X], c[Y], inv_rad, S, Sx, Sy);
                    double F = c[Z] - S;  // F = z - sum c_i Z_i ; grad F = (-dS/dx, -dS/dy, 1)
                    Fx[k] = -Sx;
                    Fy[k] = -Sy;
                    Fz[k] = 1.;
                    if (CONIC) {
                        const double r2 = c[X] * c[X] + c[Y] * c[Y];
                        const double sq = sqrt(1. - K1 * cc * cc * r2);
                        F -= cc * r2 / (1. + sq);
                        const double g = cc / sq;  // (1/rho) d(sag)/d(rho)
                        Fx[k] -= g * c[X];
                        Fy[k] -= g * c[Y];
                    }
                    const double Fp = Fx[k] * c[L] + Fy[k] * c[M] + Fz[k] * c[N];
                    const double delta = -f_div(F, Fp);
                    c[X] = c[X] + c[L] * delta;
                    c[Y] = c[Y] + c[M] * delta;
                    c[Z] = c[Z] + c[N] * delta;
                    t[k] = t[k] + delta;
                    act[k] = fabs(delta) > PXF_TOL_1EM10;
                    any |= act[k];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < V; k++) {
            double *c = r[k].c;
            const double iFp = f_rsq(Fx[k] * Fx[k] + Fy[k] * Fy[k] + Fz[k] * Fz[k]);
            c[UX] = Fx[k] * iFp;
            c[UY] = Fy[k] * iFp;
            c[UZ] = Fz[k] * iFp;
            c[OPD] = c[OPD] + t[k] * nr;  // zernsurf.f95:199 -- once, from the accumulated path length
            wm[k] = POS | NRM | (has_opd ? B(OPD) : 0u);
        }
    }
};

// The opd column is optional.  The generic kernel loads every column an Op declares in RD, so the
// OPD and plain variants are distinct types that differ only in their RD/WR sets (the plain one
// never touches the opd column: 96 B/ray instead of 112).
template <int NMAX, bool WOPD, bool CONIC = false>
struct ZernOpSel : ZernOp<NMAX, CONIC> {
    static constexpr unsigned RD = POS | DIR | (WOPD ? B(OPD) : 0u);
    static constexpr unsigned WR = POS | NRM | (WOPD ? B(OPD) : 0u);
    template <int V>
    __device__ __forceinline__ void apply(Ray (&r)[V], const bool (&sel)[V], unsigned (&wm)[V]) const
    {
        if (!WOPD) {
#pragma unroll
            for (int k = 0; k < V; k++) r[k].c[OPD] = 0.;
        }
        ZernOp<NMAX, CONIC>::template apply<V>(r, sel, wm);
    }
};

// ---- host: fold (coeff, rorder, aorder) into the (n,|m|) table ----------------------------
// `rot`: the set is evaluated at theta+rot (tracezernrot's second set, zernsurf.f95:302): by the
// angle-addition identities  Ac cos(m(t+rot)) + As sin(m(t+rot))  =
//   [Ac cos(m rot) + As sin(m rot)] cos(m t) + [As cos(m rot) - Ac sin(m rot)] sin(m t),
// so a rotated set folds into the SAME table and costs the kernel nothing.  `accumulate` adds to
// an already folded table.
inline int fold_table(const double *coeff, const int32_t *rorder, const int32_t *aorder, int32_t arrsize,
                      ZernTab &tab, int &nmax, double rot = 0., bool accumulate = false)
{
    if (arrsize < 0 || (arrsize > 0 && (!coeff || !rorder || !aorder))) return PXF_EINVAL;
    if (!accumulate) {
        for (auto &a : tab.a) a = make_double2(0., 0.);
        nmax = 0;
    }
    const double sqrt_half_f32 = (double)sqrtf(0.5f);  // specialFunctions.f95:225: REAL*4 sqrt(0.5)
    for (int i = 0; i < arrsize; i++) {
        const int n = rorder[i], mm = aorder[i], m = mm < 0 ? -mm : mm;
        if (n < 0 || m > n || ((n - m) & 1)) return PXF_EINVAL;  // the reference reads garbage here
        if (n > kZernMaxN) return PXF_ERANGE;
        if (n > nmax) nmax = n;
        const double norm = sqrt(2. * (n + 1));
        double2 &a = tab.a[zern_pair(n, m)];
        double ac = 0., as = 0.;
        if (mm < 0)
            as = coeff[i] * norm;
        else if (mm > 0)
            ac = coeff[i] * norm;
        else
            ac = coeff[i] * (norm * sqrt_half_f32);
        if (rot != 0. && m > 0) {
            const double cr = cos(m * rot), sr = sin(m * rot);
            const double ac2 = ac * cr + as * sr, as2 = as * cr - ac * sr;
            ac = ac2;
            as = as2;
        }
        a.x += ac;
        a.y += as;
    }
    return PXF_OK;
}

// ---- host: (Ac,As) pairs -> Horner-ordered monomial table (Cartesian form) ---------------------------------
// R_n^m(rho) [cos, sin](m theta) with  R_n^m = sum_k (-1)^k (n-k)! / (k! ((n+m)/2-k)! ((n-m)/2-k)!) rho^(n-2k):
// rho^(n-2k) [cos,sin](m theta) = (u^2+v^2)^((n-m)/2-k) [Re,Im](u+iv)^m, all integer coefficients
// (< 2^53 for n <= 15), accumulated in long double and rounded once.  `out` receives zern_nmono(N) doubles.
inline void zern_monomials(const ZernTab &tab, int N, double *out)
{
    constexpr int D = kZernMaxN + 1;
    static long double fact[2 * D];
    static bool init = false;
    if (!init) {
        fact[0] = 1.L;
        for (int i = 1; i < 2 * D; i++) fact[i] = fact[i - 1] * i;
        init = true;
    }
    auto binom = [&](int a, int b) { return fact[a] / (fact[b] * fact[a - b]); };
    long double c[D][D] = {};
    for (int n = 0; n <= N; n++)
        for (int m = n & 1; m <= n; m += 2) {
            const double2 a = tab.a[zern_pair(n, m)];
            if (a.x == 0. && a.y == 0.) continue;
            for (int k = 0; k <= (n - m) / 2; k++) {
                const long double rk = ((k & 1) ? -1.L : 1.L) * fact[n - k] /
                                       (fact[k] * fact[(n + m) / 2 - k] * fact[(n - m) / 2 - k]);
                const int p = (n - m) / 2 - k;  // power of (u^2 + v^2)
                for (int s = 0; s <= p; s++) {
                    const long double b = binom(p, s);  // u^(2(p-s)) v^(2s)
                    for (int q = 0; q <= m; q++) {      // (u+iv)^m = sum_q C(m,q) u^(m-q) (iv)^q
                        const long double w = rk * b * binom(m, q) * (((q / 2) & 1) ? -1.L : 1.L);
                        const int i = 2 * (p - s) + m - q, j = 2 * s + q;
                        c[i][j] += (q & 1) ? (long double)a.y * w : (long double)a.x * w;
                    }
                }
            }
        }
    for (int j = 0; j <= N; j++)
        for (int i = 0; i <= N - j; i++) out[zern_mono_off(N, j) + i] = (double)c[N - j - i][j];
}

// Fill an Op's by-value table for the instantiation NMAX (>= the largest order present in `tab`).
template <int NMAX>
inline void zern_fill_table(double (&dst)[zern_nmono(NMAX)], const ZernTab &tab)
{
    zern_monomials(tab, NMAX, dst);
}

}  // namespace pxf
