/*
 * pxf.h -- C ABI of libpxf_hip.so, the MI355X (gfx950) ray-trace core that drops in behind
 * PyXFocus' f2py extension modules (compiletrace.sh:1-6 in the reference tree).
 *
 * One entry point per reference Fortran subroutine on the hot path.  Argument ORDER and
 * MEANING follow the Fortran subroutine each one replaces (cited per function, file:line in
 * the reference); the differences are the ones a C ABI needs:
 *     - `num` is int64_t by value (the reference's default INTEGER caps a call at 2^31-1 rays),
 *     - scalars are passed by value,
 *     - every ray column is a DEVICE pointer to `num` contiguous float64 (structure-of-arrays:
 *       the PyXFocus ray bundle is the list [opd,x,y,z,l,m,n,ux,uy,uz], sources.py:1-15),
 *     - two trailing arguments: `mask` (NULL, or a device uint8[num]; rays with mask[i]==0 are
 *       left untouched -- this is the reference wrappers' `ind=` gather/compute/scatter,
 *       surfaces.py:17-24 / transformations.py:20-27, done as an in-kernel predicate) and
 *       `stream` (a hipStream_t passed as void*; NULL = the default stream).
 *
 * All calls are asynchronous on `stream`, stateless and re-entrant.  Pointers are borrowed for
 * the duration of the launch; nothing is allocated or retained, except where a `workspace`
 * argument is documented.  No exceptions cross the boundary: the return value is PXF_OK or a
 * negative PXF_E* code.  Physics failures stay IN BAND exactly as in the reference (zeroed
 * direction cosines, NaN, ...), see DESIGN.md.
 *
 * Arithmetic is fp64 throughout, compiled without FMA contraction, in the reference's operation
 * order, so flat/reflect/transform/conic/Wolter results are bit-identical to the x86 reference.
 */
#ifndef PXF_H
#define PXF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PXF_ABI_VERSION 1

#define PXF_OK 0
#define PXF_EINVAL (-1)   /* NULL column, negative num, bad order table, ... */
#define PXF_EHIP (-2)     /* a HIP runtime call failed; see pxf_last_hip_error() */
#define PXF_ENODEV (-3)   /* no gfx950 device / code object not loadable */
#define PXF_ERANGE (-4)   /* parameter outside the supported range (e.g. Zernike order > 15) */

typedef void *pxf_stream_t; /* hipStream_t */

int pxf_abi_version(void);
const char *pxf_strerror(int code);
int pxf_last_hip_error(void); /* hipError_t of the last failing HIP call on this thread */
/* PXF_OK if the calling thread's current HIP device is a gfx950 (the only code object the library carries),
 * PXF_ENODEV otherwise (no device, or another architecture).  Every launch makes the same check and returns
 * PXF_ENODEV itself; this entry lets a binding ask up front.  Cached per device ordinal. */
int pxf_device_check(void);
/* Self-test of the shared-reciprocal divisions inside the bit-exact kernels (csrc/pxf_exactdiv.h): `ntuples` random
 * operand tuples -- exponents across and around the fast path's window, the whole fp64 range, zeros, denormals,
 * infinities, NaNs -- through every such routine and through `/`; *mismatches_dev (device uint64) receives the number
 * of results that differ in any bit (NaN == NaN).  Must be 0.  Run by the GPU test suite. */
int pxf_selftest_exact_division(int64_t ntuples, uint64_t seed, uint64_t *mismatches_dev, pxf_stream_t stream);

/* ---- surfacesf.f95 --------------------------------------------------------------------- */

/* flat(x,y,z,l,m,n,ux,uy,uz,num)            surfacesf.f95:4-29
 * Propagate to z=0; normal (0,0,1).  The propagation distance is rounded to fp32 as in the
 * reference (implicit REAL*4 `delta`). */
int pxf_flat(double *x, double *y, double *z, const double *l, const double *m, const double *n,
             double *ux, double *uy, double *uz, int64_t num, const uint8_t *mask, pxf_stream_t stream);

/* flatOPD(x,y,z,l,m,n,ux,uy,uz,opd,num,nr)  surfacesf.f95:32-53 */
int pxf_flatopd(double *x, double *y, double *z, const double *l, const double *m, const double *n,
                double *ux, double *uy, double *uz, double *opd, int64_t num, double nr,
                const uint8_t *mask, pxf_stream_t stream);

/* conic(x,y,z,l,m,n,ux,uy,uz,num,R,K)       surfacesf.f95:302-360 */
int pxf_conic(double *x, double *y, double *z, double *l, double *m, double *n, double *ux,
              double *uy, double *uz, int64_t num, double R, double K, const uint8_t *mask,
              pxf_stream_t stream);

/* conicopd(opd,x,y,z,l,m,n,ux,uy,uz,num,R,K,nr)  surfacesf.f95:366-420 */
int pxf_conicopd(double *opd, double *x, double *y, double *z, double *l, double *m, double *n,
                 double *ux, double *uy, double *uz, int64_t num, double R, double K, double nr,
                 const uint8_t *mask, pxf_stream_t stream);

/* tracesphere(x,...,uz,num,rad) / tracesphereOPD(opd,x,...,uz,num,rad,nr)   surfacesf.f95:57-98 / 101-148
 * Sphere of radius rad about the origin, nearest root; a miss zeroes position and direction. */
int pxf_tracesphere(double *x, double *y, double *z, double *l, double *m, double *n, double *ux,
                    double *uy, double *uz, int64_t num, double rad, const uint8_t *mask, pxf_stream_t stream);
int pxf_tracesphereopd(double *opd, double *x, double *y, double *z, double *l, double *m, double *n,
                       double *ux, double *uy, double *uz, int64_t num, double rad, double nr,
                       const uint8_t *mask, pxf_stream_t stream);
/* tracecyl / tracecylOPD   surfacesf.f95:153-196 / 199-246   (cylinder axis = y) */
int pxf_tracecyl(double *x, double *y, double *z, double *l, double *m, double *n, double *ux, double *uy,
                 double *uz, int64_t num, double rad, const uint8_t *mask, pxf_stream_t stream);
int pxf_tracecylopd(double *opd, double *x, double *y, double *z, double *l, double *m, double *n,
                    double *ux, double *uy, double *uz, int64_t num, double rad, double nr,
                    const uint8_t *mask, pxf_stream_t stream);
/* cylconic(x,...,uz,num,rad,k)   surfacesf.f95:251-296   (Newton; `rad` is a curvature) */
int pxf_cylconic(double *x, double *y, double *z, const double *l, const double *m, const double *n,
                 double *ux, double *uy, double *uz, int64_t num, double rad, double k,
                 const uint8_t *mask, pxf_stream_t stream);
/* paraxial / paraxialY (x,...,uz,num,F)   surfacesf.f95:423-439 / 443-459   (only x,y,l,m are touched) */
int pxf_paraxial(const double *x, const double *y, double *l, double *m, int64_t num, double F,
                 const uint8_t *mask, pxf_stream_t stream);
int pxf_paraxialy(const double *y, double *m, int64_t num, double F, const uint8_t *mask, pxf_stream_t stream);

/* torus(x,...,uz,num,rin,rout)   surfacesf.f95:468-507   (Newton; normal as the reference builds it) */
int pxf_torus(double *x, double *y, double *z, const double *l, const double *m, const double *n, double *ux,
              double *uy, double *uz, int64_t num, double rin, double rout, const uint8_t *mask,
              pxf_stream_t stream);
/* conicplus(x,...,uz,num,R,K,p,Np) / conicplusopd(opd,x,...,uz,num,R,K,p,Np,nr)   surfacesf.f95:514-638
 * p = HOST array of Np (<= 16) even-polynomial coefficients (p[0] quadratic, p[1] quartic, ...). */
int pxf_conicplus(double *x, double *y, double *z, const double *l, const double *m, const double *n,
                  double *ux, double *uy, double *uz, int64_t num, double R, double K, const double *p,
                  int32_t Np, const uint8_t *mask, pxf_stream_t stream);
int pxf_conicplusopd(double *opd, double *x, double *y, double *z, const double *l, const double *m,
                     const double *n, double *ux, double *uy, double *uz, int64_t num, double R, double K,
                     const double *p, int32_t Np, double nr, const uint8_t *mask, pxf_stream_t stream);
/* legSurf(x,...,uz,xwidth,ywidth,order,coeff,xo,yo,Nc,num)   surfacesf.f95:642-668
 * 2-D Legendre phase surface; only x,y,l,m,n are touched.  coeff/xo/yo HOST arrays, Nc <= 48, orders <= 12. */
int pxf_legsurf(const double *x, const double *y, double *l, double *m, double *n, int64_t num, double xwidth,
                double ywidth, double order, const double *coeff, const int32_t *xo, const int32_t *yo,
                int32_t Nc, const uint8_t *mask, pxf_stream_t stream);

/* ---- woltsurf.f95 ---------------------------------------------------------------------- */

/* Van Speybroeck constants p,d,e from (r0,z0,psi), woltsurf.f95:18-23, evaluated on the host
 * with libm exactly as the reference does at the top of every Wolter routine. */
void pxf_wolter_params(double r0, double z0, double psi, double *p, double *d, double *e);

/* wolterprimary(x,...,uz,num,r0,z0,psi)      woltsurf.f95:7-54   (Newton, tol 1.e-8 fp32 literal) */
int pxf_wolterprimary(double *x, double *y, double *z, const double *l, const double *m,
                      const double *n, double *ux, double *uy, double *uz, int64_t num, double r0,
                      double z0, double psi, const uint8_t *mask, pxf_stream_t stream);

/* wolterprimaryopd(opd,x,...,uz,num,r0,z0,psi,nr)  woltsurf.f95:60-108 (tol 1.e-10) */
int pxf_wolterprimaryopd(double *opd, double *x, double *y, double *z, const double *l,
                         const double *m, const double *n, double *ux, double *uy, double *uz,
                         int64_t num, double r0, double z0, double psi, double nr,
                         const uint8_t *mask, pxf_stream_t stream);

/* woltersecondary(x,...,uz,num,r0,z0,psi)    woltsurf.f95:114-161 */
int pxf_woltersecondary(double *x, double *y, double *z, const double *l, const double *m,
                        const double *n, double *ux, double *uy, double *uz, int64_t num, double r0,
                        double z0, double psi, const uint8_t *mask, pxf_stream_t stream);

/* spoCone(x,...,uz,num,R0,tg)   woltsurf.f95:591-638   (SPO cone; a miss zeroes the direction) */
int pxf_spocone(double *x, double *y, double *z, double *l, double *m, double *n, double *ux, double *uy,
                double *uz, int64_t num, double R0, double tg, const uint8_t *mask, pxf_stream_t stream);

/* woltersine(x,...,uz,num,r0,z0,amp,freq)   woltsurf.f95:167-215   (primary with a sinusoidal axial ripple) */
int pxf_woltersine(double *x, double *y, double *z, const double *l, const double *m, const double *n,
                   double *ux, double *uy, double *uz, int64_t num, double r0, double z0, double amp,
                   double freq, const uint8_t *mask, pxf_stream_t stream);

/* Legendre-Legendre deformed mirrors.  coeff/axial/az are HOST arrays of length cnum (folded on the
 * host into a dense (axial x az) matrix that the kernel stages in LDS; PXF_ERANGE if that matrix
 * would exceed 256 entries).
 * wolterprimLL(x,...,uz,num,r0,z0,zmax,zmin,dphi,coeff,axial,az,cnum)          woltsurf.f95:220-288
 * woltersecLL(x,...,uz,num,r0,z0,psi,zmax,zmin,dphi,coeff,axial,az,cnum)       woltsurf.f95:293-379
 * ellipsoidWoltLL(x,...,uz,num,r0,z0,psi,S,zmax,zmin,dphi,coeff,axial,az,cnum) woltsurf.f95:643-718 */
int pxf_wolterprimll(double *x, double *y, double *z, const double *l, const double *m, const double *n,
                     double *ux, double *uy, double *uz, int64_t num, double r0, double z0, double zmax,
                     double zmin, double dphi, const double *coeff, const int32_t *axial, const int32_t *az,
                     int32_t cnum, const uint8_t *mask, pxf_stream_t stream);
int pxf_woltersecll(double *x, double *y, double *z, const double *l, const double *m, const double *n,
                    double *ux, double *uy, double *uz, int64_t num, double r0, double z0, double psi,
                    double zmax, double zmin, double dphi, const double *coeff, const int32_t *axial,
                    const int32_t *az, int32_t cnum, const uint8_t *mask, pxf_stream_t stream);
int pxf_ellipsoidwoltll(double *x, double *y, double *z, const double *l, const double *m, const double *n,
                        double *ux, double *uy, double *uz, int64_t num, double r0, double z0, double psi,
                        double S, double zmax, double zmin, double dphi, const double *coeff,
                        const int32_t *axial, const int32_t *az, int32_t cnum, const uint8_t *mask,
                        pxf_stream_t stream);

/* wsprimary / wssecondary (x,...,uz,num,alpha,z0,psi)   woltsurf.f95:387-476 / 484-588
 * Wolter-Schwarzschild pair; 26-iteration cap: a ray that does not converge gets its start
 * position back and keeps its previous normal (callers vignette on "did not move"). */
int pxf_wsprimary(double *x, double *y, double *z, const double *l, const double *m, const double *n,
                  double *ux, double *uy, double *uz, int64_t num, double alpha, double z0, double psi,
                  const uint8_t *mask, pxf_stream_t stream);
int pxf_wssecondary(double *x, double *y, double *z, const double *l, const double *m, const double *n,
                    double *ux, double *uy, double *uz, int64_t num, double alpha, double z0, double psi,
                    const uint8_t *mask, pxf_stream_t stream);

/* wsprimaryBack / wssecondaryBack (x,...,uz,num,alpha,z0,psi,thick)   woltsurf.f95:726-816 / 826-933 */
int pxf_wsprimaryback(double *x, double *y, double *z, const double *l, const double *m, const double *n,
                      double *ux, double *uy, double *uz, int64_t num, double alpha, double z0, double psi,
                      double thick, const uint8_t *mask, pxf_stream_t stream);
int pxf_wssecondaryback(double *x, double *y, double *z, const double *l, const double *m, const double *n,
                        double *ux, double *uy, double *uz, int64_t num, double alpha, double z0, double psi,
                        double thick, const uint8_t *mask, pxf_stream_t stream);
/* Host utility (no GPU): beta = atan2(r, z), cos(beta) and tan(beta/2) exactly as the Wolter-Schwarzschild
 * secondary kernels form them -- a restatement of the x86-64 C library's table-driven cosine and of its
 * small-argument tangent polynomial, whose rounding the reference's `(1 - cos(beta))` (woltsurf.f95:518-519)
 * amplifies to 2.6e-9 mm per ulp.  NaN where the pair lies outside the routines' range (r/z >= 1.15, z <= 0;
 * tan: beta/2 >= 0.0607): the kernels then use the correctly rounded quotient / the half-angle identity.
 * Exported so that a CPU test can hold them to libm's atan2 / cos / tan bit for bit. */
int pxf_host_cos_atan2(const double *r, const double *z, double *beta, double *cosbeta, double *tanhalf,
                       int64_t num);
/* Host utility (no GPU): theta = atan2(y, x) and sincos(theta) exactly as the back-surface W-S kernels form the
 * substrate offset (woltsurf.f95:760-763, 862-865): the correctly rounded arctangent and a restatement of the C
 * library's `sincos` (the non-FMA build of the same algorithm).  NaN for zero / non-finite / extreme operands. */
int pxf_host_sincos_atan2(const double *y, const double *x, double *theta, double *sintheta, double *costheta,
                          int64_t num);

/* ---- zernsurf.f95 (+ zernset, specialFunctions.f95:142-232) ------------------------------- */

/* tracezern(x,...,uz,num,coeff,rorder,aorder,arrsize,rad)   zernsurf.f95:8-101
 * coeff/rorder/aorder are HOST arrays of length arrsize (they are expanded on the host into the
 * monomial coefficients of ONE bivariate polynomial -- pxf_zernike_monomials -- which the kernel stages
 * in LDS and evaluates by Horner schemes).  aorder<0 -> sin, >0 -> cos, 0 -> radial term.  Orders up to
 * n=15 are supported (PXF_ERANGE beyond). */
int pxf_tracezern(double *x, double *y, double *z, const double *l, const double *m, const double *n,
                  double *ux, double *uy, double *uz, int64_t num, const double *coeff,
                  const int32_t *rorder, const int32_t *aorder, int32_t arrsize, double rad,
                  const uint8_t *mask, pxf_stream_t stream);

/* tracezernOPD(opd,x,...,uz,num,coeff,rorder,aorder,arrsize,rad,nr)   zernsurf.f95:108-203 */
int pxf_tracezernopd(double *opd, double *x, double *y, double *z, const double *l, const double *m,
                     const double *n, double *ux, double *uy, double *uz, int64_t num,
                     const double *coeff, const int32_t *rorder, const int32_t *aorder,
                     int32_t arrsize, double rad, double nr, const uint8_t *mask, pxf_stream_t stream);

/* EXTENSION (no reference counterpart; BASELINE config 4's "Zernike-perturbed conic"): Newton trace to
 *   z = c rho^2 / (1 + sqrt(1 - (K+1) c^2 rho^2)) + sum_i coeff_i Z_i(rho/rad, theta),  c = 1/R,
 * vertex at the origin opening toward +z like `conic`; Zernike orders up to n = 10.  opd may be NULL
 * (then nr is ignored).  Parity unpinned: validated against pxf_conic (coeff = 0) and an independent
 * NumPy Newton (tests/test_gpu_wrappers.py). */
int pxf_tracezernconic(double *opd, double *x, double *y, double *z, const double *l, const double *m,
                       const double *n, double *ux, double *uy, double *uz, int64_t num, double R, double K,
                       const double *coeff, const int32_t *rorder, const int32_t *aorder, int32_t arrsize,
                       double rad, double nr, const uint8_t *mask, pxf_stream_t stream);
/* tracezernrot(x,...,uz,num,coeff1,rorder1,aorder1,arrsize1,coeff2,rorder2,aorder2,arrsize2,rad,rot)
 * zernsurf.f95:257-359: sum of two Zernike sets, the second evaluated at theta+rot (folded into one
 * table on the host by the angle-addition identities). */
int pxf_tracezernrot(double *x, double *y, double *z, const double *l, const double *m, const double *n,
                     double *ux, double *uy, double *uz, int64_t num, const double *coeff1,
                     const int32_t *rorder1, const int32_t *aorder1, int32_t arrsize1, const double *coeff2,
                     const int32_t *rorder2, const int32_t *aorder2, int32_t arrsize2, double rad, double rot,
                     const uint8_t *mask, pxf_stream_t stream);
/* Host utility (no GPU): the bivariate polynomial the Zernike kernels of radial order <= 10 evaluate.
 * sum_i coeff_i Z_i(rho, theta) -- the reference's normalised Zernikes, zernset specialFunctions.f95:142-232
 * -- expanded into monomials of u = x/rad, v = y/rad:  out[j*(degree+1) - j*(j-1)/2 + i] multiplies
 * u^(degree-j-i) v^j  (Horner order; (degree+1)(degree+2)/2 doubles; degree >= the largest rorder, <= 15). */
int pxf_zernike_monomials(const double *coeff, const int32_t *rorder, const int32_t *aorder, int32_t arrsize,
                          int32_t degree, double *out);

/* zernphase(opd,x,...,uz,num,coeff,rorder,aorder,arrsize,rad,wave)   zernsurf.f95:206-250
 * Zernike phase plate in the z=0 plane; only opd,x,y,l,m,n are touched. */
int pxf_zernphase(double *opd, const double *x, const double *y, double *l, double *m, double *n, int64_t num,
                  const double *coeff, const int32_t *rorder, const int32_t *aorder, int32_t arrsize,
                  double rad, double wave, const uint8_t *mask, pxf_stream_t stream);

/* ---- transformationsf.f95 -------------------------------------------------------------- */

/* transform(x,...,uz,num,tx,ty,tz,rx,ry,rz)  transformationsf.f95:134-163
 * p += t, then Rx(rx) Ry(ry) Rz(rz) on position, direction and normal.  (The Python wrapper
 * negates all six parameters before this call, transformations.py:29.)  Every column is READ and every number
 * goes through the reference's arithmetic; a column is REWRITTEN only where a value changed (bit compare), so a
 * pure translation along one axis moves 72 + 8 B per ray instead of 144 -- same results, NaN / -0 propagation
 * through the zero-angle rotations included. */
int pxf_transform(double *x, double *y, double *z, double *l, double *m, double *n, double *ux,
                  double *uy, double *uz, int64_t num, double tx, double ty, double tz, double rx,
                  double ry, double rz, const uint8_t *mask, pxf_stream_t stream);

/* itransform(...)                            transformationsf.f95:168-201 */
int pxf_itransform(double *x, double *y, double *z, double *l, double *m, double *n, double *ux,
                   double *uy, double *uz, int64_t num, double tx, double ty, double tz, double rx,
                   double ry, double rz, const uint8_t *mask, pxf_stream_t stream);

/* reflect(l,m,n,ux,uy,uz,num)                transformationsf.f95:60-79 */
int pxf_reflect(double *l, double *m, double *n, const double *ux, const double *uy,
                const double *uz, int64_t num, const uint8_t *mask, pxf_stream_t stream);

/* radgrat(x,y,l,m,n,wave,num,dpermm,order)   transformationsf.f95:205-238  (scalar wavelength;
 * output sign of n from sign(n); evanescent -> NaN; pi is the reference's fp32 acos(-1.)) */
int pxf_radgrat(const double *x, const double *y, double *l, double *m, double *n, double wave,
                int64_t num, double dpermm, double order, const uint8_t *mask, pxf_stream_t stream);

/* radgratW(x,y,l,m,n,wave,num,dpermm,order)  transformationsf.f95:242-272  (per-ray wavelength,
 * device column; output sign of n from sign(y)) */
int pxf_radgratw(const double *x, const double *y, double *l, double *m, double *n,
                 const double *wave, int64_t num, double dpermm, double order, const uint8_t *mask,
                 pxf_stream_t stream);

/* grat(x,y,l,m,n,num,d,order,wave)   transformationsf.f95:277-305   (linear grating, grooves along
 * y; x,y are unused by the reference loop and not taken here).  Only the product order*wave
 * enters: pass it as a scalar, or per ray as a device column (then the scalar is ignored). */
int pxf_grat(double *l, double *m, double *n, int64_t num, double d, double order_times_wave,
             const double *order_times_wave_col, const uint8_t *mask, pxf_stream_t stream);

/* refract(l,m,n,ux,uy,uz,num,n1,n2)   transformationsf.f95:82-130 */
int pxf_refract(double *l, double *m, double *n, double *ux, double *uy, double *uz, int64_t num, double n1,
                double n2, const uint8_t *mask, pxf_stream_t stream);

/* ---- transformations.py host-side helpers that touch every ray ----------------------------- */

/* applyT(rays,coords,inverse)                transformations.py:257-280
 * Out-of-place: positions through the 4x4 `mpos` (rotation+translation, w=1), direction cosines
 * and normals through the 4x4 `mdir` (rotation only).  Matrices are HOST row-major double[16].
 * Replaces three dense np.dot(4x4, 4xN) products; K=4 is far too thin for MFMA. */
int pxf_applyt(const double *x, const double *y, const double *z, const double *l, const double *m,
               const double *n, const double *ux, const double *uy, const double *uz, double *ox,
               double *oy, double *oz, double *ol, double *om, double *on, double *oux, double *ouy,
               double *ouz, int64_t num, const double *mpos, const double *mdir, pxf_stream_t stream);

/* applyTPos(x,y,z,coords,inverse)            transformations.py:282-289 */
int pxf_applytpos(const double *x, const double *y, const double *z, double *ox, double *oy,
                  double *oz, int64_t num, const double *mpos, pxf_stream_t stream);

/* pointTo(rays,x0,y0,z0,reverse)             transformations.py:91-100 */
int pxf_pointto(const double *x, const double *y, const double *z, double *l, double *m, double *n,
                int64_t num, double x0, double y0, double z0, double reverse, pxf_stream_t stream);

/* vignette(rays,ind)                         transformations.py:214-225
 * Stable stream compaction (wavefront ballot + prefix sums).  Three steps:
 *   pxf_vignette_mask     default predicate  l^2+m^2+n^2 > .1  -> uint8 mask (NaN -> dropped)
 *   pxf_compact_scan      per-tile counts + exclusive scan into `workspace`; total -> *total_dev
 *   pxf_compact_scatter   out[c][rank(i)] = in[c][i] for every kept ray, all columns at once
 * `workspace` is caller-owned device memory of pxf_compact_workspace_bytes(num) bytes.
 * `in_cols` / `out_cols` are HOST arrays of `ncols` (<= 16) device pointers. */
size_t pxf_compact_workspace_bytes(int64_t num);
int pxf_vignette_mask(const double *l, const double *m, const double *n, int64_t num,
                      uint8_t *mask_out, pxf_stream_t stream);
int pxf_compact_scan(const uint8_t *mask, int64_t num, void *workspace, int64_t *total_dev,
                     pxf_stream_t stream);
int pxf_compact_scatter(const double *const *in_cols, double *const *out_cols, int32_t ncols,
                        int64_t num, const uint8_t *mask, const void *workspace, pxf_stream_t stream);
/* rays[i][ind] with an integer index array (any order, repeats allowed, negatives wrap) */
int pxf_gather(const double *const *in_cols, double *const *out_cols, int32_t ncols,
               const int64_t *index, int64_t nidx, int64_t num, pxf_stream_t stream);
/* ind (index array) -> uint8 mask of length num for the in-place `ind=` paths */
int pxf_mask_from_index(const int64_t *index, int64_t nidx, int64_t num, uint8_t *mask_out,
                        pxf_stream_t stream);

/* ---- analyses.py reductions (focal-plane metrics) --------------------------------------------- */

/* One pass, deterministic two-level tree (no atomics).  out_dev[PXF_NMOM] (device) receives
 *   [0] S w   [1] S w x   [2] S w y   [3] S w l/n   [4] S w m/n   [5] S w x l/n   [6] S w y m/n
 *   [7] S w (l/n)^2   [8] S w (m/n)^2   [9] S w l   [10] S w m   [11] S w n
 * from which centroid (analyses.py:16-22), analyticImagePlane/X/Y (:118-155) and steerX/Y
 * (transformations.py:79-89) follow.  w==NULL means unit weights.  l,m,n may be NULL (then
 * only [0..2] are meaningful).  `workspace`: pxf_reduce_workspace_bytes() bytes. */
#define PXF_NMOM 12
size_t pxf_reduce_workspace_bytes(void);
int pxf_moments(const double *x, const double *y, const double *l, const double *m, const double *n,
                const double *w, int64_t num, double *out_dev, void *workspace, pxf_stream_t stream);
/* Second pass about a centre: out_dev[4] = [S w, S w ((x-cx)^2+(y-cy)^2), S w (x-cx)^2, S w (y-cy)^2]
 * -> rmsCentroid / rmsX / rmsY (analyses.py:24-58). */
int pxf_central_moments(const double *x, const double *y, const double *w, int64_t num, double cx,
                        double cy, double *out_dev, void *workspace, pxf_stream_t stream);

/* One step of surfaces.focus / focusX / focusY / focusI (surfaces.py:502-519) in ONE pass over the bundle:
 *     transform(x..uz, tx,ty,tz, rx,ry,rz)   transformationsf.f95:134-163   (the wrapper's negated arguments)
 *     flat(x..uz)  /  flatOPD(x..uz, opd, nr) if opd != NULL                  surfacesf.f95:4-53
 *     moments_dev[PXF_NMOM] = the twelve sums above of the RESULT, weights w (NULL: unit), if moments_dev != NULL
 * Per ray the same arithmetic as the three separate entries, in the same order: rays and sums come out with the same
 * bits as pxf_transform -> pxf_flat[opd] -> pxf_moments.  104 B/ray (96 without the sums) instead of 80 + 96 + 48.
 * `workspace` as pxf_moments (only used with moments_dev).  No mask: focus() has no `ind`. */
int pxf_transform_flat_moments(double *opd, double *x, double *y, double *z, double *l, double *m, double *n,
                               double *ux, double *uy, double *uz, const double *w, int64_t num, double tx, double ty,
                               double tz, double rx, double ry, double rz, double nr, double *moments_dev,
                               void *workspace, pxf_stream_t stream);

/* analyses.rho (analyses.py:60-71): out[i] = sqrt((x-cx)^2 + (y-cy)^2) */
int pxf_rho(const double *x, const double *y, double cx, double cy, double *out, int64_t num,
            pxf_stream_t stream);
/* analyses.measureOPD (analyses.py:231-244): out[i] = |(x,y,z)_i - (px,py,pz)| */
int pxf_dist3(const double *x, const double *y, const double *z, double px, double py, double pz, double *out,
              int64_t num, pxf_stream_t stream);
/* analyses.indAngle / grazeAngle (analyses.py:164-187): out[i] = acos(d.u)  (graze!=0: pi/2 - acos(d.u)) */
int pxf_indangle(const double *l, const double *m, const double *n, const double *ux, const double *uy,
                 const double *uz, double *out, int64_t num, int32_t graze, pxf_stream_t stream);
/* Order statistics without sorting (radix select on the device): out_dev[j] = the k[j]-th smallest
 * (0-based) element of v, for nk <= 2 ranks given as a HOST array; NaN anywhere -> NaN (np.median).
 * analyses.hpd (analyses.py:88-97) = 2 * mean of the two middle radii.  `workspace`:
 * pxf_select_workspace_bytes() bytes of device memory. */
size_t pxf_select_workspace_bytes(void);
int pxf_kth_values(const double *v, int64_t num, const int64_t *k, int32_t nk, double *out_dev,
                   void *workspace, pxf_stream_t stream);

/* The same selection step by step, for a bundle SHARDED across GPUs ("hpd median via
 * histogram all-reduce"): begin (global ranks k[]), then for pass = 0 .. pxf_select_passes()-1:
 * pxf_select_hist on the local shard, SUM the histogram block (and, after pass 0, the NaN counter:
 * both uint64, located by pxf_select_state_layout inside the workspace) over all ranks, then
 * pxf_select_pick.  After the last pick out_dev[j] holds the exact global k[j]-th value on every rank. */
int32_t pxf_select_passes(void);
void pxf_select_state_layout(int64_t *hist_offset, int64_t *hist_bytes, int64_t *nan_offset);
int pxf_select_begin(const int64_t *k, int32_t nk, void *workspace, pxf_stream_t stream);
int pxf_select_hist(const double *v, int64_t num, int32_t nk, void *workspace, int32_t pass, pxf_stream_t stream);
int pxf_select_pick(int32_t nk, void *workspace, int32_t pass, double *out_dev, pxf_stream_t stream);

/* Order statistics of the RADII rho_i = sqrt((x_i-cx)^2+(y_i-cy)^2) (mode 0) or |y_i-cy| (mode 1)
 * straight from x, y in one pass over the rays (analyses.hpd / hpdY = twice their median,
 * analyses.py:88-116): a strided
 * sample brackets the wanted ranks, one pass counts the radii below the bracket and collects those
 * inside it, the exact k[j]-th values are selected among the collected ones.  *ok_dev = 1 when the
 * bracket held (then out_dev[j] is exactly the k[j]-th smallest radius; NaN if any radius is NaN),
 * 0 when it did not (many tied radii, or two ranks far apart: ONE bracket spans both): fall back to
 * pxf_rho + pxf_kth_values.  16 B/ray instead of 72.
 * `workspace`: pxf_radial_kth_workspace_bytes(num) bytes of device memory. */
size_t pxf_radial_kth_workspace_bytes(int64_t num);
int pxf_radial_kth(const double *x, const double *y, double cx, double cy, int32_t mode, int64_t num,
                   const int64_t *k, int32_t nk, double *out_dev, int32_t *ok_dev, void *workspace,
                   pxf_stream_t stream);

/* Weighted quantiles without a sort (analyses.hpd with weights, analyses.py:73-97): out_dev[j] =
 * r[argmin |cdf - q[j]|] of the reference (cdf = cumsum of the weights in order of increasing v,
 * normalised), for nq <= 2 quantiles q[] (HOST array, 0 < q <= 1) and positive weights w.  Weighted
 * radix descent + one neighbour pass: 7 reads of v and 6 of w.  `workspace`:
 * pxf_wselect_workspace_bytes().  The step-wise entries serve a bundle sharded across GPUs: after
 * every pxf_wselect_hist SUM the fp64 histogram block over ranks; after pxf_wselect_neighbors take
 * the MAX of the int64 `prev` pair and the SUM of the uint64 `count` pair (pxf_wselect_state_layout);
 * then pxf_wselect_finish. */
size_t pxf_wselect_workspace_bytes(void);
int pxf_wquantiles(const double *v, const double *w, int64_t num, const double *q, int32_t nq, double *out_dev,
                   void *workspace, pxf_stream_t stream);
void pxf_wselect_state_layout(int64_t *hist_offset, int64_t *hist_bytes, int64_t *prev_offset, int64_t *count_offset);
int pxf_wselect_begin(const double *q, int32_t nq, void *workspace, pxf_stream_t stream);
int pxf_wselect_hist(const double *v, const double *w, int64_t num, int32_t nq, void *workspace, int32_t pass,
                     pxf_stream_t stream);
int pxf_wselect_pick(int32_t nq, void *workspace, int32_t pass, pxf_stream_t stream);
int pxf_wselect_neighbors(const double *v, int64_t num, int32_t nq, void *workspace, pxf_stream_t stream);
int pxf_wselect_finish(int32_t nq, void *workspace, double *out_dev, pxf_stream_t stream);

/* ---- wavefront sensing: reconstruct.f95 ---------------------------------------------------------
 * Grids are Fortran-ordered like the reference's arrays: a(xi,yi) = a[xi + xdim*yi] (device memory).
 *
 * southwellbin(x,y,l,m,num,binsize,xang,yang,phase,xdim,ydim)               reconstruct.f95:139-187
 * Bins l, m by ray position on an xdim x ydim lenslet grid of pitch `binsize`; per bin the mean
 * cosine becomes a slope tan(asin(mean)); empty bins get 100. in xang, yang AND phase (other phase
 * cells are set to 0).  Rays outside the grid are dropped (the reference indexes out of bounds).
 * Sums are accumulated with atomics (order-nondeterministic, like the reference's OpenMP loop).
 * `workspace`: pxf_southwellbin_workspace_bytes(xdim, ydim) bytes of device memory. */
size_t pxf_southwellbin_workspace_bytes(int32_t xdim, int32_t ydim);
int pxf_southwellbin(const double *x, const double *y, const double *l, const double *m, int64_t num,
                     double binsize, double *xang, double *yang, double *phase, int32_t xdim, int32_t ydim,
                     void *workspace, pxf_stream_t stream);
/* reconstruct(xang,yang,xdim,ydim,criteria,h,phase,phasec,maxiter)           reconstruct.f95:1-131
 * Southwell SOR in the reference's Gauss-Seidel order (executed by anti-diagonals: identical
 * updates).  xang, yang, phase are updated in place as in the reference (isolated lenslets are set
 * to 100.; phase ends equal to phasec); phasec receives the reconstructed phase.  `sweeps_dev`
 * (device int32, may be NULL) receives the number of sweeps performed. */
int pxf_reconstruct(double *xang, double *yang, int32_t xdim, int32_t ydim, double criteria, double h,
                    double *phase, double *phasec, int32_t maxiter, int32_t *sweeps_dev, pxf_stream_t stream);

/* ---- sources.py on the device ------------------------------------------------------------------ */

/* Counter-based Philox4x32-10 keyed on (seed, first_index + i): the bundle is a pure function of
 * the GLOBAL ray index, so a sharded run reproduces the single-GPU bundle exactly.  (The
 * reference draws from NumPy's global MT19937 state, sources.py:40-41; the Python layer keeps
 * that bit-exact host path as the default and offers this one for bundles too large for it.)
 * cols = the ten columns in bundle order opd,x,y,z,l,m,n,ux,uy,uz (HOST array of device ptrs). */
#define PXF_SRC_POINTSOURCE 0  /* p0 = ang                         sources.py:20-53   */
#define PXF_SRC_CIRCULARBEAM 1 /* p0 = rad                         sources.py:56-88   */
#define PXF_SRC_ANNULUS 2      /* p0 = rin, p1 = rout, p3 = zhat   sources.py:91-127  */
#define PXF_SRC_SUBANNULUS 3   /* p0 = rin, p1 = rout, p2 = dphi, p3 = zhat  sources.py:130-170 */
#define PXF_SRC_RECTBEAM 4     /* p0 = xhalfwidth, p1 = yhalfwidth sources.py:348-377 */
/* the remaining sources.py generators; parameters in a HOST double[7] (pxf_source_ex) */
#define PXF_SRC_XSLIT 5            /* xin, xout, -, zhat                       sources.py:173-207  (grid) */
#define PXF_SRC_RECTARRAY 6        /* xsize, ysize, n per side                 sources.py:210-247  (grid) */
#define PXF_SRC_CONVERGINGBEAM 7   /* zset, rin, rout, tmin, tmax, lscat       sources.py:250-297  */
#define PXF_SRC_CONVERGINGBEAM2 8  /* zset, xmin, xmax, ymin, ymax, lscat      sources.py:300-345  */
#define PXF_SRC_GAUSSIANBEAM 9     /* ang                                      sources.py:380-413  */
#define PXF_SRC_FANBEAM 10         /* xang, yang, n per side                   sources.py:416-438  (grid) */
#define PXF_SRC_CIRCFAN 11         /* halfang, rings, arms                     sources.py:441-471  (grid) */
int pxf_source(int32_t kind, double *const *cols, int64_t num, double p0, double p1, double p2,
               double p3, uint64_t seed, uint64_t first_index, pxf_stream_t stream);

/* Extended kinds 5..11 (sources.py:173-345, 380-471: xslit, rectArray, convergingbeam[2], gaussianBeam,
 * fanBeam, circFan).  `num` rays are generated for global indices first_index .. first_index+num-1
 * of a bundle of `total` rays (the grid kinds need the full length: np.linspace(.., total)); the grid
 * kinds reproduce NumPy's linspace/meshgrid positions bit for bit (direction cosines to an ulp of sin/cos), the random ones are distribution-equal. */
int pxf_source_ex(int32_t kind, double *const *cols, int64_t num, int64_t total, const double *params,
                  uint64_t seed, uint64_t first_index, pxf_stream_t stream);

/* ---- fused ray program ---------------------------------------------------------------------------
 * The reference makes one pass over the bundle per API call (576 B/ray of HBM traffic for the
 * two-bounce chain).  pxf_program_run executes a SEQUENCE of the per-ray operations above in one
 * launch: every column is loaded at most once, all operations run on the ray in registers, every
 * modified column is stored once (120 B/ray for the same chain).  The per-ray arithmetic is the
 * same code as the stand-alone entry points, so results are bit-identical to calling them one by
 * one.  Operations whose result is overwritten before anything reads it are dropped (e.g. the
 * rotation of the surface normal by a `transform` that precedes a surface intersection).
 *
 * `ops` is a HOST array.  `a[]` holds the trailing scalar arguments of the corresponding entry
 * point, in its order (e.g. PXF_OP_TRANSFORM: tx,ty,tz,rx,ry,rz; PXF_OP_WOLTERPRIMARYOPD:
 * r0,z0,psi,nr; PXF_OP_RADGRAT: wave,dpermm,order; PXF_OP_RADGRATW: dpermm,order with `aux` = the
 * device wavelength column).  `mask` as for the stand-alone call (NULL = all rays).
 * `cols` = HOST array of the ten device column pointers in bundle order opd,x,y,z,l,m,n,ux,uy,uz;
 * columns no operation touches may be NULL. */
#define PXF_OP_FLAT 1
#define PXF_OP_FLATOPD 2
#define PXF_OP_CONIC 3
#define PXF_OP_CONICOPD 4
#define PXF_OP_WOLTERPRIMARY 5
#define PXF_OP_WOLTERPRIMARYOPD 6
#define PXF_OP_WOLTERSECONDARY 7
#define PXF_OP_TRANSFORM 8
#define PXF_OP_ITRANSFORM 9
#define PXF_OP_REFLECT 10
#define PXF_OP_RADGRAT 11
#define PXF_OP_RADGRATW 12
#define PXF_OP_SPOCONE 13      /* a: R0, tg */
#define PXF_OP_SPHERE 14       /* a: rad */
#define PXF_OP_SPHEREOPD 15    /* a: rad, nr */
#define PXF_OP_CYL 16          /* a: rad */
#define PXF_OP_CYLOPD 17       /* a: rad, nr */
#define PXF_OP_CYLCONIC 18     /* a: rad, k */
#define PXF_OP_PARAXIAL 19     /* a: F */
#define PXF_OP_PARAXIALY 20    /* a: F */
#define PXF_OP_GRAT 21         /* a: d, order*wave ; aux (optional) = per-ray order*wave column */
#define PXF_OP_REFRACT 22      /* a: n1, n2 */
#define PXF_OP_TRACEZERN 23    /* a: rad      ; aux = (const double *)&pxf_zern_args_t on the HOST */
#define PXF_OP_INDANGLE 25     /* a: graze (0: acos(d.u), 1: pi/2 - acos(d.u)) ; aux = OUTPUT column (analyses.py:164-187): */
                               /* the angle of every ray as it stands at that point of the program, e.g. the two  */
                               /* graze angles behind the reflectivity weights of sector.py:592-612               */
#define PXF_OP_TRACEZERNOPD 24 /* a: rad, nr  ; aux as above.  At most ONE Zernike operation per program, */
                               /* such a program holds at most PXF_PROGRAM_MAX_OPS_ZERN operations, unmasked, out of FLAT[OPD], */
                               /* [I]TRANSFORM, REFLECT (PXF_ERANGE otherwise: run the trace as its own launch) */
/* Tolerance-parity MIRRORS inside a program (woltsurf.f95:167-215, 220-379, 387-588, 643-933): a program that
 * holds any of them may hold at most PXF_PROGRAM_MAX_OPS_MIRROR operations, unmasked, out of these codes and
 * FLAT[OPD], [I]TRANSFORM, REFLECT (PXF_ERANGE otherwise: run the mirror as its own launch); the Legendre-Legendre
 * kinds take azimuthal orders <= 11, at most 15 axial orders each, and all their matrices -- zero-padded to a
 * common multiple-of-four column count and an even row count -- must fit 288 cells (two 11 x 11 sets do). */
#define PXF_OP_WSPRIMARY 26        /* a: alpha, z0, psi */
#define PXF_OP_WSSECONDARY 27      /* a: alpha, z0, psi */
#define PXF_OP_WSPRIMARYBACK 28    /* a: alpha, z0, psi, thick */
#define PXF_OP_WSSECONDARYBACK 29  /* a: alpha, z0, psi, thick */
#define PXF_OP_WOLTERSINE 30       /* a: r0, z0, amp, freq */
#define PXF_OP_WOLTERPRIMLL 31     /* a: r0, z0, zmax, zmin, dphi       ; aux = (const double *)&pxf_ll_args_t on the HOST */
#define PXF_OP_WOLTERSECLL 32      /* a: r0, z0, psi, zmax, zmin, dphi  ; aux as above */
#define PXF_OP_ELLIPSOIDWOLTLL 33  /* a: r0, z0, psi, zmax, zmin, dphi  ; aux as above (its S field) */
#define PXF_PROGRAM_MAX_OPS 32
#define PXF_PROGRAM_MAX_OPS_ZERN 16
#define PXF_PROGRAM_MAX_OPS_MIRROR 12
#define PXF_PROGRAM_ZERN_MAX_ORDER 15 /* highest radial order of a FUSED Zernike trace (= of any Zernike trace) */

/* host-side coefficient table of a PXF_OP_TRACEZERN[OPD] operation (the arguments of pxf_tracezern) */
typedef struct pxf_zern_args {
    const double *coeff;   /* HOST arrays of `arrsize` entries */
    const int32_t *rorder;
    const int32_t *aorder;
    int32_t arrsize;
} pxf_zern_args_t;

/* host-side coefficient list of a Legendre-Legendre operation (the trailing arguments of pxf_wolterprimll ...) */
typedef struct pxf_ll_args {
    const double *coeff;   /* HOST arrays of `cnum` entries */
    const int32_t *axial;
    const int32_t *az;
    int32_t cnum;
    double S;              /* PXF_OP_ELLIPSOIDWOLTLL only */
} pxf_ll_args_t;

typedef struct pxf_op {
    int32_t code;        /* PXF_OP_* */
    int32_t reserved;    /* must be 0 */
    const uint8_t *mask; /* device uint8[num] or NULL */
    const double *aux;   /* device float64[num] extra column (RADGRATW wavelengths), NULL, or see TRACEZERN */
    double a[6];         /* scalar arguments */
} pxf_op_t;

int pxf_program_run(const pxf_op_t *ops, int32_t nops, double *const *cols, int64_t num,
                    pxf_stream_t stream);
/* What the planner decided for a program (for tests / reports): columns loaded, columns stored
 * (bit c = column c in bundle order) and the number of operation groups dropped as dead. */
int pxf_program_plan(const pxf_op_t *ops, int32_t nops, uint32_t *load_mask, uint32_t *store_mask,
                     int32_t *dropped);

/* ---- segmented bundles (multi-shell optics in one launch) ------------------------------------------
 * A bundle made of `nseg` contiguous segments of `rays_per_seg` rays (mirror shells, modules, ...)
 * that run the SAME operation sequence with DIFFERENT parameters.  The reference loops over shells
 * in Python (examples/arcus/sector.py:556-626); here the per-segment parameter blocks sit in a
 * device table, each workgroup stages the blocks of the segments its rays belong to in LDS, and the
 * whole bundle is traced by one launch.
 *   pxf_program_params        host: the internal parameter block (9 doubles per LIVE op, in plan
 *                             order) the planner derives for one op list -> build the table by calling
 *                             it once per segment; *nlive = number of live ops (same for every segment)
 *   pxf_program_run_segments  `ops` gives the sequence (codes, masks, aux; its a[] is ignored),
 *                             seg_params_dev = device double[nseg][nlive][9]; num == nseg*rays_per_seg,
 *                             rays_per_seg even and >= 172; columns 16-byte aligned
 *   pxf_source_segments       device source whose (p0..p3) come per segment from device double[nseg][4] */
int pxf_program_params(const pxf_op_t *ops, int32_t nops, double *params_out, int32_t *nlive);
int pxf_program_run_segments(const pxf_op_t *ops, int32_t nops, const double *seg_params_dev, int32_t nseg,
                             int64_t rays_per_seg, double *const *cols, int64_t num, pxf_stream_t stream);
int pxf_source_segments(int32_t kind, double *const *cols, int64_t num, const double *seg_params_dev,
                        int32_t nseg, int64_t rays_per_seg, uint64_t seed, uint64_t first_index,
                        pxf_stream_t stream);

/* ---- a fanned array of radial gratings, per ray (EXTENSION: no reference function) --------------------------
 * The `while` loop of the reference's examples/arcus/sector.py:663-704 (`gratArray`): a stack of radial gratings
 * fanned about a common hub, walked round by round -- rotate the frame of ALL rays by one grating pitch
 * (transform(0,0,0,ang,0,0)), flat() the rays that have not met a grating yet and are not grazing
 * (|asin n| > thr), test  hubdist < -sqrt(x^2+y^2) sign(y) < hubdist+length, diffract the rays that passed
 * (reflect, radgrat) at the start of the next round -- until every ray has met a grating.  The reference does
 * this with a host loop over the whole bundle; here every ray walks the fan in registers with the same
 * TransformOp / FlatOp / ReflectOp / RadgratOp code, bit-identical to the loop:
 *   pxf_fan_search  rounds until each ray meets a grating: it_hit[i] = its round (-1: none within `cap`),
 *                   *imax = the largest round (= the host loop's round count), *nmiss = rays without a hit
 *   pxf_fan_finish  diffracts every ray at the start of round it_hit+1 and keeps rotating it up to round imax
 * cols[1..9] = x y z l m n ux uy uz (cols[0] unused); it_hit, imax, nmiss are DEVICE buffers. */
int pxf_fan_search(double *const *cols, int64_t num, double hubdist, double length, double ang, double thr, int32_t cap,
                   int32_t *it_hit, int32_t *imax, uint64_t *nmiss, pxf_stream_t stream);
int pxf_fan_finish(double *const *cols, int64_t num, double ang, double dpermm, double order, double wave,
                   const int32_t *it_hit, int32_t imax, pxf_stream_t stream);

/* ---- analyses.interpolateVec on the device (analyses.py:189-230; scipy.interpolate.griddata 'nearest' / 'linear') ----
 * Values v of n scattered points (x, y) interpolated at m query points: method 0 = the value of the nearest point,
 * 1 = barycentric interpolation inside the DELAUNAY triangle that contains the query (found per query by a simplex
 * walk, csrc/pxf_interp.hip; no triangulation is stored), NaN outside the convex hull of the points -- what
 * scipy's griddata returns, to the rounding of the barycentric weights, for points in general position.
 * The caller supplies the points SORTED BY CELL of a uniform gx x gy grid over their bounding box -- cell (i, j) =
 * [x0 + i / inv_cell_x, ...) x [y0 + j / inv_cell_y, ...), id j * gx + i -- and cell_start[gx * gy + 1] (device int32,
 * offsets of each cell's points); about two points per cell is a good density.  support (HOST double[64], may be NULL):
 * max_i (x_i cos a_k + y_i sin a_k), a_k = 2 pi k / 64 -- a cheap "outside the hull" test for queries beyond the cloud.
 * All other pointers are device pointers; out[m]; workspace: pxf_griddata_workspace_bytes(m) bytes (the queue of the
 * nodes at the rim of the cloud, which a second kernel settles with one workgroup each). */
size_t pxf_griddata_workspace_bytes(int64_t m);
int pxf_griddata(const double *xs, const double *ys, const double *vs, int64_t n, const int32_t *cell_start, int32_t gx,
                 int32_t gy, double x0, double y0, double inv_cell_x, double inv_cell_y, const double *support,
                 const double *qx, const double *qy, int64_t m, int32_t method, double *out, void *workspace,
                 pxf_stream_t stream);

/* ---- OPT-IN tolerance mode of the fused program ------------------------------------------------------
 * pxf_program_run / pxf_program_run_segments are bit-identical to the reference's loops
 * (surfacesf.f95 / woltsurf.f95 / transformationsf.f95) and, for Newton-heavy programs, bound by IEEE
 * fp64 divides instead of HBM.  The _fast entries run the same program to a TOLERANCE instead: the
 * Wolter-I Newton loops iterate the exact quadratic F(s) along the ray (same steps, same stopping
 * rule, same stale-gradient normal as woltsurf.f95:28-47,134-154), divides and square roots are
 * reciprocal / rsqrt refinements, FMA contraction is on.  Agreement with the reference (the figures
 * are asserted in tests/test_gpu_fused.py): the all-ray RMS position error is ~1e-7 mm -- NOT under the 1e-10 mm
 * bar -- because about 2.5 rays per million have the fp32-rounded propagation distance of a `flat`
 * (surfacesf.f95:4-15) fall on the other side and land one fp32 quantum of their path (2.6e-5 mm at 8.4 m) aside;
 * with those fp32-flat flips set aside it is 5.5e-11 (two-bounce) / 7.5e-11 mm (two-bounce + grating) RMS, the
 * size of the reference's own rounding jitter under a 1-ulp change of an input -- for focal lengths z0 <= 12 m;
 * the jitter grows as z0^2 (1-6e-10 mm at 15-20 m).
 * Supported operations: FLAT, TRANSFORM, ITRANSFORM, REFLECT, WOLTERPRIMARY, WOLTERSECONDARY,
 * RADGRAT, SPOCONE, INDANGLE (aux = the OUTPUT column), unmasked; anything else returns PXF_ERANGE and nothing is
 * launched (run the exact entry instead).  Never selected implicitly. */
int pxf_program_run_fast(const pxf_op_t *ops, int32_t nops, double *const *cols, int64_t num, pxf_stream_t stream);
int pxf_program_run_segments_fast(const pxf_op_t *ops, int32_t nops, const double *seg_params_dev, int32_t nseg,
                                  int64_t rays_per_seg, double *const *cols, int64_t num, pxf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PXF_H */
