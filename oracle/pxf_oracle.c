/*
 * pxf_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the PyXFocus per-ray loops that make up the hot path
 * (SURVEY.md section 8a).  Every function follows the reference's floating-point operation
 * ORDER and reproduces its numerical quirks, so that it is bit-identical to the reference
 * Fortran compiled for x86-64 (no FMA, glibc libm).  It is pinned against
 *   (1) oracle/_ref  = the reference's own .f95 files compiled in place with flang
 *       (tests/test_oracle_vs_ref.py, runs wherever oracle/_ref exists), and
 *   (2) tests/golden/*.npz = input/output vectors produced by (1) with
 *       oracle/make_golden.py and committed.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library -- as the checker / the reported CPU baseline, never as a compute fallback.
 *
 * Build:  gcc -O2 -std=c11 -ffp-contract=off -fPIC -shared pxf_oracle.c -lm
 *         (-ffp-contract=off is REQUIRED: the reference binary has no fused multiply-add).
 *
 * Citations are file:line in the reference tree (/root/reference).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

/* cos(theta), sin(theta) of ONE angle as the reference BINARY forms them: flang (like gfortran / gcc) merges the pair
 * of calls in rotatevector (transformationsf.f95:10-22), spoCone (woltsurf.f95:605-606) ... into one call of the C
 * library's `sincos`, which in glibc 2.35 is the non-FMA build of the algorithm and differs from sin() / cos() in
 * the last bit for about one argument in a thousand.  The restatement calls it EXPLICITLY, so that it does not
 * depend on what this file's compiler happens to merge: round 2 found orc_itransform's `cos(-rx), sin(-rx)` folded to
 * cos(rx), -sin(rx) -- two separate calls, no sincos -- and therefore an ulp off the reference for such angles
 * (tests/test_oracle.py::test_port_matches_reference_where_sincos_differs). */
extern void sincos(double, double *, double *);
static inline void ref_cos_sin(double theta, double *c, double *s) { sincos(theta, s, c); }

/* Fortran single-precision literals widened to double. */
#define TOL_1EM8 ((double)1.e-8f)   /* woltsurf.f95:30,136  "1.e-8"  */
#define TOL_1EM10 ((double)1.e-10f) /* woltsurf.f95:83, zernsurf.f95:31 "1.e-10" */

/* The reference has no iteration cap on the Wolter-I / Zernike Newton loops (it would hang).
 * The oracle carries the same safety cap as the HIP kernels; it never triggers on inputs the
 * reference terminates on in fewer iterations. */
#define ORC_NEWTON_CAP 1000

/* ------------------------------------------------------------------------------------ */
/* Van Speybroeck constants, woltsurf.f95:18-23 (same in :71-76, :125-130).              */
void orc_wolter_params(double r0, double z0, double psi, double *p, double *d, double *e)
{
    double alpha = .25 * atan(r0 / z0);
    double thetah = 2 * (1 + 2 * psi) / (1 + psi) * alpha;
    double thetap = 2 * psi / (1 + psi) * alpha;
    *p = z0 * tan(4 * alpha) * tan(thetap);
    *d = z0 * tan(4 * alpha) * tan(4 * alpha - thetah);
    *e = cos(4 * alpha) * (1 + tan(4 * alpha) * tan(thetah));
}

/* ------------------------------------------------------------------------------------ */
/* surfacesf.f95:4-29.  `delta` is implicitly REAL*4 (no `implicit none`).               */
void orc_flat(double *x, double *y, double *z, const double *l, const double *m, const double *n,
              double *ux, double *uy, double *uz, int64_t num)
{
#pragma omp parallel for
    for (int64_t i = 0; i < num; i++) {
        float delta = (float)(-z[i] / n[i]);
        z[i] = 0.;
        x[i] = x[i] + (double)delta * l[i];
        y[i] = y[i] + (double)delta * m[i];
        ux[i] = 0.;
        uy[i] = 0.;
        uz[i] = 1.;
    }
}

/* surfacesf.f95:32-53 */
void orc_flatopd(double *x, double *y, double *z, const double *l, const double *m, const double *n,
                 double *ux, double *uy, double *uz, double *opd, int64_t num, double nr)
{
#pragma omp parallel for
    for (int64_t i = 0; i < num; i++) {
        float delta = (float)(-z[i] / n[i]);
        z[i] = 0.;
        x[i] = x[i] + (double)delta * l[i];
        y[i] = y[i] + (double)delta * m[i];
        ux[i] = 0.;
        uy[i] = 0.;
        uz[i] = 1.;
        opd[i] = opd[i] + (double)delta * nr;
    }
}

/* ------------------------------------------------------------------------------------ */
/* surfacesf.f95:302-360 (opd==NULL) and :366-420 (opd!=NULL).  The two differ in the
 * order of the terms of `c` (:322 vs :386) and in the OPD update (:398).                */
static void conic_core(double *opd, double *x, double *y, double *z, double *l, double *m, double *n,
                       double *ux, double *uy, double *uz, int64_t num, double R, double K, double nr)
{
#pragma omp parallel for
    for (int64_t i = 0; i < num; i++) {
        double s = 0., denom, b, c, disc, s1, s2;
        if (K == -1 && fabs(n[i]) == 1.) {
            s = (x[i] * x[i] + y[i] * y[i] - 2 * R * z[i]) / (2 * R * n[i]);
        } else {
            denom = l[i] * l[i] + m[i] * m[i] + (K + 1) * (n[i] * n[i]);
            b = x[i] * l[i] + y[i] * m[i] + ((K + 1) * z[i] - R) * n[i];
            b = b / denom;
            if (opd)
                c = x[i] * x[i] + y[i] * y[i] + (K + 1) * (z[i] * z[i]) - 2 * R * z[i];
            else
                c = x[i] * x[i] + y[i] * y[i] - 2 * R * z[i] + (K + 1) * (z[i] * z[i]);
            c = c / denom;
            disc = b * b - c;
            if (disc >= 0.) {
                s1 = -b + sqrt(disc);
                s2 = -b - sqrt(disc);
                if (fabs(s1) <= fabs(s2))
                    s = s1;
                else
                    s = s2;
            }
        }
        if (s == 0.) {
            l[i] = 0.;
            m[i] = 0.;
            n[i] = 0.;
        } else {
            x[i] = x[i] + l[i] * s;
            y[i] = y[i] + m[i] * s;
            z[i] = z[i] + n[i] * s;
            if (opd)
                opd[i] = opd[i] + s * nr;
            double r2 = x[i] * x[i] + y[i] * y[i];
            denom = sqrt(R * R - K * r2);
            ux[i] = -x[i] / denom;
            uy[i] = -y[i] / denom;
            uz[i] = -(R / fabs(R) * sqrt(R * R - (K + 1) * r2));
            uz[i] = -uz[i] / denom;
        }
    }
}

void orc_conic(double *x, double *y, double *z, double *l, double *m, double *n, double *ux,
               double *uy, double *uz, int64_t num, double R, double K)
{
    conic_core(NULL, x, y, z, l, m, n, ux, uy, uz, num, R, K, 0.);
}

void orc_conicopd(double *opd, double *x, double *y, double *z, double *l, double *m, double *n,
                  double *ux, double *uy, double *uz, int64_t num, double R, double K, double nr)
{
    conic_core(opd, x, y, z, l, m, n, ux, uy, uz, num, R, K, nr);
}

/* ------------------------------------------------------------------------------------ */
/* woltsurf.f95:7-54 (opd==NULL, tol 1.e-8) and :60-108 (opd, tol 1.e-10).
 * Normal is built from the gradient evaluated BEFORE the last Newton step (:44-47).     */
static void wolterprimary_core(double *opd, double *x, double *y, double *z, const double *l,
                               const double *m, const double *n, double *ux, double *uy, double *uz,
                               int64_t num, double r0, double z0, double psi, double nr, int64_t *iters)
{
    double p, d, e;
    orc_wolter_params(r0, z0, psi, &p, &d, &e);
    const double Fz = 2 * p;
    const double tol = opd ? TOL_1EM10 : TOL_1EM8;
    int64_t itsum = 0;
#pragma omp parallel for reduction(+ : itsum)
    for (int64_t i = 0; i < num; i++) {
        double delt = 100., F, Fx = 0., Fy = 0., Fp;
        int it = 0;
        while (fabs(delt) > tol && it < ORC_NEWTON_CAP) {
            F = 2 * p * z[i] + p * p + 4 * (e * e) * p * d / (e * e - 1) - x[i] * x[i] - y[i] * y[i];
            Fx = -2. * x[i];
            Fy = -2. * y[i];
            Fp = Fx * l[i] + Fy * m[i] + Fz * n[i];
            delt = -F / Fp;
            x[i] = x[i] + l[i] * delt;
            y[i] = y[i] + m[i] * delt;
            z[i] = z[i] + n[i] * delt;
            if (opd)
                opd[i] = opd[i] + nr * delt;
            it++;
        }
        itsum += it;
        Fp = sqrt(Fx * Fx + Fy * Fy + Fz * Fz);
        ux[i] = Fx / Fp;
        uy[i] = Fy / Fp;
        uz[i] = Fz / Fp;
    }
    if (iters)
        *iters = itsum;
}

void orc_wolterprimary(double *x, double *y, double *z, const double *l, const double *m,
                       const double *n, double *ux, double *uy, double *uz, int64_t num, double r0,
                       double z0, double psi)
{
    wolterprimary_core(NULL, x, y, z, l, m, n, ux, uy, uz, num, r0, z0, psi, 0., NULL);
}

void orc_wolterprimaryopd(double *opd, double *x, double *y, double *z, const double *l,
                          const double *m, const double *n, double *ux, double *uy, double *uz,
                          int64_t num, double r0, double z0, double psi, double nr)
{
    wolterprimary_core(opd, x, y, z, l, m, n, ux, uy, uz, num, r0, z0, psi, nr, NULL);
}

/* woltsurf.f95:114-161 */
void orc_woltersecondary(double *x, double *y, double *z, const double *l, const double *m,
                         const double *n, double *ux, double *uy, double *uz, int64_t num, double r0,
                         double z0, double psi)
{
    double p, d, e;
    orc_wolter_params(r0, z0, psi, &p, &d, &e);
    (void)p;
#pragma omp parallel for
    for (int64_t i = 0; i < num; i++) {
        double delt = 100., F, Fx = 0., Fy = 0., Fz = 0., Fp;
        int it = 0;
        while (fabs(delt) > TOL_1EM8 && it < ORC_NEWTON_CAP) {
            double dz = d + z[i];
            F = (e * e) * (dz * dz) - z[i] * z[i] - x[i] * x[i] - y[i] * y[i];
            Fx = -2. * x[i];
            Fy = -2. * y[i];
            Fz = 2 * (e * e) * dz - 2 * z[i];
            Fp = Fx * l[i] + Fy * m[i] + Fz * n[i];
            delt = -F / Fp;
            x[i] = x[i] + l[i] * delt;
            y[i] = y[i] + m[i] * delt;
            z[i] = z[i] + n[i] * delt;
            it++;
        }
        Fp = sqrt(Fx * Fx + Fy * Fy + Fz * Fz);
        ux[i] = Fx / Fp;
        uy[i] = Fy / Fp;
        uz[i] = Fz / Fp;
    }
}

/* ------------------------------------------------------------------------------------ */
/* specialFunctions.f95:142-232.  rnm/rprime are radnum x radnum work tables; the reference
 * heap-allocates them per call (:165-166), the oracle takes caller scratch of the same shape.
 * Index (n, m) <-> rnm[n*radnum + m].  `radnum` is derived from znum exactly as :156-161.  */
static int zern_radnum(int znum)
{
    int tznum = 1, radnum = 1;
    while (tznum < znum) {
        tznum = tznum + (radnum + 1);
        radnum = radnum + 1;
    }
    return radnum;
}

static void zernset(double rho, double theta, const int32_t *rorder, const int32_t *aorder, int znum,
                    double *polyout, double *derrho, double *dertheta, double *rnm, double *rprime,
                    int radnum)
{
#define RNM(i, j) rnm[((i)-1) * radnum + ((j)-1)]
#define RPR(i, j) rprime[((i)-1) * radnum + ((j)-1)]
    for (int i = 1; i <= radnum; i++) {
        double n = (double)i - 1;
        for (int j = (int)n + 1; j >= 1; j -= 2) {
            double m = (double)j - 1;
            if (rho == 0) {
                if (m == 1) {
                    RPR(i, j) = pow(-1., (n - 1) / 2) * (n + 1) / 2;
                    RNM(i, j) = 0.;
                } else if (m == 0) {
                    RPR(i, j) = 0.;
                    RNM(i, j) = pow(-1., n / 2);
                } else {
                    RPR(i, j) = 0.;
                    RNM(i, j) = 0.;
                }
            } else {
                if (n == m) {
                    RNM(i, j) = pow(rho, n);
                    RPR(i, j) = n * pow(rho, n - 1);
                } else if (m == n - 2) {
                    RNM(i, j) = n * RNM(i, i) - (n - 1) * RNM(i - 2, i - 2);
                    RPR(i, j) = n * RPR(i, i) - (n - 1) * RPR(i - 2, i - 2);
                } else {
                    double h3 = -(4 * (m + 2) * (m + 1) / (n + m + 2) / (n - m));
                    double h2 = h3 * (n + m + 4) * (n - m - 2) / 4. / (m + 3) + (m + 2);
                    double h1 = .5 * (m + 4) * (m + 3) - (m + 4) * h2 + h3 * (n + m + 6) * (n - m - 4) / 8.;
                    double rho2 = rho * rho;
                    double rho3 = rho * rho * rho;
                    RNM(i, j) = h1 * RNM(i, j + 4) + (h2 + h3 / rho2) * RNM(i, j + 2);
                    RPR(i, j) = h1 * RPR(i, j + 4) + (h2 + h3 / rho2) * RPR(i, j + 2) -
                                2 * h3 / rho3 * RNM(i, j + 2);
                }
            }
        }
    }
    /* :208-230.  sqrt(0.5) is a single-precision expression in the reference. */
    const double sqrt_half_f32 = (double)sqrtf(0.5f);
    for (int i = 0; i < znum; i++) {
        double n = rorder[i];
        double mm = aorder[i];
        double m = fabs(mm);
        double norm = sqrt(2 * (n + 1));
        double r = RNM((int)n + 1, (int)m + 1);
        double rp = RPR((int)n + 1, (int)m + 1);
        if (mm < 0) {
            polyout[i] = norm * r * sin(m * theta);
            derrho[i] = norm * rp * sin(m * theta);
            dertheta[i] = norm * r * cos(m * theta) * m;
        } else if (mm > 0) {
            polyout[i] = norm * r * cos(m * theta);
            derrho[i] = norm * rp * cos(m * theta);
            dertheta[i] = -norm * r * sin(m * theta) * m;
        } else {
            polyout[i] = norm * sqrt_half_f32 * r;
            derrho[i] = norm * sqrt_half_f32 * rp;
            dertheta[i] = 0.;
        }
    }
#undef RNM
#undef RPR
}

/* Exposed for the unit test that pins zernset itself against the compiled reference. */
void orc_zernset(double rho, double theta, const int32_t *rorder, const int32_t *aorder, int32_t znum,
                 double *polyout, double *derrho, double *dertheta)
{
    int radnum = zern_radnum(znum);
    double *rnm = (double *)calloc((size_t)radnum * radnum, sizeof(double));
    double *rpr = (double *)calloc((size_t)radnum * radnum, sizeof(double));
    zernset(rho, theta, rorder, aorder, znum, polyout, derrho, dertheta, rnm, rpr, radnum);
    free(rnm);
    free(rpr);
}

/* zernsurf.f95:8-101 (opd==NULL) and :108-203 (opd!=NULL: opd += t*nr once, :199).
 * Returns -1 if an order pair addresses outside the reference's radnum x radnum table
 * (the reference reads out of bounds there: undefined, so nothing to be in parity with). */
static int tracezern_core(double *opd, double *x, double *y, double *z, const double *l,
                          const double *m, const double *n, double *ux, double *uy, double *uz,
                          int64_t num, const double *coeff, const int32_t *rorder,
                          const int32_t *aorder, int32_t arrsize, double rad, double nr)
{
    const int radnum = zern_radnum(arrsize);
    for (int c = 0; c < arrsize; c++) {
        int am = aorder[c] < 0 ? -aorder[c] : aorder[c];
        if (rorder[c] < 0 || rorder[c] >= radnum || am > rorder[c] || ((rorder[c] - am) & 1))
            return -1;
    }
#pragma omp parallel
    {
        double *zern = (double *)malloc(sizeof(double) * 3 * (size_t)arrsize);
        double *rhoder = zern + arrsize, *thetader = zern + 2 * arrsize;
        double *rnm = (double *)calloc(2 * (size_t)radnum * radnum, sizeof(double));
        double *rpr = rnm + (size_t)radnum * radnum;
#pragma omp for
        for (int64_t i = 0; i < num; i++) {
            double t = 0., delta = 100.;
            double Fx = 0., Fy = 0., Fz = 0., Fp;
            int it = 0;
            while (fabs(delta) > TOL_1EM10 && it < ORC_NEWTON_CAP) {
                double rho = sqrt(x[i] * x[i] + y[i] * y[i]);
                double theta = atan2(y[i], x[i]);
                double F = z[i], Frho = 0., Ftheta = 0.;
                zernset(rho / rad, theta, rorder, aorder, arrsize, zern, rhoder, thetader, rnm, rpr, radnum);
                for (int c = 0; c < arrsize; c++) {
                    F = F - coeff[c] * zern[c];
                    Frho = Frho - coeff[c] * rhoder[c] / rad;
                    Ftheta = Ftheta - coeff[c] * thetader[c];
                }
                double Frhox = (x[i] / rho) * Frho;
                double Frhoy = (y[i] / rho) * Frho;
                double Fthetax = (-y[i] / rho) * Ftheta / rho;
                double Fthetay = (x[i] / rho) * Ftheta / rho;
                Fx = Frhox + Fthetax;
                Fy = Frhoy + Fthetay;
                Fz = 1.;
                Fp = Fx * l[i] + Fy * m[i] + Fz * n[i];
                delta = -F / Fp;
                x[i] = x[i] + l[i] * delta;
                y[i] = y[i] + m[i] * delta;
                z[i] = z[i] + n[i] * delta;
                t = t + delta;
                it++;
            }
            Fp = sqrt(Fx * Fx + Fy * Fy + Fz * Fz);
            ux[i] = Fx / Fp;
            uy[i] = Fy / Fp;
            uz[i] = Fz / Fp;
            if (opd)
                opd[i] = opd[i] + t * nr;
        }
        free(zern);
        free(rnm);
    }
    return 0;
}

int orc_tracezern(double *x, double *y, double *z, const double *l, const double *m, const double *n,
                  double *ux, double *uy, double *uz, int64_t num, const double *coeff,
                  const int32_t *rorder, const int32_t *aorder, int32_t arrsize, double rad)
{
    return tracezern_core(NULL, x, y, z, l, m, n, ux, uy, uz, num, coeff, rorder, aorder, arrsize, rad, 0.);
}

int orc_tracezernopd(double *opd, double *x, double *y, double *z, const double *l, const double *m,
                     const double *n, double *ux, double *uy, double *uz, int64_t num,
                     const double *coeff, const int32_t *rorder, const int32_t *aorder,
                     int32_t arrsize, double rad, double nr)
{
    return tracezern_core(opd, x, y, z, l, m, n, ux, uy, uz, num, coeff, rorder, aorder, arrsize, rad, nr);
}

/* ------------------------------------------------------------------------------------ */
/* transformationsf.f95:3-28.  sin/cos are re-evaluated by every call in the reference;
 * they are pure functions of theta, so hoisting them is value-preserving.               */
static inline void rotatevector(double *x, double *y, double *z, double c, double s, int axis)
{
    double o1, o2, o3;
    if (axis == 1) {
        o1 = *x;
        o2 = c * *y - s * *z;
        o3 = s * *y + c * *z;
    } else if (axis == 2) {
        o1 = c * *x + s * *z;
        o2 = *y;
        o3 = -s * *x + c * *z;
    } else {
        o1 = c * *x - s * *y;
        o2 = s * *x + c * *y;
        o3 = *z;
    }
    *x = o1;
    *y = o2;
    *z = o3;
}

/* transformationsf.f95:134-163 */
void orc_transform(double *x, double *y, double *z, double *l, double *m, double *n, double *ux,
                   double *uy, double *uz, int64_t num, double tx, double ty, double tz, double rx,
                   double ry, double rz)
{
    double cx, sx, cy, sy, cz, sz;
    ref_cos_sin(rx, &cx, &sx);
    ref_cos_sin(ry, &cy, &sy);
    ref_cos_sin(rz, &cz, &sz);
#pragma omp parallel for
    for (int64_t i = 0; i < num; i++) {
        x[i] = x[i] + tx;
        y[i] = y[i] + ty;
        z[i] = z[i] + tz;
        rotatevector(&x[i], &y[i], &z[i], cx, sx, 1);
        rotatevector(&l[i], &m[i], &n[i], cx, sx, 1);
        rotatevector(&ux[i], &uy[i], &uz[i], cx, sx, 1);
        rotatevector(&x[i], &y[i], &z[i], cy, sy, 2);
        rotatevector(&l[i], &m[i], &n[i], cy, sy, 2);
        rotatevector(&ux[i], &uy[i], &uz[i], cy, sy, 2);
        rotatevector(&x[i], &y[i], &z[i], cz, sz, 3);
        rotatevector(&l[i], &m[i], &n[i], cz, sz, 3);
        rotatevector(&ux[i], &uy[i], &uz[i], cz, sz, 3);
    }
}

/* transformationsf.f95:168-201 */
void orc_itransform(double *x, double *y, double *z, double *l, double *m, double *n, double *ux,
                    double *uy, double *uz, int64_t num, double tx, double ty, double tz, double rx,
                    double ry, double rz)
{
    double cx, sx, cy, sy, cz, sz;       /* tmp = -rz ...: rotatevector(..., tmp, axis) (:180-191) */
    ref_cos_sin(-rx, &cx, &sx);
    ref_cos_sin(-ry, &cy, &sy);
    ref_cos_sin(-rz, &cz, &sz);
#pragma omp parallel for
    for (int64_t i = 0; i < num; i++) {
        rotatevector(&x[i], &y[i], &z[i], cz, sz, 3);
        rotatevector(&l[i], &m[i], &n[i], cz, sz, 3);
        rotatevector(&ux[i], &uy[i], &uz[i], cz, sz, 3);
        rotatevector(&x[i], &y[i], &z[i], cy, sy, 2);
        rotatevector(&l[i], &m[i], &n[i], cy, sy, 2);
        rotatevector(&ux[i], &uy[i], &uz[i], cy, sy, 2);
        rotatevector(&x[i], &y[i], &z[i], cx, sx, 1);
        rotatevector(&l[i], &m[i], &n[i], cx, sx, 1);
        rotatevector(&ux[i], &uy[i], &uz[i], cx, sx, 1);
        x[i] = x[i] - tx;
        y[i] = y[i] - ty;
        z[i] = z[i] - tz;
    }
}

/* transformationsf.f95:60-79 */
void orc_reflect(double *l, double *m, double *n, const double *ux, const double *uy, const double *uz,
                 int64_t num)
{
#pragma omp parallel for
    for (int64_t i = 0; i < num; i++) {
        double dot = ux[i] * l[i] + uy[i] * m[i] + uz[i] * n[i];
        l[i] = l[i] - 2 * dot * ux[i];
        m[i] = m[i] - 2 * dot * uy[i];
        n[i] = n[i] - 2 * dot * uz[i];
    }
}

/* transformationsf.f95:205-238 (wave scalar, sign from n) and :242-272 (per-ray wave, sign
 * from y).  pi = acos(-1.) is evaluated in SINGLE precision (:214, :251).                 */
static void radgrat_core(const double *x, const double *y, double *l, double *m, double *n,
                         const double *wavev, double wave, int64_t num, double dpermm, double order)
{
    const double pi = (double)acosf(-1.f);
    /* The reference loops are serial (no omp directive). */
    for (int64_t i = 0; i < num; i++) {
        double sn, d, yaw, w;
        if (wavev) {
            sn = y[i] / fabs(y[i]);
            w = wavev[i];
        } else {
            sn = n[i] / fabs(n[i]);
            w = wave;
        }
        d = dpermm * sqrt(y[i] * y[i] + x[i] * x[i]);
        yaw = -pi / 2 - atan2(x[i], y[i]);
        l[i] = l[i] + sin(yaw) * order * w / d;
        m[i] = m[i] - cos(yaw) * order * w / d;
        n[i] = sn * sqrt(1. - l[i] * l[i] - m[i] * m[i]);
    }
}

void orc_radgrat(const double *x, const double *y, double *l, double *m, double *n, double wave,
                 int64_t num, double dpermm, double order)
{
    radgrat_core(x, y, l, m, n, NULL, wave, num, dpermm, order);
}

void orc_radgratw(const double *x, const double *y, double *l, double *m, double *n,
                  const double *wave, int64_t num, double dpermm, double order)
{
    radgrat_core(x, y, l, m, n, wave, 0., num, dpermm, order);
}

/* ==================================================================================== */
/* Widening batch 1: SPO cones, linear grating, sphere, cylinder,
 * cylindrical conic, paraxial lenses, refraction.                                         */

/* woltsurf.f95:591-638 */
void orc_spocone(double *x, double *y, double *z, double *l, double *m, double *n, double *ux,
                 double *uy, double *uz, int64_t num, double R0, double tg)
{
    const double sl = tan(tg);
    double ctg, stg;
    ref_cos_sin(tg, &ctg, &stg);
#pragma omp parallel for
    for (int64_t i = 0; i < num; i++) {
        double A = n[i] * n[i] * (sl * sl) - m[i] * m[i] - l[i] * l[i];
        double B = 2 * n[i] * sl * R0 + 2 * z[i] * (sl * sl) * n[i] - 2 * x[i] * l[i] - 2 * y[i] * m[i];
        double C = R0 * R0 + 2 * sl * R0 * z[i] + z[i] * z[i] * (sl * sl) - x[i] * x[i] - y[i] * y[i];
        double det = B * B - 4 * A * C;
        if (det >= 0) {
            double t1 = (-B + sqrt(det)) / (2 * A);
            double t2 = (-B - sqrt(det)) / (2 * A);
            if (fabs(t2) < fabs(t1)) t1 = t2;
            x[i] = x[i] + t1 * l[i];
            y[i] = y[i] + t1 * m[i];
            z[i] = z[i] + t1 * n[i];
            double rr = sqrt(x[i] * x[i] + y[i] * y[i]);
            ux[i] = -x[i] / rr * ctg;
            uy[i] = -y[i] / rr * ctg;
            uz[i] = stg;
        } else {
            l[i] = 0.;
            m[i] = 0.;
            n[i] = 0.;
        }
    }
}

/* transformationsf.f95:277-305; order/wave are per-ray arrays in the Fortran signature */
void orc_grat(const double *x, const double *y, double *l, double *m, double *n, int64_t num, double d,
              const double *order, const double *wave)
{
    for (int64_t i = 0; i < num; i++) {
        double sn = n[i] / fabs(n[i]);
        l[i] = l[i] - order[i] * wave[i] / d;
        n[i] = sn * sqrt(1 - l[i] * l[i] - m[i] * m[i]);
        if ((l[i] * l[i] + m[i] * m[i]) > 1) {
            l[i] = 0.;
            m[i] = 0.;
            n[i] = 0.;
        }
    }
}

/* surfacesf.f95:57-98 (opd==NULL) and :101-148 */
static void sphere_core(double *opd, double *x, double *y, double *z, double *l, double *m, double *n,
                        double *ux, double *uy, double *uz, int64_t num, double rad, double nr)
{
    for (int64_t i = 0; i < num; i++) {
        double dotol = l[i] * x[i] + m[i] * y[i] + n[i] * z[i];
        double mago = x[i] * x[i] + y[i] * y[i] + z[i] * z[i];
        double determinant = dotol * dotol - mago + rad * rad;
        if (determinant < 0) {
            x[i] = y[i] = z[i] = 0.;
            l[i] = m[i] = n[i] = 0.;
        } else {
            double d1 = -dotol + sqrt(determinant);
            double d2 = -dotol - sqrt(determinant);
            if (fabs(d2) < fabs(d1)) d1 = d2;
            x[i] = x[i] + d1 * l[i];
            y[i] = y[i] + d1 * m[i];
            z[i] = z[i] + d1 * n[i];
            if (opd) opd[i] = opd[i] + d1 * nr;
        }
        mago = sqrt(x[i] * x[i] + y[i] * y[i] + z[i] * z[i]);
        ux[i] = x[i] / mago;
        uy[i] = y[i] / mago;
        uz[i] = z[i] / mago;
    }
}

void orc_tracesphere(double *x, double *y, double *z, double *l, double *m, double *n, double *ux,
                     double *uy, double *uz, int64_t num, double rad)
{
    sphere_core(NULL, x, y, z, l, m, n, ux, uy, uz, num, rad, 0.);
}

void orc_tracesphereopd(double *opd, double *x, double *y, double *z, double *l, double *m, double *n,
                        double *ux, double *uy, double *uz, int64_t num, double rad, double nr)
{
    sphere_core(opd, x, y, z, l, m, n, ux, uy, uz, num, rad, nr);
}

/* surfacesf.f95:153-196 (opd==NULL) and :199-246; cylinder axis = y */
static void cyl_core(double *opd, double *x, double *y, double *z, double *l, double *m, double *n,
                     double *ux, double *uy, double *uz, int64_t num, double rad, double nr)
{
    for (int64_t i = 0; i < num; i++) {
        double a = l[i] * l[i] + n[i] * n[i];
        double b = 2 * (x[i] * l[i] + z[i] * n[i]);
        double c = x[i] * x[i] + z[i] * z[i] - rad * rad;
        double det = b * b - 4 * a * c;
        if (det < 0) {
            x[i] = y[i] = z[i] = 0.;
            l[i] = m[i] = n[i] = 0.;
        } else {
            double d1 = (-b + sqrt(det)) / 2 / a;
            double d2 = (-b - sqrt(det)) / 2 / a;
            if (fabs(d2) < fabs(d1)) d1 = d2;
            x[i] = x[i] + l[i] * d1;
            y[i] = y[i] + m[i] * d1;
            z[i] = z[i] + n[i] * d1;
            if (opd) opd[i] = opd[i] + d1 * nr;
        }
        double mag = sqrt(x[i] * x[i] + z[i] * z[i]);
        ux[i] = x[i] / mag;
        uz[i] = z[i] / mag;
        uy[i] = 0.;
    }
}

void orc_tracecyl(double *x, double *y, double *z, double *l, double *m, double *n, double *ux,
                  double *uy, double *uz, int64_t num, double rad)
{
    cyl_core(NULL, x, y, z, l, m, n, ux, uy, uz, num, rad, 0.);
}

void orc_tracecylopd(double *opd, double *x, double *y, double *z, double *l, double *m, double *n,
                     double *ux, double *uy, double *uz, int64_t num, double rad, double nr)
{
    cyl_core(opd, x, y, z, l, m, n, ux, uy, uz, num, rad, nr);
}

/* surfacesf.f95:251-296 (Newton, tol 1.e-10 fp32 literal, no iteration cap in the reference) */
void orc_cylconic(double *x, double *y, double *z, const double *l, const double *m, const double *n,
                  double *ux, double *uy, double *uz, int64_t num, double rad, double k)
{
    for (int64_t i = 0; i < num; i++) {
        double delt = 100., Fx = 0., Fy = 0., Fp;
        int it = 0;
        while (fabs(delt) > TOL_1EM10 && it < ORC_NEWTON_CAP) {
            double q = 1 - (1 + k) * (rad * rad) * (x[i] * x[i]);
            double low = 1 + sqrt(q);
            double high = rad * (x[i] * x[i]);
            double dL = -((1 + k) * (rad * rad) * x[i]) / sqrt(q);
            double dH = 2 * rad * x[i];
            double F = y[i] - high / low;
            Fx = (high * dL - low * dH) / (low * low);
            Fy = 1.;
            Fp = Fx * l[i] + Fy * m[i];
            delt = -F / Fp;
            x[i] = x[i] + l[i] * delt;
            y[i] = y[i] + m[i] * delt;
            z[i] = z[i] + n[i] * delt;
            it++;
        }
        Fp = sqrt(Fx * Fx + Fy * Fy);
        ux[i] = Fx / Fp;
        uy[i] = Fy / Fp;
        uz[i] = 0.;
    }
}

/* surfacesf.f95:423-439 / 443-459 */
void orc_paraxial(const double *x, const double *y, double *l, double *m, int64_t num, double F, int yonly)
{
    for (int64_t i = 0; i < num; i++) {
        if (!yonly) l[i] = l[i] - x[i] / F;
        m[i] = m[i] - y[i] / F;
    }
}

/* transformationsf.f95:32-55 */
static void rotateaxis(double *x, double *y, double *z, double theta, double *ux, double *uy, double *uz)
{
    double mag = sqrt(*ux * *ux + *uy * *uy + *uz * *uz);
    *ux = *ux / mag;
    *uy = *uy / mag;
    *uz = *uz / mag;
    double c = cos(theta), s = sin(theta);
    double o1 = (c + *ux * *ux * (1 - c)) * *x + (*ux * *uy * (1 - c) - *uz * s) * *y + (*ux * *uz * (1 - c) + *uy * s) * *z;
    double o2 = (*uy * *ux * (1 - c) + *uz * s) * *x + (c + *uy * *uy * (1 - c)) * *y + (*uy * *uz * (1 - c) - *ux * s) * *z;
    double o3 = (*uz * *ux * (1 - c) - *uy * s) * *x + (*uz * *uy * (1 - c) + *ux * s) * *y + (c + *uz * *uz * (1 - c)) * *z;
    *x = o1;
    *y = o2;
    *z = o3;
}

/* transformationsf.f95:82-130 */
void orc_refract(double *l, double *m, double *n, double *ux, double *uy, double *uz, int64_t num,
                 double n1, double n2)
{
    for (int64_t i = 0; i < num; i++) {
        double dot = l[i] * ux[i] + m[i] * uy[i] + n[i] * uz[i];
        if (dot < 0) {
            ux[i] = -ux[i];
            uy[i] = -uy[i];
            uz[i] = -uz[i];
            dot = -dot;
        }
        if (dot == 1) continue;
        double t1 = acos(dot);
        double t2 = asin((n1 / n2) * sin(t1));
        double cx = uy[i] * n[i] - m[i] * uz[i];
        double cy = l[i] * uz[i] - ux[i] * n[i];
        double cz = ux[i] * m[i] - l[i] * uy[i];
        double dt = t2 - t1;
        rotateaxis(&l[i], &m[i], &n[i], dt, &cx, &cy, &cz);
        double alpha = sqrt(l[i] * l[i] + m[i] * m[i] + n[i] * n[i]);
        l[i] = l[i] / alpha;
        m[i] = m[i] / alpha;
        n[i] = n[i] / alpha;
    }
}

/* ==================================================================================== */
/* Widening batch 2: sinusoidally perturbed primary, Legendre-Legendre
 * deformed Wolter / ellipsoid mirrors, Wolter-Schwarzschild primary and secondary.        */

/* x**k with a run-time INTEGER exponent: binary exponentiation, as compiler-rt's __powidf2
 * (what the flang-compiled reference calls for integer powers). */
static double powi(double a, int b)
{
    const int recip = b < 0;
    double r = 1;
    while (1) {
        if (b & 1) r *= a;
        b /= 2;
        if (b == 0) break;
        a *= a;
    }
    return recip ? 1 / r : r;
}

static double ipow_int(int base, int e) /* integer ** integer (e >= 0), as Fortran */
{
    long long r = 1;
    for (int i = 0; i < e; i++) r *= base;
    return (double)(int)r;
}

static double factorial_d(int n) /* specialFunctions.f95:2-14 */
{
    double a = 1;
    for (int i = 1; i <= n; i++) a = a * (double)i;
    return a;
}

/* specialFunctions.f95:337-360 */
static double legendre(double x, int n)
{
    double x2 = fabs(x) > 1. ? x / fabs(x) : x;
    double s = 0.;
    if (n == 0) return 1.;
    for (int i = 0; i <= n / 2; i++)
        s = s + ipow_int(-1, i) * factorial_d(2 * n - 2 * i) / factorial_d(i) / factorial_d(n - i) /
                    factorial_d(n - 2 * i) / ipow_int(2, n) * powi(x2, n - 2 * i);
    return s;
}

/* specialFunctions.f95:363-388 */
static double legendrep(double x, int n)
{
    double s = 0.;
    if (n == 0)
        s = 0.;
    else if (n == 1)
        s = 1.;
    else if (x == 0. && n % 2 == 0)
        s = 0.;
    else
        for (int i = 0; i <= n / 2; i++)
            s = s + ipow_int(-1, i) * factorial_d(2 * n - 2 * i) / factorial_d(i) / factorial_d(n - i) /
                        factorial_d(n - 2 * i) / ipow_int(2, n) * (n - 2 * i) * powi(x, n - 2 * i - 1);
    if (fabs(x) > 1.) s = 0.;
    return s;
}

void orc_legendre(double x, int32_t n, double *p, double *dp)
{
    *p = legendre(x, n);
    *dp = legendrep(x, n);
}

/* woltsurf.f95:167-215 */
void orc_woltersine(double *x, double *y, double *z, const double *l, const double *m, const double *n,
                    double *ux, double *uy, double *uz, int64_t num, double r0, double z0, double amp,
                    double freq)
{
    double alpha = .25 * atan(r0 / z0);
    double thetah = 3. * alpha, thetap = alpha;
    double p = z0 * tan(4 * alpha) * tan(thetap);
    double d = z0 * tan(4 * alpha) * tan(4 * alpha - thetah);
    double e = cos(4 * alpha) * (1 + tan(4 * alpha) * tan(thetah));
    const double pi32 = (double)acosf(-1.f);
    const double twopi32 = (double)(2.f * acosf(-1.f)); /* 2*acos(-1.) is a REAL*4 product */
    for (int64_t i = 0; i < num; i++) {
        double delt = 100., Fx = 0., Fy = 0., Fz = 0., Fp;
        int it = 0;
        while (fabs(delt) > TOL_1EM10 && it < ORC_NEWTON_CAP) {
            double rad = sqrt(x[i] * x[i] + y[i] * y[i]) + amp * sin(twopi32 * freq * z[i]);
            double F = 2 * p * z[i] + p * p + 4 * (e * e) * p * d / (e * e - 1) - rad * rad;
            Fx = -2. * x[i];
            Fy = -2. * y[i];
            Fz = 2. * p - 2 * rad * amp * 2 * pi32 * freq * cos(twopi32 * freq * z[i]);
            Fp = Fx * l[i] + Fy * m[i] + Fz * n[i];
            delt = -F / Fp;
            x[i] = x[i] + l[i] * delt;
            y[i] = y[i] + m[i] * delt;
            z[i] = z[i] + n[i] * delt;
            it++;
        }
        Fp = sqrt(Fx * Fx + Fy * Fy + Fz * Fz);
        ux[i] = Fx / Fp;
        uy[i] = Fy / Fp;
        uz[i] = Fz / Fp;
    }
}

/* Shared body of wolterprimLL (woltsurf.f95:220-288), woltersecLL (:293-379) and
 * ellipsoidWoltLL (:643-718).  kind: 0 primary, 1 secondary, 2 ellipsoid. */
static void ll_core(int kind, double *x, double *y, double *z, const double *l, const double *m,
                    const double *n, double *ux, double *uy, double *uz, int64_t num, double p, double d,
                    double e, double aa, double bb, double zfoc, double tol, double zmax, double zmin,
                    double dphi, const double *coeff, const int32_t *axial, const int32_t *az, int cnum)
{
    for (int64_t i = 0; i < num; i++) {
        double delt = 100., Fx = 0., Fy = 0., Fz = 0., Fp, F;
        int it = 0;
        while (fabs(delt) > tol && it < ORC_NEWTON_CAP) {
            double ang = atan2(y[i], x[i]);
            double zarg = (z[i] - ((zmax + zmin) / 2.)) / ((zmax - zmin) / 2.);
            double targ = 2 * ang / dphi;
            double add = 0., addx = 0., addy = 0., addz = 0.;
            for (int a = 0; a < cnum; a++) {
                add = add + coeff[a] * legendre(zarg, axial[a]) * legendre(targ, az[a]);
                addx = addx - coeff[a] * legendre(zarg, axial[a]) * legendrep(targ, az[a]) * (2 / dphi) *
                                  (y[i] / (y[i] * y[i] + x[i] * x[i]));
                addy = addy + coeff[a] * legendre(zarg, axial[a]) * legendrep(targ, az[a]) * (2 / dphi) *
                                  (x[i] / (y[i] * y[i] + x[i] * x[i]));
                addz = addz + coeff[a] * legendrep(zarg, axial[a]) * legendre(targ, az[a]) * 2 / (zmax - zmin);
            }
            double rr = sqrt(x[i] * x[i] + y[i] * y[i]);
            double G = rr + add;
            if (kind == 0) {
                F = -(G * G - p * p - 2 * p * z[i] - 4 * (e * e) * p * d / (e * e - 1));
                Fx = -(2 * G * (x[i] / rr + addx));
                Fy = -(2 * G * (y[i] / rr + addy));
                Fz = 2 * p - 2 * G * addz;
            } else if (kind == 1) {
                double dz = d + z[i];
                F = -(G * G - (e * e) * (dz * dz) + z[i] * z[i]);
                Fx = -(2 * G * (x[i] / rr + addx));
                Fy = -(2 * G * (y[i] / rr + addy));
                Fz = 2 * (e * e) * dz - 2 * z[i] - 2 * G * addz;
            } else {
                double zz = z[i] - zfoc;
                F = zz * zz / (aa * aa) + G * G / (bb * bb) - 1.;
                Fx = 2 * G / (bb * bb) * (x[i] / rr + addx);
                Fy = 2 * G / (bb * bb) * (y[i] / rr + addy);
                Fz = 2 * zz / (aa * aa) + (2 * G / (bb * bb)) * (addz);
            }
            Fp = Fx * l[i] + Fy * m[i] + Fz * n[i];
            delt = -F / Fp;
            x[i] = x[i] + l[i] * delt;
            y[i] = y[i] + m[i] * delt;
            z[i] = z[i] + n[i] * delt;
            it++;
        }
        Fp = sqrt(Fx * Fx + Fy * Fy + Fz * Fz);
        ux[i] = Fx / Fp;
        uy[i] = Fy / Fp;
        uz[i] = Fz / Fp;
    }
}

void orc_wolterprimll(double *x, double *y, double *z, const double *l, const double *m, const double *n,
                      double *ux, double *uy, double *uz, int64_t num, double r0, double z0, double zmax,
                      double zmin, double dphi, const double *coeff, const int32_t *axial,
                      const int32_t *az, int32_t cnum)
{
    double alpha = .25 * atan(r0 / z0);
    double thetah = 3. * alpha, thetap = alpha;
    double p = z0 * tan(4 * alpha) * tan(thetap);
    double d = z0 * tan(4 * alpha) * tan(4 * alpha - thetah);
    double e = cos(4 * alpha) * (1 + tan(4 * alpha) * tan(thetah));
    ll_core(0, x, y, z, l, m, n, ux, uy, uz, num, p, d, e, 0., 0., 0., TOL_1EM10, zmax, zmin, dphi, coeff,
            axial, az, cnum);
}

void orc_woltersecll(double *x, double *y, double *z, const double *l, const double *m, const double *n,
                     double *ux, double *uy, double *uz, int64_t num, double r0, double z0, double psi,
                     double zmax, double zmin, double dphi, const double *coeff, const int32_t *axial,
                     const int32_t *az, int32_t cnum)
{
    double p, d, e;
    orc_wolter_params(r0, z0, psi, &p, &d, &e);
    ll_core(1, x, y, z, l, m, n, ux, uy, uz, num, p, d, e, 0., 0., 0., (double)1.e-7f, zmax, zmin, dphi,
            coeff, axial, az, cnum);
}

void orc_ellipsoid_params(double R0, double z0, double psi, double S, double *aa, double *bb, double *zfoc)
{
    double P = R0 / sin((psi * asin(R0 / z0) - asin(R0 / S)) / (1 + psi));
    double ff = (S + P) / 2.;
    double bq = -(R0 * R0 + (ff - P) * (ff - P) + ff * ff);
    double cq = ff * ff * ((ff - P) * (ff - P));
    *aa = sqrt((-bq + sqrt(bq * bq - 4 * cq)) / 2.);
    *bb = sqrt(*aa * *aa - ff * ff);
    *zfoc = ff - P + z0;
}

void orc_ellipsoidwoltll(double *x, double *y, double *z, const double *l, const double *m, const double *n,
                         double *ux, double *uy, double *uz, int64_t num, double r0, double z0, double psi,
                         double S, double zmax, double zmin, double dphi, const double *coeff,
                         const int32_t *axial, const int32_t *az, int32_t cnum)
{
    double aa, bb, zfoc;
    orc_ellipsoid_params(r0, z0, psi, S, &aa, &bb, &zfoc);
    ll_core(2, x, y, z, l, m, n, ux, uy, uz, num, 0., 0., 0., aa, bb, zfoc, TOL_1EM10, zmax, zmin, dphi, coeff,
            axial, az, cnum);
}

/* woltsurf.f95:387-476.  26-iteration cap; a non-converged / NaN ray gets its start position
 * back and keeps its old normal. */
static void wsprimary_core(double *x, double *y, double *z, const double *l, const double *m, const double *n,
                           double *ux, double *uy, double *uz, int64_t num, double alpha, double z0, double psi,
                           int back, double thick)
{
    const double betas = 4 * alpha, ff = z0 / cos(betas), g = ff / psi;
    const double tb = tan(betas / 2), k = tb * tb;
    for (int64_t i = 0; i < num; i++) {
        double delt = 100., F, Fx = 0., Fy = 0., Fz = 0., Fp, Fb, beta, kterm, r, dbdx, dbdy;
        int c = 0, flag;
        const double xi = x[i], yi = y[i], zi = z[i];
        while (fabs(delt) > TOL_1EM8) {
            /* wsprimaryBack (woltsurf.f95:726-816): the surface is evaluated at the radius shrunk by `thick` */
            double xe = x[i], ye = y[i];
            if (back) {
                double rb = sqrt(x[i] * x[i] + y[i] * y[i]);
                double th = atan2(y[i], x[i]);
                xe = (rb - thick) * cos(th);
                ye = (rb - thick) * sin(th);
            }
            beta = asin(sqrt(xe * xe + ye * ye) / ff);
            flag = 0;
            if (beta <= betas) {
                beta = betas;
                flag = 1;
                kterm = 0.;
            } else {
                double t = tan(beta / 2);
                kterm = (1 / k) * (t * t) - 1;
            }
            {
                double sb2 = sin(betas / 2), sb = sin(beta), cb2 = cos(beta / 2);
                F = -z[i] - ff * (sb2 * sb2) + ff * ff * (sb * sb) / (4 * ff * (sb2 * sb2)) +
                    g * (((cb2 * cb2) * cb2) * cb2) * pow(kterm, 1 - k) /* x**4 with a literal exponent: sequential products */;
                Fb = ff * ff * sb * cos(beta) / (2 * ff * (sb2 * sb2)) -
                     2 * g * ((cb2 * cb2) * cb2) * sin(beta / 2) * pow(kterm, 1 - k) +
                     g * (1 - k) * cb2 * sin(beta / 2) * pow(kterm, -k) * (1 / k);
            }
            Fz = -1.;
            if (flag == 1) {
                double sb2 = sin(betas / 2);
                r = sqrt(xe * xe + ye * ye);
                Fb = ff * ff * sin(betas) * cos(betas) / (2 * ff * (sb2 * sb2)) +
                     g * (1 - k) * cos(betas / 2) * sin(betas / 2) * (1 / k);
                F = F + (r - ff * sin(betas)) * z[i] / (r * r + z[i] * z[i]) * Fb;
                Fz = Fz + (r - ff * sin(betas)) * (r * r - z[i] * z[i]) /
                              ((r * r + z[i] * z[i]) * (r * r + z[i] * z[i])) * Fb;
            }
            {
                double r2 = xe * xe + ye * ye;
                dbdx = xe / sqrt(1 - r2 / (ff * ff)) / ff / sqrt(r2);
                dbdy = ye / sqrt(1 - r2 / (ff * ff)) / ff / sqrt(r2);
            }
            Fx = Fb * dbdx;
            Fy = Fb * dbdy;
            Fp = Fx * l[i] + Fy * m[i] + Fz * n[i];
            delt = -F / Fp;
            x[i] = x[i] + l[i] * delt;
            y[i] = y[i] + m[i] * delt;
            z[i] = z[i] + n[i] * delt;
            if (c > 25 || isnan(delt)) {
                delt = 0.;
                x[i] = xi;
                y[i] = yi;
                z[i] = zi;
                c = 1000;
            }
            c = c + 1;
        }
        if (c < 26) {
            Fp = sqrt(Fx * Fx + Fy * Fy + Fz * Fz);
            ux[i] = -Fx / Fp;
            uy[i] = -Fy / Fp;
            uz[i] = -Fz / Fp;
        }
    }
}

void orc_wsprimary(double *x, double *y, double *z, const double *l, const double *m, const double *n,
                   double *ux, double *uy, double *uz, int64_t num, double alpha, double z0, double psi)
{
    wsprimary_core(x, y, z, l, m, n, ux, uy, uz, num, alpha, z0, psi, 0, 0.);
}

void orc_wsprimaryback(double *x, double *y, double *z, const double *l, const double *m, const double *n,
                       double *ux, double *uy, double *uz, int64_t num, double alpha, double z0, double psi,
                       double thick)
{
    wsprimary_core(x, y, z, l, m, n, ux, uy, uz, num, alpha, z0, psi, 1, thick);
}

/* woltsurf.f95:484-588 (and wssecondaryBack, :826-933, with back=1) */
static void wssecondary_core(double *x, double *y, double *z, const double *l, const double *m, const double *n,
                             double *ux, double *uy, double *uz, int64_t num, double alpha, double z0,
                             double psi, int back, double thick)
{
    const double betas = 4 * alpha, ff = z0 / cos(betas), g = ff / psi;
    const double tb = tan(betas / 2), k = tb * tb;
    for (int64_t i = 0; i < num; i++) {
        double delt = 100., F, Fx = 0., Fy = 0., Fz = 0., Fp, Fb, beta, kterm, a, dadb, dbdx, dbdy, dbdz;
        int c = 0, flag;
        const double xi = x[i], yi = y[i], zi = z[i];
        while (fabs(delt) > TOL_1EM8) {
            double xe = x[i], ye = y[i];
            if (back) {
                double rb = sqrt(x[i] * x[i] + y[i] * y[i]);
                double th = atan2(y[i], x[i]);
                xe = (rb - thick) * cos(th);
                ye = (rb - thick) * sin(th);
            }
            double rr = sqrt(xe * xe + ye * ye);
            beta = atan2(rr, z[i]);
            flag = 0;
            if (beta <= betas) {
                beta = betas;
                kterm = 0;
                a = 1 / ff;
                flag = 1;
            } else {
                double t = tan(beta / 2);
                kterm = (1 / k) * (t * t) - 1;
                a = (1 - cos(beta)) / (1 - cos(betas)) / ff + (1 + cos(beta)) / (2 * g) * pow(kterm, 1 + k);
            }
            F = -z[i] + cos(beta) / a;
            if (flag == 1) {
                double cb2 = cos(betas / 2);
                double dadbs = sin(betas) / ff / (1 - cos(betas)) +
                               (k + 1) * (cos(betas) + 1) * tan(betas / 2) / (cb2 * cb2) / 2 / g / k;
                double sbs = sin(betas);
                double dbdzs = -(sbs * sbs) / rr;
                double gam = (-ff * sin(betas) - ff * ff * cos(betas) * dadbs) * dbdzs;
                Fb = 0.;
                F = F + gam * (z[i] - rr / tan(betas));
                Fx = -2. / tan(betas) * xe / rr;
                Fy = -2. / tan(betas) * ye / rr;
                Fz = gam - 1.;
            } else {
                double cb2 = cos(beta / 2);
                double r3 = xe * xe + ye * ye + z[i] * z[i];
                dadb = sin(beta) / ff / (1 - cos(betas)) - sin(beta) / (2 * g) * pow(kterm, 1 + k) +
                       (k + 1) * (cos(beta) + 1) * tan(beta / 2) * pow(kterm, k) / 2 / g / k / (cb2 * cb2);
                Fb = -sin(beta) / a - cos(beta) / (a * a) * dadb;
                dbdx = xe * z[i] / r3 / rr;
                dbdy = ye * z[i] / r3 / rr;
                dbdz = -rr / r3;
                Fx = Fb * dbdx;
                Fy = Fb * dbdy;
                Fz = -1. + Fb * dbdz;
            }
            Fp = Fx * l[i] + Fy * m[i] + Fz * n[i];
            delt = -F / Fp;
            x[i] = x[i] + l[i] * delt;
            y[i] = y[i] + m[i] * delt;
            z[i] = z[i] + n[i] * delt;
            if (c > 25 || isnan(delt)) {
                delt = 0.;
                x[i] = xi;
                y[i] = yi;
                z[i] = zi;
                c = 1000;
            }
            c = c + 1;
        }
        if (c < 26) {
            Fp = sqrt(Fx * Fx + Fy * Fy + Fz * Fz);
            ux[i] = Fx / Fp;
            uy[i] = Fy / Fp;
            uz[i] = Fz / Fp;
        }
    }
}

void orc_wssecondary(double *x, double *y, double *z, const double *l, const double *m, const double *n,
                     double *ux, double *uy, double *uz, int64_t num, double alpha, double z0, double psi)
{
    wssecondary_core(x, y, z, l, m, n, ux, uy, uz, num, alpha, z0, psi, 0, 0.);
}

void orc_wssecondaryback(double *x, double *y, double *z, const double *l, const double *m, const double *n,
                         double *ux, double *uy, double *uz, int64_t num, double alpha, double z0, double psi,
                         double thick)
{
    wssecondary_core(x, y, z, l, m, n, ux, uy, uz, num, alpha, z0, psi, 1, thick);
}

/* ==================================================================================== */
/* Widening batch 3: torus, conicplus[opd], legSurf, zernphase, tracezernrot.               */

/* surfacesf.f95:468-507.  Note the normal: Fp = sqrt(Fx^2+Fy^2) WITHOUT Fz (:501) -- reproduced. */
void orc_torus(double *x, double *y, double *z, const double *l, const double *m, const double *n,
               double *ux, double *uy, double *uz, int64_t num, double rin, double rout)
{
    for (int64_t i = 0; i < num; i++) {
        double delt = 100., Fx = 0., Fy = 0., Fz = 0., Fp;
        int it = 0;
        while (fabs(delt) > TOL_1EM10 && it < ORC_NEWTON_CAP) {
            double zr = z[i] + rin + rout;
            double t = zr * zr + y[i] * y[i] + x[i] * x[i] + rout * rout - rin * rin;
            double F = t * t - (4 * (rout * rout) * (y[i] * y[i] + zr * zr));
            double zr2 = rin + rout + z[i];
            Fx = 4 * x[i] * (-(rin * rin) + zr2 * zr2 + rout * rout + x[i] * x[i] + y[i] * y[i]);
            double q = 2 * rin * (rout + z[i]) + 2 * rout * z[i] + z[i] * z[i] + y[i] * y[i] + x[i] * x[i];
            Fy = 4 * y[i] * q;
            Fz = 4 * (rout + rin + z[i]) * q;
            Fp = Fx * l[i] + Fy * m[i] + Fz * n[i];
            delt = -F / Fp;
            x[i] = x[i] + l[i] * delt;
            y[i] = y[i] + m[i] * delt;
            z[i] = z[i] + n[i] * delt;
            it++;
        }
        Fp = sqrt(Fx * Fx + Fy * Fy);
        ux[i] = Fx / Fp;
        uy[i] = Fy / Fp;
        uz[i] = Fz / Fp;
    }
}

/* surfacesf.f95:514-572 (opd==NULL) and :579-638 */
static void conicplus_core(double *opd, double *x, double *y, double *z, const double *l, const double *m,
                           const double *n, double *ux, double *uy, double *uz, int64_t num, double R,
                           double K, const double *p, int Np, double nr)
{
    const double Fz = 1., c = 1 / R;
    for (int64_t i = 0; i < num; i++) {
        double delt = 100., Fx = 0., Fy = 0., Fp;
        int it = 0;
        while (fabs(delt) > TOL_1EM10 && it < ORC_NEWTON_CAP) {
            double rad = sqrt(x[i] * x[i] + y[i] * y[i]);
            double a0 = 0., a1 = 0.;
            for (int j = 1; j <= Np; j++) {
                a0 = a0 + p[j - 1] * powi(rad, 2 * j);
                a1 = a1 + p[j - 1] * (2 * j) * powi(rad, 2 * j - 1);
            }
            double sq = sqrt(1 - (K + 1) * (c * c) * (rad * rad));
            double denom = sq + 1;
            double F = z[i] - c * (rad * rad) / denom + a0;
            double Fr = -(2 * c * rad / denom + ((K + 1) * ((rad * rad) * rad) * ((c * c) * c)) / ((denom * denom) * sq)) + a1;
            Fx = Fr * x[i] / rad;
            Fy = Fr * y[i] / rad;
            Fp = Fx * l[i] + Fy * m[i] + Fz * n[i];
            delt = -F / Fp;
            x[i] = x[i] + l[i] * delt;
            y[i] = y[i] + m[i] * delt;
            z[i] = z[i] + n[i] * delt;
            if (opd) opd[i] = opd[i] + delt * nr;
            it++;
        }
        Fp = sqrt(Fx * Fx + Fy * Fy + Fz * Fz);
        ux[i] = Fx / Fp;
        uy[i] = Fy / Fp;
        uz[i] = Fz / Fp;
    }
}

void orc_conicplus(double *x, double *y, double *z, const double *l, const double *m, const double *n,
                   double *ux, double *uy, double *uz, int64_t num, double R, double K, const double *p,
                   int32_t Np)
{
    conicplus_core(NULL, x, y, z, l, m, n, ux, uy, uz, num, R, K, p, Np, 0.);
}

void orc_conicplusopd(double *opd, double *x, double *y, double *z, const double *l, const double *m,
                      const double *n, double *ux, double *uy, double *uz, int64_t num, double R, double K,
                      const double *p, int32_t Np, double nr)
{
    conicplus_core(opd, x, y, z, l, m, n, ux, uy, uz, num, R, K, p, Np, nr);
}

/* surfacesf.f95:642-668 */
void orc_legsurf(const double *x, const double *y, double *l, double *m, double *n, int64_t num,
                 double xwidth, double ywidth, double order, const double *coeff, const int32_t *xo,
                 const int32_t *yo, int32_t Nc)
{
    for (int64_t i = 0; i < num; i++) {
        double dphidx = 0, dphidy = 0;
        for (int j = 0; j < Nc; j++) {
            dphidx = dphidx + coeff[j] * legendre(y[i] / ywidth, yo[j]) * legendrep(x[i] / xwidth, xo[j]);
            dphidy = dphidy + coeff[j] * legendrep(y[i] / ywidth, yo[j]) * legendre(x[i] / xwidth, xo[j]);
        }
        l[i] = l[i] + dphidx * order / xwidth;
        m[i] = m[i] + dphidy * order / ywidth;
        n[i] = n[i] / fabs(n[i]) * sqrt(1. - l[i] * l[i] - m[i] * m[i]);
    }
}

/* zernsurf.f95:206-250 */
int orc_zernphase(double *opd, const double *x, const double *y, double *l, double *m, double *n,
                  int64_t num, const double *coeff, const int32_t *rorder, const int32_t *aorder,
                  int32_t arrsize, double rad, double wave)
{
    const int radnum = zern_radnum(arrsize);
    for (int c = 0; c < arrsize; c++) {
        int am = aorder[c] < 0 ? -aorder[c] : aorder[c];
        if (rorder[c] < 0 || rorder[c] >= radnum || am > rorder[c] || ((rorder[c] - am) & 1)) return -1;
    }
    double *zern = (double *)malloc(sizeof(double) * 3 * (size_t)arrsize);
    double *rhoder = zern + arrsize, *thetader = zern + 2 * arrsize;
    double *rnm = (double *)calloc(2 * (size_t)radnum * radnum, sizeof(double));
    double *rpr = rnm + (size_t)radnum * radnum;
    for (int64_t i = 0; i < num; i++) {
        double rho = sqrt(x[i] * x[i] + y[i] * y[i]);
        double theta = atan2(y[i], x[i]);
        zernset(rho / rad, theta, rorder, aorder, arrsize, zern, rhoder, thetader, rnm, rpr, radnum);
        double F = 0., Frho = 0., Ftheta = 0.;
        for (int c = 0; c < arrsize; c++) {
            F = F + coeff[c] * zern[c];
            Frho = Frho + coeff[c] * rhoder[c] / rad;
            Ftheta = Ftheta + coeff[c] * thetader[c];
        }
        double Fx = (x[i] / rho) * Frho + (-y[i] / rho) * Ftheta / rho;
        double Fy = (y[i] / rho) * Frho + (x[i] / rho) * Ftheta / rho;
        l[i] = l[i] + Fx * wave;
        m[i] = m[i] + Fy * wave;
        n[i] = copysign(sqrt(1. - l[i] * l[i] - m[i] * m[i]), n[i]);
        opd[i] = opd[i] + F * wave;
    }
    free(zern);
    free(rnm);
    return 0;
}

/* zernsurf.f95:257-359 (serial; the reference's OpenMP build has a data race here) */
int orc_tracezernrot(double *x, double *y, double *z, const double *l, const double *m, const double *n,
                     double *ux, double *uy, double *uz, int64_t num, const double *coeff1,
                     const int32_t *rorder1, const int32_t *aorder1, int32_t n1, const double *coeff2,
                     const int32_t *rorder2, const int32_t *aorder2, int32_t n2, double rad, double rot)
{
    const int rn1 = zern_radnum(n1), rn2 = zern_radnum(n2);
    for (int c = 0; c < n1; c++) {
        int am = aorder1[c] < 0 ? -aorder1[c] : aorder1[c];
        if (rorder1[c] < 0 || rorder1[c] >= rn1 || am > rorder1[c] || ((rorder1[c] - am) & 1)) return -1;
    }
    for (int c = 0; c < n2; c++) {
        int am = aorder2[c] < 0 ? -aorder2[c] : aorder2[c];
        if (rorder2[c] < 0 || rorder2[c] >= rn2 || am > rorder2[c] || ((rorder2[c] - am) & 1)) return -1;
    }
    int nmx = n1 > n2 ? n1 : n2, rmx = rn1 > rn2 ? rn1 : rn2;
    double *zern = (double *)malloc(sizeof(double) * 3 * (size_t)nmx);
    double *rhoder = zern + nmx, *thetader = zern + 2 * nmx;
    double *rnm = (double *)calloc(2 * (size_t)rmx * rmx, sizeof(double));
    double *rpr = rnm + (size_t)rmx * rmx;
    for (int64_t i = 0; i < num; i++) {
        double delta = 100., Fx = 0., Fy = 0., Fz = 0., Fp;
        int it = 0;
        while (fabs(delta) > TOL_1EM10 && it < ORC_NEWTON_CAP) {
            double rho = sqrt(x[i] * x[i] + y[i] * y[i]);
            double theta = atan2(y[i], x[i]);
            double F = z[i], Frho = 0., Ftheta = 0.;
            zernset(rho / rad, theta, rorder1, aorder1, n1, zern, rhoder, thetader, rnm, rpr, rn1);
            for (int c = 0; c < n1; c++) {
                F = F - coeff1[c] * zern[c];
                Frho = Frho - coeff1[c] * rhoder[c] / rad;
                Ftheta = Ftheta - coeff1[c] * thetader[c];
            }
            zernset(rho / rad, theta + rot, rorder2, aorder2, n2, zern, rhoder, thetader, rnm, rpr, rn2);
            for (int c = 0; c < n2; c++) {
                F = F - coeff2[c] * zern[c];
                Frho = Frho - coeff2[c] * rhoder[c] / rad;
                Ftheta = Ftheta - coeff2[c] * thetader[c];
            }
            Fx = (x[i] / rho) * Frho + (-y[i] / rho) * Ftheta / rho;
            Fy = (y[i] / rho) * Frho + (x[i] / rho) * Ftheta / rho;
            Fz = 1.;
            Fp = Fx * l[i] + Fy * m[i] + Fz * n[i];
            delta = -F / Fp;
            x[i] = x[i] + l[i] * delta;
            y[i] = y[i] + m[i] * delta;
            z[i] = z[i] + n[i] * delta;
            it++;
        }
        Fp = sqrt(Fx * Fx + Fy * Fy + Fz * Fz);
        ux[i] = Fx / Fp;
        uy[i] = Fy / Fp;
        uz[i] = Fz / Fp;
    }
    free(zern);
    free(rnm);
    return 0;
}

/* ---- wavefront sensing: reconstruct.f95 ------------------------------------------------------
 * Arrays are Fortran-ordered (xdim fastest): a(xi,yi) = a[xi + xdim*yi], xi,yi 0-based here. */

/* southwellbin, reconstruct.f95:139-187.  Bins ray slopes into a lenslet grid and converts the mean
 * direction cosine of each bin to a slope tan(asin(.)); empty bins get the sentinel 100.  The
 * reference indexes out of bounds for rays outside the grid (undefined); here they are skipped.
 * xang/yang are accumulated from zero (the reference's intent(out) arrays are "init to 0" by its
 * caller, :137). */
void orc_southwellbin(const double *x, const double *y, const double *l, const double *m, int64_t num,
                      double binsize, double *xang, double *yang, double *phase, int32_t xdim, int32_t ydim)
{
    int64_t nb = (int64_t)xdim * ydim;
    int32_t *accum = (int32_t *)calloc((size_t)nb, sizeof(int32_t));
    for (int64_t b = 0; b < nb; b++) {
        xang[b] = 0.;
        yang[b] = 0.;
        phase[b] = 0.;
    }
    for (int64_t i = 0; i < num; i++) {
        long xb, yb; /* 0-based (the Fortran's 1-based index minus one) */
        if (xdim % 2 == 0)
            xb = (long)floor(x[i] / binsize) + xdim / 2 + 1;
        else
            xb = (long)floor((x[i] + binsize / 2) / binsize) + (xdim - 1) / 2;
        if (ydim % 2 == 0)
            yb = (long)floor(y[i] / binsize) + ydim / 2 + 1;
        else
            yb = (long)floor((y[i] + binsize / 2) / binsize) + (ydim - 1) / 2;
        if (xb < 0 || xb >= xdim || yb < 0 || yb >= ydim) continue;
        xang[xb + (int64_t)xdim * yb] += l[i];
        yang[xb + (int64_t)xdim * yb] += m[i];
        accum[xb + (int64_t)xdim * yb] += 1;
    }
    for (int64_t b = 0; b < nb; b++) {
        if (accum[b] == 0) {
            phase[b] = 100.;
            xang[b] = 100.;
            yang[b] = 100.;
        } else {
            xang[b] = tan(asin(xang[b] / accum[b]));
            yang[b] = tan(asin(yang[b] / accum[b]));
        }
    }
    free(accum);
}

/* reconstruct, reconstruct.f95:1-131.  Successive over-relaxation (Gauss-Seidel order: xi outer,
 * yi inner) of the Southwell geometry; 100. marks invalid lenslets.  Returns the number of sweeps. */
int32_t orc_reconstruct(double *xang, double *yang, int32_t xdim, int32_t ydim, double criteria, double h,
                        double *phase, double *phasec, int32_t maxiter)
{
    const double pi = (double)3.1415926535897931f; /* default-real literal, :13 */
    const double w = 2 / (1 + sin(pi / (sqrt((double)xdim * (double)ydim) + 1)));
    const int64_t nb = (int64_t)xdim * ydim;
    const int64_t sx = 1, sy = xdim;
    int32_t counter = 0, sweeps = 0;
    for (int64_t b = 0; b < nb; b++) phasec[b] = phase[b];
    for (;;) {
        double rms = 0.;
        int32_t nannum = 0;
        for (int32_t xi = 1; xi < xdim - 1; xi++) {
            for (int32_t yi = 1; yi < ydim - 1; yi++) {
                const int64_t o = xi * sx + yi * sy;
                if (phasec[o] == 100.) continue;
                double yplus = yang[o + sy], yneg = yang[o - sy], xplus = xang[o + sx], xneg = xang[o - sx];
                if (yplus == 100.) yplus = -yang[o];
                if (yneg == 100.) yneg = -yang[o];
                if (xplus == 100.) xplus = -xang[o];
                if (xneg == 100.) xneg = -xang[o];
                const double bk = .5 * (yplus - yneg + xplus - xneg) * h;
                double goodpix = 4.;
                double pyplus = phasec[o + sy], pyneg = phasec[o - sy], pxplus = phasec[o + sx], pxneg = phasec[o - sx];
                if (pyplus == 100.) { pyplus = 0.; goodpix -= 1; }
                if (pyneg == 100.) { pyneg = 0.; goodpix -= 1; }
                if (pxplus == 100.) { pxplus = 0.; goodpix -= 1; }
                if (pxneg == 100.) { pxneg = 0.; goodpix -= 1; }
                const double psum = pyplus + pyneg + pxplus + pxneg;
                if (goodpix == 0.) {
                    phasec[o] = 100.;
                    phase[o] = 100.;
                    xang[o] = 100.;
                    yang[o] = 100.;
                    continue;
                }
                phasec[o] = phasec[o] + w * ((psum + bk) / goodpix - phasec[o]);
                nannum += 1;
                const double df = phasec[o] - phase[o];
                rms = rms + df * df;
            }
        }
        rms = sqrt(rms / nannum);
        for (int64_t b = 0; b < nb; b++) phase[b] = phasec[b];
        sweeps++;
        if (rms < criteria) break;
        counter++;
        if (counter > maxiter) break;
    }
    return sweeps;
}

/* ------------------------------------------------------------------------------------ */
/* beta = atan2(r, z), cos(beta), tan(beta/2) by the C library -- the calls woltsurf.f95:510,      */
/* 516-520 make per Newton step of the W-S secondary; checker of the product's               */
/* pxf_host_cos_atan2.                                                                         */
void orc_cos_atan2(const double *r, const double *z, double *beta, double *cosbeta, double *tanhalf, int64_t num)
{
    for (int64_t i = 0; i < num; i++) {
        double sb;
        beta[i] = atan2(r[i], z[i]);
        sincos(beta[i], &sb, &cosbeta[i]);   /* the reference binary: one sincos for sin(beta), cos(beta) */
        tanhalf[i] = tan(beta[i] / 2);
    }
}

/* theta = atan2(y, x), sin(theta), cos(theta) by the C library (gcc merges the pair into one `sincos` call, as    */
/* flang does in the reference's wsprimaryBack / wssecondaryBack, woltsurf.f95:760-763, 862-865);              */
/* checker of the product's pxf_host_sincos_atan2.                                                             */
void orc_sincos_atan2(const double *y, const double *x, double *theta, double *sintheta, double *costheta, int64_t num)
{
    extern void sincos(double, double *, double *);   /* glibc; what both compilers emit for the sin/cos pair */
    for (int64_t i = 0; i < num; i++) {
        theta[i] = atan2(y[i], x[i]);
        sincos(theta[i], &sintheta[i], &costheta[i]);
    }
}
