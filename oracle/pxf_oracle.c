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
    const double cx = cos(rx), sx = sin(rx), cy = cos(ry), sy = sin(ry), cz = cos(rz), sz = sin(rz);
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
    const double cx = cos(-rx), sx = sin(-rx), cy = cos(-ry), sy = sin(-ry), cz = cos(-rz), sz = sin(-rz);
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
    const double sl = tan(tg), ctg = cos(tg), stg = sin(tg);
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
