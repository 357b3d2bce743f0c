/* abi_smoke.c -- the C ABI of include/pxf.h used from plain C with the HIP runtime only (no Python,
 * no torch): allocates a small bundle with hipMalloc, runs the config-2 two-bounce chain through the
 * pxf_* entry points on a non-default stream, runs the same chain as one fused program, compares
 * the two bit for bit, exercises the error returns, and prints the focal RMS.
 * Build:  gcc -std=c11 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tests/c/abi_smoke.c \
 *             -Lpyxfocus_amd/csrc -lpxf_hip -L/opt/rocm/lib -lamdhip64 -lm -o build/abi_smoke
 * (tests/test_gpu_cabi.py builds and runs it on the GPU box). */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pxf.h"

#define CHECK(x)                                                          \
    do {                                                                  \
        int rc_ = (x);                                                    \
        if (rc_ != 0) {                                                   \
            fprintf(stderr, "%s failed: %d (%s)\n", #x, rc_, pxf_strerror(rc_)); \
            return 1;                                                     \
        }                                                                 \
    } while (0)

int main(void)
{
    const int64_t n = 100001; /* odd: exercises the scalar tail */
    const double r0 = 220., z0 = 8400., L = 100.;
    double *col[10], *col2[10];
    hipStream_t s;
    if (hipStreamCreate(&s) != hipSuccess) {
        fprintf(stderr, "no HIP device\n");
        return 2;
    }
    for (int c = 0; c < 10; c++) {
        if (hipMalloc((void **)&col[c], n * sizeof(double)) != hipSuccess) return 2;
        if (hipMalloc((void **)&col2[c], n * sizeof(double)) != hipSuccess) return 2;
    }
    double p, d, e;
    pxf_wolter_params(r0, z0, 1., &p, &d, &e);
    const double rin = sqrt(p * p + 2 * p * z0 + 4 * e * e * p * d / (e * e - 1));
    const double rout = sqrt(p * p + 2 * p * (z0 + L) + 4 * e * e * p * d / (e * e - 1));
    if (pxf_abi_version() != PXF_ABI_VERSION) return 3;
    CHECK(pxf_source(PXF_SRC_ANNULUS, col, n, rin, rout, 0., -1., 7, 0, s));
    for (int c = 0; c < 10; c++) hipMemcpyAsync(col2[c], col[c], n * sizeof(double), hipMemcpyDeviceToDevice, s);

    /* one launch per call */
    double *x = col[1], *y = col[2], *z = col[3], *l = col[4], *m = col[5], *nn = col[6], *ux = col[7],
           *uy = col[8], *uz = col[9];
    CHECK(pxf_transform(x, y, z, l, m, nn, ux, uy, uz, n, 0., 0., z0 + 2 * L, 0., 0., 0., NULL, s));
    CHECK(pxf_wolterprimary(x, y, z, l, m, nn, ux, uy, uz, n, r0, z0, 1., NULL, s));
    CHECK(pxf_reflect(l, m, nn, ux, uy, uz, n, NULL, s));
    CHECK(pxf_woltersecondary(x, y, z, l, m, nn, ux, uy, uz, n, r0, z0, 1., NULL, s));
    CHECK(pxf_reflect(l, m, nn, ux, uy, uz, n, NULL, s));
    CHECK(pxf_flat(x, y, z, l, m, nn, ux, uy, uz, n, NULL, s));

    /* the same as one fused program */
    pxf_op_t ops[6];
    memset(ops, 0, sizeof(ops));
    ops[0].code = PXF_OP_TRANSFORM;
    ops[0].a[2] = z0 + 2 * L;
    ops[1].code = PXF_OP_WOLTERPRIMARY;
    ops[1].a[0] = r0; ops[1].a[1] = z0; ops[1].a[2] = 1.;
    ops[2].code = PXF_OP_REFLECT;
    ops[3].code = PXF_OP_WOLTERSECONDARY;
    ops[3].a[0] = r0; ops[3].a[1] = z0; ops[3].a[2] = 1.;
    ops[4].code = PXF_OP_REFLECT;
    ops[5].code = PXF_OP_FLAT;
    uint32_t lm, sm;
    int32_t dropped;
    CHECK(pxf_program_plan(ops, 6, &lm, &sm, &dropped));
    if (lm != 0x7e || sm != 0x3fe || dropped != 1) {
        fprintf(stderr, "unexpected plan %x %x %d\n", lm, sm, dropped);
        return 4;
    }
    CHECK(pxf_program_run(ops, 6, col2, n, s));

    /* focal-plane moments on the device */
    void *ws;
    double *mom;
    hipMalloc(&ws, pxf_reduce_workspace_bytes());
    hipMalloc((void **)&mom, PXF_NMOM * sizeof(double));
    CHECK(pxf_moments(x, y, l, m, nn, NULL, n, mom, ws, s));
    double hm[PXF_NMOM];
    hipMemcpyAsync(hm, mom, sizeof(hm), hipMemcpyDeviceToHost, s);
    double *ha = (double *)malloc(n * sizeof(double)), *hb = (double *)malloc(n * sizeof(double));
    int bad = 0;
    for (int c = 1; c < 10; c++) {
        hipMemcpyAsync(ha, col[c], n * sizeof(double), hipMemcpyDeviceToHost, s);
        hipMemcpyAsync(hb, col2[c], n * sizeof(double), hipMemcpyDeviceToHost, s);
        hipStreamSynchronize(s);
        if (memcmp(ha, hb, n * sizeof(double)) != 0) bad++;
    }
    hipStreamSynchronize(s);
    const double cx = hm[1] / hm[0], cy = hm[2] / hm[0];
    double *cm;
    hipMalloc((void **)&cm, 4 * sizeof(double));
    CHECK(pxf_central_moments(x, y, NULL, n, cx, cy, cm, ws, s));
    double hc[4];
    hipMemcpy(hc, cm, sizeof(hc), hipMemcpyDeviceToHost);
    const double rms = sqrt(hc[1] / hc[0]);

    /* exact median radius in one pass over x, y (hpd = 2 * median), cross-checked on the host */
    void *kws;
    double *kout;
    int32_t *kok;
    hipMalloc(&kws, pxf_radial_kth_workspace_bytes(n));
    hipMalloc((void **)&kout, 2 * sizeof(double));
    hipMalloc((void **)&kok, sizeof(int32_t));
    const int64_t kk[2] = {(n - 1) / 2, n / 2};
    CHECK(pxf_radial_kth(x, y, cx, cy, 0, n, kk, 2, kout, kok, kws, s));
    double hk[2];
    int32_t hok;
    hipMemcpyAsync(hk, kout, sizeof(hk), hipMemcpyDeviceToHost, s);
    hipMemcpyAsync(&hok, kok, sizeof(hok), hipMemcpyDeviceToHost, s);
    hipMemcpyAsync(ha, col[1], n * sizeof(double), hipMemcpyDeviceToHost, s);
    hipMemcpyAsync(hb, col[2], n * sizeof(double), hipMemcpyDeviceToHost, s);
    hipStreamSynchronize(s);
    int64_t below = 0, equal = 0;
    for (int64_t i = 0; i < n; i++) {
        const double dx = ha[i] - cx, dy = hb[i] - cy, r = sqrt(dx * dx + dy * dy);
        below += r < hk[0];
        equal += r == hk[0];
    }
    const int median_ok = hok == 1 && below <= kk[0] && kk[0] < below + equal;

    /* wavefront sensing: bin the focal-plane slopes on a 16 x 16 lenslet grid, reconstruct */
    enum { XD = 16, YD = 16 };
    double *gx, *gy, *gp, *gpc;
    void *bws;
    int32_t *sweeps;
    hipMalloc((void **)&gx, XD * YD * sizeof(double));
    hipMalloc((void **)&gy, XD * YD * sizeof(double));
    hipMalloc((void **)&gp, XD * YD * sizeof(double));
    hipMalloc((void **)&gpc, XD * YD * sizeof(double));
    hipMalloc(&bws, pxf_southwellbin_workspace_bytes(XD, YD));
    hipMalloc((void **)&sweeps, sizeof(int32_t));
    CHECK(pxf_southwellbin(x, y, l, m, n, 4. * rms / XD * 2., gx, gy, gp, XD, YD, bws, s));
    CHECK(pxf_reconstruct(gy, gx, XD, YD, 1e-12, 1., gp, gpc, 1000, sweeps, s));
    int32_t hsw;
    double hg[XD * YD];
    hipMemcpyAsync(&hsw, sweeps, sizeof(hsw), hipMemcpyDeviceToHost, s);
    hipMemcpyAsync(hg, gpc, sizeof(hg), hipMemcpyDeviceToHost, s);
    hipStreamSynchronize(s);
    int valid = 0;
    for (int i = 0; i < XD * YD; i++) valid += hg[i] != 100.;
    printf("median_ok=%d hpd_mm=%.6e lenslets_valid=%d sweeps=%d\n", median_ok, hk[0] + hk[1], valid, (int)hsw);
    if (!median_ok || valid < 4 || hsw < 1) return 7;

    /* error returns: no exception, no crash */
    int e1 = pxf_flat(NULL, y, z, l, m, nn, ux, uy, uz, n, NULL, s);
    int e2 = pxf_flat(x, y, z, l, m, nn, ux, uy, uz, -1, NULL, s);
    int e3 = pxf_flat(NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, 0, NULL, s);
    ops[0].code = 999;
    int e4 = pxf_program_run(ops, 6, col2, n, s);
    int32_t ro[2] = {3, 40}, ao[2] = {1, 0};
    double cf[2] = {1., 1.};
    int e5 = pxf_tracezern(x, y, z, l, m, nn, ux, uy, uz, n, cf, ro, ao, 2, 50., NULL, s);
    printf("n=%lld fused_mismatch_columns=%d rms_centroid_mm=%.6e errs=%d,%d,%d,%d,%d\n", (long long)n, bad, rms, e1,
           e2, e3, e4, e5);
    if (bad || !(rms > 1e-7 && rms < 1e-4)) return 5;
    if (e1 != PXF_EINVAL || e2 != PXF_EINVAL || e3 != PXF_OK || e4 != PXF_EINVAL || e5 != PXF_ERANGE) return 6;
    printf("abi_smoke ok\n");
    return 0;
}
