// bw_probe.hip -- dev tool: which streaming shape gets the most HBM bandwidth on MI355X for the
// SoA column pattern of the ray kernels (R columns read, W columns written, fp64).
// Variants: rays per thread (V = 2, 4), non-temporal loads/stores, workgroups per CU, block size.
// Build: hipcc --offload-arch=gfx950 -O3 -o bw_probe tools/bw_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));

struct Ptrs {
    double *p[10];
};

template <int R, int W, int V, bool NTL, bool NTS>
__global__ void probe(Ptrs c, long nvec, double add)
{
    const long stride = (long)gridDim.x * blockDim.x;
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
        d2 r[R][V / 2];
#pragma unroll
        for (int k = 0; k < R; k++)
#pragma unroll
            for (int j = 0; j < V / 2; j++) {
                const d2 *p = reinterpret_cast<const d2 *>(c.p[k]) + v * (V / 2) + j;
                r[k][j] = NTL ? __builtin_nontemporal_load(p) : *p;
            }
#pragma unroll
        for (int k = 0; k < W; k++)
#pragma unroll
            for (int j = 0; j < V / 2; j++) {
                d2 o = r[k][j] + r[(k + 1) % R][j] * add;
                d2 *p = reinterpret_cast<d2 *>(c.p[k]) + v * (V / 2) + j;
                if (NTS)
                    __builtin_nontemporal_store(o, p);
                else
                    *p = o;
            }
    }
}

// transform-shaped kernel that touches its nine columns as three sequential groups of three
// (position, direction, normal are independent): fewer concurrent DRAM streams per workgroup
template <bool NT>
__global__ void probe_grouped(Ptrs c, long nvec, double add)
{
    const long stride = (long)gridDim.x * blockDim.x;
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
#pragma unroll
        for (int g = 0; g < 3; g++) {
            d2 r[3];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const d2 *p = reinterpret_cast<const d2 *>(c.p[3 * g + k]) + v;
                r[k] = NT ? __builtin_nontemporal_load(p) : *p;
            }
#pragma unroll
            for (int k = 0; k < 3; k++) {
                d2 o = r[k] + r[(k + 1) % 3] * add;
                d2 *p = reinterpret_cast<d2 *>(c.p[3 * g + k]) + v;
                if (NT) __builtin_nontemporal_store(o, p); else *p = o;
            }
        }
    }
}

template <bool NT>
static void run_grouped(Ptrs c, long n, int cus, int reps)
{
    const long nvec = n / 2;
    int grid = (int)((nvec + 255) / 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL((probe_grouped<NT>), dim3(grid), dim3(256), 0, 0, c, nvec, 1e-9);
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((probe_grouped<NT>), dim3(grid), dim3(256), 0, 0, c, nvec, 1e-9);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("grouped3x3 nt%d full grid : %7.3f ms  %7.1f GB/s\n", (int)NT, ms, 18 * 8.0 * n / (ms * 1e-3) / 1e9);
}

template <int R, int W, int V, bool NTL, bool NTS>
static void run(const char *name, Ptrs c, long n, int block, int bpc, int cus, int reps)
{
    const long nvec = n / V;
    long want = (nvec + block - 1) / block;
    long cap = bpc > 0 ? (long)cus * bpc : want;
    int grid = (int)(want < cap ? want : cap);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL((probe<R, W, V, NTL, NTS>), dim3(grid), dim3(block), 0, 0, c, nvec, 1e-9);
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((probe<R, W, V, NTL, NTS>), dim3(grid), dim3(block), 0, 0, c, nvec, 1e-9);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    double gbs = (double)(R + W) * 8.0 * n / (ms * 1e-3) / 1e9;
    printf("%-10s R%d W%d V%d ntl%d nts%d block%4d bpc%3d grid%7d : %7.3f ms  %7.1f GB/s  (%.1f%% of 8 TB/s)\n", name, R, W,
           V, (int)NTL, (int)NTS, block, bpc, grid, ms, gbs, gbs / 80.0);
}

int main(int argc, char **argv)
{
    long n = argc > 1 ? atol(argv[1]) : 100000000L;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    int cus = prop.multiProcessorCount;
    printf("device %s CUs %d n %ld\n", prop.name, cus, n);
    Ptrs c;
    for (int k = 0; k < 10; k++) {
        hipMalloc(&c.p[k], n * sizeof(double));
        hipMemset(c.p[k], 0, n * sizeof(double));
    }
    const int reps = 5;
    int blocks[] = {256, 512, 1024};
    int bpcs[] = {4, 8, 16, 32, 0};
    for (int rep = 0; rep < 3; rep++) {   // interleaved: plain 9+9 vs grouped 3x(3+3), full grid
        run<9, 9, 2, false, false>("transform", c, n, 256, 0, cus, reps);
        run_grouped<false>(c, n, cus, reps);
        run<9, 9, 2, true, true>("transform", c, n, 256, 0, cus, reps);
        run_grouped<true>(c, n, cus, reps);
    }
    return 0;
    // transform shape
    for (int b : blocks)
        for (int bpc : bpcs) {
            if (b != 256 && bpc != 8 && bpc != 0) continue;
            run<9, 9, 2, false, false>("transform", c, n, b, bpc, cus, reps);
        }
    run<9, 9, 2, true, false>("transform", c, n, 256, 8, cus, reps);
    run<9, 9, 2, false, true>("transform", c, n, 256, 8, cus, reps);
    run<9, 9, 2, true, true>("transform", c, n, 256, 8, cus, reps);
    run<9, 9, 2, true, true>("transform", c, n, 256, 0, cus, reps);
    run<9, 9, 4, false, false>("transform", c, n, 256, 8, cus, reps);
    run<9, 9, 4, true, true>("transform", c, n, 256, 8, cus, reps);
    run<9, 9, 4, false, false>("transform", c, n, 256, 0, cus, reps);
    // flat / wolter shape (6 read, 6 written) and reflect (6 read, 3 written), copy (1,1)
    for (int bpc : bpcs) run<6, 6, 2, false, false>("wolter", c, n, 256, bpc, cus, reps);
    run<6, 6, 2, true, true>("wolter", c, n, 256, 8, cus, reps);
    run<6, 6, 4, false, false>("wolter", c, n, 256, 8, cus, reps);
    run<6, 6, 4, true, true>("wolter", c, n, 256, 8, cus, reps);
    for (int bpc : bpcs) run<6, 3, 2, false, false>("reflect", c, n, 256, bpc, cus, reps);
    run<6, 3, 2, true, true>("reflect", c, n, 256, 8, cus, reps);
    run<6, 3, 4, false, false>("reflect", c, n, 256, 8, cus, reps);
    run<6, 3, 4, true, true>("reflect", c, n, 256, 8, cus, reps);
    run<2, 1, 2, false, false>("copy-ish", c, n, 256, 8, cus, reps);
    run<2, 1, 4, false, false>("copy-ish", c, n, 256, 8, cus, reps);
    run<2, 1, 4, true, true>("copy-ish", c, n, 256, 8, cus, reps);
    run<2, 1, 4, false, false>("copy-ish", c, n, 256, 0, cus, reps);
    return 0;
}
