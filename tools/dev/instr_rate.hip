// instr_rate.hip -- issue cost of the fp64 instructions the bit-exact kernels are made of, on gfx950.
// Every wave runs REPS x 16 independent copies of ONE instruction (inline asm, no dependence between consecutive
// copies beyond a 16-register rotation), 8 waves per SIMD on every CU, so the result is the pipe's issue rate:
// cycles per wave-instruction per SIMD = elapsed cycles x SIMDs / (waves x instructions per wave).
// build: hipcc --offload-arch=gfx950 -O2 -o instr_rate tools/dev/instr_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#define REPS 2000

#define KERNEL(name, ASM)                                                                         \
    __global__ __launch_bounds__(256) void k_##name(double *out, double seed)                    \
    {                                                                                             \
        double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 1.0000001, c = 0.5; \
        float f0 = 1.5f, f1 = 2.5f;                                                               \
        int i0 = 0;                                                                               \
        for (int r = 0; r < REPS; r++) {                                                          \
            _Pragma("unroll") for (int u = 0; u < 4; u++) {                                       \
                asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(f0), "+v"(f1), "+v"(i0) : "v"(b), "v"(c) : "vcc", "scc"); \
            }                                                                                     \
        }                                                                                         \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + f0 + f1 + i0;                   \
    }

// four instructions per asm block (x4 unroll = 16 per trip)
KERNEL(fma, "v_fma_f64 %0, %0, %7, %8\n v_fma_f64 %1, %1, %7, %8\n v_fma_f64 %2, %2, %7, %8\n v_fma_f64 %3, %3, %7, %8")
KERNEL(mul, "v_mul_f64 %0, %0, %7\n v_mul_f64 %1, %1, %7\n v_mul_f64 %2, %2, %7\n v_mul_f64 %3, %3, %7")
KERNEL(add, "v_add_f64 %0, %0, %7\n v_add_f64 %1, %1, %7\n v_add_f64 %2, %2, %7\n v_add_f64 %3, %3, %7")
KERNEL(rcp, "v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3")
KERNEL(rsq, "v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3")
KERNEL(sqrt, "v_sqrt_f64 %0, %0\n v_sqrt_f64 %1, %1\n v_sqrt_f64 %2, %2\n v_sqrt_f64 %3, %3")
KERNEL(divscale, "v_div_scale_f64 %0, vcc, %0, %7, %0\n v_div_scale_f64 %1, vcc, %1, %7, %1\n v_div_scale_f64 %2, vcc, %2, %7, %2\n v_div_scale_f64 %3, vcc, %3, %7, %3")
KERNEL(divfmas, "v_div_fmas_f64 %0, %0, %7, %8\n v_div_fmas_f64 %1, %1, %7, %8\n v_div_fmas_f64 %2, %2, %7, %8\n v_div_fmas_f64 %3, %3, %7, %8")
KERNEL(divfixup, "v_div_fixup_f64 %0, %0, %7, %8\n v_div_fixup_f64 %1, %1, %7, %8\n v_div_fixup_f64 %2, %2, %7, %8\n v_div_fixup_f64 %3, %3, %7, %8")
KERNEL(cvt_f32_f64, "v_cvt_f32_f64 %4, %0\n v_cvt_f32_f64 %5, %1\n v_cvt_f32_f64 %4, %2\n v_cvt_f32_f64 %5, %3")
KERNEL(cvt_f64_f32, "v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %4\n v_cvt_f64_f32 %3, %5")
KERNEL(cmp_class, "v_cmp_class_f64 vcc, %0, %6\n v_cmp_class_f64 vcc, %1, %6\n v_cmp_class_f64 vcc, %2, %6\n v_cmp_class_f64 vcc, %3, %6")
KERNEL(cmp_gt, "v_cmp_gt_f64 vcc, %0, %7\n v_cmp_gt_f64 vcc, %1, %7\n v_cmp_gt_f64 vcc, %2, %7\n v_cmp_gt_f64 vcc, %3, %7")
KERNEL(mov_b64, "v_mov_b64 %0, %7\n v_mov_b64 %1, %7\n v_mov_b64 %2, %7\n v_mov_b64 %3, %7")
KERNEL(mov_b32, "v_mov_b32 %4, %5\n v_mov_b32 %5, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %4")
KERNEL(cndmask, "v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %4, vcc\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %4, vcc")
KERNEL(ldexp, "v_ldexp_f64 %0, %0, %6\n v_ldexp_f64 %1, %1, %6\n v_ldexp_f64 %2, %2, %6\n v_ldexp_f64 %3, %3, %6")
KERNEL(frexp_exp, "v_frexp_exp_i32_f64 %6, %0\n v_frexp_exp_i32_f64 %6, %1\n v_frexp_exp_i32_f64 %6, %2\n v_frexp_exp_i32_f64 %6, %3")
KERNEL(max_f64, "v_max_f64 %0, %0, %7\n v_max_f64 %1, %1, %7\n v_max_f64 %2, %2, %7\n v_max_f64 %3, %3, %7")
KERNEL(fma_f32, "v_fma_f32 %4, %4, %5, %5\n v_fma_f32 %5, %5, %4, %4\n v_fma_f32 %4, %4, %5, %5\n v_fma_f32 %5, %5, %4, %4")
KERNEL(add_u32, "v_add_u32 %6, %6, %6\n v_add_u32 %6, %6, %6\n v_add_u32 %6, %6, %6\n v_add_u32 %6, %6, %6")
KERNEL(and_b32, "v_and_b32 %6, %6, %6\n v_and_b32 %6, %6, %6\n v_and_b32 %6, %6, %6\n v_and_b32 %6, %6, %6")

struct Entry { const char *name; void (*fn)(double *, double); };

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const int waves_per_simd = 8, blocks = cus * waves_per_simd;      // 256 threads = 4 waves = one per SIMD
    double *out;
    hipMalloc(&out, (size_t)blocks * 256 * sizeof(double));
    const double ghz = prop.clockRate * 1e-6;
#define E(n) {#n, k_##n}
    Entry es[] = {E(fma), E(mul), E(add), E(rcp), E(rsq), E(sqrt), E(divscale), E(divfmas), E(divfixup), E(cvt_f32_f64),
                  E(cvt_f64_f32), E(cmp_class), E(cmp_gt), E(mov_b64), E(mov_b32), E(cndmask), E(ldexp), E(frexp_exp),
                  E(max_f64), E(fma_f32), E(add_u32), E(and_b32)};
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_GHz\": %.3f, \"waves_per_simd\": %d, \"note\": \"cycles per wave-instruction "
           "per SIMD at the reported peak clock (the real clock under load is lower, so read the numbers relative to fma)\", \"cycles\": {",
           prop.name, cus, ghz, waves_per_simd);
    fflush(stdout);
    for (size_t e = 0; e < sizeof(es) / sizeof(es[0]); e++) {
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        hipLaunchKernelGGL(es[e].fn, dim3(blocks), dim3(256), 0, 0, out, 1.0);   // warm-up
        hipEventRecord(a);
        hipLaunchKernelGGL(es[e].fn, dim3(blocks), dim3(256), 0, 0, out, 1.0);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        const double insts_per_wave = (double)REPS * 16;
        const double cyc = ms * 1e-3 * ghz * 1e9 / (waves_per_simd * insts_per_wave);
        printf("%s\"%s\": %.2f", e ? ", " : "", es[e].name, cyc);
        fflush(stdout);
    }
    printf("}}\n");
    return 0;
}
