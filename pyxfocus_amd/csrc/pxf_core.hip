// pxf_core.hip -- ABI version, error strings, per-thread HIP error slot, device query.
#include <stdlib.h>
#include <string.h>

#include "pxf_exactdiv.h"
#include "pxf_kernel.h"

namespace pxf {

thread_local int g_last_hip_error = 0;

// Per-device facts, cached per device ORDINAL (a process may drive several GPUs; a CU count read on one must not
// size a reduction launched on another): CU count, and whether the device can run this library's gfx950 code object.
namespace {
constexpr int kMaxDevices = 64;
struct DevInfo {
    int state = 0;  // 0 unknown, 1 usable gfx950, -1 not usable
    int cus = 256;
};
DevInfo g_dev[kMaxDevices];

const DevInfo &dev_info()
{
    static const DevInfo none{-1, 256};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return none;
    DevInfo &d = g_dev[dev];
    if (d.state == 0) {  // (a benign race: every thread would write the same values)
        hipDeviceProp_t prop;
        DevInfo t;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) {
            if (prop.multiProcessorCount > 0) t.cus = prop.multiProcessorCount;
            t.state = strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : -1;
        } else {
            t.state = -1;
        }
        d.cus = t.cus;
        d.state = t.state;
    }
    return d;
}
}  // namespace

int device_cus() { return dev_info().cus; }

bool device_usable() { return dev_info().state == 1; }

int blocks_per_cu_cap()
{
    static int bpc = -1;
    if (bpc < 0) {
        const char *e = getenv("PXF_BPC");
        bpc = e ? atoi(e) : kDefaultBlocksPerCU;
        if (bpc < 0) bpc = 0;
    }
    return bpc;
}

bool use_nt(int64_t num)
{
    static int64_t min_rays = -1;
    if (min_rays < 0) {
        const char *e = getenv("PXF_NT_MIN_RAYS");
        // default: the ten-column bundle (80 B/ray) no longer fits the 256 MiB Infinity Cache
        min_rays = e ? atoll(e) : (256LL << 20) / 80;
    }
    return num >= min_rays;
}

// ---- self-test of pxf_exactdiv.h ----------------------------------------------------------------------------------
// Every thread draws operand tuples (n0, n1, n2, d) from a counter hash -- random mantissas; exponents across the
// fast-path window, in the few binades either side of its two edges, over the whole fp64 range, and the special
// values -- and compares div3_exact / div2_exact / unit_sign_exact with `/` as 64-bit patterns (NaN == NaN).
__device__ __forceinline__ uint64_t splitmix(uint64_t x)
{
    x += 0x9e3779b97f4a7c15ull;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ double selftest_operand(uint64_t h)
{
    const uint64_t mant = h & 0x000fffffffffffffull, sign = (h >> 52) & 1;
    const unsigned kind = (unsigned)(h >> 53) & 15, u = (unsigned)(h >> 57) & 127;
    int e;  // biased exponent
    if (kind < 8) e = 1023 - 300 + (int)(((h >> 40) & 0xfff) % 601);      // inside the window
    else if (kind < 10) e = 1023 - 300 - 4 + (int)(u % 8);                // around its lower edge
    else if (kind < 12) e = 1023 + 300 - 4 + (int)(u % 8);                // around its upper edge
    else if (kind < 14) e = (int)(((h >> 40) & 0xfff) % 2047);            // anywhere (0 = zero / denormal)
    else if (kind == 14) e = (u & 1) ? 0 : 2047;                          // denormal / zero ... inf / NaN
    else return __longlong_as_double((long long)((sign << 63) | ((u & 1) ? 0ull : 0x7ff0000000000000ull)));   // +-0, +-inf
    return __longlong_as_double((long long)((sign << 63) | ((uint64_t)e << 52) | mant));
}
__device__ __forceinline__ bool same_bits(double a, double b)
{
    return __double_as_longlong(a) == __double_as_longlong(b) || (a != a && b != b);
}

__global__ __launch_bounds__(kBlock, 2) void selftest_div_kernel(int64_t ntuples, uint64_t seed,
                                                                 unsigned long long *__restrict__ mismatches)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    unsigned long long bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < ntuples; i += stride) {
        const uint64_t h = splitmix(seed + 4 * (uint64_t)i);
        double n0 = selftest_operand(h), n1 = selftest_operand(splitmix(h + 1)), n2 = selftest_operand(splitmix(h + 2));
        double d = selftest_operand(splitmix(h + 3));
        if ((h >> 61) == 0) d = sqrt(n0 * n0 + n1 * n1 + n2 * n2);   // one tuple in eight: a vector over its length
        double q0, q1, q2, p0, p1;
        div3_exact(n0, n1, n2, d, q0, q1, q2);
        div2_exact(n1, n2, d, p0, p1);
        // (the reference quotients must not be folded into the same code: opaque copies of the operands)
        double a0 = n0, a1 = n1, a2 = n2, ad = d;
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(ad));
        const double r0 = a0 / ad, r1 = a1 / ad, r2 = a2 / ad;
        bad += !same_bits(q0, r0) + !same_bits(q1, r1) + !same_bits(q2, r2) + !same_bits(p0, r1) + !same_bits(p1, r2);
        bad += !same_bits(unit_sign_exact(n0), a0 / __builtin_fabs(a0));
    }
    if (bad) atomicAdd(mismatches, bad);
}

}  // namespace pxf

extern "C" {

int pxf_selftest_exact_division(int64_t ntuples, uint64_t seed, uint64_t *mismatches_dev, pxf_stream_t stream)
{
    if (ntuples < 0 || !mismatches_dev) return PXF_EINVAL;
    if (!pxf::device_usable()) return PXF_ENODEV;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(mismatches_dev, 0, sizeof(uint64_t), s);
    if (e != hipSuccess) return pxf::hip_fail(e);
    if (ntuples == 0) return PXF_OK;
    hipLaunchKernelGGL(pxf::selftest_div_kernel, dim3(pxf::grid_capped((ntuples + 15) / 16)), dim3(pxf::kBlock), 0, s, ntuples,
                       seed, reinterpret_cast<unsigned long long *>(mismatches_dev));
    e = hipGetLastError();
    return e == hipSuccess ? PXF_OK : pxf::hip_fail(e);
}

int pxf_abi_version(void) { return PXF_ABI_VERSION; }

int pxf_last_hip_error(void) { return pxf::g_last_hip_error; }

int pxf_device_check(void) { return pxf::device_usable() ? PXF_OK : PXF_ENODEV; }

const char *pxf_strerror(int code)
{
    switch (code) {
        case PXF_OK: return "ok";
        case PXF_EINVAL: return "invalid argument (NULL column, negative count or bad table)";
        case PXF_EHIP: return "HIP runtime error (see pxf_last_hip_error)";
        case PXF_ENODEV: return "no usable gfx950 device";
        case PXF_ERANGE: return "parameter outside the supported range";
        default: return "unknown pxf error";
    }
}

}  // extern "C"
