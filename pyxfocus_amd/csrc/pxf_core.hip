// pxf_core.hip -- ABI version, error strings, per-thread HIP error slot, device query.
#include <stdlib.h>
#include <string.h>

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

bool xcd_remap_enabled()
{
    static int on = -1;
    if (on < 0) {
        const char *e = getenv("PXF_XCD_REMAP");
        on = e ? (atoi(e) != 0) : 0;  // measured on MI355X: the XCD-contiguous mapping LOSES 4 %
    }
    return on != 0;
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

}  // namespace pxf

extern "C" {

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
