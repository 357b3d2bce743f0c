// pxf_core.hip -- ABI version, error strings, per-thread HIP error slot, device query.
#include <stdlib.h>

#include "pxf_kernel.h"

namespace pxf {

thread_local int g_last_hip_error = 0;

int device_cus()
{
    // cached per process; a ray bundle lives on one device and MI355X nodes are homogeneous
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
            prop.multiProcessorCount > 0)
            cus = prop.multiProcessorCount;
        else
            cus = 256;  // MI355X
    }
    return cus;
}

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
