// pxf_core.hip -- ABI version, error strings, per-thread HIP error slot, device query.
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
