// api.hip -- library-level entry points of libl3d_hip.so (version, status strings, last HIP error).
#include "common.h"

thread_local int g_l3d_last_hip_error = 0;

extern "C" int l3d_version(void) { return 100; /* 0.1.0 */ }

extern "C" const char *l3d_status_string(int status)
{
    switch (status) {
        case L3D_OK: return "ok";
        case L3D_ERR_INVALID_ARG: return "invalid argument (null pointer, non-positive size or k out of range)";
        case L3D_ERR_UNSUPPORTED: return "shape not supported by the gfx950 kernels";
        case L3D_ERR_LAUNCH: return "HIP kernel launch failed (see l3d_last_hip_error)";
        default: return "unknown status";
    }
}

extern "C" int l3d_last_hip_error(void) { return g_l3d_last_hip_error; }
