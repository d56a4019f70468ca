// capi.hip -- identification and error strings of libmdt_hip.so (include/mdt_hip.h).
#include <hip/hip_runtime.h>
#include "mdt_hip.h"

extern "C" {

const char *mdt_version(void) { return "mdt_hip 0.1 (gfx950, HIP, C ABI)"; }

const char *mdt_error_string(int code)
{
    switch (code) {
        case MDT_OK: return "ok";
        case MDT_ERR_INVALID_ARGUMENT: return "invalid argument";
        case MDT_ERR_WORKSPACE_TOO_SMALL: return "workspace missing or too small";
        case MDT_ERR_LAUNCH_FAILED: return "kernel launch failed";
        case MDT_ERR_UNSUPPORTED: return "shape not supported by this build";
        default: return "unknown error code";
    }
}

}  // extern "C"
