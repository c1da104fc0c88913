// Library-level entry points: ABI version, error text, last HIP error.
#include "tp3d_common.h"

namespace tp3d {
static thread_local int g_last_hip_error = 0;
void set_last_hip_error(hipError_t e) { g_last_hip_error = (int)e; }
}  // namespace tp3d

TP3D_EXPORT int tp3d_abi_version(void) { return TP3D_ABI_VERSION; }

TP3D_EXPORT int tp3d_last_hip_error(void) { return tp3d::g_last_hip_error; }

TP3D_EXPORT const char *tp3d_strerror(int code)
{
    switch (code) {
        case TP3D_OK: return "ok";
        case TP3D_E_BADARG: return "bad argument (negative or inconsistent size, or null pointer)";
        case TP3D_E_LAUNCH: return "HIP launch failed (see tp3d_last_hip_error)";
        case TP3D_E_UNSORTED: return "batch vector is not sorted";
        case TP3D_E_TOOBIG: return "size exceeds the kernel's index range";
        default: return "unknown error";
    }
}
