// ABI version and error text of libgm3d_hip.so (see include/gm3d.h).
#include "common.hpp"

extern "C" int gm3d_abi_version(void) { return 1; }

extern "C" const char* gm3d_strerror(int code) {
    switch (code) {
        case GM3D_OK: return "ok";
        case GM3D_EINVAL: return "invalid argument (null pointer, non-positive size or inconsistent shape)";
        case GM3D_EUNSUPPORTED: return "shape outside the limits documented in include/gm3d.h";
        case GM3D_ELAUNCH: return "HIP kernel launch failed";
        default: return "unknown gm3d error code";
    }
}
