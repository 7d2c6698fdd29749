// Library introspection entry points of the C ABI (include/snap_hip.h).
#include "common.h"

extern "C" int snap_abi_version(void) { return 22; }

extern "C" const char* snap_build_arch(void) { return "gfx950"; }

extern "C" const char* snap_status_string(int status) {
  switch (status) {
    case SNAP_OK: return "ok";
    case SNAP_ERR_BAD_SHAPE: return "bad or inconsistent shape";
    case SNAP_ERR_UNSUPPORTED: return "unsupported option";
    case SNAP_ERR_NULL: return "required pointer is NULL";
    case SNAP_ERR_LAUNCH: return "HIP launch error";
    case SNAP_ERR_WORKSPACE: return "workspace too small";
    default: return "unknown status";
  }
}
