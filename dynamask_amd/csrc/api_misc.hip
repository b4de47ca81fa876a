// Library identity / error strings.
#include "common.h"

extern "C" const char* dm_error_string(int code) {
  switch (code) {
    case DM_OK: return "ok";
    case DM_ERR_INVALID_ARG: return "invalid argument (shape, null pointer or unsupported parameter)";
    case DM_ERR_LAUNCH: return "HIP kernel launch failed";
    case DM_ERR_UNSUPPORTED: return "configuration not supported by libdynamask_hip";
    default: return "unknown error";
  }
}

extern "C" int dm_abi_version(void) { return 3; }
