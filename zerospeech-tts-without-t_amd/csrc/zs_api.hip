// zs_api.hip -- error state and version of libzs_amd.so (see include/zs_amd.h).
#include <stdarg.h>

#include "zs_common.h"

static thread_local char g_err[512] = "";

void zs_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int zs_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    zs_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return ZS_ELAUNCH;
  }
  return ZS_OK;
}

extern "C" int zs_abi_version(void) { return ZS_ABI_VERSION; }
extern "C" const char* zs_last_error(void) { return g_err; }
