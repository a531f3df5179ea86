// moq_core.hip -- error reporting and device queries shared by every C-ABI entry point.
#include <stdarg.h>
#include <stdio.h>

#include "moq_common.h"

namespace moq {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: HIP error %d (%s)", what, (int)e, hipGetErrorString(e));
    return MOQ_ERR_LAUNCH;
  }
  return MOQ_OK;
}

}  // namespace moq

extern "C" int moq_abi_version(void) { return MOQ_ABI_VERSION; }
extern "C" const char* moq_last_error(void) { return moq::g_err; }
extern "C" int moq_device_cu_count(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) return -1;
  return p.multiProcessorCount;
}
