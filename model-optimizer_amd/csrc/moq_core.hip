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

// A kernel that needs more than 64 KiB of dynamic LDS must opt in per device (hipFuncAttributeMaxDynamicSharedMemorySize).
// A device or driver that refuses (anything but gfx950's 160 KiB CUs) would otherwise surface as an opaque "invalid
// value" at the launch: the refusal is remembered here and reported by the check_launch that follows the launch.
static thread_local char g_optin[256] = "";

bool lds_opt_in(const void* kernel, int bytes, const char* kernel_name) {
  hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) return true;
  (void)hipGetLastError();
  snprintf(g_optin, sizeof(g_optin),
           "%s needs %d bytes of dynamic LDS per workgroup (gfx950: 160 KiB per CU) and the device refused the opt-in: "
           "HIP error %d (%s)", kernel_name, bytes, (int)e, hipGetErrorString(e));
  return false;
}

int check_launch(const char* what) {
  if (g_optin[0]) {
    (void)hipGetLastError();  // the launch that followed failed for the reason already recorded
    set_error("%s: %s", what, g_optin);
    g_optin[0] = 0;
    return MOQ_ERR_UNSUPPORTED;
  }
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
