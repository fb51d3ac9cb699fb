// C-ABI housekeeping: error string, version / architecture probes.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

extern "C" void nv_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* nv_last_error(void) { return g_err; }

extern "C" int nv_version(void) { return 1; }

// 1 when the current HIP device is gfx950 (MI355X), 0 otherwise, negative on HIP error.
extern "C" int nv_arch_ok(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { nv_set_error("nv_arch_ok: no HIP device"); return -2; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { nv_set_error("nv_arch_ok: hipGetDeviceProperties failed"); return -2; }
  return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}
