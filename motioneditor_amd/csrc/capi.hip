// Library-level entry points of libmotioned.so: ABI version, last-error string, device probe.
#include "me_common.h"
#include "../../include/motioned.h"
#include <stdio.h>
#include <string.h>

namespace {
thread_local char g_err[512] = "";
thread_local char g_kernel[96] = "";
}

// name of the device kernel the last compute entry point of this thread launched (bench.py: per-kernel roofline)
extern "C" void me_set_kernel(const char* name) {
  strncpy(g_kernel, name ? name : "", sizeof(g_kernel) - 1);
  g_kernel[sizeof(g_kernel) - 1] = 0;
}
extern "C" const char* me_last_kernel(void) { return g_kernel; }

extern "C" void me_set_error(const char* msg) {
  strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
}

// "<what>: <hipGetErrorName>: <hipGetErrorString>"
extern "C" void me_set_hip_error(const char* what, int err) {
  snprintf(g_err, sizeof(g_err), "%s: %s: %s", what ? what : "", hipGetErrorName((hipError_t)err), hipGetErrorString((hipError_t)err));
}

extern "C" int me_abi_version(void) { return ME_ABI_VERSION; }

extern "C" const char* me_last_error(void) { return g_err; }

extern "C" int me_device_info(int* cus, int* lds_bytes, char* arch, int arch_len) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { me_set_error("me_device_info: no HIP device"); return ME_EHIP; }
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) { me_set_error("me_device_info: hipGetDeviceProperties failed"); return ME_EHIP; }
  if (cus) *cus = p.multiProcessorCount;
  if (lds_bytes) *lds_bytes = (int)p.maxSharedMemoryPerMultiProcessor;
  if (arch && arch_len > 0) {
    strncpy(arch, p.gcnArchName, arch_len - 1);
    arch[arch_len - 1] = 0;
  }
  return ME_OK;
}
