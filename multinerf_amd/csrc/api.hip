// Library-level entry points: error string, ABI version, device info.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "common.h"

static thread_local char g_err[512] = "";

void mnr_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* mnr_last_error(void) { return g_err; }

extern "C" int mnr_abi_version(void) { return 20; }

extern "C" int mnr_device_info(int device, int* cu_count, int* lds_bytes_per_cu, char* arch_name,
                               int arch_name_len) {
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) {
    mnr_set_error("hipGetDeviceProperties(%d): %s", device, hipGetErrorString(e));
    return MNR_ERR_HIP;
  }
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (lds_bytes_per_cu) *lds_bytes_per_cu = (int)prop.maxSharedMemoryPerMultiProcessor;
  if (arch_name && arch_name_len > 0) {
    strncpy(arch_name, prop.gcnArchName, arch_name_len - 1);
    arch_name[arch_name_len - 1] = 0;
  }
  return MNR_OK;
}
