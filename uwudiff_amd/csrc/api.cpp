// Error plumbing + version for libuwu_hip.so.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/uwu_hip.h"

static thread_local char g_err[512] = "";

void uwu_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* uwu_last_error(void) { return g_err; }
extern "C" int uwu_version(void) { return 1; }

// generation counter of the cached environment switches (common.h UwuEnv)
static int g_env_gen = 0;
int uwu_env_generation() { return g_env_gen; }
extern "C" int uwu_env_refresh(void) { return ++g_env_gen; }

extern "C" int uwu_memset_zero(void* p, uint64_t bytes, void* stream) {
  if (!p || !bytes) {
    uwu_set_error("memset_zero: bad argument");
    return UWU_EINVAL;
  }
  if (hipMemsetAsync(p, 0, (size_t)bytes, (hipStream_t)stream) != hipSuccess) {
    uwu_set_error("memset_zero: hipMemsetAsync failed");
    return UWU_ELAUNCH;
  }
  return UWU_OK;
}

// ---- per-device facts for the kernels that own a whole CU (persistent grids, > 64 KB of dynamic LDS) -------------------------
// Cached PER DEVICE: a process may drive several GPUs (a function-local `static once` applied the first device's CU
// count and LDS attribute to every later device).
static const int kMaxDev = 64;
static int g_dev_cus[kMaxDev];
static size_t g_dev_lds[kMaxDev];
static bool g_dev_known[kMaxDev];
static int cur_dev() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) dev = 0;
  if (!g_dev_known[dev]) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) {
      g_dev_cus[dev] = prop.multiProcessorCount;
      g_dev_lds[dev] = prop.sharedMemPerBlockOptin ? prop.sharedMemPerBlockOptin : prop.sharedMemPerBlock;
    } else {
      g_dev_cus[dev] = 0;
      g_dev_lds[dev] = 0;
    }
    g_dev_known[dev] = true;
  }
  return dev;
}
int uwu_dev_index() { return cur_dev(); }
int uwu_dev_cus() { return g_dev_cus[cur_dev()]; }
bool uwu_dev_lds_fits(size_t bytes) { return g_dev_lds[cur_dev()] >= bytes; }
// hipFuncAttributeMaxDynamicSharedMemorySize for `fn` on the current device, once per device (`done`: kMaxDev flags owned by
// the caller, one array per kernel); false when the device cannot give a workgroup that much LDS or the call fails
bool uwu_func_lds(const void* fn, size_t bytes, unsigned char* done) {
  const int dev = cur_dev();
  if (done[dev] == 1) return true;
  if (done[dev] == 2) return false;
  const bool ok = g_dev_lds[dev] >= bytes && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
  done[dev] = ok ? 1 : 2;
  return ok;
}
