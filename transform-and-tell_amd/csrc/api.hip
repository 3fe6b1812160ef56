// Error reporting + misc entry points of libtell_hip.so.
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

extern "C" void tell_set_error(const char* msg) {
  strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
}
extern "C" const char* tell_last_error(void) { return g_err; }
extern "C" int tell_abi_version(void) { return 1; }

// number of visible HIP devices (0 on a CPU-only box); never throws
extern "C" int tell_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}

// host-side evaluation of the dropout hash (same function the kernels use) - lets CPU tests pin
// the numpy restatement in tell_amd/rng.py without a GPU
const uint32_t* g_tell_rng_step = nullptr;
extern "C" int tell_set_rng_step_ptr(const void* counter, hipStream_t) {
  g_tell_rng_step = static_cast<const uint32_t*>(counter);
  return TELL_OK;
}
// position offset of a captured decode step (adaptive.hip embed_finalize): separate from the dropout counter, so that
// a captured TRAINING step (fresh masks per replay, positions fixed) and a captured DECODE step (position advances per
// replay) can both exist
const uint32_t* g_tell_pos_step = nullptr;
extern "C" int tell_set_pos_step_ptr(const void* counter, hipStream_t) {
  g_tell_pos_step = static_cast<const uint32_t*>(counter);
  return TELL_OK;
}
extern "C" uint32_t tell_keep_field_host(uint32_t seed, uint32_t salt, uint64_t idx) { return tell_keep_field(seed, salt, idx); }
extern "C" uint32_t tell_drop_threshold_host(float p) { return tell_drop_threshold(p); }

// rate of the device wall clock (wall_clock64) that the GEMM kernels' execution-span stamps use (bench.py roofline)
extern "C" int tell_wall_clock_khz(void) {
  int dev = 0, khz = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return khz;
}
