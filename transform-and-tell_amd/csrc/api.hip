// Error reporting + misc entry points of libtell_hip.so.
#include "common.h"
#include "options.h"
#include <limits.h>
#include <string.h>

static thread_local char g_err[512] = "";

extern "C" void tell_set_error(const char* msg) {
  strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
}
extern "C" const char* tell_last_error(void) { return g_err; }
extern "C" int tell_abi_version(void) { return 2; }

// ---- run-time options (options.h): the library's only tunable state, set explicitly - no environment variable is read
#define TELL_X(e, k, d) {d},
std::atomic<long> g_tell_opt[TELL_OPT_COUNT] = {
  TELL_OPTION_LIST(TELL_X)
#ifdef TELL_PROBES
  TELL_PROBE_LIST(TELL_X)
#endif
};
#undef TELL_X
#define TELL_X(e, k, d) k,
static const char* const g_opt_keys[TELL_OPT_COUNT] = {
  TELL_OPTION_LIST(TELL_X)
#ifdef TELL_PROBES
  TELL_PROBE_LIST(TELL_X)
#endif
};
#undef TELL_X
#define TELL_X(e, k, d) d,
static const long g_opt_defaults[TELL_OPT_COUNT] = {
  TELL_OPTION_LIST(TELL_X)
#ifdef TELL_PROBES
  TELL_PROBE_LIST(TELL_X)
#endif
};
#undef TELL_X
static int opt_index(const char* key) {
  if (!key) return -1;
  for (int i = 0; i < TELL_OPT_COUNT; ++i)
    if (strcmp(key, g_opt_keys[i]) == 0) return i;
  return -1;
}
extern "C" int tell_set_option(const char* key, long value) {
  const int i = opt_index(key);
  if (i < 0) {
    char msg[160];
    snprintf(msg, sizeof(msg), "set_option: unknown option '%.64s'", key ? key : "(null)");
    tell_set_error(msg);
    return TELL_ERR_ARG;
  }
  g_tell_opt[i].store(value, std::memory_order_relaxed);
  return TELL_OK;
}
extern "C" long tell_get_option(const char* key) {
  const int i = opt_index(key);
  return i < 0 ? LONG_MIN : g_tell_opt[i].load(std::memory_order_relaxed);
}
extern "C" const char* tell_option_key(int index) { return index >= 0 && index < TELL_OPT_COUNT ? g_opt_keys[index] : nullptr; }
extern "C" long tell_option_default(int index) { return index >= 0 && index < TELL_OPT_COUNT ? g_opt_defaults[index] : LONG_MIN; }
extern "C" int tell_probe_build(void) {
#ifdef TELL_PROBES
  return 1;
#else
  return 0;
#endif
}

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
// Round 6: the per-token bookkeeping launch INSIDE the captured decode step.  A second registered word holds the offset of
// the NEXT step: the step's first kernel (tell_embed_gather_step, every block) reads it and its block 0 copies it into the
// counter above, which every later kernel of the step reads; the bookkeeping kernel - the step's last - reads the counter
// (every block) and its block 0 writes the next offset into the second word.  Nobody writes a word while another block of
// the same launch may still read it, and the host's part of a decode step shrinks to one graph replay.
uint32_t* g_tell_pos_next = nullptr;
extern "C" int tell_set_pos_next_ptr(void* next, hipStream_t) {
  g_tell_pos_next = static_cast<uint32_t*>(next);
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

// Contention rehearsal (bench.py --cu-hog N): n workgroups that do nothing but HOLD a compute unit each for `ticks` of the
// device wall clock (64 KB of LDS: no 256x256 GEMM workgroup - 133 KB - fits beside one; the dispatcher places one per CU
// while free CUs exist).  What RCCL's channel kernels will do to a step whose dominant kernel needs whole CUs, measurable on
// one GPU.  Bounded: a launch never spins longer than 4 s.
__global__ __launch_bounds__(64) void cu_hog_kernel(unsigned long long ticks, int* stop) {
  __shared__ int hog[16384];
  hog[threadIdx.x] = (int)threadIdx.x;
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {
    if (stop && __hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;     // (L2: another stream sets it)
    __builtin_amdgcn_s_sleep(64);
  }
  if (stop && hog[(threadIdx.x * 7) & 16383] == -12345) *stop = 2;
}
extern "C" int tell_cu_hog(int n_workgroups, long ticks, int* stop, hipStream_t stream) {
  TELL_REQUIRE(n_workgroups >= 0 && n_workgroups <= 256 && ticks >= 0, "cu_hog: 0..256 workgroups");
  if (n_workgroups == 0) return TELL_OK;
  const int khz = tell_wall_clock_khz();
  const long cap = 4000L * (khz > 0 ? khz : 100000);
  hipLaunchKernelGGL(cu_hog_kernel, dim3(n_workgroups), dim3(64), 0, stream, (unsigned long long)(ticks < cap ? ticks : cap), stop);
  return tell_check_launch("cu_hog");
}
