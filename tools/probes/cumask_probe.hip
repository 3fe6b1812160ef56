// Probe: which CUs does a hipExtStreamCreateWithCUMask stream use on MI355X (bit -> XCD / CU), and does a hipGraph that
// was captured on an ordinary stream honour the mask of the stream it is LAUNCHED on?
// hipcc --offload-arch=gfx950 -O2 tools/probes/cumask_probe.hip -o gpurun_out/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <map>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void where_kernel(uint32_t* out, int spin) {
  uint32_t xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  // keep the workgroup alive a little so that the whole grid spreads over the CUs it may use
  long t0 = clock64();
  while (clock64() - t0 < spin) {}
  if (threadIdx.x == 0) { out[blockIdx.x * 2] = xcc & 0xf; out[blockIdx.x * 2 + 1] = hw; }
}
__global__ void stream_kernel(const uint4* __restrict__ a, uint4* __restrict__ b, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x, st = (long)gridDim.x * blockDim.x;
  uint4 acc = {0, 0, 0, 0};
  for (; i < n; i += st) { uint4 v = a[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
  if (acc.x == 0x12345 && acc.y == 7) b[0] = acc;
}

static void report(const char* what, const std::vector<uint32_t>& h, int n) {
  std::map<int, std::set<int>> per;
  for (int i = 0; i < n; ++i) {
    int xcc = h[2 * i], hw = h[2 * i + 1];
    int cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
    per[xcc].insert(se * 32 + sh * 16 + cu);
  }
  int tot = 0;
  printf("%s:", what);
  for (auto& kv : per) { printf(" xcc%d=%zu", kv.first, kv.second.size()); tot += (int)kv.second.size(); }
  printf("  total CUs %d\n", tot);
}

int main() {
  int n = 4096;
  uint32_t* d; CK(hipMalloc(&d, n * 8));
  std::vector<uint32_t> h(2 * n);
  hipStream_t plain; CK(hipStreamCreate(&plain));
  auto run = [&](hipStream_t s, const char* what) -> int {
    CK(hipMemsetAsync(d, 0xff, n * 8, s));
    hipLaunchKernelGGL(where_kernel, dim3(n), dim3(256), 0, s, d, 20000);
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost));
    report(what, h, n);
    return 0;
  };
  run(plain, "plain stream");
  struct M { const char* name; uint32_t w[8]; } masks[] = {
    {"bits 0-31", {0xffffffffu, 0, 0, 0, 0, 0, 0, 0}},
    {"bits 0-7", {0xffu, 0, 0, 0, 0, 0, 0, 0}},
    {"bits 0,8,16,24", {0x01010101u, 0, 0, 0, 0, 0, 0, 0}},
    {"bits 0-223", {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0}},
    {"bits 224-255", {0, 0, 0, 0, 0, 0, 0, 0xffffffffu}},
    {"bits 192-255", {0, 0, 0, 0, 0, 0, 0xffffffffu, 0xffffffffu}},
  };
  // a graph captured on the plain stream
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(plain, hipStreamCaptureModeGlobal));
  hipLaunchKernelGGL(where_kernel, dim3(n), dim3(256), 0, plain, d, 20000);
  CK(hipStreamEndCapture(plain, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  long nbytes = 1L << 30;
  uint4 *a, *b; CK(hipMalloc(&a, nbytes)); CK(hipMalloc(&b, 64)); CK(hipMemset(a, 1, nbytes));
  for (auto& m : masks) {
    hipStream_t s;
    CK(hipExtStreamCreateWithCUMask(&s, 8, m.w));
    char buf[128];
    snprintf(buf, sizeof buf, "mask %-16s direct launch", m.name); run(s, buf);
    CK(hipMemsetAsync(d, 0xff, n * 8, s));
    CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost));
    snprintf(buf, sizeof buf, "mask %-16s graph replay ", m.name); report(buf, h, n);
    // streaming bandwidth through the masked stream
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0, s));
      hipLaunchKernelGGL(stream_kernel, dim3(2048), dim3(256), 0, s, a, b, nbytes / 16);
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("mask %-16s read 1 GiB: %.1f us = %.2f TB/s\n", m.name, ms * 1e3, nbytes / (ms * 1e-3) / 1e12);
    CK(hipStreamDestroy(s));
  }
  return 0;
}
