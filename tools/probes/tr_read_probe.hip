#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(const uint16_t* in, uint16_t* out, int rowstride_bytes) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = in[i];
  __syncthreads();
  const int l = threadIdx.x;
  // lane i of a 16-lane group supplies the address of row (i>>2), 8-byte piece (i&3)
  const int i = l & 15, g = l >> 4;
  const int off = (i >> 2) * rowstride_bytes + (i & 3) * 8 + g * 32;   // group g starts 16 columns further
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)((__attribute__((address_space(3))) char*)lds + off));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)v[j];
}
int main() {
  uint16_t h[4096], *d, *o, r[256];
  for (int i = 0; i < 4096; ++i) h[i] = i;   // value = row*64 + col for rowstride 128 B (64 elements)
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(r));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, o, 128);
  hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d  (row,col) = (%d,%d) (%d,%d) (%d,%d) (%d,%d)\n", l, r[l*4], r[l*4+1], r[l*4+2], r[l*4+3],
     r[l*4]/64, r[l*4]%64, r[l*4+1]/64, r[l*4+1]%64, r[l*4+2]/64, r[l*4+2]%64, r[l*4+3]/64, r[l*4+3]%64);
  return 0;
}
