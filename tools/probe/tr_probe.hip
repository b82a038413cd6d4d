// Probe of ds_read_b64_tr_b16 semantics (gfx950): which (row, column) element lands in which lane/element.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
  __shared__ short lds[64 * 16];
  int t = threadIdx.x;
  for (int i = t; i < 64 * 16; i += 64) lds[i] = (short)i;  // element (row r, col c) = r*16 + c, rows of 16 shorts (32 B)
  __syncthreads();
  int g = t >> 4, q = (t & 15) >> 2, p = t & 3;
  const short* addr = lds + (g * 4 + q) * 16 + 4 * p;  // group g reads rows 4g..4g+3
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)addr);
  for (int e = 0; e < 4; ++e) out[t * 4 + e] = v[e];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int t = 0; t < 64; ++t) {
    printf("lane %2d:", t);
    for (int e = 0; e < 4; ++e) printf(" (r%d,c%d)", h[t * 4 + e] / 16, h[t * 4 + e] % 16);
    printf("\n");
  }
  return 0;
}
