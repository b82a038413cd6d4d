// Shared device helpers for the gfx950 (CDNA4) kernels of the lane-segmentation hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rln {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_16x16x4_f32: D[i][j] += sum_k A[i][k] * B[k][j], exact fp32 (fmaf chain).
// Lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]; it receives
// D[i = 4*(l>>4) + r][j = l&15] in register r.
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// sum over the four 16-lane groups (lanes l, l^16, l^32, l^48)
__device__ __forceinline__ float group4_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

constexpr int round_mod32(int x, int r) {  // smallest y >= x with y % 32 == r
  int y = (x / 32) * 32 + r;
  return y >= x ? y : y + 32;
}
constexpr int cmax(int a, int b) { return a > b ? a : b; }
constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }

}  // namespace rln
