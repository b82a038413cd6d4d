// Split-operand 16-bit MFMA helpers shared by the dense 3x3 (dense3.hip) and 1x1 transition (pw1.hip) kernel families.
#pragma once
#include "common.h"
#include "dense3.h"

namespace rln {

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---- operand splitting: x = part0 + part1 (+ part2), every part the 16-bit rounding of the remainder -------------
template <int DT>
__device__ __forceinline__ unsigned pack2(f32x2 v) {
  if constexpr (DT == D3_BF16) {
    union { bf16x2 h; unsigned u; } c;
    c.h = __builtin_convertvector(v, bf16x2);
    return c.u;
  } else {
    union { f16x2 h; unsigned u; } c;
    c.h = __builtin_convertvector(v, f16x2);
    return c.u;
  }
}
template <int DT>
__device__ __forceinline__ f32x2 unpack2(unsigned u) {
  if constexpr (DT == D3_BF16) {
    union { bf16x2 h; unsigned u; } c;
    c.u = u;
    return __builtin_convertvector(c.h, f32x2);
  } else {
    union { f16x2 h; unsigned u; } c;
    c.u = u;
    return __builtin_convertvector(c.h, f32x2);
  }
}
template <int DT, int NP>
__device__ __forceinline__ void split2(float x0, float x1, unsigned (&out)[NP]) {
  f32x2 r = {x0, x1};
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    out[p] = pack2<DT>(r);
    if (p + 1 < NP) r = r - unpack2<DT>(out[p]);
  }
}

// ---- f16 range handling (forward kernels; bf16 parts have fp32's exponent range and need none of this) -------------
// f16 parts cover |x| <= 65504 and lose relative precision below 2^-3 (the low part of a two-part split becomes
// subnormal: absolute error <= 2^-25 instead of relative 2^-22).  Three rules keep the f16x2 forward in the accuracy
// class of the fp32 MFMA chain and free of inf / NaN:
//   * activations behind BatchNorm + ReLU are clamped to [0, 65504] by the v_med3_f32 that replaces the ReLU's v_max
//     (same instruction count); raw inputs (first convolution, TransitionUp) are clamped to +-65504 before the split;
//     values beyond that range saturate instead of turning into inf - inf = NaN;
//   * forward-orientation weights are packed times 2^8 (Kaiming-scale weights ~1e-2 would put their low part into the
//     subnormal range: 2^-17 instead of 2^-22 relative) and saturate at |w| = 255.9; the kernels' epilogues multiply
//     the fp32 accumulator by 2^-8, which is exact;
//   * small activations keep an ABSOLUTE error <= 2^-25 per element, far below the fp32 rounding of the O(1) sums
//     they enter.
constexpr float kF16Max = 65504.f;
template <int DT>
__device__ __forceinline__ constexpr float w_prescale() { return DT == D3_F16 ? 256.f : 1.f; }
template <int DT>
__device__ __forceinline__ constexpr float w_unscale() { return DT == D3_F16 ? (1.f / 256.f) : 1.f; }
template <int DT>
__device__ __forceinline__ float relu16(float v) {  // ReLU, saturating at the part type's largest finite value
  if constexpr (DT == D3_F16) return __builtin_amdgcn_fmed3f(v, 0.f, kF16Max);
  else return fmaxf(v, 0.f);
}
template <int DT>
__device__ __forceinline__ float sat16(float v) {
  if constexpr (DT == D3_F16) return __builtin_amdgcn_fmed3f(v, -kF16Max, kF16Max);
  else return v;
}

template <int DT>
__device__ __forceinline__ f32x4 mfma32(const uint4& a, const uint4& b, f32x4 c) {
  if constexpr (DT == D3_BF16) {
    union { uint4 u; bf16x8 v; } ca, cb;
    ca.u = a;
    cb.u = b;
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(ca.v, cb.v, c, 0, 0, 0);
  } else {
    union { uint4 u; f16x8 v; } ca, cb;
    ca.u = a;
    cb.u = b;
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(ca.v, cb.v, c, 0, 0, 0);
  }
}
// acc += sum of the leading cross terms of (a0+a1+a2)*(b0+b1+b2), smallest terms first
template <int DT, int NP>
__device__ __forceinline__ f32x4 mfma_split(const uint4 (&a)[NP], const uint4 (&b)[NP], f32x4 c) {
  if constexpr (NP == 3) {
    c = mfma32<DT>(a[2], b[0], c);
    c = mfma32<DT>(a[1], b[1], c);
    c = mfma32<DT>(a[0], b[2], c);
  }
  if constexpr (NP >= 2) {
    c = mfma32<DT>(a[1], b[0], c);
    c = mfma32<DT>(a[0], b[1], c);
  }
  return mfma32<DT>(a[0], b[0], c);
}


// sum over the 16 lanes of a DPP row (lanes with equal l>>4): four rotate-adds, every lane receives the total
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, true));
  return v;
}

// LDS float add without return value (ds_add_f32): nothing to wait for.  Used for per-wave statistics slots that one wave
// owns, so the issue order is the summation order (deterministic).
// Wait states between the last MFMA of a dependent chain and the first ordinary read of its accumulator.  With the three-
// part first convolution -- whose epilogue reads the accumulators at once -- the compiler (ROCm 7.2, gfx950) placed
// `s_nop 6` + one instruction there, and rows 13 / 15 of the 16x16 tile (the rows a 16x16x32 MFMA writes in its last
// pass) came back different from run to run (tools/forward_bisect.py).  Sixteen explicit wait states after the chain.
__device__ __forceinline__ void mfma_drain() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 15");
  __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ void lds_add_f32(float* p, float v) {
  (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

}  // namespace rln
