// Shared device helpers for the gfx950 (CDNA4) kernels of the lane-segmentation hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <cstdlib>

namespace rln {

// Experiment switches (RLN_* environment variables) exist in diagnostic builds only (build.sh -DRLN_DIAG): the shipped
// library answers "unset" at compile time, so every switch folds away and no launch path reads the environment.
#ifdef RLN_DIAG
inline const char* rln_env(const char* name) { return std::getenv(name); }
#else
constexpr const char* rln_env(const char*) { return nullptr; }
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));

// One-time-per-DEVICE latch for per-kernel function attributes (hipFuncAttributeMaxDynamicSharedMemorySize is a
// per-device property: a process that drives a second device must set it there too).  first() is true once per device;
// undo() re-arms the current device after a failed attempt.
struct DevOnce {
  std::atomic<unsigned long long> mask{0};
  static unsigned long long bit() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return 0;
    return 1ull << d;
  }
  bool first() {
    const unsigned long long b = bit();
    if (b == 0) return true;  // unknown device index: set the attribute every time
    return !(mask.fetch_or(b) & b);
  }
  void undo() { mask.fetch_and(~bit()); }
};

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

// XCD-aware block order.  Workgroups are dealt round-robin over the 8 XCDs (linear ids b and b+8 share an L2), so
// neighbouring tiles of one sample would each pull their shared halo rows into a different L2.  This maps the
// dispatch-order id to a logical id such that every XCD walks ONE contiguous range of the logical order
// (bijective for any grid size).  Placement is a speed matter only; results never depend on it.
__device__ inline unsigned xcd_logical_block(unsigned lin, unsigned total) {
  const unsigned xcd = lin & 7u, slot = lin >> 3;
  const unsigned q = total >> 3, r = total & 7u;
  const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + slot;
}

}  // namespace rln
