// Calibration microbenchmarks for the fp32 MFMA inner loops (not part of the product).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I sim2real_lane_segment_amd/csrc tools/microbench.hip -o tools/microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "common.h"
using namespace rln;

// A: MFMA from registers, NACC independent accumulators
template <int NACC>
__global__ __launch_bounds__(256) void k_reg(float* out, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 9; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = mfma16(a, b, acc[i]);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// B: the dense-forward LDS -> MFMA block (same layout constants as IgCfg<3,1,...,8,32>), LDS filled once
constexpr int CHS = 368, PITCH = 34, NS = 9, TW = 32;
template <bool SYNC, int UNROLL_KG>
__global__ __launch_bounds__(256) void k_lds(float* out, int iters, int lds_floats) {
  extern __shared__ float smem[];
  for (int i = threadIdx.x; i < lds_floats; i += 256) smem[i] = (i % 97) * 1e-3f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lj = lane & 15, lk = lane >> 4;
  const float* zbase = smem + lk * CHS + ((wave * 4 * 16) / TW) * PITCH + lj;
  const float* wlane = smem + 16 * CHS + lane;
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll UNROLL_KG
    for (int kg = 0; kg < 4; ++kg) {
      const float* zk = zbase + kg * 4 * CHS;
      const float* wk = wlane + kg * NS * 64;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const float bw = wk[s * 64];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const float a = zk[((m * 16) / TW) * PITCH + (m * 16) % TW + (s / 3) * PITCH + (s % 3)];
          acc[m] = mfma16(a, bw, acc[m]);
        }
      }
    }
    if (SYNC) __syncthreads();
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F>
float time_ms(F f) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  f();
  hipDeviceSynchronize();
  hipEventRecord(a);
  f();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms;
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 8192 * sizeof(float));
  const int iters = 400;
  for (int bpc : {1, 2, 4}) {
    const int grid = 256 * bpc;
    auto rep = [&](const char* name, float ms, double mfma_per_wave) {
      double flops = (double)grid * 4 * mfma_per_wave * 2048.0;
      printf("%-34s blocks/CU %d : %.3f ms  %.1f TFLOP/s\n", name, bpc, ms, flops / ms / 1e9);
    };
    rep("reg 4 acc", time_ms([&] { hipLaunchKernelGGL(k_reg<4>, dim3(grid), dim3(256), 0, 0, out, iters); }), iters * 36.0);
    rep("reg 1 acc", time_ms([&] { hipLaunchKernelGGL(k_reg<1>, dim3(grid), dim3(256), 0, 0, out, iters); }), iters * 9.0);
    // LDS size chosen so that exactly bpc blocks fit per CU
    const int lds_bytes = (bpc == 1) ? 96 * 1024 : (bpc == 2) ? 66 * 1024 : 34 * 1024;
    const int lds_floats = 16 * CHS + 2304;
    hipFuncSetAttribute((const void*)k_lds<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    hipFuncSetAttribute((const void*)k_lds<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    hipFuncSetAttribute((const void*)k_lds<false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    hipFuncSetAttribute((const void*)k_lds<true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    rep("lds kg-rolled nosync", time_ms([&] { hipLaunchKernelGGL((k_lds<false, 1>), dim3(grid), dim3(256), lds_bytes, 0, out, iters, lds_floats); }), iters * 144.0);
    rep("lds kg-rolled sync/chunk", time_ms([&] { hipLaunchKernelGGL((k_lds<true, 1>), dim3(grid), dim3(256), lds_bytes, 0, out, iters, lds_floats); }), iters * 144.0);
    rep("lds kg-unrolled nosync", time_ms([&] { hipLaunchKernelGGL((k_lds<false, 4>), dim3(grid), dim3(256), lds_bytes, 0, out, iters, lds_floats); }), iters * 144.0);
    rep("lds kg-unrolled sync/chunk", time_ms([&] { hipLaunchKernelGGL((k_lds<true, 4>), dim3(grid), dim3(256), lds_bytes, 0, out, iters, lds_floats); }), iters * 144.0);
  }
  return 0;
}
