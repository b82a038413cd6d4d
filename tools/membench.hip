// Memory-side calibration: the dense-forward tile staging access pattern without LDS/MFMA (not part of the product).
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int H = 120, W = 160, CT = 288, N = 64, TH = 8, TW = 32;
#ifndef PAD
#define PAD 0
#endif
constexpr long long PLANE = (long long)H * W + PAD;

// A: dword loads, (TH+2)x(TW+2) tile positions, 16 channels per chunk (scalar staging pattern)
template <bool BAR>
__global__ __launch_bounds__(256) void k_dword(const float* __restrict__ S, float* out, int nchunk) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x;
  const int tile_y = blockIdx.x / 5, tile_x = blockIdx.x % 5, n = blockIdx.z;
  int goff[2];
  for (int i = 0; i < 2; ++i) {
    const int e = tid + 256 * i;
    int off = 0;
    if (e < 340) {
      const int r = e / 34, c = e % 34;
      const int iy = tile_y * TH - 1 + r, ix = tile_x * TW - 1 + c;
      if (iy >= 0 && iy < H && ix >= 0 && ix < W) off = iy * W + ix;
    }
    goff[i] = off;
  }
  const float* Sn = S + (long long)n * CT * PLANE;
  float acc = 0.f;
  for (int ch = 0; ch < nchunk; ++ch) {
    float v[32];
#pragma unroll
    for (int cc = 0; cc < 16; ++cc) {
      const float* src = Sn + (long long)(ch * 16 + cc) * PLANE;
#pragma unroll
      for (int i = 0; i < 2; ++i) v[cc * 2 + i] = src[goff[i]];
    }
#pragma unroll
    for (int k = 0; k < 32; ++k) acc += v[k];
    if (BAR) __syncthreads();
  }
  out[((long long)n * gridDim.x + blockIdx.x) * 256 + tid] = acc;
}

// B: float4 loads, aligned image of (TH+2) rows x (TW+8) columns
template <bool BAR>
__global__ __launch_bounds__(256) void k_quad(const float* __restrict__ S, float* out, int nchunk) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x;
  const int tile_y = blockIdx.x / 5, tile_x = blockIdx.x % 5, n = blockIdx.z;
  int goff[7];
  for (int i = 0; i < 7; ++i) {
    const int e = tid + 256 * i;
    int off = 0;
    if (e < 1600) {
      const int cc = e / 100, rem = e % 100, r = rem / 10, q = rem % 10;
      const int iy = tile_y * TH - 1 + r, ix = tile_x * TW - 4 + 4 * q;
      off = cc * (int)PLANE + ((iy >= 0 && iy < H && ix >= 0 && ix < W) ? iy * W + ix : 0);
    }
    goff[i] = off;
  }
  const float* Sn = S + (long long)n * CT * PLANE;
  float acc = 0.f;
  for (int ch = 0; ch < nchunk; ++ch) {
    const float* src = Sn + (long long)(ch * 16) * PLANE;
    float4 v[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) v[i] = *reinterpret_cast<const float4*>(src + goff[i]);
#pragma unroll
    for (int i = 0; i < 7; ++i) acc += v[i].x + v[i].y + v[i].z + v[i].w;
    if (BAR) __syncthreads();
  }
  out[((long long)n * gridDim.x + blockIdx.x) * 256 + tid] = acc;
}

// C: float4 loads, exact tile rows (no halo), fully aligned 128-byte rows
template <bool BAR>
__global__ __launch_bounds__(256) void k_quad_nohalo(const float* __restrict__ S, float* out, int nchunk) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x;
  const int tile_y = blockIdx.x / 5, tile_x = blockIdx.x % 5, n = blockIdx.z;
  int goff[4];
  for (int i = 0; i < 4; ++i) {
    const int e = tid + 256 * i;  // < 1024
    const int cc = e / 64, rem = e % 64, r = rem / 8, q = rem % 8;
    goff[i] = cc * (int)PLANE + (tile_y * TH + r) * W + tile_x * TW + 4 * q;
  }
  const float* Sn = S + (long long)n * CT * PLANE;
  float acc = 0.f;
  for (int ch = 0; ch < nchunk; ++ch) {
    const float* src = Sn + (long long)(ch * 16) * PLANE;
    float4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const float4*>(src + goff[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc += v[i].x + v[i].y + v[i].z + v[i].w;
    if (BAR) __syncthreads();
  }
  out[((long long)n * gridDim.x + blockIdx.x) * 256 + tid] = acc;
}

// D: plain linear float4 streaming copy-read of the same number of bytes (reference ceiling)
__global__ __launch_bounds__(256) void k_stream(const float4* __restrict__ S, float* out, long long n4) {
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 v = S[i];
    acc += v.x + v.y + v.z + v.w;
  }
  out[(long long)blockIdx.x * 256 + threadIdx.x] = acc;
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
  const long long total = (long long)N * CT * PLANE;
  float *S, *out;
  hipMalloc(&S, total * sizeof(float));
  hipMalloc(&out, 75LL * N * 256 * 64 * sizeof(float));
  hipMemset(S, 0, total * sizeof(float));
  const int nchunk = 17;  // 272 input channels
  dim3 grid(75, 1, N);
  for (int lds : {33 * 1024, 66 * 1024, 16 * 1024}) {
    const int bpc = 160 * 1024 / lds > 8 ? 8 : 160 * 1024 / lds;
    hipFuncSetAttribute((const void*)k_dword<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    hipFuncSetAttribute((const void*)k_dword<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    hipFuncSetAttribute((const void*)k_quad<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    hipFuncSetAttribute((const void*)k_quad<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    hipFuncSetAttribute((const void*)k_quad_nohalo<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    auto rep = [&](const char* name, float ms, double bytes) {
      printf("%-28s ~%d blocks/CU: %.3f ms  %.0f GB/s useful\n", name, bpc, ms, bytes / ms / 1e6);
    };
    const double halo_bytes = 75.0 * N * nchunk * 16 * 340 * 4, quad_bytes = 75.0 * N * nchunk * 16 * 400 * 4,
                 tile_bytes = 75.0 * N * nchunk * 16 * 256 * 4;
    rep("dword halo", time_ms([&] { hipLaunchKernelGGL(k_dword<false>, grid, dim3(256), lds, 0, S, out, nchunk); }), halo_bytes);
    rep("dword halo + barrier", time_ms([&] { hipLaunchKernelGGL(k_dword<true>, grid, dim3(256), lds, 0, S, out, nchunk); }), halo_bytes);
    rep("float4 aligned(+8) halo", time_ms([&] { hipLaunchKernelGGL(k_quad<false>, grid, dim3(256), lds, 0, S, out, nchunk); }), quad_bytes);
    rep("float4 aligned halo + barrier", time_ms([&] { hipLaunchKernelGGL(k_quad<true>, grid, dim3(256), lds, 0, S, out, nchunk); }), quad_bytes);
    rep("float4 no halo", time_ms([&] { hipLaunchKernelGGL(k_quad_nohalo<false>, grid, dim3(256), lds, 0, S, out, nchunk); }), tile_bytes);
  }
  const long long n4 = (long long)N * 272 * PLANE / 4;
  float ms = time_ms([&] { hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, 0, (const float4*)S, out, n4); });
  printf("linear float4 stream: %.3f ms %.0f GB/s\n", ms, n4 * 16.0 / ms / 1e6);
  return 0;
}
