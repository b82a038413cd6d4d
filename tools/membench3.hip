// Memory-side calibration 3: the data-gradient epilogue traffic (read S, read G, write G; 16 channels x 8x32 pixels per
// step) with two lane mappings: A = row-contiguous (64 lanes cover 1 KiB of one channel), B = the MFMA accumulator
// layout used today (lane = (channel, 4-pixel group): 16 channels x 64 B per wave-instruction).  Not product code.
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int H = 120, W = 160, CT = 288, N = 64;
constexpr long long PLANE = (long long)H * W;

template <int MAP, bool RMW>
__global__ __launch_bounds__(256) void k(const float* __restrict__ S, float* __restrict__ G, float* out, int nstep) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile_y = blockIdx.x / 5, tile_x = blockIdx.x % 5, n = blockIdx.z;
  int off[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    int cc, ty, tx;
    if (MAP == 0) {  // A: e = quad index in (cc, row, quad-in-row) order
      const int e = tid + 256 * m;
      cc = e >> 6; ty = (e & 63) >> 3; tx = (e & 7) * 4;
    } else {         // B: accumulator layout
      cc = lane & 15;
      const int q = (wave * 4 + m) * 16 + (lane >> 4) * 4;
      ty = q >> 5; tx = q & 31;
    }
    off[m] = cc * (int)PLANE + (tile_y * 8 + ty) * W + tile_x * 32 + tx;
  }
  const long long nb = (long long)n * CT * PLANE;
  float acc = 0.f;
  for (int st = 0; st < nstep; ++st) {
    const float* Sc = S + nb + (long long)(st * 16) * PLANE;
    float* Gc = G + nb + (long long)(st * 16) * PLANE;
    float4 s[4], g[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) s[m] = *reinterpret_cast<const float4*>(Sc + off[m]);
#pragma unroll
    for (int m = 0; m < 4; ++m) g[m] = *reinterpret_cast<const float4*>(Gc + off[m]);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      g[m].x += s[m].x; g[m].y += s[m].y; g[m].z += s[m].z; g[m].w += s[m].w;
      acc += g[m].x;
    }
    if (RMW) {
#pragma unroll
      for (int m = 0; m < 4; ++m) *reinterpret_cast<float4*>(Gc + off[m]) = g[m];
    }
    __syncthreads();
  }
  out[((long long)n * gridDim.x + blockIdx.x) * 256 + tid] = acc;
}

template <typename F>
float time_ms(F f) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  f();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  f();
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms;
  (void)hipEventElapsedTime(&ms, a, b);
  return ms;
}

int main() {
  const long long total = (long long)N * CT * PLANE;
  float *S, *G, *out;
  (void)hipMalloc(&S, total * sizeof(float));
  (void)hipMalloc(&G, total * sizeof(float));
  (void)hipMalloc(&out, 75LL * N * 256 * sizeof(float));
  (void)hipMemset(S, 0, total * sizeof(float));
  (void)hipMemset(G, 0, total * sizeof(float));
  const int nstep = 17;
  dim3 grid(75, 1, N);
  const double bytes_r = 2.0 * N * 272 * PLANE * 4, bytes_w = 1.0 * N * 272 * PLANE * 4;
  for (int rep = 0; rep < 2; ++rep)
    for (int lds : {50 << 10, 76 << 10}) {
      auto run = [&](auto kern, const char* name, double bytes) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 100 << 10);
        const float ms = time_ms([&] { hipLaunchKernelGGL(kern, grid, dim3(256), lds, 0, S, G, out, nstep); });
        printf("%-34s lds %2d KB: %.3f ms  %5.0f GB/s\n", name, lds >> 10, ms, bytes / ms / 1e6);
      };
      run(k<0, false>, "A row-contiguous, read S+G", bytes_r);
      run(k<1, false>, "B accumulator layout, read S+G", bytes_r);
      run(k<0, true>, "A row-contiguous, read S+G, write G", bytes_r + bytes_w);
      run(k<1, true>, "B accumulator layout, r S+G, w G", bytes_r + bytes_w);
    }
  return 0;
}
