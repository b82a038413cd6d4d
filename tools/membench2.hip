// Memory-side calibration 2: useful (tile) bytes per second of halo'd tile patterns, with and without the XCD-aware
// block order.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int H = 120, W = 160, CT = 288, N = 64;
constexpr long long PLANE = (long long)H * W;

__device__ inline unsigned xcd_logical_block(unsigned lin, unsigned total) {
  const unsigned xcd = lin & 7u, slot = lin >> 3;
  const unsigned q = total >> 3, r = total & 7u;
  const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + slot;
}

// generic: tile TH x TW (TW multiple of 4, tile x origin multiple of 4), halo HALO rows each side (0/1), quad columns
// [x0-4*HC, x0+TW+4*HC), CK channels per chunk; each thread loads NQ quads per chunk.
template <int TH, int TW, int HALO, int CK, bool XCD>
__global__ __launch_bounds__(256) void k_tile(const float* __restrict__ S, float* out, int nchunk) {
  constexpr int ROWS = TH + 2 * HALO, QW = TW / 4 + 2 * HALO, QPC = ROWS * QW, NQ = (CK * QPC + 255) / 256;
  constexpr int TX = W / TW, TY = H / TH;
  unsigned b = blockIdx.x;
  if (XCD) b = xcd_logical_block(b, gridDim.x);
  const int n = b / (TX * TY), t = b % (TX * TY);
  const int tile_y = t / TX, tile_x = t % TX;
  const int tid = threadIdx.x;
  int goff[NQ];
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int e = tid + 256 * i;
    int off = 0;
    if (e < CK * QPC) {
      const int cc = e / QPC, rem = e % QPC, r = rem / QW, q = rem % QW;
      const int iy = tile_y * TH - HALO + r, ix = tile_x * TW - 4 * HALO + 4 * q;
      off = cc * (int)PLANE + ((iy >= 0 && iy < H && ix >= 0 && ix < W) ? iy * W + ix : 0);
    }
    goff[i] = off;
  }
  const float* Sn = S + (long long)n * CT * PLANE;
  float acc = 0.f;
  for (int ch = 0; ch < nchunk; ++ch) {
    const float* src = Sn + (long long)(ch * CK) * PLANE;
    float4 v[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) v[i] = *reinterpret_cast<const float4*>(src + goff[i]);
#pragma unroll
    for (int i = 0; i < NQ; ++i) acc += v[i].x + v[i].y + v[i].z + v[i].w;
    __syncthreads();
  }
  out[(long long)blockIdx.x * 256 + tid] = acc;
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

template <int TH, int TW, int HALO, int CK, bool XCD>
void run(const float* S, float* out, int lds, const char* name) {
  auto k = k_tile<TH, TW, HALO, CK, XCD>;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  const int nchunk = 272 / CK;
  const int blocks = (H / TH) * (W / TW) * N;
  const float ms = time_ms([&] { hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, 0, S, out, nchunk); });
  const double useful = (double)N * 272 * PLANE * 4;
  printf("%-34s lds %3d KB: %.3f ms  %5.0f GB/s useful\n", name, lds / 1024, ms, useful / ms / 1e6);
}

int main() {
  const long long total = (long long)N * CT * PLANE;
  float *S, *out;
  hipMalloc(&S, total * sizeof(float));
  hipMalloc(&out, 4800LL * N * 256 * sizeof(float));
  hipMemset(S, 0, total * sizeof(float));
  for (int rep = 0; rep < 2; ++rep) {
    run<8, 32, 1, 16, false>(S, out, 40 << 10, "8x32 halo ck16");
    run<8, 32, 1, 16, true>(S, out, 40 << 10, "8x32 halo ck16 xcd");
    run<4, 80, 1, 16, false>(S, out, 52 << 10, "4x80 halo ck16");
    run<4, 80, 1, 16, true>(S, out, 52 << 10, "4x80 halo ck16 xcd");
    run<4, 160, 1, 16, false>(S, out, 76 << 10, "4x160 halo ck16");
    run<4, 160, 1, 16, true>(S, out, 76 << 10, "4x160 halo ck16 xcd");
    run<8, 160, 1, 8, false>(S, out, 76 << 10, "8x160 halo ck8");
    run<8, 160, 1, 8, true>(S, out, 76 << 10, "8x160 halo ck8 xcd");
    run<8, 80, 1, 8, false>(S, out, 52 << 10, "8x80 halo ck8");
    run<8, 80, 1, 8, true>(S, out, 52 << 10, "8x80 halo ck8 xcd");
    run<2, 160, 0, 16, false>(S, out, 52 << 10, "2x160 nohalo ck16 (rolling)");
    run<2, 160, 0, 16, true>(S, out, 52 << 10, "2x160 nohalo ck16 (rolling) xcd");
    run<4, 160, 0, 16, false>(S, out, 76 << 10, "4x160 nohalo ck16");
    run<8, 32, 0, 16, false>(S, out, 40 << 10, "8x32 nohalo ck16");
  }
  return 0;
}
