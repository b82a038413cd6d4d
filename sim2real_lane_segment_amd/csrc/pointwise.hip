// HBM-bound kernels (see pointwise.h).  Everything here is wave64 code: per-lane streaming loads
// that are coalesced along the pixel axis of NCHW planes, wave64 shuffles for reductions, block
// partials + fixed-order second stage instead of float atomics (bitwise reproducible results).
#include "pointwise.h"
#include "storage.h"

#include <cstring>

namespace rln {

#define RLN_LAUNCH_CHECK() return (int)hipGetLastError()

// =============================================================================================
// BatchNorm bookkeeping
// =============================================================================================

// Column sums of the [nblk][J][2] partial rows for channel j by one 256-thread block, in double and in a fixed order
// (thread t: rows t, t + 256, ...; butterfly inside the wave; waves 0..3).  These reductions sit between two dependent
// launches ~120 times per step and are pure load latency: a block per channel needs nblk / 256 rounds of loads where a
// wave per channel needed nblk / 64 (4800 rows at 120x160: 9 -> 5 us per launch).
__device__ __forceinline__ void block_rows_sum2(const float* __restrict__ partial, long long nblk, int J, int j, double& s1,
                                                double& s2) {
  __shared__ double red2[8];
  double a = 0.0, b = 0.0;
#pragma unroll 4
  for (long long r = threadIdx.x; r < nblk; r += 256) {
    const float2 v = *reinterpret_cast<const float2*>(partial + (r * J + j) * 2);
    a += (double)v.x;
    b += (double)v.y;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    a += __shfl_xor(a, o, 64);
    b += __shfl_xor(b, o, 64);
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    red2[w * 2 + 0] = a;
    red2[w * 2 + 1] = b;
  }
  __syncthreads();
  s1 = ((red2[0] + red2[2]) + red2[4]) + red2[6];
  s2 = ((red2[1] + red2[3]) + red2[5]) + red2[7];
}

// one block per channel
__global__ __launch_bounds__(256) void bn_finalize_k(const float* __restrict__ partial, long long nblk, int J,
                                                     double count, float eps, float* mean, float* var,
                                                     float* invstd, float* stdv) {
  const int j = blockIdx.x;
  double s1, s2;
  block_rows_sum2(partial, nblk, J, j, s1, s2);
  if (threadIdx.x == 0) {
    const double m = s1 / count;
    double v = s2 / count - m * m;
    if (v < 0.0) v = 0.0;
    mean[j] = (float)m;
    var[j] = (float)v;
    const float sd = sqrtf((float)v + eps);
    invstd[j] = 1.0f / sd;
    stdv[j] = sd;
  }
}

// bn_finalize of the channels a layer has just written + bn_prep of the NEXT layer's BatchNorm over its whole input
// range, in one launch (they are adjacent in the stream and each costs a dependent-dispatch gap).  Blocks [0, Jnew):
// one block per new channel new_lo + blockIdx.x (statistics from the partial rows, then its table entry); the blocks
// after them: one thread per channel that already has statistics.
__global__ __launch_bounds__(256) void bn_finalize_prep_k(const float* __restrict__ partial, long long nblk, int Jnew,
                                                          int new_lo, double count, float eps, float* mean, float* var,
                                                          float* invstd, float* stdv, int C,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* running_mean, float* running_var, float momentum,
                                                          float unbias, float* a, float* b) {
  float m, vr, is;
  int c;
  if ((int)blockIdx.x < Jnew) {
    const int j = blockIdx.x;
    c = new_lo + j;
    double s1, s2;
    block_rows_sum2(partial, nblk, Jnew, j, s1, s2);
    if (threadIdx.x != 0) return;
    const double md = s1 / count;
    double v = s2 / count - md * md;
    if (v < 0.0) v = 0.0;
    m = (float)md;
    vr = (float)v;
    const float sd = sqrtf(vr + eps);
    is = 1.0f / sd;
    mean[c] = m;
    var[c] = vr;
    invstd[c] = is;
    stdv[c] = sd;
  } else {
    const int i = ((int)blockIdx.x - Jnew) * 256 + threadIdx.x;  // index among the C - Jnew older channels
    if (i >= C - Jnew) return;
    c = i < new_lo ? i : i + Jnew;
    m = mean[c];
    vr = var[c];
    is = invstd[c];
  }
  const float av = gamma[c] * is;
  a[c] = av;
  b[c] = beta[c] - m * av;
  if (running_mean != nullptr) {
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (vr * unbias);
  }
}

int bn_finalize_prep(const float* partial, long long nblk, int Jnew, int new_lo, double count, float eps, float* mean,
                     float* var, float* invstd, float* stdv, int C, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, float momentum, float* a, float* b, hipStream_t s) {
  const float unbias = count > 1.0 ? (float)(count / (count - 1.0)) : 1.0f;
  hipLaunchKernelGGL(bn_finalize_prep_k, dim3((unsigned)(Jnew + (C - Jnew + 255) / 256)), dim3(256), 0, s, partial, nblk, Jnew, new_lo, count, eps,
                     mean, var, invstd, stdv, C, gamma, beta, running_mean, running_var, momentum, unbias, a, b);
  RLN_LAUNCH_CHECK();
}

int bn_finalize(const float* partial, long long nblk, int J, double count, float eps, float* mean, float* var,
                float* invstd, float* stdv, hipStream_t s) {
  hipLaunchKernelGGL(bn_finalize_k, dim3((unsigned)J), dim3(256), 0, s, partial, nblk, J, count, eps, mean, var,
                     invstd, stdv);
  RLN_LAUNCH_CHECK();
}

__global__ __launch_bounds__(256) void bn_prep_k(int training, int C, const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, const float* __restrict__ mean,
                                                 const float* __restrict__ var, const float* __restrict__ invstd,
                                                 float* running_mean, float* running_var, float momentum,
                                                 float unbias, float eps, float* a, float* b) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  if (training) {
    const float m = mean[c], is = invstd[c];
    const float av = gamma[c] * is;
    a[c] = av;
    b[c] = beta[c] - m * av;
    if (running_mean != nullptr) {
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (var[c] * unbias);
    }
  } else {
    const float is = 1.0f / sqrtf(running_var[c] + eps);
    const float av = gamma[c] * is;
    a[c] = av;
    b[c] = beta[c] - running_mean[c] * av;
  }
}

int bn_prep(int training, int C, const float* gamma, const float* beta, const float* mean, const float* var,
            const float* invstd, float* running_mean, float* running_var, float momentum, double count, float eps,
            float* a, float* b, hipStream_t s) {
  const float unbias = count > 1.0 ? (float)(count / (count - 1.0)) : 1.0f;
  hipLaunchKernelGGL(bn_prep_k, dim3((C + 255) / 256), dim3(256), 0, s, training, C, gamma, beta, mean, var, invstd,
                     running_mean, running_var, momentum, unbias, eps, a, b);
  RLN_LAUNCH_CHECK();
}

// one block per channel j (block_rows_sum2)
__device__ __forceinline__ void bn_bwd_finalize_body(int j, const float* __restrict__ partial, long long nblk, int J,
                                                     const float* __restrict__ gamma, float* dgamma, float* dbeta,
                                                     float* S1, float* S2) {
  double s1, s2;
  block_rows_sum2(partial, nblk, J, j, s1, s2);
  if (threadIdx.x == 0) {
    dbeta[j] = (float)s1;
    dgamma[j] = (float)s2;
    const float g = gamma[j];
    S1[j] += g * (float)s1;
    S2[j] += g * (float)s2;
  }
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_k(const float* __restrict__ partial, long long nblk, int J,
                                                         const float* __restrict__ gamma, float* dgamma, float* dbeta,
                                                         float* S1, float* S2) {
  bn_bwd_finalize_body((int)blockIdx.x, partial, nblk, J, gamma, dgamma, dbeta, S1, S2);
}

int bn_bwd_finalize(const float* partial, long long nblk, int J, const float* gamma, float* dgamma, float* dbeta,
                    float* S1, float* S2, hipStream_t s) {
  hipLaunchKernelGGL(bn_bwd_finalize_k, dim3((unsigned)J), dim3(256), 0, s, partial, nblk, J, gamma, dgamma, dbeta, S1,
                     S2);
  RLN_LAUNCH_CHECK();
}

// =============================================================================================
// gradient finalisation:  d/dx = invstd * (G - S1/M - xhat * S2/M)  [* dropout scale]
// =============================================================================================

template <int ST, int YT>  // storage element types of S (the level's activations) and of dst (its finalised gradients)
__global__ __launch_bounds__(256) void grad_finalize_k(const GradFinParams p) {
  const int c = blockIdx.y, n = blockIdx.z;
  const int tid = threadIdx.x;
  const float mean = p.mean[c], is = p.invstd[c];
  const float k1 = p.S1[c] * p.invM, k2 = p.S2[c] * p.invM;
  float sc = 1.f;
  if (p.nscale != nullptr) sc = p.nscale[(long long)n * p.C + c];
  const long long plane = (long long)p.H * p.W;
  const SP<ST> Sp = SP<ST>(p.S) + ((long long)n * p.ns + (long long)c * plane);
  const float* Gp = p.G + (long long)n * p.ns + (long long)c * plane;
  const long long dplane = (long long)p.Hd * p.Wd;
  const SP<YT> dp = SP<YT>(p.dst) + ((long long)n * p.C + c) * dplane;
  float bsum = 0.f;
  const long long e0 = (long long)blockIdx.x * 1024;
  if (p.pool_idx == nullptr && p.vec4) {  // planes of a multiple of 4 elements, 16-byte aligned views: one quad per thread
    const long long e = e0 + 4 * tid;
    if (e < dplane) {
      const float4 x4 = Sp.ld4(e);
      const float4 g4 = *reinterpret_cast<const float4*>(Gp + e);
      const float xs[4] = {x4.x, x4.y, x4.z, x4.w}, gs[4] = {g4.x, g4.y, g4.z, g4.w};
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float xh = (xs[r] - mean) * is;
        v[r] = st_round<YT>(sc * is * (gs[r] - k1 - xh * k2));  // the bias gradient sums what is stored
      }
      dp.st4(e, v[0], v[1], v[2], v[3]);
      if (p.dst16 != nullptr) (SP<ST_BF16>(p.dst16) + ((long long)n * p.C + c) * dplane).st4(e, v[0], v[1], v[2], v[3]);
      // same summation order per lane as the scalar form is not required: the block sum below is order-fixed
      bsum = (v[0] + v[1]) + (v[2] + v[3]);
    }
  } else if (p.pool_idx == nullptr) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long long e = e0 + i * 256 + tid;
      if (e < dplane) {
        const float xh = (Sp.ld1(e) - mean) * is;
        const float v = st_round<YT>(sc * is * (Gp[e] - k1 - xh * k2));  // the bias gradient sums what is stored
        dp.st1(e, v);
        if (p.dst16 != nullptr) (SP<ST_BF16>(p.dst16) + ((long long)n * p.C + c) * dplane).st1(e, v);
        bsum += v;
      }
    }
  } else {
    const unsigned char* ip = p.pool_idx + ((long long)n * p.C + c) * plane;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long long e = e0 + i * 256 + tid;
      if (e < dplane) {
        const int yd = (int)(e / p.Wd), xd = (int)(e - (long long)yd * p.Wd);
        const int py = yd >> 1, px = xd >> 1;
        float v = 0.f;
        if (py < p.H && px < p.W) {
          const long long q = (long long)py * p.W + px;
          if ((int)ip[q] == (((yd & 1) << 1) | (xd & 1))) {
            const float xh = (Sp.ld1(q) - mean) * is;
            v = st_round<YT>(sc * is * (Gp[q] - k1 - xh * k2));
          }
        }
        dp.st1(e, v);
        bsum += v;
      }
    }
  }
  if (p.bias_partial != nullptr) {
    __shared__ float red[4];
    bsum = wave_sum64(bsum);
    if ((tid & 63) == 0) red[tid >> 6] = bsum;
    __syncthreads();
    if (tid == 0) {
      const long long row = (long long)n * gridDim.x + blockIdx.x;
      p.bias_partial[row * p.C + c] = red[0] + red[1] + red[2] + red[3];
    }
  }
}

long long grad_finalize_rows(int N, int Hd, int Wd) {
  const long long nbx = ((long long)Hd * Wd + 1023) / 1024;
  return nbx * N;
}

int grad_finalize(const GradFinParams& p, int N, long long* rows, hipStream_t s) {
  const long long nbx = ((long long)p.Hd * p.Wd + 1023) / 1024;
  if (rows) *rows = nbx * N;
  const dim3 grid((unsigned)nbx, (unsigned)p.C, (unsigned)N);
  GradFinParams q = p;
  {
    const long long plane = (long long)p.H * p.W;
    auto al = [](const void* a, uintptr_t m) { return (reinterpret_cast<uintptr_t>(a) & m) == 0; };
    q.vec4 = (p.pool_idx == nullptr && (plane % 4) == 0 && (p.ns % 4) == 0 && al(p.G, 15) &&
              al(p.S, p.st == ST_BF16 ? 7 : 15) && al(p.dst, p.yt == ST_BF16 ? 7 : 15) && (p.dst16 == nullptr || al(p.dst16, 7)))
                 ? 1 : 0;
  }
  if (p.st == ST_BF16 && p.yt == ST_BF16)
    hipLaunchKernelGGL((grad_finalize_k<ST_BF16, ST_BF16>), grid, dim3(256), 0, s, q);
  else if (p.st == ST_BF16)
    hipLaunchKernelGGL((grad_finalize_k<ST_BF16, ST_F32>), grid, dim3(256), 0, s, q);
  else if (p.yt == ST_BF16)
    return -4;
  else
    hipLaunchKernelGGL((grad_finalize_k<ST_F32, ST_F32>), grid, dim3(256), 0, s, q);
  RLN_LAUNCH_CHECK();
}

__global__ __launch_bounds__(256) void splitk_finish_k(const float* __restrict__ part, int nsplit, long long split_stride,
                                                       int J, int HW, const float* __restrict__ bias,
                                                       const float* __restrict__ nscale, float* out, long long out_ns,
                                                       float* stat_partial) {
  const int j = blockIdx.y, n = blockIdx.z;
  const int px = blockIdx.x * 256 + threadIdx.x;
  float v = 0.f;
  if (px < HW) {
    const float* pp = part + ((long long)n * J + j) * HW + px;
    float acc = 0.f;
    for (int s = 0; s < nsplit; ++s) acc += pp[(long long)s * split_stride];
    const float sc = nscale ? nscale[(long long)n * J + j] : 1.f;
    v = (acc + (bias ? bias[j] : 0.f)) * sc;
    out[(long long)n * out_ns + (long long)j * HW + px] = v;
  }
  if (stat_partial != nullptr) {
    __shared__ float red[4][2];
    const float a1 = wave_sum64(v), a2 = wave_sum64(v * v);
    if ((threadIdx.x & 63) == 0) {
      red[threadIdx.x >> 6][0] = a1;
      red[threadIdx.x >> 6][1] = a2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      const long long row = (long long)n * gridDim.x + blockIdx.x;
      stat_partial[(row * J + j) * 2 + 0] = red[0][0] + red[1][0] + red[2][0] + red[3][0];
      stat_partial[(row * J + j) * 2 + 1] = red[0][1] + red[1][1] + red[2][1] + red[3][1];
    }
  }
}

int splitk_finish(const float* part, int nsplit, long long split_stride, int N, int J, int HW, const float* bias,
                  const float* nscale, float* out, long long out_ns, float* stat_partial, long long* nblk,
                  hipStream_t s) {
  const int gx = (HW + 255) / 256;
  if (nblk) *nblk = (long long)gx * N;
  hipLaunchKernelGGL(splitk_finish_k, dim3(gx, J, N), dim3(256), 0, s, part, nsplit, split_stride, J, HW, bias, nscale,
                     out, out_ns, stat_partial);
  RLN_LAUNCH_CHECK();
}

__device__ __forceinline__ void reduce_rows_body(long long vb, const float* __restrict__ src, long long rows,
                                                 long long len, float* dst) {
  const long long e = vb * 256 + threadIdx.x;
  if (e >= len) return;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  long long r = 0;
  for (; r + 3 < rows; r += 4) {
    a0 += src[r * len + e];
    a1 += src[(r + 1) * len + e];
    a2 += src[(r + 2) * len + e];
    a3 += src[(r + 3) * len + e];
  }
  for (; r < rows; ++r) a0 += src[r * len + e];
  dst[e] = (a0 + a1) + (a2 + a3);
}

__global__ __launch_bounds__(256) void reduce_rows_k(const float* __restrict__ src, long long rows, long long len,
                                                     float* dst) {
  reduce_rows_body((long long)blockIdx.x, src, rows, len, dst);
}

// few long columns (bias gradients): one wave per column, double accumulation
__device__ __forceinline__ void reduce_rows_tall_body(long long vb, const float* __restrict__ src, long long rows,
                                                      long long len, float* dst) {
  const int lane = threadIdx.x & 63;
  const long long e = vb * 4 + (threadIdx.x >> 6);
  if (e >= len) return;
  double a = 0.0;
#pragma unroll 8
  for (long long r = lane; r < rows; r += 64) a += (double)src[r * len + e];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
  if (lane == 0) dst[e] = (float)a;
}

__global__ __launch_bounds__(256) void reduce_rows_tall_k(const float* __restrict__ src, long long rows, long long len,
                                                          float* dst) {
  reduce_rows_tall_body((long long)blockIdx.x, src, rows, len, dst);
}

// The three small reductions that end a dense layer's backward (BatchNorm-backward sums of the data gradient, the
// weight-gradient slabs, the bias-gradient rows) in ONE launch: they are independent, and as separate launches
// each costs a ~5 us dependent-dispatch gap on top of its few microseconds of work.  Same arithmetic and
// summation order as the stand-alone kernels.
__global__ __launch_bounds__(256) void dense_tail_k(const DenseTail t) {
  long long b = blockIdx.x;
  if (b < t.nA) {
    bn_bwd_finalize_body((int)b, t.bn_partial, t.bn_rows, t.J, t.gamma, t.dgamma, t.dbeta, t.S1, t.S2);
    return;
  }
  b -= t.nA;
  if (b < t.nB) {
    if (t.w_tall) reduce_rows_tall_body(b, t.w_src, t.w_rows, t.w_len, t.w_dst);
    else reduce_rows_body(b, t.w_src, t.w_rows, t.w_len, t.w_dst);
    return;
  }
  b -= t.nB;
  if (t.b_tall) reduce_rows_tall_body(b, t.b_src, t.b_rows, t.b_len, t.b_dst);
  else reduce_rows_body(b, t.b_src, t.b_rows, t.b_len, t.b_dst);
}

static inline bool rows_tall(long long rows, long long len) { return len <= 4096 && rows >= 64; }

int dense_tail(DenseTail t, hipStream_t s) {
  t.nA = t.J;  // one block per channel (bn_bwd_finalize_body)
  t.w_tall = rows_tall(t.w_rows, t.w_len) ? 1 : 0;
  t.b_tall = rows_tall(t.b_rows, t.b_len) ? 1 : 0;
  t.nB = t.w_tall ? (t.w_len + 3) / 4 : (t.w_len + 255) / 256;
  const long long nC = t.b_tall ? (t.b_len + 3) / 4 : (t.b_len + 255) / 256;
  hipLaunchKernelGGL(dense_tail_k, dim3((unsigned)(t.nA + t.nB + nC)), dim3(256), 0, s, t);
  RLN_LAUNCH_CHECK();
}

int reduce_rows(const float* src, long long rows, long long len, float* dst, hipStream_t s) {
  if (rows_tall(rows, len)) {
    hipLaunchKernelGGL(reduce_rows_tall_k, dim3((unsigned)((len + 3) / 4)), dim3(256), 0, s, src, rows, len, dst);
  } else {
    hipLaunchKernelGGL(reduce_rows_k, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, s, src, rows, len, dst);
  }
  RLN_LAUNCH_CHECK();
}

// =============================================================================================
// head: F.normalize (tiramisu.py:105) + 1x1 classifier, /T, softmax (tiramisu.py:120-125)
// =============================================================================================

constexpr int HEAD_MAXC = 1024;  // feature channels supported by the LDS weight image

template <int NC>
__device__ __forceinline__ void load_head_weights(float* wt, float* bt, const HeadParams& p) {
  for (int e = threadIdx.x; e < p.C * NC; e += 256) {
    const int c = e / NC, k = e - c * NC;
    wt[e] = (k < p.ncls) ? p.w[(long long)k * p.C + c] : 0.f;
  }
  if (threadIdx.x < NC) bt[threadIdx.x] = (threadIdx.x < p.ncls) ? p.b[threadIdx.x] : 0.f;
}

template <int NC>
__device__ __forceinline__ void softmax_nc(float* l, int ncls) {
  float m = l[0];
#pragma unroll
  for (int k = 1; k < NC; ++k)
    if (k < ncls) m = fmaxf(m, l[k]);
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    l[k] = (k < ncls) ? expf(l[k] - m) : 0.f;
    sum += l[k];
  }
  const float inv = 1.f / sum;
#pragma unroll
  for (int k = 0; k < NC; ++k) l[k] *= inv;
}

// NORMALIZE: input is the raw stack (fused feature extractor tail); else input already normalised
template <int NC, bool NORMALIZE, int ST>
__global__ __launch_bounds__(256) void head_fwd_k(const HeadParams p, float* out, int use_softmax, float* feat_out) {
  extern __shared__ __align__(16) float hsm[];
  float* wt = hsm;
  float* bt = hsm + (long long)p.C * NC;
  load_head_weights<NC>(wt, bt, p);
  __syncthreads();
  const int n = blockIdx.y;
  const long long px = (long long)blockIdx.x * 256 + threadIdx.x;
  if (px >= p.HW) return;
  const SP<ST> xp = SP<ST>(p.S) + ((long long)n * p.ns + px);
  float d[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) d[k] = 0.f;
  float ss = 0.f;
#pragma unroll 8
  for (int c = 0; c < p.C; ++c) {
    const float x = xp.ld1((long long)c * p.HW);
    ss = fmaf(x, x, ss);
#pragma unroll
    for (int k = 0; k < NC; ++k) d[k] = fmaf(wt[c * NC + k], x, d[k]);
  }
  float inv = 1.f;
  if (NORMALIZE) inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
  float l[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) l[k] = (d[k] * inv + bt[k]) / p.T;
  if (use_softmax) softmax_nc<NC>(l, p.ncls);
  if (out != nullptr) {
#pragma unroll
    for (int k = 0; k < NC; ++k)
      if (k < p.ncls) out[((long long)n * p.ncls + k) * p.HW + px] = l[k];
  }
  if (feat_out != nullptr) {
    float* fp = feat_out + (long long)n * p.C * p.HW + px;
#pragma unroll 8
    for (int c = 0; c < p.C; ++c) fp[(long long)c * p.HW] = xp.ld1((long long)c * p.HW) * inv;
  }
}

template <bool NORMALIZE>
static int head_launch(const HeadParams& p, int N, float* out, int use_softmax, float* feat_out, hipStream_t s) {
  if (p.C > HEAD_MAXC || p.ncls > 16 || p.ncls < 1) return -4;
  dim3 grid((unsigned)((p.HW + 255) / 256), (unsigned)N);
  if (p.st == ST_BF16) {
    if (p.ncls <= 4) {
      hipLaunchKernelGGL((head_fwd_k<4, NORMALIZE, ST_BF16>), grid, dim3(256), (p.C * 4 + 4) * 4, s, p, out, use_softmax,
                         feat_out);
    } else if (p.ncls <= 8) {
      hipLaunchKernelGGL((head_fwd_k<8, NORMALIZE, ST_BF16>), grid, dim3(256), (p.C * 8 + 8) * 4, s, p, out, use_softmax,
                         feat_out);
    } else {
      hipLaunchKernelGGL((head_fwd_k<16, NORMALIZE, ST_BF16>), grid, dim3(256), (p.C * 16 + 16) * 4, s, p, out,
                         use_softmax, feat_out);
    }
  } else if (p.ncls <= 4) {
    hipLaunchKernelGGL((head_fwd_k<4, NORMALIZE, ST_F32>), grid, dim3(256), (p.C * 4 + 4) * 4, s, p, out, use_softmax,
                       feat_out);
  } else if (p.ncls <= 8) {
    hipLaunchKernelGGL((head_fwd_k<8, NORMALIZE, ST_F32>), grid, dim3(256), (p.C * 8 + 8) * 4, s, p, out, use_softmax,
                       feat_out);
  } else {
    hipLaunchKernelGGL((head_fwd_k<16, NORMALIZE, ST_F32>), grid, dim3(256), (p.C * 16 + 16) * 4, s, p, out, use_softmax,
                       feat_out);
  }
  RLN_LAUNCH_CHECK();
}

int head_forward(const HeadParams& p, int N, float* out, int use_softmax, float* feat_out, hipStream_t s) {
  return head_launch<true>(p, N, out, use_softmax, feat_out, s);
}
int classifier_forward(const HeadParams& p, int N, float* out, int use_softmax, hipStream_t s) {
  return head_launch<false>(p, N, out, use_softmax, nullptr, s);
}

// =============================================================================================
// loss: class histogram -> weights, CE on probabilities (double softmax), argmax, accuracy
// (SimpleTrain.py:15-20, TrainingBase.py:12-23,84-87)
// =============================================================================================

__global__ __launch_bounds__(256) void loss_hist_k(const long long* __restrict__ y, long long npix, int ncls,
                                                   int* counts) {
  __shared__ int h[17];
  if (threadIdx.x < 17) h[threadIdx.x] = 0;
  __syncthreads();
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long long)gridDim.x * 256) {
    const long long v = y[i];
    const int k = (v >= 0 && v < ncls) ? (int)v : 16;
    atomicAdd(&h[k], 1);
  }
  __syncthreads();
  if (threadIdx.x < ncls && h[threadIdx.x]) atomicAdd(&counts[threadIdx.x], h[threadIdx.x]);
  if (threadIdx.x == 16 && h[16]) atomicAdd(&counts[ncls], h[16]);
}

__global__ void loss_weights_k(const int* counts, int ncls, int weighted, float* result) {
  const int k = threadIdx.x;
  if (k < ncls) result[4 + k] = weighted ? (1.0f / (float)counts[k]) : 1.0f;
}

constexpr int LOSS_PIX_PER_BLOCK = 1024;
long long loss_blocks(long long npix) { return (npix + LOSS_PIX_PER_BLOCK - 1) / LOSS_PIX_PER_BLOCK; }

template <int NC>
__global__ __launch_bounds__(256) void loss_main_k(const float* __restrict__ probs, const long long* __restrict__ y,
                                                   int ncls, long long HW, long long npix,
                                                   const float* __restrict__ result, float* partial,
                                                   long long* argmax_out, int* conf) {
  float num = 0.f, den = 0.f, cor = 0.f;
#pragma unroll
  for (int i = 0; i < LOSS_PIX_PER_BLOCK / 256; ++i) {
    const long long g = (long long)blockIdx.x * LOSS_PIX_PER_BLOCK + i * 256 + threadIdx.x;
    if (g < npix) {
      const long long n = g / HW, px = g - n * HW;
      float pr[NC];
#pragma unroll
      for (int k = 0; k < NC; ++k) pr[k] = (k < ncls) ? probs[(n * ncls + k) * HW + px] : 0.f;
      float m = pr[0];
      int am = 0;
#pragma unroll
      for (int k = 1; k < NC; ++k)
        if (k < ncls && pr[k] > m) {
          m = pr[k];
          am = k;
        }
      float se = 0.f;
#pragma unroll
      for (int k = 0; k < NC; ++k)
        if (k < ncls) se += expf(pr[k] - m);
      const float lse = m + logf(se);
      const long long lab = y[g];
      if (lab >= 0 && lab < ncls) {
        float py = 0.f;
#pragma unroll
        for (int k = 0; k < NC; ++k)
          if (k == (int)lab) py = pr[k];
        const float w = result[4 + (int)lab];
        num += w * (lse - py);
        den += w;
        cor += (am == (int)lab) ? 1.f : 0.f;
        if (conf != nullptr) atomicAdd(&conf[(int)lab * ncls + am], 1);
      }
      if (argmax_out != nullptr) argmax_out[g] = am;
    }
  }
  __shared__ float red[4][3];
  num = wave_sum64(num);
  den = wave_sum64(den);
  cor = wave_sum64(cor);
  if ((threadIdx.x & 63) == 0) {
    red[threadIdx.x >> 6][0] = num;
    red[threadIdx.x >> 6][1] = den;
    red[threadIdx.x >> 6][2] = cor;
  }
  __syncthreads();
  if (threadIdx.x < 3)
    partial[(long long)blockIdx.x * 4 + threadIdx.x] =
        red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__global__ __launch_bounds__(256) void loss_finalize_k(const float* __restrict__ partial, long long nblk, long long npix,
                                                       int ncls, const int* counts, float* result, float* out,
                                                       const int* conf, long long* confusion_out) {
  double a[3] = {0.0, 0.0, 0.0};
  for (long long b = threadIdx.x; b < nblk; b += 256) {
    a[0] += (double)partial[b * 4 + 0];
    a[1] += (double)partial[b * 4 + 1];
    a[2] += (double)partial[b * 4 + 2];
  }
  __shared__ double red[4][3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a[k] += __shfl_xor(a[k], o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = a[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double num = red[0][0] + red[1][0] + red[2][0] + red[3][0];
    const double den = red[0][1] + red[1][1] + red[2][1] + red[3][1];
    const double cor = red[0][2] + red[1][2] + red[2][2] + red[3][2];
    result[0] = (float)num;
    result[1] = (float)den;
    result[2] = (float)cor;
    if (out != nullptr) {
      out[0] = (float)(num / den);
      out[1] = (float)(cor / (double)npix);
      out[2] = (float)counts[ncls];
      for (int k = 0; k < ncls; ++k) out[3 + k] = (float)counts[k];
    }
  }
  if (confusion_out != nullptr && conf != nullptr && threadIdx.x < ncls * ncls)
    confusion_out[threadIdx.x] = (long long)conf[threadIdx.x];
}

int loss_forward(const float* probs, const long long* y, int N, int ncls, int HW, int weighted, const LossScratch& sc,
                 float* out, long long* argmax_out, long long* confusion_out, hipStream_t s) {
  if (ncls > 16 || ncls < 1) return -4;
  const long long npix = (long long)N * HW;
  int* conf = confusion_out ? (sc.counts + 32) : nullptr;
  hipError_t e = hipMemsetAsync(sc.counts, 0, sizeof(int) * (32 + 256), s);
  if (e != hipSuccess) return (int)e;
  long long hb = (npix + 255) / 256;
  if (hb > 1024) hb = 1024;
  hipLaunchKernelGGL(loss_hist_k, dim3((unsigned)hb), dim3(256), 0, s, y, npix, ncls, sc.counts);
  hipLaunchKernelGGL(loss_weights_k, dim3(1), dim3(64), 0, s, sc.counts, ncls, weighted, sc.result);
  const long long nblk = loss_blocks(npix);
  if (ncls <= 4) {
    hipLaunchKernelGGL(loss_main_k<4>, dim3((unsigned)nblk), dim3(256), 0, s, probs, y, ncls, (long long)HW, npix,
                       sc.result, sc.partial, argmax_out, conf);
  } else if (ncls <= 8) {
    hipLaunchKernelGGL(loss_main_k<8>, dim3((unsigned)nblk), dim3(256), 0, s, probs, y, ncls, (long long)HW, npix,
                       sc.result, sc.partial, argmax_out, conf);
  } else {
    hipLaunchKernelGGL(loss_main_k<16>, dim3((unsigned)nblk), dim3(256), 0, s, probs, y, ncls, (long long)HW, npix,
                       sc.result, sc.partial, argmax_out, conf);
  }
  hipLaunchKernelGGL(loss_finalize_k, dim3(1), dim3(256), 0, s, sc.partial, nblk, npix, ncls, sc.counts, sc.result, out,
                     conf, confusion_out);
  RLN_LAUNCH_CHECK();
}

// =============================================================================================
// head backward.  Per pixel: recompute forward, d(loss)/d(probabilities) of the weighted CE on
// probabilities, back through softmax, /T, 1x1 conv and F.normalize.
// =============================================================================================

template <int NC>
__global__ __launch_bounds__(256) void head_bwd_data_k(const HeadBwdParams q) {
  const float ls = q.loss_scale * (q.loss_scale_dev ? *q.loss_scale_dev : 1.f);  // d(loss) from autograd stays on the device
  const HeadParams& p = q.h;
  extern __shared__ __align__(16) float hsm[];
  float* wt = hsm;
  float* bt = hsm + (long long)p.C * NC;
  float* red = bt + NC;  // [4][NC]
  load_head_weights<NC>(wt, bt, p);
  __syncthreads();
  const int n = blockIdx.y;
  const long long px = (long long)blockIdx.x * 256 + threadIdx.x;
  const bool active = px < p.HW;
  float gl[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) gl[k] = 0.f;
  if (active) {
    const float* xp = p.S + (long long)n * p.ns + px;
    float d[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) d[k] = 0.f;
    float ss = 0.f;
#pragma unroll 8
    for (int c = 0; c < p.C; ++c) {
      const float x = xp[(long long)c * p.HW];
      ss = fmaf(x, x, ss);
#pragma unroll
      for (int k = 0; k < NC; ++k) d[k] = fmaf(wt[c * NC + k], x, d[k]);
    }
    const float nrm = sqrtf(ss);
    const bool proj = nrm > 1e-12f;
    const float inv = 1.f / fmaxf(nrm, 1e-12f);
    float pr[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) pr[k] = (d[k] * inv + bt[k]) / p.T;
    softmax_nc<NC>(pr, p.ncls);
    // second softmax (cross_entropy applies log_softmax to the probabilities)
    float qq[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) qq[k] = pr[k];
    softmax_nc<NC>(qq, p.ncls);
    float gp[NC];
    float dot = 0.f;
    if (q.mode == 0) {
      const long long lab = q.y[(long long)n * p.HW + px];
      float wy = 0.f;
      if (lab >= 0 && lab < p.ncls) wy = q.lossres[4 + (int)lab];
      const float coef = ls * wy / q.lossres[1];
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        gp[k] = (k < p.ncls) ? coef * (qq[k] - ((k == (int)lab) ? 1.f : 0.f)) : 0.f;
        dot = fmaf(pr[k], gp[k], dot);
      }
    } else if (q.mode == 2) {  // caller-supplied d(loss)/d(probabilities): differentiable module forward
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        gp[k] = (k < p.ncls) ? ls * q.gext[((long long)n * p.ncls + k) * p.HW + px] : 0.f;
        dot = fmaf(pr[k], gp[k], dot);
      }
    } else {  // d/dp of lamda * mean(sum_k p*log(p+1e-5))
      const float coef = ls * q.lamda * q.inv_count;
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        gp[k] = (k < p.ncls) ? coef * (logf(pr[k] + 1e-5f) + pr[k] / (pr[k] + 1e-5f)) : 0.f;
        dot = fmaf(pr[k], gp[k], dot);
      }
    }
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      gl[k] = (k < p.ncls) ? pr[k] * (gp[k] - dot) / p.T : 0.f;
      t = fmaf(gl[k], d[k], t);
    }
    t = proj ? t * inv : 0.f;  // = sum_c gxn[c] * xn[c]
    float* gout = q.G + (long long)n * q.g_ns + px;
#pragma unroll 8
    for (int c = 0; c < p.C; ++c) {
      const float x = xp[(long long)c * p.HW];
      float gxn = 0.f;
#pragma unroll
      for (int k = 0; k < NC; ++k) gxn = fmaf(wt[c * NC + k], gl[k], gxn);
      const float gx = q.feat_sign * (gxn - x * inv * t) * inv;
      gout[(long long)c * p.HW] = gx / q.invstd[c];
    }
#pragma unroll
    for (int k = 0; k < NC; ++k)
      if (k < p.ncls) q.glin[((long long)n * p.ncls + k) * p.HW + px] = gl[k] * inv;
  }
  // bias gradient partial: sum of gl[k] over the block
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    const float v = wave_sum64(gl[k]);
    if ((threadIdx.x & 63) == 0) red[(threadIdx.x >> 6) * NC + k] = v;
  }
  __syncthreads();
  if (threadIdx.x < p.ncls) {
    const int k = threadIdx.x;
    const long long row = (long long)n * gridDim.x + blockIdx.x;
    q.bias_partial[row * p.ncls + k] = red[k] + red[NC + k] + red[2 * NC + k] + red[3 * NC + k];
  }
}

int head_backward_data(const HeadBwdParams& q, int N, long long* bias_rows, hipStream_t s) {
  const HeadParams& p = q.h;
  if (p.C > HEAD_MAXC || p.ncls > 16 || p.ncls < 1) return -4;
  dim3 grid((unsigned)((p.HW + 255) / 256), (unsigned)N);
  if (bias_rows) *bias_rows = (long long)grid.x * N;
  if (p.ncls <= 4) {
    hipLaunchKernelGGL(head_bwd_data_k<4>, grid, dim3(256), (p.C * 4 + 4 + 16) * 4, s, q);
  } else if (p.ncls <= 8) {
    hipLaunchKernelGGL(head_bwd_data_k<8>, grid, dim3(256), (p.C * 8 + 8 + 32) * 4, s, q);
  } else {
    hipLaunchKernelGGL(head_bwd_data_k<16>, grid, dim3(256), (p.C * 16 + 16 + 64) * 4, s, q);
  }
  RLN_LAUNCH_CHECK();
}

// dW[k][c] partial per sample: block = (channel c, sample n)
template <int NC>
__global__ __launch_bounds__(256) void head_bwd_weight_k(const HeadParams p, const float* __restrict__ glin,
                                                         float* partial) {
  const int c = blockIdx.x, n = blockIdx.y;
  const float* xp = p.S + (long long)n * p.ns + (long long)c * p.HW;
  const float* gp = glin + (long long)n * p.ncls * p.HW;
  float acc[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) acc[k] = 0.f;
  for (long long px = threadIdx.x; px < p.HW; px += 256) {
    const float x = xp[px];
#pragma unroll
    for (int k = 0; k < NC; ++k)
      if (k < p.ncls) acc[k] = fmaf(gp[(long long)k * p.HW + px], x, acc[k]);
  }
  __shared__ float red[4][NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    const float v = wave_sum64(acc[k]);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < p.ncls) {
    const int k = threadIdx.x;
    partial[(long long)n * p.ncls * p.C + (long long)k * p.C + c] = red[0][k] + red[1][k] + red[2][k] + red[3][k];
  }
}

int head_backward_weight(const HeadParams& p, int N, const float* glin, float* partial, hipStream_t s) {
  dim3 grid((unsigned)p.C, (unsigned)N);
  if (p.ncls <= 4) {
    hipLaunchKernelGGL(head_bwd_weight_k<4>, grid, dim3(256), 0, s, p, glin, partial);
  } else if (p.ncls <= 8) {
    hipLaunchKernelGGL(head_bwd_weight_k<8>, grid, dim3(256), 0, s, p, glin, partial);
  } else {
    hipLaunchKernelGGL(head_bwd_weight_k<16>, grid, dim3(256), 0, s, p, glin, partial);
  }
  RLN_LAUNCH_CHECK();
}

// Fused head backward: ONE pass over the feature stack.  A block walks HEAD_TPB tiles of 32 pixels; each tile's
// C x 32 feature values are staged in LDS while it (1) recomputes the forward head, (2) writes the gradient of the
// features (x read from LDS, not a second time from HBM) and (3) adds the tile's share of the classifier weight
// gradient dW[k][c] += sum_px glin[k][px] * x[c][px] into thread-owned registers.  Per block one partial row of
// dW (ncls*C) and of db (ncls); a row reduction in fixed order follows (bitwise reproducible).
// Threads: lane & 31 = pixel of the tile, tid >> 5 = one of 8 channel groups (c = g + 8 i).
constexpr int HEAD_TPX = 32;   // pixels per tile
constexpr int HEAD_TPB = 10;   // tiles per block
constexpr int HEAD_XS = 33;    // LDS row stride of the staged tile (odd: conflict-free for both access patterns)

template <int NC, int ST>  // ST: storage element type of the feature stack
__global__ __launch_bounds__(256) void head_bwd_fused_k(const HeadBwdParams q, float* __restrict__ wpartial) {
  const float ls = q.loss_scale * (q.loss_scale_dev ? *q.loss_scale_dev : 1.f);  // d(loss) from autograd stays on the device
  const HeadParams& p = q.h;
  extern __shared__ __align__(16) float hsm[];
  float* wt = hsm;                                  // [C][NC]
  float* bt = wt + (long long)p.C * NC;             // [NC]
  float* red = bt + NC;                             // [8][1 + NC][32]
  float* gls = red + 8 * (1 + NC) * HEAD_TPX;       // [NC][32]   glin = gl * inv
  float* xs = gls + NC * HEAD_TPX;                  // [C][HEAD_XS]
  float* sdt = xs + (long long)p.C * HEAD_XS;       // [C] 1 / invstd (a multiply per element instead of a divide)
  load_head_weights<NC>(wt, bt, p);
  for (int e = threadIdx.x; e < p.C; e += 256) sdt[e] = 1.f / q.invstd[e];
  const int tid = threadIdx.x, pl = tid & 31, g = tid >> 5;
  const int n = blockIdx.y;
  const SP<ST> Sn = SP<ST>(p.S) + (long long)n * p.ns;
  float* Gn = q.G + (long long)n * q.g_ns;
  float wacc[2][NC], bsum[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) wacc[0][k] = wacc[1][k] = bsum[k] = 0.f;
  const int ntile = (p.HW + HEAD_TPX - 1) / HEAD_TPX;
  const int t0 = blockIdx.x * HEAD_TPB, t1 = min(ntile, t0 + HEAD_TPB);
  for (int tile = t0; tile < t1; ++tile) {
    __syncthreads();  // previous tile fully consumed (first pass: weights visible)
    const int px = tile * HEAD_TPX + pl;
    const bool active = px < p.HW;
    const int pxs = active ? px : p.HW - 1;
    // ---- (1) stage + partial dot products of this thread's channels ----
    float ss = 0.f, d[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) d[k] = 0.f;
#pragma unroll 18
    for (int c = g; c < p.C; c += 8) {
      const float x = Sn.ld1((long long)c * p.HW + pxs);
      xs[c * HEAD_XS + pl] = x;
      ss = fmaf(x, x, ss);
#pragma unroll
      for (int k = 0; k < NC; ++k) d[k] = fmaf(wt[c * NC + k], x, d[k]);
    }
    red[(g * (1 + NC)) * HEAD_TPX + pl] = ss;
#pragma unroll
    for (int k = 0; k < NC; ++k) red[(g * (1 + NC) + 1 + k) * HEAD_TPX + pl] = d[k];
    __syncthreads();
    ss = 0.f;
#pragma unroll
    for (int k = 0; k < NC; ++k) d[k] = 0.f;
#pragma unroll
    for (int gg = 0; gg < 8; ++gg) {  // fixed order: every group computes the identical per-pixel values
      ss += red[(gg * (1 + NC)) * HEAD_TPX + pl];
#pragma unroll
      for (int k = 0; k < NC; ++k) d[k] += red[(gg * (1 + NC) + 1 + k) * HEAD_TPX + pl];
    }
    // ---- per-pixel head: forward recompute, d loss / d logits, back through softmax, /T, normalize ----
    const float nrm = sqrtf(ss);
    const bool proj = nrm > 1e-12f;
    const float inv = 1.f / fmaxf(nrm, 1e-12f);
    float pr[NC], qq[NC], gp[NC], gl[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) pr[k] = (d[k] * inv + bt[k]) / p.T;
    softmax_nc<NC>(pr, p.ncls);
#pragma unroll
    for (int k = 0; k < NC; ++k) qq[k] = pr[k];
    softmax_nc<NC>(qq, p.ncls);  // cross_entropy applies log_softmax to the probabilities
    float dot = 0.f;
    if (q.mode == 0) {
      const long long lab = q.y[(long long)n * p.HW + pxs];
      float wy = 0.f;
      if (lab >= 0 && lab < p.ncls) wy = q.lossres[4 + (int)lab];
      const float coef = ls * wy / q.lossres[1];
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        gp[k] = (k < p.ncls) ? coef * (qq[k] - ((k == (int)lab) ? 1.f : 0.f)) : 0.f;
        dot = fmaf(pr[k], gp[k], dot);
      }
    } else if (q.mode == 2) {  // caller-supplied d(loss)/d(probabilities): differentiable module forward
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        gp[k] = (k < p.ncls) ? ls * q.gext[((long long)n * p.ncls + k) * p.HW + pxs] : 0.f;
        dot = fmaf(pr[k], gp[k], dot);
      }
    } else {  // d/dp of lamda * mean(sum_k p*log(p+1e-5))
      const float coef = ls * q.lamda * q.inv_count;
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        gp[k] = (k < p.ncls) ? coef * (logf(pr[k] + 1e-5f) + pr[k] / (pr[k] + 1e-5f)) : 0.f;
        dot = fmaf(pr[k], gp[k], dot);
      }
    }
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      gl[k] = (active && k < p.ncls) ? pr[k] * (gp[k] - dot) / p.T : 0.f;
      t = fmaf(gl[k], d[k], t);
    }
    t = proj ? t * inv : 0.f;  // = sum_c gxn[c] * xn[c]
    if (g == 0) {
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        gls[k * HEAD_TPX + pl] = gl[k] * inv;
        bsum[k] += gl[k];
      }
    }
    // ---- (2) feature gradient of this thread's channels ----
#pragma unroll 18
    for (int c = g; c < p.C; c += 8) {
      const float x = xs[c * HEAD_XS + pl];
      float gxn = 0.f;
#pragma unroll
      for (int k = 0; k < NC; ++k) gxn = fmaf(wt[c * NC + k], gl[k], gxn);
      const float gx = q.feat_sign * (gxn - x * inv * t) * inv;
      if (active) Gn[(long long)c * p.HW + px] = gx * sdt[c];
    }
    __syncthreads();  // gls visible
    // ---- (3) classifier weight gradient: thread-owned channels tid and tid + 256 ----
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
      const int c = tid + 256 * sl;
      if (c < p.C) {
#pragma unroll 8
        for (int j = 0; j < HEAD_TPX; ++j) {
          const float x = xs[c * HEAD_XS + j];
#pragma unroll
          for (int k = 0; k < NC; ++k) wacc[sl][k] = fmaf(gls[k * HEAD_TPX + j], x, wacc[sl][k]);
        }
      }
    }
  }
  const long long row = (long long)n * gridDim.x + blockIdx.x;
#pragma unroll
  for (int sl = 0; sl < 2; ++sl) {
    const int c = tid + 256 * sl;
    if (c < p.C) {
#pragma unroll
      for (int k = 0; k < NC; ++k)
        if (k < p.ncls) wpartial[(row * p.ncls + k) * p.C + c] = wacc[sl][k];
    }
  }
  if (tid < 64) {  // group 0 = lanes 0..31 of wave 0 hold the bias sums; lanes 32..63 (group 1) contribute 0
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      const float v = wave_sum64(g == 0 ? bsum[k] : 0.f);
      if (tid == 0 && k < p.ncls) q.bias_partial[row * p.ncls + k] = v;
    }
  }
}

long long head_backward_rows(int N, int HW) {
  const int ntile = (HW + HEAD_TPX - 1) / HEAD_TPX;
  return (long long)N * ((ntile + HEAD_TPB - 1) / HEAD_TPB);
}

// feature gradient + partial rows of dW ([rows][ncls*C]) and db ([rows][ncls]); rows = head_backward_rows(N, HW)
int head_backward_fused(const HeadBwdParams& q, int N, float* wpartial, long long* rows, hipStream_t s) {
  const HeadParams& p = q.h;
  if (p.C > 512 || p.ncls > 16 || p.ncls < 1) return -4;
  const int ntile = (p.HW + HEAD_TPX - 1) / HEAD_TPX;
  dim3 grid((unsigned)((ntile + HEAD_TPB - 1) / HEAD_TPB), (unsigned)N);
  if (rows) *rows = (long long)grid.x * N;
  auto lds = [&](int nc) {
    return (size_t)(p.C * nc + nc + 8 * (1 + nc) * HEAD_TPX + nc * HEAD_TPX + p.C * HEAD_XS + p.C) * 4;
  };
#define RLN_HEAD_FUSED(NCV, STV)                                                                                    \
  {                                                                                                                 \
    static DevOnce attr_once;                                                                                       \
    if (attr_once.first()) {                                                                                        \
      const hipError_t attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(head_bwd_fused_k<NCV, STV>),    \
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);     \
      if (attr_err != hipSuccess) {                                                                                 \
        (void)hipGetLastError();                                                                                    \
        attr_once.undo();                                                                                           \
        return (int)attr_err;                                                                                       \
      }                                                                                                             \
    }                                                                                                               \
    hipLaunchKernelGGL((head_bwd_fused_k<NCV, STV>), grid, dim3(256), lds(NCV), s, q, wpartial);                    \
  }
  if (p.st == ST_BF16) {
    if (p.ncls <= 4) RLN_HEAD_FUSED(4, ST_BF16)
    else if (p.ncls <= 8) RLN_HEAD_FUSED(8, ST_BF16)
    else RLN_HEAD_FUSED(16, ST_BF16)
  } else {
    if (p.ncls <= 4) RLN_HEAD_FUSED(4, ST_F32)
    else if (p.ncls <= 8) RLN_HEAD_FUSED(8, ST_F32)
    else RLN_HEAD_FUSED(16, ST_F32)
  }
#undef RLN_HEAD_FUSED
  RLN_LAUNCH_CHECK();
}

// =============================================================================================
// Input transform on device (dataManagement/myTransforms.py:15-19, non-augmenting branch):
// Resize(height, width) -> [ToGray] -> Normalize() -> ToTensorV2, for uint8 HWC frames resident in HBM; the label
// mask is resized with nearest neighbour.  Resize follows the published fixed-point algorithm of cv2.resize
// INTER_LINEAR for 8-bit images (the library albumentations 0.5.2 calls; neither ships with the reference, so this
// row's parity is UNPINNED): source coordinate (d + 0.5) * scale - 0.5, 11-bit coefficients, horizontal pass in
// int32, vertical pass ((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2.
// =============================================================================================
__device__ __forceinline__ void lin_coef(int d, int dst, int src, int* s0, int* s1, int* c0, int* c1) {
  const double scale = (double)src / (double)dst;
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int i = (int)floorf(f);
  f -= (float)i;
  if (i < 0) {
    i = 0;
    f = 0.f;
  }
  if (i >= src - 1) {
    i = src - 1;
    f = 0.f;
  }
  *s0 = i;
  *s1 = min(i + 1, src - 1);
  const float k0 = (1.f - f) * 2048.f, k1 = f * 2048.f;
  *c0 = (int)rintf(k0);
  *c1 = (int)rintf(k1);
}

__global__ __launch_bounds__(256) void preprocess_u8_k(const unsigned char* __restrict__ frames, int hs, int ws,
                                                       const unsigned char* __restrict__ labels, int h, int w, int gray,
                                                       float m0, float m1, float m2, float i0, float i1, float i2,
                                                       float* __restrict__ x, long long* __restrict__ y) {
  const int n = blockIdx.y;
  const int px = blockIdx.x * 256 + threadIdx.x;
  if (px >= h * w) return;
  const int dy = px / w, dx = px - dy * w;
  int x0, x1, a0, a1, y0, y1, b0, b1;
  lin_coef(dx, w, ws, &x0, &x1, &a0, &a1);
  lin_coef(dy, h, hs, &y0, &y1, &b0, &b1);
  const unsigned char* f = frames + (long long)n * hs * ws * 3;
  int v[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int r0 = f[((long long)y0 * ws + x0) * 3 + c] * a0 + f[((long long)y0 * ws + x1) * 3 + c] * a1;
    const int r1 = f[((long long)y1 * ws + x0) * 3 + c] * a0 + f[((long long)y1 * ws + x1) * 3 + c] * a1;
    int o = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
    v[c] = min(max(o, 0), 255);
  }
  if (gray) {  // ToGray: cv2 RGB2GRAY fixed point on the stored channel order, replicated to 3 channels
    const int g = (v[0] * 4899 + v[1] * 9617 + v[2] * 1868 + (1 << 13)) >> 14;
    v[0] = v[1] = v[2] = g;
  }
  const long long hw = (long long)h * w;
  float* xo = x + (long long)n * 3 * hw + px;
  xo[0] = ((float)v[0] - m0) * i0;  // Normalize: (img - mean*255) * (1 / (std*255)), float32
  xo[hw] = ((float)v[1] - m1) * i1;
  xo[2 * hw] = ((float)v[2] - m2) * i2;
  if (labels != nullptr && y != nullptr) {  // cv2 INTER_NEAREST: floor(d * scale)
    const int sy = min((int)floor((double)dy * ((double)hs / (double)h)), hs - 1);
    const int sx = min((int)floor((double)dx * ((double)ws / (double)w)), ws - 1);
    y[(long long)n * hw + px] = (long long)labels[((long long)n * hs + sy) * ws + sx];
  }
}

int preprocess_u8(const unsigned char* frames, int N, int hs, int ws, const unsigned char* labels, int h, int w, int gray,
                  const float* mean3, const float* std3, float* x, long long* y, hipStream_t s) {
  const float m0 = mean3[0] * 255.f, m1 = mean3[1] * 255.f, m2 = mean3[2] * 255.f;
  const float i0 = 1.f / (std3[0] * 255.f), i1 = 1.f / (std3[1] * 255.f), i2 = 1.f / (std3[2] * 255.f);
  dim3 grid((unsigned)((h * w + 255) / 256), (unsigned)N);
  hipLaunchKernelGGL(preprocess_u8_k, grid, dim3(256), 0, s, frames, hs, ws, labels, h, w, gray, m0, m1, m2, i0, i1, i2,
                     x, y);
  RLN_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------
// Demo overlay (makeDemoVideo.py:36-46): pred = argmax_k probs (first maximum wins, torch.max(out, 1)), frame_out =
// cv2.resize(frame, (w, h)) [the script passes INTER_LANCZOS4 in the `dst` position, so the interpolation is the
// default INTER_LINEAR: the same 8-bit fixed-point resize as above], then frame_out[pred == k] = colour[k] for the
// classes the script paints.  One pass: uint8 HWC frame + probabilities in, uint8 HWC frame out.
struct OverlayColors {
  unsigned char bgr[16][3];
  unsigned paint;  // bit k set -> class k is painted
};
__global__ __launch_bounds__(256) void overlay_u8_k(const unsigned char* __restrict__ frames, int hs, int ws,
                                                    const float* __restrict__ probs, int ncls, int h, int w,
                                                    OverlayColors col, unsigned char* __restrict__ out,
                                                    unsigned char* __restrict__ pred_out) {
  const int n = blockIdx.y;
  const int px = blockIdx.x * 256 + threadIdx.x;
  if (px >= h * w) return;
  const long long hw = (long long)h * w;
  const float* pr = probs + (long long)n * ncls * hw + px;
  int best = 0;
  float bv = pr[0];
  for (int k = 1; k < ncls; ++k) {
    const float v = pr[k * hw];
    if (v > bv) {
      bv = v;
      best = k;
    }
  }
  if (pred_out) pred_out[(long long)n * hw + px] = (unsigned char)best;
  unsigned char* o = out + ((long long)n * hw + px) * 3;
  if ((col.paint >> best) & 1u) {
    o[0] = col.bgr[best][0];
    o[1] = col.bgr[best][1];
    o[2] = col.bgr[best][2];
    return;
  }
  const int dy = px / w, dx = px - dy * w;
  int x0, x1, a0, a1, y0, y1, b0, b1;
  lin_coef(dx, w, ws, &x0, &x1, &a0, &a1);
  lin_coef(dy, h, hs, &y0, &y1, &b0, &b1);
  const unsigned char* f = frames + (long long)n * hs * ws * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int r0 = f[((long long)y0 * ws + x0) * 3 + c] * a0 + f[((long long)y0 * ws + x1) * 3 + c] * a1;
    const int r1 = f[((long long)y1 * ws + x0) * 3 + c] * a0 + f[((long long)y1 * ws + x1) * 3 + c] * a1;
    const int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
    o[c] = (unsigned char)min(max(v, 0), 255);
  }
}

int overlay_u8(const unsigned char* frames, int N, int hs, int ws, const float* probs, int ncls, int h, int w,
               const unsigned char* colors_host, unsigned paint_mask, unsigned char* out, unsigned char* pred_out,
               hipStream_t s) {
  OverlayColors col;
  memset(&col, 0, sizeof(col));
  for (int k = 0; k < ncls && k < 16; ++k)
    for (int c = 0; c < 3; ++c) col.bgr[k][c] = colors_host[k * 3 + c];
  col.paint = paint_mask;
  dim3 grid((unsigned)((h * w + 255) / 256), (unsigned)N);
  hipLaunchKernelGGL(overlay_u8_k, grid, dim3(256), 0, s, frames, hs, ws, probs, ncls, h, w, col, out, pred_out);
  RLN_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------
// Augmenting branch of the input transform (myTransforms.py:8-13): HueSaturationValue -> RandomSizedCrop (resized
// to height x width) -> OneOf(MotionBlur, GaussNoise), then Normalize + CHW.  The per-image random parameters are
// drawn on the host (the reference draws them with Python's `random` per image) and arrive as a table; the
// per-pixel Gaussian noise is generated here from a counter hash.  Same operations as albumentations 0.5.2 / cv2
// by their published definitions; parity with those libraries is unpinned (absent here), and a random stream can
// only match in distribution anyway.
// params[n][AUG_NP]: 0 hue, 1 sat, 2 val shift | 3 crop_y, 4 crop_x, 5 crop_h, 6 crop_w | 7 choice (0 blur, 1 noise)
//                    8 ksize | 9 sigma | 10 noise seed | 16.. 7x7 blur kernel (row-major, centred, zeros outside)
constexpr int AUG_NP = 80;

__device__ __forceinline__ void hsv_shift_u8(int& c0, int& c1, int& c2, float dh, float ds, float dv) {
#pragma clang fp contract(off)  // plain IEEE multiply/add sequence (checkable against the numpy restatement)
  // RGB -> HSV (H in [0,180), S, V in [0,255]), LUT shifts as albumentations' uint8 path, HSV -> RGB
  const int vmax = max(c0, max(c1, c2)), vmin = min(c0, min(c1, c2));
  const int d = vmax - vmin;
  float hf = 0.f;
  if (d != 0) {
    if (vmax == c0) hf = 30.f * (float)(c1 - c2) / (float)d;
    else if (vmax == c1) hf = 60.f + 30.f * (float)(c2 - c0) / (float)d;
    else hf = 120.f + 30.f * (float)(c0 - c1) / (float)d;
    if (hf < 0.f) hf += 180.f;
  }
  int H = (int)rintf(hf);
  if (H >= 180) H -= 180;
  const int S = vmax == 0 ? 0 : (int)rintf(255.f * (float)d / (float)vmax);
  const int V = vmax;
  float h2 = fmodf((float)H + dh, 180.f);
  if (h2 < 0.f) h2 += 180.f;
  const int H2 = (int)h2;                                           // astype(uint8): truncation
  const int S2 = (int)fminf(fmaxf((float)S + ds, 0.f), 255.f);
  const int V2 = (int)fminf(fmaxf((float)V + dv, 0.f), 255.f);
  const float s = (float)S2 * (1.f / 255.f), v = (float)V2 * (1.f / 255.f);
  const float h6 = (float)H2 * (1.f / 30.f);
  const int sector = (int)h6;
  const float f = h6 - (float)sector;
  const float pp = v * (1.f - s), qq = v * (1.f - s * f), tt = v * (1.f - s * (1.f - f));
  float r, g, b;
  switch (sector) {
    case 0: r = v; g = tt; b = pp; break;
    case 1: r = qq; g = v; b = pp; break;
    case 2: r = pp; g = v; b = tt; break;
    case 3: r = pp; g = qq; b = v; break;
    case 4: r = tt; g = pp; b = v; break;
    default: r = v; g = pp; b = qq; break;
  }
  c0 = min(max((int)rintf(r * 255.f), 0), 255);
  c1 = min(max((int)rintf(g * 255.f), 0), 255);
  c2 = min(max((int)rintf(b * 255.f), 0), 255);
}

// pass 1: HSV jitter on the four source pixels, crop + bilinear resize (cv2 8-bit fixed point) -> uint8 HWC scratch
__global__ __launch_bounds__(256) void aug_resize_k(const unsigned char* __restrict__ frames, int hs, int ws,
                                                    const unsigned char* __restrict__ labels, int h, int w,
                                                    const float* __restrict__ params, unsigned char* __restrict__ tmp,
                                                    long long* __restrict__ y) {
  const int n = blockIdx.y;
  const int px = blockIdx.x * 256 + threadIdx.x;
  if (px >= h * w) return;
  const float* pr = params + (long long)n * AUG_NP;
  const int cy = (int)pr[3], cx = (int)pr[4], ch = (int)pr[5], cw = (int)pr[6];
  const int dy = px / w, dx = px - dy * w;
  int x0, x1, a0, a1, y0, y1, b0, b1;
  lin_coef(dx, w, cw, &x0, &x1, &a0, &a1);
  lin_coef(dy, h, ch, &y0, &y1, &b0, &b1);
  const unsigned char* f = frames + (long long)n * hs * ws * 3;
  int q[4][3];
  const int ys[2] = {cy + y0, cy + y1}, xs[2] = {cx + x0, cx + x1};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const unsigned char* s = f + ((long long)ys[k >> 1] * ws + xs[k & 1]) * 3;
    q[k][0] = s[0];
    q[k][1] = s[1];
    q[k][2] = s[2];
    hsv_shift_u8(q[k][0], q[k][1], q[k][2], pr[0], pr[1], pr[2]);
  }
  unsigned char* o = tmp + ((long long)n * h * w + px) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int r0 = q[0][c] * a0 + q[1][c] * a1, r1 = q[2][c] * a0 + q[3][c] * a1;
    const int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
    o[c] = (unsigned char)min(max(v, 0), 255);
  }
  if (labels != nullptr && y != nullptr) {
    const int sy = min((int)floor((double)dy * ((double)ch / (double)h)), ch - 1);
    const int sx = min((int)floor((double)dx * ((double)cw / (double)w)), cw - 1);
    y[(long long)n * h * w + px] = (long long)labels[((long long)n * hs + cy + sy) * ws + cx + sx];
  }
}

__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

// pass 2: MotionBlur (filter2D, BORDER_REFLECT_101) or GaussNoise on the resized image, then Normalize + CHW
__global__ __launch_bounds__(256) void aug_post_k(const unsigned char* __restrict__ tmp, int h, int w,
                                                  const float* __restrict__ params, float m0, float m1, float m2,
                                                  float i0, float i1, float i2, float* __restrict__ x) {
  const int n = blockIdx.y;
  const int px = blockIdx.x * 256 + threadIdx.x;
  if (px >= h * w) return;
  const float* pr = params + (long long)n * AUG_NP;
  const int dy = px / w, dx = px - dy * w;
  const unsigned char* t = tmp + (long long)n * h * w * 3;
  int v[3];
  if (pr[7] < 0.5f) {  // motion blur
    float acc[3] = {0.f, 0.f, 0.f};
    for (int ky = 0; ky < 7; ++ky) {
      int sy = dy + ky - 3;
      sy = sy < 0 ? -sy : (sy >= h ? 2 * h - 2 - sy : sy);
      sy = min(max(sy, 0), h - 1);
      for (int kx = 0; kx < 7; ++kx) {
        const float wgt = pr[16 + ky * 7 + kx];
        if (wgt == 0.f) continue;
        int sx = dx + kx - 3;
        sx = sx < 0 ? -sx : (sx >= w ? 2 * w - 2 - sx : sx);
        sx = min(max(sx, 0), w - 1);
        const unsigned char* s = t + ((long long)sy * w + sx) * 3;
        acc[0] = fmaf(wgt, (float)s[0], acc[0]);
        acc[1] = fmaf(wgt, (float)s[1], acc[1]);
        acc[2] = fmaf(wgt, (float)s[2], acc[2]);
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = min(max((int)rintf(acc[c]), 0), 255);
  } else {  // Gaussian noise, independent per channel: clip(img + N(0, sigma)) -> uint8 (truncation)
    const float sigma = pr[9];
    const unsigned seed = (unsigned)pr[10];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const unsigned idx = (unsigned)((n * h * w + px) * 3 + c);
      const unsigned r1 = hash32(idx * 2u + 1u + seed * 0x9E3779B9u), r2 = hash32(idx * 2u + 2u + seed * 0x85EBCA6Bu);
      const float u1 = ((float)(r1 >> 8) + 1.f) * (1.f / 16777216.f), u2 = (float)(r2 >> 8) * (1.f / 16777216.f);
      const float g = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2) * sigma;
      const float o = (float)t[(long long)px * 3 + c] + g;
      v[c] = (int)fminf(fmaxf(o, 0.f), 255.f);
    }
  }
  const long long hw = (long long)h * w;
  float* xo = x + (long long)n * 3 * hw + px;
  xo[0] = ((float)v[0] - m0) * i0;
  xo[hw] = ((float)v[1] - m1) * i1;
  xo[2 * hw] = ((float)v[2] - m2) * i2;
}

int augment_u8(const unsigned char* frames, int N, int hs, int ws, const unsigned char* labels, int h, int w,
               const float* params, const float* mean3, const float* std3, unsigned char* tmp, float* x, long long* y,
               hipStream_t s) {
  const float m0 = mean3[0] * 255.f, m1 = mean3[1] * 255.f, m2 = mean3[2] * 255.f;
  const float i0 = 1.f / (std3[0] * 255.f), i1 = 1.f / (std3[1] * 255.f), i2 = 1.f / (std3[2] * 255.f);
  dim3 grid((unsigned)((h * w + 255) / 256), (unsigned)N);
  hipLaunchKernelGGL(aug_resize_k, grid, dim3(256), 0, s, frames, hs, ws, labels, h, w, params, tmp, y);
  hipLaunchKernelGGL(aug_post_k, grid, dim3(256), 0, s, tmp, h, w, params, m0, m1, m2, i0, i1, i2, x);
  RLN_LAUNCH_CHECK();
}

// =============================================================================================
// EncDecNet pieces
// =============================================================================================

__global__ __launch_bounds__(256) void bn_affine_from_sums_k(const float* __restrict__ sums, int C, double count,
                                                             int training, const float* gamma, const float* beta,
                                                             float* running_mean, float* running_var, float momentum,
                                                             float eps, float* a, float* b) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float g = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
  if (training) {
    const double m = (double)sums[2 * c] / count;
    double v = (double)sums[2 * c + 1] / count - m * m;
    if (v < 0.0) v = 0.0;
    const float is = 1.0f / sqrtf((float)v + eps);
    a[c] = g * is;
    b[c] = be - (float)m * g * is;
    if (running_mean != nullptr) {
      const double unb = count > 1.0 ? count / (count - 1.0) : 1.0;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(v * unb);
    }
  } else {
    const float is = 1.0f / sqrtf(running_var[c] + eps);
    a[c] = g * is;
    b[c] = be - running_mean[c] * g * is;
  }
}

int bn_affine_from_sums(const float* sums, int C, double count, int training, const float* gamma, const float* beta,
                        float* running_mean, float* running_var, float momentum, float eps, float* a, float* b,
                        hipStream_t s) {
  hipLaunchKernelGGL(bn_affine_from_sums_k, dim3((C + 255) / 256), dim3(256), 0, s, sums, C, count, training, gamma,
                     beta, running_mean, running_var, momentum, eps, a, b);
  RLN_LAUNCH_CHECK();
}

__global__ __launch_bounds__(256) void bn_drop_maxpool_k(const float* __restrict__ x, int C, int H, int W,
                                                         const float* __restrict__ a, const float* __restrict__ b,
                                                         const float* __restrict__ mask, int k, int Ho, int Wo,
                                                         float* out) {
  const int c = blockIdx.y, n = blockIdx.z;
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= Ho * Wo) return;
  const int oy = o / Wo, ox = o - oy * Wo;
  const float av = a ? a[c] : 1.f, bv = b ? b[c] : 0.f;
  const long long base = ((long long)n * C + c) * H * W;
  const int pad = k / 2;
  float best = -INFINITY;
  for (int dy = 0; dy < k; ++dy) {
    const int iy = 2 * oy - pad + dy;
    if (iy < 0 || iy >= H) continue;
    for (int dx = 0; dx < k; ++dx) {
      const int ix = 2 * ox - pad + dx;
      if (ix < 0 || ix >= W) continue;
      const long long idx = base + (long long)iy * W + ix;
      float v = fmaf(av, x[idx], bv);
      if (mask) v *= mask[idx];
      best = fmaxf(best, v);
    }
  }
  out[((long long)n * C + c) * Ho * Wo + o] = best;
}

int bn_drop_maxpool(const float* x, int N, int C, int H, int W, const float* a, const float* b, const float* mask,
                    int k, float* out, hipStream_t s) {
  const int Ho = (H + 2 * (k / 2) - k) / 2 + 1, Wo = (W + 2 * (k / 2) - k) / 2 + 1;
  hipLaunchKernelGGL(bn_drop_maxpool_k, dim3((Ho * Wo + 255) / 256, C, N), dim3(256), 0, s, x, C, H, W, a, b, mask, k,
                     Ho, Wo, out);
  RLN_LAUNCH_CHECK();
}

__global__ __launch_bounds__(256) void bn_drop_upsample2_k(const float* __restrict__ x, int C, int H, int W,
                                                           const float* __restrict__ a, const float* __restrict__ b,
                                                           const float* __restrict__ mask, float* out) {
  const int c = blockIdx.y, n = blockIdx.z;
  const int Ho = 2 * H, Wo = 2 * W;
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= Ho * Wo) return;
  const int oy = o / Wo, ox = o - oy * Wo;
  const float av = a ? a[c] : 1.f, bv = b ? b[c] : 0.f;
  const long long base = ((long long)n * C + c) * H * W;
  // align_corners=True: src = dst * (in - 1) / (out - 1)
  const float sy = Ho > 1 ? (float)oy * (float)(H - 1) / (float)(Ho - 1) : 0.f;
  const float sx = Wo > 1 ? (float)ox * (float)(W - 1) / (float)(Wo - 1) : 0.f;
  const int y0 = min((int)sy, H - 1), x0 = min((int)sx, W - 1);
  const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
  const float fy = sy - (float)y0, fx = sx - (float)x0;
  auto val = [&](int yy, int xx) {
    const long long idx = base + (long long)yy * W + xx;
    float v = fmaf(av, x[idx], bv);
    if (mask) v *= mask[idx];
    return v;
  };
  const float v00 = val(y0, x0), v01 = val(y0, x1), v10 = val(y1, x0), v11 = val(y1, x1);
  const float top = v00 + (v01 - v00) * fx, bot = v10 + (v11 - v10) * fx;
  out[((long long)n * C + c) * Ho * Wo + o] = top + (bot - top) * fy;
}

int bn_drop_upsample2(const float* x, int N, int C, int H, int W, const float* a, const float* b, const float* mask,
                      float* out, hipStream_t s) {
  hipLaunchKernelGGL(bn_drop_upsample2_k, dim3((4 * H * W + 255) / 256, C, N), dim3(256), 0, s, x, C, H, W, a, b, mask,
                     out);
  RLN_LAUNCH_CHECK();
}

// out = softmax over channels of (x / T) (use_softmax) or x / T; T = 1 is the plain Softmax(dim=-3) of EncDecNet.py:97
__global__ __launch_bounds__(256) void softmax_channels_k(const float* __restrict__ x, int C, long long HW, float T,
                                                          int use_softmax, float* out) {
  const int n = blockIdx.y;
  const long long px = (long long)blockIdx.x * 256 + threadIdx.x;
  if (px >= HW) return;
  const float* xp = x + (long long)n * C * HW + px;
  float* op = out + (long long)n * C * HW + px;
  if (!use_softmax) {
    for (int c = 0; c < C; ++c) op[(long long)c * HW] = xp[(long long)c * HW] / T;
    return;
  }
  float m = xp[0] / T;
  for (int c = 1; c < C; ++c) m = fmaxf(m, xp[(long long)c * HW] / T);
  float sum = 0.f;
  for (int c = 0; c < C; ++c) sum += expf(xp[(long long)c * HW] / T - m);
  const float inv = 1.f / sum;
  for (int c = 0; c < C; ++c) op[(long long)c * HW] = expf(xp[(long long)c * HW] / T - m) * inv;
}

int softmax_channels(const float* x, int N, int C, int HW, float* out, hipStream_t s, float T, int use_softmax) {
  hipLaunchKernelGGL(softmax_channels_k, dim3((HW + 255) / 256, N), dim3(256), 0, s, x, C, (long long)HW, T, use_softmax,
                     out);
  RLN_LAUNCH_CHECK();
}

// adentropy forward
template <int NC>
__global__ __launch_bounds__(256) void entropy_k(const float* __restrict__ probs, int ncls, long long HW, long long npix,
                                                 float* partial) {
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < LOSS_PIX_PER_BLOCK / 256; ++i) {
    const long long g = (long long)blockIdx.x * LOSS_PIX_PER_BLOCK + i * 256 + threadIdx.x;
    if (g < npix) {
      const long long n = g / HW, px = g - n * HW;
#pragma unroll
      for (int k = 0; k < NC; ++k)
        if (k < ncls) {
          const float pk = probs[(n * ncls + k) * HW + px];
          acc += pk * logf(pk + 1e-5f);
        }
    }
  }
  __shared__ float red[4];
  acc = wave_sum64(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[(long long)blockIdx.x * 4] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void entropy_finalize_k(const float* __restrict__ partial, long long nblk, double scale,
                                                          float* out) {
  double a = 0.0;
  for (long long b = threadIdx.x; b < nblk; b += 256) a += (double)partial[b * 4];
  __shared__ double red[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (float)((red[0] + red[1] + red[2] + red[3]) * scale);
}

int entropy_forward(const float* probs, int N, int ncls, int HW, float lamda, const LossScratch& sc, float* out,
                    hipStream_t s) {
  if (ncls > 16 || ncls < 1) return -4;
  const long long npix = (long long)N * HW;
  const long long nblk = loss_blocks(npix);
  if (ncls <= 4) {
    hipLaunchKernelGGL(entropy_k<4>, dim3((unsigned)nblk), dim3(256), 0, s, probs, ncls, (long long)HW, npix, sc.partial);
  } else if (ncls <= 8) {
    hipLaunchKernelGGL(entropy_k<8>, dim3((unsigned)nblk), dim3(256), 0, s, probs, ncls, (long long)HW, npix, sc.partial);
  } else {
    hipLaunchKernelGGL(entropy_k<16>, dim3((unsigned)nblk), dim3(256), 0, s, probs, ncls, (long long)HW, npix, sc.partial);
  }
  hipLaunchKernelGGL(entropy_finalize_k, dim3(1), dim3(256), 0, s, sc.partial, nblk, (double)lamda / (double)npix, out);
  RLN_LAUNCH_CHECK();
}

__global__ __launch_bounds__(256) void sgd_nesterov_k(float* __restrict__ p, const float* __restrict__ g,
                                                      float* __restrict__ buf, long long count, float lr, float mu,
                                                      float wd, int first_step, float grad_scale) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  const float pi = p[i];
  const float gi = g[i] * grad_scale + wd * pi;
  const float bi = first_step ? gi : fmaf(mu, buf[i], gi);
  buf[i] = bi;
  p[i] = pi - lr * (gi + mu * bi);
}

int sgd_nesterov(float* p, const float* g, float* buf, long long count, float lr, float momentum, float wd,
                 int first_step, float grad_scale, hipStream_t s) {
  hipLaunchKernelGGL(sgd_nesterov_k, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, p, g, buf, count, lr,
                     momentum, wd, first_step, grad_scale);
  RLN_LAUNCH_CHECK();
}

// =============================================================================================
// AdamW (torch.optim.AdamW single-tensor formulas, SimpleTrain.py:28), dropout masks, counters
// =============================================================================================

__global__ __launch_bounds__(256) void adamw_k(float* __restrict__ p, const float* __restrict__ g,
                                               float* __restrict__ m, float* __restrict__ v, long long count,
                                               float decay_mul, float b1, float b2, float eps, float step_size,
                                               float bc2_sqrt, float grad_scale) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  const float gi = g[i] * grad_scale;
  float pi = p[i] * decay_mul;
  float mi = m[i];
  mi = mi + (gi - mi) * (1.f - b1);          // exp_avg.lerp_(grad, 1 - beta1)
  const float vi = v[i] * b2 + (1.f - b2) * gi * gi;  // mul_(beta2).addcmul_(grad, grad, 1 - beta2)
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  pi = pi - step_size * (mi / denom);
  p[i] = pi;
  m[i] = mi;
  v[i] = vi;
}

int adamw(float* p, const float* g, float* m, float* v, long long count, float lr, float b1, float b2, float eps,
          float wd, int step, float grad_scale, hipStream_t s) {
  const double bc1 = 1.0 - pow((double)b1, (double)step);
  const double bc2 = 1.0 - pow((double)b2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  const float decay_mul = (float)(1.0 - (double)lr * (double)wd);
  hipLaunchKernelGGL(adamw_k, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, p, g, m, v, count, decay_mul, b1,
                     b2, eps, step_size, bc2_sqrt, grad_scale);
  RLN_LAUNCH_CHECK();
}

__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// counter-based Bernoulli(keep) per (dropout call, sample, channel): 0 or 1/keep
__global__ __launch_bounds__(256) void dropout_k(float* dst, long long count, float keep, unsigned long long seed) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  const unsigned long long r = mix64(mix64(seed) ^ (unsigned long long)i);
  const float u = (float)(r >> 40) * (1.0f / 16777216.0f);
  dst[i] = (u < keep) ? (1.0f / keep) : 0.f;
}

int dropout_scales(float* dst, long long count, float keep, unsigned long long seed, hipStream_t s) {
  hipLaunchKernelGGL(dropout_k, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, dst, count, keep, seed);
  RLN_LAUNCH_CHECK();
}

__global__ void add_one_k(long long* p, long long count) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < count) p[i] += 1;
}

int add_one_i64(long long* p, long long count, hipStream_t s) {
  hipLaunchKernelGGL(add_one_k, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, p, count);
  RLN_LAUNCH_CHECK();
}

}  // namespace rln
