// First-convolution kernels on the 16-bit MFMA pipe with split fp32 operands (see fc3.h).
#include "fc3.h"

#include <algorithm>

#include "split16.h"
#include "storage.h"

namespace rln {

constexpr int F3_MT = 4;  // M tiles (16 output channels each) at most

// =============================================================================================
// forward: out[o][p] = bias[o] + sum_{k = c*9 + tap < Cin*9} W[o][k] * x[c][p + tap]
//
// Block = 4 persistent waves; the weight fragments (M tiles x parts, one K step) are built in LDS by the block itself.
// A wave walks super tiles of 64 consecutive pixels: lane (n = l&15, kb = l>>4) owns pixels 4n..4n+3 (the four N tiles,
// so its four accumulators of a channel are one 16-byte store) and the K entries 8kb..8kb+7, i.e. 8 (channel, tap)
// pairs whose shifted pixels it loads directly (zero outside the image).
// =============================================================================================
template <int NP, int DT, int OT>
__global__ __launch_bounds__(256, 2) void f3_fwd_k(const F3Fwd p) {
  __shared__ __align__(16) uint4 wl[F3_MT * 3 * 64];  // [mtile][part][lane] (NP <= 3)
  __shared__ float btab[F3_MT * 16];
  __shared__ float slot[4 * F3_MT * 16 * 2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n16 = lane & 15, kb = lane >> 4;
  const int K = p.Cin * 9;
  const int MT = (p.Cout + 15) >> 4;
  for (int e = tid; e < MT * 64; e += 256) {
    const int m = e >> 6, l = e & 63;
    const int o = m * 16 + (l & 15);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 8 * (l >> 4) + j;
      v[j] = (o < p.Cout && k < K) ? sat16<DT>(p.w[(long long)o * K + k] * w_prescale<DT>()) : 0.f;  // split16.h
    }
    unsigned parts[4][NP];
#pragma unroll
    for (int j = 0; j < 4; ++j) split2<DT, NP>(v[2 * j], v[2 * j + 1], parts[j]);
#pragma unroll
    for (int pt = 0; pt < NP; ++pt) wl[(m * 3 + pt) * 64 + l] = make_uint4(parts[0][pt], parts[1][pt], parts[2][pt], parts[3][pt]);
  }
  for (int e = tid; e < F3_MT * 16; e += 256) btab[e] = (p.bias && e < p.Cout) ? p.bias[e] : 0.f;
  for (int e = tid; e < 4 * F3_MT * 32; e += 256) slot[e] = 0.f;
  __syncthreads();

  // the lane's 8 (channel, tap) pairs
  int koff[8], kdy[8], kdx[8];
  bool kval[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = 8 * kb + j;
    const int c = k / 9, t = k - c * 9;
    kdy[j] = t / 3 - 1;
    kdx[j] = t % 3 - 1;
    kval[j] = k < K;
    koff[j] = kval[j] ? c * p.H * p.W + kdy[j] * p.W + kdx[j] : 0;
  }
  const int HW = p.H * p.W;
  const long long total = (long long)p.N * HW;
  const long long nsuper = (total + 63) >> 6;
  const long long sstride = (long long)gridDim.x * 4;
  for (long long ST = (long long)blockIdx.x * 4 + wave; ST < nsuper; ST += sstride) {
    const long long pix = ST * 64 + 4 * n16;  // W % 4 == 0: the lane's 4 pixels share a row
    const bool pv = pix < total;
    const long long pc = pv ? pix : 0;
    const int ns_ = (int)(pc / HW);
    const int rem = (int)(pc - (long long)ns_ * HW);
    const int y = rem / p.W, x0 = rem - y * p.W;
    const float* xb = p.X + (long long)ns_ * p.Cin * HW + rem;
    float v[4][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool rv = pv && kval[j] && (unsigned)(y + kdy[j]) < (unsigned)p.H;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const bool ok = rv && (unsigned)(x0 + t + kdx[j]) < (unsigned)p.W;
        v[t][j] = ok ? sat16<DT>(xb[koff[j] + t]) : 0.f;  // raw input: saturate into the part type's range
      }
    }
    uint4 bf[4][NP];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      unsigned w[4][NP];
#pragma unroll
      for (int j = 0; j < 4; ++j) split2<DT, NP>(v[t][2 * j], v[t][2 * j + 1], w[j]);
#pragma unroll
      for (int pt = 0; pt < NP; ++pt) bf[t][pt] = make_uint4(w[0][pt], w[1][pt], w[2][pt], w[3][pt]);
    }
    int kb4 = 4 * kb;
    asm volatile("" : "+v"(kb4));
#pragma unroll
    for (int m = 0; m < F3_MT; ++m) {
      if (m < MT) {
        uint4 A[NP];
#pragma unroll
        for (int pt = 0; pt < NP; ++pt) A[pt] = wl[(m * 3 + pt) * 64 + lane];
        f32x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = mfma_split<DT, NP>(A, bf[t], f32x4{0.f, 0.f, 0.f, 0.f});
        mfma_drain();  // (the accumulators are read right away: see split16.h)
        const float4 b4 = *reinterpret_cast<const float4*>(btab + m * 16 + kb4);
        const float bia[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = m * 16 + kb4 + r;
          const bool st = pv && o < p.Cout;
          // (rounded to the storage type first: the statistics describe the tensor as it is stored)
          const float4 ov = make_float4(st_round<OT>(fmaf(acc[0][r], w_unscale<DT>(), bia[r])),
                                        st_round<OT>(fmaf(acc[1][r], w_unscale<DT>(), bia[r])),
                                        st_round<OT>(fmaf(acc[2][r], w_unscale<DT>(), bia[r])),
                                        st_round<OT>(fmaf(acc[3][r], w_unscale<DT>(), bia[r])));
          if (st) SP<OT>(p.out).st4((long long)ns_ * p.out_ns + (long long)o * p.out_cs + rem, ov.x, ov.y, ov.z, ov.w);
          float s1 = st ? (ov.x + ov.y) + (ov.z + ov.w) : 0.f;
          float s2 = st ? (ov.x * ov.x + ov.y * ov.y) + (ov.z * ov.z + ov.w * ov.w) : 0.f;
          s1 = row16_sum(s1);
          s2 = row16_sum(s2);
          if (n16 == 0) {
            lds_add_f32(slot + ((wave * F3_MT * 16) + m * 16 + kb4 + r) * 2, s1);
            lds_add_f32(slot + ((wave * F3_MT * 16) + m * 16 + kb4 + r) * 2 + 1, s2);
          }
        }
      }
    }
  }
  __syncthreads();
  if (p.stat_partial != nullptr && tid < p.Cout) {
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      a1 += slot[((w * F3_MT * 16) + tid) * 2 + 0];
      a2 += slot[((w * F3_MT * 16) + tid) * 2 + 1];
    }
    p.stat_partial[((long long)blockIdx.x * p.Cout + tid) * 2 + 0] = a1;
    p.stat_partial[((long long)blockIdx.x * p.Cout + tid) * 2 + 1] = a2;
  }
}

bool f3_fwd_supported(const F3Fwd& p) {
  if (p.Cin < 1 || p.Cin * 9 > 32 || p.Cout < 1 || p.Cout > F3_MT * 16 || p.N < 1) return false;
  if (p.H < 1 || p.W < 4 || (p.W & 3) || (p.out_cs & 3) || (p.out_ns & 3)) return false;
  if (reinterpret_cast<uintptr_t>(p.out) & (p.ot == ST_BF16 ? 7 : 15)) return false;
  return true;
}

void f3_fwd_plan(F3Fwd* p) {
  const long long nsuper = ((long long)p->N * p->H * p->W + 63) / 64;
  p->blocks = (int)std::max(1ll, std::min((nsuper + 3) / 4, 512ll));
}

int f3_fwd_launch(const F3Fwd& p, int np, int dt, hipStream_t s) {
  if (!f3_fwd_supported(p) || p.blocks < 1) return -4;
  const dim3 grid((unsigned)p.blocks);
#define F3_FWD(NP_, DT_) hipLaunchKernelGGL((f3_fwd_k<NP_, DT_, ST_F32>), grid, dim3(256), 0, s, p)
  if (p.ot == ST_BF16) {  // bf16 storage = plain bf16 operands
    if (np != 1 || dt != D3_BF16) return -4;
    hipLaunchKernelGGL((f3_fwd_k<1, D3_BF16, ST_BF16>), grid, dim3(256), 0, s, p);
  } else if (dt == D3_BF16) {
    if (np == 1) F3_FWD(1, D3_BF16);
    else if (np == 2) F3_FWD(2, D3_BF16);
    else if (np == 3) F3_FWD(3, D3_BF16);
    else return -4;
  } else if (dt == D3_F16) {
    if (np == 1) F3_FWD(1, D3_F16);
    else if (np == 2) F3_FWD(2, D3_F16);
    else return -4;
  } else {
    return -4;
  }
#undef F3_FWD
  return (int)hipGetLastError();
}

// =============================================================================================
// weight gradient: dW[o][k = c*9 + tap] = sum_{n,p} dY[n][o][p] * x[n][c][p + tap]
//
// K = pixels: a K step is 32 consecutive pixels (lane kb owns 8 of them; W % 8 == 0 keeps a lane's run inside one row).
// A = dY rows (two 16-byte loads per M tile), B = the shifted input run of the lane's (channel, tap) column (8 scalar
// loads, zero outside the image).  Waves take K steps round-robin; the block's 4 waves are summed through LDS and one
// partial row per block is written (reduced afterwards in fixed order).
// =============================================================================================
template <int NP, int DT, int YT>
__global__ __launch_bounds__(256, 2) void f3_wgrad_k(const F3Wgrad p) {
  __shared__ float red[4 * F3_MT * 2 * 256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n16 = lane & 15, kb = lane >> 4;
  const int K = p.Cin * 9;
  const int MT = (p.Cout + 15) >> 4;
  const int HW = p.H * p.W;
  const long long total = (long long)p.N * HW;
  const long long ksteps = (total + 31) >> 5;
  // the lane's two (channel, tap) columns
  int coff[2], cdy[2], cdx[2];
  bool cval[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int k = 16 * i + n16;
    const int c = k / 9, t = k - c * 9;
    cdy[i] = t / 3 - 1;
    cdx[i] = t % 3 - 1;
    cval[i] = k < K;
    coff[i] = cval[i] ? c * HW + cdy[i] * p.W + cdx[i] : 0;
  }
  f32x4 acc[F3_MT][2];
#pragma unroll
  for (int m = 0; m < F3_MT; ++m) acc[m][0] = acc[m][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  const long long kstride = (long long)gridDim.x * 4;
  for (long long ks = (long long)blockIdx.x * 4 + wave; ks < ksteps; ks += kstride) {
    const long long pix = ks * 32 + 8 * kb;
    const bool pv = pix < total;
    const long long pc = pv ? pix : 0;
    const int ns_ = (int)(pc / HW);
    const int rem = (int)(pc - (long long)ns_ * HW);
    const int y = rem / p.W, x0 = rem - y * p.W;
    // B fragments
    uint4 bf[2][NP];
    const float* xb = p.X + (long long)ns_ * p.Cin * HW + rem;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool rv = pv && cval[i] && (unsigned)(y + cdy[i]) < (unsigned)p.H;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const bool ok = rv && (unsigned)(x0 + e + cdx[i]) < (unsigned)p.W;
        v[e] = ok ? xb[coff[i] + e] : 0.f;
      }
      unsigned w[4][NP];
#pragma unroll
      for (int j = 0; j < 4; ++j) split2<DT, NP>(v[2 * j], v[2 * j + 1], w[j]);
#pragma unroll
      for (int pt = 0; pt < NP; ++pt) bf[i][pt] = make_uint4(w[0][pt], w[1][pt], w[2][pt], w[3][pt]);
    }
    // A fragments and products
    const SP<YT> yb = SP<YT>(p.dY) + ((long long)ns_ * p.Cout * HW + rem);
#pragma unroll
    for (int m = 0; m < F3_MT; ++m) {
      if (m < MT) {
        const int o = min(m * 16 + n16, p.Cout - 1);
        const bool ov = pv && (m * 16 + n16 < p.Cout);
        const float4 u0 = yb.ld4((long long)o * HW);
        const float4 u1 = yb.ld4((long long)o * HW + 4);
        unsigned w[4][NP];
        split2<DT, NP>(ov ? u0.x : 0.f, ov ? u0.y : 0.f, w[0]);
        split2<DT, NP>(ov ? u0.z : 0.f, ov ? u0.w : 0.f, w[1]);
        split2<DT, NP>(ov ? u1.x : 0.f, ov ? u1.y : 0.f, w[2]);
        split2<DT, NP>(ov ? u1.z : 0.f, ov ? u1.w : 0.f, w[3]);
        uint4 A[NP];
#pragma unroll
        for (int pt = 0; pt < NP; ++pt) A[pt] = make_uint4(w[0][pt], w[1][pt], w[2][pt], w[3][pt]);
        acc[m][0] = mfma_split<DT, NP>(A, bf[0], acc[m][0]);
        acc[m][1] = mfma_split<DT, NP>(A, bf[1], acc[m][1]);
      }
    }
  }
mfma_drain();  // wait states between the MFMA chain and the first accumulator read (split16.h)
  // ---- block reduction (fixed order) and store: acc[m][i][r] = dW[o = 16m + 4kb + r][k = 16i + n16] ----
#pragma unroll
  for (int m = 0; m < F3_MT; ++m)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[((wave * F3_MT + m) * 2 + i) * 256 + r * 64 + lane] = acc[m][i][r];
  __syncthreads();
  for (int e = tid; e < F3_MT * 2 * 256; e += 256) {
    const int mi = e >> 8, q = e & 255;
    const int m = mi >> 1, i = mi & 1, r = q >> 6, l = q & 63;
    const float v = (red[((0 * F3_MT + m) * 2 + i) * 256 + q] + red[((1 * F3_MT + m) * 2 + i) * 256 + q]) +
                    (red[((2 * F3_MT + m) * 2 + i) * 256 + q] + red[((3 * F3_MT + m) * 2 + i) * 256 + q]);
    const int o = m * 16 + 4 * (l >> 4) + r, k = 16 * i + (l & 15);
    if (o < p.Cout && k < K) p.partial[((long long)blockIdx.x * p.Cout + o) * K + k] = v;
  }
}

bool f3_wgrad_supported(const F3Wgrad& p) {
  if (p.Cin < 1 || p.Cin * 9 > 32 || p.Cout < 1 || p.Cout > F3_MT * 16 || p.N < 1) return false;
  if (p.H < 1 || p.W < 8 || (p.W & 7)) return false;
  if (reinterpret_cast<uintptr_t>(p.dY) & (p.yt == ST_BF16 ? 7 : 15)) return false;
  return true;
}

void f3_wgrad_plan(F3Wgrad* p) {
  const long long ksteps = ((long long)p->N * p->H * p->W + 31) / 32;
  p->blocks = (int)std::max(1ll, std::min((ksteps + 3) / 4, 512ll));
}

int f3_wgrad_launch(const F3Wgrad& p, int np, int dt, hipStream_t s) {
  if (!f3_wgrad_supported(p) || p.blocks < 1) return -4;
  const dim3 grid((unsigned)p.blocks);
#define F3_WG(NP_, DT_) hipLaunchKernelGGL((f3_wgrad_k<NP_, DT_, ST_F32>), grid, dim3(256), 0, s, p)
  if (p.yt == ST_BF16) {  // bf16 storage = plain bf16 operands
    if (np != 1 || dt != D3_BF16) return -4;
    hipLaunchKernelGGL((f3_wgrad_k<1, D3_BF16, ST_BF16>), grid, dim3(256), 0, s, p);
  } else if (dt == D3_BF16) {
    if (np == 1) F3_WG(1, D3_BF16);
    else if (np == 2) F3_WG(2, D3_BF16);
    else if (np == 3) F3_WG(3, D3_BF16);
    else return -4;
  } else if (dt == D3_F16) {
    if (np == 1) F3_WG(1, D3_F16);
    else if (np == 2) F3_WG(2, D3_F16);
    else return -4;
  } else {
    return -4;
  }
#undef F3_WG
  return (int)hipGetLastError();
}

}  // namespace rln
