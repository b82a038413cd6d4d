// Dense-block 3x3 kernels on the 16-bit MFMA pipe with split fp32 operands (see dense3.h).
#include "dense3.h"

#include <algorithm>
#include <cstdio>

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

// =============================================================================================
// weight packing
// =============================================================================================
template <int DT, int NP>
__global__ __launch_bounds__(256) void d3_pack_k(const float* __restrict__ params, const D3PackDesc* __restrict__ desc,
                                                 int n_desc, int total_units, uint4* __restrict__ packed) {
  const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (unit >= total_units) return;
  int d = 0;
  while (d + 1 < n_desc && desc[d + 1].unit_begin <= unit) ++d;
  const D3PackDesc q = desc[d];
  int u = unit - q.unit_begin;
  const int nch = (q.cin + 15) >> 4;
  const int per = nch * 5;
  const bool fwd_avail = q.wf_off >= 0;
  const bool backward = fwd_avail ? (u >= per) : true;
  if (backward) {
    if (fwd_avail) u -= per;
    if (q.wb_off < 0) return;
  }
  const int grp = u / 5, s = u - grp * 5;
  const int n = lane & 15, g = lane >> 4;
  const float* w = params + q.w_off;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = 8 * g + e;
    const int tap = 2 * s + (k >> 4);
    const int kk = k & 15;
    float val = 0.f;
    if (tap < 9) {
      if (!backward) {
        const int ch = grp * 16 + kk;
        if (n < q.cout && ch < q.cin) val = w[((long long)n * q.cin + ch) * 9 + tap];
      } else {
        const int c = grp * 16 + n;
        if (kk < q.cout && c < q.cin) val = w[((long long)kk * q.cin + c) * 9 + (8 - tap)];
      }
    }
    v[e] = val;
  }
  unsigned parts[4][NP];
#pragma unroll
  for (int i = 0; i < 4; ++i) split2<DT, NP>(v[2 * i], v[2 * i + 1], parts[i]);
  uint4* dst = packed + (backward ? q.wb_off : q.wf_off) + ((long long)(grp * 5 + s) * NP) * 64 + lane;
#pragma unroll
  for (int p = 0; p < NP; ++p) dst[p * 64] = make_uint4(parts[0][p], parts[1][p], parts[2][p], parts[3][p]);
}

int d3_pack_weights(const float* params, const D3PackDesc* desc_dev, int n_desc, int total_units, uint4* packed, int np,
                    int dt, hipStream_t s) {
  if (total_units <= 0) return 0;
  dim3 grid((unsigned)((total_units + 3) / 4));
#define D3_PACK(DT_, NP_)                                                                                         \
  hipLaunchKernelGGL((d3_pack_k<DT_, NP_>), grid, dim3(256), 0, s, params, desc_dev, n_desc, total_units, packed)
  if (dt == D3_BF16) {
    if (np == 1) D3_PACK(D3_BF16, 1);
    else if (np == 2) D3_PACK(D3_BF16, 2);
    else D3_PACK(D3_BF16, 3);
  } else {
    if (np == 1) D3_PACK(D3_F16, 1);
    else if (np == 2) D3_PACK(D3_F16, 2);
    else D3_PACK(D3_F16, 3);
  }
#undef D3_PACK
  return (int)hipGetLastError();
}

// =============================================================================================
// forward: out[n][j][p] = nscale[n][j] * (bias[j] + sum_{c,tap} relu(a[c]*S[n][c][p+tap] + b[c]) * W[j][c][tap])
//
// Block = one th x tw pixel tile of one sample, 4 waves; wave w owns M-tiles [w*MPW, (w+1)*MPW) (16 consecutive
// tile pixels each).  K loop over 16-channel chunks.  LDS image: [part][row 0..th+1][col 0..tw+1][16 channels]
// 16-bit, 32 bytes per pixel, odd pixel pitch (bank-conflict-free 16-byte writes from 4 rows x 2 channel octets per
// 8-lane group; A-fragment reads of 16 consecutive pixels are conflict-free for any start).  Out-of-image cells are
// zeroed once and never written (zero padding applies AFTER the activation).  Per chunk: the interior is fetched as
// 16-byte row segments (NR rounds of 8 channels x 4 pixels per thread), the two halo columns as scalars; the next
// chunk's global loads are issued before the MFMA phase of the current one and committed to LDS after it.
// =============================================================================================
template <int MPW, int NR, int NP, int DT>
__global__ __launch_bounds__(256, 2) void d3_fwd_k(const D3Fwd p) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lp = lane & 15, lg = lane >> 4;
  const int P = p.tw + 3;  // odd pixel pitch
  const int rows = p.th + 2;
  const int PLANE = rows * P * 32;
  const int nchunk_all = (p.Cin + 15) >> 4;
  const int Cpad = nchunk_all * 16;
  unsigned char* img = smem;
  float* abtab = reinterpret_cast<float*>(smem + NP * PLANE);
  float* red = abtab + 2 * Cpad;

  const int bx = blockIdx.x;
  const int tile_y = bx / p.tiles_x, tile_x = bx - tile_y * p.tiles_x;
  const int gy0 = tile_y * p.th, gx0 = tile_x * p.tw;
  const int n = blockIdx.z;
  const float* Sn = p.S + (long long)n * p.ns;

  // chunk range of this block (split-K over blockIdx.y)
  const int per = (nchunk_all + (int)gridDim.y - 1) / (int)gridDim.y;
  const int c_begin = (int)blockIdx.y * per;
  const int c_end = min(nchunk_all, c_begin + per);

  // ---- one-time LDS setup: zero image, BN affine table ----
  {
    uint4* z = reinterpret_cast<uint4*>(img);
    const int n16 = NP * PLANE / 16;
    for (int i = tid; i < n16; i += 256) z[i] = make_uint4(0u, 0u, 0u, 0u);
    for (int i = tid; i < Cpad; i += 256) {
      abtab[i] = i < p.Cin ? p.pa[i] : 0.f;
      abtab[Cpad + i] = i < p.Cin ? p.pb[i] : 0.f;
    }
  }

  // ---- staging plan (chunk-invariant) ----
  const int nq = p.tw >> 2;
  const int nrq = (rows + 3) >> 2;
  int s_goff[NR], s_lds[NR];
  bool s_ok[NR];
  int s_o[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int lu = tid + 256 * i;
    const int o = lu & 1, rr = (lu >> 1) & 3, u = lu >> 3;
    const int R = u / nq, Q = u - R * nq;
    const int r = 4 * R + rr;
    const int iy = gy0 - 1 + r, ix = gx0 + 4 * Q;
    const bool ok = (R < nrq) && (r < rows) && (iy >= 0) && (iy < p.H) && (ix < p.W);
    s_ok[i] = ok;
    s_goff[i] = ok ? iy * p.W + ix : 0;
    s_lds[i] = ((ok ? r : 0) * P + 1 + 4 * Q) * 32 + o * 16;
    s_o[i] = o;
  }
  // halo columns: unit = (row, side, channel pair)
  const int h_cp = tid & 7, h_side = (tid >> 3) & 1, h_r = tid >> 4;
  const int h_iy = gy0 - 1 + h_r, h_ix = h_side ? gx0 + p.tw : gx0 - 1;
  const bool h_ok = (h_r < rows) && (h_iy >= 0) && (h_iy < p.H) && (h_ix >= 0) && (h_ix < p.W);
  const int h_goff = h_ok ? h_iy * p.W + h_ix : 0;
  const int h_lds = ((h_ok ? h_r : 0) * P + (h_side ? p.tw + 1 : 0)) * 32 + h_cp * 4;

  // ---- MFMA-phase geometry ----
  const int npix = p.th * p.tw;
  int basem[MPW];
  int m_cnt = 0;  // number of M-tiles of this wave that hold pixels (wave-uniform)
#pragma unroll
  for (int m = 0; m < MPW; ++m) {
    const int mt = wave * MPW + m;
    if (mt * 16 < npix) m_cnt = m + 1;
    const int q = min(mt * 16 + lp, npix - 1);
    const int ty = q / p.tw, tx = q - ty * p.tw;
    basem[m] = ((ty + 1) * P + tx + 1) * 32 + (lg & 1) * 16;
  }
  int toff[5];
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    const int tap = min(2 * s + (lg >> 1), 8);
    toff[s] = ((tap / 3 - 1) * P + (tap % 3 - 1)) * 32;
  }

  float4 sreg[NR][8];
  float hreg[2];
  auto issue = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int cb = chunk * 16 + s_o[i] * 8;
#pragma unroll
      for (int cc = 0; cc < 8; ++cc) {
        const int ch = min(cb + cc, p.Cin - 1);
        sreg[i][cc] = *reinterpret_cast<const float4*>(Sn + (long long)ch * p.cs + s_goff[i]);
      }
    }
    {
      const int c0 = min(chunk * 16 + 2 * h_cp, p.Cin - 1), c1 = min(chunk * 16 + 2 * h_cp + 1, p.Cin - 1);
      hreg[0] = Sn[(long long)c0 * p.cs + h_goff];
      hreg[1] = Sn[(long long)c1 * p.cs + h_goff];
    }
  };
  auto commit = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const float* ab = abtab + chunk * 16 + s_o[i] * 8;
      const float4 a0 = *reinterpret_cast<const float4*>(ab), a1 = *reinterpret_cast<const float4*>(ab + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(ab + Cpad), b1 = *reinterpret_cast<const float4*>(ab + Cpad + 4);
      const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
      const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      float z[4][8];
#pragma unroll
      for (int cc = 0; cc < 8; ++cc) {
        z[0][cc] = fmaxf(fmaf(av[cc], sreg[i][cc].x, bv[cc]), 0.f);
        z[1][cc] = fmaxf(fmaf(av[cc], sreg[i][cc].y, bv[cc]), 0.f);
        z[2][cc] = fmaxf(fmaf(av[cc], sreg[i][cc].z, bv[cc]), 0.f);
        z[3][cc] = fmaxf(fmaf(av[cc], sreg[i][cc].w, bv[cc]), 0.f);
      }
      if (s_ok[i]) {
#pragma unroll
        for (int px = 0; px < 4; ++px) {
          unsigned parts[4][NP];
#pragma unroll
          for (int k = 0; k < 4; ++k) split2<DT, NP>(z[px][2 * k], z[px][2 * k + 1], parts[k]);
#pragma unroll
          for (int pt = 0; pt < NP; ++pt)
            *reinterpret_cast<uint4*>(img + pt * PLANE + s_lds[i] + px * 32) =
                make_uint4(parts[0][pt], parts[1][pt], parts[2][pt], parts[3][pt]);
        }
      }
    }
    {
      const int c0 = chunk * 16 + 2 * h_cp;
      const float z0 = fmaxf(fmaf(abtab[c0], hreg[0], abtab[Cpad + c0]), 0.f);
      const float z1 = fmaxf(fmaf(abtab[c0 + 1], hreg[1], abtab[Cpad + c0 + 1]), 0.f);
      if (h_ok) {
        unsigned parts[NP];
        split2<DT, NP>(z0, z1, parts);
#pragma unroll
        for (int pt = 0; pt < NP; ++pt) *reinterpret_cast<unsigned*>(img + pt * PLANE + h_lds) = parts[pt];
      }
    }
  };

  f32x4 acc[MPW];
#pragma unroll
  for (int m = 0; m < MPW; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (c_begin < c_end) {
    issue(c_begin);
    __syncthreads();  // zeroed image + affine table visible
    commit(c_begin);
    __syncthreads();
    uint4 bf[5][NP];
    auto load_b = [&](int chunk) {
      const uint4* wp = p.wpk + ((long long)chunk * 5 * NP) * 64 + lane;
#pragma unroll
      for (int s = 0; s < 5; ++s)
#pragma unroll
        for (int pt = 0; pt < NP; ++pt) bf[s][pt] = wp[(s * NP + pt) * 64];
    };
    load_b(c_begin);
    for (int chunk = c_begin; chunk < c_end; ++chunk) {
      if (chunk + 1 < c_end) issue(chunk + 1);
#pragma unroll
      for (int s = 0; s < 5; ++s) {
#pragma unroll
        for (int m = 0; m < MPW; ++m) {
          if (m < m_cnt) {
            const int ad = basem[m] + toff[s];
            uint4 af[NP];
#pragma unroll
            for (int pt = 0; pt < NP; ++pt) af[pt] = *reinterpret_cast<const uint4*>(img + pt * PLANE + ad);
            acc[m] = mfma_split<DT, NP>(af, bf[s], acc[m]);
          }
        }
      }
      if (chunk + 1 < c_end) load_b(chunk + 1);  // consumed after the commit phase
      __syncthreads();  // every wave is done reading this chunk's image
      if (chunk + 1 < c_end) commit(chunk + 1);
      __syncthreads();
    }
  } else {
    __syncthreads();
  }

  // ---- epilogue: lane holds 4 consecutive pixels (rows 4*lg..4*lg+3 of the M-tile) of output channel lp ----
  const int j = lp;
  const bool jv = j < p.Cout;
  const bool raw = p.ksplit > 1;
  const float bias = (jv && !raw && p.bias) ? p.bias[j] : 0.f;
  const float sc = (jv && !raw && p.nscale) ? p.nscale[(long long)n * p.Cout + j] : 1.f;
  float* outn = p.out + (raw ? (long long)blockIdx.y * p.split_stride : 0) + (long long)n * p.out_ns +
                (long long)(jv ? j : 0) * p.out_cs;
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int m = 0; m < MPW; ++m) {
    const int q = (wave * MPW + m) * 16 + lg * 4;
    const int ty = q / p.tw, tx = q - ty * p.tw;
    const int gy = gy0 + ty, gx = gx0 + tx;
    const bool ok = jv && (q < npix) && (gy < p.H) && (gx < p.W);
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[r] = (acc[m][r] + bias) * sc;
      if (ok) {
        s1 += v[r];
        s2 += v[r] * v[r];
      }
    }
    if (ok) *reinterpret_cast<float4*>(outn + (long long)gy * p.W + gx) = make_float4(v[0], v[1], v[2], v[3]);
  }
  if (p.stat_partial != nullptr && !raw) {
    s1 = group4_sum(s1);
    s2 = group4_sum(s2);
    if (lg == 0) {
      red[(wave * 16 + lp) * 2 + 0] = s1;
      red[(wave * 16 + lp) * 2 + 1] = s2;
    }
    __syncthreads();
    if (tid < 2 * p.Cout) {
      const int jj = tid >> 1, w2 = tid & 1;
      const float t = red[(0 * 16 + jj) * 2 + w2] + red[(1 * 16 + jj) * 2 + w2] + red[(2 * 16 + jj) * 2 + w2] +
                      red[(3 * 16 + jj) * 2 + w2];
      const long long brow = (long long)n * gridDim.x + blockIdx.x;
      p.stat_partial[(brow * p.Cout + jj) * 2 + w2] = t;
    }
  }
}

bool d3_fwd_supported(const D3Fwd& p) {
  if (p.Cout > 16 || p.Cout < 1 || p.Cin < 1) return false;
  if ((p.W & 3) || p.W < 40 || p.H < 4) return false;
  if ((p.cs & 3) || (p.ns & 3) || (p.out_cs & 3) || (p.out_ns & 3)) return false;
  if ((reinterpret_cast<uintptr_t>(p.S) & 15) || (reinterpret_cast<uintptr_t>(p.out) & 15)) return false;
  return true;
}

void d3_fwd_pick_tile(int H, int W, int* th, int* tw) {
  (void)H;
  *th = 8;
  *tw = (W % 80 == 0 || W > 120) ? 80 : 40;
}

template <int MPW, int NR, int NP, int DT>
static int d3_fwd_launch_t(const D3Fwd& p, int N, hipStream_t s) {
  const int P = p.tw + 3, rows = p.th + 2;
  const int Cpad = ((p.Cin + 15) / 16) * 16;
  const size_t lds = (size_t)NP * rows * P * 32 + (size_t)2 * Cpad * 4 + 4 * 16 * 2 * 4;
  auto kern = d3_fwd_k<MPW, NR, NP, DT>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    (void)hipGetLastError();
    attr_done = true;
  }
  if (lds > 160 * 1024) return -4;
  dim3 grid((unsigned)(p.tiles_x * p.tiles_y), (unsigned)std::max(1, p.ksplit), (unsigned)N);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, p);
  return (int)hipGetLastError();
}

int d3_fwd_launch(const D3Fwd& p, int N, int np, int dt, hipStream_t s) {
  if (!d3_fwd_supported(p)) return -4;
  // tile capacity: 64*MPW pixels; staging rounds NR = ceil(8 * ceil((th+2)/4) * (tw/4) / 256)
  const int npix = p.th * p.tw;
  const int lane_units = 8 * ((p.th + 2 + 3) / 4) * (p.tw / 4);
  const int nr = (lane_units + 255) / 256;
  if ((p.th + 2) * 16 > 256 || nr > 2 || (p.tw & 3)) return -4;
#define D3_FWD(MPW_, NR_)                                                                         \
  do {                                                                                            \
    if (dt == D3_BF16) {                                                                          \
      if (np == 1) return d3_fwd_launch_t<MPW_, NR_, 1, D3_BF16>(p, N, s);                        \
      if (np == 2) return d3_fwd_launch_t<MPW_, NR_, 2, D3_BF16>(p, N, s);                        \
      return d3_fwd_launch_t<MPW_, NR_, 3, D3_BF16>(p, N, s);                                     \
    } else {                                                                                      \
      if (np == 1) return d3_fwd_launch_t<MPW_, NR_, 1, D3_F16>(p, N, s);                         \
      return d3_fwd_launch_t<MPW_, NR_, 2, D3_F16>(p, N, s);                                      \
    }                                                                                             \
  } while (0)
  if (npix <= 320 && nr == 1) D3_FWD(5, 1);
  if (npix <= 320) D3_FWD(5, 2);
  if (npix <= 640) D3_FWD(10, 2);
#undef D3_FWD
  return -4;
}

}  // namespace rln
