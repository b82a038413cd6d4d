// Dense-block 3x3 kernels on the 16-bit MFMA pipe with split fp32 operands (see dense3.h).
#include "dense3.h"

#include <algorithm>
#include <cstdio>
#include <type_traits>

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
        // the last chunk of a Cin that is no multiple of 16 covers channels [Cin-16, Cin): the kernel then never reads
        // past the layer's input range; channels an earlier chunk already covered get zero weights
        const int cb = min(grp * 16, max(q.cin - 16, 0));
        const int ch = cb + kk;
        if (n < q.cout && ch < q.cin && ch >= grp * 16) val = w[((long long)n * q.cin + ch) * 9 + tap];
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
// Block = one th x tw pixel tile of one sample, 12 waves with fixed roles:
//   waves 0-3  "consumers": wave w owns M-tiles [w*MPW, (w+1)*MPW) (16 consecutive tile pixels each) and only issues
//               LDS fragment reads + MFMAs (150 per 16-channel chunk at 2 parts);
//   waves 4-7 / 8-11 two "producer" groups taking alternate chunks: fetch a chunk's fp32 rows from HBM, apply BN+ReLU,
//               split into 16-bit parts and write the [pixel][channel] image of chunk k+1 into the OTHER LDS buffer
//               while chunk k multiplies.  A group re-issues its loads (chunk k+3) right after committing chunk k+1,
//               so every load has a whole iteration in flight and ~2 chunks (128 KB per CU) are outstanding.
// The conversion costs about as many VALU cycles per chunk as the MFMAs cost matrix cycles; with a consumer wave and
// two producer waves per SIMD the matrix pipe, the VALU and the memory system run side by side (one barrier per
// chunk) instead of one after the other.
// LDS image: [buffer][part][row 0..th+1][col 0..tw+1][16 channels] 16-bit, 32 bytes per pixel, odd pixel pitch
// (bank-conflict-free 16-byte writes from 4 rows x 2 channel octets per 8-lane group; A-fragment reads of 16
// consecutive pixels are conflict-free for any start).  Out-of-image cells are zeroed once and never written (zero
// padding applies AFTER the activation).  Interior rows are fetched as 16-byte segments (NR rounds of 8 channels x 4
// pixels per producer thread), the two halo columns as scalars.
// =============================================================================================
template <int MPW, int NR, int NP, int DT>
__global__ __launch_bounds__(768, 3) void d3_fwd_k(const D3Fwd p) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool producer = wave >= 4;
  const int P = p.tw + 3;  // odd pixel pitch
  const int rows = p.th + 2;
  const int PLANE = rows * P * 32;
  const int IMG = NP * PLANE + 5 * NP * 1024;  // one buffer: image parts + weight fragments of the chunk
  const int nchunk_all = (p.Cin + 15) >> 4;
  const int Cpad = nchunk_all * 16;
  float* abtab = reinterpret_cast<float*>(smem + 2 * IMG);
  float* red = abtab + 2 * Cpad;

  // XCD-aware tile order: blocks are dealt round-robin over the 8 XCDs, so vertically adjacent tiles (which share two
  // halo rows per channel) would never meet in one L2.  Each XCD walks a contiguous range of (sample, tile) pairs
  // instead: the ~32 tiles an XCD runs at a time are neighbours and the halo rows are served by its L2 rather than by
  // HBM a second time.  Placement is a speed matter only.
  int bx = blockIdx.x, n = blockIdx.z;
  if (gridDim.y == 1) {
    const unsigned lin = blockIdx.x + gridDim.x * blockIdx.z;
    const unsigned lg = xcd_logical_block(lin, gridDim.x * gridDim.z);
    n = (int)(lg / gridDim.x);
    bx = (int)(lg - (unsigned)n * gridDim.x);
  }
  const int tile_y = bx / p.tiles_x, tile_x = bx - tile_y * p.tiles_x;
  const int gy0 = tile_y * p.th, gx0 = tile_x * p.tw;
  const float* Sn = p.S + (long long)n * p.ns;

  // chunk range of this block (split-K over blockIdx.y)
  const int per = (nchunk_all + (int)gridDim.y - 1) / (int)gridDim.y;
  const int c_begin = (int)blockIdx.y * per;
  const int c_end = min(nchunk_all, c_begin + per);

  // ---- one-time LDS setup: zero the padding cells (image cells outside the picture; everything else is rewritten by
  // every chunk), BN affine table.  The producers start their first global loads before this (see below). ----
  auto lds_setup = [&]() __attribute__((always_inline)) {
    const int cells = rows * P;
    for (int i = tid; i < cells; i += 768) {
      const int r = i / P, cpos = i - r * P;
      const int iy = gy0 - 1 + r, ix = gx0 - 1 + cpos;
      if (iy < 0 || iy >= p.H || ix < 0 || ix >= p.W || cpos > p.tw + 1) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int pt = 0; pt < NP; ++pt) {
            uint4* z = reinterpret_cast<uint4*>(smem + b * IMG + pt * PLANE + i * 32);
            z[0] = make_uint4(0u, 0u, 0u, 0u);
            z[1] = make_uint4(0u, 0u, 0u, 0u);
          }
      }
    }
    for (int i = tid; i < Cpad; i += 768) {
      abtab[i] = i < p.Cin ? p.pa[i] : 0.f;
      abtab[Cpad + i] = i < p.Cin ? p.pb[i] : 0.f;
    }
  };

  f32x4 acc[MPW];
#pragma unroll
  for (int m = 0; m < MPW; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int npix = p.th * p.tw;

  if (producer) {
    // =========================== producer waves ===========================
    const int pgroup = (wave - 4) >> 2;  // 0: chunks c_begin+1, +3, ...   1: chunks c_begin, +2, ...
    const int ptid = tid & 255;
    const int nq = p.tw >> 2;
    int s_off[NR][8], s_lds[NR];
    bool s_ok[NR];
    int s_o[NR];
    // 8-lane group = 2 channel octets x RG rows x (4/RG) pixel quads (RG = 4: conflict-free 16-byte LDS writes;
    // RG = 2: 2-way, fewer idle lanes when th+2 is not a multiple of 4)
    const int rgs = p.rg == 2 ? 1 : 2;        // log2(RG)
    const int qpg = 4 >> rgs;                 // quads per group
    const int nqg = (nq + qpg - 1) / qpg;     // groups per row band
    const int nrg = (rows + p.rg - 1) >> rgs; // row bands
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int lu = ptid + 256 * i;
      const int o = lu & 1, rr = (lu >> 1) & (p.rg - 1), qq = (lu >> (1 + rgs)) & (qpg - 1), u = lu >> 3;
      const int R = u / nqg, Q = (u - R * nqg) * qpg + qq;
      const int r = (R << rgs) + rr;
      const int iy = gy0 - 1 + r, ix = gx0 + 4 * Q;
      bool ok = (R < nrg) && (r < rows) && (Q < nq) && (iy >= 0) && (iy < p.H) && (ix < p.W);
#ifdef RLN_DIAG
      if ((p.dbg & 16) && (r == 0 || r == rows - 1)) ok = false;  // timing ablation: no halo rows
#endif
      s_ok[i] = ok;
      const int goff = ok ? iy * p.W + ix : 0;
#pragma unroll
      for (int cc = 0; cc < 8; ++cc) s_off[i][cc] = (o * 8 + cc) * p.cs + goff;  // relative to the chunk's first plane
      s_lds[i] = ((ok ? r : 0) * P + 1 + 4 * (ok ? Q : 0)) * 32 + o * 16;
      s_o[i] = o;
    }
    // halo columns: unit = (row, side, channel pair)
    const int h_cp = ptid & 7, h_side = (ptid >> 3) & 1, h_r = ptid >> 4;
    const int h_iy = gy0 - 1 + h_r, h_ix = h_side ? gx0 + p.tw : gx0 - 1;
    const bool h_ok = (h_r < rows) && (h_iy >= 0) && (h_iy < p.H) && (h_ix >= 0) && (h_ix < p.W);
    const int h_goff = (h_ok ? h_iy * p.W + h_ix : 0) + 2 * h_cp * p.cs;
    const int h_lds = ((h_ok ? h_r : 0) * P + (h_side ? p.tw + 1 : 0)) * 32 + h_cp * 4;
    constexpr int NB = (5 * NP * 64 + 255) / 256;  // weight-fragment entries per producer thread

    float4 sreg[NR][8];
    float hreg[2];
    uint4 breg[NB];
    // the tail chunk of a Cin that is no multiple of 16 starts at Cin-16 (see d3_pack_k): no clamping needed
    auto chunk_base = [&](int chunk) __attribute__((always_inline)) { return min(chunk * 16, max(p.Cin - 16, 0)); };
    // issue_part / commit_part: part 0 = weight fragments + halo columns, part 1+i = staging round i
    auto issue_part = [&](int chunk, auto PART) __attribute__((always_inline)) {
      constexpr int part = decltype(PART)::value;
#ifdef RLN_DIAG
      if (p.dbg & 1) return;
#endif
      const float* base = Sn + (long long)chunk_base(chunk) * p.cs;  // wave-uniform
      if constexpr (part == 0) {
        const uint4* wp = p.wpk + (long long)chunk * 5 * NP * 64;
#pragma unroll
        for (int i = 0; i < NB; ++i) breg[i] = wp[min(ptid + 256 * i, 5 * NP * 64 - 1)];
        hreg[0] = base[h_goff];
        hreg[1] = base[h_goff + p.cs];
      } else {
        constexpr int i = part - 1;
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) sreg[i][cc] = *reinterpret_cast<const float4*>(base + s_off[i][cc]);
      }
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    using P2 = std::integral_constant<int, 2>;
    auto issue = [&](int chunk) __attribute__((always_inline)) {
      issue_part(chunk, P0{});
      issue_part(chunk, P1{});
      if constexpr (NR > 1) issue_part(chunk, P2{});
    };
    auto commit_part = [&](int chunk, unsigned char* buf, auto PART) __attribute__((always_inline)) {
      constexpr int part = decltype(PART)::value;
#ifdef RLN_DIAG
      if (p.dbg & 2) return;
#endif
      const int cb = chunk_base(chunk);
      if constexpr (part == 0) {
        uint4* btile = reinterpret_cast<uint4*>(buf + NP * PLANE);
#pragma unroll
        for (int i = 0; i < NB; ++i)
          if (ptid + 256 * i < 5 * NP * 64) btile[ptid + 256 * i] = breg[i];
        if (h_ok) {
          const int c0 = cb + 2 * h_cp;
          unsigned parts[NP];
          split2<DT, NP>(fmaxf(fmaf(abtab[c0], hreg[0], abtab[Cpad + c0]), 0.f),
                         fmaxf(fmaf(abtab[c0 + 1], hreg[1], abtab[Cpad + c0 + 1]), 0.f), parts);
#pragma unroll
          for (int pt = 0; pt < NP; ++pt) *reinterpret_cast<unsigned*>(buf + pt * PLANE + h_lds) = parts[pt];
        }
        return;
      }
      constexpr int i = part > 0 ? part - 1 : 0;
      const float* ab = abtab + cb + s_o[i] * 8;
      const float4 a0 = *reinterpret_cast<const float4*>(ab), a1 = *reinterpret_cast<const float4*>(ab + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(ab + Cpad);
      const float4 b1 = *reinterpret_cast<const float4*>(ab + Cpad + 4);
      const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
      const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      if (s_ok[i]) {
#pragma unroll
        for (int px = 0; px < 4; ++px) {
          unsigned parts[4][NP];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float4 u0 = sreg[i][2 * k], u1 = sreg[i][2 * k + 1];
            const float x0 = px == 0 ? u0.x : px == 1 ? u0.y : px == 2 ? u0.z : u0.w;
            const float x1 = px == 0 ? u1.x : px == 1 ? u1.y : px == 2 ? u1.z : u1.w;
            split2<DT, NP>(fmaxf(fmaf(av[2 * k], x0, bv[2 * k]), 0.f),
                           fmaxf(fmaf(av[2 * k + 1], x1, bv[2 * k + 1]), 0.f), parts[k]);
          }
#pragma unroll
          for (int pt = 0; pt < NP; ++pt)
            *reinterpret_cast<uint4*>(buf + pt * PLANE + s_lds[i] + px * 32) =
                make_uint4(parts[0][pt], parts[1][pt], parts[2][pt], parts[3][pt]);
        }
      }
    };
    auto commit = [&](int chunk, unsigned char* buf) __attribute__((always_inline)) {
      commit_part(chunk, buf, P0{});
      commit_part(chunk, buf, P1{});
      if constexpr (NR > 1) commit_part(chunk, buf, P2{});
    };
    // steady state: every part's registers are refilled (chunk `next`) right after they are consumed (chunk `cur`)
    auto commit_issue = [&](int cur, unsigned char* buf, int next) __attribute__((always_inline)) {
      commit_part(cur, buf, P0{});
      issue_part(next, P0{});
      commit_part(cur, buf, P1{});
      issue_part(next, P1{});
      if constexpr (NR > 1) {
        commit_part(cur, buf, P2{});
        issue_part(next, P2{});
      }
    };

    // Iteration i (consumers multiply chunk c_begin+i, one barrier at its end): the group with (i & 1) == pgroup
    // commits chunk c_begin+i+1 and re-issues its registers for chunk c_begin+i+3; the other group only waits.
    // Group 1 also stages the first chunk before iteration 0.
    const int nck = c_end - c_begin;
    if (pgroup == 1) {
      if (nck > 0) issue(c_begin);     // in flight while the padding cells are zeroed
      lds_setup();
      __syncthreads();
      if (nck > 0) {
        commit(c_begin, smem + (c_begin & 1) * IMG);
        if (nck > 2) issue(c_begin + 2);
      }
      __syncthreads();                 // start of iteration 0
      if (nck > 0) __syncthreads();    // iteration 0: the other group commits
      int i = 1;
      for (; i + 3 < nck; i += 2) {    // steady state: no branches between the loads and their use
        commit_issue(c_begin + i + 1, smem + ((c_begin + i + 1) & 1) * IMG, c_begin + i + 3);
        __syncthreads();
        __syncthreads();
      }
      for (; i < nck; ++i) {
        if (((i & 1) == 1) && i + 1 < nck) commit(c_begin + i + 1, smem + ((c_begin + i + 1) & 1) * IMG);
        __syncthreads();
      }
    } else {
      if (nck > 1) issue(c_begin + 1);
      lds_setup();
      __syncthreads();
      __syncthreads();                 // start of iteration 0
      int i = 0;
      for (; i + 3 < nck; i += 2) {
        commit_issue(c_begin + i + 1, smem + ((c_begin + i + 1) & 1) * IMG, c_begin + i + 3);
        __syncthreads();
        __syncthreads();
      }
      for (; i < nck; ++i) {
        if (((i & 1) == 0) && i + 1 < nck) commit(c_begin + i + 1, smem + ((c_begin + i + 1) & 1) * IMG);
        __syncthreads();
      }
    }
  } else {
    // =========================== consumer waves ===========================
    const int lp = lane & 15, lg = lane >> 4;
    int basem[MPW];
#pragma unroll
    for (int m = 0; m < MPW; ++m) {
      const int q = min((wave * MPW + m) * 16 + lp, npix - 1);
      const int ty = q / p.tw, tx = q - ty * p.tw;
      basem[m] = ((ty + 1) * P + tx + 1) * 32 + (lg & 1) * 16;
    }
    int toff[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      const int tap = min(2 * s + (lg >> 1), 8);
      toff[s] = ((tap / 3 - 1) * P + (tap % 3 - 1)) * 32;
    }
    lds_setup();
    __syncthreads();
    __syncthreads();  // first chunk staged
    for (int chunk = c_begin; chunk < c_end; ++chunk) {
      const unsigned char* img = smem + (chunk & 1) * IMG;
      const uint4* btile = reinterpret_cast<const uint4*>(img + NP * PLANE);
      // straight-line MFMA phase (M-tiles beyond the tile read clamped addresses and are dropped in the epilogue);
      // the fragments of step i+1 are read while step i multiplies
      // straight-line MFMA phase (M-tiles beyond the tile read clamped addresses and are dropped in the epilogue).
      // One consumer wave per SIMD: LDS latency (~200 cycles under load) is covered by reading the A fragments
      // DEPTH steps (48 MFMA cycles each at 2 parts) ahead through a register ring.
      constexpr int DEPTH = 5, RING = DEPTH + 1, STEPS = 5 * MPW;
      uint4 af[RING][NP], bf[2][NP];
#ifdef RLN_DIAG
      if (p.dbg & 4) {
        __syncthreads();
        continue;
      }
#endif
#pragma unroll
      for (int pt = 0; pt < NP; ++pt) bf[0][pt] = btile[pt * 64 + lane];
#pragma unroll
      for (int i = 0; i < DEPTH && i < STEPS; ++i) {
        const int s0 = i / MPW, m0 = i - s0 * MPW;
#pragma unroll
        for (int pt = 0; pt < NP; ++pt)
          af[i % RING][pt] = *reinterpret_cast<const uint4*>(img + pt * PLANE + basem[m0] + toff[s0]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < STEPS; ++i) {
        const int s = i / MPW, m = i - s * MPW;
        if (i + DEPTH < STEPS) {
          const int s1 = (i + DEPTH) / MPW, m1 = (i + DEPTH) - s1 * MPW;
#pragma unroll
          for (int pt = 0; pt < NP; ++pt)
            af[(i + DEPTH) % RING][pt] = *reinterpret_cast<const uint4*>(img + pt * PLANE + basem[m1] + toff[s1]);
        }
        if (m == 0 && s + 1 < 5) {
#pragma unroll
          for (int pt = 0; pt < NP; ++pt) bf[(s + 1) & 1][pt] = btile[((s + 1) * NP + pt) * 64 + lane];
        }
        acc[m] = mfma_split<DT, NP>(af[i % RING], bf[s & 1], acc[m]);
        __builtin_amdgcn_sched_barrier(0);  // keep the read-ahead distance: the scheduler would sink the reads
      }
      __syncthreads();
    }
  }

  // ---- epilogue (consumers): lane holds 4 consecutive pixels (rows 4*lg..4*lg+3 of the M-tile) of channel lp ----
  float s1 = 0.f, s2 = 0.f;
  const bool raw = p.ksplit > 1;
  if (!producer) {
    const int lp = lane & 15, lg = lane >> 4;
    const int j = lp;
    const bool jv = j < p.Cout;
    const float bias = (jv && !raw && p.bias) ? p.bias[j] : 0.f;
    const float sc = (jv && !raw && p.nscale) ? p.nscale[(long long)n * p.Cout + j] : 1.f;
    float* outn = p.out + (raw ? (long long)blockIdx.y * p.split_stride : 0) + (long long)n * p.out_ns +
                  (long long)(jv ? j : 0) * p.out_cs;
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
  }
  if (p.stat_partial != nullptr && !raw) {
    if (!producer) {
      s1 = group4_sum(s1);
      s2 = group4_sum(s2);
      if ((lane >> 4) == 0) {
        red[(wave * 16 + (lane & 15)) * 2 + 0] = s1;
        red[(wave * 16 + (lane & 15)) * 2 + 1] = s2;
      }
    }
    __syncthreads();
    if (tid < 2 * p.Cout) {
      const int jj = tid >> 1, w2 = tid & 1;
      const float t = red[(0 * 16 + jj) * 2 + w2] + red[(1 * 16 + jj) * 2 + w2] + red[(2 * 16 + jj) * 2 + w2] +
                      red[(3 * 16 + jj) * 2 + w2];
      const long long brow = (long long)n * gridDim.x + bx;
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

// Tiles are full-width strips whenever a row fits (W <= 160): every (channel, tile) region is then ONE contiguous run
// of (th+2)*W floats in HBM (no column halo; DRAM-page friendly), which streams about twice as fast as column tiles.
void d3_fwd_pick_tile(int H, int W, int np, int* th, int* tw, int* rg) {
  (void)H;
  if (W <= 160) {
    *tw = W;
    const int cap = np >= 3 ? 320 : 640;  // pixels per tile (LDS: two image buffers)
    int t = cap / W;
    t = t >= 8 ? 8 : (t >= 4 ? 4 : (t >= 2 ? 2 : 1));
    *th = t;
  } else {
    *tw = 80;
    *th = np >= 3 ? 4 : 8;
  }
  // rows per staging lane group: 4 (conflict-free LDS writes) unless 2 saves a staging round
  auto rounds = [&](int g) {
    const int qpg = 4 / g;
    return (8 * ((*th + 2 + g - 1) / g) * ((*tw / 4 + qpg - 1) / qpg) + 255) / 256;
  };
  *rg = rounds(2) < rounds(4) ? 2 : 4;
}

template <int MPW, int NR, int NP, int DT>
static int d3_fwd_launch_t(const D3Fwd& p, int N, hipStream_t s) {
  const int P = p.tw + 3, rows = p.th + 2;
  const int Cpad = ((p.Cin + 15) / 16) * 16;
  const size_t lds = 2 * ((size_t)NP * rows * P * 32 + (size_t)5 * NP * 1024) + (size_t)2 * Cpad * 4 + 4 * 16 * 2 * 4;
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
  hipLaunchKernelGGL(kern, grid, dim3(768), lds, s, p);
  return (int)hipGetLastError();
}

int d3_fwd_launch(const D3Fwd& p, int N, int np, int dt, hipStream_t s) {
  if (!d3_fwd_supported(p)) return -4;
  // tile capacity: 64*MPW pixels; staging rounds NR = ceil(8 * ceil((th+2)/4) * (tw/4) / 256)
  const int npix = p.th * p.tw;
  const int qpg = 4 / p.rg;
  const int lane_units = 8 * ((p.th + 2 + p.rg - 1) / p.rg) * ((p.tw / 4 + qpg - 1) / qpg);
  const int nr = (lane_units + 255) / 256;
  if ((p.th + 2) * 16 > 256 || nr > 2 || (p.tw & 3) || (p.rg != 2 && p.rg != 4)) return -4;
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
