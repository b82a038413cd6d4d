// Dense-block 3x3 kernels on the 16-bit MFMA pipe with split fp32 operands (see dense3.h).
#include "dense3.h"
#include "igemm.h"
#include "split16.h"
#include "storage.h"

#include <algorithm>
#include <cstdio>
#include <type_traits>

namespace rln {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
// an empty "use" of a loaded register: makes the wait-count pass place the exact vmcnt for it here
template <class T>
__device__ __forceinline__ void touch_reg(const T& v) {
  if constexpr (sizeof(T) == 16) {
    const u32x4 t = __builtin_bit_cast(u32x4, v);
    asm volatile("" ::"v"(t));
  } else if constexpr (sizeof(T) == 8) {
    const u32x2 t = __builtin_bit_cast(u32x2, v);
    asm volatile("" ::"v"(t));
  } else {
    static_assert(sizeof(T) == 4, "register-sized value");
    const unsigned t = __builtin_bit_cast(unsigned, v);
    asm volatile("" ::"v"(t));
  }
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
        if (n < q.cout && ch < q.cin && ch >= grp * 16)
          val = sat16<DT>(w[((long long)n * q.cin + ch) * 9 + tap] * w_prescale<DT>());  // split16.h: f16 range handling
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
template <int MPW, int NR, int NP, int DT, int ST>
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
  const SP<ST> Sn = SP<ST>(p.S) + (long long)n * p.ns;

  // chunk range of this block (split-K over blockIdx.y)
  // (c_first > 0: the launch only adds the chunks from c_first on to the raw sums p.partial_in of the earlier chunks,
  // which a paired launch of the previous layer accumulated -- see d3_fwd2_k)
  constexpr bool FIN = (NP == 1);  // pairs exist for one-part operands only: the other variants keep their code
  const int c_first = FIN ? p.c_first : 0;
  const int per = (nchunk_all - c_first + (int)gridDim.y - 1) / (int)gridDim.y;
  const int c_begin = c_first + (int)blockIdx.y * per;
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
    // All 8 producer waves work on EVERY chunk: thread -> one staging unit (8 channels x 4 pixels); threads below
    // rows*16 additionally fetch one halo-column channel pair, every thread two weight-fragment entries.  Every thread
    // issues the same 12 loads per chunk (clamped where it has nothing to fetch): no branches around loads, exact wait
    // counters.  Two register sets hold the chunks of even / odd iteration: a chunk's loads stay in flight for two
    // iterations (~130 KB per CU outstanding).
    const int ptid = tid - 256;
    // pixels per staging unit.  (16-byte loads of 8 bf16 pixels were measured: no faster than 8-byte loads of 4 -- the
    // request width is not what limits this kernel -- and twice the staging registers.)
    constexpr int PXU = 4;
    const int nq = p.tw / PXU;
    const int rgs = p.rg == 2 ? 1 : 2;        // log2(rows per 8-lane group)
    const int qpg = 4 >> rgs;                 // quads per group
    const int nqg = (nq + qpg - 1) / qpg;     // groups per row band
    const int nrg = (rows + p.rg - 1) >> rgs; // row bands
    int s_off[8], s_lds, s_o;
    bool s_ok;
    {
      const int lu = ptid;
      const int o = lu & 1, rr = (lu >> 1) & (p.rg - 1), qq = (lu >> (1 + rgs)) & (qpg - 1), u = lu >> 3;
      const int R = u / nqg, Q = (u - R * nqg) * qpg + qq;
      const int r = (R << rgs) + rr;
      const int iy = gy0 - 1 + r, ix = gx0 + PXU * Q;
      bool ok = (R < nrg) && (r < rows) && (Q < nq) && (iy >= 0) && (iy < p.H) && (ix < p.W);
#ifdef RLN_DIAG
      if ((p.dbg & 16) && (r == 0 || r == rows - 1)) ok = false;  // timing ablation: no halo rows
#endif
      s_ok = ok;
      const int goff = ok ? iy * p.W + ix : 0;
#pragma unroll
      for (int cc = 0; cc < 8; ++cc) s_off[cc] = (o * 8 + cc) * p.cs + goff;  // relative to the chunk's first plane
      s_lds = ((ok ? r : 0) * P + 1 + PXU * (ok ? Q : 0)) * 32 + o * 16;
      s_o = o;
    }
    // halo columns: unit = (row, side, channel pair)
    const int h_cp = ptid & 7, h_side = (ptid >> 3) & 1, h_r = ptid >> 4;
    const int h_iy = gy0 - 1 + h_r, h_ix = h_side ? gx0 + p.tw : gx0 - 1;
    const bool h_ok = (h_r < rows) && (h_iy >= 0) && (h_iy < p.H) && (h_ix >= 0) && (h_ix < p.W);
    const int h_goff = (h_ok ? h_iy * p.W + h_ix : 0) + 2 * h_cp * p.cs;
    const int h_lds = ((h_ok ? h_r : 0) * P + (h_side ? p.tw + 1 : 0)) * 32 + h_cp * 4;
    constexpr int NBE = 5 * NP * 64;             // weight-fragment entries per chunk

    // (The weight fragments do not travel through the producers: as a third member of the register set they were kept
    // in scratch across the barrier, and every scratch reload is a vmcnt(0) -- scratch shares the counter -- that drained
    // the chunk loads issued one iteration earlier.  The consumer waves fetch them: see wload / wstore below.)
    struct Stage {
      typename SRaw<ST>::r4 s[8];  // narrow registers until the commit widens them
      typename SRaw<ST>::r1 h[2];
    };
    Stage RA;
    // the tail chunk of a Cin that is no multiple of 16 starts at Cin-16 (see d3_pack_k): no clamping needed
    auto chunk_base = [&](int chunk) __attribute__((always_inline)) { return min(chunk * 16, max(p.Cin - 16, 0)); };
    auto issue = [&](int chunk, Stage& R) __attribute__((always_inline)) {
#ifdef RLN_DIAG
      if (p.dbg & 1) return;
#endif
      const SP<ST> base = Sn + (long long)chunk_base(chunk) * p.cs;  // wave-uniform
#pragma unroll
      for (int cc = 0; cc < 8; ++cc) R.s[cc] = base.raw4(s_off[cc]);
      R.h[0] = base.raw1(h_goff);
      R.h[1] = base.raw1(h_goff + p.cs);
    };
    auto commit = [&](int chunk, unsigned char* buf, const Stage& R) __attribute__((always_inline)) {
#ifdef RLN_DIAG
      if (p.dbg & 2) return;
#endif
      const int cb = chunk_base(chunk);
      if (h_ok) {
        const int c0 = cb + 2 * h_cp;
        unsigned parts[NP];
        split2<DT, NP>(relu16<DT>(fmaf(abtab[c0], SRaw<ST>::w1(R.h[0]), abtab[Cpad + c0])),
                       relu16<DT>(fmaf(abtab[c0 + 1], SRaw<ST>::w1(R.h[1]), abtab[Cpad + c0 + 1])), parts);
#pragma unroll
        for (int pt = 0; pt < NP; ++pt) *reinterpret_cast<unsigned*>(buf + pt * PLANE + h_lds) = parts[pt];
      }
      if (s_ok) {
        const float* ab = abtab + cb + s_o * 8;
        const float4 a0 = *reinterpret_cast<const float4*>(ab), a1 = *reinterpret_cast<const float4*>(ab + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(ab + Cpad);
        const float4 b1 = *reinterpret_cast<const float4*>(ab + Cpad + 4);
        const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int px = 0; px < PXU; ++px) {
          unsigned parts[4][NP];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float4 u0 = SRaw<ST>::w4(R.s[2 * k]), u1 = SRaw<ST>::w4(R.s[2 * k + 1]);
            const float x0 = px == 0 ? u0.x : px == 1 ? u0.y : px == 2 ? u0.z : u0.w;
            const float x1 = px == 0 ? u1.x : px == 1 ? u1.y : px == 2 ? u1.z : u1.w;
            split2<DT, NP>(relu16<DT>(fmaf(av[2 * k], x0, bv[2 * k])),
                           relu16<DT>(fmaf(av[2 * k + 1], x1, bv[2 * k + 1])), parts[k]);
          }
#pragma unroll
          for (int pt = 0; pt < NP; ++pt)
            *reinterpret_cast<uint4*>(buf + pt * PLANE + s_lds + px * 32) =
                make_uint4(parts[0][pt], parts[1][pt], parts[2][pt], parts[3][pt]);
        }
      }
    };
    const int nck = c_end - c_begin;
    auto bufof = [&](int i) __attribute__((always_inline)) { return smem + ((c_begin + i) & 1) * IMG; };
    // (Measured with bf16 stacks and not kept: four register sets = loads in flight for four iterations, -7 %; 16-byte
    // loads of 8 pixels per lane, +-0; 320-pixel column tiles with one register set and two workgroups per CU, -18 %.
    // In steady state the kernel moves 4.0 - 4.6 TB/s of its strip pattern.  A persistent form was built and measured
    // too -- 8 XCDs x 32 blocks, each walking its share of the (sample, tile) order as ONE pipeline of (tile, chunk)
    // items, so that the next tile's first chunks load under the last MFMAs and the epilogue stores of this one, with
    // the padding cells and the BN table set up once per CU: parity-green, 4.67 -> 4.62 ms per step at the two large
    // levels and +0.08 ms at the 30x40 / 15x20 levels (one tile per block there; every cell re-zeroed per tile): the
    // ~55 us a level-0 launch costs beyond its per-plane rate is NOT block turnover, and the form is not kept.)
    // chunk c_begin+i lives in set i & 1.  Iteration i (consumers multiply chunk i): commit chunk i+1 into the other
    // LDS buffer, refill its register set with chunk i+3.
    // The prologue issues its loads unconditionally (a chunk index past the range is clamped: a redundant load nobody
    // commits): with "if (nck > 2) issue(...)" the steady loop is entered from a join of paths with different numbers of
    // loads in flight, and the wait-count pass then drains everything (vmcnt(0)) at the loop top of EVERY trip.
    Stage RB;
    if (nck <= 0) {  // (a split-K block without chunks: only the barriers)
      lds_setup();
      __syncthreads();
      __syncthreads();
    } else {
    issue(c_begin, RA);
    issue(c_begin + min(1, nck - 1), RB);
    lds_setup();
    __syncthreads();
    commit(c_begin, bufof(0), RA);
    issue(c_begin + min(2, nck - 1), RA);
    __syncthreads();  // start of iteration 0
    int i = 0;
    for (; i + 4 < nck; i += 2) {  // steady state, two iterations per trip, no branches around the loads
      commit(c_begin + i + 1, bufof(i + 1), RB);
      issue(c_begin + i + 3, RB);
      __syncthreads();
      commit(c_begin + i + 2, bufof(i + 2), RA);
      issue(c_begin + i + 4, RA);
      __syncthreads();
    }
    for (; i < nck; ++i) {
      if (i + 1 < nck) {
        if ((i + 1) & 1) commit(c_begin + i + 1, bufof(i + 1), RB); else commit(c_begin + i + 1, bufof(i + 1), RA);
        if (i + 3 < nck) {
          if ((i + 3) & 1) issue(c_begin + i + 3, RB); else issue(c_begin + i + 3, RA);
        }
      }
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
    // weight fragments of a chunk: global (L2-resident, 5 KB per part) -> registers at the start of the previous chunk's
    // MFMA phase -> the other image's fragment tile before the barrier that publishes that image
    constexpr int NBE = 5 * NP * 64;        // weight-fragment entries per chunk
    constexpr int NBC = (NBE + 255) / 256;  // per consumer thread
    u32x4 wreg[NBC];  // (a native vector type: as HIP's uint4 struct the array stayed a stack object)
    auto wload = [&](int chunk) __attribute__((always_inline)) {
      const u32x4* wp = reinterpret_cast<const u32x4*>(p.wpk) + (long long)chunk * NBE;
#pragma unroll
      for (int i = 0; i < NBC; ++i) wreg[i] = wp[min(tid + 256 * i, NBE - 1)];
    };
    auto wstore = [&](int chunk) __attribute__((always_inline)) {
      u32x4* bt = reinterpret_cast<u32x4*>(smem + (chunk & 1) * IMG + NP * PLANE);
#pragma unroll
      for (int i = 0; i < NBC; ++i)
        if (i + 1 < NBC || tid + 256 * i < NBE) bt[tid + 256 * i] = wreg[i];
    };
    if (c_begin < c_end) wload(c_begin);
    lds_setup();
    __syncthreads();
    if (c_begin < c_end) wstore(c_begin);
    __syncthreads();  // first chunk staged
    for (int chunk = c_begin; chunk < c_end; ++chunk) {
      const unsigned char* img = smem + (chunk & 1) * IMG;
      const uint4* btile = reinterpret_cast<const uint4*>(img + NP * PLANE);
      const int nxt = min(chunk + 1, c_end - 1);  // (the last chunk refetches its own fragments: no branch around the loads)
      wload(nxt);
      // straight-line MFMA phase (M-tiles beyond the tile read clamped addresses and are dropped in the epilogue);
      // the fragments of step i+1 are read while step i multiplies
      // straight-line MFMA phase (M-tiles beyond the tile read clamped addresses and are dropped in the epilogue).
      // One consumer wave per SIMD: LDS latency (~200 cycles under load) is covered by reading the A fragments
      // DEPTH steps (48 MFMA cycles each at 2 parts) ahead through a register ring.
      // (three parts: a 2-deep ring -- with 5 the variant needs ~190 VGPRs, and its in-loop spills made every bf16x3
      // forward differ from the last by up to 4e-3: tools/forward_stress.py)
      constexpr int DEPTH = NP >= 3 ? 2 : 5, RING = DEPTH + 1, STEPS = 5 * MPW;
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
      if (nxt != chunk) wstore(nxt);
      __syncthreads();
    }
  }

  // ---- epilogue (consumers): lane holds 4 consecutive pixels (rows 4*lg..4*lg+3 of the M-tile) of channel lp ----
  mfma_drain();  // wait states between the MFMA chain and the first accumulator read (split16.h)
  float s1 = 0.f, s2 = 0.f;
  const bool raw = p.ksplit > 1;
  if (!producer) {
    const int lp = lane & 15, lg = lane >> 4;
    const int j = lp;
    const bool jv = j < p.Cout;
    const float bias = (jv && !raw && p.bias) ? p.bias[j] : 0.f;
    const float sc = (jv && !raw && p.nscale) ? p.nscale[(long long)n * p.Cout + j] : 1.f;
    const SP<ST> outn = SP<ST>(p.out) + ((raw ? (long long)blockIdx.y * p.split_stride : 0) + (long long)n * p.out_ns +
                                         (long long)(jv ? j : 0) * p.out_cs);
#pragma unroll
    for (int m = 0; m < MPW; ++m) {
      const int q = (wave * MPW + m) * 16 + lg * 4;
      const int ty = q / p.tw, tx = q - ty * p.tw;
      const int gy = gy0 + ty, gx = gx0 + tx;
      const bool ok = jv && (q < npix) && (gy < p.H) && (gx < p.W);
      float v[4];
      float4 pin = make_float4(0.f, 0.f, 0.f, 0.f);
      if (FIN && p.partial_in != nullptr && ok)
        pin = *reinterpret_cast<const float4*>(p.partial_in + ((long long)n * p.Cout + j) * ((long long)p.H * p.W) +
                                               (long long)gy * p.W + gx);
      const float pv[4] = {pin.x, pin.y, pin.z, pin.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = st_round<ST>(fmaf(acc[m][r] + pv[r], w_unscale<DT>(), bias) * sc);  // statistics of the tensor as it is stored
        if (ok) {
          s1 += v[r];
          s2 += v[r] * v[r];
        }
      }
      if (ok) outn.st4((long long)gy * p.W + gx, v[0], v[1], v[2], v[3]);
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

// =============================================================================================
// forward of TWO consecutive layers of a block over the input channels they share (layer-alternating pipeline)
//
// Layer j+1 consumes every input channel of layer j (plus layer j's 16 outputs), each through its own BatchNorm.  The
// two LDS images of d3_fwd_k's double buffer become "layer j's image" and "layer j+1's image" of the SAME 16-channel
// chunk: a chunk is LOADED once, converted twice (iteration 2c-1: image 0 with layer j's table, iteration 2c: image 1
// with layer j+1's), and the consumers alternate between the two images and two accumulator sets.  The loads halve per
// layer, a register set stays in flight for three iterations, and the conversion of one image still overlaps the MFMAs
// on the other.  Layer j leaves through the normal epilogue; layer j+1's raw sums over the shared chunks go to
// p.partial_out, and a d3_fwd_k launch restricted to its last chunk (c_first, partial_in) completes it once layer j's
// output and statistics exist (the new channels' BatchNorm needs batch statistics: a grid-wide dependency).
// Measured (bf16 stacks, 120x160, batch 64): 317 us for the pair against 2 x 216 us, but the one-chunk finishing launch
// costs 73 us (fp32 raw sums out and back in), so a pair nets ~40 us: 18.29 -> 18.09 ms per step.  Conversion and LDS
// reads are per layer and do not shrink; only the global loads are shared.  One-part operands only (see the launcher).
// =============================================================================================
template <int MPW, int NP, int DT, int ST>
__global__ __launch_bounds__(768, 3) void d3_fwd2_k(const D3Fwd p) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool producer = wave >= 4;
  const int P = p.tw + 3;  // odd pixel pitch
  const int rows = p.th + 2;
  const int PLANE = rows * P * 32;
  const int IMG = NP * PLANE + 5 * NP * 1024;  // one image: parts + weight fragments of the (layer, chunk) item
  const int nck = p.Cin >> 4;                  // shared chunks (Cin of the first layer: a multiple of 16)
  const int Cpad = nck * 16;
  float* abtab = reinterpret_cast<float*>(smem + 2 * IMG);  // [2 layers][a[Cpad], b[Cpad]]
  float* red = abtab + 4 * Cpad;

  int bx = blockIdx.x, n = blockIdx.z;
  {
    const unsigned lin = blockIdx.x + gridDim.x * blockIdx.z;
    const unsigned lg = xcd_logical_block(lin, gridDim.x * gridDim.z);
    n = (int)(lg / gridDim.x);
    bx = (int)(lg - (unsigned)n * gridDim.x);
  }
  const int tile_y = bx / p.tiles_x, tile_x = bx - tile_y * p.tiles_x;
  const int gy0 = tile_y * p.th, gx0 = tile_x * p.tw;
  const SP<ST> Sn = SP<ST>(p.S) + (long long)n * p.ns;
  const int I = 2 * nck;  // items: (chunk t >> 1, layer t & 1); item t lives in image t & 1

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
      abtab[i] = p.pa[i];
      abtab[Cpad + i] = p.pb[i];
      abtab[2 * Cpad + i] = p.pa2[i];
      abtab[3 * Cpad + i] = p.pb2[i];
    }
  };

  const int npix = p.th * p.tw;
  float s1 = 0.f, s2 = 0.f;  // (the accumulators live in the consumer branch only: the producers' registers hold two chunks)

  if (producer) {
    const int ptid = tid - 256;
    constexpr int PXU = 4;
    const int nq = p.tw / PXU;
    const int rgs = p.rg == 2 ? 1 : 2;
    const int qpg = 4 >> rgs;
    const int nqg = (nq + qpg - 1) / qpg;
    const int nrg = (rows + p.rg - 1) >> rgs;
    int s_off[8], s_lds, s_o;
    bool s_ok;
    {
      const int lu = ptid;
      const int o = lu & 1, rr = (lu >> 1) & (p.rg - 1), qq = (lu >> (1 + rgs)) & (qpg - 1), u = lu >> 3;
      const int R = u / nqg, Q = (u - R * nqg) * qpg + qq;
      const int r = (R << rgs) + rr;
      const int iy = gy0 - 1 + r, ix = gx0 + PXU * Q;
      const bool ok = (R < nrg) && (r < rows) && (Q < nq) && (iy >= 0) && (iy < p.H) && (ix < p.W);
      s_ok = ok;
      const int goff = ok ? iy * p.W + ix : 0;
#pragma unroll
      for (int cc = 0; cc < 8; ++cc) s_off[cc] = (o * 8 + cc) * p.cs + goff;
      s_lds = ((ok ? r : 0) * P + 1 + PXU * (ok ? Q : 0)) * 32 + o * 16;
      s_o = o;
    }
    const int h_cp = ptid & 7, h_side = (ptid >> 3) & 1, h_r = ptid >> 4;
    const int h_iy = gy0 - 1 + h_r, h_ix = h_side ? gx0 + p.tw : gx0 - 1;
    const bool h_ok = (h_r < rows) && (h_iy >= 0) && (h_iy < p.H) && (h_ix >= 0) && (h_ix < p.W);
    const int h_goff = (h_ok ? h_iy * p.W + h_ix : 0) + 2 * h_cp * p.cs;
    const int h_lds = ((h_ok ? h_r : 0) * P + (h_side ? p.tw + 1 : 0)) * 32 + h_cp * 4;
    struct Stage {  // (the weight fragments are the consumers' business: see d3_fwd_k)
      typename SRaw<ST>::r4 s[8];
      typename SRaw<ST>::r1 h[2];
    };
    auto issue = [&](int chunk, Stage& R) __attribute__((always_inline)) {
#ifdef RLN_DIAG
      if (p.dbg & 1) return;
#endif
      const SP<ST> base = Sn + (long long)chunk * 16 * p.cs;  // wave-uniform
#pragma unroll
      for (int cc = 0; cc < 8; ++cc) R.s[cc] = base.raw4(s_off[cc]);
      R.h[0] = base.raw1(h_goff);
      R.h[1] = base.raw1(h_goff + p.cs);
    };
    // item (chunk, L): the chunk's raw values through layer L's table into image L
    auto commit = [&](int chunk, int L, const Stage& R) __attribute__((always_inline)) {
#ifdef RLN_DIAG
      if (p.dbg & 2) return;
#endif
      const int cb = chunk * 16;
      // the image base is re-materialised as an opaque value at every call: as loop invariants the LDS addresses of the
      // four commit sites of the steady-state loop cost ~40 VGPRs (spills at the 168-register budget of 12 waves)
      int ibase = L * IMG;
      asm volatile("" : "+v"(ibase));
      unsigned char* buf = smem + ibase;
      const float* tab = abtab + L * 2 * Cpad;
      if (h_ok) {
        const int c0 = cb + 2 * h_cp;
        unsigned parts[NP];
        split2<DT, NP>(relu16<DT>(fmaf(tab[c0], SRaw<ST>::w1(R.h[0]), tab[Cpad + c0])),
                       relu16<DT>(fmaf(tab[c0 + 1], SRaw<ST>::w1(R.h[1]), tab[Cpad + c0 + 1])), parts);
#pragma unroll
        for (int pt = 0; pt < NP; ++pt) *reinterpret_cast<unsigned*>(buf + pt * PLANE + h_lds) = parts[pt];
      }
      if (s_ok) {
        const float* ab = tab + cb + s_o * 8;
        const float4 a0 = *reinterpret_cast<const float4*>(ab), a1 = *reinterpret_cast<const float4*>(ab + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(ab + Cpad);
        const float4 b1 = *reinterpret_cast<const float4*>(ab + Cpad + 4);
        const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int px = 0; px < PXU; ++px) {
          unsigned parts[4][NP];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float4 u0 = SRaw<ST>::w4(R.s[2 * k]), u1 = SRaw<ST>::w4(R.s[2 * k + 1]);
            const float x0 = px == 0 ? u0.x : px == 1 ? u0.y : px == 2 ? u0.z : u0.w;
            const float x1 = px == 0 ? u1.x : px == 1 ? u1.y : px == 2 ? u1.z : u1.w;
            split2<DT, NP>(relu16<DT>(fmaf(av[2 * k], x0, bv[2 * k])),
                           relu16<DT>(fmaf(av[2 * k + 1], x1, bv[2 * k + 1])), parts[k]);
          }
#pragma unroll
          for (int pt = 0; pt < NP; ++pt)
            *reinterpret_cast<uint4*>(buf + pt * PLANE + s_lds + px * 32) =
                make_uint4(parts[0][pt], parts[1][pt], parts[2][pt], parts[3][pt]);
        }
      }
    };
    // Chunk c lives in register set c & 1.  Iteration t (consumers multiply item t) commits item t + 1:
    //   t = 2c     : (c, layer 1) from set(c); the set is free afterwards -> refill it with chunk c + 2
    //   t = 2c + 1 : (c + 1, layer 0) from set(c + 1)
    Stage RA, RB;  // (nck >= 1; unconditional, clamped prologue loads: see d3_fwd_k)
    issue(0, RA);
    issue(min(1, nck - 1), RB);
    lds_setup();
    __syncthreads();
    commit(0, 0, RA);
    __syncthreads();  // start of iteration 0
    int c = 0;
    for (; c + 3 < nck; c += 2) {  // steady state: four iterations per trip, no branches around the loads
      commit(c, 1, RA);
      issue(c + 2, RA);
      __syncthreads();
      commit(c + 1, 0, RB);
      __syncthreads();
      commit(c + 1, 1, RB);
      issue(c + 3, RB);
      __syncthreads();
      commit(c + 2, 0, RA);
      __syncthreads();
    }
    for (; c < nck; ++c) {  // tail: iterations 2c and 2c + 1
      if (c & 1) commit(c, 1, RB); else commit(c, 1, RA);
      if (c + 2 < nck) {
        if (c & 1) issue(c + 2, RB); else issue(c + 2, RA);
      }
      __syncthreads();
      if (c + 1 < nck) {
        if ((c + 1) & 1) commit(c + 1, 0, RB); else commit(c + 1, 0, RA);
      }
      __syncthreads();
    }
  } else {
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
    f32x4 acc[2][MPW];
#pragma unroll
    for (int L = 0; L < 2; ++L)
#pragma unroll
      for (int m = 0; m < MPW; ++m) acc[L][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    // weight fragments of item t = (chunk t >> 1, layer t & 1): fetched by the consumers one item ahead (see d3_fwd_k)
    constexpr int NBE = 5 * NP * 64;
    constexpr int NBC = (NBE + 255) / 256;
    u32x4 wreg[NBC];
    auto wload = [&](int t) __attribute__((always_inline)) {
      const u32x4* wp = reinterpret_cast<const u32x4*>((t & 1) ? p.wpk2 : p.wpk) + (long long)(t >> 1) * NBE;
#pragma unroll
      for (int i = 0; i < NBC; ++i) wreg[i] = wp[min(tid + 256 * i, NBE - 1)];
    };
    auto wstore = [&](int t) __attribute__((always_inline)) {
      u32x4* bt = reinterpret_cast<u32x4*>(smem + (t & 1) * IMG + NP * PLANE);
#pragma unroll
      for (int i = 0; i < NBC; ++i)
        if (i + 1 < NBC || tid + 256 * i < NBE) bt[tid + 256 * i] = wreg[i];
    };
    if (nck > 0) wload(0);
    lds_setup();
    __syncthreads();
    if (nck > 0) wstore(0);
    __syncthreads();  // first item staged
    auto mma_item = [&](int ioff, f32x4 (&a)[MPW]) __attribute__((always_inline)) {
#ifdef RLN_DIAG
      if (p.dbg & 4) return;
#endif
      // per-item opaque bases: hoisted out of the chunk loop, the 2 x 50 x NP fragment addresses of the two images are
      // loop invariants the register allocator spills (a scratch reload in front of every MFMA)
      asm volatile("" : "+v"(ioff));
      const unsigned char* img = smem + ioff;
      int bm[MPW];
#pragma unroll
      for (int m = 0; m < MPW; ++m) {
        bm[m] = basem[m];
        asm volatile("" : "+v"(bm[m]));
      }
      const uint4* btile = reinterpret_cast<const uint4*>(img + NP * PLANE);
      constexpr int DEPTH = 3, RING = DEPTH + 1, STEPS = 5 * MPW;
      uint4 af[RING][NP], bf[2][NP];
#pragma unroll
      for (int pt = 0; pt < NP; ++pt) bf[0][pt] = btile[pt * 64 + lane];
#pragma unroll
      for (int i = 0; i < DEPTH && i < STEPS; ++i) {
        const int s0 = i / MPW, m0 = i - s0 * MPW;
#pragma unroll
        for (int pt = 0; pt < NP; ++pt)
          af[i % RING][pt] = *reinterpret_cast<const uint4*>(img + pt * PLANE + bm[m0] + toff[s0]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < STEPS; ++i) {
        const int s = i / MPW, m = i - s * MPW;
        if (i + DEPTH < STEPS) {
          const int s1 = (i + DEPTH) / MPW, m1 = (i + DEPTH) - s1 * MPW;
#pragma unroll
          for (int pt = 0; pt < NP; ++pt)
            af[(i + DEPTH) % RING][pt] = *reinterpret_cast<const uint4*>(img + pt * PLANE + bm[m1] + toff[s1]);
        }
        if (m == 0 && s + 1 < 5) {
#pragma unroll
          for (int pt = 0; pt < NP; ++pt) bf[(s + 1) & 1][pt] = btile[((s + 1) * NP + pt) * 64 + lane];
        }
        a[m] = mfma_split<DT, NP>(af[i % RING], bf[s & 1], a[m]);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    for (int c = 0; c < nck; ++c) {
      wload(2 * c + 1);
      mma_item(0, acc[0]);
      wstore(2 * c + 1);
      __syncthreads();
      const int tn = min(2 * c + 2, 2 * nck - 2);  // (after the last chunk: a harmless refetch, no branch around the loads)
      wload(tn);
      mma_item(IMG, acc[1]);
      if (c + 1 < nck) wstore(tn);
      __syncthreads();
    }
    // ---- epilogue: layer 0 like d3_fwd_k; layer 1: raw sums over the shared chunks ----
    mfma_drain();  // wait states between the MFMA chain and the first accumulator read (split16.h)
    const int j = lp;
    const bool jv = j < p.Cout;
    const float bias = (jv && p.bias) ? p.bias[j] : 0.f;
    const float sc = (jv && p.nscale) ? p.nscale[(long long)n * p.Cout + j] : 1.f;
    const SP<ST> outn = SP<ST>(p.out) + ((long long)n * p.out_ns + (long long)(jv ? j : 0) * p.out_cs);
    float* pout = p.partial_out + ((long long)n * p.Cout + (jv ? j : 0)) * ((long long)p.H * p.W);
#pragma unroll
    for (int m = 0; m < MPW; ++m) {
      const int q = (wave * MPW + m) * 16 + lg * 4;
      const int ty = q / p.tw, tx = q - ty * p.tw;
      const int gy = gy0 + ty, gx = gx0 + tx;
      const bool ok = jv && (q < npix) && (gy < p.H) && (gx < p.W);
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = st_round<ST>(fmaf(acc[0][m][r], w_unscale<DT>(), bias) * sc);
        if (ok) {
          s1 += v[r];
          s2 += v[r] * v[r];
        }
      }
      if (ok) {
        outn.st4((long long)gy * p.W + gx, v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(pout + (long long)gy * p.W + gx) =
            make_float4(acc[1][m][0], acc[1][m][1], acc[1][m][2], acc[1][m][3]);
      }
    }
  }
  if (p.stat_partial != nullptr) {
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
  if ((p.W & 3) || p.W < 20 || p.H < 4) return false;  // (rows of 20: the 15x20 level, whole samples per tile)
  if ((p.cs & 3) || (p.ns & 3) || (p.out_cs & 3) || (p.out_ns & 3)) return false;
  const uintptr_t amask = p.st == ST_BF16 ? 7 : 15;  // a quad of elements per load / store
  if ((reinterpret_cast<uintptr_t>(p.S) & amask) || (reinterpret_cast<uintptr_t>(p.out) & amask)) return false;
  if (p.st == ST_BF16 && p.ksplit > 1) return false;  // split-K partial sums are fp32 scratch
  return true;
}

// Tiles are full-width strips whenever a row fits (W <= 160): every (channel, tile) region is then ONE contiguous run
// of (th+2)*W floats in HBM (no column halo; DRAM-page friendly), which streams about twice as fast as column tiles.
void d3_fwd_pick_tile(int H, int W, int np, int* th, int* tw, int* rg, int st) {
  (void)H;
  const int pxu = 4;  // pixels per staging unit
  (void)st;
  if (W <= 160) {  // (8 x 80 column tiles at W = 160: 4 % slower than 4 x 160 strips in round 2, +0.2 ms per step again after the scratch fix of round 3)
    *tw = W;
    const int cap = np >= 3 ? 320 : 640;  // pixels per tile (LDS: two image buffers)
    int t = cap / W;
    t = t >= 16 && W < 40 ? 16 : (t >= 8 ? 8 : (t >= 4 ? 4 : (t >= 2 ? 2 : 1)));  // narrow levels: whole samples per tile
    *th = t;
  } else {
    *tw = 80;
    *th = np >= 3 ? 4 : 8;
  }
  // rows per staging lane group: 4 (conflict-free LDS writes) unless 2 saves a staging round
  auto units = [&](int g) {
    const int qpg = 4 / g;
    return 8 * ((*th + 2 + g - 1) / g) * ((*tw / pxu + qpg - 1) / qpg);
  };
  *rg = units(4) <= 512 ? 4 : 2;  // 512 producer threads, one unit each
}

template <int MPW, int NR, int NP, int DT, int ST = ST_F32>
static int d3_fwd_launch_t(const D3Fwd& p, int N, hipStream_t s) {
  const int P = p.tw + 3, rows = p.th + 2;
  const int Cpad = ((p.Cin + 15) / 16) * 16;
  const size_t lds = 2 * ((size_t)NP * rows * P * 32 + (size_t)5 * NP * 1024) + (size_t)2 * Cpad * 4 + 4 * 16 * 2 * 4;
  auto kern = d3_fwd_k<MPW, NR, NP, DT, ST>;
  static DevOnce attr_once;
  if (attr_once.first()) {
    const hipError_t attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    if (attr_err != hipSuccess) {  // refused: report it here instead of an opaque launch failure later
      (void)hipGetLastError();
      attr_once.undo();
      return (int)attr_err;
    }
  }
  if (lds > 160 * 1024) return -4;
  dim3 grid((unsigned)(p.tiles_x * p.tiles_y), (unsigned)std::max(1, p.ksplit), (unsigned)N);
  hipLaunchKernelGGL(kern, grid, dim3(768), lds, s, p);
  return (int)hipGetLastError();
}

int d3_fwd_launch(const D3Fwd& p, int N, int np, int dt, hipStream_t s) {
  if (!d3_fwd_supported(p)) return -4;
  if ((p.c_first != 0 || p.partial_in != nullptr) && np != 1) return -4;  // finishing launches: one-part operands only
  // tile capacity: 64*MPW pixels; staging rounds NR = ceil(8 * ceil((th+2)/4) * (tw/4) / 256)
  const int npix = p.th * p.tw;
  const int qpg = 4 / p.rg;
  const int pxu = 4;
  const int lane_units = 8 * ((p.th + 2 + p.rg - 1) / p.rg) * ((p.tw / pxu + qpg - 1) / qpg);
  const int nr = 1;
  if ((p.th + 2) * 16 > 512 || lane_units > 512 || (p.tw % pxu) || (p.rg != 2 && p.rg != 4)) return -4;
#define D3_FWD(MPW_, NR_)                                                                         \
  do {                                                                                            \
    if (p.st == ST_BF16) {                                                                        \
      if (np != 1 || dt != D3_BF16) return -4; /* bf16 storage = plain bf16 operands */           \
      return d3_fwd_launch_t<MPW_, NR_, 1, D3_BF16, ST_BF16>(p, N, s);                            \
    }                                                                                             \
    if (dt == D3_BF16) {                                                                          \
      if (np == 1) return d3_fwd_launch_t<MPW_, NR_, 1, D3_BF16>(p, N, s);                        \
      if (np == 2) return d3_fwd_launch_t<MPW_, NR_, 2, D3_BF16>(p, N, s);                        \
      return d3_fwd_launch_t<MPW_, NR_, 3, D3_BF16>(p, N, s);                                     \
    } else {                                                                                      \
      if (np == 1) return d3_fwd_launch_t<MPW_, NR_, 1, D3_F16>(p, N, s);                         \
      return d3_fwd_launch_t<MPW_, NR_, 2, D3_F16>(p, N, s);                                      \
    }                                                                                             \
  } while (0)
  (void)nr;
  if (npix <= 320) D3_FWD(5, 1);
  if (npix <= 640) D3_FWD(10, 1);
#undef D3_FWD
  return -4;
}

template <int MPW, int NP, int DT, int ST = ST_F32>
static int d3_fwd_pair_launch_t(const D3Fwd& p, int N, hipStream_t s) {
  const int P = p.tw + 3, rows = p.th + 2;
  const size_t lds = 2 * ((size_t)NP * rows * P * 32 + (size_t)5 * NP * 1024) + (size_t)4 * p.Cin * 4 + 4 * 16 * 2 * 4;
  auto kern = d3_fwd2_k<MPW, NP, DT, ST>;
  static DevOnce attr_once;
  if (attr_once.first()) {
    const hipError_t attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    if (attr_err != hipSuccess) {
      (void)hipGetLastError();
      attr_once.undo();
      return (int)attr_err;
    }
  }
  if (lds > 160 * 1024) return -4;
  hipLaunchKernelGGL(kern, dim3((unsigned)(p.tiles_x * p.tiles_y), 1, (unsigned)N), dim3(768), lds, s, p);
  return (int)hipGetLastError();
}

// One-part operands only: with two parts the producer's second register set does not fit the 168-VGPR budget of a
// 12-wave block (216 bytes of scratch; measured 663 us against 2 x 300 us for the two single launches at 120x160).
int d3_fwd_pair_launch(const D3Fwd& p, int N, int np, int dt, hipStream_t s) {
  if (np != 1) return -4;
  if (!d3_fwd_supported(p) || p.ksplit > 1 || (p.Cin & 15) || p.Cin < 16 || p.c_first != 0) return -4;
  if (!p.pa2 || !p.pb2 || !p.wpk2 || !p.partial_out || (reinterpret_cast<uintptr_t>(p.partial_out) & 15)) return -4;
  const int npix = p.th * p.tw;
  const int qpg = 4 / p.rg;
  const int lane_units = 8 * ((p.th + 2 + p.rg - 1) / p.rg) * ((p.tw / 4 + qpg - 1) / qpg);
  if ((p.th + 2) * 16 > 512 || lane_units > 512 || (p.tw % 4) || (p.rg != 2 && p.rg != 4)) return -4;
#define D3_FWD2(MPW_)                                                                            \
  do {                                                                                           \
    if (p.st == ST_BF16) {                                                                       \
      if (np != 1 || dt != D3_BF16) return -4;                                                   \
      return d3_fwd_pair_launch_t<MPW_, 1, D3_BF16, ST_BF16>(p, N, s);                           \
    }                                                                                            \
    if (dt == D3_BF16) {                                                                         \
      if (np == 1) return d3_fwd_pair_launch_t<MPW_, 1, D3_BF16>(p, N, s);                       \
      return -4;                                                                                 \
    }                                                                                            \
    if (np == 1) return d3_fwd_pair_launch_t<MPW_, 1, D3_F16>(p, N, s);                          \
    return -4;                                                                                   \
  } while (0)
  if (npix <= 320) D3_FWD2(5);
  if (npix <= 640) D3_FWD2(10);
#undef D3_FWD2
  return -4;
}

// =============================================================================================
// finishing launch of a paired forward as its own light kernel (see dense3.h: d3_fin_launch)
//
// The second layer of a pair still owes the ONE chunk its partner has just written: out = epilogue(partial_in +
// conv3x3(relu(bn(x16)))).  As a d3_fwd_k launch that is a 12-wave block per CU with nothing to overlap (73 us for
// 157 MB at 120x160, batch 64, bf16 stacks).  Here: 4-wave blocks of one 256-pixel tile, three or more per CU; the
// 16-channel tile (+1 halo) is staged once through BN + ReLU as a [pixel][16 channels] image, the raw sums of the
// shared chunks are prefetched under the staging, 5 K-steps x 4 M-tiles of MFMAs per wave, the forward kernel's
// epilogue (same arithmetic, statistics of the tensor as stored).
// =============================================================================================
template <int NP, int DT, int ST, int TH, int TW>
__global__ __launch_bounds__(256, 3) void d3_fin_k(const D3Fwd p) {
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int MPW = TH * TW / 64;
  constexpr int P = TW + 3, ROWS = TH + 2, PLANE = ROWS * P * 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lp = lane & 15, lg = lane >> 4;
  const int bx = blockIdx.x;
  const int tile_y = bx / p.tiles_x, tile_x = bx - tile_y * p.tiles_x;
  const int gy0 = tile_y * TH, gx0 = tile_x * TW;
  const int n = blockIdx.z;
  float* red = reinterpret_cast<float*>(smem + NP * PLANE);  // [4 waves][16][2]
  const int cb = p.c_first * 16;

  // ---- per-lane geometry, weight fragments and the raw sums of the earlier chunks (in flight under the staging) ----
  int basem[MPW], pixoff[MPW];
  unsigned vmask = 0;
#pragma unroll
  for (int m = 0; m < MPW; ++m) {
    const int mt = wave * MPW + m;
    {
      const int q = mt * 16 + lp;
      const int ty = q / TW, tx = q - ty * TW;
      basem[m] = ((ty + 1) * P + tx + 1) * 32 + (lg & 1) * 16;
    }
    const int q = mt * 16 + lg * 4;
    const int ty = q / TW, tx = q - ty * TW;
    const int gy = gy0 + ty, gx = gx0 + tx;
    const bool ok = gy < p.H && gx < p.W;  // W % 4 == 0: a 4-pixel group is all-in or all-out
    vmask |= (ok ? 1u : 0u) << m;
    pixoff[m] = ok ? gy * p.W + gx : 0;
  }
  const int j = lp;
  const bool jv = j < p.Cout;
  const int jc = jv ? j : 0;
  float4 pin[MPW];
  {
    const float* pn = p.partial_in + ((long long)n * p.Cout + jc) * ((long long)p.H * p.W);
#pragma unroll
    for (int m = 0; m < MPW; ++m) pin[m] = *reinterpret_cast<const float4*>(pn + pixoff[m]);
  }
  uint4 bf[5][NP];
  {
    const uint4* wp = p.wpk + (long long)p.c_first * 5 * NP * 64 + lane;
#pragma unroll
    for (int s0 = 0; s0 < 5; ++s0)
#pragma unroll
      for (int pt = 0; pt < NP; ++pt) bf[s0][pt] = wp[(s0 * NP + pt) * 64];
  }
  int toff[5];
#pragma unroll
  for (int s0 = 0; s0 < 5; ++s0) {
    const int tap = min(2 * s0 + (lg >> 1), 8);
    toff[s0] = ((tap / 3 - 1) * P + (tap % 3 - 1)) * 32;
  }

  // ---- stage the 16-channel tile: thread -> image cells (16 channels each), BN + ReLU, one 16-bit part ----
  {
    const SP<ST> xn = SP<ST>(p.S) + ((long long)n * p.ns + (long long)cb * p.cs);
    float av[16], bv[16];
#pragma unroll
    for (int cc = 0; cc < 16; ++cc) {
      av[cc] = p.pa[cb + cc];
      bv[cc] = p.pb[cb + cc];
    }
    for (int e = tid; e < ROWS * P; e += 256) {
      const int r = e / P, col = e - r * P;
      const int iy = gy0 - 1 + r, ix = gx0 - 1 + col;
      const bool ok = col <= TW + 1 && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      const int off = ok ? iy * p.W + ix : 0;
      float v[16];
#pragma unroll
      for (int cc = 0; cc < 16; ++cc) v[cc] = xn.ld1((long long)cc * p.cs + off);
      unsigned parts[8][NP];
#pragma unroll
      for (int k = 0; k < 8; ++k)  // cells outside the picture are zero AFTER the activation (the convolution's padding)
        split2<DT, NP>(ok ? relu16<DT>(fmaf(av[2 * k], v[2 * k], bv[2 * k])) : 0.f,
                       ok ? relu16<DT>(fmaf(av[2 * k + 1], v[2 * k + 1], bv[2 * k + 1])) : 0.f, parts[k]);
#pragma unroll
      for (int pt = 0; pt < NP; ++pt) {
        uint4* dst = reinterpret_cast<uint4*>(smem + pt * PLANE + e * 32);
        dst[0] = make_uint4(parts[0][pt], parts[1][pt], parts[2][pt], parts[3][pt]);
        dst[1] = make_uint4(parts[4][pt], parts[5][pt], parts[6][pt], parts[7][pt]);
      }
    }
  }
  __syncthreads();  // image staged

  f32x4 acc[MPW];
#pragma unroll
  for (int m = 0; m < MPW; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    constexpr int DEPTH = 4, RING = DEPTH + 1, STEPS = 5 * MPW;
    uint4 af[RING][NP];
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) {
      const int s0 = i / MPW, m0 = i - s0 * MPW;
#pragma unroll
      for (int pt = 0; pt < NP; ++pt)
        af[i % RING][pt] = *reinterpret_cast<const uint4*>(smem + pt * PLANE + basem[m0] + toff[s0]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < STEPS; ++i) {
      const int s0 = i / MPW, m = i - s0 * MPW;
      if (i + DEPTH < STEPS) {
        const int s1 = (i + DEPTH) / MPW, m1 = (i + DEPTH) - s1 * MPW;
#pragma unroll
        for (int pt = 0; pt < NP; ++pt)
          af[(i + DEPTH) % RING][pt] = *reinterpret_cast<const uint4*>(smem + pt * PLANE + basem[m1] + toff[s1]);
      }
      acc[m] = mfma_split<DT, NP>(af[i % RING], bf[s0], acc[m]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- epilogue: lane holds 4 consecutive pixels of channel j per M-tile (d3_fwd_k's arithmetic) ----
  mfma_drain();  // wait states between the MFMA chain and the first accumulator read (split16.h)
  const float bias = (jv && p.bias) ? p.bias[j] : 0.f;
  const float sc = (jv && p.nscale) ? p.nscale[(long long)n * p.Cout + j] : 1.f;
  const SP<ST> outn = SP<ST>(p.out) + ((long long)n * p.out_ns + (long long)jc * p.out_cs);
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int m = 0; m < MPW; ++m) {
    const bool ok = jv && ((vmask >> m) & 1u);
    const float pv[4] = {pin[m].x, pin[m].y, pin[m].z, pin[m].w};
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[r] = st_round<ST>(fmaf(acc[m][r] + pv[r], w_unscale<DT>(), bias) * sc);  // statistics of the tensor as it is stored
      if (ok) {
        s1 += v[r];
        s2 += v[r] * v[r];
      }
    }
    if (ok) outn.st4(pixoff[m], v[0], v[1], v[2], v[3]);
  }
  if (p.stat_partial != nullptr) {
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
      const long long brow = (long long)n * gridDim.x + bx;
      p.stat_partial[(brow * p.Cout + jj) * 2 + w2] = t;
    }
  }
}

// tiling of the finishing kernel: the 256-pixel shape that wastes fewer pixels (ties -> 8 x 32: longer rows)
static bool d3_fin_tile16(int H, int W) {
  const long long a0 = (long long)((H + 7) / 8) * ((W + 31) / 32), a1 = (long long)((H + 15) / 16) * ((W + 15) / 16);
  return a1 < a0;
}

long long d3_fin_rows(int H, int W, int N) {
  const bool t16 = d3_fin_tile16(H, W);
  const int th = t16 ? 16 : 8, tw = t16 ? 16 : 32;
  return (long long)N * ((H + th - 1) / th) * ((W + tw - 1) / tw);
}

bool d3_fin_supported(const D3Fwd& p, int np) {
  if (np < 1 || np > 2 || p.ksplit > 1 || p.partial_in == nullptr || p.wpk == nullptr) return false;
  if (p.Cout < 1 || p.Cout > 16 || p.c_first < 1 || p.Cin != (p.c_first + 1) * 16) return false;
  if ((p.W & 3) || p.W < 16 || p.H < 4 || (p.cs & 3) || (p.out_cs & 3) || (p.ns & 3) || (p.out_ns & 3)) return false;
  if ((((long long)p.H * p.W) & 3) || (reinterpret_cast<uintptr_t>(p.partial_in) & 15)) return false;
  const uintptr_t amask = p.st == ST_BF16 ? 7 : 15;
  if ((reinterpret_cast<uintptr_t>(p.S) & amask) || (reinterpret_cast<uintptr_t>(p.out) & amask)) return false;
  return true;
}

template <int NP, int DT, int ST, int TH, int TW>
static int d3_fin_launch_t(D3Fwd p, int N, hipStream_t s) {
  constexpr int P = TW + 3, ROWS = TH + 2;
  p.tiles_y = (p.H + TH - 1) / TH;
  p.tiles_x = (p.W + TW - 1) / TW;
  const size_t lds = (size_t)NP * ROWS * P * 32 + 4 * 16 * 2 * 4;
  hipLaunchKernelGGL((d3_fin_k<NP, DT, ST, TH, TW>), dim3((unsigned)(p.tiles_x * p.tiles_y), 1, (unsigned)N), dim3(256), lds, s, p);
  return (int)hipGetLastError();
}

int d3_fin_launch(const D3Fwd& p, int N, int np, int dt, hipStream_t s) {
  if (!d3_fin_supported(p, np)) return -4;
  const bool t16 = d3_fin_tile16(p.H, p.W);
#define D3_FIN(NP_, DT_, ST_) \
  return t16 ? d3_fin_launch_t<NP_, DT_, ST_, 16, 16>(p, N, s) : d3_fin_launch_t<NP_, DT_, ST_, 8, 32>(p, N, s)
  if (p.st == ST_BF16) {
    if (dt != D3_BF16 || np != 1) return -4;
    D3_FIN(1, D3_BF16, ST_BF16);
  }
  if (np == 2) {
    if (dt == D3_BF16) D3_FIN(2, D3_BF16, ST_F32);
    D3_FIN(2, D3_F16, ST_F32);
  }
  if (dt == D3_BF16) D3_FIN(1, D3_BF16, ST_F32);
  D3_FIN(1, D3_F16, ST_F32);
#undef D3_FIN
}

// =============================================================================================
// weight gradient (see dense3.h)
//
// 12 waves: waves 0-3 consume (wave w takes the 32-pixel K-steps w, w+4, w+8 of the 320-pixel tile: per K-step one dY
// fragment and, per tap, one shifted z fragment through transposed LDS reads, 27 MFMAs at 2 parts); waves 4-7 / 8-11
// are two producer groups taking alternate TILES (z chunk with BN+ReLU and halo + the dY tile, both split into 16-bit
// parts), each re-issuing its global loads for tile i+3 right after committing tile i+1.  One barrier per tile.
// =============================================================================================
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 lds_tr16(const unsigned char* p) {
  union { s16x4 v; uint2 u; } c;
  c.v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  return c.u;
}

// (Measured with bf16 stacks and not kept: <= 80 VGPRs + one staging register set + 512 blocks = two workgroups per CU:
// 4.4 -> 5.1 ms per step.  The second register set -- a tile's loads in flight for two iterations -- is worth more than a
// partner workgroup.)
// YT: storage type of the dY the kernel reads.  (ST, YT) = (fp32, bf16) is the default mode's weight gradient: its
// one-part bf16 operand is rounded once by grad_finalize_k into a 2-byte copy (p.dY16) instead of by every one of the
// Cin / 16 chunk blocks that re-read the tile -- the dY share of the staged bytes halves; a dY unit is then 8 channels
// x 8 pixels (one 16-byte load per channel, like every other unit).
// NL = 2: two layers of one block in one launch.  Layer j+1 consumes every input channel of layer j (plus its own 16 new
// ones), each with its own BatchNorm: the z chunk is LOADED once and converted twice (two z images, two dY tiles, two
// sets of tap accumulators).  The z loads are the largest share of this kernel's time (ablation in DESIGN.md 4.1e).
// The primary layer (p.Cin, p.pa, p.dY16, p.partial) is the one with more channels and defines the chunk grid; the
// secondary (p.Cin2 <= p.Cin, p.pa2, p.dY16_2, p.partial2) takes part in the chunks it has.
// Z8 (bf16 stacks): every unit is 8 pixels wide -- 16-byte loads of 8 bf16 pixels for z, dY and the halo alike -- which
// frees the thread slots a second layer's dY tile needs (the 4-pixel form has no room for NL = 2).
template <int NP, int DT, int ST, int YT, int NL, bool Z8>
__global__ __launch_bounds__(768, 3) void d3_wgrad_k(const D3Wgrad p) {
  constexpr bool Y16 = (ST == ST_F32 && YT == ST_BF16);
  constexpr bool B16 = Y16 || Z8;  // byte-addressed 16-byte loads in every thread, 8-pixel dY units
  static_assert(ST == YT || (Y16 && NP == 1 && DT == D3_BF16), "mixed storage: fp32 stacks with a bf16 dY copy, one bf16 part");
  static_assert(!Z8 || (ST == ST_BF16 && YT == ST_BF16 && NP == 1 && DT == D3_BF16), "8-pixel units: bf16 stacks, one bf16 part");
  static_assert(NL == 1 || B16, "two-layer launches use the 8-pixel dY units");
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave >= 4;
  const int P = p.tw + 3;
  const int rows = p.th + 2;
  const int PLZ = rows * P * 32;          // one part of the z image
  const int npix = p.th * p.tw;           // 320
  const int PLY = npix * 32;              // one part of the dY image
  const int ZL = NP * PLZ, YL = NP * PLY;  // one layer's images
  const int ZB = NL * ZL, YB = NL * YL;    // one buffer each
  unsigned char* zbuf = smem;             // [2][NL][NP][PLZ]
  unsigned char* ybuf = smem + 2 * ZB;    // [2][NL][NP][PLY]
  float* abtab = reinterpret_cast<float*>(smem + 2 * ZB + 2 * YB);  // per layer: a[16], b[16] of this chunk

  const unsigned lg_id = xcd_logical_block(blockIdx.x, gridDim.x);
  const int chunk = (int)(lg_id % (unsigned)p.nchunks), range = (int)(lg_id / (unsigned)p.nchunks);
  const int cb = min(chunk * 16, max(p.Cin - 16, 0));
  const int tiles = p.tiles_x * p.tiles_y;
  const int total = tiles * p.N;
  const int per = (total + p.nranges - 1) / p.nranges;
  const int t0 = range * per;
  const int nt = max(0, min(total, t0 + per) - t0);

  const bool sec_active = NL == 2 && chunk * 16 < p.Cin2;  // the secondary layer has this chunk (block-uniform)
  if (tid < 32 * NL) {
    const int L = tid >> 5, t = tid & 31;
    const int ch = cb + (t & 15);
    const float* ta = L ? p.pa2 : p.pa;
    const float* tb = L ? p.pb2 : p.pb;
    abtab[tid] = ch < (L ? p.Cin2 : p.Cin) ? (t < 16 ? ta[ch] : tb[ch]) : 0.f;
  }
  if (p.tiles_x == 1) {  // full-width tiles (the 15x20 level: a whole sample per tile): zero padding columns, written once
    for (int i = tid; i < 2 * NL * NP * rows * 2; i += 768) {
      const int side = i & 1, r = (i >> 1) % rows, bp = (i >> 1) / rows;  // bp = (buffer * NL + layer) * NP + part
      uint4* z = reinterpret_cast<uint4*>(zbuf + bp * PLZ + (r * P + (side ? p.tw + 1 : 0)) * 32);
      z[0] = make_uint4(0u, 0u, 0u, 0u);
      z[1] = make_uint4(0u, 0u, 0u, 0u);
    }
  }

  f32x4 acc[NL][9];
#pragma unroll
  for (int L = 0; L < NL; ++L)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[L][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (producer) {
    // All 8 producer waves work on EVERY tile: thread -> one staging unit (8 channels x 4 pixels, or a halo pair):
    //   ptid   0..255  z unit      (2 octets x RG rows x quads; 240 used for the 4x80 tile)
    //   ptid 256..415  dY unit     (2 octets x 4 rows x 20 quads)
    //   ptid 416..511  z halo unit (row, side, channel quad)
    // Two register sets hold the tiles of even / odd iteration, so a tile's loads stay in flight for two iterations.
    const int ptid = tid - 256;
    const int nq = p.tw >> 2;
    const int rgs = p.rg == 2 ? 1 : 2;
    const int qpg = 4 >> rgs;
    const int nqg = (nq + qpg - 1) / qpg;
    const int nrg = (rows + p.rg - 1) >> rgs;
    const int kind = ptid < 256 ? 0 : (ptid < 416 ? 1 : 2);
    int u_r = 0, u_q = 0, u_o = 0, u_lds = 0, h_side = 0, y_layer = 0;
    bool u_ex = false;
    if (kind == 0 && Z8) {  // z unit = (octet, row, 8-pixel group)
      const int lu = ptid, nq8 = p.tw >> 3;
      u_o = lu & 1;
      u_r = (lu >> 1) / nq8;
      u_q = (lu >> 1) - u_r * nq8;
      u_ex = u_r < rows;
      u_lds = ((u_ex ? u_r : 0) * P + 1 + 8 * (u_ex ? u_q : 0)) * 32 + u_o * 16;
    } else if (kind == 0) {
      const int lu = ptid;
      u_o = lu & 1;
      const int rr = (lu >> 1) & (p.rg - 1), qq = (lu >> (1 + rgs)) & (qpg - 1), u = lu >> 3;
      const int R = u / nqg;
      u_q = (u - R * nqg) * qpg + qq;
      u_r = (R << rgs) + rr;
      u_ex = (R < nrg) && (u_r < rows) && (u_q < nq);
      u_lds = ((u_ex ? u_r : 0) * P + 1 + 4 * (u_ex ? u_q : 0)) * 32 + u_o * 16;
    } else if (kind == 1) {
      int yu = ptid - 256;
      const int nqy = B16 ? (p.tw >> 3) : nq;  // units per row: 8 pixels each with a 2-byte dY
      if constexpr (NL == 2) {  // the second layer's dY units follow the first layer's
        const int nyu = 2 * p.th * nqy;
        y_layer = yu >= nyu ? 1 : 0;
        yu -= y_layer * nyu;
      }
      u_o = yu & 1;
      const int y2 = yu >> 1;
      u_r = y2 / nqy;
      u_q = y2 - u_r * nqy;
      u_ex = u_r < p.th;
      u_lds = y_layer * YL + ((u_ex ? u_r : 0) * p.tw + (B16 ? 8 : 4) * u_q) * 32 + u_o * 16;
    } else {
      const int hu = ptid - 416;
      u_o = hu & 3;  // channel quad
      h_side = (hu >> 2) & 1;
      u_r = hu >> 3;
      u_ex = u_r < rows && p.tiles_x > 1;  // full-width tiles: both halo columns lie outside the picture (zeroed below)
      u_lds = ((u_ex ? u_r : 0) * P + (h_side ? p.tw + 1 : 0)) * 32 + u_o * 8;
    }
    // Every thread issues the SAME 8 dwordx4 loads per tile from (kind base) + choff[cc] + per-tile pixel offset: no
    // branches around the loads, so the wait counters stay exact (a join of paths with different load counts makes the
    // compiler wait for the youngest loads too, which would halve the prefetch distance).
    int choff[8];
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) {
      const int ch = kind == 0 ? cb + u_o * 8 + cc : (kind == 1 ? min(u_o * 8 + cc, p.Cout - 1) : cb + 4 * u_o + (cc & 3));
      choff[cc] = ch * p.cs;
    }
    // every unit loads 8 x Raw4 (Y16: a dY unit's 16 bytes hold 8 bf16 pixels; Z8: every unit's)
    // (Z8: a native vector type -- arrays of HIP's uint4 struct in a register set that crosses the barrier were kept as
    // stack objects, and every scratch reload is a vmcnt(0) that drains the tile loads in flight: see d3_fwd_k)
    typedef typename std::conditional<Z8, u32x4, typename SRaw<ST>::r4>::type Raw4;
    const SP<ST> kbase(kind == 1 ? p.dY : p.S);
    const long long kns = kind == 1 ? (long long)p.Cout * p.cs : p.ns;
    const unsigned char* kbytes =
        reinterpret_cast<const unsigned char*>(kind == 1 ? (y_layer ? p.dY16_2 : p.dY16) : (const void*)p.S);
    const int kes = (Z8 || (Y16 && kind == 1)) ? 2 : 4;  // bytes per element of this thread's operand
    Raw4 regA[8], regB[8];
    bool okA = false, okB = false;
    // tiles are requested strictly in order t0, t0+1, ...: a cursor replaces two integer divisions per request
    int cur_n = t0 / tiles, cur_ty = (t0 - cur_n * tiles) / p.tiles_x, cur_tx = (t0 - cur_n * tiles) - cur_ty * p.tiles_x;
    auto issue = [&](int /*t: the cursor's tile*/, Raw4 (&reg)[8], bool& okf) __attribute__((always_inline)) {
      const int n = min(cur_n, p.N - 1);  // (a request past the last tile of the launch stays inside the tensors)
      const int gy0 = cur_ty * p.th, gx0 = cur_tx * p.tw;
      if (++cur_tx == p.tiles_x) {
        cur_tx = 0;
        if (++cur_ty == p.tiles_y) {
          cur_ty = 0;
          ++cur_n;
        }
      }
      int iy, ix;
      bool ok;
      if (kind == 0) {
        iy = gy0 - 1 + u_r;
        ix = gx0 + (Z8 ? 8 : 4) * u_q;
        ok = u_ex && iy >= 0 && iy < p.H && ix < p.W;
      } else if (kind == 1) {
        iy = gy0 + u_r;
        ix = gx0 + (B16 ? 8 : 4) * u_q;
        ok = u_ex && iy < p.H && ix < p.W;
      } else {  // halo pixel: the aligned quad that holds it (left: last element of the quad, right: first)
        iy = gy0 - 1 + u_r;
        const int hx = h_side ? gx0 + p.tw : gx0 - 1;
        ok = u_ex && iy >= 0 && iy < p.H && hx >= 0 && hx < p.W;
        ix = h_side ? hx : hx - (Z8 ? 7 : 3);  // the aligned group that holds it (left: its last element, right: its first)
      }
      if constexpr (B16) {  // one byte-addressed form for both element sizes: the same 8 dwordx4 loads in every thread
        const unsigned char* srcb = kbytes + ((long long)n * kns + (ok ? iy * p.W + ix : 0)) * kes;
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) reg[cc] = *reinterpret_cast<const Raw4*>(srcb + (long long)choff[cc] * kes);
      } else {
        const SP<ST> src = kbase + ((long long)n * kns + (ok ? iy * p.W + ix : 0));
#ifdef RLN_DIAG
        if (!(p.dbg & 1) && !((p.dbg & 8) && kind == 1) && !((p.dbg & 16) && kind != 1))
#endif
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) reg[cc] = src.raw4(choff[cc]);
      }
      okf = ok;
    };
    auto commit = [&](int buf, const Raw4 (&rawreg)[8], bool okf) __attribute__((always_inline)) {
      // Every thread "uses" the set's last load before the unit kinds branch: one exact vmcnt(8) on the common path.
      // Without it the threads that own no unit never wait for their (unconditional) loads, the join leaves the set
      // pending, and the refill below is protected by a vmcnt(0) that drains the OTHER set's loads in every iteration.
      touch_reg(rawreg[7]);
      float4 reg[8];
      if constexpr (!Z8) {
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) reg[cc] = SRaw<ST>::w4(rawreg[cc]);
      }
      auto bits = [](auto v) __attribute__((always_inline)) { return __builtin_bit_cast(unsigned, v); };
      unsigned char* zb = zbuf + buf * ZB;
      unsigned char* yb = ybuf + buf * YB;
      if (!u_ex) return;
#ifdef RLN_DIAG
      if (p.dbg & 2) return;
#endif
      if (kind == 2) {
#pragma unroll
        for (int L = 0; L < NL; ++L) {
          if (L == 1 && !sec_active) break;  // block-uniform
          const float* ab = abtab + 32 * L;
          float zv[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            float x;
            if constexpr (Z8) x = h_side ? bf16_lo(bits(rawreg[k].x)) : bf16_hi(bits(rawreg[k].w));
            else x = h_side ? reg[k].x : reg[k].w;
            zv[k] = okf ? fmaxf(fmaf(ab[4 * u_o + k], x, ab[16 + 4 * u_o + k]), 0.f) : 0.f;
          }
          unsigned pa[NP], pb2[NP];
          split2<DT, NP>(zv[0], zv[1], pa);
          split2<DT, NP>(zv[2], zv[3], pb2);
#pragma unroll
          for (int pt = 0; pt < NP; ++pt)
            *reinterpret_cast<uint2*>(zb + L * ZL + pt * PLZ + u_lds) = make_uint2(pa[pt], pb2[pt]);
        }
        return;
      }
      if constexpr (B16) {
        if (kind == 1) {  // 8 channels x 8 bf16 pixels, already rounded: interleave the channels per pixel
          unsigned wv[8][4];
#pragma unroll
          for (int cc = 0; cc < 8; ++cc) {
            const bool cv = okf && (u_o * 8 + cc < p.Cout);
            wv[cc][0] = cv ? bits(rawreg[cc].x) : 0u;
            wv[cc][1] = cv ? bits(rawreg[cc].y) : 0u;
            wv[cc][2] = cv ? bits(rawreg[cc].z) : 0u;
            wv[cc][3] = cv ? bits(rawreg[cc].w) : 0u;
          }
#pragma unroll
          for (int px = 0; px < 8; ++px) {
            unsigned o4[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const unsigned a = wv[2 * k][px >> 1], b = wv[2 * k + 1][px >> 1];
              o4[k] = (px & 1) ? ((a >> 16) | (b & 0xffff0000u)) : ((a & 0xffffu) | (b << 16));
            }
            *reinterpret_cast<uint4*>(yb + u_lds + px * 32) = make_uint4(o4[0], o4[1], o4[2], o4[3]);
          }
          return;
        }
      }
      if constexpr (Z8) {  // kind 0: z = relu(a*x + b) of 8 channels x 8 bf16 pixels, once per layer of the launch
#pragma unroll 1  // (a real loop: unrolled, both layers' tables and parts are live at once -- 35 VGPRs over the budget)
        for (int L = 0; L < NL; ++L) {
          if (L == 1 && !sec_active) break;
          const float* ab = abtab + 32 * L;
          float av[8], bv[8];
#pragma unroll
          for (int cc = 0; cc < 8; ++cc) {
            av[cc] = ab[u_o * 8 + cc];
            bv[cc] = ab[16 + u_o * 8 + cc];
          }
          unsigned char* dst = zb + L * ZL + u_lds;
#pragma unroll
          for (int px = 0; px < 8; ++px) {
            unsigned parts[4][NP];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const auto& r0 = rawreg[2 * k];
              const auto& r1 = rawreg[2 * k + 1];
              const float x0 = SRaw<ST_BF16>::q(make_uint4(r0.x, r0.y, r0.z, r0.w), px);
              const float x1 = SRaw<ST_BF16>::q(make_uint4(r1.x, r1.y, r1.z, r1.w), px);
              const float z0 = fmaxf(fmaf(av[2 * k], x0, bv[2 * k]), 0.f), z1 = fmaxf(fmaf(av[2 * k + 1], x1, bv[2 * k + 1]), 0.f);
              split2<DT, NP>(okf ? z0 : 0.f, okf ? z1 : 0.f, parts[k]);
            }
            *reinterpret_cast<uint4*>(dst + px * 32) = make_uint4(parts[0][0], parts[1][0], parts[2][0], parts[3][0]);
          }
        }
        return;
      }
      // kind 0: z = relu(a*x + b) of 8 channels x 4 pixels; kind 1: dY as it is (zero beyond Cout); zero outside
#pragma unroll
      for (int L = 0; L < NL; ++L) {
      if (L == 1 && (!sec_active || kind != 0)) break;  // the second conversion of the same z chunk (block-uniform / kind 0)
      const float* ab = abtab + 32 * L;
      float av[8], bv[8];
#pragma unroll
      for (int cc = 0; cc < 8; ++cc) {
        av[cc] = kind == 0 ? ab[u_o * 8 + cc] : ((u_o * 8 + cc < p.Cout) ? 1.f : 0.f);
        bv[cc] = kind == 0 ? ab[16 + u_o * 8 + cc] : 0.f;
      }
      unsigned char* dst = (kind == 0 ? zb + L * ZL : yb) + u_lds;
      const int plane = kind == 0 ? PLZ : PLY;
#pragma unroll
      for (int px = 0; px < 4; ++px) {
        unsigned parts[4][NP];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float4 u0 = reg[2 * k], u1 = reg[2 * k + 1];
          const float x0 = px == 0 ? u0.x : px == 1 ? u0.y : px == 2 ? u0.z : u0.w;
          const float x1 = px == 0 ? u1.x : px == 1 ? u1.y : px == 2 ? u1.z : u1.w;
          float z0 = fmaf(av[2 * k], x0, bv[2 * k]), z1 = fmaf(av[2 * k + 1], x1, bv[2 * k + 1]);
          if (kind == 0) {
            z0 = fmaxf(z0, 0.f);
            z1 = fmaxf(z1, 0.f);
          }
          split2<DT, NP>(okf ? z0 : 0.f, okf ? z1 : 0.f, parts[k]);
        }
#pragma unroll
        for (int pt = 0; pt < NP; ++pt)
          *reinterpret_cast<uint4*>(dst + pt * plane + px * 32) =
              make_uint4(parts[0][pt], parts[1][pt], parts[2][pt], parts[3][pt]);
      }
      }
    };
    __syncthreads();  // affine table
    {
      // tile t0+i lives in set i & 1.  Iteration i (consumers multiply tile i): commit tile i+1, refill its set with i+3.
      // The prologue loads are unconditional (a request past the block's range reads a clamped, in-range tile nobody
      // commits): conditional issues make the steady loop a join of paths with different numbers of loads in flight, and
      // the wait-count pass then drains everything at the loop top (see d3_fwd_k).
      if (nt <= 0) {
        __syncthreads();
      } else {
      issue(t0, regA, okA);
      issue(t0 + 1, regB, okB);
      commit(0, regA, okA);
      issue(t0 + 2, regA, okA);
      __syncthreads();  // first tile staged
      int i = 0;
      for (; i + 4 < nt; i += 2) {  // steady state, two iterations per trip, no branches around the loads
        commit(1, regB, okB);
        issue(t0 + i + 3, regB, okB);
        __syncthreads();
        commit(0, regA, okA);
        issue(t0 + i + 4, regA, okA);
        __syncthreads();
      }
      for (; i < nt; ++i) {
        if (i + 1 < nt) {
          if ((i + 1) & 1) commit(1, regB, okB); else commit(0, regA, okA);
          if (i + 3 < nt) {
            if ((i + 3) & 1) issue(t0 + i + 3, regB, okB); else issue(t0 + i + 3, regA, okA);
          }
        }
        __syncthreads();
      }
      }
    }
  } else {
    // =========================== consumer waves ===========================
    // transposed-read addressing: lane 4q+pp of 16-lane group g supplies pixel record (8g + q [+4]), bytes 8pp..8pp+7
    // and receives channel (lane & 15) of those 4 pixels.  Odd groups take their two 4-pixel blocks in swapped order
    // (bank-conflict-free halves); both operands use the same order, so the k pairing inside the MFMA is unchanged.
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    constexpr int KS = 3;  // K-steps per wave (tile of 320 pixels = 10 K-steps; waves 0,1 take 3, waves 2,3 take 2)
    int ya[KS][2], za[KS][2];
    int nks = 0;
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      const int ks = wave + 4 * k;
      if (ks * 32 < npix) nks = k + 1;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int blk = (h ^ (g & 1)) * 4;  // which 4-pixel block of the lane group's 8 pixels
        const int pix = min(ks * 32 + 8 * g + blk + q, npix - 1);
        ya[k][h] = pix * 32 + 8 * pp;
        const int ty = pix / p.tw, tx = pix - ty * p.tw;
        za[k][h] = ((ty + 1) * P + tx + 1) * 32 + 8 * pp;
      }
    }
    __syncthreads();  // affine table
    __syncthreads();  // first tile staged
    __builtin_amdgcn_s_setprio(2);  // the matrix waves win issue arbitration against the two staging waves of their SIMD
    for (int i = 0; i < nt; ++i) {
      const unsigned char* zb = zbuf + (i & 1) * ZB;
      const unsigned char* yb = ybuf + (i & 1) * YB;
      // one straight sequence of KS x 9 (K-step, tap) steps with a DEPTH-deep read-ahead ring across K-step boundaries;
      // a K-step a wave does not own runs with a zero dY fragment (adds nothing) so that all four waves share one code path
      constexpr int DEPTH = 3, RING = DEPTH + 1;
      const int STEPS = nks * 9;  // wave-uniform: 27 (waves 0, 1) or 18 (waves 2, 3)
      uint4 af[2][NL][NP], bfr[RING][NL][NP];
      auto load_a = [&](int k, uint4 (&dst)[NL][NP]) __attribute__((always_inline)) {
#pragma unroll
        for (int L = 0; L < NL; ++L)
#pragma unroll
          for (int pt = 0; pt < NP; ++pt) {
            const uint2 lo = lds_tr16(yb + L * YL + pt * PLY + ya[k][0]), hi = lds_tr16(yb + L * YL + pt * PLY + ya[k][1]);
            dst[L][pt] = make_uint4(lo.x, lo.y, hi.x, hi.y);
          }
      };
      auto load_b = [&](int st, uint4 (&dst)[NL][NP]) __attribute__((always_inline)) {
        const int k = st / 9, t = st - k * 9;
        const int toff = ((t / 3 - 1) * P + (t % 3 - 1)) * 32;
#pragma unroll
        for (int L = 0; L < NL; ++L)
#pragma unroll
          for (int pt = 0; pt < NP; ++pt) {
            const uint2 lo = lds_tr16(zb + L * ZL + pt * PLZ + za[k][0] + toff),
                        hi = lds_tr16(zb + L * ZL + pt * PLZ + za[k][1] + toff);
            dst[L][pt] = make_uint4(lo.x, lo.y, hi.x, hi.y);
          }
      };
      load_a(0, af[0]);
#pragma unroll
      for (int st = 0; st < DEPTH; ++st) load_b(st, bfr[st % RING]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int st = 0; st < KS * 9; ++st) {
        const int k = st / 9, t = st - k * 9;
#ifdef RLN_DIAG
        if (p.dbg & 4) continue;
#endif
        if (st < STEPS) {  // wave-uniform
          if (st + DEPTH < KS * 9) load_b(st + DEPTH, bfr[(st + DEPTH) % RING]);  // past the wave's steps: clamped, unused
          if (t == 9 - DEPTH && k + 1 < KS) load_a(k + 1, af[(k + 1) & 1]);
#pragma unroll
          for (int L = 0; L < NL; ++L) acc[L][t] = mfma_split<DT, NP>(af[k & 1][L], bfr[st % RING][L], acc[L][t]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
    }
  }

  // ---- reduce the four consumer waves' tap accumulators through LDS and write this block's slab piece ----
  __syncthreads();
  constexpr int NT = 9 * NL;
  float4* red = reinterpret_cast<float4*>(smem);  // [4 waves][NL x 9 taps][64 lanes]
  if (!producer) {
    mfma_drain();  // wait states between the MFMA chain and the first accumulator read (split16.h)
#pragma unroll
    for (int L = 0; L < NL; ++L)
#pragma unroll
      for (int t = 0; t < 9; ++t)
        red[(wave * NT + L * 9 + t) * 64 + lane] = make_float4(acc[L][t][0], acc[L][t][1], acc[L][t][2], acc[L][t][3]);
  }
  __syncthreads();
  for (int e = tid; e < NT * 64; e += 768) {
    const int lt = e >> 6, l = e & 63;
    const int L = lt / 9, t = lt - L * 9;
    if (L == 1 && !sec_active) continue;
    const float4 a0 = red[(0 * NT + lt) * 64 + l], a1 = red[(1 * NT + lt) * 64 + l], a2 = red[(2 * NT + lt) * 64 + l],
                 a3 = red[(3 * NT + lt) * 64 + l];
    const float v[4] = {a0.x + a1.x + a2.x + a3.x, a0.y + a1.y + a2.y + a3.y, a0.z + a1.z + a2.z + a3.z,
                        a0.w + a1.w + a2.w + a3.w};
    const int c = cb + (l & 15);  // D[row = 4*(l>>4) + r = output channel o][col = l & 15 = input channel]
    const int cin = L ? p.Cin2 : p.Cin;
    if (c < cin && c >= chunk * 16) {
      float* dst = (L ? p.partial2 : p.partial) + (long long)range * p.Cout * cin * 9;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = 4 * (l >> 4) + r;
        if (o < p.Cout) dst[((long long)o * cin + c) * 9 + t] = v[r];
      }
    }
  }
}

bool d3_wgrad_supported(const D3Wgrad& p) {
  if (p.Cout < 1 || p.Cout > 16 || p.Cin < 1) return false;
  if (((p.W % 40) != 0 && !(p.W == 20 && p.H <= 16)) || p.H < 4) return false;  // 15x20 level: one sample per tile
  if ((p.cs & 3) || (p.ns & 3)) return false;
  const uintptr_t amask = p.st == ST_BF16 ? 7 : 15;
  if ((reinterpret_cast<uintptr_t>(p.S) & amask) || (reinterpret_cast<uintptr_t>(p.dY) & amask)) return false;
  return true;
}

void d3_wgrad_plan(int H, int W, int N, int Cin, D3Wgrad* p) {
  if (W == 20) {  // 16 x 20: the whole 15 x 20 sample (rows past H carry zero gradient)
    p->th = 16;
    p->tw = 20;
  } else if (W % 80 == 0) {  // 320-pixel tiles: 4 x 80, or 8 x 40 where a row is not a multiple of 80
    p->th = 4;
    p->tw = 80;
  } else {
    p->th = 8;
    p->tw = 40;
  }
  p->tiles_y = (H + p->th - 1) / p->th;
  p->tiles_x = (W + p->tw - 1) / p->tw;
  p->rg = 2;  // staged rows in bands of 2: 240 (4x80) / 200 (8x40) lane-units, one per z thread
  p->nchunks = (Cin + 15) / 16;
  const long long total = (long long)p->tiles_x * p->tiles_y * N;
  long long nr = std::max<long long>(1, 256 / p->nchunks);
  nr = std::min(nr, total);
  const long long per = (total + nr - 1) / nr;
  p->nranges = (int)((total + per - 1) / per);  // no empty ranges
}

template <int NP, int DT, int ST = ST_F32, int YT = ST, int NL = 1, bool Z8 = false>
static int d3_wgrad_launch_t(const D3Wgrad& p, hipStream_t s) {
  const int P = p.tw + 3, rows = p.th + 2;
  const size_t lds = (size_t)2 * NL * NP * rows * P * 32 + (size_t)2 * NL * NP * p.th * p.tw * 32 + 128 * NL;
  if (lds > 160 * 1024 || lds < (size_t)4 * 9 * NL * 64 * 16) return -4;
  auto kern = d3_wgrad_k<NP, DT, ST, YT, NL, Z8>;
  static DevOnce attr_once;
  if (attr_once.first()) {
    const hipError_t attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    if (attr_err != hipSuccess) {  // refused: report it here instead of an opaque launch failure later
      (void)hipGetLastError();
      attr_once.undo();
      return (int)attr_err;
    }
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(p.nchunks * p.nranges)), dim3(768), lds, s, p);
  return (int)hipGetLastError();
}

int d3_wgrad_launch(const D3Wgrad& p, int np, int dt, hipStream_t s) {
  if (!d3_wgrad_supported(p) || p.th * p.tw != 320 || (p.tw != 80 && p.tw != 40 && p.tw != 20) || p.rg != 2) return -4;
  // halo / dY / z unit budgets of the producer threads (full-width tiles have no halo units)
  if ((p.tiles_x > 1 && (p.th + 2) * 8 > 96) || p.th * (p.tw / 4) * 2 > 160) return -4;
  if (p.tw == 20 && p.tiles_x != 1) return -4;
  {
    const int nq = p.tw / 4, nqg = (nq + 1) / 2, nrg = (p.th + 2 + 1) / 2;
    if (8 * nrg * nqg > 256) return -4;
  }
  if (p.st == ST_BF16) {
    if (np != 1 || dt != D3_BF16) return -4;
    if (p.z8) {  // 8-pixel units everywhere (see d3_wgrad_k): needed for, and used with, two-layer launches
      const int rows = p.th + 2, nq8 = p.tw >> 3;
      if ((p.tw & 7) || (p.W & 7) || 2 * rows * nq8 > 256 || rows * 8 > 96 || p.dY16 == nullptr ||
          (reinterpret_cast<uintptr_t>(p.S) & 15) || (reinterpret_cast<uintptr_t>(p.dY16) & 15) ||
          (((long long)p.cs * 2) & 15) || ((p.ns * 2) & 15) || p.nl * 2 * p.th * nq8 > 160)
        return -4;
      if (p.nl == 2) {
        if (p.dY16_2 == nullptr || p.partial2 == nullptr || p.pa2 == nullptr || p.pb2 == nullptr || p.Cin2 < 16 ||
            p.Cin2 > p.Cin || (p.Cin & 15) || (p.Cin2 & 15) || (reinterpret_cast<uintptr_t>(p.dY16_2) & 15))
          return -4;
        return d3_wgrad_launch_t<1, D3_BF16, ST_BF16, ST_BF16, 2, true>(p, s);
      }
      return d3_wgrad_launch_t<1, D3_BF16, ST_BF16, ST_BF16, 1, true>(p, s);
    }
    if (p.nl == 2) return -4;
    return d3_wgrad_launch_t<1, D3_BF16, ST_BF16>(p, s);
  }
  if (p.yt == ST_BF16) {  // fp32 stacks + the bf16 copy of dY (one bf16 part; 8-pixel dY units)
    if (np != 1 || dt != D3_BF16 || p.dY16 == nullptr || (p.tw & 7) || (reinterpret_cast<uintptr_t>(p.dY16) & 15) ||
        (((long long)p.cs * 2) & 15) || p.th * (p.tw >> 3) * 2 > 160)
      return -4;
    if (p.nl == 2) {  // two layers of a block in one launch (see d3_wgrad_k)
      if (p.dY16_2 == nullptr || p.partial2 == nullptr || p.pa2 == nullptr || p.pb2 == nullptr || p.Cin2 < 16 ||
          p.Cin2 > p.Cin || (p.Cin & 15) || (p.Cin2 & 15) || (reinterpret_cast<uintptr_t>(p.dY16_2) & 15) ||
          p.th * (p.tw >> 3) * 4 > 160)
        return -4;
      return d3_wgrad_launch_t<1, D3_BF16, ST_F32, ST_BF16, 2>(p, s);
    }
    return d3_wgrad_launch_t<1, D3_BF16, ST_F32, ST_BF16>(p, s);
  }
  if (p.nl == 2) return -4;
  if (dt == D3_BF16) {
    if (np == 1) return d3_wgrad_launch_t<1, D3_BF16>(p, s);
    if (np == 2) return d3_wgrad_launch_t<2, D3_BF16>(p, s);
    return d3_wgrad_launch_t<3, D3_BF16>(p, s);
  }
  if (np == 1) return d3_wgrad_launch_t<1, D3_F16>(p, s);
  return d3_wgrad_launch_t<2, D3_F16>(p, s);
}

// =============================================================================================
// data gradient into a block's own new channels, looped over 16-channel output groups (see dense3.h)
//
// Block = 4 waves, one TH x TW pixel tile (256 pixels = 16 M-tiles, 4 per wave) of one sample.  The dY tile (+1 halo) is
// staged once as split 16-bit [pixel][16 channels] images; per output group: 5 K-steps x 4 M-tiles of MFMAs per wave
// (the A fragments read DEPTH steps ahead), the group's weight fragments refilled in place with the next group's as
// each K-step retires, S / G of the group prefetched before the MFMAs; epilogue = ReLU mask, BatchNorm-backward sums
// (per-wave LDS slots, summed in fixed order at the end), G (+)= gamma * gz as one 16-byte access per lane.
// =============================================================================================
template <int NP, int DT, int ST, int TH, int TW>
__global__ __launch_bounds__(256, 3) void d3_dgl_k(const D3Dgl p) {
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int MPW = TH * TW / 64;
  static_assert(TW % 16 == 0 && (TH * TW) % 64 == 0 && (MPW * 16) % TW == 0 || TW == 16, "whole M-tiles per wave");
  constexpr int P = TW + 3, ROWS = TH + 2, PLANE = ROWS * P * 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lp = lane & 15, lg = lane >> 4;
  const int bx = blockIdx.x;
  const int tile_y = bx / p.tiles_x, tile_x = bx - tile_y * p.tiles_x;
  const int gy0 = tile_y * TH, gx0 = tile_x * TW;
  const int n = blockIdx.z;
  const int nct = (p.J + 15) >> 4, nct16 = nct * 16;
  float* red = reinterpret_cast<float*>(smem + NP * PLANE);  // [4 waves][nct16][2]

  // ---- stage the dY tile: thread -> one image cell (16 channels), clamped unconditional loads + select ----
  {
    const SP<ST> dn = SP<ST>(p.dY) + (long long)n * p.K * p.cs;
    const int kmax = p.K - 1;
    for (int e = tid; e < ROWS * P; e += 256) {
      const int r = e / P, col = e - r * P;
      const int iy = gy0 - 1 + r, ix = gx0 - 1 + col;
      const bool ok = col <= TW + 1 && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      const int off = ok ? iy * p.W + ix : 0;
      float v[16];
#pragma unroll
      for (int cc = 0; cc < 16; ++cc) v[cc] = dn.ld1((long long)min(cc, kmax) * p.cs + off);
      unsigned parts[8][NP];
#pragma unroll
      for (int k = 0; k < 8; ++k)
        split2<DT, NP>((ok && 2 * k <= kmax) ? v[2 * k] : 0.f, (ok && 2 * k + 1 <= kmax) ? v[2 * k + 1] : 0.f, parts[k]);
#pragma unroll
      for (int pt = 0; pt < NP; ++pt) {
        uint4* dst = reinterpret_cast<uint4*>(smem + pt * PLANE + e * 32);
        dst[0] = make_uint4(parts[0][pt], parts[1][pt], parts[2][pt], parts[3][pt]);
        dst[1] = make_uint4(parts[4][pt], parts[5][pt], parts[6][pt], parts[7][pt]);
      }
    }
  }
  // ---- per-lane geometry ----
  int basem[MPW], pixoff[MPW];
  unsigned vmask = 0;
#pragma unroll
  for (int m = 0; m < MPW; ++m) {
    const int mt = wave * MPW + m;
    {
      const int q = mt * 16 + lp;
      const int ty = q / TW, tx = q - ty * TW;
      basem[m] = ((ty + 1) * P + tx + 1) * 32 + (lg & 1) * 16;
    }
    const int q = mt * 16 + lg * 4;
    const int ty = q / TW, tx = q - ty * TW;
    const int gy = gy0 + ty, gx = gx0 + tx;
    const bool ok = gy < p.H && gx < p.W;  // W % 4 == 0: a 4-pixel group is all-in or all-out
    vmask |= (ok ? 1u : 0u) << m;
    pixoff[m] = ok ? gy * p.W + gx : 0;
  }
  int toff[5];
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    const int tap = min(2 * s + (lg >> 1), 8);
    toff[s] = ((tap / 3 - 1) * P + (tap % 3 - 1)) * 32;
  }
  const SP<ST> Sn = SP<ST>(p.S) + (long long)n * p.s_ns;
  float* Gn = p.G + (long long)n * p.s_ns;

  // output groups of this block (small grids split them over blockIdx.y: disjoint outputs, no reduction)
  const int ct_per = (nct + (int)gridDim.y - 1) / (int)gridDim.y;
  const int ct_begin = (int)blockIdx.y * ct_per;
  const int ct_end = min(nct, ct_begin + ct_per);
  uint4 bf[5][NP];
  if (ct_begin < ct_end) {
    const uint4* wp = p.wpk + ((long long)ct_begin * 5 * NP) * 64 + lane;
#pragma unroll
    for (int s0 = 0; s0 < 5; ++s0)
#pragma unroll
      for (int pt = 0; pt < NP; ++pt) bf[s0][pt] = wp[(s0 * NP + pt) * 64];
  }
  __syncthreads();  // image staged

#pragma unroll 1
  for (int ct = ct_begin; ct < ct_end; ++ct) {
    const int j = ct * 16 + lp;
    const bool jv = j < p.J;
    const int jc = min(j, p.J - 1);
    const bool accum = (j >= p.acc_lo) && (j < p.acc_hi);
    const SP<ST> Sc = Sn + (long long)jc * p.cs;
    float* Gc = Gn + (long long)jc * p.cs;
    float4 sv[MPW], gv[MPW];
#pragma unroll
    for (int m = 0; m < MPW; ++m) {
      sv[m] = Sc.ld4(pixoff[m]);
      gv[m] = *reinterpret_cast<const float4*>(Gc + pixoff[m]);
    }
    const float ea = p.ea[jc], eb = p.eb[jc], emean = p.emean[jc], einv = p.einvstd[jc], egam = p.egamma[jc];
    const uint4* wpn = p.wpk + ((long long)min(ct + 1, ct_end - 1) * 5 * NP) * 64 + lane;  // next group's fragments

    f32x4 acc[MPW];
#pragma unroll
    for (int m = 0; m < MPW; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int DEPTH = 4, RING = DEPTH + 1, STEPS = 5 * MPW;
    uint4 af[RING][NP];
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) {
      const int s0 = i / MPW, m0 = i - s0 * MPW;
#pragma unroll
      for (int pt = 0; pt < NP; ++pt)
        af[i % RING][pt] = *reinterpret_cast<const uint4*>(smem + pt * PLANE + basem[m0] + toff[s0]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < STEPS; ++i) {
      const int s = i / MPW, m = i - s * MPW;
      if (i + DEPTH < STEPS) {
        const int s1 = (i + DEPTH) / MPW, m1 = (i + DEPTH) - s1 * MPW;
#pragma unroll
        for (int pt = 0; pt < NP; ++pt)
          af[(i + DEPTH) % RING][pt] = *reinterpret_cast<const uint4*>(smem + pt * PLANE + basem[m1] + toff[s1]);
      }
      acc[m] = mfma_split<DT, NP>(af[i % RING], bf[s], acc[m]);
      if (m == MPW - 1) {
#pragma unroll
        for (int pt = 0; pt < NP; ++pt) bf[s][pt] = wpn[(s * NP + pt) * 64];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- epilogue: lane holds 4 consecutive pixels of channel j per M-tile ----
    mfma_drain();  // wait states between the MFMA chain and the first accumulator read (split16.h)
    float s1 = 0.f, s2 = 0.f;
    float4 ov[MPW];
#pragma unroll
    for (int m = 0; m < MPW; ++m) {
      const bool ok = jv && ((vmask >> m) & 1u);
      const float xs[4] = {sv[m].x, sv[m].y, sv[m].z, sv[m].w};
      const float gs[4] = {gv[m].x, gv[m].y, gv[m].z, gv[m].w};
      float o4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float yv = fmaf(ea, xs[r], eb);
        const float gz = (ok && yv > 0.f) ? acc[m][r] : 0.f;
        const float xh = (xs[r] - emean) * einv;
        s1 += gz;
        s2 += gz * xh;
        o4[r] = fmaf(egam, gz, accum ? gs[r] : 0.f);
      }
      ov[m] = make_float4(o4[0], o4[1], o4[2], o4[3]);
    }
    // every prefetched register is consumed before the first store (vmcnt counts loads and stores together, in order)
#pragma unroll
    for (int m = 0; m < MPW; ++m) asm volatile("" ::"v"(ov[m].x), "v"(ov[m].y), "v"(ov[m].z), "v"(ov[m].w) : "memory");
#pragma unroll
    for (int m = 0; m < MPW; ++m)
      if (jv && ((vmask >> m) & 1u)) *reinterpret_cast<float4*>(Gc + pixoff[m]) = ov[m];
    s1 = group4_sum(s1);
    s2 = group4_sum(s2);
    if (lg == 0) {
      red[(wave * nct16 + ct * 16 + lp) * 2 + 0] = s1;
      red[(wave * nct16 + ct * 16 + lp) * 2 + 1] = s2;
    }
  }
  __syncthreads();
  if (p.stat_partial != nullptr) {
    const long long brow = (long long)n * gridDim.x + blockIdx.x;
    for (int j = ct_begin * 16 + tid; j < min(p.J, ct_end * 16); j += 256) {
      float a1 = 0.f, a2 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        a1 += red[(w * nct16 + j) * 2 + 0];
        a2 += red[(w * nct16 + j) * 2 + 1];
      }
      p.stat_partial[(brow * p.J + j) * 2 + 0] = a1;
      p.stat_partial[(brow * p.J + j) * 2 + 1] = a2;
    }
  }
}

bool d3_dgl_supported(const D3Dgl& p) {
  if (p.K < 1 || p.K > 16 || p.J < 1) return false;
  if ((p.W & 3) || p.W < 16 || p.H < 4 || (p.cs & 3) || (p.s_ns & 3)) return false;
  const uintptr_t amask = p.st == ST_BF16 ? 7 : 15;
  if ((reinterpret_cast<uintptr_t>(p.S) & amask) || (reinterpret_cast<uintptr_t>(p.G) & 15)) return false;
  return true;
}

void d3_dgl_plan(int H, int W, D3Dgl* p) {
  // covered area of each tiling; the one wasting fewer pixels (ties -> 8 x 32: longer rows)
  const long long a0 = (long long)((H + 7) / 8) * ((W + 31) / 32) * 256, a1 = (long long)((H + 15) / 16) * ((W + 15) / 16) * 256;
  p->tile = a1 < a0 ? 1 : 0;
  const int th = p->tile ? 16 : 8, tw = p->tile ? 16 : 32;
  p->tiles_y = (H + th - 1) / th;
  p->tiles_x = (W + tw - 1) / tw;
}

template <int NP, int DT, int ST, int TH, int TW>
static int d3_dgl_launch_t(const D3Dgl& p, hipStream_t s) {
  constexpr int P = TW + 3, ROWS = TH + 2;
  const int nct = (p.J + 15) / 16;
  const size_t lds = (size_t)NP * ROWS * P * 32 + (size_t)4 * nct * 16 * 2 * 4;
  if (lds > 64 * 1024) return -4;
  const long long base_blocks = (long long)p.tiles_x * p.tiles_y * p.N;
  int split = 1;  // aim for >= ~2 blocks per CU on small levels by splitting the output groups
  if (base_blocks < 512) split = (int)std::min<long long>((512 + base_blocks - 1) / base_blocks, (long long)nct);
  hipLaunchKernelGGL((d3_dgl_k<NP, DT, ST, TH, TW>), dim3((unsigned)(p.tiles_x * p.tiles_y), (unsigned)split, (unsigned)p.N),
                     dim3(256), lds, s, p);
  return (int)hipGetLastError();
}

int d3_dgl_launch(const D3Dgl& p, int np, int dt, hipStream_t s) {
  if (!d3_dgl_supported(p)) return -4;
#define D3_DGL(NP_, DT_, ST_)                                                            \
  return p.tile ? d3_dgl_launch_t<NP_, DT_, ST_, 16, 16>(p, s) : d3_dgl_launch_t<NP_, DT_, ST_, 8, 32>(p, s)
  if (p.st == ST_BF16) {
    if (np != 1 || dt != D3_BF16) return -4;
    D3_DGL(1, D3_BF16, ST_BF16);
  }
  if (dt != D3_BF16) return -4;  // gradients span far more than f16's exponent range
  if (np == 1) D3_DGL(1, D3_BF16, ST_F32);
  if (np == 2) D3_DGL(2, D3_BF16, ST_F32);
  D3_DGL(3, D3_BF16, ST_F32);
#undef D3_DGL
}

// =============================================================================================
// data gradient, pull form (see dense3.h)
//
// Persistent blocks of 8 waves loop over pixel tiles of th x tw = 160 pixels (10 M-tiles).  Per tile: all waves stage
// the nl dY tiles (+1 halo) as split 16-bit [pixel][16 channels] images (zero outside the picture); then every wave
// walks its own output-channel groups (no barriers): per group it loads S (and G where it accumulates) for its 4
// pixels x 10 M-tiles, and per layer runs 5 k-steps x 10 M-tiles of MFMAs (K = 16 output channels x 2 taps, flipped
// weights), applies the layer's ReLU mask, adds gamma_j * gz_j to the running sum and the two BatchNorm-backward sums to
// per-(layer, channel) slots in LDS.  G is written once per group.  Two waves per SIMD: one wave's epilogue VALU work
// overlaps the other's MFMAs.
// =============================================================================================
template <int NP, int DT, int ST, int MT>
__global__ __launch_bounds__(512, 2) void d3_pull_k(const D3Pull p) {
  extern __shared__ __align__(16) unsigned char smem[];
  // MT M-tiles per work item: an item = (output-channel group, 1/NSUB of the 160-pixel tile).  MT = 5 (halves) for the
  // wide input ranges of a block; MT = 2 (fifths) when the launch covers one 16-channel group (a block's own new
  // channels), so that five of the eight waves have work instead of two.
  constexpr int NSUB = 10 / MT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lp = lane & 15, lg = lane >> 4;
  const int P = p.tw + 3;
  const int rows = p.th + 2;
  const int PLANE = rows * P * 32;
  const int LIMG = NP * PLANE;  // one layer's image
  const int ngroups = (p.C + 15) >> 4;
  const int Cpad = ngroups * 16;
  float* stats = reinterpret_cast<float*>(smem + p.nl * LIMG);  // [sub][nl][Cpad][2]
  float* ctab = stats + NSUB * p.nl * Cpad * 2;                  // [nl][Cpad][4]: ea, eb, gamma, - ; then [Cpad][2] mean, invstd
  float* mtab = ctab + p.nl * Cpad * 4;
  for (int i = tid; i < NSUB * p.nl * Cpad * 2; i += 512) stats[i] = 0.f;
  for (int i = tid; i < p.nl * Cpad; i += 512) {  // per-(layer, channel) constants: tile-invariant, read from LDS later
    const int j = i / Cpad, ch = min(i - j * Cpad, p.C - 1);
    ctab[i * 4 + 0] = p.ea[j][ch];
    ctab[i * 4 + 1] = p.eb[j][ch];
    ctab[i * 4 + 2] = p.egamma[j][ch];
    ctab[i * 4 + 3] = 0.f;
  }
  for (int i = tid; i < Cpad; i += 512) {
    mtab[i * 2 + 0] = p.mean[min(i, p.C - 1)];
    mtab[i * 2 + 1] = p.invstd[min(i, p.C - 1)];
  }
  if (p.tiles_x == 1) {  // full-width tiles (rows of 20 / 40 pixels): the two halo columns are zero padding, written once
    for (int i = tid; i < p.nl * NP * rows * 2; i += 512) {
      const int side = i & 1, r = (i >> 1) % rows, jp = (i >> 1) / rows;  // jp = layer * NP + part
      uint4* z = reinterpret_cast<uint4*>(smem + jp * PLANE + (r * P + (side ? p.tw + 1 : 0)) * 32);
      z[0] = make_uint4(0u, 0u, 0u, 0u);
      z[1] = make_uint4(0u, 0u, 0u, 0u);
    }
  }

  const int tiles = p.tiles_x * p.tiles_y;
  const int total = tiles * p.N;
  const int npix = p.th * p.tw;

  // staging plan: lane-unit = (layer, 8-lane group: 2 channel octets x 4 rows, pixel quad); rounds over 512 threads
  const int nq = p.tw >> 2;
  const int nrb = (rows + 3) >> 2;
  const int lul = 8 * nrb * nq;  // lane-units per layer
  const int hul = rows * 16;     // halo units per layer: (row, side, channel pair)

  int toff[5];
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    const int tap = min(2 * s + (lg >> 1), 8);
    toff[s] = ((tap / 3 - 1) * P + (tap % 3 - 1)) * 32;
  }

#ifdef RLN_DIAG
  // phase stamps (wave cycles summed over all waves; tools/pull_stamps.py): 0 wait at the tile barrier, 1 dY staging,
  // 2 wait for the staged tile, 3 item / layer set-up, 4 MFMA loops, 5 layer epilogues, 6 G read-modify-write, 7 items
  unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
#define D3P_STAMP(k)                                             \
  do {                                                           \
    __builtin_amdgcn_sched_barrier(0);                           \
    const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); \
    __builtin_amdgcn_s_waitcnt(0xC07F);                          \
    tph[k] += tn_ - tlast;                                       \
    tlast = tn_;                                                 \
    __builtin_amdgcn_sched_barrier(0);                           \
  } while (0)
#else
#define D3P_STAMP(k) do {} while (0)
#endif
  for (int t = blockIdx.x; t < total; t += gridDim.x) {
    const int n = t / tiles, tl = t - n * tiles;
    const int tile_y = tl / p.tiles_x, tile_x = tl - tile_y * p.tiles_x;
    const int gy0 = tile_y * p.th, gx0 = tile_x * p.tw;
    __syncthreads();  // previous tile's images are no longer read (also orders the stats zeroing)
    D3P_STAMP(0);
    // ---- stage the dY images (zero outside the picture and beyond Cout): all loads first, then convert ----
    {
      constexpr int NRD = 2;  // staging rounds (nl * lul <= 1024: checked by the launcher)
      typename SRaw<ST>::r4 v[NRD][8];  // dY shares the level's storage type with S
      float hv[2];
      int sdst[NRD];
      unsigned sflag[NRD];  // bit 0: unit exists, bit 1: inside the picture, bits 8..: octet
#pragma unroll
      for (int i = 0; i < NRD; ++i) {
        const int lu = tid + 512 * i;
        const int j = min(lu / lul, p.nl - 1);
        const int l2 = lu - j * lul;
        const int o = l2 & 1, rr = (l2 >> 1) & 3, u = l2 >> 3;
        const int R = u / nq, Q = u - R * nq;
        const int r = 4 * R + rr;
        const bool ex = lu < p.nl * lul && r < rows;
        const int iy = gy0 - 1 + r, ix = gx0 + 4 * Q;
        const bool ok = ex && iy >= 0 && iy < p.H && ix < p.W;
        const SP<ST> src = SP<ST>(p.dY[j]) + (((long long)n * p.Cout) * p.cs + (ok ? iy * p.W + ix : 0));
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) {
          const int ch = min(o * 8 + cc, p.Cout - 1);
          v[i][cc] = src.raw4((long long)ch * p.cs);
        }
        sdst[i] = j * LIMG + ((ex ? r : 0) * P + 1 + 4 * Q) * 32 + o * 16;
        sflag[i] = (ex ? 1u : 0u) | (ok ? 2u : 0u) | ((unsigned)o << 8);
      }
      int hdst = 0;
      unsigned hflag = 0;
      {
        const int hu = tid;
        const int j = min(hu / hul, p.nl - 1);
        const int h2 = hu - j * hul;
        const int cp = h2 & 7, side = (h2 >> 3) & 1, r = h2 >> 4;
        const bool ex = hu < p.nl * hul && p.tiles_x > 1;  // full-width tiles: the halo columns are padding (zeroed once)
        const int iy = gy0 - 1 + r, ix = side ? gx0 + p.tw : gx0 - 1;
        const bool ok = ex && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        const SP<ST> src = SP<ST>(p.dY[j]) + (((long long)n * p.Cout) * p.cs + (ok ? iy * p.W + ix : 0));
        hv[0] = src.ld1((long long)min(2 * cp, p.Cout - 1) * p.cs);
        hv[1] = src.ld1((long long)min(2 * cp + 1, p.Cout - 1) * p.cs);
        if (!ok || 2 * cp >= p.Cout) hv[0] = 0.f;
        if (!ok || 2 * cp + 1 >= p.Cout) hv[1] = 0.f;
        hdst = j * LIMG + ((ex ? r : 0) * P + (side ? p.tw + 1 : 0)) * 32 + cp * 4;
        hflag = ex ? 1u : 0u;
      }
#pragma unroll
      for (int i = 0; i < NRD; ++i) {
        if (sflag[i] & 1u) {
          const bool ok = (sflag[i] & 2u) != 0;
          const int o = (int)(sflag[i] >> 8);
#pragma unroll
          for (int px = 0; px < 4; ++px) {
            unsigned parts[4][NP];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const float4 u0 = SRaw<ST>::w4(v[i][2 * k]), u1 = SRaw<ST>::w4(v[i][2 * k + 1]);
              float x0 = px == 0 ? u0.x : px == 1 ? u0.y : px == 2 ? u0.z : u0.w;
              float x1 = px == 0 ? u1.x : px == 1 ? u1.y : px == 2 ? u1.z : u1.w;
              if (!ok || o * 8 + 2 * k >= p.Cout) x0 = 0.f;
              if (!ok || o * 8 + 2 * k + 1 >= p.Cout) x1 = 0.f;
              split2<DT, NP>(x0, x1, parts[k]);
            }
#pragma unroll
            for (int pt = 0; pt < NP; ++pt)
              *reinterpret_cast<uint4*>(smem + sdst[i] + pt * PLANE + px * 32) =
                  make_uint4(parts[0][pt], parts[1][pt], parts[2][pt], parts[3][pt]);
          }
        }
      }
      if (hflag & 1u) {
        unsigned parts[NP];
        split2<DT, NP>(hv[0], hv[1], parts);
#pragma unroll
        for (int pt = 0; pt < NP; ++pt) *reinterpret_cast<unsigned*>(smem + hdst + pt * PLANE) = parts[pt];
      }
    }
    D3P_STAMP(1);
    __syncthreads();
    D3P_STAMP(2);

    // ---- per-wave loop over work items (no barriers) ----
    const SP<ST> Sn = SP<ST>(p.S) + (long long)n * p.s_ns;
    float* Gn = p.G + (long long)n * p.s_ns;
    uint4 bf[5][NP];
    bool bf_valid = false;
#pragma unroll 1
    for (int item = wave; item < NSUB * ngroups; item += 8) {
      const int g = item / NSUB, half = item - g * NSUB;  // "half" = which 1/NSUB of the tile
      int basem[MT], goff[MT];
      unsigned vmask = 0;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int mt = half * MT + m;
        {
          const int q = min(mt * 16 + lp, npix - 1);
          const int ty = q / p.tw, tx = q - ty * p.tw;
          basem[m] = ((ty + 1) * P + tx + 1) * 32 + (lg & 1) * 16;
        }
        const int q = mt * 16 + lg * 4;
        const int ty = q / p.tw, tx = q - ty * p.tw;
        const int gy = gy0 + ty, gx = gx0 + tx;
        const bool ok = q < npix && gy < p.H && gx < p.W;
        vmask |= (ok ? 1u : 0u) << m;
        goff[m] = ok ? gy * p.W + gx : 0;
      }
      const int c = g * 16 + lp;
      const bool cv = c < p.C;
      const int cc = cv ? c : p.C - 1;
      const bool accum = c >= p.acc_lo && c < p.acc_hi;
      const SP<ST> Sc = Sn + (long long)cc * p.cs;
      float* Gc = Gn + (long long)cc * p.cs;
      float4 sv[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) sv[m] = Sc.ld4(goff[m]);
      const float mean = mtab[(g * 16 + lp) * 2], invstd = mtab[(g * 16 + lp) * 2 + 1];
      f32x4 gsum[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) gsum[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (!bf_valid) {  // first (item, layer) of this wave in the tile: nothing was prefetched
        const uint4* wp = p.wpk[0] + ((long long)g * 5 * NP) * 64 + lane;
#pragma unroll
        for (int s0 = 0; s0 < 5; ++s0)
#pragma unroll
          for (int pt = 0; pt < NP; ++pt) bf[s0][pt] = wp[(s0 * NP + pt) * 64];
        bf_valid = true;
      }
#pragma unroll 1
      for (int j = 0; j < p.nl; ++j) {
        const unsigned char* img = smem + j * LIMG;
        // weight fragments of the NEXT (item, layer): each k-step's registers are refilled as soon as the step is done,
        // so the L2 latency of the 10 KB per (group, layer) hides behind the remaining steps and the epilogue
        const bool last_layer = j + 1 == p.nl;
        const int gn = last_layer ? ((item + 8) / NSUB) : g;
        const bool has_next = !last_layer || (item + 8 < NSUB * ngroups);
        const uint4* wpn = p.wpk[last_layer ? 0 : j + 1] + ((long long)(has_next ? gn : g) * 5 * NP) * 64 + lane;
        const float4 cst = *reinterpret_cast<const float4*>(ctab + ((long long)j * Cpad + g * 16 + lp) * 4);
        const float ea = cst.x, eb = cst.y, egam = cst.z;
        f32x4 acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        // keep the 25 x NP fragment addresses from being hoisted out of the layer loop as loop invariants (50+ VGPRs):
        // the per-tile bases are re-materialised as opaque values in every layer iteration
        int bm[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          bm[m] = basem[m];
          asm volatile("" : "+v"(bm[m]));
        }
        constexpr int DEPTH = 4, RING = DEPTH + 1, STEPS = 5 * MT;
        uint4 af[RING][NP];
        D3P_STAMP(3);
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) {
          const int s0 = i / MT, m0 = i - s0 * MT;
#pragma unroll
          for (int pt = 0; pt < NP; ++pt)
            af[i % RING][pt] = *reinterpret_cast<const uint4*>(img + pt * PLANE + bm[m0] + toff[s0]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < STEPS; ++i) {
          const int s = i / MT, m = i - s * MT;
          if (i + DEPTH < STEPS) {
            const int s1 = (i + DEPTH) / MT, m1 = (i + DEPTH) - s1 * MT;
#pragma unroll
            for (int pt = 0; pt < NP; ++pt)
              af[(i + DEPTH) % RING][pt] = *reinterpret_cast<const uint4*>(img + pt * PLANE + bm[m1] + toff[s1]);
          }
          acc[m] = mfma_split<DT, NP>(af[i % RING], bf[s], acc[m]);
          if (m == MT - 1) {
#pragma unroll
            for (int pt = 0; pt < NP; ++pt) bf[s][pt] = wpn[(s * NP + pt) * 64];
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        // epilogue of layer j: ReLU mask of the layer's BatchNorm output, BN-backward sums, gamma-weighted sum
        mfma_drain();  // wait states between the MFMA chain and the first accumulator read (split16.h)
#ifdef RLN_DIAG
        asm volatile("" ::"v"(acc[0][0]), "v"(acc[MT - 1][3]));
#endif
        D3P_STAMP(4);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const bool ok = cv && ((vmask >> m) & 1u);
          // ((x - mean) * invstd is hoisted in front of the layer loop by the compiler: 20 live registers, and the item's S
          // loads are waited for before its first MFMA.  Made opaque per layer -- 252 VGPRs, no scratch, the wait after the
          // first layer's 75 MFMAs -- the step did not move (fp32 +-0, bf16 +0.04 ms): the second wave of the SIMD already
          // covers that latency.  Not kept.)
          const float xs[4] = {sv[m].x, sv[m].y, sv[m].z, sv[m].w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float yv = fmaf(ea, xs[r], eb);
            const float gz = (ok && yv > 0.f) ? acc[m][r] : 0.f;
            const float xh = (xs[r] - mean) * invstd;
            s1 += gz;
            s2 += gz * xh;
            gsum[m][r] = fmaf(egam, gz, gsum[m][r]);
          }
        }
        s1 = group4_sum(s1);
        s2 = group4_sum(s2);
        if (lg == 0) {  // (half, layer, channel) slots are owned by exactly one wave per tile: plain read-modify-write
          float* st = stats + (((long long)half * p.nl + j) * Cpad + g * 16 + lp) * 2;
          st[0] += s1;
          st[1] += s2;
        }
        D3P_STAMP(5);
      }
      // write G once (the activations are dead by now: their registers take the old gradient where it accumulates)
#pragma unroll
      for (int m = 0; m < MT; ++m)
        sv[m] = accum ? *reinterpret_cast<const float4*>(Gc + goff[m]) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        if (cv && ((vmask >> m) & 1u))
          *reinterpret_cast<float4*>(Gc + goff[m]) = make_float4(gsum[m][0] + sv[m].x, gsum[m][1] + sv[m].y,
                                                                 gsum[m][2] + sv[m].z, gsum[m][3] + sv[m].w);
      }
      D3P_STAMP(6);
#ifdef RLN_DIAG
      tph[7] += 1;
#endif
    }
  }
#ifdef RLN_DIAG
  if (lane == 0 && p.dbg_out != nullptr) {
#pragma unroll
    for (int k = 0; k < 8; ++k) atomicAdd(&p.dbg_out[k], tph[k]);
  }
#endif
  __syncthreads();
  if (p.stat_partial != nullptr) {  // rows: [block][sub]
    float* dst = p.stat_partial + (long long)blockIdx.x * NSUB * p.nl * Cpad * 2;
    for (int i = tid; i < NSUB * p.nl * Cpad * 2; i += 512) dst[i] = stats[i];
  }
}

__global__ __launch_bounds__(256) void d3_pull_finalize_k(const D3PullFin f) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= f.C) return;
  float a1 = 0.f, a2 = 0.f;
  for (int j = 0; j < f.nl; ++j) {
    double s1 = 0.0, s2 = 0.0;
    for (int r = lane; r < f.rows; r += 64) {
      const float* q = f.partial + (((long long)r * f.nl + j) * f.Cpad + c) * 2;
      s1 += (double)q[0];
      s2 += (double)q[1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      s1 += __shfl_xor(s1, o, 64);
      s2 += __shfl_xor(s2, o, 64);
    }
    if (lane == 0) {
      f.dbeta[j][c] = (float)s1;
      f.dgamma[j][c] = (float)s2;
      const float g = f.gamma[j][c];
      a1 += g * (float)s1;
      a2 += g * (float)s2;
    }
  }
  if (lane == 0) {
    f.S1[c] += a1;
    f.S2[c] += a2;
  }
}

int d3_pull_finalize(const D3PullFin& f, hipStream_t s) {
  hipLaunchKernelGGL(d3_pull_finalize_k, dim3((unsigned)((f.C + 3) / 4)), dim3(256), 0, s, f);
  return (int)hipGetLastError();
}

// work items per (group, tile): see d3_pull_k.  (Measured and not kept: whole-tile items, MT = 10, for ranges of >= 8
// groups -- half the weight-fragment fetches per pixel, but 256 VGPRs + 476 B of spills per lane: the level-0 up-block
// launch 1.62 -> 2.93 ms.  An ablation that never re-fetches the fragments changes nothing (1.64 -> 1.61 ms): the 10 KB
// per (item, layer) from L2 are not what the item loop waits for.  In-kernel stamps of the level-0 up-block launch
// (tools/pull_stamps.py, wave cycles): MFMA loops 29 %, layer epilogues 22 %, item / layer set-up 15 %, G tail 15 %
// (26 % with bf16 stacks), dY staging + its two barriers 20 %.  The G tail guards its loads and stores per lane and the
// compiler keeps an s_cbranch_execz around each, so the wait-count pass drains every outstanding memory operation
// there; straight-line tails for whole-tile / all-or-nothing-accumulate launches (two more template variants) plus the
// first fragment load hoisted out of the item loop brought the stamped tail to 11 % but the step from 24.48 to 24.62 ms
// (A/B on one box, tools/ab_bench.sh): not kept.  Nor is an M-tile-major step order whose per-M-tile epilogue slices are
// issued between the next M-tile's MFMAs (no extra registers, bit-identical): +0.16 ms per step.)
int d3_pull_nsub(const D3Pull& p) { return p.C <= 16 ? 5 : 2; }

static size_t d3_pull_lds(const D3Pull& p, int np) {
  const int P = p.tw + 3, rows = p.th + 2;
  const int Cpad = ((p.C + 15) / 16) * 16;
  return (size_t)p.nl * np * rows * P * 32 + (size_t)d3_pull_nsub(p) * p.nl * Cpad * 2 * 4 + (size_t)p.nl * Cpad * 16 +
         (size_t)Cpad * 8;
}

bool d3_pull_supported(const D3Pull& p, int np) {
  if (p.nl < 1 || p.nl > D3_LMAX || p.Cout < 1 || p.Cout > 16 || p.C < 1) return false;
  if (p.nl * 8 * ((p.th + 2 + 3) / 4) * (p.tw / 4) > 1024) return false;                  // staging rounds
  if (p.tiles_x > 1 && p.nl * (p.th + 2) * 16 > 512) return false;                         // halo units
  if (p.th * p.tw != 160 || d3_pull_lds(p, np) > 160 * 1024) return false;
  if ((p.W & 3) || (p.cs & 3) || (p.s_ns & 3)) return false;
  if (!(p.W % 80 == 0 || p.W == 40 || p.W == 20)) return false;
  if ((p.W == 40 || p.W == 20) && p.tiles_x != 1) return false;
  const uintptr_t amask = p.st == ST_BF16 ? 7 : 15;
  if ((reinterpret_cast<uintptr_t>(p.S) & amask) || (reinterpret_cast<uintptr_t>(p.G) & 15)) return false;
  for (int j = 0; j < p.nl; ++j)
    if (reinterpret_cast<uintptr_t>(p.dY[j]) & amask) return false;
  return true;
}

void d3_pull_pick_tile(int H, int W, int* th, int* tw) {
  (void)H;
  if (W % 80 == 0) {  // (4 x 40 tiles -- dY halo 1.58x instead of 2.05x -- measured +0.15 ms per step: shorter rows)
    *tw = 80;
    *th = 2;
  } else if (W == 20) {
    *tw = 20;
    *th = 8;
  } else {
    *tw = 40;
    *th = 4;
  }
}

int d3_pull_blocks(const D3Pull& p) {
  const long long total = (long long)p.tiles_x * p.tiles_y * p.N;
  return (int)std::min<long long>(total, 256);
}

template <int NP, int DT, int ST, int MT>
static int d3_pull_launch_m(const D3Pull& p, hipStream_t s) {
  const size_t lds = d3_pull_lds(p, NP);
  if (lds > 160 * 1024) return -4;
  auto kern = d3_pull_k<NP, DT, ST, MT>;
  static DevOnce attr_once;
  if (attr_once.first()) {
    const hipError_t attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    if (attr_err != hipSuccess) {  // refused: report it here instead of an opaque launch failure later
      (void)hipGetLastError();
      attr_once.undo();
      return (int)attr_err;
    }
  }
#ifdef RLN_DIAG
  if (rln_env("RLN_PULL_STAMPS") && p.C >= atoi(rln_env("RLN_PULL_STAMPS"))) {  // stamps of the launches with >= that many channels
    D3Pull q = p;
    q.dbg_out = igemm_debug_buffer();
    hipLaunchKernelGGL(kern, dim3((unsigned)d3_pull_blocks(p)), dim3(512), lds, s, q);
    return (int)hipGetLastError();
  }
#endif
  hipLaunchKernelGGL(kern, dim3((unsigned)d3_pull_blocks(p)), dim3(512), lds, s, p);
  return (int)hipGetLastError();
}

template <int NP, int DT, int ST = ST_F32>
static int d3_pull_launch_t(const D3Pull& p, hipStream_t s) {
  return d3_pull_nsub(p) == 5 ? d3_pull_launch_m<NP, DT, ST, 2>(p, s) : d3_pull_launch_m<NP, DT, ST, 5>(p, s);
}

int d3_pull_launch(const D3Pull& p, int np, int dt, hipStream_t s) {
  if (!d3_pull_supported(p, np)) return -4;
  if (p.st == ST_BF16) {
    if (np != 1 || dt != D3_BF16) return -4;
    return d3_pull_launch_t<1, D3_BF16, ST_BF16>(p, s);
  }
  if (dt == D3_BF16) {
    if (np == 1) return d3_pull_launch_t<1, D3_BF16>(p, s);
    if (np == 2) return d3_pull_launch_t<2, D3_BF16>(p, s);
    return d3_pull_launch_t<3, D3_BF16>(p, s);
  }
  if (np == 1) return d3_pull_launch_t<1, D3_F16>(p, s);
  return d3_pull_launch_t<2, D3_F16>(p, s);
}

}  // namespace rln
