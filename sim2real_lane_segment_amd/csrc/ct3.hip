// TransitionUp (ConvTranspose2d 3x3, stride 2) kernels on the 16-bit MFMA pipe with split fp32 operands (see ct3.h).
#include "ct3.h"

#include <algorithm>
#include <cstdio>

#include "split16.h"
#include "storage.h"

namespace rln {

constexpr int C3_LDS_BUDGET = 150 * 1024;
constexpr int C3_MT = 2;  // M tiles per block at most (LDS: 9 taps x K steps x parts per tile)

// lane i <- lane i-1 within a 16-lane row (lane 0 <- 0)
__device__ __forceinline__ unsigned row_from_prev(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
}
// lane i <- lane i+1 within a 16-lane row (lane 15 <- 0)
__device__ __forceinline__ unsigned row_from_next(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xf, 0xf, true);
}
__device__ __forceinline__ uint4 and4(uint4 v, unsigned m) { return make_uint4(v.x & m, v.y & m, v.z & m, v.w & m); }

// =============================================================================================
// weight packing
// =============================================================================================
template <int DT, int NP>
__global__ __launch_bounds__(256) void c3_pack_k(const float* __restrict__ params, const C3PackDesc* __restrict__ desc,
                                                 int n_desc, int total_units, uint4* __restrict__ packed) {
  const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (unit >= total_units) return;
  int d = 0;
  while (d + 1 < n_desc && desc[d + 1].unit_begin <= unit) ++d;
  const C3PackDesc q = desc[d];
  int u = unit - q.unit_begin;
  const int nf = q.wf_off >= 0 ? ((q.cout + 15) >> 4) * ((q.cin + 31) >> 5) * 9 : 0;
  const bool backward = u >= nf;
  if (backward) {
    u -= nf;
    if (q.wb_off < 0) return;
  }
  const int KS = backward ? (q.cout + 31) >> 5 : (q.cin + 31) >> 5;
  const int tap = u % 9;
  const int mk = u / 9;
  const int mtile = mk / KS, ks = mk - mtile * KS;
  const int i = lane & 15, kb = lane >> 4;
  const float* w = params + q.w_off;
  const int row = mtile * 16 + i;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = ks * 32 + kb * 8 + e;
    float val = 0.f;
    if (!backward) {  // row = o, k = c
      if (row < q.cout && k < q.cin)
        val = sat16<DT>(w[((long long)k * q.cout + row) * 9 + tap] * w_prescale<DT>());  // split16.h: f16 range handling
    } else {  // row = c, k = o
      if (row < q.cin && k < q.cout) val = w[((long long)row * q.cout + k) * 9 + tap];
    }
    v[e] = val;
  }
  unsigned parts[4][NP];
#pragma unroll
  for (int j = 0; j < 4; ++j) split2<DT, NP>(v[2 * j], v[2 * j + 1], parts[j]);
  long long slot_index = u;  // forward: [mtile][ks][tap]
  if (backward) {              // backward: [group][ks][ky][m in group][kx] (one K-step row of taps is contiguous)
    const int MT = (q.cin + 15) >> 4;
    const int mtb = c3_dgrad_mt(MT);
    const int grp = mtile / mtb, m = mtile - grp * mtb;
    slot_index = ((((long long)grp * KS + ks) * 3 + tap / 3) * mtb + m) * 3 + tap % 3;
  }
  uint4* dst = packed + (backward ? q.wb_off : q.wf_off) + (slot_index * NP) * 64 + lane;
#pragma unroll
  for (int p = 0; p < NP; ++p) dst[p * 64] = make_uint4(parts[0][p], parts[1][p], parts[2][p], parts[3][p]);
}

int c3_pack_weights(const float* params, const C3PackDesc* desc_dev, int n_desc, int total_units, uint4* packed, int np,
                    int dt, hipStream_t s) {
  if (total_units <= 0) return 0;
  dim3 grid((unsigned)((total_units + 3) / 4));
#define C3_PACK(DT_, NP_)                                                                                         \
  hipLaunchKernelGGL((c3_pack_k<DT_, NP_>), grid, dim3(256), 0, s, params, desc_dev, n_desc, total_units, packed)
  if (dt == D3_BF16) {
    if (np == 1) C3_PACK(D3_BF16, 1);
    else if (np == 2) C3_PACK(D3_BF16, 2);
    else C3_PACK(D3_BF16, 3);
  } else {
    if (np == 1) C3_PACK(D3_F16, 1);
    else if (np == 2) C3_PACK(D3_F16, 2);
    else C3_PACK(D3_F16, 3);
  }
#undef C3_PACK
  return (int)hipGetLastError();
}

static void c3_group_plan(int m_tiles, int ksteps, int np, int* mt, int* groups) {
  const int per = ksteps * 9 * np * 1024 + 16 * 4 + 8 * 16 * 2 * 4;
  int cap = C3_LDS_BUDGET / per;
  cap = std::max(1, std::min(C3_MT, cap));
  *groups = (m_tiles + cap - 1) / cap;
  *mt = (m_tiles + *groups - 1) / *groups;
}

// =============================================================================================
// forward
//
// Block = 8 persistent waves holding the fragments of `mt` M tiles (16 output channels each; all 9 taps, all K steps) in
// LDS.  A wave walks pairs of wave tiles (2 x 15 input pixels; the weight fragments read from LDS serve both).  Per K step
// a lane loads its pixel and the pixel above for 8 channels (scalar loads, 15 lanes = 60 contiguous bytes), splits them
// into 16-bit parts = B fragments "own" and "up"; "left" / "up-left" are the previous lane's fragments (DPP), zeroed where
// the lane's pixel starts a row.  Four accumulators per M tile, one per output parity:
//   P00 += W00*own + W02*left + W20*up + W22*upleft     P01 += W01*own + W21*up
//   P10 += W10*own + W12*left                            P11 += W11*own
// Epilogue: + bias, crop to Ho x Wo, 8-byte stores (rows 2y and 2y+1), per-channel sums for the next BatchNorm.
// =============================================================================================
// WT: wave tiles a wave multiplies per pass.  WT = 2 shares every LDS weight-fragment read between two tiles but needs
// ~310 VGPRs at two parts: 58 spilled registers, reloaded inside the K loop -- and the kernel was NOT bitwise repeatable
// (tools/tu_stress.py: 21 of 3000 launches with one wave tile's outputs and sums changed, always MFMA rows 13 / 15: the
// accumulators were spilled too soon after the MFMA that wrote them -- see mfma_drain() in split16.h).  WT = 1 fits the
// register file without scratch.
template <int NP, int DT, int XT, int OT, int WT>
__global__ __launch_bounds__(512, 2) void c3_fwd_k(const C3Fwd p) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n16 = lane & 15, kb = lane >> 4;
  const int KS = (p.Cin + 31) >> 5;
  const int g = blockIdx.x / p.bpg, b = blockIdx.x - g * p.bpg;
  const int m0 = g * p.mt;
  const int mt = min(p.mt, ((p.Cout + 15) >> 4) - m0);
  uint4* wl = reinterpret_cast<uint4*>(smem);                                          // [mt][KS][9][NP][64]
  float* btab = reinterpret_cast<float*>(smem + (size_t)p.mt * KS * 9 * NP * 1024);     // [p.mt*16]
  float* slot = btab + p.mt * 16;                                                       // [8][p.mt*16][2]
  {
    const uint4* src = p.wpk + (long long)m0 * KS * 9 * NP * 64;
    const int cnt = mt * KS * 9 * NP * 64;
    for (int i = tid; i < cnt; i += 512) wl[i] = src[i];
    for (int i = tid; i < p.mt * 16; i += 512) btab[i] = (p.bias && m0 * 16 + i < p.Cout) ? p.bias[m0 * 16 + i] : 0.f;
    for (int i = tid; i < 8 * p.mt * 32; i += 512) slot[i] = 0.f;
  }
  __syncthreads();

  const int CH = (p.Ho + 1) >> 1, CW = (p.Wo + 1) >> 1, CC = CH * CW;
  const int cells = p.N * CC;
  const int ntiles = (cells + 14) / 15;
  const int nsuper = (ntiles + WT - 1) / WT;
  const int sstride = p.bpg * 8;
  const bool vec2 = ((p.Wo | p.out_cs) & 1) == 0 && (p.out_ns & 1) == 0 &&
                    (reinterpret_cast<uintptr_t>(p.out) & (2 * SP<OT>::ES - 1)) == 0;

  struct Cell {
    SP<XT> own;
    SP<XT> up;
    SP<OT> outp;          // out at (sample, channel 0, row 2y, column 2x)
    unsigned left_mask;   // all ones when the lane has a left neighbour in its row
    bool v_own, v_up, valid;
    int oy, ox;
  };
  auto setup_cell = [&](int T) __attribute__((always_inline)) {
    const int id = T * 15 + n16 - 1;
    const bool inr = id >= 0 && id < cells;
    const int idc = min(max(id, 0), cells - 1);
    const int ns_ = idc / CC;
    const int rem = idc - ns_ * CC;
    const int y = rem / CW, x = rem - y * CW;
    Cell c;
    c.v_own = inr && y < p.H && x < p.W;
    c.v_up = inr && y >= 1 && x < p.W;
    c.valid = inr && n16 >= 1;
    c.left_mask = x >= 1 ? 0xFFFFFFFFu : 0u;
    const SP<XT> xb = SP<XT>(p.X) + ((long long)ns_ * p.ns + min(x, p.W - 1));
    c.own = xb + (long long)min(y, p.H - 1) * p.W;
    c.up = xb + (long long)max(min(y, p.H) - 1, 0) * p.W;
    c.oy = 2 * y;
    c.ox = 2 * x;
    c.outp = SP<OT>(p.out) + ((long long)ns_ * p.out_ns + (long long)c.oy * p.Wo + c.ox);
    return c;
  };

  typename SRaw<XT>::r1 qo[WT][8], qu[WT][8];  // narrow until convert widens them
  auto issue = [&](const Cell& c, int w, int ks) __attribute__((always_inline)) {
    const long long off = (long long)min(ks * 32 + kb * 8, p.Cin - 8) * p.cs;  // Cin % 8 == 0; weights past Cin are zero
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      qo[w][e] = c.own.raw1(off + (long long)e * p.cs);
      qu[w][e] = c.up.raw1(off + (long long)e * p.cs);
    }
  };
  uint4 fo[WT][NP], fu[WT][NP], fl[WT][NP], ful[WT][NP];
  auto convert = [&](const Cell& c, int w) __attribute__((always_inline)) {
    unsigned a[4][NP], u[4][NP];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // raw (un-normalised) block output: saturate into the part type's range before the split (split16.h)
      split2<DT, NP>(c.v_own ? sat16<DT>(SRaw<XT>::w1(qo[w][2 * j])) : 0.f,
                     c.v_own ? sat16<DT>(SRaw<XT>::w1(qo[w][2 * j + 1])) : 0.f, a[j]);
      split2<DT, NP>(c.v_up ? sat16<DT>(SRaw<XT>::w1(qu[w][2 * j])) : 0.f,
                     c.v_up ? sat16<DT>(SRaw<XT>::w1(qu[w][2 * j + 1])) : 0.f, u[j]);
    }
#pragma unroll
    for (int pt = 0; pt < NP; ++pt) {
      fo[w][pt] = make_uint4(a[0][pt], a[1][pt], a[2][pt], a[3][pt]);
      fu[w][pt] = make_uint4(u[0][pt], u[1][pt], u[2][pt], u[3][pt]);
      fl[w][pt] = and4(make_uint4(row_from_prev(a[0][pt]), row_from_prev(a[1][pt]), row_from_prev(a[2][pt]),
                                  row_from_prev(a[3][pt])), c.left_mask);
      ful[w][pt] = and4(make_uint4(row_from_prev(u[0][pt]), row_from_prev(u[1][pt]), row_from_prev(u[2][pt]),
                                   row_from_prev(u[3][pt])), c.left_mask);
    }
  };

  f32x4 acc[C3_MT][WT][4];
  int ST = b * 8 + wave;
  Cell c0 = setup_cell(WT * min(ST, nsuper - 1)), c1 = setup_cell(WT * min(ST, nsuper - 1) + (WT - 1));
  if (ST < nsuper) {
    issue(c0, 0, 0);
    if constexpr (WT == 2) issue(c1, 1, 0);
  }
  while (ST < nsuper) {
#pragma unroll
    for (int m = 0; m < C3_MT; ++m)
#pragma unroll
      for (int w = 0; w < WT; ++w)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[m][w][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int STn = ST + sstride;
    Cell n0 = c0, n1 = c1;
    for (int ks = 0; ks < KS; ++ks) {
      convert(c0, 0);
      if constexpr (WT == 2) convert(c1, 1);
      __builtin_amdgcn_sched_barrier(0);
      if (ks + 1 < KS) {
        issue(c0, 0, ks + 1);
        if constexpr (WT == 2) issue(c1, 1, ks + 1);
      } else if (STn < nsuper) {
        n0 = setup_cell(WT * STn);
        issue(n0, 0, 0);
        if constexpr (WT == 2) {
          n1 = setup_cell(WT * STn + 1);
          issue(n1, 1, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < C3_MT; ++m) {
        if (m < mt) {
          const uint4* wb = wl + ((m * KS + ks) * 9) * NP * 64 + lane;
          auto tapA = [&](int t, uint4 (&A)[NP]) __attribute__((always_inline)) {
#pragma unroll
            for (int pt = 0; pt < NP; ++pt) A[pt] = wb[(t * NP + pt) * 64];
          };
          uint4 A[NP];
#define C3_TAP(T_, F_, P_)                                                                     \
  tapA(T_, A);                                                                                 \
  acc[m][0][P_] = mfma_split<DT, NP>(A, F_[0], acc[m][0][P_]);                                 \
  if constexpr (WT == 2) acc[m][WT - 1][P_] = mfma_split<DT, NP>(A, F_[WT - 1], acc[m][WT - 1][P_]);
          C3_TAP(0, fo, 0)
          C3_TAP(2, fl, 0)
          C3_TAP(6, fu, 0)
          C3_TAP(8, ful, 0)
          C3_TAP(1, fo, 1)
          C3_TAP(7, fu, 1)
          C3_TAP(3, fo, 2)
          C3_TAP(5, fl, 2)
          C3_TAP(4, fo, 3)
#undef C3_TAP
        }
      }
    }
    // ---- epilogue ----
    mfma_drain();  // wait states between the MFMA chain and the first accumulator read (split16.h)
    int kb4 = 4 * kb;
    asm volatile("" : "+v"(kb4));
#pragma unroll
    for (int w = 0; w < WT; ++w) {
      const Cell& c = w == 0 ? c0 : c1;
      const bool r0v = c.valid && c.oy < p.Ho, r1v = c.valid && c.oy + 1 < p.Ho;
      const bool x0v = c.ox < p.Wo, x1v = c.ox + 1 < p.Wo;
#pragma unroll
      for (int m = 0; m < C3_MT; ++m) {
        if (m < mt) {
          const float4 b4 = *reinterpret_cast<const float4*>(btab + m * 16 + kb4);
          const float bia[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int ol = m * 16 + kb4 + r;
            const int o = m0 * 16 + ol;
            const bool ov = o < p.Cout;
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)  // (rounded first: statistics of the stored tensor)
              v[q] = st_round<OT>(fmaf(acc[m][w][q][r], w_unscale<DT>(), bia[r]));
            const SP<OT> dst = c.outp + (long long)o * p.out_cs;
            const bool m00 = ov && r0v && x0v, m01 = ov && r0v && x1v, m10 = ov && r1v && x0v, m11 = ov && r1v && x1v;
            if (vec2) {  // Wo even: a block column pair is all-in or all-out
              if (m00) dst.st2(0, v[0], v[1]);
              if (m10) dst.st2(p.Wo, v[2], v[3]);
            } else {
              if (m00) dst.st1(0, v[0]);
              if (m01) dst.st1(1, v[1]);
              if (m10) dst.st1(p.Wo, v[2]);
              if (m11) dst.st1(p.Wo + 1, v[3]);
            }
            float s1 = (m00 ? v[0] : 0.f) + (m01 ? v[1] : 0.f) + (m10 ? v[2] : 0.f) + (m11 ? v[3] : 0.f);
            float s2 = (m00 ? v[0] * v[0] : 0.f) + (m01 ? v[1] * v[1] : 0.f) + (m10 ? v[2] * v[2] : 0.f) +
                       (m11 ? v[3] * v[3] : 0.f);
            s1 = row16_sum(s1);
            s2 = row16_sum(s2);
            if (n16 == 0) {
              float* sl = slot + ((wave * p.mt * 16) + ol) * 2;
              lds_add_f32(sl, s1);
              lds_add_f32(sl + 1, s2);
            }
          }
        }
      }
    }
    ST = STn;
    c0 = n0;
    c1 = n1;
  }
  __syncthreads();
  if (p.stat_partial != nullptr && tid < mt * 16) {
    const int o = m0 * 16 + tid;
    if (o < p.Cout) {
      float a1 = 0.f, a2 = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) {
        a1 += slot[((w * p.mt * 16) + tid) * 2 + 0];
        a2 += slot[((w * p.mt * 16) + tid) * 2 + 1];
      }
      p.stat_partial[((long long)b * p.Cout + o) * 2 + 0] = a1;
      p.stat_partial[((long long)b * p.Cout + o) * 2 + 1] = a2;
    }
  }
}

bool c3_fwd_supported(const C3Fwd& p) {
  if (p.H < 1 || p.W < 1 || p.N < 1 || p.Cin < 8 || (p.Cin & 7) || p.Cout < 1) return false;
  if (p.Ho < 1 || p.Wo < 1 || p.Ho > 2 * p.H + 1 || p.Wo > 2 * p.W + 1) return false;
  if ((long long)p.N * ((p.Ho + 1) / 2) * ((p.Wo + 1) / 2) + 64 >= (1ll << 31)) return false;
  return true;
}

bool c3_fwd_fits(const C3Fwd& p, int np) {
  const int KS = (p.Cin + 31) / 32;
  return (size_t)KS * 9 * np * 1024 + 16 * 4 + 8 * 32 * 4 <= (size_t)C3_LDS_BUDGET;
}

void c3_fwd_plan(C3Fwd* p, int np) {
  const int KS = (p->Cin + 31) / 32;
  c3_group_plan((p->Cout + 15) / 16, KS, np, &p->mt, &p->groups);
  const long long cells = (long long)p->N * ((p->Ho + 1) / 2) * ((p->Wo + 1) / 2);
  const long long nsuper = ((cells + 14) / 15 + 1) / 2;
  p->bpg = (int)std::max(1ll, std::min((nsuper + 7) / 8, (long long)std::max(1, 256 / p->groups)));
}

template <int NP, int DT, int ST = ST_F32, int OT = ST_F32>
static int c3_fwd_launch_t(const C3Fwd& p, hipStream_t s) {
  constexpr int WT = 1;  // see c3_fwd_k: two wave tiles per pass spill, and the spilling build was not repeatable
  const int KS = (p.Cin + 31) / 32;
  const size_t lds = (size_t)p.mt * KS * 9 * NP * 1024 + (size_t)p.mt * 16 * 4 + (size_t)8 * p.mt * 32 * 4;
  if (lds > 160 * 1024) return -4;
  auto kern = c3_fwd_k<NP, DT, ST, OT, WT>;
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
  hipLaunchKernelGGL(kern, dim3((unsigned)(p.groups * p.bpg)), dim3(512), lds, s, p);
  return (int)hipGetLastError();
}

int c3_fwd_launch(const C3Fwd& p, int np, int dt, hipStream_t s) {
  if (!c3_fwd_supported(p) || p.mt < 1 || p.mt > C3_MT || p.groups < 1 || p.bpg < 1) return -4;
  if (p.mt * p.groups * 16 < p.Cout) return -4;
  if (p.st == ST_BF16 || p.ot == ST_BF16) {  // bf16 storage on either side = plain bf16 operands
    if (np != 1 || dt != D3_BF16) return -4;
    if (p.st == ST_BF16 && p.ot == ST_BF16) return c3_fwd_launch_t<1, D3_BF16, ST_BF16, ST_BF16>(p, s);
    if (p.st == ST_BF16) return c3_fwd_launch_t<1, D3_BF16, ST_BF16, ST_F32>(p, s);
    return c3_fwd_launch_t<1, D3_BF16, ST_F32, ST_BF16>(p, s);
  }
  if (dt == D3_BF16) {
    if (np == 1) return c3_fwd_launch_t<1, D3_BF16>(p, s);
    if (np == 2) return c3_fwd_launch_t<2, D3_BF16>(p, s);
    if (np == 3) return c3_fwd_launch_t<3, D3_BF16>(p, s);
  } else if (dt == D3_F16) {
    if (np == 1) return c3_fwd_launch_t<1, D3_F16>(p, s);
    if (np == 2) return c3_fwd_launch_t<2, D3_F16>(p, s);
  }
  return -4;
}

// =============================================================================================
// data gradient
//
// Lane = input pixel (15 per wave tile + the next pixel as halo lane 15), M = input channels, K = output channels.  The
// weights of all (<= 5) M tiles do not fit LDS for all taps and K steps at once, so a block walks "steps" = (K step,
// kernel row ky): the 3 kernel-column fragments of every M tile for that step are streamed L2 -> registers -> LDS one
// step ahead (double buffer, one barrier per step) while the waves, in lock step, load the dU row 2y+ky of their pixels
// (one 8-byte load per channel: columns 2x, 2x+1), split it into the kx = 0 / 1 fragments, take kx = 2 from the next
// lane's kx = 0 fragment (DPP) and issue 3 x mt x parts MFMAs per wave tile.  dU is read once per 80 input channels.
// =============================================================================================
constexpr int C3_DMT = 5;
template <int NP, int DT, int YT>
__global__ __launch_bounds__(512, 2) void c3_dgrad_k(const C3Dgrad p) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n16 = lane & 15, kb = lane >> 4;
  const int KS = (p.Cout + 31) >> 5;
  const int S = KS * 3;  // steps per super tile
  const int g = blockIdx.x / p.bpg, b = blockIdx.x - g * p.bpg;
  const int m0 = g * p.mt;
  const int mt = min(p.mt, ((p.C + 15) >> 4) - m0);
  const int STEP = p.mt * 3 * NP * 64;  // uint4 entries per step
  uint4* wbuf = reinterpret_cast<uint4*>(smem);  // [2][STEP]
  const uint4* wsrc = p.wpk + (long long)g * S * STEP;

  const int cells = p.N * p.H * p.W;
  const int HW = p.H * p.W;
  const int ntiles = (cells + 14) / 15;
  const int nsuper = (ntiles + 1) >> 1;
  const int sstride = p.bpg * 8;
  const int rounds = (nsuper + sstride - 1) / sstride;  // every wave of every block runs the same number of rounds
  const long long oplane = (long long)p.Ho * p.Wo;

  struct Cell {
    SP<YT> du;        // dU at (sample, channel 0, row 2y, column 2x)
    float* gp;        // G at (sample, channel 0, y, x)
    unsigned right_mask;
    int y2;
    bool inr, valid;
  };
  auto setup_cell = [&](int T) __attribute__((always_inline)) {
    const int id = T * 15 + n16;
    Cell c;
    c.inr = id < cells;
    c.valid = c.inr && n16 < 15;
    const int idc = min(id, cells - 1);
    const int ns_ = idc / HW;
    const int rem = idc - ns_ * HW;
    const int y = rem / p.W, x = rem - y * p.W;
    c.right_mask = (x + 1 < p.W) ? 0xFFFFFFFFu : 0u;
    c.y2 = 2 * y;
    c.du = SP<YT>(p.dU) + ((long long)ns_ * p.Cout * oplane + 2 * x);
    c.gp = p.G + (long long)ns_ * p.ns + rem;
    return c;
  };

  // weight streaming: thread copies entries tid + 512*i of the step
  constexpr int WCH = (C3_DMT * 3 * NP * 64 + 511) / 512;  // <= 6
  uint4 w0, w1, w2, w3, w4, w5;  // named registers (an array written in one lambda and read in another went to scratch)
  w0 = w1 = w2 = w3 = w4 = w5 = make_uint4(0u, 0u, 0u, 0u);
  auto wissue = [&](int step) __attribute__((always_inline)) {
    const uint4* q = wsrc + (long long)step * STEP;
    w0 = q[min(tid, STEP - 1)];
    if constexpr (WCH > 1) w1 = q[min(tid + 512, STEP - 1)];
    if constexpr (WCH > 2) w2 = q[min(tid + 1024, STEP - 1)];
    if constexpr (WCH > 3) w3 = q[min(tid + 1536, STEP - 1)];
    if constexpr (WCH > 4) w4 = q[min(tid + 2048, STEP - 1)];
    if constexpr (WCH > 5) w5 = q[min(tid + 2560, STEP - 1)];
  };
  auto wcommit = [&](int buf) __attribute__((always_inline)) {
    uint4* d = wbuf + buf * STEP;
    if (tid < STEP) d[tid] = w0;
    if constexpr (WCH > 1) if (tid + 512 < STEP) d[tid + 512] = w1;
    if constexpr (WCH > 2) if (tid + 1024 < STEP) d[tid + 1024] = w2;
    if constexpr (WCH > 3) if (tid + 1536 < STEP) d[tid + 1536] = w3;
    if constexpr (WCH > 4) if (tid + 2048 < STEP) d[tid + 2048] = w4;
    if constexpr (WCH > 5) if (tid + 2560 < STEP) d[tid + 2560] = w5;
  };

  float2 raw[2][8];
  auto issue = [&](const Cell& c, int w, int step) __attribute__((always_inline)) {
    const int ks = step / 3, ky = step - ks * 3;
    const int Y = min(c.y2 + ky, p.Ho - 1);
    const SP<YT> q = c.du + ((long long)min(ks * 32 + kb * 8, p.Cout - 8) * oplane + (long long)Y * p.Wo);
#pragma unroll
    for (int e = 0; e < 8; ++e) raw[w][e] = q.ld2((long long)e * oplane);
  };
  uint4 f0[2][NP], f1[2][NP], f2[2][NP];
  auto convert = [&](const Cell& c, int w, int step) __attribute__((always_inline)) {
    const int ky = step % 3;
    const bool rv = c.inr && (c.y2 + ky < p.Ho);
    unsigned a[4][NP], bq[4][NP];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      split2<DT, NP>(rv ? raw[w][2 * j].x : 0.f, rv ? raw[w][2 * j + 1].x : 0.f, a[j]);
      split2<DT, NP>(rv ? raw[w][2 * j].y : 0.f, rv ? raw[w][2 * j + 1].y : 0.f, bq[j]);
    }
#pragma unroll
    for (int pt = 0; pt < NP; ++pt) {
      f0[w][pt] = make_uint4(a[0][pt], a[1][pt], a[2][pt], a[3][pt]);
      f1[w][pt] = make_uint4(bq[0][pt], bq[1][pt], bq[2][pt], bq[3][pt]);
      f2[w][pt] = and4(make_uint4(row_from_next(a[0][pt]), row_from_next(a[1][pt]), row_from_next(a[2][pt]),
                                  row_from_next(a[3][pt])), c.right_mask);
    }
  };

  f32x4 acc[C3_DMT][2];
  // prologue: weights of step 0 into buffer 0
  wissue(0);
  wcommit(0);
  __syncthreads();
  int gstep = 0;  // global step counter (buffer parity)
  for (int r = 0; r < rounds; ++r) {
    const int ST = b * 8 + wave + r * sstride;
    const bool live = ST < nsuper;
    const int STc = min(ST, nsuper - 1);
    const Cell c0 = setup_cell(2 * STc), c1 = setup_cell(2 * STc + 1);
#pragma unroll
    for (int m = 0; m < C3_DMT; ++m) acc[m][0] = acc[m][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    issue(c0, 0, 0);
    issue(c1, 1, 0);
    for (int step = 0; step < S; ++step, ++gstep) {
      const int buf = gstep & 1;
      wissue(step + 1 < S ? step + 1 : 0);  // unconditional (the very last copy is never read): keeps wreg in registers
      convert(c0, 0, step);
      convert(c1, 1, step);
      __builtin_amdgcn_sched_barrier(0);
      if (step + 1 < S) {
        issue(c0, 0, step + 1);
        issue(c1, 1, step + 1);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < C3_DMT; ++m) {
        if (m < mt) {
          const uint4* wb = wbuf + buf * STEP + (m * 3) * NP * 64 + lane;
          uint4 A[NP];
#define C3_KX(KX_, F_)                                                      \
  _Pragma("unroll") for (int pt = 0; pt < NP; ++pt) A[pt] = wb[(KX_ * NP + pt) * 64]; \
  acc[m][0] = mfma_split<DT, NP>(A, F_[0], acc[m][0]);                      \
  acc[m][1] = mfma_split<DT, NP>(A, F_[1], acc[m][1]);
          C3_KX(0, f0)
          C3_KX(1, f1)
          C3_KX(2, f2)
#undef C3_KX
        }
      }
      wcommit(buf ^ 1);
      __syncthreads();
    }
    // ---- epilogue: G = cscale * acc (overwrite) ----
    mfma_drain();  // wait states between the MFMA chain and the first accumulator read (split16.h)
    if (live) {
      int kb4 = 4 * kb;
      asm volatile("" : "+v"(kb4));
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        const Cell& c = w == 0 ? c0 : c1;
#pragma unroll
        for (int m = 0; m < C3_DMT; ++m) {
          if (m < mt) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              const int ch = (m0 + m) * 16 + kb4 + rr;
              if (c.valid && ch < p.C) {
                const float sc = p.cscale ? p.cscale[ch] : 1.f;
                c.gp[(long long)ch * p.cs] = acc[m][w][rr] * sc;
              }
            }
          }
        }
      }
    }
  }
}

bool c3_dgrad_supported(const C3Dgrad& p) {
  if (p.H < 1 || p.W < 1 || p.N < 1 || p.C < 1 || p.Cout < 8 || (p.Cout & 7)) return false;
  if (p.Wo != 2 * p.W || (p.Ho != 2 * p.H && p.Ho != 2 * p.H + 1)) return false;  // 8-byte column pairs: even cropped width
  if ((reinterpret_cast<uintptr_t>(p.dU) & (p.yt == ST_BF16 ? 3 : 7)) != 0) return false;
  if ((long long)p.N * p.H * p.W + 64 >= (1ll << 31)) return false;
  return true;
}

void c3_dgrad_plan(C3Dgrad* p) {
  const int MT = (p->C + 15) / 16;
  p->mt = c3_dgrad_mt(MT);
  p->groups = (MT + p->mt - 1) / p->mt;
  const long long cells = (long long)p->N * p->H * p->W;
  const long long nsuper = ((cells + 14) / 15 + 1) / 2;
  p->bpg = (int)std::max(1ll, std::min((nsuper + 7) / 8, (long long)std::max(1, 256 / p->groups)));
}

template <int NP, int DT, int YT = ST_F32>
static int c3_dgrad_launch_t(const C3Dgrad& p, hipStream_t s) {
  const size_t lds = (size_t)2 * p.mt * 3 * NP * 1024;
  if (lds > 160 * 1024) return -4;
  auto kern = c3_dgrad_k<NP, DT, YT>;
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
  hipLaunchKernelGGL(kern, dim3((unsigned)(p.groups * p.bpg)), dim3(512), lds, s, p);
  return (int)hipGetLastError();
}

int c3_dgrad_launch(const C3Dgrad& p, int np, int dt, hipStream_t s) {
  if (!c3_dgrad_supported(p) || p.mt < 1 || p.mt > C3_DMT || p.groups < 1 || p.bpg < 1) return -4;
  if (p.mt != c3_dgrad_mt((p.C + 15) / 16) || p.mt * p.groups * 16 < p.C) return -4;
  if (p.yt == ST_BF16) {  // bf16 storage = plain bf16 operands
    if (np != 1 || dt != D3_BF16) return -4;
    return c3_dgrad_launch_t<1, D3_BF16, ST_BF16>(p, s);
  }
  if (dt == D3_BF16) {
    if (np == 1) return c3_dgrad_launch_t<1, D3_BF16>(p, s);
    if (np == 2) return c3_dgrad_launch_t<2, D3_BF16>(p, s);
    if (np == 3) return c3_dgrad_launch_t<3, D3_BF16>(p, s);
  } else if (dt == D3_F16) {
    if (np == 1) return c3_dgrad_launch_t<1, D3_F16>(p, s);
    if (np == 2) return c3_dgrad_launch_t<2, D3_F16>(p, s);
  }
  return -4;
}

// =============================================================================================
// weight gradient
//
// K = input pixels: a K step is four 8-pixel row segments (lane kb owns segment kb; W % 8 == 0, so segments never cross
// rows), numbered linearly over samples x rows x segments.  Block = 3*mo waves: wave (m, ky) owns output-channel tile m
// and kernel row ky, i.e. the three kernel-column accumulators against all (<= 5) input-channel tiles of the block:
//   A: the wave's dU row segment (row 2y+ky, 17 consecutive columns = four 16-byte loads + one) straight from memory,
//      de-interleaved into the kx = 0 / 1 / 2 fragments (even, odd, even+1 columns) -- every dU element is read once;
//   B: the x rows of the K step, staged once per block through LDS as split fragments (double buffer, one barrier per K
//      step), shared by all waves.
// partial[range][c][o][ky][kx] is reduced afterwards in fixed order.
// =============================================================================================
constexpr int C3_WNC = 5;
template <int NP, int DT, int MO, int ST, int YT>
__global__ __launch_bounds__(192 * MO, 3) void c3_wgrad_k(const C3Wgrad p) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nthreads = blockDim.x;
  const int n16 = lane & 15, kb = lane >> 4;
  uint4* bbuf = reinterpret_cast<uint4*>(smem);  // [2][nc][NP][64]
  const int BB = p.nc * NP * 64;

  const int cg = blockIdx.x % p.cgroups;
  const int og = (blockIdx.x / p.cgroups) % p.ogroups;
  const int range = blockIdx.x / (p.cgroups * p.ogroups);
  const int m = wave / 3, ky = wave - m * 3;
  const int otile = og * p.mo + m;
  const bool wave_active = otile * 16 < p.Cout;
  const int NT = (p.Cin + 15) >> 4;
  const int nc = min(p.nc, NT - cg * p.nc);
  const int W8 = p.W >> 3;
  const int SG = p.N * p.H * W8;  // segments
  const int KT = (SG + 3) >> 2;   // K steps
  const int k_begin = range * p.per, k_end = min(KT, k_begin + p.per);
  const long long oplane = (long long)p.Ho * p.Wo;

  // ---- A side ----
  const int o_lane = min(otile * 16 + n16, p.Cout - 1);
  struct ARaw {
    float4 r[4];
    float r16;
    bool ok;
  };
  ARaw arA, arB;  // two K steps in flight
  auto issue_a = [&](int kstep, ARaw& a) __attribute__((always_inline)) {
    const int sg = kstep * 4 + kb;
    const int sgc = min(sg, SG - 1);
    const int ns_ = sgc / (p.H * W8);
    const int rem = sgc - ns_ * (p.H * W8);
    const int y = rem / W8, x0 = (rem - y * W8) * 8;
    const int Y = 2 * y + ky;
    a.ok = sg < SG && Y < p.Ho;
    const SP<YT> q = SP<YT>(p.dU) + (((long long)ns_ * p.Cout + o_lane) * oplane + (long long)min(Y, p.Ho - 1) * p.Wo + 2 * x0);
#pragma unroll
    for (int i = 0; i < 4; ++i) a.r[i] = q.ld4(4 * i);
    a.r16 = (2 * x0 + 16 < p.Wo) ? q.ld1(16) : 0.f;
  };
  uint4 af[3][NP];
  auto convert_a = [&](const ARaw& a) __attribute__((always_inline)) {
    float u[17];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u[4 * i + 0] = a.ok ? a.r[i].x : 0.f;
      u[4 * i + 1] = a.ok ? a.r[i].y : 0.f;
      u[4 * i + 2] = a.ok ? a.r[i].z : 0.f;
      u[4 * i + 3] = a.ok ? a.r[i].w : 0.f;
    }
    u[16] = a.ok ? a.r16 : 0.f;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      unsigned w[4][NP];
#pragma unroll
      for (int j = 0; j < 4; ++j) split2<DT, NP>(u[4 * j + kx], u[4 * j + 2 + kx], w[j]);
#pragma unroll
      for (int pt = 0; pt < NP; ++pt) af[kx][pt] = make_uint4(w[0][pt], w[1][pt], w[2][pt], w[3][pt]);
    }
  };

  // ---- B side: thread t stages the float4 (channel c = t / 8, quarter q = t % 8 -> segment q / 2, half q % 2) ----
  constexpr int BCH = MO == 2 ? 2 : 4;  // float4 per thread: 80 channels x 8 quarters over 384 / 192 threads
  float4 br[BCH];
  auto issue_b = [&](int kstep) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < BCH; ++i) {
      const int t = tid + i * nthreads;
      const int cl = min(t >> 3, nc * 16 - 1), q = t & 7;
      const int c = min(cg * p.nc * 16 + cl, p.Cin - 1);
      const int sg = min(kstep * 4 + (q >> 1), SG - 1);
      const int ns_ = sg / (p.H * W8);
      const int rem = sg - ns_ * (p.H * W8);  // = y * W8 + x8  ->  pixel offset 8 * rem
      br[i] = SP<ST>(p.X).ld4((long long)ns_ * p.ns + (long long)c * p.cs + (long long)rem * 8 + 4 * (q & 1));
    }
  };
  auto commit_b = [&](int kstep, int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < BCH; ++i) {
      const int t = tid + i * nthreads;
      const int cl = t >> 3, q = t & 7;
      if (cl < nc * 16) {
        const bool ok = (cg * p.nc * 16 + cl < p.Cin) && (kstep * 4 + (q >> 1) < SG);
        unsigned h0[NP], h1[NP];
        split2<DT, NP>(ok ? br[i].x : 0.f, ok ? br[i].y : 0.f, h0);
        split2<DT, NP>(ok ? br[i].z : 0.f, ok ? br[i].w : 0.f, h1);
#pragma unroll
        for (int pt = 0; pt < NP; ++pt) {
          uint2* d = reinterpret_cast<uint2*>(bbuf + buf * BB + ((cl >> 4) * NP + pt) * 64 + (cl & 15) + 16 * (q >> 1));
          d[q & 1] = make_uint2(h0[pt], h1[pt]);
        }
      }
    }
  };

  f32x4 acc[3][C3_WNC];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int j = 0; j < C3_WNC; ++j) acc[kx][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (k_begin < k_end) {
    issue_b(k_begin);
    issue_a(k_begin, arA);
    if (k_begin + 1 < k_end) issue_a(k_begin + 1, arB);
    commit_b(k_begin, 0);
  }
  __syncthreads();
  auto step = [&](int k, ARaw& a) __attribute__((always_inline)) {
    const int buf = (k - k_begin) & 1;
    const bool more = k + 1 < k_end;
    if (more) issue_b(k + 1);
    convert_a(a);
    __builtin_amdgcn_sched_barrier(0);
    if (k + 2 < k_end) issue_a(k + 2, a);
    __builtin_amdgcn_sched_barrier(0);
    if (wave_active) {
#pragma unroll
      for (int j = 0; j < C3_WNC; ++j) {
        if (j < nc) {
          uint4 B[NP];
#pragma unroll
          for (int pt = 0; pt < NP; ++pt) B[pt] = bbuf[buf * BB + (j * NP + pt) * 64 + lane];
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) acc[kx][j] = mfma_split<DT, NP>(af[kx], B, acc[kx][j]);
        }
      }
    }
    if (more) commit_b(k + 1, buf ^ 1);
    __syncthreads();
  };
  for (int k = k_begin; k < k_end; k += 2) {
    step(k, arA);
    if (k + 1 < k_end) step(k + 1, arB);
  }
mfma_drain();  // wait states between the MFMA chain and the first accumulator read (split16.h)
  // ---- store: acc[kx][j][r] = dW[c = 16*(cg*nc+j) + n16][o = 16*otile + 4*kb + r][ky][kx] ----
  if (wave_active) {
    float* dst = p.partial + (long long)range * p.Cin * p.Cout * 9;
#pragma unroll
    for (int j = 0; j < C3_WNC; ++j) {
      if (j < nc) {
        const int c = (cg * p.nc + j) * 16 + n16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = otile * 16 + 4 * kb + r;
          if (c < p.Cin && o < p.Cout) {
            float* d = dst + ((long long)c * p.Cout + o) * 9 + ky * 3;
            d[0] = acc[0][j][r];
            d[1] = acc[1][j][r];
            d[2] = acc[2][j][r];
          }
        }
      }
    }
  }
}

bool c3_wgrad_supported(const C3Wgrad& p) {
  if (p.H < 1 || p.W < 8 || (p.W & 7) || p.N < 1 || p.Cin < 1 || p.Cout < 1) return false;
  if (p.Wo != 2 * p.W || (p.Ho != 2 * p.H && p.Ho != 2 * p.H + 1)) return false;
  if ((p.ns & 3) || (p.cs & 3) || (reinterpret_cast<uintptr_t>(p.X) & (p.st == ST_BF16 ? 7 : 15)) ||
      (reinterpret_cast<uintptr_t>(p.dU) & (p.yt == ST_BF16 ? 7 : 15)))
    return false;
  if ((long long)p.N * p.H * p.W + 64 >= (1ll << 31)) return false;
  return true;
}

void c3_wgrad_plan(C3Wgrad* p) {
  const int MT = (p->Cout + 15) / 16, NT = (p->Cin + 15) / 16;
  p->ogroups = (MT + 1) / 2;
  p->mo = (MT + p->ogroups - 1) / p->ogroups;
  p->cgroups = (NT + C3_WNC - 1) / C3_WNC;
  p->nc = (NT + p->cgroups - 1) / p->cgroups;
  const long long ksteps = std::max(1ll, ((long long)p->N * p->H * (p->W / 8) + 3) / 4);
  long long nr = std::max(1ll, std::min(ksteps, (long long)std::max(1, 768 / (p->ogroups * p->cgroups))));
  nr = std::min(nr, 256ll);
  const long long per = (ksteps + nr - 1) / nr;
  p->per = (int)per;
  p->nranges = (int)((ksteps + per - 1) / per);
}

template <int NP, int DT, int ST = ST_F32, int YT = ST_F32>
static int c3_wgrad_launch_t(const C3Wgrad& p, hipStream_t s) {
  const size_t lds = (size_t)2 * p.nc * NP * 1024;
  const dim3 grid((unsigned)(p.nranges * p.ogroups * p.cgroups));
  if (p.nc * 16 * 8 > 768) return -4;  // staging budget of the block's threads
  if (p.mo == 2) hipLaunchKernelGGL((c3_wgrad_k<NP, DT, 2, ST, YT>), grid, dim3(384), lds, s, p);
  else hipLaunchKernelGGL((c3_wgrad_k<NP, DT, 1, ST, YT>), grid, dim3(192), lds, s, p);
  return (int)hipGetLastError();
}

int c3_wgrad_launch(const C3Wgrad& p, int np, int dt, hipStream_t s) {
  if (!c3_wgrad_supported(p) || p.mo < 1 || p.mo > 2 || p.nc < 1 || p.nc > C3_WNC || p.ogroups < 1 || p.cgroups < 1 ||
      p.nranges < 1 || p.per < 1)
    return -4;
  if (p.mo * p.ogroups * 16 < p.Cout || p.nc * p.cgroups * 16 < p.Cin) return -4;
  if (p.st == ST_BF16 || p.yt == ST_BF16) {  // bf16 storage = plain bf16 operands
    if (np != 1 || dt != D3_BF16) return -4;
    if (p.st == ST_BF16 && p.yt == ST_BF16) return c3_wgrad_launch_t<1, D3_BF16, ST_BF16, ST_BF16>(p, s);
    if (p.st == ST_BF16) return c3_wgrad_launch_t<1, D3_BF16, ST_BF16, ST_F32>(p, s);
    return c3_wgrad_launch_t<1, D3_BF16, ST_F32, ST_BF16>(p, s);
  }
  if (dt == D3_BF16) {
    if (np == 1) return c3_wgrad_launch_t<1, D3_BF16>(p, s);
    if (np == 2) return c3_wgrad_launch_t<2, D3_BF16>(p, s);
    if (np == 3) return c3_wgrad_launch_t<3, D3_BF16>(p, s);
  } else if (dt == D3_F16) {
    if (np == 1) return c3_wgrad_launch_t<1, D3_F16>(p, s);
    if (np == 2) return c3_wgrad_launch_t<2, D3_F16>(p, s);
  }
  return -4;
}

}  // namespace rln
