// TransitionDown 1x1 kernels on the 16-bit MFMA pipe with split fp32 operands (see pw1.h).
#include "pw1.h"

#include <algorithm>
#include <cstdio>

#include "split16.h"
#include "storage.h"

namespace rln {

constexpr int P1_LDS_BUDGET = 150 * 1024;

// =============================================================================================
// weight packing
// =============================================================================================
template <int DT, int NP>
__global__ __launch_bounds__(256) void p1_pack_k(const float* __restrict__ params, const P1PackDesc* __restrict__ desc,
                                                 int n_desc, int total_units, uint4* __restrict__ packed) {
  const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (unit >= total_units) return;
  int d = 0;
  while (d + 1 < n_desc && desc[d + 1].unit_begin <= unit) ++d;
  const P1PackDesc q = desc[d];
  int u = unit - q.unit_begin;
  const int nf = q.wf_off >= 0 ? ((q.cout + 15) >> 4) * ((q.cin + 31) >> 5) : 0;
  const bool backward = u >= nf;
  if (backward) {
    u -= nf;
    if (q.wb_off < 0) return;
  }
  const int KS = backward ? (q.cout + 31) >> 5 : (q.cin + 31) >> 5;
  const int mtile = u / KS, ks = u - mtile * KS;
  const int i = lane & 15, kb = lane >> 4;
  const float* w = params + q.w_off;
  const int row = mtile * 16 + i;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = ks * 32 + kb * 8 + e;
    float val = 0.f;
    if (!backward) {
      if (row < q.cout && k < q.cin) val = sat16<DT>(w[(long long)row * q.cin + k] * w_prescale<DT>());  // split16.h
    } else {
      if (row < q.cin && k < q.cout) val = w[(long long)k * q.cin + row];
    }
    v[e] = val;
  }
  unsigned parts[4][NP];
#pragma unroll
  for (int j = 0; j < 4; ++j) split2<DT, NP>(v[2 * j], v[2 * j + 1], parts[j]);
  uint4* dst = packed + (backward ? q.wb_off : q.wf_off) + ((long long)(mtile * KS + ks) * NP) * 64 + lane;
#pragma unroll
  for (int p = 0; p < NP; ++p) dst[p * 64] = make_uint4(parts[0][p], parts[1][p], parts[2][p], parts[3][p]);
}

int p1_pack_weights(const float* params, const P1PackDesc* desc_dev, int n_desc, int total_units, uint4* packed, int np,
                    int dt, hipStream_t s) {
  if (total_units <= 0) return 0;
  dim3 grid((unsigned)((total_units + 3) / 4));
#define P1_PACK(DT_, NP_)                                                                                         \
  hipLaunchKernelGGL((p1_pack_k<DT_, NP_>), grid, dim3(256), 0, s, params, desc_dev, n_desc, total_units, packed)
  if (dt == D3_BF16) {
    if (np == 1) P1_PACK(D3_BF16, 1);
    else if (np == 2) P1_PACK(D3_BF16, 2);
    else P1_PACK(D3_BF16, 3);
  } else {
    if (np == 1) P1_PACK(D3_F16, 1);
    else if (np == 2) P1_PACK(D3_F16, 2);
    else P1_PACK(D3_F16, 3);
  }
#undef P1_PACK
  return (int)hipGetLastError();
}

// M tiles per block such that the block's weight fragments fit in LDS, balanced over the groups
static void p1_group_plan(int m_tiles, int ksteps, int np, int extra_bytes_per_mtile, int* mt, int* groups) {
  const int per = ksteps * np * 1024 + extra_bytes_per_mtile;
  int cap = (P1_LDS_BUDGET - 2 * ksteps * 32 * 4) / per;
  cap = std::max(1, std::min(8, cap));
  *groups = (m_tiles + cap - 1) / cap;
  *mt = (m_tiles + *groups - 1) / *groups;
}

// =============================================================================================
// forward
//
// Block = 8 waves, persistent: holds the weight fragments of `mt` M tiles (16 output channels each) in LDS and walks wave
// tiles of 16 pooling windows.  Per 32-channel K step a lane loads the 2x2 pixels of its window for its 8 channels (two
// 8-byte loads per channel), applies BN + ReLU, splits into 16-bit parts (these registers are the B fragments of the four
// window positions) and multiplies them with every M tile; the next K step's loads (or the first of the next wave tile)
// are issued before the MFMAs.  Epilogue: (acc + bias) * scale, maximum over the four positions (first maximum wins, as
// MaxPool2d), pooled value and argmax stored, per-channel sums of the pooled map reduced over the 16 windows with DPP
// and accumulated in the wave's own LDS slots (fixed order: deterministic).
// =============================================================================================
template <int NP, int DT, int ST, int OT>
__global__ __launch_bounds__(512, 2) void p1_fwd_k(const P1Fwd p) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n16 = lane & 15, kb = lane >> 4;
  const int KS = (p.Cin + 31) >> 5;
  const int g = blockIdx.x / p.bpg, b = blockIdx.x - g * p.bpg;
  const int m0 = g * p.mt;
  const int mt = min(p.mt, ((p.Cout + 15) >> 4) - m0);
  uint4* wl = reinterpret_cast<uint4*>(smem);                                         // [mt][KS][NP][64]
  float* abtab = reinterpret_cast<float*>(smem + (size_t)p.mt * KS * NP * 1024);       // [2][KS*32]
  float* slot = abtab + 2 * KS * 32;                                                   // [8][p.mt*16][2]
  float* btab = slot + 8 * p.mt * 32;                                                  // [p.mt*16] bias
  {
    const uint4* src = p.wpk + (long long)m0 * KS * NP * 64;
    const int cnt = mt * KS * NP * 64;
    for (int i = tid; i < cnt; i += 512) wl[i] = src[i];
    for (int i = tid; i < KS * 32; i += 512) {
      abtab[i] = i < p.Cin ? p.pa[i] : 0.f;
      abtab[KS * 32 + i] = i < p.Cin ? p.pb[i] : 0.f;
    }
    for (int i = tid; i < 8 * p.mt * 32; i += 512) slot[i] = 0.f;
    for (int i = tid; i < p.mt * 16; i += 512) btab[i] = (p.bias && m0 * 16 + i < p.Cout) ? p.bias[m0 * 16 + i] : 0.f;
  }
  __syncthreads();

  const int PH = p.H >> 1, PW = p.W >> 1, PP = PH * PW;
  const int total = p.N * PP;
  const int ntiles = (total + 15) >> 4;
  const int tstride = p.bpg * 8;

  struct Tile {
    SP<ST> base;
    int ns_, poff;
    bool valid;
  };
  auto setup_tile = [&](int T) __attribute__((always_inline)) {
    int v = T * 16 + n16;
    Tile t;
    t.valid = v < total;
    if (!t.valid) v = total - 1;
    t.ns_ = v / PP;
    t.poff = v - t.ns_ * PP;
    const int wy = t.poff / PW, wx = t.poff - wy * PW;
    t.base = SP<ST>(p.S) + ((long long)t.ns_ * p.ns + (long long)(2 * wy) * p.W + 2 * wx);
    return t;
  };

  typename SRaw<ST>::r2 q0[8], q1[8];  // raw window rows (narrow until convert widens them)
  auto issue = [&](const SP<ST> base, int ks) __attribute__((always_inline)) {
    // Cin % 8 == 0: an 8-channel block is all-in or all-out; blocks past Cin re-read the last block (their folded
    // scale / shift in abtab is zero, so they contribute relu(0) = 0)
    const SP<ST> q = base + (long long)min(ks * 32 + kb * 8, p.Cin - 8) * p.cs;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      q0[e] = q.raw2((long long)e * p.cs);
      q1[e] = q.raw2((long long)e * p.cs + p.W);
    }
  };
  bool dbg_const = false;
  uint4 bf[4][NP];
  auto convert = [&](int ks) __attribute__((always_inline)) {
    const float4* ap = reinterpret_cast<const float4*>(abtab + ks * 32 + kb * 8);
    const float4* bp = reinterpret_cast<const float4*>(abtab + KS * 32 + ks * 32 + kb * 8);
    const float4 a0 = ap[0], a1 = ap[1], b0 = bp[0], b1 = bp[1];
    const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    float z[4][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float2 r0 = dbg_const ? make_float2(0.5f, -0.25f) : SRaw<ST>::w2(q0[e]);
      const float2 r1 = dbg_const ? make_float2(0.5f, -0.25f) : SRaw<ST>::w2(q1[e]);
      z[0][e] = relu16<DT>(fmaf(av[e], r0.x, bv[e]));
      z[1][e] = relu16<DT>(fmaf(av[e], r0.y, bv[e]));
      z[2][e] = relu16<DT>(fmaf(av[e], r1.x, bv[e]));
      z[3][e] = relu16<DT>(fmaf(av[e], r1.y, bv[e]));
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      unsigned w[4][NP];
#pragma unroll
      for (int j = 0; j < 4; ++j) split2<DT, NP>(z[t][2 * j], z[t][2 * j + 1], w[j]);
#pragma unroll
      for (int pt = 0; pt < NP; ++pt) bf[t][pt] = make_uint4(w[0][pt], w[1][pt], w[2][pt], w[3][pt]);
    }
  };

#ifdef RLN_DIAG
  const int dbg = p.dbg;
#else
  constexpr int dbg = 0;
#endif
  if (dbg & 1) dbg_const = true;
  const bool sc_vec = (reinterpret_cast<uintptr_t>(p.nscale) & 15) == 0;
  f32x4 acc[8][4];
  int T = b * 8 + wave;
  Tile cur = setup_tile(min(T, ntiles - 1));
  if (T < ntiles && !(dbg & 1)) issue(cur.base, 0);
  while (T < ntiles) {
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int Tn = T + tstride;
    Tile nxt = cur;
    for (int ks = 0; ks < KS; ++ks) {
      convert(ks);
      __builtin_amdgcn_sched_barrier(0);
      if (ks + 1 < KS) {
        if (!(dbg & 1)) issue(cur.base, ks + 1);
      } else if (Tn < ntiles) {
        nxt = setup_tile(Tn);
        if (!(dbg & 1)) issue(nxt.base, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (dbg & 2) {
        asm volatile("" ::"v"(bf[0][0].x), "v"(bf[1][0].y), "v"(bf[2][0].z), "v"(bf[3][0].w));
        continue;
      }
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        if (m < mt) {
          uint4 A[NP];
#pragma unroll
          for (int pt = 0; pt < NP; ++pt) A[pt] = wl[((m * KS + ks) * NP + pt) * 64 + lane];
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[m][t] = mfma_split<DT, NP>(A, bf[t], acc[m][t]);
        }
      }
    }
    // ---- epilogue ----
    mfma_drain();  // wait states between the MFMA chain and the first accumulator read (split16.h)
    int kb4 = 4 * kb;
    asm volatile("" : "+v"(kb4));  // per-iteration opaque: keeps the 64 channel addresses out of loop-invariant registers
    if (dbg & 4) {
#pragma unroll
      for (int m = 0; m < 8; ++m) asm volatile("" ::"v"(acc[m][0][0]), "v"(acc[m][1][1]), "v"(acc[m][2][2]), "v"(acc[m][3][3]));
      T = Tn;
      cur = nxt;
      continue;
    }
    // Dropout2d scales of the lane's sample for all its channels first (one 16-byte load per M tile, issued together):
    // loads placed between the stores below could not be hoisted over them
    float4 sc4[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      sc4[m] = make_float4(1.f, 1.f, 1.f, 1.f);
      if (m < mt && p.nscale != nullptr) {
        const float* q = p.nscale + (long long)cur.ns_ * p.Cout + min(m0 * 16 + m * 16 + kb4, p.Cout - 4);
        if (sc_vec) sc4[m] = *reinterpret_cast<const float4*>(q);
        else sc4[m] = make_float4(q[0], q[1], q[2], q[3]);
      }
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (m < mt) {
        const float4 b4 = *reinterpret_cast<const float4*>(btab + m * 16 + kb4);
        const float bia[4] = {b4.x, b4.y, b4.z, b4.w};
        const float sca[4] = {sc4[m].x, sc4[m].y, sc4[m].z, sc4[m].w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ol = m * 16 + kb4 + r;
          const int o = m0 * 16 + ol;
          const bool ov = o < p.Cout;
          const float bias = bia[r];
          const float sc = sca[r];
          float best = fmaf(acc[m][0][r], w_unscale<DT>(), bias) * sc;
          int bi = 0;
#pragma unroll
          for (int t = 1; t < 4; ++t) {
            const float v = fmaf(acc[m][t][r], w_unscale<DT>(), bias) * sc;
            if (v > best) {
              best = v;
              bi = t;
            }
          }
          const bool st = ov && cur.valid;
          best = st_round<OT>(best);  // the statistics describe the pooled map as it is stored
          if (st) {
            SP<OT>(p.out).st1((long long)cur.ns_ * p.out_ns + (long long)o * p.out_cs + cur.poff, best);
            p.pool_idx[((long long)cur.ns_ * p.Cout + o) * PP + cur.poff] = (unsigned char)bi;
          }
          const float s1 = row16_sum(st ? best : 0.f);
          const float s2 = row16_sum(st ? best * best : 0.f);
          if (n16 == 0) {  // LDS float adds without return: nothing to wait for; one wave per slot, issue order = sum order
            float* sl = slot + ((wave * p.mt * 16) + ol) * 2;
            lds_add_f32(sl, s1);
            lds_add_f32(sl + 1, s2);
          }
        }
      }
    }
    T = Tn;
    cur = nxt;
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

static inline bool al8(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 7) == 0; }

bool p1_fwd_supported(const P1Fwd& p) {
  if (p.H < 2 || p.W < 2 || (p.W & 1) || (p.ns & 1) || (p.cs & 1)) return false;
  if (reinterpret_cast<uintptr_t>(p.S) & (p.st == ST_BF16 ? 3 : 7)) return false;  // a pair of elements per load
  if (p.Cin < 8 || (p.Cin & 7) || p.Cout < 4 || (p.Cout & 3) || p.N < 1) return false;
  if ((long long)p.N * (p.H / 2) * (p.W / 2) + 16 >= (1ll << 31) || (long long)8 * p.cs + p.W >= (1ll << 31)) return false;
  return true;
}

void p1_fwd_plan(P1Fwd* p, int np) {
  const int KS = (p->Cin + 31) / 32;
  p1_group_plan((p->Cout + 15) / 16, KS, np, 8 * 16 * 2 * 4 + 16 * 4, &p->mt, &p->groups);
  const long long total = (long long)p->N * (p->H / 2) * (p->W / 2);
  const long long ntiles = (total + 15) / 16;
  p->bpg = (int)std::max(1ll, std::min((ntiles + 7) / 8, (long long)std::max(1, 256 / p->groups)));
}

template <int NP, int DT, int ST = ST_F32, int OT = ST_F32>
static int p1_fwd_launch_t(const P1Fwd& p, hipStream_t s) {
  const int KS = (p.Cin + 31) / 32;
  const size_t lds = (size_t)p.mt * KS * NP * 1024 + (size_t)2 * KS * 32 * 4 + (size_t)8 * p.mt * 32 * 4 +
                     (size_t)p.mt * 16 * 4;
  if (lds > 160 * 1024) return -4;
  auto kern = p1_fwd_k<NP, DT, ST, OT>;
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

int p1_fwd_launch(const P1Fwd& p, int np, int dt, hipStream_t s) {
  if (!p1_fwd_supported(p) || p.mt < 1 || p.mt > 8 || p.groups < 1 || p.bpg < 1) return -4;
  if (p.mt * p.groups * 16 < p.Cout) return -4;
  if (p.st == ST_BF16 || p.ot == ST_BF16) {  // bf16 storage on either side = plain bf16 operands
    if (np != 1 || dt != D3_BF16) return -4;
    if (p.st == ST_BF16 && p.ot == ST_BF16) return p1_fwd_launch_t<1, D3_BF16, ST_BF16, ST_BF16>(p, s);
    if (p.st == ST_BF16) return p1_fwd_launch_t<1, D3_BF16, ST_BF16, ST_F32>(p, s);
    return p1_fwd_launch_t<1, D3_BF16, ST_F32, ST_BF16>(p, s);
  }
  if (dt == D3_BF16) {
    if (np == 1) return p1_fwd_launch_t<1, D3_BF16>(p, s);
    if (np == 2) return p1_fwd_launch_t<2, D3_BF16>(p, s);
    if (np == 3) return p1_fwd_launch_t<3, D3_BF16>(p, s);
  } else if (dt == D3_F16) {
    if (np == 1) return p1_fwd_launch_t<1, D3_F16>(p, s);
    if (np == 2) return p1_fwd_launch_t<2, D3_F16>(p, s);
  }
  return -4;
}

// =============================================================================================
// data gradient
//
// Same skeleton as the forward with M = input channels, K = output channels.  The B operand is the un-pooled gradient:
// a lane loads the pooled value and the argmax byte of its window for its 8 output channels, splits the value once and
// masks it into the window position the index names (the other three positions are zero), so the sparse pre-pool map is
// never written to memory.  Epilogue per input channel: ReLU mask from S, BatchNorm-backward sums (sum gz, sum gz*xhat;
// DPP row sums into the wave's LDS slots), G (+)= gamma * gz.  Odd H: the last row has no window (its gz is zero); it is
// covered by an extra window row whose second pixel row lies outside the image.
// =============================================================================================
template <int NP, int DT, int ST, int YT>
__global__ __launch_bounds__(512, 2) void p1_dgrad_k(const P1Dgrad p) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n16 = lane & 15, kb = lane >> 4;
  const int KS = (p.Cout + 31) >> 5;
  const int g = blockIdx.x / p.bpg, b = blockIdx.x - g * p.bpg;
  const int m0 = g * p.mt;
  const int mt = min(p.mt, ((p.C + 15) >> 4) - m0);
  uint4* wl = reinterpret_cast<uint4*>(smem);                                      // [mt][KS][NP][64]
  float* chtab = reinterpret_cast<float*>(smem + (size_t)p.mt * KS * NP * 1024);    // [p.mt*16][8]
  float* slot = chtab + p.mt * 16 * 8;                                              // [8][p.mt*16][2]
  {
    const uint4* src = p.wpk + (long long)m0 * KS * NP * 64;
    const int cnt = mt * KS * NP * 64;
    for (int i = tid; i < cnt; i += 512) wl[i] = src[i];
    for (int i = tid; i < p.mt * 16; i += 512) {
      const int c = m0 * 16 + i;
      const bool cv = c < p.C;
      float* t = chtab + i * 8;
      t[0] = cv ? p.ea[c] : 0.f;
      t[1] = cv ? p.eb[c] : 0.f;
      t[2] = cv ? p.mean[c] : 0.f;
      t[3] = cv ? p.invstd[c] : 0.f;
      t[4] = cv ? p.egamma[c] : 0.f;
      t[5] = (cv && c >= p.acc_lo && c < p.acc_hi) ? 1.f : 0.f;
      t[6] = 0.f;
      t[7] = 0.f;
    }
    for (int i = tid; i < 8 * p.mt * 32; i += 512) slot[i] = 0.f;
  }
  __syncthreads();

  const int PH = p.H >> 1, PW = p.W >> 1, PP = PH * PW;
  const int WR = (p.H + 1) >> 1, WP = WR * PW;
  const int total = p.N * WP;
  const int ntiles = (total + 15) >> 4;
  const int tstride = p.bpg * 8;

  struct Tile {
    SP<YT> gbase;                // dYp at (sample, channel 0, window)
    const unsigned char* ibase;  // pool_idx, same
    long long pix;               // pixel offset of the window's first row inside the sample (S / G views)
    bool valid, win, row1;       // lane has a window slot / the window exists (pooled) / its second pixel row exists
  };
  auto setup_tile = [&](int T) __attribute__((always_inline)) {
    int v = T * 16 + n16;
    Tile t;
    t.valid = v < total;
    if (!t.valid) v = total - 1;
    const int ns_ = v / WP;
    const int rem = v - ns_ * WP;
    const int wy = rem / PW, wx = rem - wy * PW;
    t.win = t.valid && wy < PH;
    t.row1 = 2 * wy + 1 < p.H;
    const int poff = min(wy, PH - 1) * PW + wx;
    t.gbase = SP<YT>(p.dYp) + ((long long)ns_ * p.Cout * PP + poff);
    t.ibase = p.pool_idx + (long long)ns_ * p.Cout * PP + poff;
    t.pix = (long long)ns_ * p.ns + (long long)(2 * wy) * p.W + 2 * wx;
    return t;
  };

  typename SRaw<YT>::r1 gvr[8];
  unsigned char iv[8];
  auto issue = [&](const Tile& t, int ks) __attribute__((always_inline)) {
    const long long off = (long long)min(ks * 32 + kb * 8, p.Cout - 8) * PP;  // Cout % 8 == 0; weights past Cout are zero
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      gvr[e] = t.gbase.raw1(off + (long long)e * PP);
      iv[e] = t.ibase[off + (long long)e * PP];
    }
  };
  uint4 bf[4][NP];
  auto convert = [&](bool win) __attribute__((always_inline)) {
    unsigned w[4][NP];
    float gv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) gv[e] = SRaw<YT>::w1(gvr[e]);
#pragma unroll
    for (int j = 0; j < 4; ++j) split2<DT, NP>(win ? gv[2 * j] : 0.f, win ? gv[2 * j + 1] : 0.f, w[j]);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      unsigned mk[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        mk[j] = ((int)iv[2 * j] == t ? 0x0000FFFFu : 0u) | ((int)iv[2 * j + 1] == t ? 0xFFFF0000u : 0u);
#pragma unroll
      for (int pt = 0; pt < NP; ++pt)
        bf[t][pt] = make_uint4(w[0][pt] & mk[0], w[1][pt] & mk[1], w[2][pt] & mk[2], w[3][pt] & mk[3]);
    }
  };

  f32x4 acc[8][4];
  int T = b * 8 + wave;
  Tile cur = setup_tile(min(T, ntiles - 1));
  if (T < ntiles) issue(cur, 0);
  while (T < ntiles) {
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int Tn = T + tstride;
    Tile nxt = cur;
    for (int ks = 0; ks < KS; ++ks) {
      convert(cur.win);
      __builtin_amdgcn_sched_barrier(0);
      if (ks + 1 < KS) {
        issue(cur, ks + 1);
      } else if (Tn < ntiles) {
        nxt = setup_tile(Tn);
        issue(nxt, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        if (m < mt) {
          uint4 A[NP];
#pragma unroll
          for (int pt = 0; pt < NP; ++pt) A[pt] = wl[((m * KS + ks) * NP + pt) * 64 + lane];
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[m][t] = mfma_split<DT, NP>(A, bf[t], acc[m][t]);
        }
      }
    }
    // ---- epilogue ----
    mfma_drain();  // wait states between the MFMA chain and the first accumulator read (split16.h)
    int kb4 = 4 * kb;
    asm volatile("" : "+v"(kb4));  // per-iteration opaque: keeps the channel addresses out of loop-invariant registers
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (m < mt) {
        float2 s0[4], s1[4], g0[4], g1[4];
        bool cvr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int cl = m * 16 + kb4 + r;
          const int c = m0 * 16 + cl;
          cvr[r] = c < p.C && cur.valid;
          const long long off = cur.pix + (long long)min(c, p.C - 1) * p.cs;
          const long long off1 = cur.row1 ? off + p.W : off;
          s0[r] = SP<ST>(p.S).ld2(off);
          s1[r] = SP<ST>(p.S).ld2(off1);
          g0[r] = *reinterpret_cast<const float2*>(p.G + off);
          g1[r] = *reinterpret_cast<const float2*>(p.G + off1);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int cl = m * 16 + kb4 + r;
          const int c = m0 * 16 + cl;
          const float4 t0 = *reinterpret_cast<const float4*>(chtab + cl * 8);
          const float2 t1 = *reinterpret_cast<const float2*>(chtab + cl * 8 + 4);
          const float ea = t0.x, eb = t0.y, mean = t0.z, is = t0.w, egam = t1.x;
          const bool accum = t1.y != 0.f;
          const float sv[4] = {s0[r].x, s0[r].y, s1[r].x, s1[r].y};
          const float go[4] = {g0[r].x, g0[r].y, g1[r].x, g1[r].y};
          float ov[4];
          float a1 = 0.f, a2 = 0.f;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const bool pv = cvr[r] && (t < 2 || cur.row1);
            const float yv = fmaf(ea, sv[t], eb);
            const float gz = (pv && yv > 0.f) ? acc[m][t][r] : 0.f;
            const float xh = (sv[t] - mean) * is;
            a1 += gz;
            a2 += gz * xh;
            ov[t] = fmaf(egam, gz, accum ? go[t] : 0.f);
          }
          if (cvr[r]) {
            const long long off = cur.pix + (long long)c * p.cs;
            *reinterpret_cast<float2*>(p.G + off) = make_float2(ov[0], ov[1]);
            if (cur.row1) *reinterpret_cast<float2*>(p.G + off + p.W) = make_float2(ov[2], ov[3]);
          }
          a1 = row16_sum(a1);
          a2 = row16_sum(a2);
          if (n16 == 0) {
            float* sl = slot + ((wave * p.mt * 16) + cl) * 2;
            lds_add_f32(sl, a1);
            lds_add_f32(sl + 1, a2);
          }
        }
      }
    }
    T = Tn;
    cur = nxt;
  }
  __syncthreads();
  if (p.stat_partial != nullptr && tid < mt * 16) {
    const int c = m0 * 16 + tid;
    if (c < p.C) {
      float a1 = 0.f, a2 = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) {
        a1 += slot[((w * p.mt * 16) + tid) * 2 + 0];
        a2 += slot[((w * p.mt * 16) + tid) * 2 + 1];
      }
      p.stat_partial[((long long)b * p.C + c) * 2 + 0] = a1;
      p.stat_partial[((long long)b * p.C + c) * 2 + 1] = a2;
    }
  }
}

bool p1_dgrad_supported(const P1Dgrad& p) {
  if (p.H < 2 || p.W < 2 || (p.W & 1) || (p.ns & 1) || (p.cs & 1) || !al8(p.G)) return false;
  if (reinterpret_cast<uintptr_t>(p.S) & (p.st == ST_BF16 ? 3 : 7)) return false;
  if (p.C < 1 || p.Cout < 8 || (p.Cout & 7) || p.N < 1) return false;
  if ((long long)p.N * ((p.H + 1) / 2) * (p.W / 2) + 16 >= (1ll << 31)) return false;
  return true;
}

void p1_dgrad_plan(P1Dgrad* p, int np) {
  const int KS = (p->Cout + 31) / 32;
  p1_group_plan((p->C + 15) / 16, KS, np, 16 * 8 * 4 + 8 * 16 * 2 * 4, &p->mt, &p->groups);
  const long long total = (long long)p->N * ((p->H + 1) / 2) * (p->W / 2);
  const long long ntiles = (total + 15) / 16;
  p->bpg = (int)std::max(1ll, std::min((ntiles + 7) / 8, (long long)std::max(1, 256 / p->groups)));
}

template <int NP, int DT, int ST = ST_F32, int YT = ST_F32>
static int p1_dgrad_launch_t(const P1Dgrad& p, hipStream_t s) {
  const int KS = (p.Cout + 31) / 32;
  const size_t lds = (size_t)p.mt * KS * NP * 1024 + (size_t)p.mt * 16 * 8 * 4 + (size_t)8 * p.mt * 32 * 4;
  if (lds > 160 * 1024) return -4;
  auto kern = p1_dgrad_k<NP, DT, ST, YT>;
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

int p1_dgrad_launch(const P1Dgrad& p, int np, int dt, hipStream_t s) {
  if (!p1_dgrad_supported(p) || p.mt < 1 || p.mt > 8 || p.groups < 1 || p.bpg < 1) return -4;
  if (p.mt * p.groups * 16 < p.C) return -4;
  if ((p.st == ST_BF16 || p.yt == ST_BF16) && dt != D3_BF16) return -4;
  if (dt == D3_BF16) {
    if (p.st == ST_BF16 || p.yt == ST_BF16) {  // bf16 storage = plain bf16 operands
      if (np != 1) return -4;
      if (p.st == ST_BF16 && p.yt == ST_BF16) return p1_dgrad_launch_t<1, D3_BF16, ST_BF16, ST_BF16>(p, s);
      if (p.st == ST_BF16) return p1_dgrad_launch_t<1, D3_BF16, ST_BF16, ST_F32>(p, s);
      return p1_dgrad_launch_t<1, D3_BF16, ST_F32, ST_BF16>(p, s);
    }
    if (np == 1) return p1_dgrad_launch_t<1, D3_BF16>(p, s);
    if (np == 2) return p1_dgrad_launch_t<2, D3_BF16>(p, s);
    if (np == 3) return p1_dgrad_launch_t<3, D3_BF16>(p, s);
  } else if (dt == D3_F16) {
    if (np == 1) return p1_dgrad_launch_t<1, D3_F16>(p, s);
    if (np == 2) return p1_dgrad_launch_t<2, D3_F16>(p, s);
  }
  return -4;
}

// =============================================================================================
// weight gradient
//
// K = pixels.  A K step is 8 pooling windows x 4 positions; lane (i = l&15, kb = l>>4) holds windows 2kb, 2kb+1.
//   A[o][k] = un-pooled gradient: expanded from (pooled value, argmax byte) -- staged through LDS once per block and
//             shared by its 4 waves (two K steps = one 16-window "slab" per barrier, double buffered);
//   B[k][c] = relu(a*S + b): each wave owns two 16-channel N tiles and loads its lanes' window pixels straight from the
//             NCHW planes (two 8-byte loads per window), so every activation is read once per output-channel group.
// Block = (K range, output-channel group of <= 8 M tiles, 128-input-channel block); accumulators stay in registers over
// the whole range, partial[range][o][c] is reduced afterwards (fixed order).  Slabs never straddle samples; windows past
// the end of a sample's window plane carry zero gradient.
// =============================================================================================
template <int NP, int DT, bool V4, int ST, int YT>
__global__ __launch_bounds__(256, 2) void p1_wgrad_k(const P1Wgrad p) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n16 = lane & 15, kb = lane >> 4;
  uint4* abuf = reinterpret_cast<uint4*>(smem);  // [2 buffers][2 ksteps][8 m][NP][64]
  constexpr int ABUF = 2 * 8 * NP * 64;          // uint4 units per buffer

  const int cb = blockIdx.x % p.cblocks;
  const int og = (blockIdx.x / p.cblocks) % p.ogroups;
  const int range = blockIdx.x / (p.cblocks * p.ogroups);
  const int MTtot = (p.Cout + 15) >> 4;
  const int mo = min(p.mo, MTtot - og * p.mo);
  const int PH = p.H >> 1, PW = p.W >> 1, PP = PH * PW;
  const int SL = (PP + 15) >> 4;
  const int slab_begin = range * p.per, slab_end = min(p.N * SL, slab_begin + p.per);

  // B side: this wave's two N tiles
  int cch[2];
  float ba[2], bb[2];
  SP<ST> sb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = (cb * 8 + 2 * wave + i) * 16 + n16;
    cch[i] = c;
    const int cc = min(c, p.Cin - 1);
    ba[i] = c < p.Cin ? p.pa[cc] : 0.f;
    bb[i] = c < p.Cin ? p.pb[cc] : 0.f;
    sb[i] = SP<ST>(p.S) + (long long)cc * p.cs;
  }
  const bool wave_active = (cb * 8 + 2 * wave) * 16 < p.Cin;

  // V4 (even window-row length, 16-byte aligned rows): a lane's two windows are horizontal neighbours, one 16-byte load
  // per (tile, row, K step) and the two K steps of a slab consume each 128-byte line back to back.
  typedef float2 RawSet[2][2][2][2];  // [kstep][tile][window][row]
  RawSet rawA, rawB;                  // two slabs in flight
  auto issue_b = [&](int slab, RawSet& raw) __attribute__((always_inline)) {
    const int ns_ = slab / SL;
    const int w0 = (slab - ns_ * SL) * 16;
    if constexpr (V4) {
      long long off[2];
#pragma unroll
      for (int j2 = 0; j2 < 2; ++j2) {
        const int w = min(w0 + j2 * 8 + 2 * kb, PP - 2);
        const int wy = w / PW, wx = w - wy * PW;
        off[j2] = (long long)ns_ * p.ns + (long long)(2 * wy) * p.W + 2 * wx;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int row = 0; row < 2; ++row)
#pragma unroll
          for (int j2 = 0; j2 < 2; ++j2) {
            const float4 v = sb[i].ld4(off[j2] + row * p.W);
            raw[j2][i][0][row] = make_float2(v.x, v.y);
            raw[j2][i][1][row] = make_float2(v.z, v.w);
          }
    } else {
#pragma unroll
      for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int w = min(w0 + j2 * 8 + 2 * kb + j, PP - 1);
          const int wy = w / PW, wx = w - wy * PW;
          const long long off = (long long)ns_ * p.ns + (long long)(2 * wy) * p.W + 2 * wx;
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            raw[j2][i][j][0] = sb[i].ld2(off);
            raw[j2][i][j][1] = sb[i].ld2(off + p.W);
          }
        }
    }
  };
  uint4 bfr[2][2][NP];
  auto convert_b = [&](const RawSet& raw) __attribute__((always_inline)) {
#pragma unroll
    for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        unsigned w[4][NP];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float z0 = fmaxf(fmaf(ba[i], raw[j2][i][j][0].x, bb[i]), 0.f);
          const float z1 = fmaxf(fmaf(ba[i], raw[j2][i][j][0].y, bb[i]), 0.f);
          const float z2 = fmaxf(fmaf(ba[i], raw[j2][i][j][1].x, bb[i]), 0.f);
          const float z3 = fmaxf(fmaf(ba[i], raw[j2][i][j][1].y, bb[i]), 0.f);
          split2<DT, NP>(z0, z1, w[2 * j]);
          split2<DT, NP>(z2, z3, w[2 * j + 1]);
        }
#pragma unroll
        for (int pt = 0; pt < NP; ++pt) bfr[j2][i][pt] = make_uint4(w[0][pt], w[1][pt], w[2][pt], w[3][pt]);
      }
  };

  // A side: thread builds entries (kstep j2, m, lane l) for e = tid + 256*q, q = 0..3 : l = e & 63, m = (e >> 6) & 7, j2 = e >> 9
  float ag[4][2];
  unsigned char ai[4][2];
  auto issue_a = [&](int slab) __attribute__((always_inline)) {
    const int ns_ = slab / SL;
    const int w0 = (slab - ns_ * SL) * 16;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = tid + 256 * q;
      const int l = e & 63, m = (e >> 6) & 7, j2 = e >> 9;
      const int o = min((og * p.mo + m) * 16 + (l & 15), p.Cout - 1);
      const long long base = ((long long)ns_ * p.Cout + o) * PP;
      if constexpr (V4) {  // PP even: the window pair is 8-byte (values) / 2-byte (indices) aligned
        const int w = min(w0 + j2 * 8 + 2 * (l >> 4), PP - 2);
        const float2 g2 = SP<YT>(p.dYp).ld2(base + w);
        const uchar2 i2 = *reinterpret_cast<const uchar2*>(p.pool_idx + base + w);
        ag[q][0] = g2.x;
        ag[q][1] = g2.y;
        ai[q][0] = i2.x;
        ai[q][1] = i2.y;
      } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int w = min(w0 + j2 * 8 + 2 * (l >> 4) + j, PP - 1);
          ag[q][j] = SP<YT>(p.dYp).ld1(base + w);
          ai[q][j] = p.pool_idx[base + w];
        }
      }
    }
  };
  auto commit_a = [&](int slab, int buf) __attribute__((always_inline)) {
    const int ns_ = slab / SL;
    const int w0 = (slab - ns_ * SL) * 16;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = tid + 256 * q;
      const int l = e & 63, m = (e >> 6) & 7, j2 = e >> 9;
      const int wbase = w0 + j2 * 8 + 2 * (l >> 4);
      const bool ov = (og * p.mo + m) * 16 + (l & 15) < p.Cout && m < mo;
      const float g0 = (ov && wbase < PP) ? ag[q][0] : 0.f;
      const float g1 = (ov && wbase + 1 < PP) ? ag[q][1] : 0.f;
      unsigned h[NP];
      split2<DT, NP>(g0, g1, h);
      const int i0 = ai[q][0], i1 = ai[q][1];
#pragma unroll
      for (int pt = 0; pt < NP; ++pt) {
        const unsigned lo = h[pt] & 0xFFFFu, hi = h[pt] >> 16;
        uint4 v;
        v.x = (i0 == 0 ? lo : 0u) | (i0 == 1 ? lo << 16 : 0u);
        v.y = (i0 == 2 ? lo : 0u) | (i0 == 3 ? lo << 16 : 0u);
        v.z = (i1 == 0 ? hi : 0u) | (i1 == 1 ? hi << 16 : 0u);
        v.w = (i1 == 2 ? hi : 0u) | (i1 == 3 ? hi << 16 : 0u);
        abuf[buf * ABUF + ((j2 * 8 + m) * NP + pt) * 64 + l] = v;
      }
    }
  };

  f32x4 acc[8][2];
#pragma unroll
  for (int m = 0; m < 8; ++m)
#pragma unroll
    for (int i = 0; i < 2; ++i) acc[m][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (slab_begin < slab_end) {
    issue_a(slab_begin);
    issue_b(slab_begin, rawA);
    if (slab_begin + 1 < slab_end) issue_b(slab_begin + 1, rawB);
    commit_a(slab_begin, 0);
  }
  __syncthreads();
  // one slab: A(s+1) values requested, B(s) converted, B(s+2) requested into the registers just freed, MFMAs of slab s,
  // A(s+1) committed to the other buffer, barrier
  auto step = [&](int s, RawSet& raw) __attribute__((always_inline)) {
    const int buf = (s - slab_begin) & 1;
    const bool more = s + 1 < slab_end;
    if (more) issue_a(s + 1);
    convert_b(raw);
    __builtin_amdgcn_sched_barrier(0);
    if (s + 2 < slab_end) issue_b(s + 2, raw);
    __builtin_amdgcn_sched_barrier(0);
    if (wave_active) {
#pragma unroll
      for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          if (m < mo) {
            uint4 A[NP];
#pragma unroll
            for (int pt = 0; pt < NP; ++pt) A[pt] = abuf[buf * ABUF + ((j2 * 8 + m) * NP + pt) * 64 + lane];
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[m][i] = mfma_split<DT, NP>(A, bfr[j2][i], acc[m][i]);
          }
        }
    }
    if (more) commit_a(s + 1, buf ^ 1);
    __syncthreads();
  };
  for (int s = slab_begin; s < slab_end; s += 2) {
    step(s, rawA);
    if (s + 1 < slab_end) step(s + 1, rawB);
  }
mfma_drain();  // wait states between the MFMA chain and the first accumulator read (split16.h)
  // ---- store: acc[m][i][r] = dW[o = 16*(og*mo+m) + 4*kb + r][c = cch[i]] ----
  float* dst = p.partial + (long long)range * p.Cout * p.Cin;
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    if (m < mo) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = (og * p.mo + m) * 16 + 4 * kb + r;
        if (o < p.Cout) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
            if (cch[i] < p.Cin) dst[(long long)o * p.Cin + cch[i]] = acc[m][i][r];
        }
      }
    }
  }
}

bool p1_wgrad_supported(const P1Wgrad& p) {
  if (p.H < 2 || p.W < 2 || (p.W & 1) || (p.ns & 1) || (p.cs & 1)) return false;
  if (reinterpret_cast<uintptr_t>(p.S) & (p.st == ST_BF16 ? 3 : 7)) return false;
  if (p.Cin < 1 || p.Cout < 1 || p.N < 1) return false;
  if ((long long)p.N * (p.H / 2) * (p.W / 2) + 16 >= (1ll << 31)) return false;
  return true;
}

void p1_wgrad_plan(P1Wgrad* p) {
  const int MTtot = (p->Cout + 15) / 16;
  p->ogroups = (MTtot + 7) / 8;
  p->mo = (MTtot + p->ogroups - 1) / p->ogroups;
  p->cblocks = ((p->Cin + 15) / 16 + 7) / 8;
  const int PP = (p->H / 2) * (p->W / 2);
  const long long slabs = (long long)p->N * ((PP + 15) / 16);
  long long nr = std::max(1ll, std::min(slabs, (long long)std::max(1, 512 / (p->ogroups * p->cblocks))));
  const long long per = (slabs + nr - 1) / nr;
  p->per = (int)per;
  p->nranges = (int)((slabs + per - 1) / per);
}

template <int NP, int DT, bool V4, int ST, int YT>
static int p1_wgrad_launch_v(const P1Wgrad& p, hipStream_t s) {
  const size_t lds = (size_t)2 * 2 * 8 * NP * 1024;
  auto kern = p1_wgrad_k<NP, DT, V4, ST, YT>;
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
  hipLaunchKernelGGL(kern, dim3((unsigned)(p.nranges * p.ogroups * p.cblocks)), dim3(256), lds, s, p);
  return (int)hipGetLastError();
}

template <int NP, int DT, int ST = ST_F32, int YT = ST_F32>
static int p1_wgrad_launch_t(const P1Wgrad& p, hipStream_t s) {
  const int PW = p.W / 2, PP = (p.H / 2) * PW;
  const bool v4 = (PW % 2) == 0 && (PP % 2) == 0 && PP >= 2 && (p.ns % 4) == 0 && (p.cs % 4) == 0 &&
                  (reinterpret_cast<uintptr_t>(p.S) & (4 * st_bytes(ST) - 1)) == 0 &&
                  (reinterpret_cast<uintptr_t>(p.dYp) & (2 * st_bytes(YT) - 1)) == 0 &&
                  (reinterpret_cast<uintptr_t>(p.pool_idx) & 1) == 0;
  return v4 ? p1_wgrad_launch_v<NP, DT, true, ST, YT>(p, s) : p1_wgrad_launch_v<NP, DT, false, ST, YT>(p, s);
}

int p1_wgrad_launch(const P1Wgrad& p, int np, int dt, hipStream_t s) {
  if (!p1_wgrad_supported(p) || p.mo < 1 || p.mo > 8 || p.ogroups < 1 || p.cblocks < 1 || p.nranges < 1 || p.per < 1)
    return -4;
  if (p.mo * p.ogroups * 16 < p.Cout || p.cblocks * 128 < p.Cin) return -4;
  if (p.st == ST_BF16 || p.yt == ST_BF16) {  // bf16 storage = plain bf16 operands
    if (np != 1 || dt != D3_BF16) return -4;
    if (p.st == ST_BF16 && p.yt == ST_BF16) return p1_wgrad_launch_t<1, D3_BF16, ST_BF16, ST_BF16>(p, s);
    if (p.st == ST_BF16) return p1_wgrad_launch_t<1, D3_BF16, ST_BF16, ST_F32>(p, s);
    return p1_wgrad_launch_t<1, D3_BF16, ST_F32, ST_BF16>(p, s);
  }
  if (dt == D3_BF16) {
    if (np == 1) return p1_wgrad_launch_t<1, D3_BF16>(p, s);
    if (np == 2) return p1_wgrad_launch_t<2, D3_BF16>(p, s);
    if (np == 3) return p1_wgrad_launch_t<3, D3_BF16>(p, s);
  } else if (dt == D3_F16) {
    if (np == 1) return p1_wgrad_launch_t<1, D3_F16>(p, s);
    if (np == 2) return p1_wgrad_launch_t<2, D3_F16>(p, s);
  }
  return -4;
}

}  // namespace rln
