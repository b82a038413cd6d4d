// TransitionUp (ConvTranspose2d 3x3, stride 2) kernels on the 16-bit MFMA pipe with split fp32 operands (see ct3.h).
#include "ct3.h"

#include <algorithm>
#include <cstdio>

#include "split16.h"

namespace rln {

constexpr int C3_LDS_BUDGET = 150 * 1024;
constexpr int C3_MT = 2;  // M tiles per block at most (LDS: 9 taps x K steps x parts per tile)

__device__ __forceinline__ void lds_addf(float* p, float v) {
  (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
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
      if (row < q.cout && k < q.cin) val = w[((long long)k * q.cout + row) * 9 + tap];
    } else {  // row = c, k = o
      if (row < q.cin && k < q.cout) val = w[((long long)row * q.cout + k) * 9 + tap];
    }
    v[e] = val;
  }
  unsigned parts[4][NP];
#pragma unroll
  for (int j = 0; j < 4; ++j) split2<DT, NP>(v[2 * j], v[2 * j + 1], parts[j]);
  uint4* dst = packed + (backward ? q.wb_off : q.wf_off) + ((long long)u * NP) * 64 + lane;
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
template <int NP, int DT>
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
  const int nsuper = (ntiles + 1) >> 1;
  const int sstride = p.bpg * 8;
  const bool vec2 = ((p.Wo | p.out_cs) & 1) == 0 && (p.out_ns & 1) == 0 && (reinterpret_cast<uintptr_t>(p.out) & 7) == 0;

  struct Cell {
    const float* own;
    const float* up;
    float* outp;          // out at (sample, channel 0, row 2y, column 2x)
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
    const float* xb = p.X + (long long)ns_ * p.ns + min(x, p.W - 1);
    c.own = xb + (long long)min(y, p.H - 1) * p.W;
    c.up = xb + (long long)max(min(y, p.H) - 1, 0) * p.W;
    c.oy = 2 * y;
    c.ox = 2 * x;
    c.outp = p.out + (long long)ns_ * p.out_ns + (long long)c.oy * p.Wo + c.ox;
    return c;
  };

  float ro[2][8], ru[2][8];
  auto issue = [&](const Cell& c, int w, int ks) __attribute__((always_inline)) {
    const long long off = (long long)min(ks * 32 + kb * 8, p.Cin - 8) * p.cs;  // Cin % 8 == 0; weights past Cin are zero
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      ro[w][e] = c.own[off + (long long)e * p.cs];
      ru[w][e] = c.up[off + (long long)e * p.cs];
    }
  };
  uint4 fo[2][NP], fu[2][NP], fl[2][NP], ful[2][NP];
  auto convert = [&](const Cell& c, int w) __attribute__((always_inline)) {
    unsigned a[4][NP], u[4][NP];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      split2<DT, NP>(c.v_own ? ro[w][2 * j] : 0.f, c.v_own ? ro[w][2 * j + 1] : 0.f, a[j]);
      split2<DT, NP>(c.v_up ? ru[w][2 * j] : 0.f, c.v_up ? ru[w][2 * j + 1] : 0.f, u[j]);
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

  f32x4 acc[C3_MT][2][4];
  int ST = b * 8 + wave;
  Cell c0 = setup_cell(2 * min(ST, nsuper - 1)), c1 = setup_cell(2 * min(ST, nsuper - 1) + 1);
  if (ST < nsuper) {
    issue(c0, 0, 0);
    issue(c1, 1, 0);
  }
  while (ST < nsuper) {
#pragma unroll
    for (int m = 0; m < C3_MT; ++m)
#pragma unroll
      for (int w = 0; w < 2; ++w)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[m][w][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int STn = ST + sstride;
    Cell n0 = c0, n1 = c1;
    for (int ks = 0; ks < KS; ++ks) {
      convert(c0, 0);
      convert(c1, 1);
      __builtin_amdgcn_sched_barrier(0);
      if (ks + 1 < KS) {
        issue(c0, 0, ks + 1);
        issue(c1, 1, ks + 1);
      } else if (STn < nsuper) {
        n0 = setup_cell(2 * STn);
        n1 = setup_cell(2 * STn + 1);
        issue(n0, 0, 0);
        issue(n1, 1, 0);
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
  acc[m][1][P_] = mfma_split<DT, NP>(A, F_[1], acc[m][1][P_]);
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
    int kb4 = 4 * kb;
    asm volatile("" : "+v"(kb4));
#pragma unroll
    for (int w = 0; w < 2; ++w) {
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
            for (int q = 0; q < 4; ++q) v[q] = acc[m][w][q][r] + bia[r];
            float* dst = c.outp + (long long)o * p.out_cs;
            const bool m00 = ov && r0v && x0v, m01 = ov && r0v && x1v, m10 = ov && r1v && x0v, m11 = ov && r1v && x1v;
            if (vec2) {  // Wo even: a block column pair is all-in or all-out
              if (m00) *reinterpret_cast<float2*>(dst) = make_float2(v[0], v[1]);
              if (m10) *reinterpret_cast<float2*>(dst + p.Wo) = make_float2(v[2], v[3]);
            } else {
              if (m00) dst[0] = v[0];
              if (m01) dst[1] = v[1];
              if (m10) dst[p.Wo] = v[2];
              if (m11) dst[p.Wo + 1] = v[3];
            }
            float s1 = (m00 ? v[0] : 0.f) + (m01 ? v[1] : 0.f) + (m10 ? v[2] : 0.f) + (m11 ? v[3] : 0.f);
            float s2 = (m00 ? v[0] * v[0] : 0.f) + (m01 ? v[1] * v[1] : 0.f) + (m10 ? v[2] * v[2] : 0.f) +
                       (m11 ? v[3] * v[3] : 0.f);
            s1 = row16_sum(s1);
            s2 = row16_sum(s2);
            if (n16 == 0) {
              float* sl = slot + ((wave * p.mt * 16) + ol) * 2;
              lds_addf(sl, s1);
              lds_addf(sl + 1, s2);
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

void c3_fwd_plan(C3Fwd* p, int np) {
  const int KS = (p->Cin + 31) / 32;
  c3_group_plan((p->Cout + 15) / 16, KS, np, &p->mt, &p->groups);
  const long long cells = (long long)p->N * ((p->Ho + 1) / 2) * ((p->Wo + 1) / 2);
  const long long nsuper = ((cells + 14) / 15 + 1) / 2;
  p->bpg = (int)std::max(1ll, std::min((nsuper + 7) / 8, (long long)std::max(1, 256 / p->groups)));
}

template <int NP, int DT>
static int c3_fwd_launch_t(const C3Fwd& p, hipStream_t s) {
  const int KS = (p.Cin + 31) / 32;
  const size_t lds = (size_t)p.mt * KS * 9 * NP * 1024 + (size_t)p.mt * 16 * 4 + (size_t)8 * p.mt * 32 * 4;
  if (lds > 160 * 1024) return -4;
  auto kern = c3_fwd_k<NP, DT>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    (void)hipGetLastError();
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(p.groups * p.bpg)), dim3(512), lds, s, p);
  return (int)hipGetLastError();
}

int c3_fwd_launch(const C3Fwd& p, int np, int dt, hipStream_t s) {
  if (!c3_fwd_supported(p) || p.mt < 1 || p.mt > C3_MT || p.groups < 1 || p.bpg < 1) return -4;
  if (p.mt * p.groups * 16 < p.Cout) return -4;
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

}  // namespace rln
