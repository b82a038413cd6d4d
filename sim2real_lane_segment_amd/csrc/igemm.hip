// Implicit-GEMM convolution kernels for gfx950 on v_mfma_f32_16x16x4_f32 (see igemm.h).
//
// Block = 256 threads = 4 waves.  A block owns one TH x TW pixel tile (256 or 128 pixels =
// 16-pixel M-tiles, MPW per wave) of one sample and NT*16 output channels.  The K loop walks the
// input channels 16 at a time: the tile (+halo) of those channels is staged into LDS with the
// BatchNorm affine + ReLU applied on the way (zero padding is applied AFTER the activation, as
// Conv2d(padding=1) sees it), the matching weight slab is staged in MFMA B-operand order, then
// every wave issues MPW*NT MFMAs per (4-channel group, tap).
//
// LDS layout: activations [16 ch][CHS] with CHS % 32 == 16 so that the A-operand read of a wave
// (16 consecutive pixels x 4 channels) touches 64 distinct banks; weights [kgroup][tap][ntile][64]
// so that the B-operand read is lane-linear.
#include "igemm.h"
#include "storage.h"

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

namespace rln {

// V4: the tile image is 16-byte aligned in x (it starts 4 columns left of the tile and is TW+8 wide for 3x3), so
// global loads are dwordx4 and LDS commits are ds_write_b128 (needs W % 4 == 0 and 16-byte aligned rows).
template <int KS, int NT, int PRO, int EPI, int TH, int TW, bool V4 = false>
struct IgCfg {
  static constexpr int CK = 16;
  static constexpr bool S2D = (PRO == PRO_S2D);
  static constexpr int HALO = KS - 1;
  static constexpr int XPAD = (V4 && KS == 3) ? 4 : HALO / 2;  // columns staged left of the tile
  static constexpr int PITCH = S2D ? (TW + 1) : (V4 ? (TW + (KS == 3 ? 8 : 0)) : (TW + HALO));
  static constexpr int ROWS = S2D ? (TH + 1) : (TH + HALO);
  static constexpr int PLANE = PITCH * ROWS;
  static constexpr int POS = PLANE * (S2D ? 4 : 1);
  static constexpr int CHS = round_mod32(POS, 16);
  static constexpr int NPOS = cdiv(POS, 256);
  static constexpr int NQ = cdiv(CK * POS / 4, 256);  // V4: float4 loads per thread and chunk
  static constexpr int NS = KS * KS;
  static constexpr int IN_FLOATS = CK * CHS;
  static constexpr int W_FLOATS = (CK / 4) * NS * NT * 64;
  static constexpr int WE = 16 * NT * CK * NS;
  static constexpr int NWE = cdiv(WE, 256);
  static constexpr int OLS = TH * TW + 1;
  static constexpr int POOL_FLOATS = (EPI == EPI_POOL) ? 16 * OLS : 0;
  static constexpr int MAIN_FLOATS = cmax(IN_FLOATS + W_FLOATS, POOL_FLOATS);
  static constexpr int RED_FLOATS = 4 * NT * 16 * 2;
  static constexpr int AB_MAX = 1024;  // V4 + BN: per-channel (a, b) table of the whole layer lives in LDS
  static constexpr int AB_FLOATS = (V4 && PRO == PRO_BNRELU) ? 2 * AB_MAX : 0;
  static constexpr int LDS_BYTES = (MAIN_FLOATS + RED_FLOATS + AB_FLOATS) * 4;
  static constexpr int MPW = TH * TW / 64;
  __host__ __device__ static constexpr int slot_off(int s) {
    return S2D ? ((((s / 3) & 1) * 2 + ((s % 3) & 1)) * PLANE + ((s / 3) >> 1) * PITCH + ((s % 3) >> 1))
               : ((s / KS) * PITCH + (s % KS) + (XPAD - HALO / 2));
  }
};

// Straight-line MFMA block for one staged 16-channel chunk: every LDS offset is a compile-time constant relative
// to one per-lane base, so reads shared by several (M-tile, tap) pairs are issued once and carry immediates.
// first active tap at or after s (NS when none)
template <unsigned long long MASK, int NS>
__host__ __device__ constexpr int ig_next_tap(int s) {
  return (s >= NS) ? NS : (((MASK >> s) & 1ull) ? s : ig_next_tap<MASK, NS>(s + 1));
}

template <unsigned long long MASK>
__host__ __device__ constexpr int ig_tap_rank(int s) {  // active taps below s
  return s <= 0 ? 0 : (int)((MASK >> (s - 1)) & 1ull) + ig_tap_rank<MASK>(s - 1);
}

template <typename C, int NT, int TW, unsigned long long MASK>
__device__ __forceinline__ void igemm_compute(const float* __restrict__ zbase, const float* __restrict__ wlane,
                                              f32x4 (&acc)[C::MPW][NT]) {
  // One 4-channel group per (rolled) iteration keeps the live LDS fragments small; inside, straight-line code.  The
  // fragments of tap s+1 are read BEFORE the MFMAs of tap s (two register sets, order pinned with sched_barrier): left
  // to itself the compiler puts each read group next to its use behind an lgkmcnt(0), i.e. ~8 exposed LDS latencies
  // per 36 MFMAs -- hidden at 2+ waves per SIMD, 1.7x the matrix time when a wave has its SIMD to itself.
  constexpr int S0 = ig_next_tap<MASK, C::NS>(0);
#pragma unroll 1
  for (int kg = 0; kg < 4; ++kg) {
    const float* zk = zbase + kg * 4 * C::CHS;
    const float* wk = wlane + kg * C::NS * NT * 64;
    float bw[2][NT], av[2][C::MPW];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bw[0][nt] = wk[(S0 * NT + nt) * 64];
#pragma unroll
    for (int m = 0; m < C::MPW; ++m) av[0][m] = zk[((m * 16) / TW) * C::PITCH + (m * 16) % TW + C::slot_off(S0)];
#pragma unroll
    for (int s = 0; s < C::NS; ++s) {
      if (!((MASK >> s) & 1ull)) continue;
      const int cur = ig_tap_rank<MASK>(s) & 1;
      const int s1 = ig_next_tap<MASK, C::NS>(s + 1);
      if (s1 < C::NS) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bw[cur ^ 1][nt] = wk[(s1 * NT + nt) * 64];
#pragma unroll
        for (int m = 0; m < C::MPW; ++m)
          av[cur ^ 1][m] = zk[((m * 16) / TW) * C::PITCH + (m * 16) % TW + C::slot_off(s1)];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < C::MPW; ++m) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[m][nt] = mfma16(av[cur][m], bw[cur][nt], acc[m][nt]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// ConvTranspose2d(k3, s2) forward, all four output-parity classes from ONE staged input tile: out[2y+py][2x+px]
// takes the taps (ky, kx) of matching parity, ky = py + 2*(1 - sy) for the input rows y - 1 + sy, sy in {0, 1}
// (sy = 0 only when py = 0); likewise in x.  Nine MFMAs per (pixel tile, 4-channel group), four shifted reads.
// The per-class accumulation order equals the one-class-per-block form (slots ascending), so results are
// bit-identical to it.  Weight slab slot = kernel tap (TM_ID staging).
template <typename C, int TW>
__device__ __forceinline__ void igemm_compute_convt4(const float* __restrict__ zbase, const float* __restrict__ wlane,
                                                     f32x4 (&acc)[4][C::MPW][1]) {
#pragma unroll 1
  for (int kg = 0; kg < 4; ++kg) {
    const float* zk = zbase + kg * 4 * C::CHS;
    const float* wk = wlane + kg * C::NS * 64;
    float bw[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) bw[t] = wk[t * 64];
#pragma unroll
    for (int m = 0; m < C::MPW; ++m) {
      const float* zm = zk + ((m * 16) / TW) * C::PITCH + (m * 16) % TW;
      const float a00 = zm[C::slot_off(0)], a01 = zm[C::slot_off(1)], a10 = zm[C::slot_off(3)], a11 = zm[C::slot_off(4)];
      acc[0][m][0] = mfma16(a00, bw[8], acc[0][m][0]);
      acc[0][m][0] = mfma16(a01, bw[6], acc[0][m][0]);
      acc[0][m][0] = mfma16(a10, bw[2], acc[0][m][0]);
      acc[0][m][0] = mfma16(a11, bw[0], acc[0][m][0]);
      acc[1][m][0] = mfma16(a01, bw[7], acc[1][m][0]);
      acc[1][m][0] = mfma16(a11, bw[1], acc[1][m][0]);
      acc[2][m][0] = mfma16(a10, bw[5], acc[2][m][0]);
      acc[2][m][0] = mfma16(a11, bw[3], acc[2][m][0]);
      acc[3][m][0] = mfma16(a11, bw[4], acc[3][m][0]);
    }
  }
}

// CLS: 0 plain convolution; 1 ConvTranspose2d forward, one output-parity class per block (grid.x = 4 * tiles);
//      2 ConvTranspose2d forward, all four classes per block (NT = 1)
template <int KS, int NT, int PRO, int EPI, int TH, int TW, int CLS, bool V4>
__global__ __launch_bounds__(256, (((NT == 1 && TH * TW <= 256 && CLS != 2) || (KS == 1 && EPI == EPI_POOL)) ? 4
                                   : ((NT == 1 && TH * TW <= 320) ? 3 : 2)))
    void igemm_k(const IgemmParams p) {
  using C = IgCfg<KS, NT, PRO, EPI, TH, TW, V4>;
  static_assert(!V4 || (PRO != PRO_S2D && C::CHS % 4 == 0 && C::PITCH % 4 == 0), "V4 needs 16-byte aligned LDS rows");
  constexpr int MPW = C::MPW;
  static_assert(TW % 16 == 0 && (TH * TW) % 64 == 0, "tile must be whole M-tiles per wave");
  extern __shared__ __align__(16) float smem[];
  float* zl = smem;
  float* wl = smem + C::IN_FLOATS;
  float* red = smem + C::MAIN_FLOATS;
  float* abl = smem + C::MAIN_FLOATS + C::RED_FLOATS;  // [2][AB_MAX] (V4 + BN only)

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, lj = lane & 15, lk = lane >> 4;
  if constexpr (C::AB_FLOATS > 0) {  // read by the commit phases; the first __syncthreads orders it
    for (int c = tid; c < p.K; c += 256) {
      abl[c] = p.pa[c];
      abl[C::AB_MAX + c] = p.pb[c];
    }
  }
  // logical (x, y, sample) block coordinates: dispatch order remapped so that one XCD walks neighbouring tiles
  unsigned lx = blockIdx.x, ly = blockIdx.y, lz = blockIdx.z;
  if (!(p.dbg & 32)) {
    const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    unsigned l = xcd_logical_block(lin, gridDim.x * gridDim.y * gridDim.z);
    lz = l / (gridDim.x * gridDim.y);
    l -= lz * (gridDim.x * gridDim.y);
    ly = l / gridDim.x;
    lx = l - ly * gridDim.x;
  }
  const int bx0 = (int)lx;
  int bx = bx0;
  const int tiles = p.tiles_x * p.tiles_y;
  int cls = 0;
  if constexpr (CLS == 1) {
    cls = bx / tiles;
    bx -= cls * tiles;
  }
  const int tile_y = bx / p.tiles_x, tile_x = bx - tile_y * p.tiles_x;
  const int gy0 = tile_y * TH, gx0 = tile_x * TW;
  const int n = (int)lz;
  int ky = 0, jy = (int)ly;
  if (p.ksplit > 1) {
    const int jgroups = (p.J + NT * 16 - 1) / (NT * 16);
    ky = (int)ly / jgroups;
    jy = (int)ly - ky * jgroups;
  }
  const int jbase = jy * (NT * 16);
  const int py = cls >> 1, px = cls & 1;

  // ---- per-thread staging positions (independent of the channel) ----
  constexpr int NPOSX = V4 ? 1 : C::NPOS;
  int goff[NPOSX], loff[NPOSX];
#pragma unroll
  for (int i = 0; i < NPOSX; ++i) {
    const int e = tid + 256 * i;
    goff[i] = -1;
    loff[i] = -1;
    if (!V4 && e < C::POS) {
      int iy, ix;
      if constexpr (C::S2D) {
        const int pl = e / C::PLANE, rem = e - pl * C::PLANE;
        const int r = rem / C::PITCH, col = rem - r * C::PITCH;
        iy = 2 * (gy0 + r) + (pl >> 1);
        ix = 2 * (gx0 + col) + (pl & 1);
      } else {
        const int r = e / C::PITCH, col = e - r * C::PITCH;
        iy = gy0 - C::HALO / 2 + r;
        ix = gx0 - C::HALO / 2 + col;
      }
      loff[i] = e;
      if (iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win) {
        if (p.pk_sx > 1) {  // packed samples: virtual column -> (sample in group, column); gap columns stay zero
          const int sidx = ix / (p.pk_w + 1), x = ix - sidx * (p.pk_w + 1);
          if (x < p.pk_w && (int)lz * p.pk_sx + sidx < p.pk_n) goff[i] = sidx * (int)p.in_ns + iy * p.pk_w + x;
        } else {
          goff[i] = iy * p.Win + ix;
        }
      }
    }
  }
  static_assert((MPW * 16) % TW == 0, "a wave's M-tiles must start on a tile row");
  const float* zbase = zl + lk * C::CHS + ((wave * MPW * 16) / TW) * C::PITCH + lj;
  const float* wlane = wl + lane;

  constexpr int NCL = (CLS == 2) ? 4 : 1;  // accumulator sets (output-parity classes computed by this block)
  static_assert(CLS != 2 || (NT == 1 && KS == 3 && EPI == EPI_STORE && PRO == PRO_RAW), "fused classes: convT forward");
  f32x4 accs[NCL][MPW][NT];
  f32x4(&acc)[MPW][NT] = accs[0];
#pragma unroll
  for (int cl = 0; cl < NCL; ++cl)
#pragma unroll
    for (int m = 0; m < MPW; ++m)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) accs[cl][m][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const float* in_n = p.in + (long long)n * (p.pk_sx > 1 ? p.pk_sx : 1) * p.in_ns;
  const int nchunk = (p.K + 15) >> 4;

  // Register-staged software pipeline: the global loads of chunk ch+1 are issued BEFORE the MFMA phase of
  // chunk ch and committed (activation applied, written to LDS) after it, so HBM/L2 latency hides under MFMA.
  // Loads are unconditional from clamped, always-valid addresses (uniform channel base in SGPRs + one 32-bit
  // per-thread offset); validity is applied as a select at commit time -> no exec-mask branches per load.
  float rin[V4 ? 1 : 16][NPOSX];
  float rwt[C::NWE];
  int gsafe[NPOSX];
#pragma unroll
  for (int i = 0; i < NPOSX; ++i) gsafe[i] = goff[i] >= 0 ? goff[i] : 0;
  // V4 staging map: thread -> NQ aligned quads (channel-in-chunk, row, 4 columns) of the chunk image
  constexpr int NQX = V4 ? C::NQ : 1;
  constexpr int QPR = C::PITCH / 4;                 // quads per image row
  constexpr int QPC = C::ROWS * QPR;                // quads per channel
  float4 rq[NQX];
  int qg[NQX], qmeta[NQX];  // global offset (floats, incl. channel-in-chunk), lds float index | cc << 20 | valid << 28
  if constexpr (V4) {
#pragma unroll
    for (int i = 0; i < NQX; ++i) {
      const int e = tid + 256 * i;
      qg[i] = 0;
      qmeta[i] = -1;
      if (e < 16 * QPC) {
        const int cc = e / QPC, rem = e - cc * QPC;
        const int r = rem / QPR, q = rem - r * QPR;
        const int iy = gy0 - C::HALO / 2 + r, ix = gx0 - C::XPAD + 4 * q;
        const bool ok = iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;  // W % 4 == 0: a quad is all-in or all-out
        qg[i] = cc * p.in_cs + (ok ? iy * p.Win + ix : 0);
        qmeta[i] = (cc * C::CHS + r * C::PITCH + 4 * q) | (cc << 20) | ((ok ? 1 : 0) << 28);
      }
    }
  }
  // weight slab element handled by this thread in pass i: global offset relative to (jbase, c0), LDS slot,
  // input-channel-in-chunk and validity, packed:  wmeta = lds_index | cc << 20 | valid << 28
  auto wdecode = [&](int e, int& go, int& meta) {
    go = 0;
    meta = -1;
    if (e < C::WE) {
      const int s = e % C::NS, t = e / C::NS;
      const int cc = t & 15, jj = t >> 4;
      int tap;
      if (p.tapmode == TM_ID) {
        tap = s;
      } else if (p.tapmode == TM_FLIP) {
        tap = C::NS - 1 - s;
      } else {
        const int sy = s / 3, sx = s - sy * 3;
        const int ky = (sy == 1) ? py : ((sy == 0 && py == 0) ? 2 : -1);
        const int kx = (sx == 1) ? px : ((sx == 0 && px == 0) ? 2 : -1);
        tap = (ky < 0 || kx < 0) ? -1 : ky * 3 + kx;
      }
      const bool valid = (jbase + jj < p.J) && tap >= 0;
      const int jrel = min(jbase + jj, p.J - 1) - jbase;
      go = (int)(jrel * p.w_js + cc * p.w_ks) + max(tap, 0);
      const int lidx = (((cc >> 2) * C::NS + s) * NT + (jj >> 4)) * 64 + (cc & 3) * 16 + (jj & 15);
      meta = lidx | (cc << 20) | ((valid ? 1 : 0) << 28);
    }
  };
  // small slabs: keep the decode in registers across chunks; large slabs (few chunks): recompute per chunk
  constexpr bool WHOIST = C::NWE <= 12;
  constexpr int NWH = WHOIST ? C::NWE : 1;
  int wgo_h[NWH], wmeta_h[NWH];
  if constexpr (WHOIST) {
#pragma unroll
    for (int i = 0; i < C::NWE; ++i) wdecode(tid + 256 * i, wgo_h[i], wmeta_h[i]);
  }
  auto wget = [&](int i, int tid_o, int& go, int& meta) {
    if constexpr (WHOIST) {
      go = wgo_h[i];
      meta = wmeta_h[i];
    } else {
      wdecode(tid_o + 256 * i, go, meta);
    }
  };
  const float* wj = p.w + (long long)jbase * p.w_js;
  auto issue = [&](int ch) {
    const int c0 = ch * 16;
    const int kmax = p.K - 1;
    if constexpr (V4) {
      const float* src = in_n + (long long)c0 * p.in_cs;  // uniform chunk base
      const int krem = kmax - c0;
#pragma unroll
      for (int i = 0; i < NQX; ++i) {
        int off = qg[i];
        if (krem < 15 && qmeta[i] >= 0) {  // partial last chunk: clamp the channel (padding lanes keep offset 0)
          const int cc = (qmeta[i] >> 20) & 15;
          off -= (cc - min(cc, krem)) * p.in_cs;
        }
        rq[i] = *reinterpret_cast<const float4*>(src + off);
      }
    } else {
#pragma unroll
      for (int cc = 0; cc < 16; ++cc) {
        const float* src = in_n + (long long)min(c0 + cc, kmax) * p.in_cs;  // uniform, clamped
#pragma unroll
        for (int i = 0; i < NPOSX; ++i) rin[cc][i] = src[gsafe[i]];
      }
    }
    const float* wb = wj + (long long)c0 * p.w_ks;
    int tid_o = tid;
    if constexpr (!WHOIST) asm volatile("" : "+v"(tid_o));  // opaque: do not hoist the large decode
    if (c0 + 16 <= p.K) {
#pragma unroll
      for (int i = 0; i < C::NWE; ++i) {
        int go, meta;
        wget(i, tid_o, go, meta);
        rwt[i] = wb[go];
      }
    } else {  // partial last chunk: clamp the input-channel index
      const int krem = p.K - 1 - c0;
#pragma unroll
      for (int i = 0; i < C::NWE; ++i) {
        int go, meta;
        wget(i, tid_o, go, meta);
        const int cc = meta >= 0 ? ((meta >> 20) & 15) : 0;
        rwt[i] = wb[go - (cc - min(cc, krem)) * (int)p.w_ks];
      }
    }
  };
  auto commit = [&](int ch) {
    const int c0 = ch * 16;
    const int kmax = p.K - 1;
    if constexpr (V4) {
#pragma unroll
      for (int i = 0; i < NQX; ++i) {
        if (qmeta[i] >= 0) {
          const int cc = (qmeta[i] >> 20) & 15;
          const int c = c0 + cc;
          const bool ok = ((qmeta[i] >> 28) & 1) && c <= kmax;
          float4 v = rq[i];
          if constexpr (PRO == PRO_BNRELU) {
            const float a = abl[min(c, kmax)], b = abl[C::AB_MAX + min(c, kmax)];
            v.x = fmaxf(fmaf(a, v.x, b), 0.f);
            v.y = fmaxf(fmaf(a, v.y, b), 0.f);
            v.z = fmaxf(fmaf(a, v.z, b), 0.f);
            v.w = fmaxf(fmaf(a, v.w, b), 0.f);
          }
          if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
          *reinterpret_cast<float4*>(zl + (qmeta[i] & 0xFFFFF)) = v;
        }
      }
    } else {
#pragma unroll
      for (int cc = 0; cc < 16; ++cc) {
        const int c = c0 + cc;
        const bool cv = c <= kmax;
        float a = 1.f, b = 0.f;
        if constexpr (PRO == PRO_BNRELU) {
          a = p.pa[min(c, kmax)];
          b = p.pb[min(c, kmax)];
        }
#pragma unroll
        for (int i = 0; i < NPOSX; ++i) {
          if (loff[i] >= 0) {
            float v = rin[cc][i];
            if constexpr (PRO == PRO_BNRELU) v = fmaxf(fmaf(a, v, b), 0.f);
            zl[cc * C::CHS + loff[i]] = (cv && goff[i] >= 0) ? v : 0.f;
          }
        }
      }
    }
    const int krem = p.K - c0;  // channels of this chunk that exist
    int tid_o = tid;
    if constexpr (!WHOIST) asm volatile("" : "+v"(tid_o));
#pragma unroll
    for (int i = 0; i < C::NWE; ++i) {
      int go, meta;
      wget(i, tid_o, go, meta);
      if (meta >= 0) {
        const int cc = (meta >> 20) & 15;
        const bool valid = ((meta >> 28) & 1) && cc < krem;
        wl[meta & 0xFFFFF] = valid ? rwt[i] : 0.f;
      }
    }
  };

#ifdef RLN_DIAG
  const bool stamps = (p.dbg & 16) != 0;
#else
  constexpr bool stamps = false;
#endif
  unsigned long long t_commit = 0, t_bar1 = 0, t_issue = 0, t_mfma = 0, t_bar2 = 0, t_last = 0;
  auto stamp = [&](unsigned long long& acc) {
    if (stamps) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): s_memtime returns through the scalar data path
      acc += t - t_last;
      t_last = t;
    }
  };
#ifdef RLN_DIAG
  const int dbg = p.dbg;
#else
  constexpr int dbg = 0;
#endif
  int ch_begin = 0, ch_end = nchunk;
  if (p.ksplit > 1) {
    const int per = (nchunk + p.ksplit - 1) / p.ksplit;
    ch_begin = ky * per;
    ch_end = min(nchunk, ch_begin + per);
  }
  if (!(dbg & 1) && ch_begin < ch_end) issue(ch_begin);
  if constexpr (C::AB_FLOATS > 0) __syncthreads();  // a/b table visible before the first commit
  if (stamps) {
    t_last = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
  }
  for (int ch = ch_begin; ch < ch_end; ++ch) {
    if (ch > ch_begin) __syncthreads();  // every wave is done reading the previous chunk
    stamp(t_bar2);
    if (!(dbg & 2)) commit(ch);
    stamp(t_commit);
    __syncthreads();
    stamp(t_bar1);
    if (ch + 1 < ch_end && !(dbg & 1)) issue(ch + 1);
    stamp(t_issue);
    if (dbg & 4) continue;
    // ---- MFMA (channels past K are zero-filled in LDS, so all four 4-channel groups always run) ----
    if constexpr (CLS == 2) {
      igemm_compute_convt4<C, TW>(zbase, wlane, accs);
    } else if constexpr (CLS == 1) {  // ConvTranspose2d output parity class: only the taps with matching parity exist
      switch (cls) {
        case 0: igemm_compute<C, NT, TW, 0x1Bull>(zbase, wlane, acc); break;
        case 1: igemm_compute<C, NT, TW, 0x12ull>(zbase, wlane, acc); break;
        case 2: igemm_compute<C, NT, TW, 0x18ull>(zbase, wlane, acc); break;
        default: igemm_compute<C, NT, TW, 0x10ull>(zbase, wlane, acc); break;
      }
    } else {
      igemm_compute<C, NT, TW, (1ull << C::NS) - 1ull>(zbase, wlane, acc);
    }
    if (stamps) {
      asm volatile("" ::"v"(acc[0][0][0]));  // keep the MFMAs in front of the stamp
      stamp(t_mfma);
    }
  }
  if (stamps && lane == 0 && p.dbg_out != nullptr) {
    atomicAdd(&p.dbg_out[0], t_commit);
    atomicAdd(&p.dbg_out[1], t_bar1);
    atomicAdd(&p.dbg_out[2], t_issue);
    atomicAdd(&p.dbg_out[3], t_mfma);
    atomicAdd(&p.dbg_out[4], t_bar2);
    atomicAdd(&p.dbg_out[5], (unsigned long long)nchunk);
  }

  const long long blk_lin = (long long)n * gridDim.x + bx0;

  if constexpr (EPI == EPI_STORE || EPI == EPI_DGRAD) {
    constexpr int S_ = (CLS != 0) ? 2 : 1;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int j = jbase + nt * 16 + lj;
      const bool jv = j < p.J;
      float s1 = 0.f, s2 = 0.f;
      float bias_j = 0.f, sc = 1.f;
      float ea = 0.f, eb = 0.f, emean = 0.f, einv = 0.f, egam = 0.f;
      bool accum = false;
      if constexpr (EPI == EPI_STORE) {
        if (jv) {
          if (p.bias) bias_j = p.bias[j];
          if (p.nscale && p.pk_sx <= 1) sc = p.nscale[(long long)n * p.J + j];
          if (p.cscale) sc *= p.cscale[j];
        }
      } else {
        if (jv) {
          ea = p.ea[j];
          eb = p.eb[j];
          emean = p.emean[j];
          einv = p.einvstd[j];
          egam = p.egamma[j];
        }
        accum = (j >= p.acc_lo) && (j < p.acc_hi);
      }
      bool dgrad_done = false;
      if constexpr (EPI == EPI_DGRAD) {
        if (p.out_vec) {
          // 16-byte form: the S / G quads of all pixel groups of this channel tile are loaded TOGETHER (one latency
          // instead of MPW), from clamped always-valid addresses through an opaque pointer (as a plain branch the
          // compiler folds this path into the scalar one), then masked, accumulated and stored.
          const int jc = min(j, p.J - 1);
          const float* sbase = p.S + (long long)n * p.s_ns + (long long)jc * p.out_cs;
          float* gbase = p.out + (long long)n * p.out_ns + (long long)jc * p.out_cs;
          asm volatile("" : "+v"(sbase), "+v"(gbase));
          float4 s4[MPW], g4[MPW];
          int poff[MPW];
          bool okm[MPW];
#pragma unroll
          for (int m = 0; m < MPW; ++m) {
            const int q = (wave * MPW + m) * 16 + lk * 4;
            const int ty = q / TW, tx = q - ty * TW;
            const int gy = gy0 + ty, gx = gx0 + tx;
            okm[m] = jv && gy < p.GH && gx < p.GW;  // W % 4 == 0: a 4-pixel group is all-in or all-out
            poff[m] = okm[m] ? gy * p.GW + gx : 0;
            s4[m] = *reinterpret_cast<const float4*>(sbase + poff[m]);
          }
          if (accum) {
#pragma unroll
            for (int m = 0; m < MPW; ++m) g4[m] = *reinterpret_cast<const float4*>(gbase + poff[m]);
          } else {
#pragma unroll
            for (int m = 0; m < MPW; ++m) g4[m] = make_float4(0.f, 0.f, 0.f, 0.f);
          }
          float ovv[MPW][4];
#pragma unroll
          for (int m = 0; m < MPW; ++m) {
            const float sv[4] = {s4[m].x, s4[m].y, s4[m].z, s4[m].w};
            const float gv[4] = {g4[m].x, g4[m].y, g4[m].z, g4[m].w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float yv = fmaf(ea, sv[r], eb);
              const float gyv = (okm[m] && yv > 0.f) ? acc[m][nt][r] : 0.f;
              const float xh = (sv[r] - emean) * einv;
              s1 += gyv;
              s2 += gyv * xh;
              ovv[m][r] = fmaf(egam, gyv, gv[r]);
            }
          }
#pragma unroll
          for (int m = 0; m < MPW; ++m)
            asm volatile("" ::"v"(ovv[m][0]), "v"(ovv[m][1]), "v"(ovv[m][2]), "v"(ovv[m][3]) : "memory");
#pragma unroll
          for (int m = 0; m < MPW; ++m)
            if (okm[m]) *reinterpret_cast<float4*>(gbase + poff[m]) = make_float4(ovv[m][0], ovv[m][1], ovv[m][2], ovv[m][3]);
          dgrad_done = true;
        }
      }
#pragma unroll
      for (int cl = 0; cl < NCL; ++cl) {  // output-parity classes held by this block (1 unless CLS == 2)
        if (dgrad_done) break;
        const int pyc = (CLS == 2) ? (cl >> 1) : py, pxc = (CLS == 2) ? (cl & 1) : px;
        f32x4(&accc)[MPW][NT] = accs[cl];
  #pragma unroll
        for (int m = 0; m < MPW; ++m) {
          const int q = (wave * MPW + m) * 16 + lk * 4;
          const int ty = q / TW, tx = q - ty * TW;
          const int gy = gy0 + ty, gx = gx0 + tx;
          if (!jv || gy >= p.GH) continue;
          if constexpr (EPI == EPI_STORE && CLS == 0) {
            if (p.pk_sx > 1) {  // packed samples: every pixel of the group names its own sample
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int vx = gx + r;
                const int sidx = vx / (p.pk_w + 1), x = vx - sidx * (p.pk_w + 1);
                const int ns = n * p.pk_sx + sidx;
                if (vx < p.GW && x < p.pk_w && ns < p.pk_n) {
                  float t = accc[m][nt][r] + bias_j;
                  if (p.act == 1) t = fmaxf(t, 0.f);
                  float scr = p.nscale ? p.nscale[(long long)ns * p.J + j] : 1.f;
                  if (p.cscale) scr *= p.cscale[j];
                  t *= scr;
                  p.out[(long long)ky * p.split_stride + (long long)ns * p.out_ns + (long long)j * p.out_cs + gy * p.pk_w + x] = t;
                  s1 += t;
                  s2 += t * t;
                }
              }
              continue;
            }
          }
          if constexpr (EPI == EPI_STORE) {
            const int oy = gy * S_ + pyc;
            if (oy >= p.Hout) continue;
            float* dst = p.out + (long long)ky * p.split_stride + (long long)n * p.out_ns + (long long)j * p.out_cs +
                         (long long)oy * p.Wout;
            const int ox0 = gx * S_ + pxc;
            float v[4];
  #pragma unroll
            for (int r = 0; r < 4; ++r) {
              float t = accc[m][nt][r] + bias_j;
              if (p.act != 0) {  // uniform; EncDecNet's Conv = conv -> activation (models/EncDecNet.py:29-33)
                if (p.act == 1) t = fmaxf(t, 0.f);
                else if (p.act == 2) t = t > 0.f ? t : p.act_param * t;
                else if (p.act == 3) t = 1.f / (1.f + expf(-t));
                else t = tanhf(t);
              }
              v[r] = t * sc;
            }
            if (S_ == 1 && p.out_vec && ox0 + 3 < p.Wout) {
              *reinterpret_cast<float4*>(dst + ox0) = make_float4(v[0], v[1], v[2], v[3]);
  #pragma unroll
              for (int r = 0; r < 4; ++r) {
                s1 += v[r];
                s2 += v[r] * v[r];
              }
            } else {
  #pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int ox = ox0 + r * S_;
                if (gx + r < p.GW && ox < p.Wout) {
                  dst[ox] = v[r];
                  s1 += v[r];
                  s2 += v[r] * v[r];
                }
              }
            }
          } else {
            const long long rowoff = (long long)j * p.out_cs + (long long)gy * p.GW + gx;
            const float* sp = p.S + (long long)n * p.s_ns + rowoff;
            float* gp = p.out + (long long)n * p.out_ns + rowoff;
            float sv[4], gv[4];
            const bool vec = p.out_vec && (gx + 3 < p.GW);
            if (vec) {
              const float4 t4 = *reinterpret_cast<const float4*>(sp);
              sv[0] = t4.x; sv[1] = t4.y; sv[2] = t4.z; sv[3] = t4.w;
              if (accum) {
                const float4 g4 = *reinterpret_cast<const float4*>(gp);
                gv[0] = g4.x; gv[1] = g4.y; gv[2] = g4.z; gv[3] = g4.w;
              } else {
                gv[0] = gv[1] = gv[2] = gv[3] = 0.f;
              }
            } else {
  #pragma unroll
              for (int r = 0; r < 4; ++r) {
                const bool ok = gx + r < p.GW;
                sv[r] = ok ? sp[r] : 0.f;
                gv[r] = (ok && accum) ? gp[r] : 0.f;
              }
            }
            float ov[4];
  #pragma unroll
            for (int r = 0; r < 4; ++r) {
              const bool ok = gx + r < p.GW;
              const float yv = fmaf(ea, sv[r], eb);
              const float gyv = (ok && yv > 0.f) ? accc[m][nt][r] : 0.f;
              const float xh = (sv[r] - emean) * einv;
              s1 += gyv;
              s2 += gyv * xh;
              ov[r] = fmaf(egam, gyv, gv[r]);
            }
            if (vec) {
              *reinterpret_cast<float4*>(gp) = make_float4(ov[0], ov[1], ov[2], ov[3]);
            } else {
  #pragma unroll
              for (int r = 0; r < 4; ++r)
                if (gx + r < p.GW) gp[r] = ov[r];
            }
          }
        }
      }
      s1 = group4_sum(s1);
      s2 = group4_sum(s2);
      if (lk == 0) {
        red[((wave * NT + nt) * 16 + lj) * 2 + 0] = s1;
        red[((wave * NT + nt) * 16 + lj) * 2 + 1] = s2;
      }
    }
    __syncthreads();
    if (p.stat_partial != nullptr && tid < NT * 16) {
      const int j = jbase + tid;
      if (j < p.J) {
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          a1 += red[((w * NT) * 16 + tid) * 2 + 0];
          a2 += red[((w * NT) * 16 + tid) * 2 + 1];
        }
        p.stat_partial[(blk_lin * p.J + j) * 2 + 0] = a1;
        p.stat_partial[(blk_lin * p.J + j) * 2 + 1] = a2;
      }
    }
  } else {  // EPI_POOL: conv output tile -> LDS -> 2x2 max (first max wins) -> pooled store + argmax index
    static_assert(EPI != EPI_POOL || (TH * TW == 256), "pool epilogue assumes 64 pooled pixels per channel");
    constexpr int PH = TH / 2, PW = TW / 2;
    float* ol = smem;
    const int oy0 = gy0 >> 1, ox0 = gx0 >> 1;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      __syncthreads();
      {
        const int j = jbase + nt * 16 + lj;
        float bias_j = 0.f, sc = 1.f;
        if (j < p.J) {
          if (p.bias) bias_j = p.bias[j];
          if (p.nscale) sc = p.nscale[(long long)n * p.J + j];
        }
#pragma unroll
        for (int m = 0; m < MPW; ++m) {
          const int q = (wave * MPW + m) * 16 + lk * 4;
#pragma unroll
          for (int r = 0; r < 4; ++r) ol[lj * C::OLS + q + r] = (acc[m][nt][r] + bias_j) * sc;
        }
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int e = tid + 256 * i;
        const int chl = e / (PH * PW);  // == wave + 4*i : wave-uniform
        const int rem = e - chl * (PH * PW);
        const int pyy = rem / PW, pxx = rem - pyy * PW;
        const int jo = jbase + nt * 16 + chl;
        const int oy = oy0 + pyy, ox = ox0 + pxx;
        const bool valid = (jo < p.J) && (oy < p.Hout) && (ox < p.Wout);
        const int base = chl * C::OLS + (2 * pyy) * TW + 2 * pxx;
        const float v00 = ol[base], v01 = ol[base + 1], v10 = ol[base + TW], v11 = ol[base + TW + 1];
        float best = v00;
        int bi = 0;
        if (v01 > best) { best = v01; bi = 1; }
        if (v10 > best) { best = v10; bi = 2; }
        if (v11 > best) { best = v11; bi = 3; }
        if (valid) {
          p.out[(long long)n * p.out_ns + (long long)jo * p.out_cs + (long long)oy * p.Wout + ox] = best;
          p.pool_idx[(((long long)n * p.J + jo) * p.Hout + oy) * p.Wout + ox] = (unsigned char)bi;
        }
        float a1 = valid ? best : 0.f;
        float a2 = valid ? best * best : 0.f;
        a1 = wave_sum64(a1);
        a2 = wave_sum64(a2);
        if (lane == 0 && p.stat_partial != nullptr && jo < p.J) {
          p.stat_partial[(blk_lin * p.J + jo) * 2 + 0] = a1;
          p.stat_partial[(blk_lin * p.J + jo) * 2 + 1] = a2;
        }
      }
    }
  }
}

template <int KS, int NT, int PRO, int EPI, int TH, int TW, int CLS, bool V4>
static int launch_v(const IgemmParams& p, int N, hipStream_t stream) {
  using C = IgCfg<KS, NT, PRO, EPI, TH, TW, V4>;
  if (C::AB_FLOATS > 0 && p.K > C::AB_MAX) return -4;
  static DevOnce attr_once;
  auto kern = igemm_k<KS, NT, PRO, EPI, TH, TW, CLS, V4>;
  if ((p.ncls > 1) != (CLS == 1)) return -1;
  if (attr_once.first()) {
    const hipError_t attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                              C::LDS_BYTES);
    if (attr_err != hipSuccess) {  // refused: report it here instead of an opaque launch failure later
      (void)hipGetLastError();
      attr_once.undo();
      return (int)attr_err;
    }
    if (rln_env("RLN_DEBUG_OCC")) {
      int nb = -1;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(kern), 256, C::LDS_BYTES);
      fprintf(stderr, "[rln] igemm_k<KS%d NT%d PRO%d EPI%d %dx%d CLS%d V4%d> lds %d B -> %d blocks/CU\n", KS, NT, PRO, EPI,
              TH, TW, (int)CLS, (int)V4, C::LDS_BYTES, nb);
    }
  }
  dim3 grid((unsigned)(p.ncls * p.tiles_x * p.tiles_y),
            (unsigned)(((p.J + NT * 16 - 1) / (NT * 16)) * (p.ksplit > 1 ? p.ksplit : 1)), (unsigned)N);
  static const int dbg = rln_env("RLN_DBG") ? atoi(rln_env("RLN_DBG")) : 0;
  if (dbg) {
    IgemmParams q = p;
    q.dbg = dbg;
    // stamps only for the kernel class selected by RLN_DBG_KS / RLN_DBG_NT (default: dense forward KS3 NT1 BNRELU)
    q.dbg_out = ((dbg & 16) && KS == 3 && NT == 1 && PRO == PRO_BNRELU) ? igemm_debug_buffer() : nullptr;
    if (!q.dbg_out) q.dbg &= ~16;
    hipLaunchKernelGGL(kern, grid, dim3(256), C::LDS_BYTES, stream, q);
    return (int)hipGetLastError();
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), C::LDS_BYTES, stream, p);
  return (int)hipGetLastError();
}

// picks the 16-byte staging variant when the K-side operand allows it
template <int KS, int NT, int PRO, int EPI, int TH, int TW, int CLS = 0>
static int launch_t(const IgemmParams& p, int N, hipStream_t stream) {
  if constexpr (PRO != PRO_S2D) {
    // measured: on the small tiles the 16-byte staging path is not faster than the scalar one (the tile-with-halo
    // pattern is bound by 128-byte line requests, not by instruction count); it is used by the strip tiles only.
    static const bool want_v4 = rln_env("RLN_V4_ALL") != nullptr;
    const bool v4 = want_v4 && (p.Win % 4 == 0) && (p.in_cs % 4 == 0) && (p.in_ns % 4 == 0) &&
                    ((((uintptr_t)p.in) & 15) == 0);
    if (v4) return launch_v<KS, NT, PRO, EPI, TH, TW, CLS, true>(p, N, stream);
  }
  return launch_v<KS, NT, PRO, EPI, TH, TW, CLS, false>(p, N, stream);
}

unsigned long long* igemm_debug_buffer() {
  static unsigned long long* buf = nullptr;
  if (!buf) {
    (void)hipMalloc(&buf, 8 * sizeof(unsigned long long));
    (void)hipMemset(buf, 0, 8 * sizeof(unsigned long long));
  }
  return buf;
}

void igemm_tile_dims(IgemmKind kind, int tile, int* th, int* tw) {
  if (kind == IG_S2D3) {
    *th = 8;
    *tw = 16;
  } else if (tile == 2) {  // full-width strips: no column halo, rows start on 128-byte lines
    *th = 4;
    *tw = 160;
  } else if (tile == 3) {
    *th = 8;
    *tw = 80;
  } else if (tile == 4) {  // half-width strips: 3 blocks per CU
    *th = 4;
    *tw = 80;
  } else if (tile == 5) {  // rows of <= 4 pixels (packed samples of the deepest level)
    *th = 4;
    *tw = 32;
  } else if (tile == 0) {
    *th = 8;
    *tw = 32;
  } else {
    *th = 16;
    *tw = 16;
  }
}

// strip tiles need the 16-byte staging path: W % 4 == 0 and aligned planes (checked by the caller)
int igemm_pick_strip_tile(int gw) {
  if (rln_env("RLN_NO_STRIP")) return -1;
  static const int mode = rln_env("RLN_STRIP_MODE") ? atoi(rln_env("RLN_STRIP_MODE")) : 0;
  if (mode == 1) return (gw == 160 || gw == 80) ? 4 : -1;  // experiment: 4x80 everywhere
  if (mode == 2) return gw == 160 ? 4 : (gw == 80 ? 3 : -1);
  if (mode == 3) return gw == 160 ? 2 : (gw == 80 ? 3 : -1);
  return gw == 160 ? 4 : (gw == 80 ? 3 : -1);  // default: 4x80 half strips on 160-wide levels (3 blocks/CU), 8x80 on 80-wide
}

int igemm_pick_tile(int gh, int gw) {
  // covered area of each tiling; prefer the one wasting fewer pixels (ties -> 8x32: longer rows)
  const long long a0 = (long long)((gh + 7) / 8) * ((gw + 31) / 32);
  const long long a1 = (long long)((gh + 15) / 16) * ((gw + 15) / 16);
  return (a1 < a0) ? 1 : 0;
}

int igemm_launch(IgemmKind kind, int tile, const IgemmParams& p, int N, hipStream_t stream) {
  switch (kind) {
    case IG_CONV3_BN:
      if (tile == 2) return launch_v<3, 1, PRO_BNRELU, EPI_STORE, 4, 160, 0, true>(p, N, stream);
      if (tile == 3) return launch_v<3, 1, PRO_BNRELU, EPI_STORE, 8, 80, 0, true>(p, N, stream);
      if (tile == 4) return launch_v<3, 1, PRO_BNRELU, EPI_STORE, 4, 80, 0, true>(p, N, stream);
      if (tile == 5) return launch_v<3, 1, PRO_BNRELU, EPI_STORE, 4, 32, 0, false>(p, N, stream);
      if (p.pk_sx > 1) return launch_v<3, 1, PRO_BNRELU, EPI_STORE, 8, 32, 0, false>(p, N, stream);  // scalar staging
      return tile == 0 ? launch_t<3, 1, PRO_BNRELU, EPI_STORE, 8, 32>(p, N, stream)
                       : launch_t<3, 1, PRO_BNRELU, EPI_STORE, 16, 16>(p, N, stream);
    case IG_CONVT4:
      return tile == 0 ? launch_t<3, 1, PRO_RAW, EPI_STORE, 8, 32, 2>(p, N, stream)
                       : launch_t<3, 1, PRO_RAW, EPI_STORE, 16, 16, 2>(p, N, stream);
    case IG_CONV3_RAW:
      if (p.ncls > 1)
        return tile == 0 ? launch_t<3, 1, PRO_RAW, EPI_STORE, 8, 32, 1>(p, N, stream)
                         : launch_t<3, 1, PRO_RAW, EPI_STORE, 16, 16, 1>(p, N, stream);
      return tile == 0 ? launch_t<3, 1, PRO_RAW, EPI_STORE, 8, 32>(p, N, stream)
                       : launch_t<3, 1, PRO_RAW, EPI_STORE, 16, 16>(p, N, stream);
    case IG_CONV1_POOL:
      return tile == 0 ? launch_t<1, 4, PRO_BNRELU, EPI_POOL, 8, 32>(p, N, stream)
                       : launch_t<1, 4, PRO_BNRELU, EPI_POOL, 16, 16>(p, N, stream);
    case IG_CONV1_BN:
      return tile == 0 ? launch_t<1, 4, PRO_BNRELU, EPI_STORE, 8, 32>(p, N, stream)
                       : launch_t<1, 4, PRO_BNRELU, EPI_STORE, 16, 16>(p, N, stream);
    case IG_DGRAD3:
      return tile == 0 ? launch_t<3, 4, PRO_RAW, EPI_DGRAD, 8, 32>(p, N, stream)
                       : launch_t<3, 4, PRO_RAW, EPI_DGRAD, 16, 16>(p, N, stream);
    case IG_DGRAD1:
      return tile == 0 ? launch_t<1, 4, PRO_RAW, EPI_DGRAD, 8, 32>(p, N, stream)
                       : launch_t<1, 4, PRO_RAW, EPI_DGRAD, 16, 16>(p, N, stream);
    case IG_S2D3: {
      // output-channel tiles per block.  80 channels: 2 x 48 wastes 17 % of the MFMA columns (2 x 64: 38 %) and keeps
      // two blocks per CU; 1 x 80 fits only one block per CU and measured slower (1.72 vs 1.40 ms/step).
      static const int force = rln_env("RLN_S2D_NT") ? atoi(rln_env("RLN_S2D_NT")) : 0;
      const int waste4 = (p.J + 63) / 64 * 64 - p.J, waste3 = (p.J + 47) / 48 * 48 - p.J;
      const int nt = force ? force : (waste3 < waste4 ? 3 : 4);
      if (nt == 5) return launch_t<3, 5, PRO_S2D, EPI_STORE, 8, 16>(p, N, stream);
      if (nt == 3) return launch_t<3, 3, PRO_S2D, EPI_STORE, 8, 16>(p, N, stream);
      return launch_t<3, 4, PRO_S2D, EPI_STORE, 8, 16>(p, N, stream);
    }
    case IG_CONV7_RAW:
      return tile == 0 ? launch_v<7, 1, PRO_RAW, EPI_STORE, 8, 32, false, false>(p, N, stream)
                       : launch_v<7, 1, PRO_RAW, EPI_STORE, 16, 16, false, false>(p, N, stream);
    case IG_CONV1_RAW:
      return tile == 0 ? launch_v<1, 1, PRO_RAW, EPI_STORE, 8, 32, false, false>(p, N, stream)
                       : launch_v<1, 1, PRO_RAW, EPI_STORE, 16, 16, false, false>(p, N, stream);
  }
  return -1;
}

// =============================================================================================
// dense-layer data gradient, looped form (K = growth <= 16 input channels = dY, J = Cin outputs)
//
// The 16-channel dY tile (+halo) is staged once; the block then walks the Cin output channels 16 at a time:
// weight slab ct+1 is loaded while slab ct is in the MFMA phase (two LDS slab buffers, one barrier per step),
// the epilogue operands (S for the ReLU mask / xhat, G for accumulation) are prefetched before the MFMA phase,
// and the BatchNorm-backward partial sums leave the block per wave (4 partial rows per block).
// =============================================================================================

// VEC: rows are 16-byte aligned and W % 4 == 0 (host-checked), so every 4-pixel group is all-in or all-out and the
// S / G traffic moves as one 16-byte access per lane.  A compile-time switch: as a run-time branch the compiler
// folds both forms into the scalar one (32 dword loads per lane and step instead of 8 dwordx4).
template <int TH, int TW, bool VEC, int ST>  // ST: storage element type of dY (p.in) and S; G (p.out) is fp32
__global__ __launch_bounds__(256, VEC ? 3 : 2) void dgrad_loop_k(const IgemmParams p) {  // scalar form: small levels, few blocks
  using C = IgCfg<3, 1, PRO_RAW, EPI_DGRAD, TH, TW>;
  constexpr int MPW = C::MPW;
  constexpr int NWE = C::NWE;  // 9 weight elements per thread and slab
  extern __shared__ __align__(16) float smem[];
  float* zl = smem;
  float* wl0 = smem + C::IN_FLOATS;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, lj = lane & 15, lk = lane >> 4;
  const int bx = blockIdx.x;
  const int tile_y = bx / p.tiles_x, tile_x = bx - tile_y * p.tiles_x;
  const int gy0 = tile_y * TH, gx0 = tile_x * TW;
  const int n = blockIdx.z;

  // ---- stage the dY tile once (clamped unconditional loads + select) ----
  const int pk = (!VEC && p.pk_sx > 1) ? p.pk_sx : 1;  // samples per virtual image (igemm.h: sample packing)
  {
    const SP<ST> in_n = SP<ST>(p.in) + (long long)n * pk * p.in_ns;
    const int kmax = p.K - 1;
#pragma unroll
    for (int i = 0; i < C::NPOS; ++i) {
      const int e = tid + 256 * i;
      if (e < C::POS) {
        const int r = e / C::PITCH, col = e - r * C::PITCH;
        const int iy = gy0 - 1 + r, ix = gx0 - 1 + col;
        bool ok = iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
        int off = ok ? iy * p.Win + ix : 0;
        if (!VEC && pk > 1 && ok) {
          const int sidx = ix / (p.pk_w + 1), x = ix - sidx * (p.pk_w + 1);
          ok = x < p.pk_w && n * pk + sidx < p.pk_n;
          off = ok ? sidx * (int)p.in_ns + iy * p.pk_w + x : 0;
        }
        float v[16];
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) v[cc] = in_n.ld1((long long)min(cc, kmax) * p.in_cs + off);
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) zl[cc * C::CHS + e] = (ok && cc <= kmax) ? v[cc] : 0.f;
      }
    }
  }
  // ---- weight slab map: element (s, o = cc, c = jj) of a 16-channel output tile ----
  int wgo[NWE], wmeta[NWE];
#pragma unroll
  for (int i = 0; i < NWE; ++i) {
    const int e = tid + 256 * i;  // < 2304 always (9 * 256)
    const int s = e % 9, t = e / 9;
    const int cc = t & 15, jj = t >> 4;
    wgo[i] = jj * (int)p.w_js + min(cc, p.K - 1) * (int)p.w_ks + (8 - s);
    wmeta[i] = (((cc >> 2) * 9 + s) * 64 + (cc & 3) * 16 + jj) | (jj << 20) | ((cc < p.K ? 1 : 0) << 28);
  }
  float rwt[NWE];
  const int nct = (p.J + 15) >> 4;
  auto issue_w = [&](int ct) {
    const float* wb = p.w + (long long)(ct * 16) * p.w_js;
    if (ct * 16 + 16 <= p.J) {
#pragma unroll
      for (int i = 0; i < NWE; ++i) rwt[i] = wb[wgo[i]];
    } else {
      const int jrem = p.J - 1 - ct * 16;
#pragma unroll
      for (int i = 0; i < NWE; ++i) {
        const int jj = (wmeta[i] >> 20) & 15;
        rwt[i] = wb[wgo[i] - (jj - min(jj, jrem)) * (int)p.w_js];
      }
    }
  };
  auto commit_w = [&](int ct, float* wl) {
    const int jrem = p.J - ct * 16;
#pragma unroll
    for (int i = 0; i < NWE; ++i) {
      const int jj = (wmeta[i] >> 20) & 15;
      const bool valid = ((wmeta[i] >> 28) & 1) && jj < jrem;
      wl[wmeta[i] & 0xFFFFF] = valid ? rwt[i] : 0.f;
    }
  };

  // ---- per-lane epilogue geometry: M-tile m covers 4 consecutive pixels of one row ----
  int pixoff[MPW];
  int po[VEC ? 1 : MPW][VEC ? 1 : 4];  // scalar path: element offset of every pixel (clamped to a valid one)
  unsigned vmask = 0;  // 4 validity bits per M-tile
#pragma unroll
  for (int m = 0; m < MPW; ++m) {
    const int q = (wave * MPW + m) * 16 + lk * 4;
    const int ty = q / TW, tx = q - ty * TW;
    const int gy = gy0 + ty, gx = gx0 + tx;
    unsigned bits = 0;
    if constexpr (!VEC) {
      if (pk > 1) {
        pixoff[m] = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int vx = gx + r;
          const int sidx = vx / (p.pk_w + 1), x = vx - sidx * (p.pk_w + 1);
          const bool ok = gy < p.GH && vx < p.GW && x < p.pk_w && n * pk + sidx < p.pk_n;
          bits |= (ok ? 1u : 0u) << r;
          po[m][r] = ok ? sidx * (int)p.out_ns + gy * p.pk_w + x : 0;
        }
        vmask |= bits << (4 * m);
        continue;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) bits |= ((gy < p.GH && gx + r < p.GW) ? 1u : 0u) << r;
    vmask |= bits << (4 * m);
    pixoff[m] = (bits & 1u) ? gy * p.GW + gx : 0;  // clamped to a valid pixel when the whole group is outside
    if constexpr (!VEC) {
#pragma unroll
      for (int r = 0; r < 4; ++r) po[m][r] = pixoff[m] + (((bits >> r) & 1u) ? r : 0);
    }
  }
  constexpr bool vec = VEC;
  const SP<ST> Sn = SP<ST>(p.S) + (long long)n * pk * p.s_ns;
  float* Gn = p.out + (long long)n * pk * p.out_ns;
  const float* zbase = zl + lk * C::CHS + ((wave * MPW * 16) / TW) * C::PITCH + lj;
  const int nct16 = ((p.J + 15) >> 4) * 16;
  float* red = wl0 + 2 * C::W_FLOATS;  // [4 waves][nct16][2]

  // small grids: the output-channel tiles are split over blockIdx.y (disjoint outputs, no reduction)
  const int ct_per = (nct + (int)gridDim.y - 1) / (int)gridDim.y;
  const int ct_begin = (int)blockIdx.y * ct_per;
  const int ct_end = min(nct, ct_begin + ct_per);
  if (ct_begin < ct_end) {
    issue_w(ct_begin);
    commit_w(ct_begin, wl0 + (ct_begin & 1) * C::W_FLOATS);
  }
  __syncthreads();
#ifdef RLN_DIAG
  const bool stamps = (p.dbg & 16) != 0;
#else
  constexpr bool stamps = false;
#endif
  unsigned long long t_pre = 0, t_mfma = 0, t_wait = 0, t_epi = 0, t_tail = 0, t_last = 0;
  auto stamp = [&](unsigned long long& acc) {
    if (stamps) {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_s_waitcnt(0xC07F);
      acc += t - t_last;
      t_last = t;
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if (stamps) {
    t_last = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
  }
  for (int ct = ct_begin; ct < ct_end; ++ct) {
    float* wcur = wl0 + (ct & 1) * C::W_FLOATS;
    float* wnext = wl0 + ((ct + 1) & 1) * C::W_FLOATS;
    if (ct + 1 < ct_end) issue_w(ct + 1);
    const int j = ct * 16 + lj;
    const bool jv = j < p.J;
    const int jc = min(j, p.J - 1);
    const bool accum = (j >= p.acc_lo) && (j < p.acc_hi);
    // prefetch epilogue operands (clamped addresses; invalid lanes/pixels are masked in the epilogue)
    float sv[MPW][4], gv[MPW][4];
    const SP<ST> Sc = Sn + (long long)jc * p.out_cs;
    float* Gc = Gn + (long long)jc * p.out_cs;
#ifdef RLN_DIAG
    if (p.dbg & 1) {  // timing ablation: no S/G reads
#pragma unroll
      for (int m = 0; m < MPW; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) sv[m][r] = gv[m][r] = 1.f;
    } else
#endif
    if constexpr (vec) {
#pragma unroll
      for (int m = 0; m < MPW; ++m) {
        const float4 t4 = Sc.ld4(pixoff[m]);
        sv[m][0] = t4.x; sv[m][1] = t4.y; sv[m][2] = t4.z; sv[m][3] = t4.w;
        const float4 g4 = *reinterpret_cast<const float4*>(Gc + pixoff[m]);
        gv[m][0] = g4.x; gv[m][1] = g4.y; gv[m][2] = g4.z; gv[m][3] = g4.w;
      }
    } else {
#pragma unroll
      for (int m = 0; m < MPW; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sv[m][r] = Sc.ld1(po[m][r]);
          gv[m][r] = Gc[po[m][r]];
        }
    }
    const float ea = p.ea[jc], eb = p.eb[jc], emean = p.emean[jc], einv = p.einvstd[jc], egam = p.egamma[jc];

    f32x4 acc[MPW][1];
#pragma unroll
    for (int m = 0; m < MPW; ++m) acc[m][0] = f32x4{0.f, 0.f, 0.f, 0.f};
    stamp(t_pre);
#ifdef RLN_DIAG
    if (!(p.dbg & 4))
#endif
    igemm_compute<C, 1, TW, 0x1FFull>(zbase, wcur + lane, acc);
    stamp(t_mfma);
    if (stamps) {  // diagnostic only: separate the wait for the prefetched operands from the epilogue proper
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
      stamp(t_wait);
    }

    float s1 = 0.f, s2 = 0.f;
    float ov[MPW][4];
#pragma unroll
    for (int m = 0; m < MPW; ++m) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool ok = jv && ((vmask >> (4 * m + r)) & 1u);
        const float yv = fmaf(ea, sv[m][r], eb);
        const float gyv = (ok && yv > 0.f) ? acc[m][0][r] : 0.f;
        const float xh = (sv[m][r] - emean) * einv;
        s1 += gyv;
        s2 += gyv * xh;
        ov[m][r] = fmaf(egam, gyv, accum ? gv[m][r] : 0.f);
      }
    }
    // Two things the wait-count logic needs from the source (vmcnt counts loads and stores together, in order):
    //  * every prefetched register is consumed on EVERY path -- with the arithmetic sunk into a predicated store
    //    block, the skip path leaves loads "pending" and a vmcnt(0) lands at the top of the next step;
    //  * all loads are consumed BEFORE the first store is issued -- a load consumed after a store waits for that
    //    store's completion.
#pragma unroll
    for (int m = 0; m < MPW; ++m) asm volatile("" ::"v"(ov[m][0]), "v"(ov[m][1]), "v"(ov[m][2]), "v"(ov[m][3]) : "memory");
#pragma unroll
    for (int m = 0; m < MPW; ++m) {
#ifdef RLN_DIAG
      if (p.dbg & 2) {  // timing ablation: no G writes (keep the values live)
        if (ov[m][0] + ov[m][1] + ov[m][2] + ov[m][3] == 1.2345e-30f) Gc[0] = ov[m][0];
      } else
#endif
      if constexpr (vec) {
        if (jv && ((vmask >> (4 * m)) & 1u))
          *reinterpret_cast<float4*>(Gc + pixoff[m]) = make_float4(ov[m][0], ov[m][1], ov[m][2], ov[m][3]);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (jv && ((vmask >> (4 * m + r)) & 1u)) Gc[po[m][r]] = ov[m][r];
      }
    }
    stamp(t_epi);
    s1 = group4_sum(s1);
    s2 = group4_sum(s2);
    if (lk == 0) {  // per-wave slot in LDS; summed over the 4 waves after the loop
      red[(wave * nct16 + j) * 2 + 0] = s1;
      red[(wave * nct16 + j) * 2 + 1] = s2;
    }
    if (ct + 1 < ct_end) commit_w(ct + 1, wnext);
    __syncthreads();
    stamp(t_tail);
  }
  if (stamps && lane == 0 && p.dbg_out) {
    atomicAdd(&p.dbg_out[0], t_pre);
    atomicAdd(&p.dbg_out[1], t_mfma);
    atomicAdd(&p.dbg_out[2], t_wait);
    atomicAdd(&p.dbg_out[3], t_epi);
    atomicAdd(&p.dbg_out[4], t_tail);
    atomicAdd(&p.dbg_out[5], (unsigned long long)(ct_end - ct_begin));
  }
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

template <int TH, int TW, bool VEC, int ST = ST_F32>
static int dgrad_loop_launch_t(const IgemmParams& p, int N, hipStream_t stream) {
  using C = IgCfg<3, 1, PRO_RAW, EPI_DGRAD, TH, TW>;
  const int LDS = (C::IN_FLOATS + 2 * C::W_FLOATS + 4 * (((p.J + 15) >> 4) * 16) * 2) * 4;
  static DevOnce attr_once;
  auto kern = dgrad_loop_k<TH, TW, VEC, ST>;
  if (attr_once.first()) {
    const hipError_t attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                              96 * 1024);
    if (attr_err != hipSuccess) {  // refused: report it here instead of an opaque launch failure later
      (void)hipGetLastError();
      attr_once.undo();
      return (int)attr_err;
    }
    if (rln_env("RLN_DEBUG_OCC")) {
      int nb = -1;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(kern), 256, LDS);
      fprintf(stderr, "[rln] dgrad_loop_k<%dx%d> lds %d B -> %d blocks/CU\n", TH, TW, LDS, nb);
    }
  }
  // aim for >= ~2 blocks per CU on small levels by splitting the output-channel loop
  const long long base_blocks = (long long)p.tiles_x * p.tiles_y * N;
  const int nct = (p.J + 15) / 16;
  int split = 1;
  if (base_blocks < 512) split = (int)std::min<long long>((512 + base_blocks - 1) / base_blocks, (long long)nct);
  dim3 grid((unsigned)(p.tiles_x * p.tiles_y), (unsigned)split, (unsigned)N);
  static const int dbg = rln_env("RLN_DBG") ? atoi(rln_env("RLN_DBG")) : 0;
  if ((dbg & 64) && TH == 8 && p.GW >= 160) {  // diagnostic build: phase stamps / ablations of the level-0 launches
    static const int abl = rln_env("RLN_DG_ABL") ? atoi(rln_env("RLN_DG_ABL")) : 0;
    IgemmParams q = p;
    q.dbg = 16 | abl;
    q.dbg_out = igemm_debug_buffer();
    hipLaunchKernelGGL(kern, grid, dim3(256), LDS, stream, q);
    return (int)hipGetLastError();
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), LDS, stream, p);
  return (int)hipGetLastError();
}

// K (= dY channels) must be <= 16.  One stat-partial row per block.
int dgrad_loop_launch(int tile, const IgemmParams& p, int N, hipStream_t stream) {
  if (p.K > 16 || p.ncls != 1) return -1;
  if (p.st == ST_BF16) {  // bf16 dY / S planes: quads of pixels are 8-byte aligned (host-checked like out_vec)
    if (!p.out_vec) return -4;
    return tile == 0 ? dgrad_loop_launch_t<8, 32, true, ST_BF16>(p, N, stream)
                     : dgrad_loop_launch_t<16, 16, true, ST_BF16>(p, N, stream);
  }
  if (p.pk_sx > 1 && (p.out_vec || p.s_ns != p.out_ns)) return -1;  // packed samples: scalar path, S and G views congruent
  if (p.out_vec)
    return tile == 0 ? dgrad_loop_launch_t<8, 32, true>(p, N, stream) : dgrad_loop_launch_t<16, 16, true>(p, N, stream);
  if (tile == 5) return dgrad_loop_launch_t<4, 32, false>(p, N, stream);
  return tile == 0 ? dgrad_loop_launch_t<8, 32, false>(p, N, stream) : dgrad_loop_launch_t<16, 16, false>(p, N, stream);
}

// =============================================================================================
// weight gradients
// =============================================================================================

template <int KS, int MT, int PRO, bool SHIFT_A, int TH, int TW>
struct WgCfg {
  static constexpr bool S2D = (PRO == PRO_S2D);
  static constexpr int HALO = KS - 1;
  static constexpr int PITCH = S2D ? (TW + 1) : (TW + HALO);
  static constexpr int ROWS = S2D ? (TH + 1) : (TH + HALO);
  static constexpr int PLANE = PITCH * ROWS;
  static constexpr int POS = PLANE * (S2D ? 4 : 1);
  static constexpr int NPOS = cdiv(POS, 256);
  static constexpr int VST = round_mod32(POS, 2);
  static constexpr int UST = round_mod32(TH * TW, 2);
  static constexpr int NS = KS * KS;
  static constexpr int MCH = MT * 16;
  static constexpr int NCH = 64;
  static constexpr int VCH = SHIFT_A ? MCH : NCH;
  static constexpr int UCH = SHIFT_A ? NCH : MCH;
  static constexpr int V_FLOATS = VCH * VST;
  static constexpr int U_FLOATS = UCH * UST;
  static constexpr int AB_FLOATS = (PRO == PRO_BNRELU) ? 2 * VCH : 0;
  static constexpr int LDS_BYTES = (V_FLOATS + U_FLOATS + AB_FLOATS) * 4;
  __host__ __device__ static constexpr int slot_off(int s) {
    return S2D ? ((((s / 3) & 1) * 2 + ((s % 3) & 1)) * PLANE + ((s / 3) >> 1) * PITCH + ((s % 3) >> 1))
               : ((s / KS) * PITCH + (s % KS));
  }
};

// dW[m][n][tap] = sum over (sample, pixel) of Mop[m][p] * Nop[n][p (+tap)].
// Block: MT M-tiles x 4 N-tiles (one per wave); loops over `items_per_chunk` (sample, tile) items,
// accumulating in registers, then writes one partial slab; a reduce kernel sums the slabs in order.
template <int KS, int MT, int PRO, bool SHIFT_A, int TH, int TW>
__global__ __launch_bounds__(256, 2) void wgrad_k(const WgradParams p) {
  using C = WgCfg<KS, MT, PRO, SHIFT_A, TH, TW>;
  extern __shared__ __align__(16) float smem[];
  float* vl = smem;
  float* ul = smem + C::V_FLOATS;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, lj = lane & 15, lk = lane >> 4;
  // Optional XCD-aware order (neighbouring chunks of one channel group on one XCD).  Measured 3 % SLOWER than
  // dispatch order for this kernel (its re-reads already hit the memory-side cache), so it is off by default.
  unsigned lx = blockIdx.x, ly = blockIdx.y, lz = blockIdx.z;
  if (p.xcd_remap) {
    const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    unsigned l = xcd_logical_block(lin, gridDim.x * gridDim.y * gridDim.z);
    lz = l / (gridDim.x * gridDim.y);
    l -= lz * (gridDim.x * gridDim.y);
    ly = l / gridDim.x;
    lx = l - ly * gridDim.x;
  }
  const int chunk = (int)lx;
  const int mbase = (int)ly * C::MCH;
  const int nbase = (int)lz * C::NCH;
  const int ubase = SHIFT_A ? nbase : mbase;
  const int vbase = SHIFT_A ? mbase : nbase;

  // Staging map: a thread owns one tile position (and, for images of <= 128 positions, one of two channels
  // staged per pass); a pass moves 256 values.  Per-thread state is one relative offset per sub-pass.
  constexpr int UPIX = TH * TW;                       // 128
  constexpr int UG = 256 / UPIX;                      // channels per U pass (2)
  constexpr int NUP = C::UCH / UG;                    // U passes
  constexpr int VG = (C::POS <= 128) ? 2 : 1;         // channels per V pass
  constexpr int VSUB = (VG == 2) ? 1 : cdiv(C::POS, 256);  // sub-passes per channel
  constexpr int NVP = (C::VCH / VG) * VSUB;           // V loads per thread
  static_assert(UPIX == 128 && C::UCH % UG == 0 && C::VCH % VG == 0, "staging map assumes 128-pixel tiles");
  float ru[NUP], rv[NVP];
  // partial channel groups: V channels past Vc are neither loaded nor committed (their LDS rows only feed output
  // rows/columns that are never stored), and waves whose 16 N-side channels are all past the end skip their MFMAs
  // (guarding single M-tiles inside the unrolled MFMA stream measured slower than computing the padding)
  const int nvalid_v = min(C::VCH, p.Vc - vbase);                      // >= 1
  const int nvalid_u = min(C::UCH, p.Uc - ubase);                      // >= 1
  const int n_live = SHIFT_A ? nvalid_u : nvalid_v;                    // N side: one 16-channel tile per wave
  const bool wave_live = wave * 16 < n_live;

  f32x4 acc[MT][C::NS];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int s = 0; s < C::NS; ++s) acc[mt][s] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int tiles = p.tiles_x * p.tiles_y;
  const long long total = (long long)p.N * tiles;
  const long long it0 = (long long)chunk * p.items_per_chunk;
  long long it1 = it0 + p.items_per_chunk;
  if (it1 > total) it1 = total;

  // U: thread -> (sub-channel, pixel)
  const int u_sub = tid / UPIX, u_q = tid - u_sub * UPIX;
  const int u_ty = u_q / TW, u_tx = u_q - u_ty * TW;
  // V: thread -> (sub-channel, VSUB positions): source row/col relative to the tile origin
  const int v_sub = (VG == 2) ? tid / 128 : 0;
  int v_dy[VSUB], v_dx[VSUB], v_pos[VSUB];
#pragma unroll
  for (int k = 0; k < VSUB; ++k) {
    const int pos = (VG == 2) ? (tid & 127) : (tid + 256 * k);
    v_pos[k] = (pos < C::POS) ? pos : -1;
    if constexpr (C::S2D) {
      const int pl = pos / C::PLANE, rem = pos - pl * C::PLANE;
      const int r = rem / C::PITCH, col = rem - r * C::PITCH;
      v_dy[k] = 2 * r + (pl >> 1);
      v_dx[k] = 2 * col + (pl & 1);
    } else {
      const int r = pos / C::PITCH, col = pos - r * C::PITCH;
      v_dy[k] = r - C::HALO / 2;
      v_dx[k] = col - C::HALO / 2;
    }
  }

  // BatchNorm scale/shift of this block's V channels: constant over the items, read from LDS in the commit phase
  // (applying them in the issue phase would make every load wait for its data and serialise the prefetch).
  float* abl = ul + C::U_FLOATS;  // [VCH][2]
  if constexpr (PRO == PRO_BNRELU) {
    for (int c = tid; c < C::VCH; c += 256) {  // channels past Vc get (0, 0): relu(0 * v + 0) = 0, no select needed
      const bool cv = vbase + c < p.Vc;
      const int cg = min(vbase + c, p.Vc - 1);
      abl[2 * c + 0] = cv ? p.pa[cg] : 0.f;
      abl[2 * c + 1] = cv ? p.pb[cg] : 0.f;
    }
  }
  // Register-staged pipeline (see igemm_k): loads of item it+1 fly while item it is in the MFMA phase.  The issue
  // phase only issues (raw values + validity bits); the commit phase applies activation / masks and writes LDS.
  unsigned okbits = 0;  // bit 0: U position valid, bit 1+k: V sub-pass k valid
  auto issue = [&](long long it) {
    const int n = (int)(it / tiles);
    const int t = (int)(it - (long long)n * tiles);
    const int tile_y = t / p.tiles_x, tile_x = t - tile_y * p.tiles_x;
    const int gy0 = tile_y * TH, gx0 = tile_x * TW;
    unsigned bits = 0;
    // Unconditional loads from clamped (always valid) addresses: straight-line code, no per-load branch.
    // Address = uniform channel base (SGPR pair, clamped) + ONE 32-bit per-thread offset.
    {
      const int gy = gy0 + u_ty, gx = gx0 + u_tx;
      const bool ok = gy < p.GH && gx < p.GW;
      bits |= ok ? 1u : 0u;
      // one uniform 64-bit base per operand and sample; everything per channel is a 32-bit offset
      const int cumax = max(p.Uc - UG, 0);
      const int ub0 = min(ubase, cumax);
      const int usafe = min(u_sub, p.Uc - 1);
      const int uoff = usafe * p.u_cs + (ok ? gy * p.GW + gx : 0);
      const float* un = p.u + (long long)n * p.u_ns + (long long)ub0 * p.u_cs;
#pragma unroll
      for (int i = 0; i < NUP; ++i) ru[i] = un[uoff + (min(ubase + i * UG, cumax) - ub0) * p.u_cs];
    }
    const int oy = C::S2D ? 2 * gy0 : gy0, ox = C::S2D ? 2 * gx0 : gx0;
    const int cvmax = max(p.Vc - VG, 0);
    const int vb0 = min(vbase, cvmax);
    const float* vn = p.v + (long long)n * p.v_ns + (long long)vb0 * p.v_cs;
    const int vsafe = min(v_sub, p.Vc - 1);
#pragma unroll
    for (int k = 0; k < VSUB; ++k) {
      const int iy = oy + v_dy[k], ix = ox + v_dx[k];
      const bool ok = v_pos[k] >= 0 && iy >= 0 && iy < p.Hv && ix >= 0 && ix < p.Wv;
      bits |= (ok ? 1u : 0u) << (1 + k);
      const int voff = vsafe * p.v_cs + (ok ? iy * p.Wv + ix : 0);
#pragma unroll
      for (int i = 0; i < C::VCH / VG; ++i) {
        if (i * VG >= nvalid_v) break;  // uniform
        rv[i * VSUB + k] = vn[voff + (min(vbase + i * VG, cvmax) - vb0) * p.v_cs];  // uniform 32-bit channel offset
      }
    }
    okbits = bits;
  };
  auto commit = [&]() {
    const bool uok = okbits & 1u;
#pragma unroll
    for (int i = 0; i < NUP; ++i)
      ul[(i * UG + u_sub) * C::UST + u_q] = (uok && ubase + i * UG + u_sub < p.Uc) ? ru[i] : 0.f;
#pragma unroll
    for (int k = 0; k < VSUB; ++k) {
      if (v_pos[k] >= 0) {
        const float vmul = ((okbits >> (1 + k)) & 1u) ? 1.f : 0.f;  // padding positions -> exact 0
        constexpr int NI = C::VCH / VG;
        if constexpr (PRO == PRO_BNRELU) {
          constexpr int GRP = 8;  // table reads in groups: one LDS wait per 8 channels instead of one per channel
          static_assert(NI % GRP == 0, "channel groups");
#pragma unroll
          for (int g = 0; g < NI; g += GRP) {
            float2 ab[GRP];
#pragma unroll
            for (int j = 0; j < GRP; ++j) ab[j] = *reinterpret_cast<const float2*>(abl + 2 * ((g + j) * VG + v_sub));
#pragma unroll
            for (int j = 0; j < GRP; ++j) {
              if ((g + j) * VG >= nvalid_v) break;  // uniform
              const int c = (g + j) * VG + v_sub;
              const float z = fmaxf(fmaf(ab[j].x, rv[(g + j) * VSUB + k], ab[j].y), 0.f);
              vl[c * C::VST + v_pos[k]] = z * vmul;
            }
          }
        } else {
#pragma unroll
          for (int i = 0; i < NI; ++i) {
            if (i * VG >= nvalid_v) break;  // uniform
            const int c = i * VG + v_sub;
            vl[c * C::VST + v_pos[k]] = (vmul != 0.f && vbase + c < p.Vc) ? rv[i * VSUB + k] : 0.f;
          }
        }
      }
    }
  };

#ifdef RLN_DIAG
  const bool stamps = p.dbg_out != nullptr;
#else
  constexpr bool stamps = false;
#endif
  unsigned long long t_bar1 = 0, t_commit = 0, t_bar2 = 0, t_issue = 0, t_mfma = 0, t_last = 0;
  auto stamp = [&](unsigned long long& acc) {
    if (stamps) {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_s_waitcnt(0xC07F);
      acc += t - t_last;
      t_last = t;
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if (it0 < it1) issue(it0);
  if (stamps) {
    t_last = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
  }
  for (long long it = it0; it < it1; ++it) {
    __syncthreads();  // every wave is done reading the previous item
    stamp(t_bar1);
    commit();
    stamp(t_commit);
    __syncthreads();
    stamp(t_bar2);
    if (it + 1 < it1) issue(it + 1);
    stamp(t_issue);
    // ---- MFMA over the tile's pixels, 4 consecutive x per k-step; straight-line, constexpr LDS offsets ----
    if (wave_live) {
      const float* ubase = ul + lj * C::UST + lk;
      const float* vbase = vl + lj * C::VST + lk;
      const float* uw = ubase + (SHIFT_A ? wave * 16 * C::UST : 0);
      const float* vw = vbase + (SHIFT_A ? 0 : wave * 16 * C::VST);
      constexpr int KPR = TW / 4;  // k-steps per tile row
#pragma unroll 1
      for (int ty = 0; ty < TH; ++ty) {
        const float* ur = uw + ty * TW;
        const float* vr = vw + ty * C::PITCH;
#pragma unroll 2
        for (int kx = 0; kx < KPR; ++kx) {
          const int tx0 = kx * 4;
          if constexpr (!SHIFT_A) {
            float a[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[mt] = ur[mt * 16 * C::UST + tx0];
#pragma unroll
            for (int s = 0; s < C::NS; ++s) {
              const float b = vr[C::slot_off(s) + tx0];
#pragma unroll
              for (int mt = 0; mt < MT; ++mt) acc[mt][s] = mfma16(a[mt], b, acc[mt][s]);
            }
          } else {
            const float b = ur[tx0];
#pragma unroll
            for (int s = 0; s < C::NS; ++s) {
#pragma unroll
              for (int mt = 0; mt < MT; ++mt) {
                const float a = vr[mt * 16 * C::VST + C::slot_off(s) + tx0];
                acc[mt][s] = mfma16(a, b, acc[mt][s]);
              }
            }
          }
        }
      }
    }
    stamp(t_mfma);
  }
  if (stamps && lane == 0) {
    atomicAdd(&p.dbg_out[0], t_bar1);
    atomicAdd(&p.dbg_out[1], t_commit);
    atomicAdd(&p.dbg_out[2], t_bar2);
    atomicAdd(&p.dbg_out[3], t_issue);
    atomicAdd(&p.dbg_out[4], t_mfma);
    atomicAdd(&p.dbg_out[5], (unsigned long long)(it1 - it0));
  }

  const int Mc = SHIFT_A ? p.Vc : p.Uc;
  const int Nc = SHIFT_A ? p.Uc : p.Vc;
  float* dst = p.partial + (long long)chunk * p.wsize;
  const int nch = nbase + wave * 16 + lj;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int s = 0; s < C::NS; ++s)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int mch = mbase + mt * 16 + 4 * lk + r;
        if (mch < Mc && nch < Nc) dst[(long long)mch * p.m_stride + (long long)nch * p.n_stride + s] = acc[mt][s][r];
      }
}

// ---------------------------------------------------------------------------------------------
// Dense-layer weight gradient with 16-byte staging (levels whose rows are 16-byte aligned and W % 4 == 0).
// Same contraction and partial-slab contract as wgrad_k<3,1,PRO_BNRELU,false>, but the V image is aligned in x
// (it starts 4 columns left of the tile and is TW+8 wide) so that global loads are dwordx4 (17 per thread and
// item instead of 72 dword loads) and LDS commits are ds_write_b64.
template <int TH, int TW, int NCH_>
struct WgqCfg {
  static constexpr int QW = TW / 4 + 2;       // quads per image row
  static constexpr int ROWS = TH + 2;
  static constexpr int PITCH = 4 * QW;
  static constexpr int POS = ROWS * PITCH;
  static constexpr int QPC = ROWS * QW;       // quads per channel
  static constexpr int VST = round_mod32(POS, 2);
  static constexpr int UST = round_mod32(TH * TW, 2);
  static constexpr int NCH = NCH_, MCH = 16;  // V channels per block (64: one N-tile per wave; 32: waves also split rows)
  static constexpr int NTW = NCH / 16;        // N-tiles
  static constexpr int PSPLIT = 4 / NTW;      // row groups of the tile shared out over the waves
  static constexpr int VQ = NCH * QPC;        // V quads per item
  static constexpr int NQV = cdiv(VQ, 256);
  static constexpr int NQU = MCH * (TH * TW / 4) / 256;
  static constexpr int GRP = (NQV % 5 == 0) ? 5 : 4;
  static constexpr int V_FLOATS = NCH * VST;
  static constexpr int U_FLOATS = MCH * UST;
  static constexpr int RED_FLOATS = (PSPLIT - 1) * NTW * 9 * 4 * 64;  // row-group reduction (reuses the staging area)
  static constexpr int LDS_BYTES = cmax(V_FLOATS + U_FLOATS + 2 * NCH, RED_FLOATS) * 4;
  static constexpr int BLOCKS_PER_CU = (NCH == 64) ? 2 : (NCH == 32 ? 3 : 4);
  static_assert((MCH * TH * TW / 4) % 256 == 0 && NQV % GRP == 0, "whole passes / groups");
  static_assert(TH * TW == 128 && VST % 2 == 0 && UST % 2 == 0 && TH % PSPLIT == 0, "128-pixel tiles, 8-byte LDS rows");
  __host__ __device__ static constexpr int slot_off(int s) { return (s / 3) * PITCH + (s % 3) + 3; }
};

template <int TH, int TW, int NCH_>
__global__ __launch_bounds__(256, (NCH_ == 64 ? 2 : (NCH_ == 32 ? 3 : 4))) void wgrad_dense_q_k(const WgradParams p) {
  using C = WgqCfg<TH, TW, NCH_>;
  extern __shared__ __align__(16) float smem[];
  float* vl = smem;
  float* ul = smem + C::V_FLOATS;
  float* abl = ul + C::U_FLOATS;  // [NCH][2]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, lj = lane & 15, lk = lane >> 4;
  const int ntile = wave % C::NTW, ph = wave / C::NTW;  // this wave's N-tile and row group
  const int chunk = blockIdx.x;
  const int nbase = blockIdx.z * C::NCH;  // V (conv input) channels of this block; U = all 16 dY channels

  for (int c = tid; c < C::NCH; c += 256) {  // channels past Vc get (0, 0): relu(0 * v + 0) = 0
    const bool cv = nbase + c < p.Vc;
    const int cg = min(nbase + c, p.Vc - 1);
    abl[2 * c + 0] = cv ? p.pa[cg] : 0.f;
    abl[2 * c + 1] = cv ? p.pb[cg] : 0.f;
  }

  // staging map: quad e = tid + 256 i  ->  (channel, image row, quad in row)
  int vgo[C::NQV], vmeta[C::NQV];  // global offset relative to (sample, tile origin) ; lds | r << 20 | q << 24
#pragma unroll
  for (int i = 0; i < C::NQV; ++i) {
    const int e = min(tid + 256 * i, C::VQ - 1);  // a partial last pass repeats the last quad (same value, same slot)
    const int ch = e / C::QPC, rem = e - ch * C::QPC;
    const int r = rem / C::QW, q = rem - r * C::QW;
    const int cg = min(nbase + ch, p.Vc - 1) - min(nbase, p.Vc - 1);
    vgo[i] = cg * p.v_cs + (r - 1) * p.Wv + 4 * q - 4;
    vmeta[i] = (ch * C::VST + r * C::PITCH + 4 * q) | (r << 20) | (q << 24);
  }
  int ugo[C::NQU], ulo[C::NQU], uty[C::NQU], utx[C::NQU];
#pragma unroll
  for (int i = 0; i < C::NQU; ++i) {
    const int e = tid + 256 * i;
    const int ch = e / (TH * TW / 4), pix = (e - ch * (TH * TW / 4)) * 4;
    uty[i] = pix / TW;
    utx[i] = pix - uty[i] * TW;
    ugo[i] = min(ch, p.Uc - 1) * p.u_cs + uty[i] * p.GW + utx[i];
    ulo[i] = ch * C::UST + pix;
  }
  float4 rv[C::NQV], ru[C::NQU];
  unsigned okv = 0, oku = 0;
  // Partial last channel group (Vc % NCH != 0): staging passes that hold only channels >= Vc are skipped, and so is
  // the MFMA phase of waves whose 16 channels are all >= Vc (their LDS rows are then never read).
  const int nvalid = min(C::NCH, p.Vc - nbase);                       // >= 1
  const int nqv = min(C::NQV, (nvalid * C::QPC + 255) / 256);         // passes that touch a valid channel
  const bool wave_live = ntile * 16 < nvalid;

  f32x4 acc[C::MCH / 16][9];
#pragma unroll
  for (int s = 0; s < 9; ++s) acc[0][s] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int tiles = p.tiles_x * p.tiles_y;
  const long long total = (long long)p.N * tiles;
  const long long it0 = (long long)chunk * p.items_per_chunk;
  long long it1 = it0 + p.items_per_chunk;
  if (it1 > total) it1 = total;
  const int vb0 = min(nbase, p.Vc - 1);

  // issue: raw loads from clamped (always valid) addresses + validity bits; commit: activation, masks, LDS
  auto issue = [&](long long it) {
    const int n = (int)(it / tiles);
    const int t = (int)(it - (long long)n * tiles);
    const int tile_y = t / p.tiles_x, tile_x = t - tile_y * p.tiles_x;
    const int gy0 = tile_y * TH, gx0 = tile_x * TW;
    const float* vn = p.v + (long long)n * p.v_ns + (long long)vb0 * p.v_cs + gy0 * p.Wv + gx0;
    const float* un = p.u + (long long)n * p.u_ns + gy0 * p.GW + gx0;
    unsigned bv = 0, bu = 0;
#pragma unroll
    for (int i = 0; i < C::NQV; ++i) {
      if (i >= nqv) break;  // uniform
      const int r = (vmeta[i] >> 20) & 15, q = (vmeta[i] >> 24) & 15;
      const int iy = gy0 + r - 1, ix = gx0 + 4 * q - 4;
      const bool ok = iy >= 0 && iy < p.Hv && ix >= 0 && ix < p.Wv;  // W % 4 == 0: a quad is all-in or all-out
      bv |= (ok ? 1u : 0u) << i;
      // invalid quads read the (valid) first quad of the tile instead: offset of (r = 1, q = 1) relative to origin
      const int off = ok ? vgo[i] : (vgo[i] - ((r - 1) * p.Wv + 4 * q - 4));
      rv[i] = *reinterpret_cast<const float4*>(vn + off);
    }
#pragma unroll
    for (int i = 0; i < C::NQU; ++i) {
      const bool ok = gy0 + uty[i] < p.GH && gx0 + utx[i] < p.GW;
      bu |= (ok ? 1u : 0u) << i;
      const int off = ok ? ugo[i] : (ugo[i] - (uty[i] * p.GW + utx[i]));
      ru[i] = *reinterpret_cast<const float4*>(un + off);
    }
    okv = bv;
    oku = bu;
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < C::NQU; ++i) {
      const bool ok = (oku >> i) & 1u;
      float* d = ul + ulo[i];
      *reinterpret_cast<float2*>(d) = make_float2(ok ? ru[i].x : 0.f, ok ? ru[i].y : 0.f);
      *reinterpret_cast<float2*>(d + 2) = make_float2(ok ? ru[i].z : 0.f, ok ? ru[i].w : 0.f);
    }
    constexpr int GRP = C::GRP;
#pragma unroll
    for (int g = 0; g < C::NQV; g += GRP) {
      float2 ab[GRP];
#pragma unroll
      for (int j = 0; j < GRP; ++j) {
        const int ch = min(tid + 256 * (g + j), C::VQ - 1) / C::QPC;
        ab[j] = *reinterpret_cast<const float2*>(abl + 2 * ch);
      }
#pragma unroll
      for (int j = 0; j < GRP; ++j) {
        const int i = g + j;
        if (i >= nqv) break;  // uniform
        const bool ok = (okv >> i) & 1u;  // padding quads -> exact zeros
        const float4 x = rv[i];
        float* d = vl + (vmeta[i] & 0xFFFFF);
        *reinterpret_cast<float2*>(d) = make_float2(ok ? fmaxf(fmaf(ab[j].x, x.x, ab[j].y), 0.f) : 0.f,
                                                    ok ? fmaxf(fmaf(ab[j].x, x.y, ab[j].y), 0.f) : 0.f);
        *reinterpret_cast<float2*>(d + 2) = make_float2(ok ? fmaxf(fmaf(ab[j].x, x.z, ab[j].y), 0.f) : 0.f,
                                                        ok ? fmaxf(fmaf(ab[j].x, x.w, ab[j].y), 0.f) : 0.f);
      }
    }
  };

  if (it0 < it1) issue(it0);
  for (long long it = it0; it < it1; ++it) {
    __syncthreads();  // every wave is done reading the previous item (first pass: scale/shift table visible)
    commit();
    __syncthreads();
    if (it + 1 < it1) issue(it + 1);
    if (wave_live) {
      const float* uw = ul + lj * C::UST + lk;
      const float* vw = vl + (ntile * 16 + lj) * C::VST + lk;
      constexpr int KPR = TW / 4;
      constexpr int RPW = TH / C::PSPLIT;  // tile rows per wave
#pragma unroll 1
      for (int ty = ph * RPW; ty < (ph + 1) * RPW; ++ty) {
        const float* ur = uw + ty * TW;
        const float* vr = vw + ty * C::PITCH;
#pragma unroll 2
        for (int kx = 0; kx < KPR; ++kx) {
          const int tx0 = kx * 4;
          const float a = ur[tx0];
#pragma unroll
          for (int s = 0; s < 9; ++s) acc[0][s] = mfma16(a, vr[C::slot_off(s) + tx0], acc[0][s]);
        }
      }
    }
  }

  if constexpr (C::PSPLIT > 1) {  // add the other row groups' sums to the first (fixed order)
    __syncthreads();              // the staging area is free
    if (ph > 0) {
      float* rs = smem + ((ph - 1) * C::NTW + ntile) * (9 * 4 * 64);
#pragma unroll
      for (int s = 0; s < 9; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r) rs[(s * 4 + r) * 64 + lane] = acc[0][s][r];
    }
    __syncthreads();
    if (ph == 0) {
#pragma unroll
      for (int g = 0; g < C::PSPLIT - 1; ++g) {
        const float* rs = smem + (g * C::NTW + ntile) * (9 * 4 * 64);
#pragma unroll
        for (int s = 0; s < 9; ++s)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[0][s][r] += rs[(s * 4 + r) * 64 + lane];
      }
    }
  }
  if (ph == 0) {
    float* dst = p.partial + (long long)chunk * p.wsize;
    const int nch = nbase + ntile * 16 + lj;
#pragma unroll
    for (int s = 0; s < 9; ++s)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int mch = 4 * lk + r;
        if (mch < p.Uc && nch < p.Vc) dst[(long long)mch * p.m_stride + (long long)nch * p.n_stride + s] = acc[0][s][r];
      }
  }
}

template <int TH, int TW, int NCH_>
static int wlaunch_q(const WgradParams& p, hipStream_t stream) {
  using C = WgqCfg<TH, TW, NCH_>;
  static DevOnce attr_once;
  auto kern = wgrad_dense_q_k<TH, TW, NCH_>;
  if (attr_once.first()) {
    const hipError_t attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                              C::LDS_BYTES);
    if (attr_err != hipSuccess) {  // refused: report it here instead of an opaque launch failure later
      (void)hipGetLastError();
      attr_once.undo();
      return (int)attr_err;
    }
    if (rln_env("RLN_DEBUG_OCC")) {
      int nb = -1;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(kern), 256, C::LDS_BYTES);
      fprintf(stderr, "[rln] wgrad_dense_q_k<%dx%d,%d> lds %d B -> %d blocks/CU\n", TH, TW, NCH_, C::LDS_BYTES, nb);
    }
  }
  dim3 grid((unsigned)p.nchunks, 1u, (unsigned)((p.Vc + NCH_ - 1) / NCH_));
  hipLaunchKernelGGL(kern, grid, dim3(256), C::LDS_BYTES, stream, p);
  return (int)hipGetLastError();
}

template <int KS, int MT, int PRO, bool SHIFT_A, int TH, int TW>
static int wlaunch_t(const WgradParams& p, hipStream_t stream) {
  using C = WgCfg<KS, MT, PRO, SHIFT_A, TH, TW>;
  static DevOnce attr_once;
  auto kern = wgrad_k<KS, MT, PRO, SHIFT_A, TH, TW>;
  if (attr_once.first()) {
    const hipError_t attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                              C::LDS_BYTES);
    if (attr_err != hipSuccess) {  // refused: report it here instead of an opaque launch failure later
      (void)hipGetLastError();
      attr_once.undo();
      return (int)attr_err;
    }
    if (rln_env("RLN_DEBUG_OCC")) {
      int nb = -1;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(kern), 256, C::LDS_BYTES);
      fprintf(stderr, "[rln] wgrad_k<KS%d MT%d PRO%d SA%d %dx%d> lds %d B -> %d blocks/CU\n", KS, MT, PRO, (int)SHIFT_A, TH,
              TW, C::LDS_BYTES, nb);
    }
  }
  const int Mc = SHIFT_A ? p.Vc : p.Uc;
  const int Nc = SHIFT_A ? p.Uc : p.Vc;
  dim3 grid((unsigned)p.nchunks, (unsigned)((Mc + C::MCH - 1) / C::MCH), (unsigned)((Nc + 63) / 64));
  static const int dbg = rln_env("RLN_DBG") ? atoi(rln_env("RLN_DBG")) : 0;
  if (dbg & (256 | 128)) {
    WgradParams q = p;
    if (dbg & 256) q.xcd_remap = 1;
    // diagnostic build: phase stamps of the level-0 dense launches (RLN_DBG=128)
    if ((dbg & 128) && KS == 3 && PRO == PRO_BNRELU && p.GW >= 160) q.dbg_out = igemm_debug_buffer();
    hipLaunchKernelGGL(kern, grid, dim3(256), C::LDS_BYTES, stream, q);
    return (int)hipGetLastError();
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), C::LDS_BYTES, stream, p);
  return (int)hipGetLastError();
}

void wgrad_tile_dims(WgradKind, int tile, int* th, int* tw) {
  if (tile == 0) {
    *th = 4;
    *tw = 32;
  } else {
    *th = 8;
    *tw = 16;
  }
}

int wgrad_pick_tile(int gh, int gw) {
  const long long a0 = (long long)((gh + 3) / 4) * ((gw + 31) / 32);
  const long long a1 = (long long)((gh + 7) / 8) * ((gw + 15) / 16);
  return (a1 < a0) ? 1 : 0;
}

void wgrad_block_dims(WgradKind kind, int* m_per_block, int* n_per_block) {
  *m_per_block = (kind == WG_PW1) ? 64 : 16;
  *n_per_block = 64;
}

// V channels per block of the 16-byte staging kernel the dense launch will use (0: generic kernel, 64 per block)
int wgrad_dense_q_channels(const WgradParams& p) {
  static const bool noq = rln_env("RLN_NO_WGQ") != nullptr;
  static const int nch = rln_env("RLN_WGQ_NCH") ? atoi(rln_env("RLN_WGQ_NCH")) : 16;  // measured: 16 > 32 > 64 (+0.8 %, +4 %)
  const bool al = ((reinterpret_cast<uintptr_t>(p.u) | reinterpret_cast<uintptr_t>(p.v)) & 15) == 0;
  const bool q = !noq && al && p.Uc <= 16 && (p.GW % 4) == 0 && p.Wv == p.GW && p.Hv == p.GH &&
                 (p.u_cs % 4) == 0 && (p.v_cs % 4) == 0 && (p.u_ns % 4) == 0 && (p.v_ns % 4) == 0;
  static const int nch_wide = rln_env("RLN_WGQ_NCH_WIDE") ? atoi(rln_env("RLN_WGQ_NCH_WIDE")) : 0;  // levels >= 160 wide
  const int sel = (nch_wide && p.GW >= 160) ? nch_wide : nch;
  return q ? (sel == 64 ? 64 : (sel == 16 ? 16 : 32)) : 0;
}

int wgrad_launch(WgradKind kind, int tile, const WgradParams& p, hipStream_t stream) {
  switch (kind) {
    case WG_DENSE3: {
      const int nch = wgrad_dense_q_channels(p);
      if (nch == 16) return tile == 0 ? wlaunch_q<4, 32, 16>(p, stream) : wlaunch_q<8, 16, 16>(p, stream);
      if (nch == 32) return tile == 0 ? wlaunch_q<4, 32, 32>(p, stream) : wlaunch_q<8, 16, 32>(p, stream);
      if (nch == 64) return tile == 0 ? wlaunch_q<4, 32, 64>(p, stream) : wlaunch_q<8, 16, 64>(p, stream);
      return tile == 0 ? wlaunch_t<3, 1, PRO_BNRELU, false, 4, 32>(p, stream)
                       : wlaunch_t<3, 1, PRO_BNRELU, false, 8, 16>(p, stream);
    }
    case WG_RAW3:
      return tile == 0 ? wlaunch_t<3, 1, PRO_RAW, false, 4, 32>(p, stream)
                       : wlaunch_t<3, 1, PRO_RAW, false, 8, 16>(p, stream);
    case WG_PW1:
      return tile == 0 ? wlaunch_t<1, 4, PRO_BNRELU, false, 4, 32>(p, stream)
                       : wlaunch_t<1, 4, PRO_BNRELU, false, 8, 16>(p, stream);
    case WG_CONVT:
      return tile == 0 ? wlaunch_t<3, 1, PRO_S2D, true, 4, 32>(p, stream)
                       : wlaunch_t<3, 1, PRO_S2D, true, 8, 16>(p, stream);
  }
  return -1;
}

}  // namespace rln
