// TransitionUp family (layers.py:58-68: ConvTranspose2d(C, C, 3, stride 2) + centre crop to the skip) on the 16-bit MFMA
// pipe with split fp32 operands (arithmetic of dense3.h; fp32 storage and accumulation).
//
// Same operand idea as pw1.h: the B operand comes straight from the NCHW planes.  A lane owns one input pixel (y, x) = one
// 2x2 block of output pixels (rows 2y, 2y+1; columns 2x, 2x+1), 15 consecutive pixels per 16-lane wave tile plus one
// halo lane, and the 8 channels of its K-step block.  The stride-2 structure then lives in registers:
//   forward        out[2y+py][2x+px] gathers x at (y, x), (y, x-1), (y-1, x), (y-1, x-1): the lane loads its own pixel
//                  and the one above; the left neighbours are the previous lane's fragments (DPP row shift);
//   data gradient  dx[y][x] gathers dU rows 2y..2y+2, columns 2x..2x+2: three 8-byte row loads per channel, the third
//                  column is the next lane's first (DPP);
//   weight gradient K = pixels: x rows are staged once per block through LDS, every wave owns one (output-channel tile,
//                  kernel row) pair and de-interleaves its dU row segment into the three kernel-column fragments.
// Weights are packed once per forward into A-operand fragments per tap and held in LDS by persistent blocks.
#pragma once
#include "common.h"
#include "dense3.h"

namespace rln {

// ---- weight packing: W[cin][cout][3][3] --------------------------------------------------------------------------------
//   forward  wf[mtile][kstep][tap][part][lane]: lane (i = l&15, kb = l>>4) holds W[c = 32*kstep + 8*kb + e][o = 16*mtile + i][tap]
//   backward wb[group][kstep][ky][m][kx][part][lane] (mtile = group * c3_dgrad_mt + m): lane (i, kb) holds
//            W[c = 16*mtile + i][o = 32*kstep + 8*kb + e][3*ky + kx]
struct C3PackDesc {
  long long w_off;
  int cin, cout;
  long long wf_off, wb_off;  // uint4 units into the packed buffer; -1: skip
  int unit_begin, n_units;   // units = (mtile, kstep, tap) triples, forward first
};
// M tiles (16 input channels each) per block of the data-gradient kernel: balanced groups of at most 5
__host__ __device__ inline int c3_dgrad_mt(int m_tiles) {
  const int groups = (m_tiles + 4) / 5;
  return (m_tiles + groups - 1) / groups;
}
inline int c3_units_f(int cin, int cout) { return ((cout + 15) / 16) * ((cin + 31) / 32) * 9; }
inline int c3_units_b(int cin, int cout) { return ((cin + 15) / 16) * ((cout + 31) / 32) * 9; }
// uint4 entries of the backward fragment buffer (groups are padded to c3_dgrad_mt tiles)
inline long long c3_entries_b(int cin, int cout, int np) {
  const int MT = (cin + 15) / 16, mtb = c3_dgrad_mt(MT), groups = (MT + mtb - 1) / mtb;
  return (long long)groups * mtb * ((cout + 31) / 32) * 9 * np * 64;
}
int c3_pack_weights(const float* params, const C3PackDesc* desc_dev, int n_desc, int total_units, uint4* packed, int np,
                    int dt, hipStream_t s);

// ---- forward ----------------------------------------------------------------------------------------------------------
struct C3Fwd {
  const float* X;  // input view [N][.][H][W]
  long long ns;
  int cs, H, W, Cin, N;
  const uint4* wpk;
  const float* bias;
  float* out;  // output view [N][.][Ho][Wo], top-left crop of the (2H+1) x (2W+1) transposed convolution
  long long out_ns;
  int out_cs, Cout, Ho, Wo;
  float* stat_partial;  // [bpg][Cout][2] (sum, sum of squares of what was written) or null
  int mt, groups, bpg;
  int st, ot;  // storage element types of X and of out (storage.h)
};
bool c3_fwd_supported(const C3Fwd& p);
bool c3_fwd_fits(const C3Fwd& p, int np);  // one M tile (9 taps x K steps x parts) fits the LDS budget
void c3_fwd_plan(C3Fwd* p, int np);
int c3_fwd_launch(const C3Fwd& p, int np, int dt, hipStream_t s);

// ---- data gradient: G[c][y][x] = cscale[c] * sum_{o,ky,kx} dU[o][2y+ky][2x+kx] * W[c][o][ky][kx] (overwrite) ---------------
struct C3Dgrad {
  const float* dU;  // [N][Cout][Ho][Wo] contiguous
  int Cout, Ho, Wo;
  const uint4* wpk;      // backward fragments
  const float* cscale;   // [C] or null
  float* G;              // output view [N][.][H][W]
  long long ns;
  int cs, H, W, C, N;
  int mt, groups, bpg;  // mt = c3_dgrad_mt(ceil(C/16))
  int yt;               // storage element type of dU (storage.h); G is fp32
};
bool c3_dgrad_supported(const C3Dgrad& p);
void c3_dgrad_plan(C3Dgrad* p);
int c3_dgrad_launch(const C3Dgrad& p, int np, int dt, hipStream_t s);

// ---- weight gradient: dW[c][o][ky][kx] = sum_{n,y,x} X[n][c][y][x] * dU[n][o][2y+ky][2x+kx] -------------------------------
struct C3Wgrad {
  const float* X;
  long long ns;
  int cs, H, W, Cin, N;
  const float* dU;
  int Cout, Ho, Wo;
  float* partial;  // [nranges][Cin][Cout][9]
  int mo, nc;      // output-channel tiles (<= 2: the block has 3*mo waves) and input-channel tiles (<= 5) per block
  int ogroups, cgroups, nranges, per;  // per = K steps (32 input pixels) per range
  int st, yt;      // storage element types of X and of dU (storage.h)
};
bool c3_wgrad_supported(const C3Wgrad& p);
void c3_wgrad_plan(C3Wgrad* p);
int c3_wgrad_launch(const C3Wgrad& p, int np, int dt, hipStream_t s);

}  // namespace rln
