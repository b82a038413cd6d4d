// TransitionDown family (layers.py:45-58: BN -> ReLU -> Conv2d(C, C, 1) -> Dropout2d -> MaxPool2d(2)) on the 16-bit MFMA
// pipe with split fp32 operands (same arithmetic as dense3.h: NP parts, fp32 storage and accumulation).
//
// A 1x1 convolution has no halo, so the activations never pass through LDS: the MFMA's N index is a free labelling of
// pixels, and the labelling is chosen so that what a lane loads from the NCHW planes IS its B-operand fragment.  Lane
// (n = l&15, kb = l>>4) owns one 2x2 pooling window n of a 16-window wave tile and the 8 channels 8*kb..8*kb+7 of a
// 32-channel K step; the four window positions are the four N tiles of the wave.  The max-pool (forward), the
// un-pooling of the pooled gradient (backward) and the ReLU mask / BatchNorm-backward sums therefore all happen inside
// one lane's registers.  Weights are packed once per forward into A-operand fragments and kept in LDS by a persistent
// block, which walks the wave tiles (windows are numbered linearly over samples x window rows x window columns, so any
// even width and any height fits without per-level tile shapes).
#pragma once
#include "common.h"
#include "dense3.h"

namespace rln {

// ---- weight packing -------------------------------------------------------------------------------------------------
//   forward  fragments wf[mtile][kstep][part][lane]: lane (i = l&15, kb = l>>4) holds W[o = 16*mtile + i][c = 32*kstep + 8*kb + e]
//   backward fragments wb[mtile][kstep][part][lane]: lane (i, kb) holds W[o = 32*kstep + 8*kb + e][c = 16*mtile + i]
// each entry 8 halfwords (16 bytes), zero outside the matrix.
struct P1PackDesc {
  long long w_off;   // W[cout][cin] in the parameter arena (floats)
  int cin, cout;
  long long wf_off;  // into the packed buffer (uint4 units); -1: skip
  long long wb_off;
  int unit_begin;    // first (mtile, kstep) unit of this layer in the flat unit list (forward units first)
  int n_units;
};
inline int p1_units_f(int cin, int cout) { return ((cout + 15) / 16) * ((cin + 31) / 32); }
inline int p1_units_b(int cin, int cout) { return ((cin + 15) / 16) * ((cout + 31) / 32); }
int p1_pack_weights(const float* params, const P1PackDesc* desc_dev, int n_desc, int total_units, uint4* packed, int np,
                    int dt, hipStream_t s);

// ---- forward: conv 1x1 + bias + Dropout2d scale + MaxPool2d(2) + argmax index + statistics of the pooled map ---------
struct P1Fwd {
  const float* S;  // input view (channel 0 of the input range) [N][.][H][W]
  long long ns;
  int cs, H, W, Cin, N;
  const float* pa;  // BN folded scale / shift [Cin]
  const float* pb;
  const uint4* wpk;       // forward fragments
  const float* bias;      // [Cout]
  const float* nscale;    // [N][Cout] or null
  float* out;             // pooled output view [N][.][H/2][W/2]
  long long out_ns;
  int out_cs, Cout;
  unsigned char* pool_idx;  // [N][Cout][H/2][W/2]: 2*dy + dx of the (first) maximum
  float* stat_partial;      // [bpg][Cout][2] or null
  int mt, groups, bpg;      // M tiles per block, output-channel groups, blocks per group (p1_fwd_plan)
  int dbg;                  // diagnostic builds (-DRLN_DIAG) only: 1 no global loads, 2 no MFMA phase, 4 no epilogue
  int st, ot;               // storage element types of S and of out (storage.h)
};
bool p1_fwd_supported(const P1Fwd& p);
void p1_fwd_plan(P1Fwd* p, int np);  // fills mt, groups, bpg from Cin, Cout, N, H, W
int p1_fwd_launch(const P1Fwd& p, int np, int dt, hipStream_t s);

// ---- data gradient: G[c] (+)= gamma[c] * relu'(c) * sum_o W[o][c] * unpool(dYp)[o], plus the BatchNorm-backward sums ---
struct P1Dgrad {
  const float* dYp;               // finalised gradient of the pooled output [N][Cout][H/2][W/2]
  const unsigned char* pool_idx;  // [N][Cout][H/2][W/2]
  int Cout;
  const uint4* wpk;  // backward fragments
  const float* ea;   // BN folded scale / shift (ReLU mask), gamma, level mean / invstd: indexed by input channel
  const float* eb;
  const float* egamma;
  const float* mean;
  const float* invstd;
  const float* S;  // input view (pre-BN activations)
  long long ns;
  int cs;
  float* G;  // gradient stack view, same geometry as S
  int C, acc_lo, acc_hi;
  int H, W, N;
  float* stat_partial;  // [bpg][C][2]: sum gz, sum gz * xhat
  int mt, groups, bpg;
  int st, yt;  // storage element types of S and of dYp (storage.h); G is fp32
};
bool p1_dgrad_supported(const P1Dgrad& p);
void p1_dgrad_plan(P1Dgrad* p, int np);
int p1_dgrad_launch(const P1Dgrad& p, int np, int dt, hipStream_t s);

// ---- weight gradient: dW[o][c] = sum_{n,p} unpool(dYp)[n][o][p] * relu(a[c]*S[n][c][p] + b[c]) ---------------------------
struct P1Wgrad {
  const float* dYp;
  const unsigned char* pool_idx;
  int Cout;
  const float* S;
  long long ns;
  int cs, H, W, N, Cin;
  const float* pa;
  const float* pb;
  float* partial;  // [nranges][Cout][Cin]
  int mo;          // M tiles (16 output channels each) per block
  int ogroups, cblocks, nranges, per;  // output-channel groups, 128-input-channel blocks, K ranges of `per` slabs
  int st, yt;      // storage element types of S and of dYp (storage.h)
};
bool p1_wgrad_supported(const P1Wgrad& p);
void p1_wgrad_plan(P1Wgrad* p);  // fills mo, ogroups, cblocks, nranges, per
int p1_wgrad_launch(const P1Wgrad& p, int np, int dt, hipStream_t s);

}  // namespace rln
