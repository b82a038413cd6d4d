// Dense-block 3x3 family on the 16-bit MFMA pipe (v_mfma_f32_16x16x32_{bf16,f16}) with fp32 storage.
//
// The DenseLayer convolutions of the path (layers.py:8-12: BN -> ReLU -> Conv2d(Cin,16,3) -> Dropout2d) are GEMMs
// with N = 16 output channels.  On the exact-fp32 MFMA (1/16 of the 16-bit rate) they are matrix-pipe bound; here every
// fp32 operand is split on the fly into NP 16-bit parts (x = x0 + x1 (+ x2), each part the rounding of the remainder)
// and the product is assembled from the leading cross terms, accumulated in fp32 inside the MFMA:
//   NP = 1: x0*w0                               (plain 16-bit operands: the bf16 throughput mode)
//   NP = 2: x0*w0 + x0*w1 + x1*w0               (bf16: ~2^-17 relative per term; f16: ~2^-22)
//   NP = 3: + x0*w2 + x1*w1 + x2*w0             (bf16: ~2^-25, below fp32 rounding)
// so the kernels become HBM-bound: activations stay fp32 NCHW in HBM, are read once per launch, transposed to a
// [pixel][channel] image in LDS while BN+ReLU is applied, and fed to the MFMA as 8-channel (16-byte) fragments.
#pragma once
#include "common.h"

namespace rln {

enum { D3_BF16 = 0, D3_F16 = 1 };

// ---- weight packing ---------------------------------------------------------------------------------------------
// One descriptor per dense layer; the pack kernel converts W[Cout][Cin][3][3] (fp32, parameter arena) into MFMA
// B-operand fragments, once per forward (the optimiser changes the weights every step):
//   forward  orientation  wf[chunk][s][part][lane] : n = lane&15 = output channel, k = 8*(lane>>4)+e,
//                          tap = 2s + (k>>4), input channel = 16*chunk + (k&15), value W[n][ch][tap]
//   backward orientation  wb[cgroup][s][part][lane]: n = lane&15 -> input channel 16*cgroup + n, k as above,
//                          tap' = 2s + (k>>4), o = k&15, value W[o][c][8 - tap']   (flipped taps)
// each entry 8 halfwords (16 bytes); taps >= 9 and out-of-range channels are zero.
struct D3PackDesc {
  long long w_off;      // into the parameter arena (floats)
  int cin, cout;
  long long wf_off;     // into the packed buffer (uint4 units); -1: skip
  long long wb_off;     // -1: skip
  int unit_begin;       // first (chunk|cgroup, s) unit of this layer in the flat unit list
  int n_units;          // 5 * ceil(cin/16) * (forward + backward)
};
inline long long d3_pack_entries(int cin, int np) { return (long long)((cin + 15) / 16) * 5 * np * 64; }
int d3_pack_weights(const float* params, const D3PackDesc* desc_dev, int n_desc, int total_units, uint4* packed,
                    int np, int dt, hipStream_t s);

// ---- forward ----------------------------------------------------------------------------------------------------
struct D3Fwd {
  const float* S;  // input view: channel 0 of the layer's input range, [N][.][H][W]
  long long ns;    // sample stride (floats)
  int cs;          // channel (plane) stride
  int H, W, Cin;
  const float* pa;  // BN folded scale / shift [Cin]
  const float* pb;
  const uint4* wpk;  // forward-orientation fragments of this layer
  const float* bias;    // [Cout]
  const float* nscale;  // [N][Cout] Dropout2d scale or null
  float* out;           // output view (channel 0 of the 16 new channels)
  long long out_ns;
  int out_cs, Cout;
  float* stat_partial;  // [N*tiles][Cout][2] or null
  int th, tw, tiles_x, tiles_y;
  int rg;               // rows per staging lane group (2 or 4), from d3_fwd_pick_tile
  int ksplit;           // >1: raw partial sums to out + split*split_stride (no bias / scale / statistics)
  long long split_stride;
  int dbg;              // diagnostic builds (-DRLN_DIAG) only: 1 no global loads, 2 no commit, 4 no MFMA phase, 8 no zero-init
  int st;               // storage element type of S and out (storage.h): ST_F32, or ST_BF16 (then np = 1, bf16 operands)
  // finishing launch of a pair (d3_fwd_launch): only the chunks from c_first on are multiplied; the raw sums of the
  // earlier chunks (fp32 [N][Cout][H][W], written by the paired launch of the previous layer) are added in the epilogue
  int c_first;
  const float* partial_in;
  // paired launch (d3_fwd_pair_launch): this layer (all Cin chunks; Cin % 16 == 0) and the NEXT layer of the block over the
  // same chunks: the next layer's BN table, forward fragments and raw-sum destination
  const float* pa2;
  const float* pb2;
  const uint4* wpk2;
  float* partial_out;
};
// both layers of a block over the channels they share, one load per chunk (see d3_fwd2_k); same tile plan as d3_fwd_launch
int d3_fwd_pair_launch(const D3Fwd& p, int N, int np, int dt, hipStream_t s);
// true when the launch geometry is supported (W % 4 == 0, 16-byte aligned planes, W >= 40, Cout <= 16)
bool d3_fwd_supported(const D3Fwd& p);
void d3_fwd_pick_tile(int H, int W, int np, int* th, int* tw, int* rg, int st = 0);
int d3_fwd_launch(const D3Fwd& p, int N, int np, int dt, hipStream_t s);
// the finishing launch of a pair (c_first, partial_in set; exactly one chunk left) as a light 4-wave kernel on 256-pixel
// tiles; its statistics partial rows number d3_fin_rows(H, W, N).  -4 when the geometry is not covered: the caller
// runs d3_fwd_launch instead.
bool d3_fin_supported(const D3Fwd& p, int np);
long long d3_fin_rows(int H, int W, int N);
int d3_fin_launch(const D3Fwd& p, int N, int np, int dt, hipStream_t s);

// ---- weight gradient -------------------------------------------------------------------------------------------------
// dW[o][c][tap] = sum_{n,p} dY[n][o][p] * relu(a[c]*S[n][c][p+tap] + b[c]): a GEMM whose K dimension is the pixel axis.
// Both operands are staged as the same split 16-bit [pixel][16 channels] LDS images the forward kernel uses (z with its
// halo, dY without) and read through ds_read_b64_tr_b16 (hardware transpose), so tap shifts are whole 32-byte pixel
// records.  A block owns ONE 16-channel chunk of z and a contiguous range of pixel tiles; its 9 tap accumulators stay in
// registers over the whole range.  Blocks of the same range (different chunks) are placed on one XCD so the dY tiles
// they all read come from that XCD's L2.
struct D3Wgrad {
  const float* S;  // layer input view (channel 0 of the input range)
  long long ns;
  int cs, H, W, Cin;
  const float* pa;
  const float* pb;
  const float* dY;  // [N][Cout][H][W]
  int Cout, N;
  int th, tw, tiles_x, tiles_y;
  int rg;
  int nchunks, nranges;
  float* partial;  // [nranges][Cout*Cin*9], layout [o][c][tap]
  int dbg;         // diagnostic builds (-DRLN_DIAG) only: 1 no global loads, 2 no commit, 4 no MFMA phase
  int st;          // storage element type of S and dY (storage.h)
  const void* dY16;  // fp32 stacks only: bf16 copy of dY written by grad_finalize (GradFinParams.dst16), or null
  int yt;            // ST_BF16 with st == ST_F32: read dY16 instead of dY (one bf16 part); otherwise ignored
  // two-layer launch (nl == 2, needs yt == ST_BF16): the secondary layer consumes the first Cin2 <= Cin of the same
  // input channels with its own BatchNorm table, dY copy and partial slabs
  int nl;
  int z8;          // bf16 stacks: 8-pixel units (16-byte loads) for z, dY (= dY16 / dY16_2, the bf16 dY buffers) and halo
  int Cin2;
  const float* pa2;
  const float* pb2;
  const void* dY16_2;
  float* partial2;  // [nranges][Cout*Cin2*9]
};
bool d3_wgrad_supported(const D3Wgrad& p);
void d3_wgrad_plan(int H, int W, int N, int Cin, D3Wgrad* p);  // fills th, tw, tiles, rg, nchunks, nranges
int d3_wgrad_launch(const D3Wgrad& p, int np, int dt, hipStream_t s);

// ---- data gradient of a dense block's INPUT channels ("pull" form) ----------------------------------------------------
// A DenseBlock's input channels [0, C) are consumed by every layer j of the block (layers.py:33,39).  Instead of one
// read-S / read-modify-write-G pass per layer, one launch handles all (up to D3_LMAX) layers at once: the layers' dY
// tiles are staged once per pixel tile as 16-bit [pixel][channel] images; for each 16-channel output group the
// contributions  gz_j[c] = relu'_j[c] * (W_j^T (*) dY_j)[c]  are accumulated in registers over j and
//   G[c] (+)= sum_j gamma_j[c] * gz_j[c]          is written ONCE,
// together with the BatchNorm-backward sums (sum gz_j, sum gz_j * xhat) per (layer, channel).
#define D3_LMAX 5
struct D3Pull {
  int nl;                         // layers in this launch (1..D3_LMAX)
  const float* dY[D3_LMAX];       // [N][Cout][H][W] finalised output gradients of the layers
  int Cout;
  const uint4* wpk[D3_LMAX];      // backward-orientation weight fragments of each layer
  const float* ea[D3_LMAX];       // BN folded scale / shift of layer j (ReLU mask), indexed by input channel
  const float* eb[D3_LMAX];
  const float* egamma[D3_LMAX];
  const float* mean;              // level statistics at the block's first input channel
  const float* invstd;
  const float* S;                 // activation stack view at the block's first input channel
  long long s_ns;
  int cs;                         // plane stride (= H*W)
  float* G;                       // gradient stack view (same geometry)
  int C;                          // input channels to produce
  int acc_lo, acc_hi;             // channels in [acc_lo, acc_hi) accumulate into G, the others overwrite
  int H, W, N;
  int th, tw, tiles_x, tiles_y;
  float* stat_partial;            // [nsub*blocks][nl][Cpad][2], Cpad = 16*ceil(C/16), nsub = d3_pull_nsub()
  int st;                         // storage element type of S and of the dY buffers (storage.h); G is fp32
  unsigned long long* dbg_out;    // diagnostic builds (-DRLN_DIAG): 8 phase cycle sums (see d3_pull_k), or null
};
bool d3_pull_supported(const D3Pull& p, int np);  // geometry + LDS budget (set nl, C, th, tw first)
void d3_pull_pick_tile(int H, int W, int* th, int* tw);
int d3_pull_blocks(const D3Pull& p);  // persistent grid size
int d3_pull_nsub(const D3Pull& p);    // stat_partial rows per block (2, or 5 for a one-group launch)
int d3_pull_launch(const D3Pull& p, int np, int dt, hipStream_t s);
// reduces the partial rows of a pull launch: dbeta_j[c] = sum gz_j, dgamma_j[c] = sum gz_j*xhat (c < C) and adds the
// gamma-weighted sums of all layers into the level accumulators S1/S2 (fixed summation order)
struct D3PullFin {
  int nl, C, Cpad, rows;
  const float* partial;  // [rows][nl][Cpad][2]
  const float* gamma[D3_LMAX];
  float* dgamma[D3_LMAX];
  float* dbeta[D3_LMAX];
  float* S1;
  float* S2;
};
int d3_pull_finalize(const D3PullFin& f, hipStream_t s);

// ---- data gradient of a dense layer into the block's OWN new channels ("looped" form on the 16-bit pipe) -------------
// Layer j of a block also consumes the 16 * j channels the earlier layers of the same block produced; they can only be
// finalised after layer j has contributed (BatchNorm-backward sums are grid-wide), so this contribution is a per-layer
// read-modify-write of G.  Same contraction as the exact-fp32 dgrad_loop_k (igemm.h) -- K = the layer's 16 dY channels
// x 9 flipped taps -- with the dY tile staged ONCE as a split 16-bit [pixel][16 channels] image and the backward-
// orientation weight fragments of d3_pack_k (5 K-steps of 16 channels x 2 taps per 16-channel output group).
struct D3Dgl {
  const float* dY;   // [N][K][H][W], the layer's finalised output gradient (storage type st)
  int K;             // its channels (<= 16)
  const uint4* wpk;  // backward-orientation fragments of the layer, at the first output group
  int J;             // output (new) channels
  const float* S;    // stack view at the first output channel (storage type st)
  long long s_ns;    // sample stride of S and G (elements)
  int cs;            // plane stride
  float* G;          // gradient stack view (fp32), same geometry
  const float* ea;   // BN folded scale / shift of the layer (ReLU mask), at the first output channel
  const float* eb;
  const float* emean;
  const float* einvstd;
  const float* egamma;
  int acc_lo, acc_hi;  // output channels in [acc_lo, acc_hi) accumulate into G, the others overwrite
  int H, W, N;
  int tile;            // 0: 8 x 32, 1: 16 x 16 pixel tiles
  int tiles_x, tiles_y;
  float* stat_partial;  // [N * tiles][J][2]
  int st;
};
bool d3_dgl_supported(const D3Dgl& p);
void d3_dgl_plan(int H, int W, D3Dgl* p);  // tile, tiles_x, tiles_y
inline long long d3_dgl_rows(const D3Dgl& p) { return (long long)p.tiles_x * p.tiles_y * p.N; }
int d3_dgl_launch(const D3Dgl& p, int np, int dt, hipStream_t s);

}  // namespace rln
