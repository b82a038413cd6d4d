// Implicit-GEMM convolution family on v_mfma_f32_16x16x4_f32 (exact fp32).
//
// One kernel template covers every "pixel tile x output channels" contraction of the path:
//   conv3x3 / conv1x1 forward with the BatchNorm affine + ReLU folded into LDS staging
//   (layers.py:8-11,46-49), MaxPool2d(2) epilogue (layers.py:52), ConvTranspose2d forward by output
//   parity class (layers.py:61-63), and the data gradients of all three.
#pragma once
#include "common.h"

namespace rln {

enum { PRO_RAW = 0, PRO_BNRELU = 1, PRO_S2D = 2 };
enum { EPI_STORE = 0, EPI_POOL = 1, EPI_DGRAD = 2 };
enum { TM_ID = 0, TM_FLIP = 1, TM_CONVT = 2 };

struct IgemmParams {
  // K-side operand (the tensor the taps slide over), logical channels [0,K) of an NCHW view
  const float* in;
  long long in_ns;  // sample stride (elements)
  int in_cs;        // channel stride (elements) ; rows are Win apart
  int Hin, Win, K;
  const float* pa;  // PRO_BNRELU: z = max(pa[c]*x + pb[c], 0)
  const float* pb;
  // weights: element (j, k, tap) at w[j*w_js + k*w_ks + tap]
  const float* w;
  long long w_js, w_ks;
  int tapmode;
  int J;  // output channels
  // tile grid space (= output grid, except convT-forward where out = 2*g + parity)
  int GH, GW, tiles_x, tiles_y, ncls;
  // output view
  float* out;
  long long out_ns;
  int out_cs, Hout, Wout;
  int out_vec;          // 1: rows are 16-byte aligned (float4 path legal)
  const float* bias;    // [J] or null
  const float* nscale;  // [N][J] or null (Dropout2d scale per sample/channel)
  const float* cscale;  // [J] or null
  float* stat_partial;  // [blocks][J][2] or null
  unsigned char* pool_idx;  // EPI_POOL: [N][J][Hout][Wout]
  // EPI_DGRAD extras: S = activation view the BN of this layer normalised, G(out) accumulates gamma*gy
  const float* S;
  long long s_ns;
  const float* ea;
  const float* eb;
  const float* emean;
  const float* einvstd;
  const float* egamma;
  int acc_lo, acc_hi;  // channels in [acc_lo,acc_hi) accumulate into out, others overwrite
  int act;             // EPI_STORE activation after bias: 0 none, 1 relu, 2 prelu/leaky(act_param), 3 sigmoid, 4 tanh
  float act_param;
  int ksplit;          // >1: input-channel chunks split over blockIdx.y; raw partial sums go to out + split*split_stride
  long long split_stride;
  int dbg;             // timing ablations only (RLN_DBG): 1 skip global loads, 2 skip LDS commit, 4 skip MFMA, 16 stamps, 32 no XCD remap
  unsigned long long* dbg_out;  // [8] phase cycle sums (diagnostic build path only)
  int st;              // dgrad_loop_launch only: storage element type of `in` (dY) and S (storage.h); 0 = fp32 elsewhere
  // Sample packing for levels much smaller than a tile (7x10, 3x5 ...): pk_sx > 1 samples sit side by side in one
  // VIRTUAL image of width pk_sx * (pk_w + 1) - 1, separated by one column that is zero AFTER the activation (= the
  // convolution's own padding), so a 32-wide tile carries pk_sx samples instead of one.  The kernel then runs over
  // ceil(pk_n / pk_sx) virtual samples; GH/GW/Hin/Win describe the virtual image, pk_w / pk_n the real row width and
  // sample count.  Only the staging and epilogue address maps know about it: per output pixel the contraction and its
  // order are unchanged, so results are bit-identical to the unpacked launch.  dense forward (scalar staging, raw
  // split-K epilogue) and dgrad_loop (VEC = false) only.
  int pk_sx, pk_w, pk_n;
};
// samples per virtual image for a level of width w in tiles of width tw (1 = no packing)
inline int igemm_pack_sx(int h, int w) { return (h <= 8 && w + 1 <= 16) ? 33 / (w + 1) : 1; }
unsigned long long* igemm_debug_buffer();  // device buffer of 8 counters (allocated on first use)

// which compiled variant
enum IgemmKind {
  IG_CONV3_BN = 0,   // KS3 NT1 PRO_BNRELU EPI_STORE   dense layer forward
  IG_CONV3_RAW = 1,  // KS3 NT1 PRO_RAW    EPI_STORE   first conv, convT forward (ncls=4)
  IG_CONV1_POOL = 2, // KS1 NT4 PRO_BNRELU EPI_POOL    transition down forward
  IG_DGRAD3 = 3,     // KS3 NT4 PRO_RAW    EPI_DGRAD   dense layer data gradient
  IG_DGRAD1 = 4,     // KS1 NT4 PRO_RAW    EPI_DGRAD   transition down data gradient
  IG_S2D3 = 5,       // KS3 NT4 PRO_S2D    EPI_STORE   convT data gradient
  IG_CONV1_BN = 6,   // KS1 NT4 PRO_BNRELU EPI_STORE   1x1 conv without pool (generic / tests)
  IG_CONV7_RAW = 7,  // KS7 NT1 PRO_RAW    EPI_STORE   EncDecNet 7x7 convolutions
  IG_CONV1_RAW = 8,  // KS1 NT1 PRO_RAW    EPI_STORE   EncDecNet 1x1 classifier
  IG_CONVT4 = 9,     // KS3 NT1 PRO_RAW    EPI_STORE   convT forward, all four output-parity classes per block (ncls = 1, TM_ID)
};

// tile: 0 -> 8x32, 1 -> 16x16, 5 -> 4x32 (dense forward and dgrad_loop without the 16-byte path only)
int igemm_launch(IgemmKind kind, int tile, const IgemmParams& p, int N, hipStream_t stream);
void igemm_tile_dims(IgemmKind kind, int tile, int* th, int* tw);
int igemm_pick_tile(int gh, int gw);
int igemm_pick_strip_tile(int gw);  // 2: 4x160, 3: 8x80, -1: none (dense forward only)
// number of stat-partial blocks a launch produces
inline long long igemm_stat_blocks(const IgemmParams& p, int N) {
  return (long long)p.ncls * p.tiles_x * p.tiles_y * N;
}

// dense-layer data gradient in looped form (K <= 16): same parameters as IG_DGRAD3, 4 stat-partial rows per block
int dgrad_loop_launch(int tile, const IgemmParams& p, int N, hipStream_t stream);

// ---------------------------------------------------------------------------------------------
// weight gradient family: dW[m][n][tap] = sum_pixels U[m][p] * V[n][p + tap]  (or roles swapped)
struct WgradParams {
  // U: un-shifted operand (raw), V: shifted operand (PRO_RAW / PRO_BNRELU / PRO_S2D)
  const float* u;
  long long u_ns;
  int u_cs, Uc;  // Uc channels
  const float* v;
  long long v_ns;
  int v_cs, Vc, Hv, Wv;
  const float* pa;
  const float* pb;
  int GH, GW, tiles_x, tiles_y;  // tile grid (U's spatial dims)
  int N;
  int items_per_chunk, nchunks;
  float* partial;         // [nchunks][wsize]
  long long wsize;
  long long m_stride, n_stride;  // output index = m*m_stride + n*n_stride + tap
  int xcd_remap;          // timing ablation only (RLN_DBG=256): XCD-aware chunk order
  unsigned long long* dbg_out;  // diagnostic build only: phase cycle sums
};
enum WgradKind {
  WG_DENSE3 = 0,  // M=U (dY 16ch), N=V (z = relu(a*S+b)), 9 taps
  WG_RAW3 = 1,    // as above, V raw (first conv)
  WG_PW1 = 2,     // 1x1: M=U (dY), N=V (z), MT=4
  WG_CONVT = 3,   // M=V (dU, S2D, 16ch), N=U (convT input raw), 9 taps
};
int wgrad_launch(WgradKind kind, int tile, const WgradParams& p, hipStream_t stream);
void wgrad_tile_dims(WgradKind kind, int tile, int* th, int* tw);
int wgrad_pick_tile(int gh, int gw);
void wgrad_block_dims(WgradKind kind, int* m_per_block, int* n_per_block);
int wgrad_dense_q_channels(const WgradParams& p);  // 0 / 32 / 64: V channels per block the dense launch will use

}  // namespace rln
