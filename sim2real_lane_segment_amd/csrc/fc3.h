// First convolution (tiramisu.py:33-35: Conv2d(in_channels, 48, 3, padding 1) on the raw input) on the 16-bit MFMA pipe
// with split fp32 operands (arithmetic of dense3.h).  K = Cin * 9 <= 32 is ONE MFMA K step, so the im2col never exists:
// a lane gathers its pixels' 3x3 neighbourhoods straight from the input planes into the B fragment (forward), or the
// shifted pixel runs of one (channel, tap) column into the B fragment of the pixel-K weight gradient.
#pragma once
#include "common.h"
#include "dense3.h"

namespace rln {

struct F3Fwd {
  const float* X;  // [N][Cin][H][W] contiguous
  int Cin, H, W, N;
  const float* w;     // [Cout][Cin][3][3]
  const float* bias;  // [Cout]
  float* out;         // output view [N][.][H][W]
  long long out_ns;
  int out_cs, Cout;
  float* stat_partial;  // [blocks][Cout][2] or null
  int blocks;
  int ot;  // storage element type of out (storage.h); X is the fp32 input batch
};
bool f3_fwd_supported(const F3Fwd& p);
void f3_fwd_plan(F3Fwd* p);
int f3_fwd_launch(const F3Fwd& p, int np, int dt, hipStream_t s);

struct F3Wgrad {
  const float* X;
  int Cin, H, W, N;
  const float* dY;  // [N][Cout][H][W] contiguous
  int Cout;
  float* partial;  // [blocks][Cout][Cin*9]
  int blocks;
  int yt;  // storage element type of dY (storage.h)
};
bool f3_wgrad_supported(const F3Wgrad& p);
void f3_wgrad_plan(F3Wgrad* p);
int f3_wgrad_launch(const F3Wgrad& p, int np, int dt, hipStream_t s);

}  // namespace rln
