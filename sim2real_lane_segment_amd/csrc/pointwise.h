// HBM-bound kernels of the path: BatchNorm statistics / affine folding, gradient finalisation,
// head (normalize + classifier + softmax), loss, AdamW, dropout masks.
#pragma once
#include "common.h"

namespace rln {

// ---- BatchNorm bookkeeping -----------------------------------------------------------------
// partial: [nblk][J][2] (sum, sum of squares) written by a producing kernel; finalises
// mean / biased var / invstd for J channels into level arrays at channel offset.
int bn_finalize(const float* partial, long long nblk, int J, double count, float eps, float* mean, float* var,
                float* invstd, float* stdv, hipStream_t s);
// folds a BatchNorm2d into per-channel (a, b): train -> batch stats (+ running-stat update, momentum,
// unbiased var); eval -> running stats.
// bn_finalize of the new channels + bn_prep (training form) of the next BatchNorm over [0, C) (new at [new_lo, new_lo+Jnew))
int bn_finalize_prep(const float* partial, long long nblk, int Jnew, int new_lo, double count, float eps, float* mean,
                     float* var, float* invstd, float* stdv, int C, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, float momentum, float* a, float* b, hipStream_t s);
int bn_prep(int training, int C, const float* gamma, const float* beta, const float* mean, const float* var,
            const float* invstd, float* running_mean, float* running_var, float momentum, double count, float eps,
            float* a, float* b, hipStream_t s);
// backward: reduces [nblk][J][2] partials (sum gy, sum gy*xhat) -> dgamma/dbeta and adds gamma-weighted
// sums into the level accumulators S1/S2.
int bn_bwd_finalize(const float* partial, long long nblk, int J, const float* gamma, float* dgamma, float* dbeta,
                    float* S1, float* S2, hipStream_t s);

// ---- gradient finalisation -----------------------------------------------------------------
struct GradFinParams {
  const float* S;  // activation view (channel 0 of the range)
  const float* G;  // raw gradient accumulation view
  long long ns;    // sample stride of S and G
  int C, H, W;     // channels in range, spatial dims of S/G
  const float* mean;
  const float* invstd;
  const float* S1;
  const float* S2;
  float invM;
  const float* nscale;  // [N][C] or null
  float* dst;           // [N][C][Hd][Wd] contiguous
  float* bias_partial;  // [rows][C] or null ; rows = N * blocks per plane
  // pool-backward variant: dst is the pre-pool map (Hd x Wd), S/G are the pooled maps (H x W)
  const unsigned char* pool_idx;  // [N][C][H][W] or null
  int Hd, Wd;
  int st, yt;  // storage element types of S and of dst (storage.h); G is fp32
  int vec4;     // set by grad_finalize(): 16-byte path legal (plane % 4 == 0, aligned views)
  void* dst16;  // optional second copy of dst rounded to bf16 (the dense weight gradient's one-part operand), or null
};
// returns number of bias_partial rows through *rows
int grad_finalize(const GradFinParams& p, int N, long long* rows, hipStream_t s);
long long grad_finalize_rows(int N, int Hd, int Wd);

// split-K finish: out[n][j][p] = (sum_s part[s][n][j][p] + bias[j]) * nscale[n][j]; stats partial [N*gx][J][2]
int splitk_finish(const float* part, int nsplit, long long split_stride, int N, int J, int HW, const float* bias,
                  const float* nscale, float* out, long long out_ns, float* stat_partial, long long* nblk,
                  hipStream_t s);

// dst[e] = sum_r src[r*len + e]   (fixed order)
// end of a dense layer's backward: bn_bwd_finalize + reduce_rows(weight slabs) + reduce_rows(bias rows), one launch
struct DenseTail {
  const float* bn_partial; long long bn_rows; int J;
  const float* gamma; float* dgamma; float* dbeta; float* S1; float* S2;
  const float* w_src; long long w_rows, w_len; float* w_dst;
  const float* b_src; long long b_rows, b_len; float* b_dst;
  long long nA, nB; int w_tall, b_tall;  // filled by dense_tail()
};
int dense_tail(DenseTail t, hipStream_t s);
int reduce_rows(const float* src, long long rows, long long len, float* dst, hipStream_t s);

// ---- head ----------------------------------------------------------------------------------
struct HeadParams {
  const float* S;  // [N][C][HW] view of the final stack
  long long ns;
  int C, HW, ncls;
  const float* w;  // [ncls][C]
  const float* b;  // [ncls]
  float T;
  int st;  // storage element type of S (storage.h)
};
int head_forward(const HeadParams& p, int N, float* out, int use_softmax, float* feat_out, hipStream_t s);
// classifier alone on given features [N][C][HW] contiguous
int classifier_forward(const HeadParams& p, int N, float* out, int use_softmax, hipStream_t s);

struct LossScratch {   // device scratch owned by the ctx
  int* counts;         // [ncls + 1] (last = bad labels)
  float* partial;      // [nblk][4]
  float* result;       // [0]=num [1]=den [2]=correct ; weights at [4..4+ncls)
};
int loss_forward(const float* probs, const long long* y, int N, int ncls, int HW, int weighted, const LossScratch& sc,
                 float* out, long long* argmax_out, long long* confusion_out, hipStream_t s);
long long loss_blocks(long long npix);

// adentropy (MMETrainingModule.py:10-11): out[0] = lamda * mean_pixels(sum_k p*log(p+1e-5))
int entropy_forward(const float* probs, int N, int ncls, int HW, float lamda, const LossScratch& sc, float* out,
                    hipStream_t s);
// torch.optim.SGD with momentum + nesterov (dampening 0) on a flat range
int sgd_nesterov(float* p, const float* g, float* buf, long long count, float lr, float momentum, float wd,
                 int first_step, float grad_scale, hipStream_t s);

struct HeadBwdParams {
  int mode;        // 0: class-weighted CE on probabilities, 1: entropy (adentropy), 2: external d(loss)/d(probs) in gext
  const float* gext;  // mode 2: [N][ncls][HW]
  float lamda;     // entropy weight
  float inv_count; // 1 / (N*H*W)
  float feat_sign; // -1 when a gradient-reversal layer sits between features and classifier
  HeadParams h;
  const long long* y;
  const float* lossres;  // LossScratch.result
  float loss_scale;
  const float* loss_scale_dev;  // optional device scalar multiplied on top (d(loss) handed in by autograd)
  float* G;              // gradient view [N][C][HW] (written: gx / invstd)
  long long g_ns;
  const float* invstd;   // [C]
  float* glin;           // scratch [N][ncls][HW]
  float* bias_partial;   // [blocks][ncls]
};
int head_backward_data(const HeadBwdParams& p, int N, long long* bias_rows, hipStream_t s);
// dW partial [N][ncls*C]
int head_backward_weight(const HeadParams& h, int N, const float* glin, float* partial, hipStream_t s);
// one-pass form: feature gradient + partial rows of dW ([rows][ncls*C]) and db ([rows][ncls]); C <= 512
long long head_backward_rows(int N, int HW);
int head_backward_fused(const HeadBwdParams& p, int N, float* wpartial, long long* rows, hipStream_t s);

// ---- EncDecNet pieces (models/EncDecNet.py:29-37,68-69,97,100-112) ----------------------------------------
// a,b from per-channel (sum, sumsq) of the activated conv output (train) or running stats (eval); updates running stats
int bn_affine_from_sums(const float* sums, int C, double count, int training, const float* gamma, const float* beta,
                        float* running_mean, float* running_var, float momentum, float eps, float* a, float* b,
                        hipStream_t s);
// y = mask * (a[c]*x + b[c]); then MaxPool2d(k, stride 2, padding k/2)
int bn_drop_maxpool(const float* x, int N, int C, int H, int W, const float* a, const float* b, const float* mask,
                    int k, float* out, hipStream_t s);
// y = mask * (a[c]*x + b[c]); then UpsamplingBilinear2d(scale_factor=2) (align_corners=True)
int bn_drop_upsample2(const float* x, int N, int C, int H, int W, const float* a, const float* b, const float* mask,
                      float* out, hipStream_t s);
// out = softmax over channels of x / T (use_softmax) or x / T
int softmax_channels(const float* x, int N, int C, int HW, float* out, hipStream_t s, float T = 1.f, int use_softmax = 1);
// input transform (myTransforms.py:15-19): uint8 HWC frames -> resized, normalised float CHW (+ nearest-resized labels)
int preprocess_u8(const unsigned char* frames, int N, int hs, int ws, const unsigned char* labels, int h, int w, int gray,
                  const float* mean3, const float* std3, float* x, long long* y, hipStream_t s);
// augmenting branch (myTransforms.py:8-13); params: device [N][80] (layout in pointwise.hip), tmp: device [N][h][w][3] bytes
int augment_u8(const unsigned char* frames, int N, int hs, int ws, const unsigned char* labels, int h, int w,
               const float* params, const float* mean3, const float* std3, unsigned char* tmp, float* x, long long* y,
               hipStream_t s);

// demo overlay (makeDemoVideo.py:36-46): argmax + bilinear-resized frame + class colours, uint8 HWC out
int overlay_u8(const unsigned char* frames, int N, int hs, int ws, const float* probs, int ncls, int h, int w,
               const unsigned char* colors_host, unsigned paint_mask, unsigned char* out, unsigned char* pred_out,
               hipStream_t s);

// ---- misc ----------------------------------------------------------------------------------
int adamw(float* p, const float* g, float* m, float* v, long long count, float lr, float b1, float b2, float eps,
          float wd, int step, float grad_scale, hipStream_t s);
int dropout_scales(float* dst, long long count, float keep, unsigned long long seed, hipStream_t s);
int add_one_i64(long long* p, long long count, hipStream_t s);

}  // namespace rln
