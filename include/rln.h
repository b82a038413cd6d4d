/*
 * rln.h -- C ABI of the MI355X-native lane-segmentation hot path ("right lane network").
 *
 * Drop-in boundary for the FC-DenseNet training / inference path of
 * timurlenk07/sim2real_lane_segment.  The reference has no native code: each entry point below
 * replaces a group of torch ops at the cited reference lines (paths relative to
 * rightLaneNetwork/).  Python binds this with ctypes (sim2real_lane_segment_amd/_lib.py).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter name ends in _host;
 *   - tensors are contiguous NCHW fp32, labels are int64 (torch.long);
 *   - `stream` is a hipStream_t passed as void* (0 = default stream); all work is enqueued on it,
 *     nothing synchronises the device;
 *   - every function returns 0 on success, <0 for an rln error, >0 for a hipError_t;
 *     rln_last_error() returns a thread-local message;
 *   - one rln_ctx per (process, device); a ctx is not thread-safe.
 */
#ifndef RLN_H_
#define RLN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RLN_MAX_BLOCKS 8

enum rln_error {
  RLN_OK = 0,
  RLN_ERR_ARG = -1,       /* bad argument / shape */
  RLN_ERR_STATE = -2,     /* call order (nothing bound, no forward before backward, ...) */
  RLN_ERR_WORKSPACE = -3, /* workspace missing or too small */
  RLN_ERR_UNSUPPORTED = -4
};

/* Constructor knobs of FCDenseNetFeatureExtractor / FCDenseNetClassifier
 * (models/FCDenseNet/tiramisu.py:21-24,112-118) + the constants of the layers
 * (layers.py:8,12: BatchNorm2d eps 1e-5 momentum 0.1, Dropout2d 0.2). */
typedef struct rln_config {
  int in_channels;
  int n_down;
  int down_blocks[RLN_MAX_BLOCKS];
  int n_up;
  int up_blocks[RLN_MAX_BLOCKS];
  int bottleneck_layers;
  int growth_rate;
  int first_conv_channels;
  int n_classes;
  float temperature;
  float bn_eps;
  float bn_momentum;
  float drop_p;
} rln_config;

typedef struct rln_ctx rln_ctx;

/* tensor kinds reported by rln_tensor_info */
enum { RLN_T_PARAM = 0, RLN_T_RUNNING_MEAN = 1, RLN_T_RUNNING_VAR = 2, RLN_T_NUM_BATCHES = 3 };

const char* rln_last_error(void);
int rln_version(void);

/* Builds the execution plan for a net (replaces the module construction in
 * trainingModules/TrainingBase.py:27-40 / tiramisu.py:21-87). Host-only, allocates nothing on the device. */
int rln_create(const rln_config* cfg, rln_ctx** out);
void rln_destroy(rln_ctx* ctx);

/* ---- parameter arenas ------------------------------------------------------------------
 * Parameters live in ONE flat fp32 arena laid out in forward execution order (so that gradients
 * complete back-to-front during backward and DDP buckets are contiguous slices); BN running
 * statistics live in a second flat fp32 arena, num_batches_tracked in a flat int64 arena.
 * rln_tensor_info enumerates every state_dict entry (434 for FCDenseNet67, reference key names)
 * with its arena offset, so the host can create views with the reference's names and shapes. */
int rln_num_tensors(const rln_ctx* ctx);
int64_t rln_param_count(const rln_ctx* ctx);  /* floats in the parameter (and gradient) arena */
int64_t rln_bnstat_count(const rln_ctx* ctx); /* floats in the running-stat arena */
int64_t rln_nbt_count(const rln_ctx* ctx);    /* int64 entries (one per BatchNorm2d) */
int rln_tensor_info(const rln_ctx* ctx, int idx, char* name, int name_cap, int* kind, int64_t* offset,
                    int* ndim, int64_t shape[4]);
int rln_feature_channels(const rln_ctx* ctx);
/* number of Dropout2d calls in one forward and total channels over them (mask layout: call-major,
 * each call [N, C_call]) */
int rln_num_dropouts(const rln_ctx* ctx);
int64_t rln_dropout_channels(const rln_ctx* ctx, int* per_call /* may be NULL, else [num_dropouts] */);

/* Arithmetic of the DenseLayer 3x3 convolutions (86 % of the path's FLOPs).  Storage is fp32 in every mode.
 *   parts = 0 : exact fp32 MFMA (v_mfma_f32_16x16x4_f32), the round-1 kernels;
 *   parts = 1..3 : 16-bit MFMA on operands split into `parts` bf16 (dtype 0) / f16 (dtype 1) parts, fp32 accumulate
 *                  (csrc/dense3.h; 2 parts = 3 products ~ 2^-17 (bf16) / 2^-22 (f16), 3 bf16 parts = 6 products < fp32 eps).
 * fwd_* selects the forward kernels, bwd_* the data / weight gradient kernels.  Call before rln_workspace_bytes. */
int rln_set_dense_arith(rln_ctx* ctx, int fwd_parts, int fwd_dtype, int bwd_parts, int bwd_dtype);
/* Operand parts of the dense 3x3 weight-gradient GEMMs (d3_wgrad_k).  Every entry of dW is a sum over N*H*W pixels
 * (>= 2400 where this kernel runs), over which the 16-bit rounding of the operands averages out: with ONE bf16 part
 * (plain bf16 MFMA, fp32 accumulation) the gradient error against the exact-fp32 family is 5.8e-4 (relative L2 over all
 * parameters, FCDenseNet67 2x120x160) against 4.3e-4 with two parts (tests/test_gpu_geometries.py).  rln_set_dense_arith
 * resets it to 1 for two-part backward arithmetic and to bwd_parts otherwise; 0 = bwd_parts. */
int rln_set_wgrad_parts(rln_ctx* ctx, int parts);

int rln_get_wgrad_parts(const rln_ctx* ctx);
/* Storage element type of the activation stacks and of the finalised output gradients in HBM (csrc/storage.h).
 *   0 : fp32 everywhere (default; the parity mode of BASELINE.json's north_star);
 *   1 : bf16 on the resolution levels the 16-bit kernel families cover (rows of >= 40 pixels in whole octets: 97.5 % of
 *       the activation elements at 120x160), fp32 on the deep levels.  Matches the reference's mixed-precision runs
 *       (Lightning --precision 16, train.py:100-101; BASELINE.json configs[1], [3]).  Operands are plain bf16 (forces
 *       rln_set_dense_arith(1, bf16, 1, bf16)); BatchNorm statistics, accumulation, the gradient stacks, parameters,
 *       the head's probabilities and every reduction stay fp32; values are rounded to nearest-even once, when stored.
 * Call before rln_workspace_bytes.  Input / output tensors of the ABI stay fp32. */
int rln_set_storage(rln_ctx* ctx, int mode);
int rln_get_storage(const rln_ctx* ctx);

int rln_bind_params(rln_ctx* ctx, float* params, float* grads, float* bn_running, int64_t* num_batches_tracked);

/* rln_set_eval_cache: frozen-model loops (makeDemoVideo.py:15-47, test.py:80-94: model.eval(), then one forward per
 * frame).  An eval forward (training = 0) normally rebuilds the MFMA weight fragments and the BatchNorm tables folded
 * from the running statistics on every call, because the optimiser may have changed the arena in between (66 small
 * launches, 0.36 ms of a 4.7 ms 480x640 frame).  With enable = 1 the next eval forward builds them and the following
 * ones reuse them until something invalidates them: a training forward, rln_bind_params, rln_set_workspace,
 * rln_set_dense_arith, rln_set_storage or another rln_set_eval_cache call.  The caller promises not to write the
 * parameter arena or the running statistics in between (rln_adamw_step / rln_sgd_step take raw pointers: call
 * rln_set_eval_cache again after them).  Results are bit-identical to the uncached forward.  Default: off. */
int rln_set_eval_cache(rln_ctx* ctx, int enable);

/* ---- workspace -------------------------------------------------------------------------
 * Activation stacks, gradient stacks, statistics and scratch for a given input geometry.
 * The host allocates (torch caching allocator) and hands the block over.  rln_set_workspace is a set-up call (once per
 * geometry): it drains the device (hipDeviceSynchronize) before it writes its descriptor tables into the block, because
 * the block usually reuses the memory of the previous workspace while that geometry's kernels may still be in flight. */
size_t rln_workspace_bytes(const rln_ctx* ctx, int n, int h, int w, int with_backward);
int rln_set_workspace(rln_ctx* ctx, void* ws, size_t bytes, int n, int h, int w, int with_backward);

/* ---- forward: TrainingBase.forward (TrainingBase.py:54-57) = featureExtractor (tiramisu.py:89-106)
 * + classifier (tiramisu.py:120-125).
 *   training     : 1 = BatchNorm batch statistics + running-stat update + Dropout2d, 0 = eval
 *   drop_scales  : NULL -> masks drawn on the device from `seed`; else [sum_c N*C_call] floats holding
 *                  0 or 1/(1-p) per (call, sample, channel) (parity tests inject the oracle's masks)
 *   probs_out    : [N, n_classes, H, W] softmax probabilities (or scaled logits if use_softmax==0); may be NULL
 *   feat_out     : [N, feature_channels, H, W] L2-normalised features (F.normalize, tiramisu.py:105); may be NULL
 */
int rln_forward(rln_ctx* ctx, const float* x, int n, int h, int w, int training, const float* drop_scales,
                uint64_t seed, float* probs_out, float* feat_out, int use_softmax, void* stream);

/* Classifier alone on caller-provided features (FCDenseNetClassifier.forward, tiramisu.py:120-125). */
int rln_classifier_forward(rln_ctx* ctx, const float* feat, int n, int h, int w, float* out, int use_softmax,
                           void* stream);

/* ---- loss: SimpleTrainModule.training_step (SimpleTrain.py:15-20) / evaluate_batch (TrainingBase.py:84-87)
 *   weighted=1: class weights = 1/count per batch (getClassWeight, TrainingBase.py:12-23), cross_entropy on the
 *   probabilities (double softmax); weighted=0: plain mean CE.
 *   out[0]=loss, out[1]=accuracy in [0,1], out[2]=number of labels >= n_classes (reference asserts on those),
 *   out[3..3+n_classes) = class pixel counts.  argmax_out: optional int64 [N,H,W] (torch.max(outputs,1)[1]).
 *   confusion_out: optional int64 [n_classes*n_classes], row = label, col = prediction. */
int rln_loss(rln_ctx* ctx, const float* probs, const int64_t* y, int n, int h, int w, int weighted, float* out,
             int64_t* argmax_out, int64_t* confusion_out, void* stream);

/* ---- MME unlabelled branch (MMETrainingModule.py:10-11,28-33): out[0] = lamda * mean_pixels(sum_k p*log(p+1e-5)).
 * A following rln_backward differentiates THIS loss and applies the gradient reversal of
 * GradReverse (tiramisu.py:7-18) between classifier and feature extractor: classifier gradients keep their sign,
 * everything upstream of the features is negated. */
int rln_entropy_loss(rln_ctx* ctx, const float* probs, int n, int h, int w, float lamda, float* out, void* stream);

/* ---- differentiable module forward (TrainingBase.forward in user-written training code, SimpleTrain.py:15,
 * MMETrainingModule.py:34-35): instead of one of the fused losses above, the caller hands in d(loss)/d(probabilities)
 * [N][n_classes][H][W] for the probabilities the last TRAINING rln_forward returned; a following rln_backward
 * differentiates through softmax, /T, classifier, F.normalize and the feature extractor.  The buffer must stay alive
 * until rln_backward has run. */
int rln_set_output_grad(rln_ctx* ctx, const float* dprobs, int n, int h, int w);

/* torch.optim.SGD(momentum, nesterov=True, dampening=0, weight_decay) on a flat range, as configured in
 * MMETrainingModule.py:17-20 (one call per parameter group; first_step=1 initialises the momentum buffer). */
int rln_sgd_step(float* params, const float* grads, float* momentum_buf, int64_t count, float lr, float momentum,
                 float weight_decay, int first_step, float grad_scale, void* stream);

/* ---- backward of loss∘classifier∘featureExtractor for the last training rln_forward + rln_loss(weighted)
 * (autograd of the torch graph in the reference).  Fills the bound gradient arena (overwrites).
 * The plan is cut into rln_backward_segments() segments that complete contiguous slices of the gradient
 * arena back to front; [seg_begin, seg_end) lets the host interleave gradient all-reduce with backward.
 * grad range completed by a segment: rln_backward_segment_range. loss_scale multiplies d(loss). */
int rln_backward_segments(const rln_ctx* ctx);
int rln_backward_segment_range(const rln_ctx* ctx, int seg, int64_t* grad_begin, int64_t* grad_end);
int rln_backward(rln_ctx* ctx, float loss_scale, int seg_begin, int seg_end, void* stream);
/* The same with d(loss) additionally multiplied by a DEVICE scalar (may be NULL): the incoming gradient of
 * loss.backward() (trainingModules/SimpleTrain.py:25 returns the loss to Lightning, which calls backward on it) never
 * has to visit the host.  The effective scale is loss_scale * (*loss_scale_dev). */
int rln_backward_scaled(rln_ctx* ctx, float loss_scale, const float* loss_scale_dev, int seg_begin, int seg_end,
                        void* stream);
/* Re-points the gradient arena only (same layout as in rln_bind_params): the module path hands every backward a fresh
 * flat buffer that autograd then owns, instead of copying out of a fixed arena. */
int rln_bind_grads(rln_ctx* ctx, float* grads);

/* ---- optimiser: torch.optim.AdamW single group as configured in SimpleTrain.py:27-28, on flat arenas. */
int rln_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t count, float lr,
                   float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream);

/* ---- single-op entry points (parity tests exercise each kernel family alone) --------------
 * rln_op_conv_bnrelu: y = scale[n,o] * (conv_KxK(relu(a[c]*x+b[c])) + bias[o]) written at channel offset
 *   out_coff of a [N, out_ctot, H, W] tensor (layers.py:8-12 with BN folded to a,b; a==NULL -> raw input).
 *   pool=1 appends MaxPool2d(2) (layers.py:46-52) and writes [N, out_ctot, H/2, W/2] + uint8 argmax idx.
 *   stats: optional [cout,2] per-channel (sum, sum of squares) of what was written. */
int rln_op_conv_bnrelu(const float* x, int n, int cin, int x_ctot, int x_coff, int h, int w, const float* a,
                       const float* b, const float* weight, const float* bias, int cout, int ksize,
                       const float* scale, float* out, int out_ctot, int out_coff, int pool, uint8_t* pool_idx,
                       float* stats, void* workspace, size_t workspace_bytes, void* stream);
/* rln_op_dense3_fwd: the same dense-layer forward (ksize 3, cout <= 16, folded affine required) on the 16-bit MFMA pipe
 * with split fp32 operands (csrc/dense3.h): parts = 1..3 operand parts, dtype 0 = bf16, 1 = f16.  workspace holds the
 * packed weight fragments and the statistics partials.  Returns RLN_ERR_UNSUPPORTED for geometries the kernel does
 * not cover (W % 4 != 0, W < 40, unaligned views): callers fall back to rln_op_conv_bnrelu (exact fp32 MFMA). */
int rln_op_dense3_fwd(const float* x, int n, int cin, int x_ctot, int x_coff, int h, int w, const float* a,
                      const float* b, const float* weight, const float* bias, int cout, const float* scale, float* out,
                      int out_ctot, int out_coff, float* stats, int parts, int dtype, void* workspace,
                      size_t workspace_bytes, void* stream);
/* rln_op_dense3_fwd_pair: TWO consecutive dense layers of a block in one pass over the input channels they share
 * (csrc/dense3.h: d3_fwd_pair_launch; parts = 1 -- parts = 2 returns RLN_ERR_UNSUPPORTED: the two-part pair kernel is
 * written and parity-green but not instantiated, it does not fit the register budget).  stack [N, ctot, H, W]: layer 1 reads channels [coff, coff + cin)
 * (cin % 16 == 0) and writes [coff + cin, +16); layer 2 reads [coff, coff + cin + 16) and writes the 16 channels after
 * them.  (a1, b1) / (a2, b2): folded affines over cin / cin + 16 channels; w1 [16, cin, 3, 3], w2 [16, cin + 16, 3, 3];
 * scale1 / scale2 [N, 16] optional; stats1 / stats2 optional [16, 2]; scratch: N * 16 * H * W floats (layer 2's raw sums
 * over the shared channels).  Results equal two rln_op_dense3_fwd calls with the same parts up to fp32 summation order. */
int rln_op_dense3_fwd_pair(float* stack, int n, int cin, int ctot, int coff, int h, int w, const float* a1,
                           const float* b1, const float* w1, const float* bias1, const float* scale1, const float* a2,
                           const float* b2, const float* w2, const float* bias2, const float* scale2, float* stats1,
                           float* stats2, int parts, int dtype, float* scratch, void* workspace,
                           size_t workspace_bytes, void* stream);
/* rln_op_td_fwd: the TransitionDown forward (layers.py:45-58: BN -> ReLU -> Conv2d 1x1 -> Dropout2d scale -> MaxPool2d(2))
 * on the 16-bit MFMA pipe with split fp32 operands (csrc/pw1.h); arguments as rln_op_conv_bnrelu with ksize 1, pool 1,
 * plus parts / dtype as rln_op_dense3_fwd.  Needs an even W and 8-byte aligned views. */
int rln_op_td_fwd(const float* x, int n, int cin, int x_ctot, int x_coff, int h, int w, const float* a, const float* b,
                  const float* weight, const float* bias, int cout, const float* scale, float* out, int out_ctot,
                  int out_coff, uint8_t* pool_idx, float* stats, int parts, int dtype, void* workspace,
                  size_t workspace_bytes, void* stream);
/* rln_op_td_bwd: backward of that transition on the split-operand kernels.  dyp [N, cout, h/2, w/2] is the gradient of
 * the pooled output (Dropout2d scale already applied), pool_idx the forward's argmax bytes; x [N, cin, h, w] contiguous.
 * g (optional): channels in [acc_lo, acc_hi) accumulate gamma[c] * gz, the others are overwritten, gz = relu'(a x + b) *
 * (W^T unpool(dyp)); stats (optional) [cin, 2] = sum gz, sum gz * (x - mean) * invstd.  dw (optional) [cout, cin]. */
int rln_op_td_bwd(const float* x, const float* dyp, const uint8_t* pool_idx, const float* weight, int n, int cin, int cout,
                  int h, int w, const float* a, const float* b, const float* gamma, const float* mean, const float* invstd,
                  int acc_lo, int acc_hi, float* g, float* stats, float* dw, int parts, int dtype, void* workspace,
                  size_t workspace_bytes, void* stream);
/* rln_op_fc_fwd / rln_op_fc_wgrad: the first convolution (Conv2d(cin <= 3, cout <= 64, 3, padding 1) on the raw input,
 * tiramisu.py:33-35) and its weight gradient on the split-operand kernels (csrc/fc3.h).  x [N, cin, h, w] and
 * dy [N, cout, h, w] contiguous, dw [cout, cin, 3, 3]; w % 4 == 0 (forward), w % 8 == 0 (weight gradient). */
int rln_op_fc_fwd(const float* x, int n, int cin, int h, int w, const float* weight, const float* bias, int cout,
                  float* out, int out_ctot, int out_coff, float* stats, int parts, int dtype, void* workspace,
                  size_t workspace_bytes, void* stream);
int rln_op_fc_wgrad(const float* x, const float* dy, int n, int cin, int cout, int h, int w, float* dw, int parts, int dtype,
                    void* workspace, size_t workspace_bytes, void* stream);
/* rln_op_tu_fwd: the TransitionUp forward (ConvTranspose2d k3 s2 + bias, top-left crop to (hout, wout)) on the 16-bit
 * MFMA pipe with split fp32 operands (csrc/ct3.h); x is a channel range [x_coff, x_coff + cin) of a [N, x_ctot, H, W]
 * tensor, stats optional [cout, 2] sums of what was written; parts / dtype as rln_op_dense3_fwd. */
int rln_op_tu_fwd(const float* x, int n, int cin, int x_ctot, int x_coff, int h, int w, const float* weight,
                  const float* bias, int cout, float* out, int out_ctot, int out_coff, int hout, int wout, float* stats,
                  int parts, int dtype, void* workspace, size_t workspace_bytes, void* stream);
/* rln_op_tu_bwd: backward of the same transposed convolution on the split-operand kernels: dx[N, cin, h, w] =
 * cscale[c] * d(out)/d(x) . du (overwritten; cscale may be NULL) and dw[cin, cout, 3, 3]; either output may be NULL.
 * x and du are contiguous.  The weight-gradient kernel needs w % 8 == 0, both need wout == 2 w. */
int rln_op_tu_bwd(const float* x, const float* du, const float* weight, int n, int cin, int cout, int h, int w, int hout,
                  int wout, const float* cscale, float* dx, float* dw, int parts, int dtype, void* workspace,
                  size_t workspace_bytes, void* stream);
/* rln_op_convt: ConvTranspose2d(k3,s2,p0)+bias cropped top-left to (hout,wout) (layers.py:58-67,82-86). */
int rln_op_convt(const float* x, int n, int cin, int h, int w, const float* weight, const float* bias, int cout,
                 float* out, int out_ctot, int out_coff, int hout, int wout, void* stream);

/* ---- in-library timing: HIP events recorded on the launch stream around every kernel class --------------
 * rln_profile_enable(ctx,1) clears and starts collecting, (ctx,0) stops.  rln_profile_read waits for the
 * recorded events and returns, per kernel class, the summed device time (ms), the ALGORITHMIC flops and bytes
 * of the launches (2*Cin*Cout*taps*H*W*N; each operand read once / result written once) and the launch count. */
int rln_profile_enable(rln_ctx* ctx, int on);
int rln_profile_num_classes(void);
const char* rln_profile_class_name(int cls);
int rln_profile_read(rln_ctx* ctx, double* ms, double* flops, double* bytes, int64_t* launches);
/* the same collection launch by launch (in issue order): class id, device time, algorithmic flops / bytes of up to `cap`
 * entries; returns the number of entries recorded (may exceed cap), < 0 on error */
int64_t rln_profile_entries(rln_ctx* ctx, int* cls, double* ms, double* flops, double* bytes, int64_t cap);

/* diagnostic only: in-kernel cycle stamps of the dense forward kernel (enabled by env RLN_DBG=16) */
int rln_debug_read_stamps(unsigned long long* out8);

/* ---- EncDecNet (models/EncDecNet.py:5-112, legacy secondary model) building blocks ---------------------------
 * rln_op_conv_act: Conv2d(k in {1,3,7}, padding k/2) + bias + activation (EncDecNet.py:14,31-32);
 *   act: 0 none, 1 relu, 2 prelu / leaky-relu with slope act_param, 3 sigmoid, 4 tanh;
 *   stats (optional) = per-channel [sum, sum of squares] of the activated output (BatchNorm batch statistics).
 * rln_op_bn_affine: BatchNorm2d folded to y = a*x+b from those sums (train; updates running stats) or from the
 *   running statistics (eval) (EncDecNet.py:19-22,33).
 * rln_op_bn_drop_maxpool: a*x+b, elementwise Dropout mask (0 or 1/(1-p), may be NULL), MaxPool2d(k, 2, k/2)
 *   (EncDecNet.py:24-27,68,103-104).  rln_op_bn_drop_upsample2: same prologue, UpsamplingBilinear2d(x2),
 *   align_corners=True (EncDecNet.py:69,107-108).  rln_op_softmax_channels: Softmax(dim=-3) (EncDecNet.py:97). */
int rln_op_conv_act(const float* x, int n, int cin, int h, int w, const float* weight, const float* bias, int cout,
                    int ksize, int act, float act_param, float* out, float* stats, void* workspace,
                    size_t workspace_bytes, void* stream);
int rln_op_bn_affine(const float* sums, int c, double count, int training, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, float momentum, float eps, float* a, float* b,
                     void* stream);
int rln_op_bn_drop_maxpool(const float* x, int n, int c, int h, int w, const float* a, const float* b,
                           const float* mask, int k, float* out, void* stream);
int rln_op_bn_drop_upsample2(const float* x, int n, int c, int h, int w, const float* a, const float* b,
                             const float* mask, float* out, void* stream);
int rln_op_softmax_channels(const float* x, int n, int c, int hw, float* out, void* stream);
int rln_op_dropout_mask(float* dst, int64_t count, float keep, uint64_t seed, void* stream);
/* rln_op_scaled_softmax: the tail of FCDenseNetClassifier.forward (tiramisu.py:120-125) for classifiers whose finalConv is
 * not 1x1 (FCDenseNet57(n_classes, kernel_size), tiramisu.py:113-115,150): out = softmax over channels of x / T
 * (use_softmax) or x / T, on the [n][c][hw] logits rln_op_conv_act produced. */
int rln_op_scaled_softmax(const float* x, int n, int c, int hw, float T, int use_softmax, float* out, void* stream);

/* rln_preprocess_u8: the non-augmenting input transform of dataManagement/myTransforms.py:15-19 on device:
 * Resize(h, w) [bilinear, cv2 INTER_LINEAR fixed point for 8-bit] -> optional ToGray -> Normalize(mean, std, 255) ->
 * CHW float32; labels (may be null together with y) are resized with nearest neighbour (cv2 INTER_NEAREST) to int64.
 * frames: [n][hs][ws][3] uint8 in the stored channel order (the reference normalises BGR frames with the RGB
 * constants, myDatasets.py:51); mean3/std3: HOST pointers to 3 floats; x: [n][3][h][w]; y: [n][h][w].
 * Parity of this row is unpinned (cv2/albumentations are not available to the reference here). */
int rln_preprocess_u8(const uint8_t* frames, int n, int hs, int ws, const uint8_t* labels, int h, int w, int gray,
                      const float* mean3, const float* std3, float* x, int64_t* y, void* stream);

/* rln_augment_u8: the augmenting branch of the same transform (myTransforms.py:8-13): HueSaturationValue ->
 * RandomSizedCrop resized to h x w -> OneOf(MotionBlur, GaussNoise) -> Normalize -> CHW.  The caller draws the
 * per-image random parameters and passes them as a DEVICE table params[n][80]:
 *   0..2 hue/sat/val shift | 3..6 crop y, x, height, width (inside the frame; validated by the caller) |
 *   7 choice (0 blur, 1 noise) | 8 blur kernel size | 9 noise sigma | 10 noise seed (integer < 2^24) |
 *   16..64 the 7x7 blur kernel, row-major, centred.
 * scratch: device [n][h][w][3] bytes.  mean3/std3: HOST pointers.  Parity with albumentations/cv2: unpinned. */
int rln_augment_u8(const uint8_t* frames, int n, int hs, int ws, const uint8_t* labels, int h, int w,
                   const float* params, const float* mean3, const float* std3, uint8_t* scratch, float* x, int64_t* y,
                   void* stream);

/* rln_overlay_u8: the per-frame tail of makeDemoVideo.py:36-46 on device.  pred = argmax over classes of probs
 * [n][ncls][h][w] (first maximum wins, torch.max(out, 1)); out [n][h][w][3] uint8 = the frame resized to (h, w) with
 * cv2's 8-bit INTER_LINEAR arithmetic (the script's `cv2.resize(frame, framesize, cv2.INTER_LANCZOS4)` passes the flag
 * in the dst position, so the default interpolation applies), with out[pred == k] = colors_host[k] for every class k
 * whose bit is set in paint_mask (the script paints 1 -> (0,255,0), 2 -> (255,0,0), 3 -> (0,0,255), BGR).
 * colors_host: HOST pointer to ncls*3 bytes; pred_out: optional [n][h][w] uint8 class map (pred.byte()).
 * Resize parity with cv2 is unpinned (cv2 absent); argmax and painting are exact. */
int rln_overlay_u8(const uint8_t* frames, int n, int hs, int ws, const float* probs, int ncls, int h, int w,
                   const uint8_t* colors_host, unsigned paint_mask, uint8_t* out, uint8_t* pred_out, void* stream);

/* rln_op_classifier: FCDenseNetClassifier.forward on caller-provided weights (tiramisu.py:120-125):
 * out[n,k,p] = softmax_k((sum_c w[k,c]*feat[n,c,p] + b[k]) / T). */
int rln_op_classifier(const float* feat, int n, int c, int hw, const float* w, const float* b, int ncls, float T,
                      float* out, int use_softmax, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RLN_H_ */
