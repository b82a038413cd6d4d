// Execution plan + C ABI of the lane-segmentation hot path (see include/rln.h).
//
// Memory design (MI355X-first, not a translation of the torch graph):
//   * one channel-stack buffer per resolution level, laid out [convT out | skip stack | up-block new];
//     the down block works in place on the middle range, the up block on the whole buffer, so the
//     reference's torch.cat calls (layers.py:33,39,68) never copy anything;
//   * per-channel batch statistics are produced once by the kernel that writes a channel and shared by
//     every BatchNorm that later normalises it (they differ only in gamma/beta);
//   * backward keeps ONE raw gradient stack per level: each BatchNorm consumer adds gamma*gy into it and
//     its (sum gy, sum gy*xhat) into per-channel accumulators; the mean-subtraction part of the
//     BatchNorm gradient is applied lazily when the channel's producer runs its own backward.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rln.h"
#include "dense3.h"
#include "pw1.h"
#include "ct3.h"
#include "fc3.h"
#include "igemm.h"
#include "pointwise.h"
#include "storage.h"

using namespace rln;

static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
#define RLN_TRY(expr)                                                                   \
  do {                                                                                  \
    int _e = (expr);                                                                    \
    if (_e != 0) {                                                                      \
      char _inner[256];                                                                 \
      snprintf(_inner, sizeof(_inner), "%s", g_err);                                    \
      g_err[0] = 0;                                                                     \
      return fail(_e, "%s failed with %d (%s:%d)%s%s", #expr, _e, __FILE__, __LINE__,   \
                  _inner[0] ? " <- " : "", _inner);                                     \
    }                                                                                   \
  } while (0)

namespace {

struct TensorInfo {
  std::string name;
  int kind;
  int64_t offset;
  int ndim;
  int64_t shape[4];
};

struct BNRef {
  int C = 0;
  int64_t gamma = -1, beta = -1;  // parameter arena
  int64_t rmean = -1, rvar = -1;  // running-stat arena
  int64_t ab = -1;                // folded affine arena (a at ab, b at ab + ab_total)
};
struct ConvRef {
  int64_t w = -1, b = -1;
  int cin = 0, cout = 0, ks = 0;
};
enum OpType { OP_FIRST = 0, OP_DENSE = 1, OP_TD = 2, OP_TU = 3 };
struct Op {
  OpType type;
  int src_level = 0, dst_level = 0;
  int in_off = 0, cin = 0;
  int out_off = 0, cout = 0;
  BNRef bn;
  ConvRef conv;
  int64_t drop_ch = -1;  // cumulative channel offset of this op's Dropout2d call (-1: none)
  int seg = 0;
  int acc_lo = 0, acc_hi = 0;  // backward: logical input channels that already hold gradient
  int64_t grad_begin = 0, grad_end = 0;
};
struct Level {
  int C = 0, H = 0, W = 0;
  int64_t stat_off = 0;  // into per-channel arrays
  float* S = nullptr;    // activation stack; holds bf16 elements when st == ST_BF16 (storage.h): address it through sp()
  float* G = nullptr;    // gradient stack (fp32 in every mode)
  int st = 0;            // storage element type of S and of the finalised output gradients dY of this level
  // channel `ch` of sample 0 of the activation stack
  float* sp(long long ch) const { return st_at(S, ch * (long long)H * W, st); }
};

}  // namespace

enum ProfClass {
  PC_DENSE_FWD = 0, PC_FIRST_FWD, PC_TD_FWD, PC_TU_FWD, PC_DENSE_DGRAD, PC_TD_DGRAD, PC_TU_DGRAD, PC_DENSE_WGRAD,
  PC_FIRST_WGRAD, PC_TD_WGRAD, PC_TU_WGRAD, PC_BN, PC_GRADFIN, PC_REDUCE, PC_HEAD_FWD, PC_LOSS, PC_HEAD_BWD,
  PC_D3_FWD, PC_D3_PULL, PC_D3_WGRAD, PC_D3_FWD_S, PC_D3_FWD2, PC_D3_FIN, PC_COUNT
};
static const char* kProfNames[PC_COUNT] = {
    "dense_conv3x3_fwd", "first_conv_fwd", "transition_down_fwd", "transition_up_fwd", "dense_conv3x3_dgrad",
    "transition_down_dgrad", "transition_up_dgrad", "dense_conv3x3_wgrad", "first_conv_wgrad",
    "transition_down_wgrad", "transition_up_wgrad", "bn_stats_affine", "grad_finalize", "partial_reduce",
    "head_fwd", "loss", "head_bwd", "dense3_fwd", "dense3_dgrad_pull", "dense3_wgrad", "dense3_fwd_small", "dense3_fwd_pair", "dense3_fwd_finish"};
struct ProfEntry {
  hipEvent_t a, b;
  int cls;
  double flops, bytes;
};
struct Profiler {
  bool on = false;
  std::vector<ProfEntry> entries;
  std::vector<hipEvent_t> pool;
  hipEvent_t get() {
    if (!pool.empty()) {
      hipEvent_t e = pool.back();
      pool.pop_back();
      return e;
    }
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
  }
};

struct rln_ctx {
  Profiler prof;
  rln_config cfg;
  std::vector<TensorInfo> tensors;
  std::vector<Op> ops;
  std::vector<Level> levels;  // n_down resolution levels + bottleneck
  int64_t n_param = 0, n_bnstat = 0, n_nbt = 0, n_ab = 0, n_chan = 0;
  int feat_C = 0;
  std::vector<int> drop_per_call;
  int64_t drop_total = 0;
  ConvRef cls;
  int n_seg = 0;
  std::vector<int64_t> seg_begin, seg_end;
  // bound arenas
  float* params = nullptr;
  float* grads = nullptr;
  float* bnrun = nullptr;
  int64_t* nbt = nullptr;
  // workspace
  int N = 0, H = 0, W = 0, with_bwd = 0;
  float *mean = nullptr, *var = nullptr, *invstd = nullptr, *stdv = nullptr, *S1 = nullptr, *S2 = nullptr;
  float* ab = nullptr;
  float* masks = nullptr;
  float* stat_partial = nullptr;
  float* dY = nullptr;
  float* fsplit = nullptr;  // forward split-K partial sums (small levels)
  float* wpartial = nullptr;
  float* bpartial = nullptr;
  float* glin = nullptr;
  unsigned char* pool_idx = nullptr;
  std::vector<int64_t> pool_off;  // per TD op index in ops
  LossScratch loss;
  // backward concurrency: weight gradients run on a side stream next to the data-gradient chain
  hipStream_t side = nullptr;
  hipEvent_t ev_dy[2] = {nullptr, nullptr};  // dY[buf] written (main stream)
  hipEvent_t ev_wg[2] = {nullptr, nullptr};  // last weight-gradient reading dY[buf] finished (side stream)
  bool wg_pending[2] = {false, false};
  float* dYbuf[2] = {nullptr, nullptr};
  void* dy16 = nullptr;  // bf16 copies of dense layers' finalised output gradients (the weight gradient's one-part operand):
  size_t dy16_half = 0;  // two buffers of this many bytes, by layer parity (two-layer weight-gradient launches)
  float* wpartial2 = nullptr;  // slabs of the secondary layer of a two-layer launch
  float* fpair = nullptr;      // raw sums of the second layer of a paired dense forward (dense3.h: D3Fwd.partial_out)
  long long pair_finish = -1;  // op whose forward completes a paired launch (c_first / partial_in), or -1
  int dy_flip = 0;
  bool use_side = false;
  // forward/backward hand-over state
  const float* last_x = nullptr;
  const int64_t* last_y = nullptr;
  int have_train_fwd = 0, have_loss = 0;
  long long prep_done = -1;  // op whose bn_prep already ran with the previous op's bn_finalize
  int eval_cache_on = 0;     // rln_set_eval_cache
  int eval_tables_valid = 0; // packed forward weights + folded BN tables are those of the current arena (eval mode)
  int eval_reuse = 0;        // this forward reuses them
  int loss_mode = 0;      // 0: weighted CE (rln_loss), 1: entropy with gradient reversal (rln_entropy_loss)
  float loss_lamda = 0.f;
  const float* last_scales = nullptr;
  const float* gext = nullptr;  // loss_mode 2: caller-supplied gradient of the probabilities
  // dense-layer arithmetic (rln_set_dense_arith): 0 parts = exact fp32 MFMA kernels, else split 16-bit MFMA (dense3.h)
  int d3_fwd_np = 0, d3_fwd_dt = 0, d3_bwd_np = 0, d3_bwd_dt = 0;
  int wg_parts = 0;  // operand parts of the dense 3x3 weight-gradient GEMMs (rln_set_wgrad_parts)
  int storage = 0;   // rln_set_storage: 0 = fp32 stacks, 1 = bf16 stacks on the levels the 16-bit kernel families cover
  std::vector<D3PackDesc> d3_desc_f, d3_desc_b;  // host copies, one entry per dense op
  std::vector<long long> d3_wf_off, d3_wb_off;   // per op index (uint4 units into d3_packed), -1: none
  D3PackDesc* d3_desc_f_dev = nullptr;
  D3PackDesc* d3_desc_b_dev = nullptr;
  uint4* d3_packed = nullptr;
  int d3_units_f = 0, d3_units_b = 0;
  std::vector<P1PackDesc> p1_desc_f, p1_desc_b;  // TransitionDown 1x1 weights (same packed buffer)
  std::vector<long long> p1_wf_off, p1_wb_off;
  P1PackDesc* p1_desc_f_dev = nullptr;
  P1PackDesc* p1_desc_b_dev = nullptr;
  int p1_units_f = 0, p1_units_b = 0;
  std::vector<C3PackDesc> c3_desc_f, c3_desc_b;  // TransitionUp 3x3 transposed-conv weights (same packed buffer)
  std::vector<long long> c3_wf_off, c3_wb_off;
  C3PackDesc* c3_desc_f_dev = nullptr;
  C3PackDesc* c3_desc_b_dev = nullptr;
  int c3_units_f = 0, c3_units_b = 0;
  std::vector<float*> dyblk;  // one finalised-output-gradient buffer per layer of a dense block (pull-form backward)
};

namespace {

void add_tensor(rln_ctx* c, const std::string& name, int kind, int64_t off, std::initializer_list<int64_t> shp) {
  TensorInfo t;
  t.name = name;
  t.kind = kind;
  t.offset = off;
  t.ndim = (int)shp.size();
  int i = 0;
  for (auto v : shp) t.shape[i++] = v;
  for (; i < 4; ++i) t.shape[i] = 1;
  c->tensors.push_back(t);
}

BNRef add_bn(rln_ctx* c, const std::string& prefix, int C) {
  BNRef b;
  b.C = C;
  b.gamma = c->n_param;
  add_tensor(c, prefix + ".weight", RLN_T_PARAM, c->n_param, {C});
  c->n_param += C;
  b.beta = c->n_param;
  add_tensor(c, prefix + ".bias", RLN_T_PARAM, c->n_param, {C});
  c->n_param += C;
  b.rmean = c->n_bnstat;
  add_tensor(c, prefix + ".running_mean", RLN_T_RUNNING_MEAN, c->n_bnstat, {C});
  c->n_bnstat += C;
  b.rvar = c->n_bnstat;
  add_tensor(c, prefix + ".running_var", RLN_T_RUNNING_VAR, c->n_bnstat, {C});
  c->n_bnstat += C;
  c->tensors.push_back(TensorInfo{prefix + ".num_batches_tracked", RLN_T_NUM_BATCHES, c->n_nbt, 0, {1, 1, 1, 1}});
  c->n_nbt += 1;
  b.ab = c->n_ab;
  c->n_ab += C;
  return b;
}

ConvRef add_conv(rln_ctx* c, const std::string& prefix, int d0, int d1, int ks, int cin, int cout) {
  ConvRef r;
  r.cin = cin;
  r.cout = cout;
  r.ks = ks;
  r.w = c->n_param;
  add_tensor(c, prefix + ".weight", RLN_T_PARAM, c->n_param, {d0, d1, ks, ks});
  c->n_param += (int64_t)d0 * d1 * ks * ks;
  r.b = c->n_param;
  add_tensor(c, prefix + ".bias", RLN_T_PARAM, c->n_param, {cout});
  c->n_param += cout;
  return r;
}

std::string fmt(const char* f, int a, int b = 0) {
  char buf[160];
  snprintf(buf, sizeof(buf), f, a, b);
  return buf;
}

// Builds ops in forward execution order; the parameter arena follows the same order.
int build_plan(rln_ctx* c) {
  const rln_config& g = c->cfg;
  if (g.n_down < 1 || g.n_down > RLN_MAX_BLOCKS || g.n_up != g.n_down)
    return fail(RLN_ERR_ARG, "need 1 <= n_down == n_up <= %d", RLN_MAX_BLOCKS);
  if (g.growth_rate < 1 || g.first_conv_channels < 1 || g.in_channels < 1 || g.n_classes < 1 || g.n_classes > 16 ||
      g.bottleneck_layers < 1)
    return fail(RLN_ERR_ARG, "bad channel configuration");
  const int gr = g.growth_rate, nd = g.n_down;
  for (int i = 0; i < nd; ++i)
    if (g.down_blocks[i] < 1 || g.up_blocks[i] < 1) return fail(RLN_ERR_ARG, "blocks need >= 1 layer");
  // channel bookkeeping (tiramisu.py:26-87)
  std::vector<int> skip(nd), ct(nd), upnew(nd), down_in(nd);
  int cur = g.first_conv_channels;
  for (int i = 0; i < nd; ++i) {
    down_in[i] = cur;
    cur += gr * g.down_blocks[i];
    skip[i] = cur;
  }
  const int bott_in = cur, bott_new = gr * g.bottleneck_layers;
  int prev = bott_new;
  for (int i = 0; i < nd; ++i) {
    const int L = nd - 1 - i;
    ct[L] = prev;
    upnew[L] = gr * g.up_blocks[i];
    prev = upnew[L];
  }
  c->levels.resize(nd + 1);
  int64_t so = 0;
  for (int L = 0; L < nd; ++L) {
    c->levels[L].C = ct[L] + skip[L] + upnew[L];
    c->levels[L].stat_off = so;
    so += c->levels[L].C;
  }
  c->levels[nd].C = bott_in + bott_new;
  c->levels[nd].stat_off = so;
  so += c->levels[nd].C;
  c->n_chan = so;
  c->feat_C = c->levels[0].C;

  const std::string fe = "featureExtractor.";
  int seg_of_down_stage0 = 0;
  (void)seg_of_down_stage0;
  // segments in BACKWARD order: 0 = head, 1..nd = up stages (level 0 first), nd+1 = bottleneck,
  // nd+2 .. 2nd+1 = down stages (deepest first), 2nd+2 = first conv
  c->n_seg = 2 * nd + 3;
  auto seg_up = [&](int i /*up stage index, forward order*/) { return 1 + (nd - 1 - i); };
  auto seg_down = [&](int i) { return nd + 2 + (nd - 1 - i); };
  int64_t drop_ch = 0;

  {  // first conv (tiramisu.py:33-35)
    Op o;
    o.type = OP_FIRST;
    o.src_level = -1;
    o.dst_level = 0;
    o.cin = g.in_channels;
    o.out_off = ct[0];
    o.cout = g.first_conv_channels;
    o.grad_begin = c->n_param;
    o.conv = add_conv(c, fe + "firstconv", o.cout, o.cin, 3, o.cin, o.cout);
    o.grad_end = c->n_param;
    o.seg = 2 * nd + 2;
    c->ops.push_back(o);
  }
  auto add_dense = [&](const std::string& prefix, int level, int in_off, int cin, int seg) {
    Op o;
    o.type = OP_DENSE;
    o.src_level = o.dst_level = level;
    o.in_off = in_off;
    o.cin = cin;
    o.out_off = in_off + cin;
    o.cout = gr;
    o.grad_begin = c->n_param;
    o.bn = add_bn(c, prefix + ".norm", cin);
    o.conv = add_conv(c, prefix + ".conv", gr, cin, 3, cin, gr);
    o.grad_end = c->n_param;
    o.drop_ch = drop_ch;
    drop_ch += gr;
    c->drop_per_call.push_back(gr);
    o.seg = seg;
    c->ops.push_back(o);
  };
  for (int i = 0; i < nd; ++i) {  // down path (tiramisu.py:41-49)
    for (int j = 0; j < g.down_blocks[i]; ++j)
      add_dense(fe + fmt("denseBlocksDown.%d.layers.%d", i, j), i, ct[i], down_in[i] + j * gr, seg_down(i));
    Op o;
    o.type = OP_TD;
    o.src_level = i;
    o.dst_level = i + 1;
    o.in_off = ct[i];
    o.cin = skip[i];
    o.out_off = (i + 1 < nd) ? ct[i + 1] : 0;
    o.cout = skip[i];
    o.grad_begin = c->n_param;
    o.bn = add_bn(c, fe + fmt("transDownBlocks.%d.norm", i), skip[i]);
    o.conv = add_conv(c, fe + fmt("transDownBlocks.%d.conv", i), skip[i], skip[i], 1, skip[i], skip[i]);
    o.grad_end = c->n_param;
    o.drop_ch = drop_ch;
    drop_ch += skip[i];
    c->drop_per_call.push_back(skip[i]);
    o.seg = seg_down(i);
    c->ops.push_back(o);
  }
  for (int j = 0; j < g.bottleneck_layers; ++j)  // bottleneck (tiramisu.py:55-58)
    add_dense(fe + fmt("bottleneck.bottleneck.layers.%d", j), nd, 0, bott_in + j * gr, nd + 1);
  for (int i = 0; i < nd; ++i) {  // up path (tiramisu.py:64-85)
    const int L = nd - 1 - i;
    Op o;
    o.type = OP_TU;
    o.src_level = L + 1;
    o.dst_level = L;
    o.cin = ct[L];
    o.cout = ct[L];
    if (L + 1 == nd) {
      o.in_off = bott_in;
    } else {
      o.in_off = ct[L + 1] + skip[L + 1];
    }
    o.out_off = 0;
    o.grad_begin = c->n_param;
    o.conv = add_conv(c, fe + fmt("transUpBlocks.%d.convTrans", i), ct[L], ct[L], 3, ct[L], ct[L]);
    o.grad_end = c->n_param;
    o.seg = seg_up(i);
    c->ops.push_back(o);
    for (int j = 0; j < g.up_blocks[i]; ++j)
      add_dense(fe + fmt("denseBlocksUp.%d.layers.%d", i, j), L, 0, ct[L] + skip[L] + j * gr, seg_up(i));
  }
  c->drop_total = drop_ch;
  {  // classifier (tiramisu.py:112-118)
    c->cls.cin = c->feat_C;
    c->cls.cout = g.n_classes;
    c->cls.ks = 1;
    c->cls.w = c->n_param;
    add_tensor(c, "classifier.finalConv.weight", RLN_T_PARAM, c->n_param, {g.n_classes, c->feat_C, 1, 1});
    c->n_param += (int64_t)g.n_classes * c->feat_C;
    c->cls.b = c->n_param;
    add_tensor(c, "classifier.finalConv.bias", RLN_T_PARAM, c->n_param, {g.n_classes});
    c->n_param += g.n_classes;
  }
  // segment gradient ranges (contiguous by construction: arena order == execution order)
  c->seg_begin.assign(c->n_seg, INT64_MAX);
  c->seg_end.assign(c->n_seg, -1);
  for (const Op& o : c->ops) {
    if (o.grad_begin < c->seg_begin[o.seg]) c->seg_begin[o.seg] = o.grad_begin;
    if (o.grad_end > c->seg_end[o.seg]) c->seg_end[o.seg] = o.grad_end;
  }
  c->seg_begin[0] = c->cls.w;
  c->seg_end[0] = c->n_param;

  // backward accumulate-vs-overwrite ranges: simulate which gradient channels are initialised
  std::vector<std::vector<char>> init(nd + 1);
  for (int L = 0; L <= nd; ++L) init[L].assign(c->levels[L].C, 0);
  std::fill(init[0].begin(), init[0].end(), 1);  // head backward writes every level-0 channel
  for (int k = (int)c->ops.size() - 1; k >= 0; --k) {
    Op& o = c->ops[k];
    if (o.type == OP_TU) {
      for (int ch = 0; ch < o.cin; ++ch) init[o.src_level][o.in_off + ch] = 1;
    } else if (o.type == OP_DENSE || o.type == OP_TD) {
      std::vector<char>& v = init[o.src_level];
      int lo = -1, hi = -1;
      bool contiguous = true;
      for (int ch = 0; ch < o.cin; ++ch) {
        if (v[o.in_off + ch]) {
          if (lo < 0) lo = ch;
          if (hi >= 0 && ch != hi) contiguous = false;
          hi = ch + 1;
        }
      }
      if (!contiguous) return fail(RLN_ERR_UNSUPPORTED, "non-contiguous gradient initialisation in op %d", k);
      o.acc_lo = lo < 0 ? 0 : lo;
      o.acc_hi = lo < 0 ? 0 : hi;
      for (int ch = 0; ch < o.cin; ++ch) v[o.in_off + ch] = 1;
    }
  }
  return 0;
}

struct Carver {
  char* base;
  size_t off = 0;
  explicit Carver(void* b) : base((char*)b) {}
  template <typename T>
  T* take(size_t count) {
    off = (off + 255) & ~(size_t)255;
    T* p = base ? (T*)(base + off) : nullptr;
    off += count * sizeof(T);
    return p;
  }
};

void level_dims(const rln_ctx* c, int h, int w, std::vector<int>& hs, std::vector<int>& ws) {
  const int nd = c->cfg.n_down;
  hs.resize(nd + 1);
  ws.resize(nd + 1);
  hs[0] = h;
  ws[0] = w;
  for (int L = 1; L <= nd; ++L) {
    hs[L] = hs[L - 1] / 2;
    ws[L] = ws[L - 1] / 2;
  }
}

// Forward split-K factor of a dense layer.  Small levels do not fill the chip with one block per tile, so the
// input-channel loop is split over blocks.  The factor depends on the level's pixel count ONLY (not on the batch
// size), so a sample's result does not depend on which batch it is evaluated in.
int dense_fwd_split(int hw, int cin) {
  static const int mid = rln_env("RLN_SPLIT_MID") ? atoi(rln_env("RLN_SPLIT_MID")) : 2;  // 512 < hw <= 1200 (measured: 2)
  if (hw > 1200 || hw <= 0 || rln_env("RLN_NO_SPLITK")) return 1;
  if (hw > 512 && mid <= 1) return 1;
  const int nchunk = (cin + 15) / 16;
  const int want = hw <= 128 ? 8 : (hw <= 512 ? 4 : mid);
  return std::max(1, std::min(want, nchunk));
}

// Sample packing of the levels much smaller than a tile (igemm.h): rewrites the grid of a dense forward / looped data
// gradient launch to virtual images of pk samples; returns the number of virtual samples (N when not packed).
int pack_small_level(IgemmParams& p, int H, int W, int N, int* tile) {
  const int pk = igemm_pack_sx(H, W);
  if (pk <= 1 || rln_env("RLN_NO_PACK")) return N;
  if ((long long)(pk - 1) * std::max(p.in_ns, p.out_ns) >= 0x7fffffffLL) return N;
  p.pk_sx = pk;
  p.pk_w = W;
  p.pk_n = N;
  p.GH = p.Hin = H;
  p.GW = p.Win = pk * (W + 1) - 1;
  *tile = H <= 4 ? 5 : 0;
  p.tiles_x = p.tiles_y = 1;
  p.out_vec = 0;
  return (N + pk - 1) / pk;
}
// split-K factor of a packed dense forward: fewer, fuller tiles -> more input-channel ranges (a function of the level
// only, like dense_fwd_split: a sample's result does not depend on the batch it is in)
int dense_fwd_split_packed(int hw, int cin) {
  // one 16-channel chunk per block: with ~1 block per CU a wave has its SIMD to itself and the LDS -> MFMA chain of a
  // chunk runs at 2.6x its issue time (ablation, tools/abl_small.sh); more, shorter blocks interleave instead
  (void)hw;
  return std::max(1, std::min(40, (cin + 15) / 16));
}

long long wgrad_chunks(long long total_items, int mgroups, int ngroups, int* ipc, int want_blocks = 1024) {
  long long want = (want_blocks + (long long)mgroups * ngroups - 1) / ((long long)mgroups * ngroups);
  if (want < 1) want = 1;
  if (want > total_items) want = total_items;
  long long per = (total_items + want - 1) / want;
  *ipc = (int)per;
  return (total_items + per - 1) / per;
}

// Lays out (or just sizes, when base == nullptr) the workspace for geometry (n,h,w).
size_t carve(rln_ctx* c, void* base, int n, int h, int w, int with_bwd, bool assign) {
  const int nd = c->cfg.n_down;
  std::vector<int> hs, ws;
  level_dims(c, h, w, hs, ws);
  Carver cv(base);
  std::vector<float*> S(nd + 1), G(nd + 1);
  // bf16 storage: the levels the split-operand kernel families cover -- forward AND backward: rows of 40 pixels or of a
  // multiple of 80 (d3_pull_supported; 97.5 % of the activation elements at 120x160, every level down to 30x40 at
  // 480x640) -- keep their stack as bf16 planes; the deep levels and levels of any other width stay fp32 (a 232x312
  // frame therefore runs the one-part bf16 arithmetic on fp32 stacks throughout: the exact-fp32 fall-back kernels of
  // its dense blocks read fp32 planes only)
  std::vector<int> lst(nd + 1, ST_F32);
  for (int L = 0; L <= nd; ++L)
    if (c->storage == 1 && (ws[L] == 40 || (ws[L] >= 80 && ws[L] % 80 == 0)) && hs[L] >= 4 && c->cfg.growth_rate <= 16)
      lst[L] = ST_BF16;
  for (int L = 0; L <= nd; ++L)
    S[L] = reinterpret_cast<float*>(
        cv.take<unsigned char>((size_t)n * c->levels[L].C * hs[L] * ws[L] * (size_t)st_bytes(lst[L])));
  for (int L = 0; L <= nd; ++L) G[L] = with_bwd ? cv.take<float>((size_t)n * c->levels[L].C * hs[L] * ws[L]) : nullptr;
  float* mean = cv.take<float>(c->n_chan);
  float* var = cv.take<float>(c->n_chan);
  float* invstd = cv.take<float>(c->n_chan);
  float* stdv = cv.take<float>(c->n_chan);
  float* S12 = cv.take<float>(2 * c->n_chan);
  float* ab = cv.take<float>(2 * c->n_ab);
  float* masks = cv.take<float>((size_t)n * c->drop_total);
  // scratch maxima
  size_t stat_max = 0, dy_max = 0, wp_max = 0, bp_max = 0, pool_bytes = 0, fs_max = 0;
  std::vector<int64_t> pool_off(c->ops.size(), -1);
  for (size_t k = 0; k < c->ops.size(); ++k) {
    const Op& o = c->ops[k];
    const int Hd = hs[o.dst_level], Wd = ws[o.dst_level];
    if (o.type == OP_FIRST || o.type == OP_DENSE) {
      int th, tw;
      const int tile = igemm_pick_tile(Hd, Wd);
      igemm_tile_dims(IG_CONV3_BN, tile, &th, &tw);
      const size_t blocks = (size_t)n * ((Hd + th - 1) / th) * ((Wd + tw - 1) / tw);
      stat_max = std::max(stat_max, blocks * o.cout * 2);
      if (o.type == OP_FIRST) {
        stat_max = std::max(stat_max, (size_t)512 * o.cout * 2);
        wp_max = std::max(wp_max, (size_t)512 * o.cout * o.cin * 9);
      }
      if (o.type == OP_DENSE && c->d3_fwd_np > 0) {
        int dth, dtw, drg;
        d3_fwd_pick_tile(Hd, Wd, c->d3_fwd_np, &dth, &dtw, &drg, lst[o.dst_level]);
        stat_max = std::max(stat_max, (size_t)n * ((Hd + dth - 1) / dth) * ((Wd + dtw - 1) / dtw) * o.cout * 2);
        if (c->d3_fwd_np == 1) stat_max = std::max(stat_max, (size_t)d3_fin_rows(Hd, Wd, n) * o.cout * 2);
      }
      if (o.type == OP_DENSE) {
        const int sp = std::max(dense_fwd_split(Hd * Wd, o.cin),
                                igemm_pack_sx(Hd, Wd) > 1 ? dense_fwd_split_packed(Hd * Wd, o.cin) : 1);
        if (sp > 1) {
          fs_max = std::max(fs_max, (size_t)sp * n * o.cout * Hd * Wd);
          stat_max = std::max(stat_max, (size_t)n * ((Hd * Wd + 255) / 256) * o.cout * 2);
        }
      }
      if (with_bwd) {
        stat_max = std::max(stat_max, blocks * o.cin * 2);
        if (c->d3_bwd_np > 0) stat_max = std::max(stat_max, (size_t)512 * D3_LMAX * (((size_t)o.cin + 15) / 16 * 16) * 2);
        dy_max = std::max(dy_max, (size_t)n * o.cout * Hd * Wd);
        bp_max = std::max(bp_max, (size_t)grad_finalize_rows(n, Hd, Wd) * o.cout);
        int wth, wtw, ipc;
        wgrad_tile_dims(WG_DENSE3, wgrad_pick_tile(Hd, Wd), &wth, &wtw);
        const long long items = (long long)n * ((Hd + wth - 1) / wth) * ((Wd + wtw - 1) / wtw);
        // both block shapes of the dense weight gradient (64 V channels / ~1024 blocks, 32 / ~1536 blocks)
        const long long nch = std::max({wgrad_chunks(items, (o.cout + 15) / 16, (o.cin + 63) / 64, &ipc),
                                        wgrad_chunks(items, (o.cout + 15) / 16, (o.cin + 31) / 32, &ipc, 1536),
                                        wgrad_chunks(items, (o.cout + 15) / 16, (o.cin + 15) / 16, &ipc, 2048)});
        wp_max = std::max(wp_max, (size_t)nch * o.cout * o.cin * 9);
        if (c->d3_bwd_np > 0 && o.type == OP_DENSE) {
          D3Wgrad g;
          memset(&g, 0, sizeof(g));
          g.st = lst[o.dst_level];
          d3_wgrad_plan(Hd, Wd, n, o.cin, &g);
          wp_max = std::max(wp_max, (size_t)g.nranges * o.cout * o.cin * 9);
        }
      }
    } else if (o.type == OP_TD) {
      const int Hs = hs[o.src_level], Ws = ws[o.src_level];
      int th, tw;
      const int tile = igemm_pick_tile(Hs, Ws);
      igemm_tile_dims(IG_CONV1_POOL, tile, &th, &tw);
      const size_t blocks = (size_t)n * ((Hs + th - 1) / th) * ((Ws + tw - 1) / tw);
      stat_max = std::max(stat_max, blocks * o.cout * 2);
      if (c->d3_fwd_np > 0 || c->d3_bwd_np > 0) stat_max = std::max(stat_max, (size_t)256 * std::max(o.cout, o.cin) * 2);
      pool_off[k] = (int64_t)pool_bytes;
      pool_bytes += (size_t)n * o.cout * Hd * Wd;
      pool_bytes = (pool_bytes + 255) & ~(size_t)255;
      if (with_bwd) {
        dy_max = std::max(dy_max, (size_t)n * o.cout * Hs * Ws);
        bp_max = std::max(bp_max, (size_t)grad_finalize_rows(n, Hs, Ws) * o.cout);
        int wth, wtw, ipc;
        wgrad_tile_dims(WG_PW1, wgrad_pick_tile(Hs, Ws), &wth, &wtw);
        const long long items = (long long)n * ((Hs + wth - 1) / wth) * ((Ws + wtw - 1) / wtw);
        const long long nch = wgrad_chunks(items, (o.cout + 63) / 64, (o.cin + 63) / 64, &ipc);
        wp_max = std::max(wp_max, (size_t)nch * o.cout * o.cin);
        if (c->d3_bwd_np > 0) {
          P1Wgrad g;
          memset(&g, 0, sizeof(g));
          g.Cout = o.cout;
          g.Cin = o.cin;
          g.H = Hs;
          g.W = Ws;
          g.N = n;
          p1_wgrad_plan(&g);
          wp_max = std::max(wp_max, (size_t)g.nranges * o.cout * o.cin);
        }
      }
    } else {  // OP_TU
      const int GHc = (Hd + 1) / 2, GWc = (Wd + 1) / 2;
      int th, tw;
      const int tile = igemm_pick_tile(GHc, GWc);
      igemm_tile_dims(IG_CONV3_RAW, tile, &th, &tw);
      const size_t blocks = (size_t)n * 4 * ((GHc + th - 1) / th) * ((GWc + tw - 1) / tw);
      stat_max = std::max(stat_max, blocks * o.cout * 2);
      if (c->d3_fwd_np > 0) stat_max = std::max(stat_max, (size_t)256 * o.cout * 2);
      if (with_bwd) {
        const int Hs = hs[o.src_level], Ws = ws[o.src_level];
        dy_max = std::max(dy_max, (size_t)n * o.cout * Hd * Wd);
        bp_max = std::max(bp_max, (size_t)grad_finalize_rows(n, Hd, Wd) * o.cout);
        int wth, wtw, ipc;
        wgrad_tile_dims(WG_CONVT, wgrad_pick_tile(Hs, Ws), &wth, &wtw);
        const long long items = (long long)n * ((Hs + wth - 1) / wth) * ((Ws + wtw - 1) / wtw);
        const long long nch = wgrad_chunks(items, (o.cout + 15) / 16, (o.cin + 63) / 64, &ipc);
        wp_max = std::max(wp_max, (size_t)nch * o.cout * o.cin * 9);
        if (c->d3_bwd_np > 0 && (Ws % 8) == 0 && Ws >= 8) {
          C3Wgrad g;
          memset(&g, 0, sizeof(g));
          g.Cout = o.cout;
          g.Cin = o.cin;
          g.H = Hs;
          g.W = Ws;
          g.N = n;
          c3_wgrad_plan(&g);
          wp_max = std::max(wp_max, (size_t)g.nranges * o.cout * o.cin * 9);
        }
      }
    }
  }
  const size_t hw0 = (size_t)h * w;
  if (with_bwd) {
    bp_max = std::max(bp_max, (size_t)n * ((hw0 + 255) / 256) * c->cfg.n_classes);
    wp_max = std::max(wp_max, (size_t)std::max<long long>(n, head_backward_rows(n, (int)hw0)) * c->cfg.n_classes *
                                  c->feat_C);
  }
  float* stat_partial = cv.take<float>(stat_max);
  unsigned char* pool_idx = cv.take<unsigned char>(pool_bytes);
  float* fsplit = cv.take<float>(fs_max);
  float* dY = with_bwd ? cv.take<float>(dy_max) : nullptr;
  float* dY2 = with_bwd ? cv.take<float>(dy_max) : nullptr;
  float* dy16 = with_bwd ? cv.take<float>(2 * ((dy_max + 1) / 2)) : nullptr;  // 2 x dy_max bf16 elements (layer parity)
  float* wpartial = with_bwd ? cv.take<float>(wp_max) : nullptr;
  float* wpartial2 = with_bwd ? cv.take<float>(wp_max) : nullptr;
  float* bpartial = with_bwd ? cv.take<float>(bp_max) : nullptr;
  float* glin = with_bwd ? cv.take<float>((size_t)n * c->cfg.n_classes * hw0) : nullptr;
  // split-operand dense kernels: packed weight fragments + descriptor tables
  std::vector<D3PackDesc> dfh, dbh;
  std::vector<long long> wf_off(c->ops.size(), -1), wb_off(c->ops.size(), -1);
  long long pk_total = 0;
  int units_f = 0, units_b = 0;
  for (size_t k = 0; k < c->ops.size(); ++k) {
    const Op& o = c->ops[k];
    if (o.type != OP_DENSE || o.cout > 16) continue;
    D3PackDesc d;
    d.w_off = o.conv.w;
    d.cin = o.cin;
    d.cout = o.cout;
    d.n_units = 5 * ((o.cin + 15) / 16);
    if (c->d3_fwd_np > 0) {
      d.wf_off = pk_total;
      d.wb_off = -1;
      d.unit_begin = units_f;
      wf_off[k] = pk_total;
      pk_total += d3_pack_entries(o.cin, c->d3_fwd_np);
      units_f += d.n_units;
      dfh.push_back(d);
    }
    if (c->d3_bwd_np > 0 && with_bwd) {
      d.wf_off = -1;
      d.wb_off = pk_total;
      d.unit_begin = units_b;
      wb_off[k] = pk_total;
      pk_total += d3_pack_entries(o.cin, c->d3_bwd_np);
      units_b += d.n_units;
      dbh.push_back(d);
    }
  }
  std::vector<P1PackDesc> pfh, pbh;
  std::vector<long long> p1f_off(c->ops.size(), -1), p1b_off(c->ops.size(), -1);
  int pu_f = 0, pu_b = 0;
  for (size_t k = 0; k < c->ops.size(); ++k) {
    const Op& o = c->ops[k];
    if (o.type != OP_TD) continue;
    P1PackDesc d;
    d.w_off = o.conv.w;
    d.cin = o.cin;
    d.cout = o.cout;
    if (c->d3_fwd_np > 0) {
      d.wf_off = pk_total;
      d.wb_off = -1;
      d.unit_begin = pu_f;
      d.n_units = p1_units_f(o.cin, o.cout);
      p1f_off[k] = pk_total;
      pk_total += (long long)d.n_units * c->d3_fwd_np * 64;
      pu_f += d.n_units;
      pfh.push_back(d);
    }
    if (c->d3_bwd_np > 0 && with_bwd) {
      d.wf_off = -1;
      d.wb_off = pk_total;
      d.unit_begin = pu_b;
      d.n_units = p1_units_b(o.cin, o.cout);
      p1b_off[k] = pk_total;
      pk_total += (long long)d.n_units * c->d3_bwd_np * 64;
      pu_b += d.n_units;
      pbh.push_back(d);
    }
  }
  std::vector<C3PackDesc> cfh, cbh;
  std::vector<long long> c3f_off(c->ops.size(), -1), c3b_off(c->ops.size(), -1);
  int cu_f = 0, cu_b = 0;
  for (size_t k = 0; k < c->ops.size(); ++k) {
    const Op& o = c->ops[k];
    if (o.type != OP_TU) continue;
    C3PackDesc d;
    d.w_off = o.conv.w;
    d.cin = o.cin;
    d.cout = o.cout;
    if (c->d3_fwd_np > 0) {
      d.wf_off = pk_total;
      d.wb_off = -1;
      d.unit_begin = cu_f;
      d.n_units = c3_units_f(o.cin, o.cout);
      c3f_off[k] = pk_total;
      pk_total += (long long)d.n_units * c->d3_fwd_np * 64;
      cu_f += d.n_units;
      cfh.push_back(d);
    }
    if (c->d3_bwd_np > 0 && with_bwd) {
      d.wf_off = -1;
      d.wb_off = pk_total;
      d.unit_begin = cu_b;
      d.n_units = c3_units_b(o.cin, o.cout);
      c3b_off[k] = pk_total;
      pk_total += c3_entries_b(o.cin, o.cout, c->d3_bwd_np);
      cu_b += d.n_units;
      cbh.push_back(d);
    }
  }
  // pull-form data gradient: the dY of every layer of a block stays alive until the block's input channels are done
  int max_block_layers = 0;
  size_t dy_dense_max = 0;
  if (c->d3_bwd_np > 0 && with_bwd) {
    int run = 0;
    for (size_t k = 0; k < c->ops.size(); ++k) {
      const Op& o = c->ops[k];
      const bool cont = o.type == OP_DENSE && k > 0 && c->ops[k - 1].type == OP_DENSE &&
                        c->ops[k - 1].src_level == o.src_level && c->ops[k - 1].in_off == o.in_off;
      run = o.type == OP_DENSE ? (cont ? run + 1 : 1) : 0;
      max_block_layers = std::max(max_block_layers, run);
      if (o.type == OP_DENSE)
        dy_dense_max = std::max(dy_dense_max, (size_t)n * o.cout * hs[o.dst_level] * ws[o.dst_level]);
    }
  }
  std::vector<float*> dyblk(max_block_layers, nullptr);
  for (int i = 0; i < max_block_layers; ++i) dyblk[i] = cv.take<float>(dy_dense_max);
  uint4* d3_packed = cv.take<uint4>((size_t)pk_total);
  D3PackDesc* d3_df = cv.take<D3PackDesc>(dfh.size());
  D3PackDesc* d3_db = cv.take<D3PackDesc>(dbh.size());
  P1PackDesc* p1_df = cv.take<P1PackDesc>(pfh.size());
  P1PackDesc* p1_db = cv.take<P1PackDesc>(pbh.size());
  C3PackDesc* c3_df = cv.take<C3PackDesc>(cfh.size());
  C3PackDesc* c3_db = cv.take<C3PackDesc>(cbh.size());
  int* lcounts = cv.take<int>(32 + 256);
  float* lpartial = cv.take<float>((size_t)loss_blocks((long long)n * hw0) * 4);
  float* lresult = cv.take<float>(64);
  // second layer's raw sums of a paired dense forward (one-part operands only; last, so that every other buffer keeps
  // its offset whether or not the mode uses it)
  float* fpair = c->d3_fwd_np == 1 ? cv.take<float>((size_t)n * 16 * h * w) : nullptr;
  if (assign) {
    for (int L = 0; L <= nd; ++L) {
      c->levels[L].H = hs[L];
      c->levels[L].W = ws[L];
      c->levels[L].S = S[L];
      c->levels[L].G = G[L];
      c->levels[L].st = lst[L];
    }
    c->mean = mean;
    c->var = var;
    c->invstd = invstd;
    c->stdv = stdv;
    c->S1 = S12;
    c->S2 = S12 + c->n_chan;
    c->ab = ab;
    c->masks = masks;
    c->stat_partial = stat_partial;
    c->pool_idx = pool_idx;
    c->pool_off = pool_off;
    c->fsplit = fsplit;
    c->fpair = fpair;
    c->dY = dY;
    c->dYbuf[0] = dY;
    c->dYbuf[1] = dY2;
    c->dy16 = dy16;
    c->dy16_half = (size_t)((dy_max + 1) / 2) * sizeof(float);
    c->wpartial = wpartial;
    c->wpartial2 = wpartial2;
    c->bpartial = bpartial;
    c->glin = glin;
    c->dyblk = dyblk;
    c->d3_desc_f = dfh;
    c->d3_desc_b = dbh;
    c->d3_wf_off = wf_off;
    c->d3_wb_off = wb_off;
    c->d3_packed = d3_packed;
    c->d3_desc_f_dev = d3_df;
    c->d3_desc_b_dev = d3_db;
    c->d3_units_f = units_f;
    c->d3_units_b = units_b;
    c->p1_desc_f = pfh;
    c->p1_desc_b = pbh;
    c->p1_wf_off = p1f_off;
    c->p1_wb_off = p1b_off;
    c->p1_desc_f_dev = p1_df;
    c->p1_desc_b_dev = p1_db;
    c->p1_units_f = pu_f;
    c->p1_units_b = pu_b;
    c->c3_desc_f = cfh;
    c->c3_desc_b = cbh;
    c->c3_wf_off = c3f_off;
    c->c3_wb_off = c3b_off;
    c->c3_desc_f_dev = c3_df;
    c->c3_desc_b_dev = c3_db;
    c->c3_units_f = cu_f;
    c->c3_units_b = cu_b;
    c->loss.counts = lcounts;
    c->loss.partial = lpartial;
    c->loss.result = lresult;
  }
  return (cv.off + 255) & ~(size_t)255;
}

inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// Brackets one launch (or a small group) with HIP events on the launch stream when profiling is on.
#ifdef RLN_DIAG
// Diagnostic builds, RLN_POISON_LDS=1: before every profiled launch each CU's LDS is filled with 0xFF (NaN patterns), so a
// kernel that reads LDS cells it has not written shows up as changed / non-finite results (tools/poison_probe.py) instead
// of depending on what the previous kernel -- or the previous tenant of the GPU -- left there.
__global__ __launch_bounds__(1024) void lds_poison_k() {
  extern __shared__ unsigned lds_words[];
  for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 1024) lds_words[i] = 0xFFFFFFFFu;
  __syncthreads();
  if (lds_words[(threadIdx.x * 37) % (160 * 1024 / 4)] != 0xFFFFFFFFu) __builtin_trap();  // (keeps the stores alive)
}
static void lds_poison(hipStream_t s) {
  static const bool on = rln_env("RLN_POISON_LDS") != nullptr;
  if (!on) return;
  static DevOnce once;
  if (once.first())
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lds_poison_k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(lds_poison_k, dim3(2048), dim3(1024), 160 * 1024, s);
}
#else
static inline void lds_poison(hipStream_t) {}
#endif

struct ProfScope {
  rln_ctx* c;
  hipStream_t s;
  int idx = -1;
  ProfScope(rln_ctx* c_, int cls, double flops, double bytes, hipStream_t s_) : c(c_), s(s_) {
    lds_poison(s);
    if (!c->prof.on) return;
    ProfEntry e;
    e.a = c->prof.get();
    e.b = c->prof.get();
    e.cls = cls;
    e.flops = flops;
    e.bytes = bytes;
    if (!e.a || !e.b) return;
    (void)hipEventRecord(e.a, s);
    c->prof.entries.push_back(e);
    idx = (int)c->prof.entries.size() - 1;
  }
  ~ProfScope() {
    if (idx >= 0) (void)hipEventRecord(c->prof.entries[idx].b, s);
  }
};

// ---------------------------------------------------------------------------------------------
// forward ops
// ---------------------------------------------------------------------------------------------

// k: index of the op that produced the channels; when the next op normalises a range of the same level that
// contains them, its bn_prep rides in the same launch (c->prep_done = index of that op).
int finalize_stats(rln_ctx* c, int level, int ch_off, int J, long long nblk, hipStream_t s, long long k = -1) {
  const Level& lv = c->levels[level];
  const double count = (double)c->N * lv.H * lv.W;
  const int64_t so = lv.stat_off + ch_off;
  ProfScope ps(c, PC_BN, 0, 0, s);
  static const bool nofuse = rln_env("RLN_NO_BNFUSE") != nullptr;
  if (!nofuse && k >= 0 && (size_t)(k + 1) < c->ops.size()) {
    const Op& nx = c->ops[(size_t)k + 1];
    if ((nx.type == OP_DENSE || nx.type == OP_TD) && nx.src_level == level && nx.in_off <= ch_off &&
        ch_off + J <= nx.in_off + nx.bn.C) {
      const int64_t son = lv.stat_off + nx.in_off;
      RLN_TRY(bn_finalize_prep(c->stat_partial, nblk, J, ch_off - nx.in_off, count, c->cfg.bn_eps, c->mean + son,
                               c->var + son, c->invstd + son, c->stdv + son, nx.bn.C, c->params + nx.bn.gamma,
                               c->params + nx.bn.beta, c->bnrun + nx.bn.rmean, c->bnrun + nx.bn.rvar,
                               c->cfg.bn_momentum, c->ab + nx.bn.ab, c->ab + c->n_ab + nx.bn.ab, s));
      c->prep_done = k + 1;
      return 0;
    }
  }
  RLN_TRY(bn_finalize(c->stat_partial, nblk, J, count, c->cfg.bn_eps, c->mean + so, c->var + so, c->invstd + so,
                      c->stdv + so, s));
  return 0;
}

int prep_bn(rln_ctx* c, const Op& o, int training, hipStream_t s, long long k = -1) {
  if (!training && c->eval_reuse) return 0;  // rln_set_eval_cache: the tables of the previous eval forward stand
  if (training && k >= 0 && c->prep_done == k) {  // done together with the previous op's statistics
    c->prep_done = -1;
    return 0;
  }
  const Level& lv = c->levels[o.src_level];
  const int64_t so = lv.stat_off + o.in_off;
  const double count = (double)c->N * lv.H * lv.W;
  ProfScope ps(c, PC_BN, 0, 0, s);
  RLN_TRY(bn_prep(training, o.bn.C, c->params + o.bn.gamma, c->params + o.bn.beta, c->mean + so, c->var + so,
                  c->invstd + so, c->bnrun + o.bn.rmean, c->bnrun + o.bn.rvar, c->cfg.bn_momentum, count,
                  c->cfg.bn_eps, c->ab + o.bn.ab, c->ab + c->n_ab + o.bn.ab, s));
  return 0;
}

static bool convt_fused() {
  static const bool off = rln_env("RLN_NO_CONVT4") != nullptr;
  return !off;
}

int fwd_op(rln_ctx* c, size_t k, const float* x, int training, hipStream_t s) {
  const Op& o = c->ops[k];
  const int N = c->N;
  const Level& dl = c->levels[o.dst_level];
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.ncls = 1;
  p.tapmode = TM_ID;
  p.J = o.cout;
  p.K = o.cin;
  p.bias = c->params + o.conv.b;
  p.nscale = (training && o.drop_ch >= 0) ? (c->masks + (size_t)N * o.drop_ch) : nullptr;
  p.stat_partial = training ? c->stat_partial : nullptr;
  p.out = dl.sp(o.out_off);
  p.out_ns = (long long)dl.C * dl.H * dl.W;
  p.out_cs = dl.H * dl.W;
  p.Hout = dl.H;
  p.Wout = dl.W;
  p.w = c->params + o.conv.w;
  IgemmKind kind;
  int tile;
  if (o.type == OP_FIRST) {
    kind = IG_CONV3_RAW;
    p.in = x;
    p.in_ns = (long long)o.cin * dl.H * dl.W;
    p.in_cs = dl.H * dl.W;
    p.Hin = dl.H;
    p.Win = dl.W;
    p.w_js = (long long)o.cin * 9;
    p.w_ks = 9;
    p.GH = dl.H;
    p.GW = dl.W;
  } else if (o.type == OP_DENSE) {
    kind = IG_CONV3_BN;
    RLN_TRY(prep_bn(c, o, training, s, (long long)k));
    p.in = dl.sp(o.in_off);
    p.in_ns = p.out_ns;
    p.in_cs = p.out_cs;
    p.Hin = dl.H;
    p.Win = dl.W;
    p.pa = c->ab + o.bn.ab;
    p.pb = c->ab + c->n_ab + o.bn.ab;
    p.w_js = (long long)o.cin * 9;
    p.w_ks = 9;
    p.GH = dl.H;
    p.GW = dl.W;
  } else if (o.type == OP_TD) {
    kind = IG_CONV1_POOL;
    RLN_TRY(prep_bn(c, o, training, s, (long long)k));
    const Level& sl = c->levels[o.src_level];
    p.in = sl.sp(o.in_off);
    p.in_ns = (long long)sl.C * sl.H * sl.W;
    p.in_cs = sl.H * sl.W;
    p.Hin = sl.H;
    p.Win = sl.W;
    p.pa = c->ab + o.bn.ab;
    p.pb = c->ab + c->n_ab + o.bn.ab;
    p.w_js = o.cin;
    p.w_ks = 1;
    p.GH = sl.H;
    p.GW = sl.W;
    p.pool_idx = c->pool_idx + c->pool_off[k];
  } else {  // OP_TU: ConvTranspose2d by output parity class
    kind = IG_CONV3_RAW;
    const Level& sl = c->levels[o.src_level];
    p.in = sl.sp(o.in_off);
    p.in_ns = (long long)sl.C * sl.H * sl.W;
    p.in_cs = sl.H * sl.W;
    p.Hin = sl.H;
    p.Win = sl.W;
    p.w_ks = (long long)o.cout * 9;  // weight[c_in][c_out][ky][kx]
    p.w_js = 9;
    p.tapmode = TM_CONVT;
    p.ncls = 4;
    if (convt_fused()) {  // all four output-parity classes from one staged input tile
      kind = IG_CONVT4;
      p.tapmode = TM_ID;
      p.ncls = 1;
    }
    p.GH = (dl.H + 1) / 2;
    p.GW = (dl.W + 1) / 2;
  }
  if (o.type == OP_DENSE && c->d3_fwd_np > 0 && c->d3_wf_off[k] >= 0) {  // split-operand 16-bit MFMA kernel
    D3Fwd q;
    memset(&q, 0, sizeof(q));
    q.S = p.in;
    q.ns = p.in_ns;
    q.cs = p.in_cs;
    q.H = dl.H;
    q.W = dl.W;
    q.Cin = o.cin;
    q.pa = p.pa;
    q.pb = p.pb;
    q.wpk = c->d3_packed + c->d3_wf_off[k];
    q.bias = p.bias;
    q.nscale = p.nscale;
    q.out = p.out;
    q.out_ns = p.out_ns;
    q.out_cs = p.out_cs;
    q.Cout = o.cout;
    q.stat_partial = p.stat_partial;
    q.ksplit = 1;
    q.st = dl.st;
    // Small levels (one tile does not fill the chip): the input-channel loop is split over blocks exactly like the
    // exact-fp32 family does it -- raw partial sums to scratch, bias / Dropout2d scale / statistics in the finish pass
    // (only where one tile per sample leaves the chip empty: rows of < 40 pixels; the 30x40 level is faster unsplit)
    const int sp3 = (dl.st == ST_F32 && dl.W < 40) ? dense_fwd_split(dl.H * dl.W, o.cin) : 1;
    if (sp3 > 1) {
      q.ksplit = sp3;
      q.split_stride = (long long)N * o.cout * dl.H * dl.W;
      q.out = c->fsplit;
      q.out_ns = (long long)o.cout * dl.H * dl.W;
      q.out_cs = dl.H * dl.W;
      q.bias = nullptr;
      q.nscale = nullptr;
      q.stat_partial = nullptr;
    }
    if (d3_fwd_supported(q)) {
      d3_fwd_pick_tile(q.H, q.W, c->d3_fwd_np, &q.th, &q.tw, &q.rg, q.st);
      q.tiles_y = (q.H + q.th - 1) / q.th;
      q.tiles_x = (q.W + q.tw - 1) / q.tw;
      int e3;
      long long fin_rows = 0;  // statistics partial rows when the finishing kernel ran (its own tiling)
      // Paired forward (dense3.h: d3_fwd_pair_launch): this layer and the next one of the block over the input channels
      // they share, one load per chunk; the next op then only adds its last chunk (c_first / partial_in).
      static const bool no_fwd_pair = rln_env("RLN_NO_FWD_PAIR") != nullptr;
      bool paired = false;
      if (c->pair_finish == (long long)k) {  // second layer of a pair: the chunks before c_first are in fpair already
        const Op& prev = c->ops[k - 1];
        q.c_first = prev.cin / 16;
        q.partial_in = c->fpair;
        c->pair_finish = -1;
      } else if (!no_fwd_pair && c->d3_fwd_np == 1 && sp3 == 1 && c->fpair != nullptr && o.cout == 16 && (o.cin % 16) == 0) {
        auto chained = [&](size_t a) {  // op a+1 is the next layer of the same block
          if (a + 1 >= c->ops.size()) return false;
          const Op& x0 = c->ops[a];
          const Op& x1 = c->ops[a + 1];
          return x1.type == OP_DENSE && x0.type == OP_DENSE && x1.dst_level == x0.dst_level && x1.in_off == x0.in_off &&
                 x1.cin == x0.cin + x0.cout && x1.cout == 16 && x0.cout == 16 && c->d3_wf_off[a + 1] >= 0 &&
                 x1.out_off == x0.out_off + x0.cout;
        };
        int run = 1;  // consecutive chained dense layers from k on: pair (k, k+1) when that number is even
        while (chained(k + run - 1)) ++run;
        if (run >= 2 && (run % 2) == 0) {
          const Op& nx = c->ops[k + 1];
          // the next layer's BN table over the channels both layers read (their statistics exist; no running-stat update here:
          // the statistics pass of this layer prepares the complete table, running statistics included)
          {
            const Level& lvn = c->levels[nx.src_level];
            const int64_t son = lvn.stat_off + nx.in_off;
            const double count = (double)c->N * lvn.H * lvn.W;
            ProfScope psb(c, PC_BN, 0, 0, s);
            if (!(!training && c->eval_reuse))
              RLN_TRY(bn_prep(training, o.cin, c->params + nx.bn.gamma, c->params + nx.bn.beta, c->mean + son, c->var + son,
                              c->invstd + son, training ? nullptr : c->bnrun + nx.bn.rmean,
                              training ? nullptr : c->bnrun + nx.bn.rvar, c->cfg.bn_momentum, count, c->cfg.bn_eps,
                              c->ab + nx.bn.ab, c->ab + c->n_ab + nx.bn.ab, s));
          }
          D3Fwd q2 = q;
          q2.pa2 = c->ab + nx.bn.ab;
          q2.pb2 = c->ab + c->n_ab + nx.bn.ab;
          q2.wpk2 = c->d3_packed + c->d3_wf_off[k + 1];
          q2.partial_out = c->fpair;
          const double flops = 2.0 * 2.0 * o.cin * o.cout * 9.0 * q.H * q.W * N;
          const double bytes = (double)N * q.H * q.W * (st_bytes(dl.st) * ((double)o.cin + o.cout) + 4.0 * o.cout);
          ProfScope ps(c, PC_D3_FWD2, flops, bytes, s);  // its own class: one kernel symbol (d3_fwd2_k) per class
          const int e2 = d3_fwd_pair_launch(q2, N, c->d3_fwd_np, c->d3_fwd_dt, s);
          if (e2 == 0) {
            paired = true;
            c->pair_finish = (long long)k + 1;
          } else if (e2 != RLN_ERR_UNSUPPORTED) {
            return fail(e2, "d3_fwd_pair_launch failed with %d (op %zu)", e2, k);
          }
        }
      }
      if (paired) {
        e3 = 0;
      } else {
        // (a finishing launch contracts only the 16 channels its partner wrote and reads the fp32 raw sums)
        const double kin = q.partial_in ? 16.0 : (double)o.cin;
        const double flops = 2.0 * kin * o.cout * 9.0 * q.H * q.W * N;
        const double bytes = (double)N * q.H * q.W * (st_bytes(dl.st) * (kin + o.cout) + (q.partial_in ? 4.0 * o.cout : 0.0));
        // one class per kernel instantiation, so that a class's average launch time is a row of the rocprofv3 summary:
        // d3_fwd_k<10, ...> (tiles of up to 640 pixels: the wide levels) / d3_fwd_k<5, ...> (<= 320 pixels)
        static const bool no_fin = rln_env("RLN_NO_FIN") != nullptr;
        if (q.partial_in != nullptr && !no_fin && d3_fin_supported(q, c->d3_fwd_np)) {
          // the one chunk a pair's second layer still owes: light 4-wave kernel, several blocks per CU (dense3.h)
          ProfScope ps(c, PC_D3_FIN, flops, bytes, s);
          e3 = d3_fin_launch(q, N, c->d3_fwd_np, c->d3_fwd_dt, s);
          if (e3 == 0) fin_rows = d3_fin_rows(q.H, q.W, N);
        } else {
          ProfScope ps(c, q.th * q.tw > 320 ? PC_D3_FWD : PC_D3_FWD_S, flops, bytes, s);
          e3 = d3_fwd_launch(q, N, c->d3_fwd_np, c->d3_fwd_dt, s);
        }
      }
      if (e3 != 0 && q.partial_in != nullptr) return fail(e3, "the finishing launch of a paired forward failed (op %zu)", k);
      // RLN_ERR_UNSUPPORTED from the launcher (tile / LDS budget of an unusual geometry) falls through to the exact-fp32
      // family below, like a geometry d3_fwd_supported() rejects; nothing was launched in that case
      if (e3 != 0 && (e3 != RLN_ERR_UNSUPPORTED || dl.st != ST_F32))
        return fail(e3, "d3_fwd_launch failed with %d (op %zu)", e3, k);
      if (e3 == 0 && sp3 > 1) {
        long long nblk = 0;
        ProfScope ps(c, PC_REDUCE, 0, 0, s);
        RLN_TRY(splitk_finish(c->fsplit, sp3, q.split_stride, N, o.cout, dl.H * dl.W, p.bias, p.nscale, p.out, p.out_ns,
                              p.stat_partial, &nblk, s));
        if (training) RLN_TRY(finalize_stats(c, o.dst_level, o.out_off, o.cout, nblk, s, (long long)k));
        return 0;
      }
      if (e3 == 0) {
        if (training)
          RLN_TRY(finalize_stats(c, o.dst_level, o.out_off, o.cout,
                                 fin_rows > 0 ? fin_rows : (long long)N * q.tiles_x * q.tiles_y, s, (long long)k));
        return 0;
      }
    }
  }
  if (o.type == OP_FIRST && c->d3_fwd_np > 0) {  // split-operand 16-bit MFMA kernel (fc3.h)
    F3Fwd q;
    memset(&q, 0, sizeof(q));
    q.X = x;
    q.Cin = o.cin;
    q.H = dl.H;
    q.W = dl.W;
    q.N = N;
    q.w = p.w;
    q.bias = p.bias;
    q.out = p.out;
    q.out_ns = p.out_ns;
    q.out_cs = p.out_cs;
    q.Cout = o.cout;
    q.stat_partial = p.stat_partial;
    q.ot = dl.st;
    if (f3_fwd_supported(q)) {
      f3_fwd_plan(&q);
      {
        const double flops = 2.0 * o.cin * o.cout * 9.0 * dl.H * dl.W * N;
        const double bytes = (double)N * (4.0 * o.cin + (double)st_bytes(dl.st) * o.cout) * dl.H * dl.W;
        ProfScope ps(c, PC_FIRST_FWD, flops, bytes, s);
        RLN_TRY(f3_fwd_launch(q, c->d3_fwd_np, c->d3_fwd_dt, s));
      }
      if (training) RLN_TRY(finalize_stats(c, o.dst_level, o.out_off, o.cout, (long long)q.blocks, s, (long long)k));
      return 0;
    }
  }
  if (o.type == OP_TU && c->d3_fwd_np > 0 && c->c3_wf_off[k] >= 0) {  // split-operand 16-bit MFMA kernel (ct3.h)
    const Level& sl = c->levels[o.src_level];
    C3Fwd q;
    memset(&q, 0, sizeof(q));
    q.X = p.in;
    q.ns = p.in_ns;
    q.cs = p.in_cs;
    q.H = sl.H;
    q.W = sl.W;
    q.Cin = o.cin;
    q.N = N;
    q.wpk = c->d3_packed + c->c3_wf_off[k];
    q.bias = p.bias;
    q.out = p.out;
    q.out_ns = p.out_ns;
    q.out_cs = p.out_cs;
    q.Cout = o.cout;
    q.Ho = dl.H;
    q.Wo = dl.W;
    q.stat_partial = p.stat_partial;
    q.st = sl.st;
    q.ot = dl.st;
    if (c3_fwd_supported(q) && c3_fwd_fits(q, c->d3_fwd_np)) {
      c3_fwd_plan(&q, c->d3_fwd_np);
      {
        const double flops = 2.0 * o.cin * o.cout * 9.0 * sl.H * sl.W * N;
        const double bytes = (double)N * ((double)st_bytes(sl.st) * o.cin * sl.H * sl.W +
                                          (double)st_bytes(dl.st) * o.cout * dl.H * dl.W);
        ProfScope ps(c, PC_TU_FWD, flops, bytes, s);
        RLN_TRY(c3_fwd_launch(q, c->d3_fwd_np, c->d3_fwd_dt, s));
      }
      if (training) RLN_TRY(finalize_stats(c, o.dst_level, o.out_off, o.cout, (long long)q.bpg, s, (long long)k));
      return 0;
    }
  }
  if (o.type == OP_TD && c->d3_fwd_np > 0 && c->p1_wf_off[k] >= 0) {  // split-operand 16-bit MFMA kernel (pw1.h)
    const Level& sl = c->levels[o.src_level];
    P1Fwd q;
    memset(&q, 0, sizeof(q));
    q.S = p.in;
    q.ns = p.in_ns;
    q.cs = p.in_cs;
    q.H = sl.H;
    q.W = sl.W;
    q.Cin = o.cin;
    q.N = N;
    q.pa = p.pa;
    q.pb = p.pb;
    q.wpk = c->d3_packed + c->p1_wf_off[k];
    q.bias = p.bias;
    q.nscale = p.nscale;
    q.out = p.out;
    q.out_ns = p.out_ns;
    q.out_cs = p.out_cs;
    q.Cout = o.cout;
    q.pool_idx = p.pool_idx;
    q.stat_partial = p.stat_partial;
    q.st = sl.st;
    q.ot = dl.st;
    if (p1_fwd_supported(q) && dl.H == sl.H / 2 && dl.W == sl.W / 2) {
      p1_fwd_plan(&q, c->d3_fwd_np);
      {
        const double flops = 2.0 * o.cin * o.cout * sl.H * sl.W * N;
        const double bytes = (double)N * ((double)st_bytes(sl.st) * o.cin * sl.H * sl.W +
                                          ((double)st_bytes(dl.st) + 1.0) * o.cout * dl.H * dl.W);
        ProfScope ps(c, PC_TD_FWD, flops, bytes, s);
        RLN_TRY(p1_fwd_launch(q, c->d3_fwd_np, c->d3_fwd_dt, s));
      }
      if (training) RLN_TRY(finalize_stats(c, o.dst_level, o.out_off, o.cout, (long long)q.bpg, s, (long long)k));
      return 0;
    }
  }
  // the exact-fp32 family below reads and writes fp32 stacks only
  if (dl.st != ST_F32 || (o.src_level >= 0 && c->levels[o.src_level].st != ST_F32))
    return fail(RLN_ERR_UNSUPPORTED, "op %zu: geometry not covered by the bf16-storage kernels (level %d -> %d)", k,
                o.src_level, o.dst_level);
  tile = igemm_pick_tile(p.GH, p.GW);
  if (kind == IG_CONV3_BN && igemm_pick_strip_tile(p.GW) >= 0 && aligned16(p.in) && (p.in_cs % 4) == 0 &&
      (p.in_ns % 4) == 0)
    tile = igemm_pick_strip_tile(p.GW);
  int th, tw;
  igemm_tile_dims(kind, tile, &th, &tw);
  p.tiles_y = (p.GH + th - 1) / th;
  p.tiles_x = (p.GW + tw - 1) / tw;
  p.out_vec = (p.ncls == 1 && (p.Wout % 4) == 0 && aligned16(p.out)) ? 1 : 0;
  {
    const double taps = (o.type == OP_TD) ? 1.0 : 9.0;
    const double flops = 2.0 * o.cin * o.cout * taps * p.Hin * p.Win * N;  // convT counted on its input grid
    const double bytes = 4.0 * N * ((double)o.cin * p.Hin * p.Win + (double)o.cout * p.Hout * p.Wout);
    const int cls = o.type == OP_FIRST ? PC_FIRST_FWD : o.type == OP_DENSE ? PC_DENSE_FWD
                    : o.type == OP_TD  ? PC_TD_FWD : PC_TU_FWD;
    ProfScope ps(c, cls, flops, bytes, s);
    int sp = (o.type == OP_DENSE) ? dense_fwd_split(dl.H * dl.W, o.cin) : 1;
    if (sp > 1) {  // split-K: raw partial sums to scratch, then bias / dropout scale / statistics in the finish pass
      IgemmParams q = p;
      const int HW = dl.H * dl.W;
      int qtile = tile;
      const int ng = (kind == IG_CONV3_BN) ? pack_small_level(q, dl.H, dl.W, N, &qtile) : N;
      if (q.pk_sx > 1) sp = dense_fwd_split_packed(HW, o.cin);
      q.ksplit = sp;
      q.split_stride = (long long)N * o.cout * HW;
      q.out = c->fsplit;
      q.out_ns = (long long)o.cout * HW;
      q.out_cs = HW;
      q.bias = nullptr;
      q.nscale = nullptr;
      q.stat_partial = nullptr;
      q.out_vec = (q.pk_sx <= 1 && (dl.W % 4) == 0 && aligned16(q.out)) ? 1 : 0;
      RLN_TRY(igemm_launch(kind, qtile, q, ng, s));
      long long nblk = 0;
      RLN_TRY(splitk_finish(c->fsplit, sp, q.split_stride, N, o.cout, HW, p.bias, p.nscale, p.out, p.out_ns,
                            p.stat_partial, &nblk, s));
      if (training) RLN_TRY(finalize_stats(c, o.dst_level, o.out_off, o.cout, nblk, s, (long long)k));
      return 0;
    }
    RLN_TRY(igemm_launch(kind, tile, p, N, s));
  }
  if (training) RLN_TRY(finalize_stats(c, o.dst_level, o.out_off, o.cout, igemm_stat_blocks(p, N), s, (long long)k));
  return 0;
}

HeadParams head_params(rln_ctx* c) {
  const Level& l0 = c->levels[0];
  HeadParams h;
  h.S = l0.S;
  h.ns = (long long)l0.C * l0.H * l0.W;
  h.C = l0.C;
  h.HW = l0.H * l0.W;
  h.ncls = c->cfg.n_classes;
  h.w = c->params + c->cls.w;
  h.b = c->params + c->cls.b;
  h.T = c->cfg.temperature;
  h.st = l0.st;
  return h;
}

// ---------------------------------------------------------------------------------------------
// backward ops
// ---------------------------------------------------------------------------------------------

int finalize_grad_range(rln_ctx* c, int level, int ch_off, int C, const float* nscale, long long* rows,
                        hipStream_t s, float* dst = nullptr, int yt = -1, void* dst16 = nullptr) {
  const Level& lv = c->levels[level];
  GradFinParams g;
  memset(&g, 0, sizeof(g));
  const size_t plane = (size_t)lv.H * lv.W;
  g.S = lv.sp(ch_off);
  g.G = lv.G + ch_off * plane;
  g.ns = (long long)lv.C * plane;
  g.C = C;
  g.H = lv.H;
  g.W = lv.W;
  const int64_t so = lv.stat_off + ch_off;
  g.mean = c->mean + so;
  g.invstd = c->invstd + so;
  g.S1 = c->S1 + so;
  g.S2 = c->S2 + so;
  g.invM = (float)(1.0 / ((double)c->N * plane));
  g.nscale = nscale;
  g.dst = dst ? dst : c->dY;
  g.bias_partial = c->bpartial;
  g.Hd = lv.H;
  g.Wd = lv.W;
  g.st = lv.st;
  g.yt = yt < 0 ? lv.st : yt;  // the finalised gradients share the level's storage type unless the consumer asks otherwise
  g.dst16 = dst16;
  ProfScope ps(c, PC_GRADFIN, 0, (4.0 + st_bytes(lv.st) + st_bytes(g.yt) + (dst16 ? 2.0 : 0.0)) * c->N * C * plane, s);
  RLN_TRY(grad_finalize(g, c->N, rows, s));
  return 0;
}

int run_wgrad(rln_ctx* c, WgradKind kind, WgradParams& w, int Mc, int Nc, int64_t grad_off, hipStream_t s,
              long long* defer_rows = nullptr) {
  const int pcls = kind == WG_DENSE3 ? PC_DENSE_WGRAD : kind == WG_RAW3 ? PC_FIRST_WGRAD
                   : kind == WG_PW1  ? PC_TD_WGRAD : PC_TU_WGRAD;
  const double taps = (kind == WG_PW1) ? 1.0 : 9.0;
  const double wflops = 2.0 * Mc * Nc * taps * w.GH * w.GW * c->N;
  const double wbytes = 4.0 * c->N * ((double)w.Uc * w.GH * w.GW + (double)w.Vc * w.Hv * w.Wv);
  const int tile = wgrad_pick_tile(w.GH, w.GW);
  int th, tw, mpb, npb;
  wgrad_tile_dims(kind, tile, &th, &tw);
  wgrad_block_dims(kind, &mpb, &npb);
  int want_blocks = 1024;
  if (kind == WG_DENSE3 && wgrad_dense_q_channels(w) == 32) {  // 32-channel blocks, three per CU
    npb = 32;
    want_blocks = 1536;
  } else if (kind == WG_DENSE3 && wgrad_dense_q_channels(w) == 16) {
    npb = 16;
    want_blocks = 2048;
  }
  w.tiles_y = (w.GH + th - 1) / th;
  w.tiles_x = (w.GW + tw - 1) / tw;
  w.N = c->N;
  const long long items = (long long)c->N * w.tiles_x * w.tiles_y;
  int ipc;
  const long long nch = wgrad_chunks(items, (Mc + mpb - 1) / mpb, (Nc + npb - 1) / npb, &ipc, want_blocks);
  w.items_per_chunk = ipc;
  w.nchunks = (int)nch;
  w.partial = c->wpartial;
  {
    ProfScope ps(c, pcls, wflops, wbytes, s);
    RLN_TRY(wgrad_launch(kind, tile, w, s));
  }
  if (defer_rows) {  // the caller reduces the slabs together with the layer's other small reductions
    *defer_rows = nch;
    return 0;
  }
  ProfScope ps2(c, PC_REDUCE, 0, 4.0 * (nch + 1) * w.wsize, s);
  RLN_TRY(reduce_rows(c->wpartial, nch, w.wsize, c->grads + grad_off, s));
  return 0;
}

// Picks the dY buffer for the next op and makes the main stream wait until the weight-gradient kernel that last
// read it (side stream) has finished.
int dy_acquire(rln_ctx* c, hipStream_t s) {
  c->dy_flip ^= 1;
  const int b = c->dy_flip;
  c->dY = c->dYbuf[b];
  if (c->use_side && c->wg_pending[b]) {
    hipError_t e = hipStreamWaitEvent(s, c->ev_wg[b], 0);
    if (e != hipSuccess) return fail((int)e, "hipStreamWaitEvent failed");
    c->wg_pending[b] = false;
  }
  return 0;
}
// Called after dY[current] has been written on the main stream: returns the stream the weight gradient runs on.
int wg_begin(rln_ctx* c, hipStream_t s, hipStream_t* ws) {
  *ws = s;
  if (!c->use_side) return 0;
  const int b = c->dy_flip;
  hipError_t e = hipEventRecord(c->ev_dy[b], s);
  if (e == hipSuccess) e = hipStreamWaitEvent(c->side, c->ev_dy[b], 0);
  if (e != hipSuccess) return fail((int)e, "side-stream hand-over failed");
  *ws = c->side;
  return 0;
}
int wg_end(rln_ctx* c) {
  if (!c->use_side) return 0;
  const int b = c->dy_flip;
  hipError_t e = hipEventRecord(c->ev_wg[b], c->side);
  if (e != hipSuccess) return fail((int)e, "hipEventRecord failed");
  c->wg_pending[b] = true;
  return 0;
}

// Operand parts of a weight-gradient GEMM whose K dimension runs over `pixels` positions: the reduced setting of
// rln_set_wgrad_parts applies where the pixel sum is long enough for the operand rounding to average out.
static inline int wgrad_parts(const rln_ctx* c, long long pixels) {
  return (pixels >= 2400) ? c->wg_parts : c->d3_bwd_np;
}

int bwd_op(rln_ctx* c, size_t k, hipStream_t s) {
  const Op& o = c->ops[k];
  const int N = c->N;
  long long rows = 0;
  hipStream_t ws = s;
  RLN_TRY(dy_acquire(c, s));
  if (o.type == OP_DENSE || o.type == OP_FIRST) {
    const Level& lv = c->levels[o.dst_level];
    const size_t plane = (size_t)lv.H * lv.W;
    const float* nscale = (o.drop_ch >= 0) ? (c->masks + (size_t)N * o.drop_ch) : nullptr;
    RLN_TRY(finalize_grad_range(c, o.dst_level, o.out_off, o.cout, nscale, &rows, s));
    // dense layers on one stream: the three small reductions of the layer run as ONE launch after the weight gradient
    static const bool no_tail = rln_env("RLN_NO_TAIL") != nullptr;
    const bool fuse_tail = o.type == OP_DENSE && !c->use_side && !no_tail;
    DenseTail tail;
    memset(&tail, 0, sizeof(tail));
    if (!fuse_tail) RLN_TRY(reduce_rows(c->bpartial, rows, o.cout, c->grads + o.conv.b, s));
    if (o.type == OP_DENSE) {
      IgemmParams p;
      memset(&p, 0, sizeof(p));
      p.in = c->dY;
      p.in_ns = (long long)o.cout * plane;
      p.in_cs = (int)plane;
      p.Hin = lv.H;
      p.Win = lv.W;
      p.K = o.cout;
      p.w = c->params + o.conv.w;  // W[o][c][tap]: j = c, k = o
      p.w_ks = (long long)o.cin * 9;
      p.w_js = 9;
      p.tapmode = TM_FLIP;
      p.J = o.cin;
      p.GH = lv.H;
      p.GW = lv.W;
      p.ncls = 1;
      p.out = lv.G + (size_t)o.in_off * plane;
      p.out_ns = (long long)lv.C * plane;
      p.out_cs = (int)plane;
      p.Hout = lv.H;
      p.Wout = lv.W;
      p.S = lv.sp(o.in_off);
      p.s_ns = p.out_ns;
      p.st = lv.st;
      const int64_t so = lv.stat_off + o.in_off;
      p.ea = c->ab + o.bn.ab;
      p.eb = c->ab + c->n_ab + o.bn.ab;
      p.emean = c->mean + so;
      p.einvstd = c->invstd + so;
      p.egamma = c->params + o.bn.gamma;
      p.acc_lo = o.acc_lo;
      p.acc_hi = o.acc_hi;
      p.stat_partial = c->stat_partial;
      if (lv.st != ST_F32 && o.cout > 16)
        return fail(RLN_ERR_UNSUPPORTED, "bf16 storage needs growth_rate <= 16");
      int tile = igemm_pick_tile(p.GH, p.GW);
      int th, tw;
      igemm_tile_dims(IG_DGRAD3, tile, &th, &tw);
      p.tiles_y = (p.GH + th - 1) / th;
      p.tiles_x = (p.GW + tw - 1) / tw;
      p.out_vec = ((lv.W % 4) == 0 && aligned16(p.out) && aligned16(p.S)) ? 1 : 0;
      const int ng = (o.cout <= 16 && lv.st == ST_F32) ? pack_small_level(p, lv.H, lv.W, N, &tile) : N;
      {
        const double flops = 2.0 * o.cin * o.cout * 9.0 * plane * N;
        const double bytes = 4.0 * N * plane * ((double)o.cout + 2.0 * o.cin + (double)(o.acc_hi - o.acc_lo));
        ProfScope ps(c, PC_DENSE_DGRAD, flops, bytes, s);
        if (o.cout <= 16) {
          RLN_TRY(dgrad_loop_launch(tile, p, ng, s));
        } else {
          RLN_TRY(igemm_launch(IG_DGRAD3, tile, p, N, s));
        }
      }
      if (fuse_tail) {
        tail.bn_partial = c->stat_partial;
        tail.bn_rows = igemm_stat_blocks(p, ng);
        tail.J = o.cin;
        tail.gamma = c->params + o.bn.gamma;
        tail.dgamma = c->grads + o.bn.gamma;
        tail.dbeta = c->grads + o.bn.beta;
        tail.S1 = c->S1 + so;
        tail.S2 = c->S2 + so;
      } else {
        ProfScope psb(c, PC_BN, 0, 0, s);
        RLN_TRY(bn_bwd_finalize(c->stat_partial, igemm_stat_blocks(p, ng), o.cin,
                                c->params + o.bn.gamma,
                                c->grads + o.bn.gamma, c->grads + o.bn.beta, c->S1 + so, c->S2 + so, s));
      }
    }
    WgradParams w;
    memset(&w, 0, sizeof(w));
    w.u = c->dY;
    w.u_ns = (long long)o.cout * plane;
    w.u_cs = (int)plane;
    w.Uc = o.cout;
    w.GH = lv.H;
    w.GW = lv.W;
    w.Hv = lv.H;
    w.Wv = lv.W;
    w.Vc = o.cin;
    w.v_cs = (int)plane;
    w.wsize = (long long)o.cout * o.cin * 9;
    w.m_stride = (long long)o.cin * 9;
    w.n_stride = 9;
    RLN_TRY(wg_begin(c, s, &ws));
    if (o.type == OP_DENSE && lv.st != ST_F32)
      return fail(RLN_ERR_UNSUPPORTED, "op %zu: dense block not covered by the bf16-storage backward kernels", k);
    if (o.type == OP_DENSE) {
      w.v = lv.sp(o.in_off);
      w.v_ns = (long long)lv.C * plane;
      w.pa = c->ab + o.bn.ab;
      w.pb = c->ab + c->n_ab + o.bn.ab;
      bool d3w = false;
      if (fuse_tail && c->d3_bwd_np > 0 && o.cout <= 16) {  // transposed-read 16-bit MFMA kernel where it covers the level
        D3Wgrad g;
        memset(&g, 0, sizeof(g));
        g.S = w.v;
        g.ns = w.v_ns;
        g.cs = (int)plane;
        g.H = lv.H;
        g.W = lv.W;
        g.Cin = o.cin;
        g.pa = w.pa;
        g.pb = w.pb;
        g.dY = c->dY;
        g.Cout = o.cout;
        g.N = N;
        g.partial = c->wpartial;
        g.st = lv.st;
        if (d3_wgrad_supported(g)) {
          d3_wgrad_plan(lv.H, lv.W, N, o.cin, &g);
          {
            const double wflops = 2.0 * o.cout * o.cin * 9.0 * plane * N;
            const double wbytes = (double)st_bytes(lv.st) * N * ((double)o.cout + o.cin) * plane;
            ProfScope ps(c, PC_D3_WGRAD, wflops, wbytes, s);
            RLN_TRY(d3_wgrad_launch(g, wgrad_parts(c, (long long)N * lv.H * lv.W), c->d3_bwd_dt, s));
          }
          tail.w_src = c->wpartial;
          tail.w_rows = g.nranges;
          tail.w_len = w.wsize;
          tail.w_dst = c->grads + o.conv.w;
          tail.b_src = c->bpartial;
          tail.b_rows = rows;
          tail.b_len = o.cout;
          tail.b_dst = c->grads + o.conv.b;
          ProfScope pst(c, PC_REDUCE, 0, 4.0 * (g.nranges + 1) * w.wsize, s);
          RLN_TRY(dense_tail(tail, s));
          d3w = true;
        }
      }
      if (d3w) {
      } else if (fuse_tail) {
        long long wrows = 0;
        RLN_TRY(run_wgrad(c, WG_DENSE3, w, o.cout, o.cin, o.conv.w, ws, &wrows));
        tail.w_src = c->wpartial;
        tail.w_rows = wrows;
        tail.w_len = w.wsize;
        tail.w_dst = c->grads + o.conv.w;
        tail.b_src = c->bpartial;
        tail.b_rows = rows;
        tail.b_len = o.cout;
        tail.b_dst = c->grads + o.conv.b;
        ProfScope pst(c, PC_REDUCE, 0, 4.0 * (wrows + 1) * w.wsize, s);
        RLN_TRY(dense_tail(tail, s));
      } else {
        RLN_TRY(run_wgrad(c, WG_DENSE3, w, o.cout, o.cin, o.conv.w, ws));
      }
    } else {
      bool done = false;
      if (c->d3_bwd_np > 0) {  // split-operand 16-bit MFMA kernel (fc3.h)
        F3Wgrad g;
        memset(&g, 0, sizeof(g));
        g.X = c->last_x;
        g.Cin = o.cin;
        g.H = lv.H;
        g.W = lv.W;
        g.N = N;
        g.dY = c->dY;
        g.Cout = o.cout;
        g.partial = c->wpartial;
        g.yt = lv.st;
        if (f3_wgrad_supported(g)) {
          f3_wgrad_plan(&g);
          {
            const double flops = 2.0 * o.cin * o.cout * 9.0 * plane * N;
            const double bytes = 4.0 * N * ((double)o.cin + o.cout) * plane;
            ProfScope ps(c, PC_FIRST_WGRAD, flops, bytes, ws);
            RLN_TRY(f3_wgrad_launch(g, c->d3_bwd_np, c->d3_bwd_dt, ws));
          }
          ProfScope ps2(c, PC_REDUCE, 0, 4.0 * (g.blocks + 1) * w.wsize, ws);
          RLN_TRY(reduce_rows(c->wpartial, g.blocks, w.wsize, c->grads + o.conv.w, ws));
          done = true;
        }
      }
      if (!done && lv.st != ST_F32)
        return fail(RLN_ERR_UNSUPPORTED, "first convolution not covered by the bf16-storage weight-gradient kernel");
      if (!done) {
        w.v = c->last_x;
        w.v_ns = (long long)o.cin * plane;
        RLN_TRY(run_wgrad(c, WG_RAW3, w, o.cout, o.cin, o.conv.w, ws));
      }
    }
    RLN_TRY(wg_end(c));
  } else if (o.type == OP_TD) {
    const Level& sl = c->levels[o.src_level];
    const Level& dl = c->levels[o.dst_level];
    const size_t splane = (size_t)sl.H * sl.W, dplane = (size_t)dl.H * dl.W;
    if (c->d3_bwd_np > 0 && c->p1_wb_off[k] >= 0 && dl.H == sl.H / 2 && dl.W == sl.W / 2) {
      // split-operand 16-bit MFMA kernels (pw1.h): the pooled gradient stays pooled; both kernels un-pool it on the fly
      const int64_t so = sl.stat_off + o.in_off;
      P1Dgrad q;
      memset(&q, 0, sizeof(q));
      q.dYp = c->dY;
      q.pool_idx = c->pool_idx + c->pool_off[k];
      q.Cout = o.cout;
      q.wpk = c->d3_packed + c->p1_wb_off[k];
      q.ea = c->ab + o.bn.ab;
      q.eb = c->ab + c->n_ab + o.bn.ab;
      q.egamma = c->params + o.bn.gamma;
      q.mean = c->mean + so;
      q.invstd = c->invstd + so;
      q.S = sl.sp(o.in_off);
      q.ns = (long long)sl.C * splane;
      q.cs = (int)splane;
      q.G = sl.G + (size_t)o.in_off * splane;
      q.C = o.cin;
      q.acc_lo = o.acc_lo;
      q.acc_hi = o.acc_hi;
      q.H = sl.H;
      q.W = sl.W;
      q.N = N;
      q.stat_partial = c->stat_partial;
      q.st = sl.st;
      q.yt = dl.st;
      P1Wgrad g;
      memset(&g, 0, sizeof(g));
      g.dYp = c->dY;
      g.pool_idx = q.pool_idx;
      g.Cout = o.cout;
      g.S = q.S;
      g.ns = q.ns;
      g.cs = q.cs;
      g.H = sl.H;
      g.W = sl.W;
      g.N = N;
      g.Cin = o.cin;
      g.pa = q.ea;
      g.pb = q.eb;
      g.partial = c->wpartial;
      g.st = sl.st;
      g.yt = dl.st;
      if (p1_dgrad_supported(q) && p1_wgrad_supported(g)) {
        {  // finalise the pooled gradient (BatchNorm-backward correction of the level below, Dropout2d scale)
          GradFinParams f;
          memset(&f, 0, sizeof(f));
          f.S = dl.sp(o.out_off);
          f.G = dl.G + (size_t)o.out_off * dplane;
          f.ns = (long long)dl.C * dplane;
          f.C = o.cout;
          f.H = dl.H;
          f.W = dl.W;
          const int64_t sd = dl.stat_off + o.out_off;
          f.mean = c->mean + sd;
          f.invstd = c->invstd + sd;
          f.S1 = c->S1 + sd;
          f.S2 = c->S2 + sd;
          f.invM = (float)(1.0 / ((double)N * dplane));
          f.nscale = c->masks + (size_t)N * o.drop_ch;
          f.dst = c->dY;
          f.bias_partial = c->bpartial;
          f.Hd = dl.H;
          f.Wd = dl.W;
          f.st = f.yt = dl.st;
          ProfScope ps(c, PC_GRADFIN, 0, (4.0 + 2.0 * st_bytes(dl.st)) * N * o.cout * dplane, s);
          RLN_TRY(grad_finalize(f, N, &rows, s));
        }
        p1_dgrad_plan(&q, c->d3_bwd_np);
        {
          const double flops = 2.0 * o.cin * o.cout * splane * N;
          const double bytes = 4.0 * N * (1.25 * o.cout * dplane + 3.0 * o.cin * splane);
          ProfScope ps(c, PC_TD_DGRAD, flops, bytes, s);
          RLN_TRY(p1_dgrad_launch(q, c->d3_bwd_np, c->d3_bwd_dt, s));
        }
        p1_wgrad_plan(&g);
        RLN_TRY(wg_begin(c, s, &ws));
        {
          const double flops = 2.0 * o.cin * o.cout * splane * N;
          const double bytes = 4.0 * N * (1.25 * o.cout * dplane + (double)o.cin * splane);
          ProfScope ps(c, PC_TD_WGRAD, flops, bytes, ws);
          RLN_TRY(p1_wgrad_launch(g, c->d3_bwd_np, c->d3_bwd_dt, ws));
        }
        {  // bias rows, BatchNorm-backward sums and weight slabs in one launch
          DenseTail tail;
          memset(&tail, 0, sizeof(tail));
          tail.bn_partial = c->stat_partial;
          tail.bn_rows = q.bpg;
          tail.J = o.cin;
          tail.gamma = c->params + o.bn.gamma;
          tail.dgamma = c->grads + o.bn.gamma;
          tail.dbeta = c->grads + o.bn.beta;
          tail.S1 = c->S1 + so;
          tail.S2 = c->S2 + so;
          tail.w_src = c->wpartial;
          tail.w_rows = g.nranges;
          tail.w_len = (long long)o.cout * o.cin;
          tail.w_dst = c->grads + o.conv.w;
          tail.b_src = c->bpartial;
          tail.b_rows = rows;
          tail.b_len = o.cout;
          tail.b_dst = c->grads + o.conv.b;
          ProfScope ps2(c, PC_REDUCE, 0, 4.0 * (g.nranges + 1) * tail.w_len, ws);
          RLN_TRY(dense_tail(tail, ws));
        }
        RLN_TRY(wg_end(c));
        return 0;
      }
    }
    if (sl.st != ST_F32 || dl.st != ST_F32)
      return fail(RLN_ERR_UNSUPPORTED, "op %zu: transition down not covered by the bf16-storage backward kernels", k);
    {  // pooled gradient -> pre-pool map (MaxPool2d backward) with the Dropout2d scale
      GradFinParams g;
      memset(&g, 0, sizeof(g));
      g.S = dl.sp(o.out_off);
      g.G = dl.G + (size_t)o.out_off * dplane;
      g.ns = (long long)dl.C * dplane;
      g.C = o.cout;
      g.H = dl.H;
      g.W = dl.W;
      const int64_t so = dl.stat_off + o.out_off;
      g.mean = c->mean + so;
      g.invstd = c->invstd + so;
      g.S1 = c->S1 + so;
      g.S2 = c->S2 + so;
      g.invM = (float)(1.0 / ((double)N * dplane));
      g.nscale = c->masks + (size_t)N * o.drop_ch;
      g.dst = c->dY;
      g.bias_partial = c->bpartial;
      g.pool_idx = c->pool_idx + c->pool_off[k];
      g.Hd = sl.H;
      g.Wd = sl.W;
      ProfScope ps(c, PC_GRADFIN, 0, 4.0 * N * o.cout * (splane + 2.25 * dplane), s);
      RLN_TRY(grad_finalize(g, N, &rows, s));
      RLN_TRY(reduce_rows(c->bpartial, rows, o.cout, c->grads + o.conv.b, s));
    }
    IgemmParams p;
    memset(&p, 0, sizeof(p));
    p.in = c->dY;
    p.in_ns = (long long)o.cout * splane;
    p.in_cs = (int)splane;
    p.Hin = sl.H;
    p.Win = sl.W;
    p.K = o.cout;
    p.w = c->params + o.conv.w;  // W[o][c]: j = c, k = o
    p.w_ks = o.cin;
    p.w_js = 1;
    p.tapmode = TM_ID;
    p.J = o.cin;
    p.GH = sl.H;
    p.GW = sl.W;
    p.ncls = 1;
    p.out = sl.G + (size_t)o.in_off * splane;
    p.out_ns = (long long)sl.C * splane;
    p.out_cs = (int)splane;
    p.Hout = sl.H;
    p.Wout = sl.W;
    p.S = sl.sp(o.in_off);
    p.s_ns = p.out_ns;
    const int64_t so = sl.stat_off + o.in_off;
    p.ea = c->ab + o.bn.ab;
    p.eb = c->ab + c->n_ab + o.bn.ab;
    p.emean = c->mean + so;
    p.einvstd = c->invstd + so;
    p.egamma = c->params + o.bn.gamma;
    p.acc_lo = o.acc_lo;
    p.acc_hi = o.acc_hi;
    p.stat_partial = c->stat_partial;
    const int tile = igemm_pick_tile(p.GH, p.GW);
    int th, tw;
    igemm_tile_dims(IG_DGRAD1, tile, &th, &tw);
    p.tiles_y = (p.GH + th - 1) / th;
    p.tiles_x = (p.GW + tw - 1) / tw;
    p.out_vec = ((sl.W % 4) == 0 && aligned16(p.out) && aligned16(p.S)) ? 1 : 0;
    {
      const double flops = 2.0 * o.cin * o.cout * splane * N;
      const double bytes = 4.0 * N * splane * ((double)o.cout + 3.0 * o.cin);
      ProfScope ps(c, PC_TD_DGRAD, flops, bytes, s);
      RLN_TRY(igemm_launch(IG_DGRAD1, tile, p, N, s));
    }
    RLN_TRY(bn_bwd_finalize(c->stat_partial, igemm_stat_blocks(p, N), o.cin, c->params + o.bn.gamma,
                            c->grads + o.bn.gamma, c->grads + o.bn.beta, c->S1 + so, c->S2 + so, s));
    WgradParams w;
    memset(&w, 0, sizeof(w));
    w.u = c->dY;
    w.u_ns = (long long)o.cout * splane;
    w.u_cs = (int)splane;
    w.Uc = o.cout;
    w.GH = sl.H;
    w.GW = sl.W;
    w.v = sl.sp(o.in_off);
    w.v_ns = (long long)sl.C * splane;
    w.v_cs = (int)splane;
    w.Vc = o.cin;
    w.Hv = sl.H;
    w.Wv = sl.W;
    w.pa = c->ab + o.bn.ab;
    w.pb = c->ab + c->n_ab + o.bn.ab;
    w.wsize = (long long)o.cout * o.cin;
    w.m_stride = o.cin;
    w.n_stride = 1;
    RLN_TRY(wg_begin(c, s, &ws));
    RLN_TRY(run_wgrad(c, WG_PW1, w, o.cout, o.cin, o.conv.w, ws));
    RLN_TRY(wg_end(c));
  } else {  // OP_TU
    const Level& sl = c->levels[o.src_level];
    const Level& dl = c->levels[o.dst_level];
    const size_t splane = (size_t)sl.H * sl.W, dplane = (size_t)dl.H * dl.W;
    // dU takes the destination level's storage type when the split-operand weight-gradient kernel covers the geometry;
    // at the boundary to the deep fp32 levels (input rows of < 8 pixels in octets) it is finalised as fp32 for the
    // exact-fp32 weight-gradient kernel
    int yt = dl.st;
    if (yt != ST_F32) {
      C3Wgrad probe;
      memset(&probe, 0, sizeof(probe));
      probe.X = sl.sp(o.in_off);
      probe.ns = (long long)sl.C * splane;
      probe.cs = (int)splane;
      probe.H = sl.H;
      probe.W = sl.W;
      probe.Cin = o.cin;
      probe.N = N;
      probe.dU = c->dY;
      probe.Cout = o.cout;
      probe.Ho = dl.H;
      probe.Wo = dl.W;
      probe.st = sl.st;
      probe.yt = yt;
      if (!c3_wgrad_supported(probe)) yt = ST_F32;
      if (yt == ST_F32 && sl.st != ST_F32)
        return fail(RLN_ERR_UNSUPPORTED, "op %zu: transition up not covered by the bf16-storage weight-gradient kernel", k);
    }
    RLN_TRY(finalize_grad_range(c, o.dst_level, 0, o.cout, nullptr, &rows, s, nullptr, yt));
    // (the bias rows are reduced together with the weight slabs below)
    IgemmParams p;
    memset(&p, 0, sizeof(p));
    p.in = c->dY;  // dU [N][cout][H][W]
    p.in_ns = (long long)o.cout * dplane;
    p.in_cs = (int)dplane;
    p.Hin = dl.H;
    p.Win = dl.W;
    p.K = o.cout;
    p.w = c->params + o.conv.w;  // weight[c_in][c_out][tap]: j = c_in, k = c_out
    p.w_js = (long long)o.cout * 9;
    p.w_ks = 9;
    p.tapmode = TM_ID;
    p.J = o.cin;
    p.GH = sl.H;
    p.GW = sl.W;
    p.ncls = 1;
    p.out = sl.G + (size_t)o.in_off * splane;
    p.out_ns = (long long)sl.C * splane;
    p.out_cs = (int)splane;
    p.Hout = sl.H;
    p.Wout = sl.W;
    p.cscale = c->stdv + sl.stat_off + o.in_off;
    const int tile = 0;
    int th, tw;
    igemm_tile_dims(IG_S2D3, tile, &th, &tw);
    p.tiles_y = (p.GH + th - 1) / th;
    p.tiles_x = (p.GW + tw - 1) / tw;
    p.out_vec = ((sl.W % 4) == 0 && aligned16(p.out)) ? 1 : 0;
    bool dgrad_done = false;
    if (c->d3_bwd_np > 0 && c->c3_wb_off[k] >= 0) {  // split-operand 16-bit MFMA kernel (ct3.h)
      C3Dgrad q;
      memset(&q, 0, sizeof(q));
      q.dU = c->dY;
      q.Cout = o.cout;
      q.Ho = dl.H;
      q.Wo = dl.W;
      q.wpk = c->d3_packed + c->c3_wb_off[k];
      q.cscale = p.cscale;
      q.G = p.out;
      q.ns = p.out_ns;
      q.cs = p.out_cs;
      q.H = sl.H;
      q.W = sl.W;
      q.C = o.cin;
      q.N = N;
      q.yt = yt;
      if (c3_dgrad_supported(q)) {
        c3_dgrad_plan(&q);
        const double flops = 2.0 * o.cin * o.cout * 9.0 * splane * N;
        const double bytes = 4.0 * N * ((double)o.cout * dplane + (double)o.cin * splane);
        ProfScope ps(c, PC_TU_DGRAD, flops, bytes, s);
        RLN_TRY(c3_dgrad_launch(q, c->d3_bwd_np, c->d3_bwd_dt, s));
        dgrad_done = true;
      }
    }
    if (!dgrad_done && yt != ST_F32)
      return fail(RLN_ERR_UNSUPPORTED, "op %zu: transition up not covered by the bf16-storage data-gradient kernel", k);
    if (!dgrad_done) {
      const double flops = 2.0 * o.cin * o.cout * 9.0 * splane * N;
      const double bytes = 4.0 * N * ((double)o.cout * dplane + (double)o.cin * splane);
      ProfScope ps(c, PC_TU_DGRAD, flops, bytes, s);
      RLN_TRY(igemm_launch(IG_S2D3, tile, p, N, s));
    }
    if (c->d3_bwd_np > 0) {
      C3Wgrad g;
      memset(&g, 0, sizeof(g));
      g.X = sl.sp(o.in_off);
      g.ns = (long long)sl.C * splane;
      g.cs = (int)splane;
      g.H = sl.H;
      g.W = sl.W;
      g.Cin = o.cin;
      g.N = N;
      g.dU = c->dY;
      g.Cout = o.cout;
      g.Ho = dl.H;
      g.Wo = dl.W;
      g.partial = c->wpartial;
      g.st = sl.st;
      g.yt = yt;
      if (c3_wgrad_supported(g)) {
        c3_wgrad_plan(&g);
        RLN_TRY(wg_begin(c, s, &ws));
        {
          const double flops = 2.0 * o.cin * o.cout * 9.0 * splane * N;
          const double bytes = 4.0 * N * ((double)o.cout * dplane + (double)o.cin * splane);
          ProfScope ps(c, PC_TU_WGRAD, flops, bytes, ws);
          RLN_TRY(c3_wgrad_launch(g, c->d3_bwd_np, c->d3_bwd_dt, ws));
        }
        {
          DenseTail tail;
          memset(&tail, 0, sizeof(tail));
          tail.w_src = c->wpartial;
          tail.w_rows = g.nranges;
          tail.w_len = (long long)o.cin * o.cout * 9;
          tail.w_dst = c->grads + o.conv.w;
          tail.b_src = c->bpartial;
          tail.b_rows = rows;
          tail.b_len = o.cout;
          tail.b_dst = c->grads + o.conv.b;
          ProfScope ps2(c, PC_REDUCE, 0, 4.0 * (g.nranges + 1) * tail.w_len, ws);
          RLN_TRY(dense_tail(tail, ws));
        }
        RLN_TRY(wg_end(c));
        return 0;
      }
    }
    if (sl.st != ST_F32 || yt != ST_F32)
      return fail(RLN_ERR_UNSUPPORTED, "op %zu: transition up not covered by the bf16-storage weight-gradient kernel", k);
    RLN_TRY(reduce_rows(c->bpartial, rows, o.cout, c->grads + o.conv.b, s));
    WgradParams w;
    memset(&w, 0, sizeof(w));
    w.u = sl.sp(o.in_off);  // convT input (raw)
    w.u_ns = (long long)sl.C * splane;
    w.u_cs = (int)splane;
    w.Uc = o.cin;
    w.GH = sl.H;
    w.GW = sl.W;
    w.v = c->dY;
    w.v_ns = (long long)o.cout * dplane;
    w.v_cs = (int)dplane;
    w.Vc = o.cout;
    w.Hv = dl.H;
    w.Wv = dl.W;
    w.wsize = (long long)o.cin * o.cout * 9;
    w.m_stride = 9;                       // m = c_out
    w.n_stride = (long long)o.cout * 9;   // n = c_in
    RLN_TRY(wg_begin(c, s, &ws));
    RLN_TRY(run_wgrad(c, WG_CONVT, w, o.cout, o.cin, o.conv.w, ws));
    RLN_TRY(wg_end(c));
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Backward of a whole DenseBlock (ops k0..k1, same level, same input offset) with the pull-form data gradient:
//   for j = L-1 .. 0:  finalise dY_j (kept), weight/bias gradient of layer j, and the layer's data gradient restricted
//                      to the block's NEW channels [C0, C0 + growth*j) (they must be complete before layer j-1 is
//                      finalised: read-modify-write on at most 64 channels);
//   then ONE pass over the block's input channels [0, C0) that pulls the contributions of all layers (d3_pull_k).
// Returns kNotCovered when the block is not covered (caller falls back to the per-layer path), 0 on success.
// ---------------------------------------------------------------------------------------------
constexpr int kNotCovered = 1 << 20;

int bwd_dense_block(rln_ctx* c, int k0, int k1, hipStream_t s) {
  const int L = k1 - k0 + 1;
  const Op& first = c->ops[(size_t)k0];
  const Level& lv = c->levels[first.src_level];
  const size_t plane = (size_t)lv.H * lv.W;
  const int N = c->N;
  const int C0 = first.cin;
  if (c->d3_bwd_np <= 0 || L > (int)c->dyblk.size()) return kNotCovered;
  D3Pull q;
  memset(&q, 0, sizeof(q));
  q.Cout = first.cout;
  q.mean = c->mean + lv.stat_off + first.in_off;
  q.invstd = c->invstd + lv.stat_off + first.in_off;
  q.S = lv.sp(first.in_off);
  q.G = lv.G + (size_t)first.in_off * plane;
  q.s_ns = (long long)lv.C * plane;
  q.cs = (int)plane;
  q.C = C0;
  q.H = lv.H;
  q.W = lv.W;
  q.N = N;
  q.st = lv.st;
  q.nl = std::min(L, D3_LMAX);
  for (int i = 0; i < q.nl; ++i) q.dY[i] = c->dyblk[(size_t)i];
  d3_pull_pick_tile(q.H, q.W, &q.th, &q.tw);
  q.tiles_y = (q.H + q.th - 1) / q.th;
  q.tiles_x = (q.W + q.tw - 1) / q.tw;
  if (!d3_pull_supported(q, c->d3_bwd_np) || first.cout > 16) return kNotCovered;
  for (int j = 0; j < L; ++j)
    if (c->d3_wb_off[(size_t)k0 + j] < 0) return kNotCovered;

  // The block's own new channels in pull form too (16-channel groups on the packed-weight grid): group j = the output
  // of layer j is consumed by the layers i > j of the block; once those are finalised, ONE launch pulls their
  // contributions into G[group j] (S read once, G written once, nothing read-modify-written per layer), and only then is
  // dY_j finalised.  Falls back to the per-layer exact-fp32 kernel when the channel grid does not line up.
  const Op& last_op = c->ops[(size_t)k1];
  const int gr = first.cout;
  // Measured on MI355X (batch 64): the one-group launches are latency-bound (30 tiles per block, each paying a full load
  // round trip: 113 us per launch whatever the layer count) and the form loses 1.1 ms per step against the per-layer
  // kernel; it stays available to diagnostic builds (RLN_PULL_NEW).
  const bool pull_new = gr == 16 && (C0 % 16) == 0 && L - 1 <= D3_LMAX && rln_env("RLN_PULL_NEW") != nullptr;
  auto pull_range = [&](int ch_lo, int C, int i_lo, int i_hi, int acc_lo, int acc_hi, long long* defer_rows) -> int {
    // input channels [ch_lo, ch_lo + C) of the block (relative to its first input channel) from layers i_hi .. i_lo
    D3Pull g = q;
    D3PullFin f;
    memset(&f, 0, sizeof(f));
    const int nl = i_hi - i_lo + 1;
    g.nl = nl;
    g.C = C;
    g.S = lv.sp(first.in_off + ch_lo);
    g.G = lv.G + (size_t)(first.in_off + ch_lo) * plane;
    g.mean = c->mean + lv.stat_off + first.in_off + ch_lo;
    g.invstd = c->invstd + lv.stat_off + first.in_off + ch_lo;
    for (int i = 0; i < nl; ++i) {
      const int j = i_hi - i;
      const Op& o = c->ops[(size_t)k0 + j];
      g.dY[i] = c->dyblk[(size_t)j];
      g.wpk[i] = c->d3_packed + c->d3_wb_off[(size_t)k0 + j] + (long long)(ch_lo / 16) * 5 * c->d3_bwd_np * 64;
      g.ea[i] = c->ab + o.bn.ab + ch_lo;
      g.eb[i] = c->ab + c->n_ab + o.bn.ab + ch_lo;
      g.egamma[i] = c->params + o.bn.gamma + ch_lo;
      f.gamma[i] = c->params + o.bn.gamma + ch_lo;
      f.dgamma[i] = c->grads + o.bn.gamma + ch_lo;
      f.dbeta[i] = c->grads + o.bn.beta + ch_lo;
    }
    g.acc_lo = acc_lo;
    g.acc_hi = acc_hi;
    g.stat_partial = c->stat_partial;
    if (!d3_pull_supported(g, c->d3_bwd_np)) return fail(RLN_ERR_UNSUPPORTED, "pull launch not covered");
    {
      const double flops = 2.0 * C * first.cout * 9.0 * plane * N * nl;
      const double eb = (double)st_bytes(lv.st);  // dY and S in the level's storage type, G in fp32
      const double bytes = (double)N * plane * (eb * nl * first.cout + (eb + 4.0) * C + 4.0 * (acc_hi - acc_lo));
      ProfScope ps(c, PC_D3_PULL, flops, bytes, s);
      RLN_TRY(d3_pull_launch(g, c->d3_bwd_np, c->d3_bwd_dt, s));
    }
    f.nl = nl;
    f.C = C;
    f.Cpad = ((C + 15) / 16) * 16;
    f.rows = d3_pull_nsub(g) * d3_pull_blocks(g);
    if (defer_rows) {  // one layer, C a multiple of 16: the partial rows have dense_tail's [rows][C][2] layout
      *defer_rows = f.rows;
      return 0;
    }
    f.partial = c->stat_partial;
    f.S1 = c->S1 + lv.stat_off + first.in_off + ch_lo;
    f.S2 = c->S2 + lv.stat_off + first.in_off + ch_lo;
    ProfScope psb(c, PC_BN, 0, 0, s);
    RLN_TRY(d3_pull_finalize(f, s));
    return 0;
  };
  // The per-layer data gradient of the block's own channels as a one-layer pull launch (16-bit MFMA family instead of
  // the exact-fp32 kernel, which is matrix-pipe bound: profiles/r03_pmc_sq_f32.json, dgrad_loop_k: 49 % of the MFMA
  // cycles at 1/16 of the 16-bit rate).  Measured on MI355X, batch 64: 25.0 -> 25.4 ms per step -- a pull launch over
  // few channel groups pays a load round trip per tile (30 tiles per block), and prefetching the next tile's dY across
  // the work items does not hide it (vmcnt retires in order: the items' own S loads wait for the older prefetch too;
  // 25.6 ms).  Kept for diagnostic builds (RLN_LAYER_PULL).
  const bool layer_pull = gr == 16 && (C0 % 16) == 0 && rln_env("RLN_LAYER_PULL") != nullptr;
  // The per-layer data gradient of the block's own new channels on the 16-bit pipe (d3_dgl_k) instead of the exact-fp32
  // dgrad_loop_k, which is matrix-pipe bound at 1/16 of the 16-bit rate (profiles/r03_pmc_sq_f32.json: 50 % MFMA busy).
  const bool dgl16 = c->d3_bwd_np > 0 && c->d3_bwd_dt == D3_BF16 && first.cout <= 16 && (C0 % 16) == 0 &&
                     rln_env("RLN_NO_DGL") == nullptr;
  D3Dgl dg;

  // Two-layer weight-gradient launches (dense3.h: D3Wgrad.nl): layer j+1 is deferred until layer j's dY is finalised,
  // then ONE launch loads every z chunk once for both.  pend_j = the deferred layer (or -1).
  int pend_j = -1;
  static const bool no_pair = rln_env("RLN_NO_WG_PAIR") != nullptr;
  for (int j = L - 1; j >= 0; --j) {
    const Op& o = c->ops[(size_t)k0 + j];
    long long rows = 0;
    const float* nscale = (o.drop_ch >= 0) ? (c->masks + (size_t)N * o.drop_ch) : nullptr;
    float* dYj = c->dyblk[(size_t)j];
    if (pull_new && j < L - 1) {  // G[group j] += contributions of layers j+1 .. L-1 (all finalised by now)
      const int lo = C0 + gr * j;
      const int alo = std::min(std::max(last_op.acc_lo - lo, 0), gr), ahi = std::min(std::max(last_op.acc_hi - lo, 0), gr);
      RLN_TRY(pull_range(lo, gr, j + 1, L - 1, alo, ahi, nullptr));
    }
    // the dense weight gradient's one-part bf16 operand is rounded here, once, into a 2-byte copy (dense3.h: dY16)
    static const bool no_dy16 = rln_env("RLN_NO_DY16") != nullptr;
    const bool use_dy16 = !no_dy16 && lv.st == ST_F32 && c->d3_bwd_np > 0 && c->d3_bwd_dt == D3_BF16 && o.cout <= 16 &&
                          wgrad_parts(c, (long long)N * lv.H * lv.W) == 1 && (lv.W % 40) == 0 && c->dy16 != nullptr &&
                          ((lv.H * lv.W) % 8) == 0;
    void* dy16_j = use_dy16 ? static_cast<unsigned char*>(c->dy16) + (size_t)(j & 1) * c->dy16_half : nullptr;
    RLN_TRY(finalize_grad_range(c, o.dst_level, o.out_off, o.cout, nscale, &rows, s, dYj, -1, dy16_j));
    // the layer's three small reductions (bias rows, weight slabs, BatchNorm-backward sums of its new-channel data
    // gradient) run as ONE launch at the end of the iteration when the weight gradient went through d3_wgrad_k
    DenseTail tail;
    memset(&tail, 0, sizeof(tail));
    bool use_tail = false;
    {  // weight gradient of layer j
      WgradParams w;
      memset(&w, 0, sizeof(w));
      w.u = dYj;
      w.u_ns = (long long)o.cout * plane;
      w.u_cs = (int)plane;
      w.Uc = o.cout;
      w.GH = lv.H;
      w.GW = lv.W;
      w.Hv = lv.H;
      w.Wv = lv.W;
      w.Vc = o.cin;
      w.v_cs = (int)plane;
      w.wsize = (long long)o.cout * o.cin * 9;
      w.m_stride = (long long)o.cin * 9;
      w.n_stride = 9;
      w.v = lv.sp(o.in_off);
      w.v_ns = (long long)lv.C * plane;
      w.pa = c->ab + o.bn.ab;
      w.pb = c->ab + c->n_ab + o.bn.ab;
      D3Wgrad g;
      memset(&g, 0, sizeof(g));
      g.S = w.v;
      g.ns = w.v_ns;
      g.cs = (int)plane;
      g.H = lv.H;
      g.W = lv.W;
      g.Cin = o.cin;
      g.pa = w.pa;
      g.pb = w.pb;
      g.dY = dYj;
      g.Cout = o.cout;
      g.N = N;
      g.partial = c->wpartial;
      g.st = lv.st;
      if (lv.st != ST_F32 && !d3_wgrad_supported(g))
        return fail(RLN_ERR_UNSUPPORTED, "dense weight gradient not covered by the bf16-storage kernel");
      if (d3_wgrad_supported(g)) {  // transposed-read 16-bit MFMA kernel
        d3_wgrad_plan(lv.H, lv.W, N, o.cin, &g);
        static const bool no_z8 = rln_env("RLN_NO_WG_Z8") != nullptr;
        if (use_dy16 && (g.tw % 8) == 0) {
          g.dY16 = dy16_j;
          g.yt = ST_BF16;
        } else if (!no_z8 && lv.st == ST_BF16 && (g.tw % 8) == 0 && (lv.W % 8) == 0 && o.cout <= 16 &&
                   wgrad_parts(c, (long long)N * lv.H * lv.W) == 1 && c->d3_bwd_dt == D3_BF16) {
          g.z8 = 1;  // bf16 stacks: 8-pixel units; the bf16 dY buffer itself is the operand
          g.dY16 = dYj;
          g.yt = ST_BF16;
        }
#ifdef RLN_DIAG
        if (rln_env("RLN_D3_DBG")) g.dbg = atoi(rln_env("RLN_D3_DBG"));
#endif
        const bool pairable = !no_pair && g.yt == ST_BF16 && (o.cin % 16) == 0 && gr == 16 && o.cout == 16 &&
                              c->wpartial2 != nullptr;
        use_tail = true;
        tail.w_len = 0;  // no weight slabs in this iteration unless set below
        if (pairable && pend_j < 0 && j >= 1 && (c->ops[(size_t)k0 + j - 1].cin % 16) == 0) {
          pend_j = j;  // launched together with layer j-1
        } else if (pairable && pend_j == j + 1) {
          const Op& ob = c->ops[(size_t)k0 + j + 1];  // primary: the deferred layer (16 more input channels)
          D3Wgrad g2 = g;
          d3_wgrad_plan(lv.H, lv.W, N, ob.cin, &g2);
          g2.Cin = ob.cin;
          g2.pa = c->ab + ob.bn.ab;
          g2.pb = c->ab + c->n_ab + ob.bn.ab;
          g2.dY = c->dyblk[(size_t)j + 1];
          g2.dY16 = g.z8 ? static_cast<void*>(c->dyblk[(size_t)j + 1])
                         : static_cast<void*>(static_cast<unsigned char*>(c->dy16) + (size_t)((j + 1) & 1) * c->dy16_half);
          g2.partial = c->wpartial;
          g2.nl = 2;
          g2.Cin2 = o.cin;
          g2.pa2 = g.pa;
          g2.pb2 = g.pb;
          g2.dY16_2 = g.dY16;
          g2.partial2 = c->wpartial2;
          {
            const double wflops = 2.0 * o.cout * ((double)o.cin + ob.cin) * 9.0 * plane * N;
            const double wbytes = (double)st_bytes(lv.st) * N * (2.0 * o.cout + o.cin + ob.cin) * plane;
            ProfScope ps(c, PC_D3_WGRAD, wflops, wbytes, s);
            RLN_TRY(d3_wgrad_launch(g2, 1, c->d3_bwd_dt, s));
          }
          {  // the deferred layer's slabs
            const long long wsize_b = (long long)ob.cout * ob.cin * 9;
            ProfScope ps(c, PC_REDUCE, 0, 4.0 * (g2.nranges + 1) * wsize_b, s);
            RLN_TRY(reduce_rows(c->wpartial, g2.nranges, wsize_b, c->grads + ob.conv.w, s));
          }
          pend_j = -1;
          tail.w_src = c->wpartial2;
          tail.w_rows = g2.nranges;
          tail.w_len = w.wsize;
          tail.w_dst = c->grads + o.conv.w;
        } else {
          if (pend_j >= 0) return fail(RLN_ERR_STATE, "a deferred dense weight gradient was left behind (layer %d)", pend_j);
          const double wflops = 2.0 * o.cout * o.cin * 9.0 * plane * N;
          const double wbytes = (double)st_bytes(lv.st) * N * ((double)o.cout + o.cin) * plane;
          ProfScope ps(c, PC_D3_WGRAD, wflops, wbytes, s);
          RLN_TRY(d3_wgrad_launch(g, wgrad_parts(c, (long long)N * lv.H * lv.W), c->d3_bwd_dt, s));
          tail.w_src = c->wpartial;
          tail.w_rows = g.nranges;
          tail.w_len = w.wsize;
          tail.w_dst = c->grads + o.conv.w;
        }
        tail.b_src = c->bpartial;
        tail.b_rows = rows;
        tail.b_len = o.cout;
        tail.b_dst = c->grads + o.conv.b;
      } else {
        if (pend_j >= 0) return fail(RLN_ERR_STATE, "a deferred dense weight gradient was left behind (layer %d)", pend_j);
        RLN_TRY(reduce_rows(c->bpartial, rows, o.cout, c->grads + o.conv.b, s));
        RLN_TRY(run_wgrad(c, WG_DENSE3, w, o.cout, o.cin, o.conv.w, s));
      }
    }
    const int Jn = pull_new ? 0 : o.cin - C0;  // new channels this layer consumes (per-layer form only)
    bool use_dgl = false;
    if (Jn > 0 && dgl16 && !(layer_pull && use_tail)) {  // does the 16-bit looped form cover this geometry?
      memset(&dg, 0, sizeof(dg));
      dg.dY = dYj;
      dg.K = o.cout;
      dg.wpk = c->d3_packed + c->d3_wb_off[(size_t)k0 + j] + (long long)(C0 / 16) * 5 * c->d3_bwd_np * 64;
      dg.J = Jn;
      dg.S = lv.sp(o.in_off + C0);
      dg.s_ns = (long long)lv.C * plane;
      dg.cs = (int)plane;
      dg.G = lv.G + (size_t)(o.in_off + C0) * plane;
      dg.ea = c->ab + o.bn.ab + C0;
      dg.eb = c->ab + c->n_ab + o.bn.ab + C0;
      dg.emean = c->mean + lv.stat_off + o.in_off + C0;
      dg.einvstd = c->invstd + lv.stat_off + o.in_off + C0;
      dg.egamma = c->params + o.bn.gamma + C0;
      dg.acc_lo = std::max(0, o.acc_lo - C0);
      dg.acc_hi = std::max(0, o.acc_hi - C0);
      dg.H = lv.H;
      dg.W = lv.W;
      dg.N = N;
      dg.stat_partial = c->stat_partial;
      dg.st = lv.st;
      use_dgl = d3_dgl_supported(dg);
    }
    if (Jn > 0 && layer_pull && use_tail) {
      long long prow = 0;
      const int64_t so = lv.stat_off + o.in_off + C0;
      RLN_TRY(pull_range(C0, Jn, j, j, std::max(0, o.acc_lo - C0), std::max(0, o.acc_hi - C0), &prow));
      tail.bn_partial = c->stat_partial;
      tail.bn_rows = prow;
      tail.J = Jn;
      tail.gamma = c->params + o.bn.gamma + C0;
      tail.dgamma = c->grads + o.bn.gamma + C0;
      tail.dbeta = c->grads + o.bn.beta + C0;
      tail.S1 = c->S1 + so;
      tail.S2 = c->S2 + so;
    } else if (use_dgl) {
      // 16-bit MFMA form of the per-layer data gradient into the block's own new channels (dense3.h: D3Dgl)
      const int64_t so = lv.stat_off + o.in_off + C0;
      d3_dgl_plan(lv.H, lv.W, &dg);
      {
        const double flops = 2.0 * Jn * o.cout * 9.0 * plane * N;
        const double eb = (double)st_bytes(lv.st);
        const double bytes = (double)N * plane * (eb * o.cout + (eb + 4.0) * Jn + 4.0 * (dg.acc_hi - dg.acc_lo));
        ProfScope ps(c, PC_DENSE_DGRAD, flops, bytes, s);
        RLN_TRY(d3_dgl_launch(dg, c->d3_bwd_np, c->d3_bwd_dt, s));
      }
      if (use_tail) {
        tail.bn_partial = c->stat_partial;
        tail.bn_rows = d3_dgl_rows(dg);
        tail.J = Jn;
        tail.gamma = c->params + o.bn.gamma + C0;
        tail.dgamma = c->grads + o.bn.gamma + C0;
        tail.dbeta = c->grads + o.bn.beta + C0;
        tail.S1 = c->S1 + so;
        tail.S2 = c->S2 + so;
      } else {
        ProfScope psb(c, PC_BN, 0, 0, s);
        RLN_TRY(bn_bwd_finalize(c->stat_partial, d3_dgl_rows(dg), Jn, c->params + o.bn.gamma + C0,
                                c->grads + o.bn.gamma + C0, c->grads + o.bn.beta + C0, c->S1 + so, c->S2 + so, s));
      }
    } else if (Jn > 0) {
      IgemmParams p;
      memset(&p, 0, sizeof(p));
      p.in = dYj;
      p.in_ns = (long long)o.cout * plane;
      p.in_cs = (int)plane;
      p.Hin = lv.H;
      p.Win = lv.W;
      p.K = o.cout;
      p.w = c->params + o.conv.w + (long long)C0 * 9;  // W[o][c][tap], c offset C0
      p.w_ks = (long long)o.cin * 9;
      p.w_js = 9;
      p.tapmode = TM_FLIP;
      p.J = Jn;
      p.GH = lv.H;
      p.GW = lv.W;
      p.ncls = 1;
      p.out = lv.G + (size_t)(o.in_off + C0) * plane;
      p.out_ns = (long long)lv.C * plane;
      p.out_cs = (int)plane;
      p.Hout = lv.H;
      p.Wout = lv.W;
      p.S = lv.sp((o.in_off + C0));
      p.s_ns = p.out_ns;
      p.st = lv.st;
      const int64_t so = lv.stat_off + o.in_off + C0;
      p.ea = c->ab + o.bn.ab + C0;
      p.eb = c->ab + c->n_ab + o.bn.ab + C0;
      p.emean = c->mean + so;
      p.einvstd = c->invstd + so;
      p.egamma = c->params + o.bn.gamma + C0;
      p.acc_lo = std::max(0, o.acc_lo - C0);
      p.acc_hi = std::max(0, o.acc_hi - C0);
      p.stat_partial = c->stat_partial;
      int tile = igemm_pick_tile(p.GH, p.GW);
      int th, tw;
      igemm_tile_dims(IG_DGRAD3, tile, &th, &tw);
      p.tiles_y = (p.GH + th - 1) / th;
      p.tiles_x = (p.GW + tw - 1) / tw;
      p.out_vec = ((lv.W % 4) == 0 && aligned16(p.out) && aligned16(p.S)) ? 1 : 0;
      const int ng = (o.cout <= 16 && lv.st == ST_F32) ? pack_small_level(p, lv.H, lv.W, N, &tile) : N;
      {
        const double flops = 2.0 * Jn * o.cout * 9.0 * plane * N;
        const double eb = (double)st_bytes(lv.st);  // dY and S in the level's storage type, G in fp32
        const double bytes = (double)N * plane * (eb * o.cout + (eb + 4.0) * Jn + 4.0 * (p.acc_hi - p.acc_lo));
        ProfScope ps(c, PC_DENSE_DGRAD, flops, bytes, s);
        if (o.cout <= 16) {
          RLN_TRY(dgrad_loop_launch(tile, p, ng, s));
        } else {
          RLN_TRY(igemm_launch(IG_DGRAD3, tile, p, N, s));
        }
      }
      if (use_tail) {
        tail.bn_partial = c->stat_partial;
        tail.bn_rows = igemm_stat_blocks(p, ng);
        tail.J = Jn;
        tail.gamma = c->params + o.bn.gamma + C0;
        tail.dgamma = c->grads + o.bn.gamma + C0;
        tail.dbeta = c->grads + o.bn.beta + C0;
        tail.S1 = c->S1 + so;
        tail.S2 = c->S2 + so;
      } else {
        ProfScope psb(c, PC_BN, 0, 0, s);
        RLN_TRY(bn_bwd_finalize(c->stat_partial, igemm_stat_blocks(p, ng), Jn, c->params + o.bn.gamma + C0,
                                c->grads + o.bn.gamma + C0, c->grads + o.bn.beta + C0, c->S1 + so, c->S2 + so, s));
      }
    }
    if (use_tail) {
      ProfScope pst(c, PC_REDUCE, 0, 4.0 * (tail.w_rows + 1) * tail.w_len, s);
      RLN_TRY(dense_tail(tail, s));
    }
  }
  if (pend_j >= 0) return fail(RLN_ERR_STATE, "a deferred dense weight gradient was left behind (layer %d)", pend_j);
  // input channels [0, C0): all layers at once (passes of at most D3_LMAX layers)
  const Op& last = c->ops[(size_t)k1];
  for (int j0 = L - 1, pass = 0; j0 >= 0; j0 -= D3_LMAX, ++pass) {
    const int nl = std::min(D3_LMAX, j0 + 1);
    q.nl = nl;
    D3PullFin f;
    memset(&f, 0, sizeof(f));
    for (int i = 0; i < nl; ++i) {
      const int j = j0 - i;
      const Op& o = c->ops[(size_t)k0 + j];
      q.dY[i] = c->dyblk[(size_t)j];
      q.wpk[i] = c->d3_packed + c->d3_wb_off[(size_t)k0 + j];
      q.ea[i] = c->ab + o.bn.ab;
      q.eb[i] = c->ab + c->n_ab + o.bn.ab;
      q.egamma[i] = c->params + o.bn.gamma;
      f.gamma[i] = c->params + o.bn.gamma;
      f.dgamma[i] = c->grads + o.bn.gamma;
      f.dbeta[i] = c->grads + o.bn.beta;
    }
    if (pass == 0) {  // channels that held gradient before the block's backward started
      q.acc_lo = std::min(last.acc_lo, C0);
      q.acc_hi = std::min(last.acc_hi, C0);
    } else {
      q.acc_lo = 0;
      q.acc_hi = C0;
    }
    q.stat_partial = c->stat_partial;
    {
      double flops = 0.0;
      for (int i = 0; i < nl; ++i) flops += 2.0 * C0 * first.cout * 9.0 * plane * N;
      const double eb = (double)st_bytes(lv.st);  // dY and S in the level's storage type, G in fp32
      const double bytes = (double)N * plane * (eb * nl * first.cout + (eb + 4.0) * C0 + 4.0 * (q.acc_hi - q.acc_lo));
      ProfScope ps(c, PC_D3_PULL, flops, bytes, s);
      RLN_TRY(d3_pull_launch(q, c->d3_bwd_np, c->d3_bwd_dt, s));
    }
    f.nl = nl;
    f.C = C0;
    f.Cpad = ((C0 + 15) / 16) * 16;
    f.rows = d3_pull_nsub(q) * d3_pull_blocks(q);
    f.partial = c->stat_partial;
    f.S1 = c->S1 + lv.stat_off + first.in_off;
    f.S2 = c->S2 + lv.stat_off + first.in_off;
    ProfScope psb(c, PC_BN, 0, 0, s);
    RLN_TRY(d3_pull_finalize(f, s));
  }
  return 0;
}

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================

extern "C" {

const char* rln_last_error(void) { return g_err; }
int rln_version(void) { return 1; }

int rln_create(const rln_config* cfg, rln_ctx** out) {
  if (!cfg || !out) return fail(RLN_ERR_ARG, "null argument");
  rln_ctx* c = new rln_ctx();
  c->cfg = *cfg;
  const int e = build_plan(c);
  if (e != 0) {
    delete c;
    return e;
  }
  *out = c;
  return 0;
}

void rln_destroy(rln_ctx* ctx) {
  if (!ctx) return;
  if (ctx->side) {
    (void)hipStreamSynchronize(ctx->side);
    for (int b = 0; b < 2; ++b) {
      if (ctx->ev_dy[b]) (void)hipEventDestroy(ctx->ev_dy[b]);
      if (ctx->ev_wg[b]) (void)hipEventDestroy(ctx->ev_wg[b]);
    }
    (void)hipStreamDestroy(ctx->side);
  }
  delete ctx;
}

int rln_num_tensors(const rln_ctx* c) { return (int)c->tensors.size(); }
int64_t rln_param_count(const rln_ctx* c) { return c->n_param; }
int64_t rln_bnstat_count(const rln_ctx* c) { return c->n_bnstat; }
int64_t rln_nbt_count(const rln_ctx* c) { return c->n_nbt; }
int rln_feature_channels(const rln_ctx* c) { return c->feat_C; }
int rln_num_dropouts(const rln_ctx* c) { return (int)c->drop_per_call.size(); }
int64_t rln_dropout_channels(const rln_ctx* c, int* per_call) {
  if (per_call)
    for (size_t i = 0; i < c->drop_per_call.size(); ++i) per_call[i] = c->drop_per_call[i];
  return c->drop_total;
}

int rln_tensor_info(const rln_ctx* c, int idx, char* name, int name_cap, int* kind, int64_t* offset, int* ndim,
                    int64_t shape[4]) {
  if (idx < 0 || idx >= (int)c->tensors.size()) return fail(RLN_ERR_ARG, "tensor index %d out of range", idx);
  const TensorInfo& t = c->tensors[idx];
  if (name && name_cap > 0) {
    strncpy(name, t.name.c_str(), name_cap - 1);
    name[name_cap - 1] = 0;
  }
  if (kind) *kind = t.kind;
  if (offset) *offset = t.offset;
  if (ndim) *ndim = t.ndim;
  if (shape)
    for (int i = 0; i < 4; ++i) shape[i] = t.shape[i];
  return 0;
}

int rln_set_dense_arith(rln_ctx* c, int fwd_parts, int fwd_dtype, int bwd_parts, int bwd_dtype) {
  c->eval_tables_valid = 0;
  auto bad = [](int np, int dt) { return np < 0 || np > 3 || dt < 0 || dt > 1 || (dt == 1 && np == 3); };
  if (bad(fwd_parts, fwd_dtype) || bad(bwd_parts, bwd_dtype))
    return fail(RLN_ERR_ARG, "parts in 0..3 (f16: 0..2), dtype 0 (bf16) or 1 (f16)");
  if (c->storage == 1 && (fwd_parts != 1 || bwd_parts != 1 || fwd_dtype != D3_BF16 || bwd_dtype != D3_BF16))
    return fail(RLN_ERR_ARG, "bf16 storage runs on one-part bf16 operands (rln_set_storage(ctx, 0) first)");
  c->d3_fwd_np = fwd_parts;
  c->d3_fwd_dt = fwd_dtype;
  c->d3_bwd_np = bwd_parts;
  c->d3_bwd_dt = bwd_dtype;
  c->wg_parts = bwd_parts == 2 ? 1 : bwd_parts;  // see rln_set_wgrad_parts
  c->levels[0].S = nullptr;  // the workspace layout depends on the mode: it has to be set again
  c->N = c->H = c->W = 0;
  return 0;
}

int rln_set_storage(rln_ctx* c, int mode) {
  c->eval_tables_valid = 0;
  if (mode != 0 && mode != 1) return fail(RLN_ERR_ARG, "storage mode 0 (fp32 stacks) or 1 (bf16 stacks)");
  c->storage = mode;
  if (mode == 1) {  // bf16 stacks are read as plain bf16 MFMA operands everywhere (fp32 accumulation and statistics)
    c->d3_fwd_np = c->d3_bwd_np = c->wg_parts = 1;
    c->d3_fwd_dt = c->d3_bwd_dt = D3_BF16;
  }
  c->levels[0].S = nullptr;  // the workspace layout depends on the mode: it has to be set again
  c->N = c->H = c->W = 0;
  return 0;
}
int rln_get_storage(const rln_ctx* c) { return c->storage; }
int rln_get_wgrad_parts(const rln_ctx* c) { return c->wg_parts; }

int rln_set_wgrad_parts(rln_ctx* c, int parts) {
  if (parts < 0 || parts > 3) return fail(RLN_ERR_ARG, "parts in 0..3 (0 = as many as the backward arithmetic)");
  c->wg_parts = (parts == 0 || parts > c->d3_bwd_np) ? c->d3_bwd_np : parts;
  return 0;
}

int rln_bind_params(rln_ctx* c, float* params, float* grads, float* bn_running, int64_t* nbt) {
  c->eval_tables_valid = 0;
  if (!params || !bn_running) return fail(RLN_ERR_ARG, "params and bn_running are required");
  c->params = params;
  c->grads = grads;
  c->bnrun = bn_running;
  c->nbt = nbt;
  return 0;
}

size_t rln_workspace_bytes(const rln_ctx* c, int n, int h, int w, int with_backward) {
  if (n < 1 || h < 1 || w < 1 || (h >> c->cfg.n_down) < 1 || (w >> c->cfg.n_down) < 1) return 0;  // rln_set_workspace reports
  return carve(const_cast<rln_ctx*>(c), nullptr, n, h, w, with_backward, false);
}

int rln_set_workspace(rln_ctx* c, void* ws, size_t bytes, int n, int h, int w, int with_backward) {
  c->eval_tables_valid = 0;
  if (n < 1 || h < 1 || w < 1) return fail(RLN_ERR_ARG, "bad geometry %dx%dx%d", n, h, w);
  if ((h >> c->cfg.n_down) < 1 || (w >> c->cfg.n_down) < 1)
    return fail(RLN_ERR_ARG, "input %dx%d too small for %d poolings (Output size is too small)", h, w, c->cfg.n_down);
  const size_t need = rln_workspace_bytes(c, n, h, w, with_backward);
  if (!ws || bytes < need) return fail(RLN_ERR_WORKSPACE, "workspace of %zu bytes needed, got %zu", need, bytes);
  if (((uintptr_t)ws) & 255) return fail(RLN_ERR_WORKSPACE, "workspace must be 256-byte aligned");
  // The block usually reuses the memory of the previous workspace (the caching allocator hands a freed block straight
  // back), and the descriptor tables below are written with synchronous host copies that are not ordered after the
  // caller's stream: kernels of the previous geometry still in flight would see their activations overwritten -- a
  // batch-64 eval forward differed from its halves by 1e-4 once in ~15 fresh-box runs (cold GPU, host far ahead).  A
  // set-up call, once per input geometry: drain the device first.
  {
    const hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return fail((int)e, "hipDeviceSynchronize failed");
  }
  carve(c, ws, n, h, w, with_backward, true);
  if (!c->d3_desc_f.empty() || !c->d3_desc_b.empty()) {  // descriptor tables of the weight-pack kernel (setup time)
    hipError_t e = hipSuccess;
    if (!c->d3_desc_f.empty())
      e = hipMemcpy(c->d3_desc_f_dev, c->d3_desc_f.data(), c->d3_desc_f.size() * sizeof(D3PackDesc),
                    hipMemcpyHostToDevice);
    if (e == hipSuccess && !c->d3_desc_b.empty())
      e = hipMemcpy(c->d3_desc_b_dev, c->d3_desc_b.data(), c->d3_desc_b.size() * sizeof(D3PackDesc),
                    hipMemcpyHostToDevice);
    if (e != hipSuccess) return fail((int)e, "descriptor upload failed");
  }
  if (!c->c3_desc_f.empty() || !c->c3_desc_b.empty()) {
    hipError_t e = hipSuccess;
    if (!c->c3_desc_f.empty())
      e = hipMemcpy(c->c3_desc_f_dev, c->c3_desc_f.data(), c->c3_desc_f.size() * sizeof(C3PackDesc),
                    hipMemcpyHostToDevice);
    if (e == hipSuccess && !c->c3_desc_b.empty())
      e = hipMemcpy(c->c3_desc_b_dev, c->c3_desc_b.data(), c->c3_desc_b.size() * sizeof(C3PackDesc),
                    hipMemcpyHostToDevice);
    if (e != hipSuccess) return fail((int)e, "descriptor upload failed");
  }
  if (!c->p1_desc_f.empty() || !c->p1_desc_b.empty()) {
    hipError_t e = hipSuccess;
    if (!c->p1_desc_f.empty())
      e = hipMemcpy(c->p1_desc_f_dev, c->p1_desc_f.data(), c->p1_desc_f.size() * sizeof(P1PackDesc),
                    hipMemcpyHostToDevice);
    if (e == hipSuccess && !c->p1_desc_b.empty())
      e = hipMemcpy(c->p1_desc_b_dev, c->p1_desc_b.data(), c->p1_desc_b.size() * sizeof(P1PackDesc),
                    hipMemcpyHostToDevice);
    if (e != hipSuccess) return fail((int)e, "descriptor upload failed");
  }
  c->N = n;
  c->H = h;
  c->W = w;
  c->with_bwd = with_backward;
  c->have_train_fwd = 0;
  c->have_loss = 0;
  return 0;
}

int rln_forward(rln_ctx* c, const float* x, int n, int h, int w, int training, const float* drop_scales, uint64_t seed,
                float* probs_out, float* feat_out, int use_softmax, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!c->params) return fail(RLN_ERR_STATE, "rln_bind_params not called");
  if (n != c->N || h != c->H || w != c->W || !c->levels[0].S)
    return fail(RLN_ERR_WORKSPACE, "workspace not set for geometry %dx%dx%d", n, h, w);
  if (!x) return fail(RLN_ERR_ARG, "x is null");
  if (training) {
    const size_t cnt = (size_t)n * c->drop_total;
    if (drop_scales) {
      hipError_t e = hipMemcpyAsync(c->masks, drop_scales, cnt * sizeof(float), hipMemcpyDeviceToDevice, s);
      if (e != hipSuccess) return fail((int)e, "mask copy failed");
    } else {
      RLN_TRY(dropout_scales(c->masks, (long long)cnt, 1.0f - c->cfg.drop_p, (unsigned long long)seed, s));
    }
    if (c->nbt) RLN_TRY(add_one_i64((long long*)c->nbt, c->n_nbt, s));
  }
  c->prep_done = -1;
  c->pair_finish = -1;
  // frozen-model loops (rln_set_eval_cache): the forward weight fragments and the folded BatchNorm tables of the
  // previous eval forward are still those of the arena
  c->eval_reuse = (!training && c->eval_cache_on && c->eval_tables_valid) ? 1 : 0;
  const bool pack_fwd = !c->eval_reuse;
  if (training) c->eval_tables_valid = 0;  // batch-statistics tables replace the eval ones
  if (c->d3_units_f > 0 && pack_fwd)
    RLN_TRY(d3_pack_weights(c->params, c->d3_desc_f_dev, (int)c->d3_desc_f.size(), c->d3_units_f, c->d3_packed,
                            c->d3_fwd_np, c->d3_fwd_dt, s));
  if (c->d3_units_b > 0 && training && c->with_bwd)
    RLN_TRY(d3_pack_weights(c->params, c->d3_desc_b_dev, (int)c->d3_desc_b.size(), c->d3_units_b, c->d3_packed,
                            c->d3_bwd_np, c->d3_bwd_dt, s));
  if (c->c3_units_f > 0 && pack_fwd)
    RLN_TRY(c3_pack_weights(c->params, c->c3_desc_f_dev, (int)c->c3_desc_f.size(), c->c3_units_f, c->d3_packed,
                            c->d3_fwd_np, c->d3_fwd_dt, s));
  if (c->c3_units_b > 0 && training && c->with_bwd)
    RLN_TRY(c3_pack_weights(c->params, c->c3_desc_b_dev, (int)c->c3_desc_b.size(), c->c3_units_b, c->d3_packed,
                            c->d3_bwd_np, c->d3_bwd_dt, s));
  if (c->p1_units_f > 0 && pack_fwd)
    RLN_TRY(p1_pack_weights(c->params, c->p1_desc_f_dev, (int)c->p1_desc_f.size(), c->p1_units_f, c->d3_packed,
                            c->d3_fwd_np, c->d3_fwd_dt, s));
  if (c->p1_units_b > 0 && training && c->with_bwd)
    RLN_TRY(p1_pack_weights(c->params, c->p1_desc_b_dev, (int)c->p1_desc_b.size(), c->p1_units_b, c->d3_packed,
                            c->d3_bwd_np, c->d3_bwd_dt, s));
  for (size_t k = 0; k < c->ops.size(); ++k) RLN_TRY(fwd_op(c, k, x, training, s));
  if (probs_out || feat_out) {
    HeadParams hp = head_params(c);
    ProfScope ps(c, PC_HEAD_FWD, 2.0 * n * hp.HW * hp.C * (hp.ncls + 1),
                 4.0 * n * hp.HW * (hp.C + hp.ncls + (feat_out ? 2.0 * hp.C : 0.0)), s);
    RLN_TRY(head_forward(hp, n, probs_out, use_softmax, feat_out, s));
  }
  c->last_x = x;
  c->have_train_fwd = training ? 1 : 0;
  c->have_loss = 0;
  if (!training && c->eval_cache_on) c->eval_tables_valid = 1;
  c->eval_reuse = 0;
  return 0;
}

int rln_set_eval_cache(rln_ctx* c, int enable) {
  c->eval_cache_on = enable ? 1 : 0;
  c->eval_tables_valid = 0;
  return 0;
}

int rln_classifier_forward(rln_ctx* c, const float* feat, int n, int h, int w, float* out, int use_softmax,
                           void* stream) {
  if (!c->params) return fail(RLN_ERR_STATE, "rln_bind_params not called");
  HeadParams hp;
  hp.st = ST_F32;
  hp.S = feat;
  hp.C = c->feat_C;
  hp.HW = h * w;
  hp.ns = (long long)hp.C * hp.HW;
  hp.ncls = c->cfg.n_classes;
  hp.w = c->params + c->cls.w;
  hp.b = c->params + c->cls.b;
  hp.T = c->cfg.temperature;
  RLN_TRY(classifier_forward(hp, n, out, use_softmax, (hipStream_t)stream));
  return 0;
}

int rln_loss(rln_ctx* c, const float* probs, const int64_t* y, int n, int h, int w, int weighted, float* out,
             int64_t* argmax_out, int64_t* confusion_out, void* stream) {
  if (!c->loss.counts || n != c->N || h != c->H || w != c->W)
    return fail(RLN_ERR_WORKSPACE, "workspace not set for geometry %dx%dx%d", n, h, w);
  {
    ProfScope ps(c, PC_LOSS, 0, (4.0 * c->cfg.n_classes + 16.0) * n * h * w, (hipStream_t)stream);
    RLN_TRY(loss_forward(probs, (const long long*)y, n, c->cfg.n_classes, h * w, weighted, c->loss, out,
                         (long long*)argmax_out, (long long*)confusion_out, (hipStream_t)stream));
  }
  c->last_y = y;
  c->have_loss = weighted ? 1 : 0;
  c->loss_mode = 0;
  return 0;
}

int rln_entropy_loss(rln_ctx* c, const float* probs, int n, int h, int w, float lamda, float* out, void* stream) {
  if (!c->loss.counts || n != c->N || h != c->H || w != c->W)
    return fail(RLN_ERR_WORKSPACE, "workspace not set for geometry %dx%dx%d", n, h, w);
  RLN_TRY(entropy_forward(probs, n, c->cfg.n_classes, h * w, lamda, c->loss, out, (hipStream_t)stream));
  c->have_loss = 1;
  c->loss_mode = 1;
  c->loss_lamda = lamda;
  return 0;
}

int rln_set_output_grad(rln_ctx* c, const float* dprobs, int n, int h, int w) {
  if (!dprobs) return fail(RLN_ERR_ARG, "null gradient");
  if (n != c->N || h != c->H || w != c->W || !c->with_bwd)
    return fail(RLN_ERR_WORKSPACE, "backward workspace not set for geometry %dx%dx%d", n, h, w);
  if (!c->have_train_fwd) return fail(RLN_ERR_STATE, "rln_set_output_grad needs a training rln_forward first");
  c->gext = dprobs;
  c->loss_mode = 2;
  c->have_loss = 1;
  return 0;
}

int rln_sgd_step(float* params, const float* grads, float* momentum_buf, int64_t count, float lr, float momentum,
                 float weight_decay, int first_step, float grad_scale, void* stream) {
  if (!params || !grads || !momentum_buf || count < 0) return fail(RLN_ERR_ARG, "bad argument");
  RLN_TRY(sgd_nesterov(params, grads, momentum_buf, count, lr, momentum, weight_decay, first_step, grad_scale,
                       (hipStream_t)stream));
  return 0;
}

int rln_backward_segments(const rln_ctx* c) { return c->n_seg; }

int rln_backward_segment_range(const rln_ctx* c, int seg, int64_t* b, int64_t* e) {
  if (seg < 0 || seg >= c->n_seg) return fail(RLN_ERR_ARG, "segment %d out of range", seg);
  if (b) *b = c->seg_begin[seg];
  if (e) *e = c->seg_end[seg];
  return 0;
}

int rln_bind_grads(rln_ctx* c, float* grads) {
  if (!grads) return fail(RLN_ERR_ARG, "null gradient arena");
  c->grads = grads;
  return 0;
}

int rln_backward(rln_ctx* c, float loss_scale, int seg_begin, int seg_end, void* stream) {
  return rln_backward_scaled(c, loss_scale, nullptr, seg_begin, seg_end, stream);
}

int rln_backward_scaled(rln_ctx* c, float loss_scale, const float* loss_scale_dev, int seg_begin, int seg_end,
                        void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!c->with_bwd || !c->grads) return fail(RLN_ERR_STATE, "no gradient arena / backward workspace");
  if (!c->have_train_fwd || !c->have_loss)
    return fail(RLN_ERR_STATE, "rln_backward needs a training rln_forward followed by rln_loss(weighted) or "
                               "rln_entropy_loss");
  if (seg_begin < 0 || seg_end > c->n_seg || seg_begin >= seg_end) return fail(RLN_ERR_ARG, "bad segment range");
  const int N = c->N;
  // measured on MI355X: no gain (the kernels' LDS footprints do not co-reside on a CU and both are bound by the
  // memory system), so the concurrent weight-gradient stream is opt-in.
  if (!c->side && rln_env("RLN_SIDE_STREAM")) {
    hipError_t e = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking);
    for (int b = 0; b < 2 && e == hipSuccess; ++b) {
      e = hipEventCreateWithFlags(&c->ev_dy[b], hipEventDisableTiming);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_wg[b], hipEventDisableTiming);
    }
    if (e != hipSuccess) return fail((int)e, "side stream creation failed");
  }
  c->use_side = c->side != nullptr;
  if (seg_begin == 0) {
    hipError_t e = hipMemsetAsync(c->S1, 0, sizeof(float) * 2 * c->n_chan, s);
    if (e != hipSuccess) return fail((int)e, "memset failed");
    HeadBwdParams q;
    q.mode = c->loss_mode;
    q.gext = c->gext;
    q.lamda = c->loss_lamda;
    q.inv_count = (float)(1.0 / ((double)N * c->H * c->W));
    q.feat_sign = c->loss_mode == 1 ? -1.f : 1.f;  // grad_reverse between features and classifier (MME)
    q.h = head_params(c);
    q.y = (const long long*)c->last_y;
    q.lossres = c->loss.result;
    q.loss_scale = loss_scale;
    q.loss_scale_dev = loss_scale_dev;
    q.G = c->levels[0].G;
    q.g_ns = q.h.ns;
    q.invstd = c->invstd + c->levels[0].stat_off;
    q.glin = c->glin;
    q.bias_partial = c->bpartial;
    long long rows = 0;
    ProfScope ps(c, PC_HEAD_BWD, 2.0 * N * q.h.HW * q.h.C * (3.0 * q.h.ncls + 2), 4.0 * N * q.h.HW * 2.0 * q.h.C, s);
    if (q.h.C > 512 && q.h.st != ST_F32)
      return fail(RLN_ERR_UNSUPPORTED, "bf16 storage needs <= 512 feature channels in the head backward");
    if (q.h.C <= 512) {  // one pass over the feature stack
      RLN_TRY(head_backward_fused(q, N, c->wpartial, &rows, s));
      RLN_TRY(reduce_rows(c->bpartial, rows, c->cfg.n_classes, c->grads + c->cls.b, s));
      RLN_TRY(reduce_rows(c->wpartial, rows, (long long)c->cfg.n_classes * c->feat_C, c->grads + c->cls.w, s));
    } else {
      RLN_TRY(head_backward_data(q, N, &rows, s));
      RLN_TRY(reduce_rows(c->bpartial, rows, c->cfg.n_classes, c->grads + c->cls.b, s));
      RLN_TRY(head_backward_weight(q.h, N, c->glin, c->wpartial, s));
      RLN_TRY(reduce_rows(c->wpartial, N, (long long)c->cfg.n_classes * c->feat_C, c->grads + c->cls.w, s));
    }
  }
  for (int seg = std::max(seg_begin, 1); seg < seg_end; ++seg) {
    for (int k = (int)c->ops.size() - 1; k >= 0; --k) {
      if (c->ops[k].seg != seg) continue;
      if (c->ops[k].type == OP_DENSE && c->d3_bwd_np > 0 && !c->use_side) {  // whole dense block at once (pull form)
        int k0 = k;
        while (k0 > 0 && c->ops[k0 - 1].type == OP_DENSE && c->ops[k0 - 1].src_level == c->ops[k].src_level &&
               c->ops[k0 - 1].in_off == c->ops[k].in_off)
          --k0;
        const int r = bwd_dense_block(c, k0, k, s);
        if (r != 0 && r != kNotCovered) return r;
        if (r == 0) {
          k = k0;  // the loop decrement moves on to the op in front of the block
          continue;
        }
      }
      RLN_TRY(bwd_op(c, (size_t)k, s));
    }
  }
  // join: everything this call produced (incl. side-stream weight gradients) is ordered before later work on `s`
  for (int b = 0; b < 2; ++b) {
    if (c->use_side && c->wg_pending[b]) {
      hipError_t e = hipStreamWaitEvent(s, c->ev_wg[b], 0);
      if (e != hipSuccess) return fail((int)e, "hipStreamWaitEvent failed");
      c->wg_pending[b] = false;
    }
  }
  return 0;
}

int rln_op_conv_bnrelu(const float* x, int n, int cin, int x_ctot, int x_coff, int h, int w, const float* a,
                       const float* b, const float* weight, const float* bias, int cout, int ksize,
                       const float* scale, float* out, int out_ctot, int out_coff, int pool, uint8_t* pool_idx,
                       float* stats, void* workspace, size_t workspace_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (ksize != 1 && ksize != 3) return fail(RLN_ERR_ARG, "ksize must be 1 or 3");
  if (pool && ksize != 1) return fail(RLN_ERR_UNSUPPORTED, "pool epilogue exists for the 1x1 transition only");
  if (ksize == 1 && !a) return fail(RLN_ERR_UNSUPPORTED, "1x1 variant needs the folded affine");
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  const size_t plane = (size_t)h * w;
  p.in = x + (size_t)x_coff * plane;
  p.in_ns = (long long)x_ctot * plane;
  p.in_cs = (int)plane;
  p.Hin = h;
  p.Win = w;
  p.K = cin;
  p.pa = a;
  p.pb = b;
  p.w = weight;
  p.w_js = (long long)cin * ksize * ksize;
  p.w_ks = ksize * ksize;
  p.tapmode = TM_ID;
  p.J = cout;
  p.GH = h;
  p.GW = w;
  p.ncls = 1;
  const int ho = pool ? h / 2 : h, wo = pool ? w / 2 : w;
  p.out = out + (size_t)out_coff * ho * wo;
  p.out_ns = (long long)out_ctot * ho * wo;
  p.out_cs = ho * wo;
  p.Hout = ho;
  p.Wout = wo;
  p.bias = bias;
  p.nscale = scale;
  p.pool_idx = pool_idx;
  IgemmKind kind = (ksize == 3) ? (a ? IG_CONV3_BN : IG_CONV3_RAW) : (pool ? IG_CONV1_POOL : IG_CONV1_BN);
  const int tile = igemm_pick_tile(h, w);
  int th, tw;
  igemm_tile_dims(kind, tile, &th, &tw);
  p.tiles_y = (h + th - 1) / th;
  p.tiles_x = (w + tw - 1) / tw;
  p.out_vec = (!pool && (wo % 4) == 0 && aligned16(p.out)) ? 1 : 0;
  const long long nblk = igemm_stat_blocks(p, n);
  if (stats) {
    if (!workspace || workspace_bytes < (size_t)nblk * cout * 2 * sizeof(float))
      return fail(RLN_ERR_WORKSPACE, "stats need %lld bytes of workspace", nblk * cout * 2 * (long long)sizeof(float));
    p.stat_partial = (float*)workspace;
  }
  RLN_TRY(igemm_launch(kind, tile, p, n, s));
  if (stats) RLN_TRY(reduce_rows(p.stat_partial, nblk, (long long)cout * 2, stats, s));
  return 0;
}

int rln_op_td_fwd(const float* x, int n, int cin, int x_ctot, int x_coff, int h, int w, const float* a, const float* b,
                  const float* weight, const float* bias, int cout, const float* scale, float* out, int out_ctot,
                  int out_coff, uint8_t* pool_idx, float* stats, int parts, int dtype, void* workspace,
                  size_t workspace_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!x || !a || !b || !weight || !out || !pool_idx || !workspace) return fail(RLN_ERR_ARG, "null pointer");
  if (parts < 1 || parts > 3 || dtype < 0 || dtype > 1 || (dtype == 1 && parts == 3))
    return fail(RLN_ERR_ARG, "parts in 1..3 (f16: 1..2), dtype 0 (bf16) or 1 (f16)");
  const size_t plane = (size_t)h * w, pplane = (size_t)(h / 2) * (w / 2);
  P1Fwd p;
  memset(&p, 0, sizeof(p));
  p.S = x + (size_t)x_coff * plane;
  p.ns = (long long)x_ctot * plane;
  p.cs = (int)plane;
  p.H = h;
  p.W = w;
  p.Cin = cin;
  p.N = n;
  p.pa = a;
  p.pb = b;
  p.bias = bias;
  p.nscale = scale;
  p.out = out + (size_t)out_coff * pplane;
  p.out_ns = (long long)out_ctot * pplane;
  p.out_cs = (int)pplane;
  p.Cout = cout;
  p.pool_idx = pool_idx;
  if (!p1_fwd_supported(p)) return fail(RLN_ERR_UNSUPPORTED, "geometry not covered by the 1x1 transition forward kernel");
  p1_fwd_plan(&p, parts);
  Carver cv(workspace);
  P1PackDesc* desc = cv.take<P1PackDesc>(1);
  uint4* packed = cv.take<uint4>((size_t)p1_units_f(cin, cout) * parts * 64);
  float* partial = stats ? cv.take<float>((size_t)p.bpg * cout * 2) : nullptr;
  if (cv.off > workspace_bytes) return fail(RLN_ERR_WORKSPACE, "workspace of %zu bytes needed", cv.off);
  P1PackDesc d;
  d.w_off = 0;
  d.cin = cin;
  d.cout = cout;
  d.wf_off = 0;
  d.wb_off = -1;
  d.unit_begin = 0;
  d.n_units = p1_units_f(cin, cout);
  hipError_t e = hipMemcpyAsync(desc, &d, sizeof(d), hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);  // `d` is a stack object
  if (e != hipSuccess) return fail((int)e, "descriptor upload failed");
  RLN_TRY(p1_pack_weights(weight, desc, 1, d.n_units, packed, parts, dtype, s));
  p.wpk = packed;
  p.stat_partial = partial;
#ifdef RLN_DIAG
  if (rln_env("RLN_P1_DBG")) p.dbg = atoi(rln_env("RLN_P1_DBG"));
#endif
  RLN_TRY(p1_fwd_launch(p, parts, dtype, s));
  if (stats) RLN_TRY(reduce_rows(partial, p.bpg, (long long)cout * 2, stats, s));
  return 0;
}

int rln_op_tu_fwd(const float* x, int n, int cin, int x_ctot, int x_coff, int h, int w, const float* weight,
                  const float* bias, int cout, float* out, int out_ctot, int out_coff, int hout, int wout, float* stats,
                  int parts, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!x || !weight || !out || !workspace) return fail(RLN_ERR_ARG, "null pointer");
  if (parts < 1 || parts > 3 || dtype < 0 || dtype > 1 || (dtype == 1 && parts == 3))
    return fail(RLN_ERR_ARG, "parts in 1..3 (f16: 1..2), dtype 0 (bf16) or 1 (f16)");
  if (hout > 2 * h + 1 || wout > 2 * w + 1) return fail(RLN_ERR_ARG, "crop larger than the transposed conv output");
  const size_t plane = (size_t)h * w, oplane = (size_t)hout * wout;
  C3Fwd p;
  memset(&p, 0, sizeof(p));
  p.X = x + (size_t)x_coff * plane;
  p.ns = (long long)x_ctot * plane;
  p.cs = (int)plane;
  p.H = h;
  p.W = w;
  p.Cin = cin;
  p.N = n;
  p.bias = bias;
  p.out = out + (size_t)out_coff * oplane;
  p.out_ns = (long long)out_ctot * oplane;
  p.out_cs = (int)oplane;
  p.Cout = cout;
  p.Ho = hout;
  p.Wo = wout;
  if (!c3_fwd_supported(p)) return fail(RLN_ERR_UNSUPPORTED, "geometry not covered by the transposed-conv forward kernel");
  c3_fwd_plan(&p, parts);
  Carver cv(workspace);
  C3PackDesc* desc = cv.take<C3PackDesc>(1);
  uint4* packed = cv.take<uint4>((size_t)c3_units_f(cin, cout) * parts * 64);
  float* partial = stats ? cv.take<float>((size_t)p.bpg * cout * 2) : nullptr;
  if (cv.off > workspace_bytes) return fail(RLN_ERR_WORKSPACE, "workspace of %zu bytes needed", cv.off);
  C3PackDesc d;
  d.w_off = 0;
  d.cin = cin;
  d.cout = cout;
  d.wf_off = 0;
  d.wb_off = -1;
  d.unit_begin = 0;
  d.n_units = c3_units_f(cin, cout);
  hipError_t e = hipMemcpyAsync(desc, &d, sizeof(d), hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);  // `d` is a stack object
  if (e != hipSuccess) return fail((int)e, "descriptor upload failed");
  RLN_TRY(c3_pack_weights(weight, desc, 1, d.n_units, packed, parts, dtype, s));
  p.wpk = packed;
  p.stat_partial = partial;
  RLN_TRY(c3_fwd_launch(p, parts, dtype, s));
  if (stats) RLN_TRY(reduce_rows(partial, p.bpg, (long long)cout * 2, stats, s));
  return 0;
}

int rln_op_tu_bwd(const float* x, const float* du, const float* weight, int n, int cin, int cout, int h, int w, int hout,
                  int wout, const float* cscale, float* dx, float* dw, int parts, int dtype, void* workspace,
                  size_t workspace_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!x || !du || !weight || !workspace || (!dx && !dw)) return fail(RLN_ERR_ARG, "null pointer");
  if (parts < 1 || parts > 3 || dtype < 0 || dtype > 1 || (dtype == 1 && parts == 3))
    return fail(RLN_ERR_ARG, "parts in 1..3 (f16: 1..2), dtype 0 (bf16) or 1 (f16)");
  const size_t plane = (size_t)h * w;
  C3Dgrad q;
  memset(&q, 0, sizeof(q));
  q.dU = du;
  q.Cout = cout;
  q.Ho = hout;
  q.Wo = wout;
  q.cscale = cscale;
  q.G = dx;
  q.ns = (long long)cin * plane;
  q.cs = (int)plane;
  q.H = h;
  q.W = w;
  q.C = cin;
  q.N = n;
  C3Wgrad g;
  memset(&g, 0, sizeof(g));
  g.X = x;
  g.ns = q.ns;
  g.cs = q.cs;
  g.H = h;
  g.W = w;
  g.Cin = cin;
  g.N = n;
  g.dU = du;
  g.Cout = cout;
  g.Ho = hout;
  g.Wo = wout;
  if ((dx && !c3_dgrad_supported(q)) || (dw && !c3_wgrad_supported(g)))
    return fail(RLN_ERR_UNSUPPORTED, "geometry not covered by the transposed-conv backward kernels");
  c3_dgrad_plan(&q);
  if (dw) c3_wgrad_plan(&g);
  Carver cv(workspace);
  C3PackDesc* desc = cv.take<C3PackDesc>(1);
  uint4* packed = cv.take<uint4>((size_t)c3_entries_b(cin, cout, parts));
  float* partial = cv.take<float>(dw ? (size_t)g.nranges * cin * cout * 9 : 0);
  if (cv.off > workspace_bytes) return fail(RLN_ERR_WORKSPACE, "workspace of %zu bytes needed", cv.off);
  if (dx) {
    C3PackDesc d;
    d.w_off = 0;
    d.cin = cin;
    d.cout = cout;
    d.wf_off = -1;
    d.wb_off = 0;
    d.unit_begin = 0;
    d.n_units = c3_units_b(cin, cout);
    hipError_t e = hipMemcpyAsync(desc, &d, sizeof(d), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);  // `d` is a stack object
    if (e != hipSuccess) return fail((int)e, "descriptor upload failed");
    RLN_TRY(c3_pack_weights(weight, desc, 1, d.n_units, packed, parts, dtype, s));
    q.wpk = packed;
    RLN_TRY(c3_dgrad_launch(q, parts, dtype, s));
  }
  if (dw) {
    g.partial = partial;
    RLN_TRY(c3_wgrad_launch(g, parts, dtype, s));
    RLN_TRY(reduce_rows(partial, g.nranges, (long long)cin * cout * 9, dw, s));
  }
  return 0;
}

int rln_op_td_bwd(const float* x, const float* dyp, const uint8_t* pool_idx, const float* weight, int n, int cin, int cout,
                  int h, int w, const float* a, const float* b, const float* gamma, const float* mean, const float* invstd,
                  int acc_lo, int acc_hi, float* g, float* stats, float* dw, int parts, int dtype, void* workspace,
                  size_t workspace_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!x || !dyp || !pool_idx || !weight || !a || !b || !workspace || (!g && !dw)) return fail(RLN_ERR_ARG, "null pointer");
  if (g && (!gamma || !mean || !invstd)) return fail(RLN_ERR_ARG, "gamma / mean / invstd are required with g");
  if (parts < 1 || parts > 3 || dtype < 0 || dtype > 1 || (dtype == 1 && parts == 3))
    return fail(RLN_ERR_ARG, "parts in 1..3 (f16: 1..2), dtype 0 (bf16) or 1 (f16)");
  const size_t plane = (size_t)h * w;
  P1Dgrad q;
  memset(&q, 0, sizeof(q));
  q.dYp = dyp;
  q.pool_idx = pool_idx;
  q.Cout = cout;
  q.ea = a;
  q.eb = b;
  q.egamma = gamma;
  q.mean = mean;
  q.invstd = invstd;
  q.S = x;
  q.ns = (long long)cin * plane;
  q.cs = (int)plane;
  q.G = g;
  q.C = cin;
  q.acc_lo = acc_lo;
  q.acc_hi = acc_hi;
  q.H = h;
  q.W = w;
  q.N = n;
  P1Wgrad gw;
  memset(&gw, 0, sizeof(gw));
  gw.dYp = dyp;
  gw.pool_idx = pool_idx;
  gw.Cout = cout;
  gw.S = x;
  gw.ns = q.ns;
  gw.cs = q.cs;
  gw.H = h;
  gw.W = w;
  gw.N = n;
  gw.Cin = cin;
  gw.pa = a;
  gw.pb = b;
  if ((g && !p1_dgrad_supported(q)) || (dw && !p1_wgrad_supported(gw)))
    return fail(RLN_ERR_UNSUPPORTED, "geometry not covered by the 1x1 transition backward kernels");
  if (g) p1_dgrad_plan(&q, parts);
  if (dw) p1_wgrad_plan(&gw);
  Carver cv(workspace);
  P1PackDesc* desc = cv.take<P1PackDesc>(1);
  uint4* packed = cv.take<uint4>((size_t)p1_units_b(cin, cout) * parts * 64);
  float* spart = (g && stats) ? cv.take<float>((size_t)q.bpg * cin * 2) : nullptr;
  float* wpart = dw ? cv.take<float>((size_t)gw.nranges * cin * cout) : nullptr;
  if (cv.off > workspace_bytes) return fail(RLN_ERR_WORKSPACE, "workspace of %zu bytes needed", cv.off);
  if (g) {
    P1PackDesc d;
    d.w_off = 0;
    d.cin = cin;
    d.cout = cout;
    d.wf_off = -1;
    d.wb_off = 0;
    d.unit_begin = 0;
    d.n_units = p1_units_b(cin, cout);
    hipError_t e = hipMemcpyAsync(desc, &d, sizeof(d), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);  // `d` is a stack object
    if (e != hipSuccess) return fail((int)e, "descriptor upload failed");
    RLN_TRY(p1_pack_weights(weight, desc, 1, d.n_units, packed, parts, dtype, s));
    q.wpk = packed;
    q.stat_partial = spart;
    RLN_TRY(p1_dgrad_launch(q, parts, dtype, s));
    if (stats) RLN_TRY(reduce_rows(spart, q.bpg, (long long)cin * 2, stats, s));
  }
  if (dw) {
    gw.partial = wpart;
    RLN_TRY(p1_wgrad_launch(gw, parts, dtype, s));
    RLN_TRY(reduce_rows(wpart, gw.nranges, (long long)cin * cout, dw, s));
  }
  return 0;
}

int rln_op_fc_fwd(const float* x, int n, int cin, int h, int w, const float* weight, const float* bias, int cout,
                  float* out, int out_ctot, int out_coff, float* stats, int parts, int dtype, void* workspace,
                  size_t workspace_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!x || !weight || !out || !workspace) return fail(RLN_ERR_ARG, "null pointer");
  const size_t plane = (size_t)h * w;
  F3Fwd p;
  memset(&p, 0, sizeof(p));
  p.X = x;
  p.Cin = cin;
  p.H = h;
  p.W = w;
  p.N = n;
  p.w = weight;
  p.bias = bias;
  p.out = out + (size_t)out_coff * plane;
  p.out_ns = (long long)out_ctot * plane;
  p.out_cs = (int)plane;
  p.Cout = cout;
  if (!f3_fwd_supported(p)) return fail(RLN_ERR_UNSUPPORTED, "geometry not covered by the first-conv forward kernel");
  f3_fwd_plan(&p);
  Carver cv(workspace);
  float* partial = stats ? cv.take<float>((size_t)p.blocks * cout * 2) : nullptr;
  if (cv.off > workspace_bytes) return fail(RLN_ERR_WORKSPACE, "workspace of %zu bytes needed", cv.off);
  p.stat_partial = partial;
  RLN_TRY(f3_fwd_launch(p, parts, dtype, s));
  if (stats) RLN_TRY(reduce_rows(partial, p.blocks, (long long)cout * 2, stats, s));
  return 0;
}

int rln_op_fc_wgrad(const float* x, const float* dy, int n, int cin, int cout, int h, int w, float* dw, int parts, int dtype,
                    void* workspace, size_t workspace_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!x || !dy || !dw || !workspace) return fail(RLN_ERR_ARG, "null pointer");
  F3Wgrad g;
  memset(&g, 0, sizeof(g));
  g.X = x;
  g.Cin = cin;
  g.H = h;
  g.W = w;
  g.N = n;
  g.dY = dy;
  g.Cout = cout;
  if (!f3_wgrad_supported(g)) return fail(RLN_ERR_UNSUPPORTED, "geometry not covered by the first-conv weight-gradient kernel");
  f3_wgrad_plan(&g);
  Carver cv(workspace);
  g.partial = cv.take<float>((size_t)g.blocks * cout * cin * 9);
  if (cv.off > workspace_bytes) return fail(RLN_ERR_WORKSPACE, "workspace of %zu bytes needed", cv.off);
  RLN_TRY(f3_wgrad_launch(g, parts, dtype, s));
  RLN_TRY(reduce_rows(g.partial, g.blocks, (long long)cout * cin * 9, dw, s));
  return 0;
}

int rln_op_dense3_fwd(const float* x, int n, int cin, int x_ctot, int x_coff, int h, int w, const float* a,
                      const float* b, const float* weight, const float* bias, int cout, const float* scale, float* out,
                      int out_ctot, int out_coff, float* stats, int parts, int dtype, void* workspace,
                      size_t workspace_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!x || !a || !b || !weight || !out || !workspace) return fail(RLN_ERR_ARG, "null pointer");
  if (parts < 1 || parts > 3 || dtype < 0 || dtype > 1 || (dtype == 1 && parts == 3))
    return fail(RLN_ERR_ARG, "parts in 1..3 (f16: 1..2), dtype 0 (bf16) or 1 (f16)");
  const size_t plane = (size_t)h * w;
  D3Fwd p;
  memset(&p, 0, sizeof(p));
  p.S = x + (size_t)x_coff * plane;
  p.ns = (long long)x_ctot * plane;
  p.cs = (int)plane;
  p.H = h;
  p.W = w;
  p.Cin = cin;
  p.pa = a;
  p.pb = b;
  p.bias = bias;
  p.nscale = scale;
  p.out = out + (size_t)out_coff * plane;
  p.out_ns = (long long)out_ctot * plane;
  p.out_cs = (int)plane;
  p.Cout = cout;
  p.ksplit = 1;
  if (!d3_fwd_supported(p)) return fail(RLN_ERR_UNSUPPORTED, "geometry not covered by the dense3 forward kernel");
  d3_fwd_pick_tile(h, w, parts, &p.th, &p.tw, &p.rg);
  p.tiles_y = (h + p.th - 1) / p.th;
  p.tiles_x = (w + p.tw - 1) / p.tw;
  // workspace: [descriptor | packed weights | statistics partials]
  Carver cv(workspace);
  D3PackDesc* desc = cv.take<D3PackDesc>(1);
  const long long entries = d3_pack_entries(cin, parts);
  uint4* packed = cv.take<uint4>((size_t)entries);
  const long long nblk = (long long)n * p.tiles_x * p.tiles_y;
  float* partial = stats ? cv.take<float>((size_t)nblk * cout * 2) : nullptr;
  if (cv.off > workspace_bytes) return fail(RLN_ERR_WORKSPACE, "workspace of %zu bytes needed", cv.off);
  // the weight may live anywhere: describe it relative to itself
  D3PackDesc d;
  d.w_off = 0;
  d.cin = cin;
  d.cout = cout;
  d.wf_off = 0;
  d.wb_off = -1;
  d.unit_begin = 0;
  d.n_units = 5 * ((cin + 15) / 16);
  hipError_t e = hipMemcpyAsync(desc, &d, sizeof(d), hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);  // `d` is a stack object
  if (e != hipSuccess) return fail((int)e, "descriptor upload failed");
  RLN_TRY(d3_pack_weights(weight, desc, 1, d.n_units, packed, parts, dtype, s));
  p.wpk = packed;
  p.stat_partial = partial;
#ifdef RLN_DIAG
  if (rln_env("RLN_D3_DBG")) p.dbg = atoi(rln_env("RLN_D3_DBG"));
#endif
  RLN_TRY(d3_fwd_launch(p, n, parts, dtype, s));
  if (stats) RLN_TRY(reduce_rows(partial, nblk, (long long)cout * 2, stats, s));
  return 0;
}

int rln_op_dense3_fwd_pair(float* stack, int n, int cin, int ctot, int coff, int h, int w, const float* a1,
                           const float* b1, const float* w1, const float* bias1, const float* scale1, const float* a2,
                           const float* b2, const float* w2, const float* bias2, const float* scale2, float* stats1,
                           float* stats2, int parts, int dtype, float* scratch, void* workspace,
                           size_t workspace_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (parts < 1 || parts > 2) return fail(RLN_ERR_ARG, "parts 1 or 2");
  if (!stack || !a1 || !b1 || !w1 || !a2 || !b2 || !w2 || !scratch || !workspace) return fail(RLN_ERR_ARG, "null pointer");
  if (dtype < 0 || dtype > 1) return fail(RLN_ERR_ARG, "dtype 0 (bf16) or 1 (f16)");
  if (cin < 16 || (cin % 16) != 0 || coff + cin + 32 > ctot)
    return fail(RLN_ERR_ARG, "cin a multiple of 16 and room for 32 new channels after the input range");
  const size_t plane = (size_t)h * w;
  const int cout = 16;
  D3Fwd p;
  memset(&p, 0, sizeof(p));
  p.S = stack + (size_t)coff * plane;
  p.ns = (long long)ctot * plane;
  p.cs = (int)plane;
  p.H = h;
  p.W = w;
  p.Cin = cin;
  p.pa = a1;
  p.pb = b1;
  p.bias = bias1;
  p.nscale = scale1;
  p.out = stack + (size_t)(coff + cin) * plane;
  p.out_ns = p.ns;
  p.out_cs = (int)plane;
  p.Cout = cout;
  p.ksplit = 1;
  if (!d3_fwd_supported(p)) return fail(RLN_ERR_UNSUPPORTED, "geometry not covered by the dense3 forward kernel");
  d3_fwd_pick_tile(h, w, parts, &p.th, &p.tw, &p.rg);
  p.tiles_y = (h + p.th - 1) / p.th;
  p.tiles_x = (w + p.tw - 1) / p.tw;
  // workspace: [2 descriptors | packed weights of both layers | statistics partials]
  Carver cv(workspace);
  D3PackDesc* desc = cv.take<D3PackDesc>(2);
  const long long e1 = d3_pack_entries(cin, parts), e2 = d3_pack_entries(cin + 16, parts);
  uint4* packed1 = cv.take<uint4>((size_t)e1);
  uint4* packed2 = cv.take<uint4>((size_t)e2);
  const long long nblk = (long long)n * p.tiles_x * p.tiles_y;
  const long long nfin = d3_fin_rows(h, w, n);
  float* partial = (stats1 || stats2) ? cv.take<float>((size_t)std::max(nblk, nfin) * cout * 2) : nullptr;
  if (cv.off > workspace_bytes) return fail(RLN_ERR_WORKSPACE, "workspace of %zu bytes needed", cv.off);
  const float* wts[2] = {w1, w2};
  uint4* pks[2] = {packed1, packed2};
  for (int i = 0; i < 2; ++i) {  // the weights may live anywhere: each is described relative to itself
    D3PackDesc d;
    d.w_off = 0;
    d.cin = cin + 16 * i;
    d.cout = cout;
    d.wf_off = 0;
    d.wb_off = -1;
    d.unit_begin = 0;
    d.n_units = 5 * ((d.cin + 15) / 16);
    hipError_t e = hipMemcpyAsync(desc + i, &d, sizeof(d), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);  // `d` is a stack object
    if (e != hipSuccess) return fail((int)e, "descriptor upload failed");
    RLN_TRY(d3_pack_weights(wts[i], desc + i, 1, d.n_units, pks[i], parts, dtype, s));
  }
  // layer 1 complete + layer 2's raw sums over the shared chunks
  D3Fwd q = p;
  q.wpk = packed1;
  q.stat_partial = stats1 ? partial : nullptr;
  q.pa2 = a2;
  q.pb2 = b2;
  q.wpk2 = packed2;
  q.partial_out = scratch;
#ifdef RLN_DIAG
  if (rln_env("RLN_D3_DBG")) q.dbg = atoi(rln_env("RLN_D3_DBG"));
#endif
  RLN_TRY(d3_fwd_pair_launch(q, n, parts, dtype, s));
  if (stats1) RLN_TRY(reduce_rows(partial, nblk, (long long)cout * 2, stats1, s));
  // layer 2: its last chunk (layer 1's output) on top of the raw sums
  D3Fwd f = p;
  f.Cin = cin + 16;
  f.pa = a2;
  f.pb = b2;
  f.wpk = packed2;
  f.bias = bias2;
  f.nscale = scale2;
  f.out = stack + (size_t)(coff + cin + 16) * plane;
  f.stat_partial = stats2 ? partial : nullptr;
  f.c_first = cin / 16;
  f.partial_in = scratch;
  if (d3_fin_supported(f, parts)) {  // the light finishing kernel where it covers the geometry (as the engine does)
    RLN_TRY(d3_fin_launch(f, n, parts, dtype, s));
    if (stats2) RLN_TRY(reduce_rows(partial, nfin, (long long)cout * 2, stats2, s));
    return 0;
  }
  RLN_TRY(d3_fwd_launch(f, n, parts, dtype, s));
  if (stats2) RLN_TRY(reduce_rows(partial, nblk, (long long)cout * 2, stats2, s));
  return 0;
}

int rln_op_convt(const float* x, int n, int cin, int h, int w, const float* weight, const float* bias, int cout,
                 float* out, int out_ctot, int out_coff, int hout, int wout, void* stream) {
  if (hout > 2 * h + 1 || wout > 2 * w + 1) return fail(RLN_ERR_ARG, "crop larger than the transposed conv output");
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.in = x;
  p.in_ns = (long long)cin * h * w;
  p.in_cs = h * w;
  p.Hin = h;
  p.Win = w;
  p.K = cin;
  p.w = weight;
  p.w_ks = (long long)cout * 9;
  p.w_js = 9;
  p.tapmode = TM_CONVT;
  p.ncls = 4;
  if (convt_fused()) {
    p.tapmode = TM_ID;
    p.ncls = 1;
  }
  p.J = cout;
  p.GH = (hout + 1) / 2;
  p.GW = (wout + 1) / 2;
  p.out = out + (size_t)out_coff * hout * wout;
  p.out_ns = (long long)out_ctot * hout * wout;
  p.out_cs = hout * wout;
  p.Hout = hout;
  p.Wout = wout;
  p.bias = bias;
  const int tile = igemm_pick_tile(p.GH, p.GW);
  int th, tw;
  igemm_tile_dims(IG_CONV3_RAW, tile, &th, &tw);
  p.tiles_y = (p.GH + th - 1) / th;
  p.tiles_x = (p.GW + tw - 1) / tw;
  RLN_TRY(igemm_launch(p.ncls == 1 ? IG_CONVT4 : IG_CONV3_RAW, tile, p, n, (hipStream_t)stream));
  return 0;
}

int rln_profile_enable(rln_ctx* c, int on) {
  for (ProfEntry& e : c->prof.entries) {
    c->prof.pool.push_back(e.a);
    c->prof.pool.push_back(e.b);
  }
  c->prof.entries.clear();
  c->prof.on = on != 0;
  return 0;
}

// diagnostic: copies the 8 in-kernel phase counters (RLN_DBG=16) to the host and clears them
int rln_debug_read_stamps(unsigned long long* out8) {
  unsigned long long* d = igemm_debug_buffer();
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpy(out8, d, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  (void)hipMemset(d, 0, 8 * sizeof(unsigned long long));
  return 0;
}

int rln_profile_num_classes(void) { return PC_COUNT; }
const char* rln_profile_class_name(int cls) { return (cls >= 0 && cls < PC_COUNT) ? kProfNames[cls] : ""; }

int rln_profile_read(rln_ctx* c, double* ms, double* flops, double* bytes, int64_t* launches) {
  for (int i = 0; i < PC_COUNT; ++i) {
    ms[i] = 0;
    flops[i] = 0;
    bytes[i] = 0;
    launches[i] = 0;
  }
  for (ProfEntry& e : c->prof.entries) {
    hipError_t err = hipEventSynchronize(e.b);
    if (err != hipSuccess) return fail((int)err, "hipEventSynchronize failed");
    float t = 0.f;
    err = hipEventElapsedTime(&t, e.a, e.b);
    if (err != hipSuccess) return fail((int)err, "hipEventElapsedTime failed");
    ms[e.cls] += t;
    flops[e.cls] += e.flops;
    bytes[e.cls] += e.bytes;
    launches[e.cls] += 1;
  }
  return 0;
}

int64_t rln_profile_entries(rln_ctx* c, int* cls, double* ms, double* flops, double* bytes, int64_t cap) {
  int64_t i = 0;
  for (ProfEntry& e : c->prof.entries) {
    if (i < cap) {
      if (hipEventSynchronize(e.b) != hipSuccess) return -1;
      float t = 0.f;
      if (hipEventElapsedTime(&t, e.a, e.b) != hipSuccess) return -1;
      cls[i] = e.cls;
      ms[i] = t;
      flops[i] = e.flops;
      bytes[i] = e.bytes;
    }
    ++i;
  }
  return i;
}

// ---- EncDecNet building blocks (models/EncDecNet.py) ------------------------------------------------------
int rln_op_conv_act(const float* x, int n, int cin, int h, int w, const float* weight, const float* bias, int cout,
                    int ksize, int act, float act_param, float* out, float* stats, void* workspace,
                    size_t workspace_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (ksize != 1 && ksize != 3 && ksize != 7) return fail(RLN_ERR_UNSUPPORTED, "kernel size %d (1, 3, 7 are built)", ksize);
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  const size_t plane = (size_t)h * w;
  p.in = x;
  p.in_ns = (long long)cin * plane;
  p.in_cs = (int)plane;
  p.Hin = h;
  p.Win = w;
  p.K = cin;
  p.w = weight;
  p.w_js = (long long)cin * ksize * ksize;
  p.w_ks = ksize * ksize;
  p.tapmode = TM_ID;
  p.J = cout;
  p.GH = h;
  p.GW = w;
  p.ncls = 1;
  p.out = out;
  p.out_ns = (long long)cout * plane;
  p.out_cs = (int)plane;
  p.Hout = h;
  p.Wout = w;
  p.bias = bias;
  p.act = act;
  p.act_param = act_param;
  const IgemmKind kind = ksize == 7 ? IG_CONV7_RAW : (ksize == 3 ? IG_CONV3_RAW : IG_CONV1_RAW);
  const int tile = igemm_pick_tile(h, w);
  int th, tw;
  igemm_tile_dims(kind, tile, &th, &tw);
  p.tiles_y = (h + th - 1) / th;
  p.tiles_x = (w + tw - 1) / tw;
  p.out_vec = ((w % 4) == 0 && aligned16(out)) ? 1 : 0;
  const long long nblk = igemm_stat_blocks(p, n);
  if (stats) {
    if (!workspace || workspace_bytes < (size_t)nblk * cout * 2 * sizeof(float))
      return fail(RLN_ERR_WORKSPACE, "stats need %lld bytes of workspace", nblk * cout * 2 * (long long)sizeof(float));
    p.stat_partial = (float*)workspace;
  }
  RLN_TRY(igemm_launch(kind, tile, p, n, s));
  if (stats) RLN_TRY(reduce_rows(p.stat_partial, nblk, (long long)cout * 2, stats, s));
  return 0;
}

int rln_op_bn_affine(const float* sums, int c, double count, int training, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, float momentum, float eps, float* a, float* b,
                     void* stream) {
  if (!training && (!running_mean || !running_var)) return fail(RLN_ERR_ARG, "eval mode needs running statistics");
  RLN_TRY(bn_affine_from_sums(sums, c, count, training, gamma, beta, running_mean, running_var, momentum, eps, a, b,
                              (hipStream_t)stream));
  return 0;
}

int rln_op_bn_drop_maxpool(const float* x, int n, int c, int h, int w, const float* a, const float* b,
                           const float* mask, int k, float* out, void* stream) {
  if (k < 1 || (k & 1) == 0) return fail(RLN_ERR_ARG, "odd pooling kernel expected");
  RLN_TRY(bn_drop_maxpool(x, n, c, h, w, a, b, mask, k, out, (hipStream_t)stream));
  return 0;
}

int rln_op_bn_drop_upsample2(const float* x, int n, int c, int h, int w, const float* a, const float* b,
                             const float* mask, float* out, void* stream) {
  RLN_TRY(bn_drop_upsample2(x, n, c, h, w, a, b, mask, out, (hipStream_t)stream));
  return 0;
}

int rln_op_softmax_channels(const float* x, int n, int c, int hw, float* out, void* stream) {
  RLN_TRY(softmax_channels(x, n, c, hw, out, (hipStream_t)stream));
  return 0;
}

int rln_preprocess_u8(const uint8_t* frames, int n, int hs, int ws, const uint8_t* labels, int h, int w, int gray,
                      const float* mean3, const float* std3, float* x, int64_t* y, void* stream) {
  if (!frames || !x || !mean3 || !std3) return fail(RLN_ERR_ARG, "null pointer");
  if (n < 1 || hs < 1 || ws < 1 || h < 1 || w < 1) return fail(RLN_ERR_ARG, "bad sizes");
  if ((labels == nullptr) != (y == nullptr)) return fail(RLN_ERR_ARG, "labels and y must be given together");
  for (int c = 0; c < 3; ++c)
    if (!(std3[c] > 0.f)) return fail(RLN_ERR_ARG, "std must be positive");
  RLN_TRY(preprocess_u8(frames, n, hs, ws, labels, h, w, gray, mean3, std3, x, (long long*)y, (hipStream_t)stream));
  return 0;
}

int rln_augment_u8(const uint8_t* frames, int n, int hs, int ws, const uint8_t* labels, int h, int w,
                   const float* params, const float* mean3, const float* std3, uint8_t* scratch, float* x, int64_t* y,
                   void* stream) {
  if (!frames || !x || !mean3 || !std3 || !params || !scratch) return fail(RLN_ERR_ARG, "null pointer");
  if (n < 1 || hs < 2 || ws < 2 || h < 1 || w < 1) return fail(RLN_ERR_ARG, "bad sizes");
  if ((labels == nullptr) != (y == nullptr)) return fail(RLN_ERR_ARG, "labels and y must be given together");
  for (int c = 0; c < 3; ++c)
    if (!(std3[c] > 0.f)) return fail(RLN_ERR_ARG, "std must be positive");
  RLN_TRY(augment_u8(frames, n, hs, ws, labels, h, w, params, mean3, std3, scratch, x, (long long*)y,
                     (hipStream_t)stream));
  return 0;
}

int rln_overlay_u8(const uint8_t* frames, int n, int hs, int ws, const float* probs, int ncls, int h, int w,
                   const uint8_t* colors_host, unsigned paint_mask, uint8_t* out, uint8_t* pred_out, void* stream) {
  if (!frames || !probs || !colors_host || !out) return fail(RLN_ERR_ARG, "null pointer");
  if (n < 1 || hs < 1 || ws < 1 || h < 1 || w < 1 || ncls < 1 || ncls > 16) return fail(RLN_ERR_ARG, "bad sizes");
  RLN_TRY(overlay_u8(frames, n, hs, ws, probs, ncls, h, w, colors_host, paint_mask, out, pred_out,
                     (hipStream_t)stream));
  return 0;
}

int rln_op_scaled_softmax(const float* x, int n, int c, int hw, float T, int use_softmax, float* out, void* stream) {
  if (!(T > 0.f) || n < 1 || c < 1 || hw < 1) return fail(RLN_ERR_ARG, "bad arguments");
  RLN_TRY(softmax_channels(x, n, c, hw, out, (hipStream_t)stream, T, use_softmax));
  return 0;
}

int rln_op_dropout_mask(float* dst, int64_t count, float keep, uint64_t seed, void* stream) {
  RLN_TRY(dropout_scales(dst, count, keep, (unsigned long long)seed, (hipStream_t)stream));
  return 0;
}

int rln_op_classifier(const float* feat, int n, int c, int hw, const float* w, const float* b, int ncls, float T,
                      float* out, int use_softmax, void* stream) {
  HeadParams hp;
  hp.st = ST_F32;
  hp.S = feat;
  hp.C = c;
  hp.HW = hw;
  hp.ns = (long long)c * hw;
  hp.ncls = ncls;
  hp.w = w;
  hp.b = b;
  hp.T = T;
  RLN_TRY(classifier_forward(hp, n, out, use_softmax, (hipStream_t)stream));
  return 0;
}

int rln_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t count, float lr,
                   float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                   void* stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || count < 0 || step < 1) return fail(RLN_ERR_ARG, "bad argument");
  RLN_TRY(adamw(params, grads, exp_avg, exp_avg_sq, count, lr, beta1, beta2, eps, weight_decay, step, grad_scale,
                (hipStream_t)stream));
  return 0;
}

}  // extern "C"
