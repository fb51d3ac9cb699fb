// Whole-encoder engine: sequences the gfx950 kernels for ViT.forward (vit_3d.py:112-126) and its
// backward as ONE C-ABI call each, over a flat parameter arena and a caller-owned workspace.
// Host-side only (no kernels here): a native "executor" so the Python layer issues 1 call per
// forward / backward instead of ~100 per-op launches.  Never allocates, never synchronises.
#include <stdlib.h>
#include <string.h>

#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/neurovit_hip.h"

#include "common.h"

namespace {

inline long align_up(long v, long a) { return (v + a - 1) / a * a; }

struct Dims {
  int B, N, n, P, Ppad, d, inner, m, L, C, heads, dh, M, T;   // M = B*n rows of the stream, T = B*N tokens
  int gf, gh, gw;
  int pool_mean;
};

// pool = 'cls' (the NeuroEncoder path, NeuroEncoder.py:194): behind the last block's attention only the B cls rows reach the head, and in
// the backward pass the residual gradient entering the last block is exactly zero in every other row.  In the cls-rows form the
// last block's out-projection, LayerNorm and FeedForward - forward and backward - run on those B rows only, as strided views (row
// stride n) through the weight-streaming kernels of skinny.hip: the same values for the logits and every gradient (the skipped rows
// contribute exact zeros; in the fp8 inference path the block's FeedForward then runs in bf16), but rows 1..n-1 of the last block's
// x1 / xn2 / u / h / x2 buffers are not produced.  The form is an ARGUMENT of every call (rows_form: nv_vit_input for the forwards,
// a parameter of nv_vit_backward_stages16): 1 = every row, as the reference computes it; 2 = cls rows when eligible; 0 = the process
// default set by nv_vit_set_cls_tail (on unless switched off).  The decision is a pure function of (config, B, training, dropout,
// rows_form), so a backward given the forward's values takes the forward's form - nothing is remembered per workspace.
// The skinny kernels apply no dropout: the cls-rows form needs the block dropout off, in training AND in a train-mode forward
// that records no graph (the frozen encoder of the 4D model under Trainer.train, config4D.yaml TRAINING_DROPOUT 0.2).
static int g_cls_tail = 1;
extern "C" int nv_vit_set_cls_tail(int on) { g_cls_tail = on ? 1 : 0; return 0; }
static int g_head_step = 1;      // nv_vit_train_step: head forward + loss + head backward as one launch (nv_head_step) where it exists
extern "C" int nv_vit_set_head_step(int on) { g_head_step = on ? 1 : 0; return 0; }
static bool cls_tail_wanted(const Dims& D, int training, float drop_p, int rows_form) {
  const bool want = rows_form == 1 ? false : (rows_form == 2 ? true : g_cls_tail != 0);
  (void)drop_p;      // (round 4: the cls-row kernels carry the nn.Dropout masks of the dense tensors, skinny.hip)
  return want && !D.pool_mean && (!training || D.B <= 4);
}

// nn.Dropout behind the output projection (vit_3d.py:45): absent - with the projection itself - when heads == 1 and dim_head == dim
// (vit_3d.py:32,43-46: to_out = nn.Identity()); the engine still runs that geometry's projection slot (identity weight, zero bias)
static inline float proj_drop_p(const nv_vit_config* c, float drop_p) { return c->no_proj_dropout ? 0.f : drop_p; }

// image width / width of a patch: nv_vit_config.image_width / patch_width, 0 = square (vit_3d.py:80-81 takes pairs)
static inline int img_w(const nv_vit_config* c) { return c->image_width > 0 ? c->image_width : c->image_size; }
static inline int pat_w(const nv_vit_config* c) { return c->patch_width > 0 ? c->patch_width : c->image_patch_size; }

int make_dims(const nv_vit_config* c, int B, Dims& D) {
  NV_CHECK_ARG(c && B > 0, "nv_vit: null config or B <= 0");
  NV_CHECK_ARG(c->image_size > 0 && c->image_patch_size > 0 && c->frames > 0 && c->frame_patch_size > 0 &&
                   c->image_size % c->image_patch_size == 0 && c->image_width >= 0 && c->patch_width >= 0 && img_w(c) % pat_w(c) == 0,
               "Image dimensions must be divisible by the patch size.");
  NV_CHECK_ARG(c->frames % c->frame_patch_size == 0, "Frames must be divisible by frame patch size");
  NV_CHECK_ARG(c->dim_head >= 8 && c->dim_head <= 128 && c->dim_head % 8 == 0,
               "nv_vit: dim_head=%d unsupported (multiples of 8 up to 128; 64 runs the MFMA attention kernels, the others scalar ones)", c->dim_head);
  NV_CHECK_ARG(c->dim % 8 == 0 && c->dim <= 2048 && c->mlp_dim % 8 == 0, "nv_vit: dim must be a multiple of 8 and <= 2048, mlp_dim a multiple of 8");
  NV_CHECK_ARG(c->depth >= 1 && c->num_classes >= 1 && c->channels >= 1, "nv_vit: bad depth/classes/channels");
  D.B = B;
  D.gf = c->frames / c->frame_patch_size;
  D.gh = c->image_size / c->image_patch_size;
  D.gw = img_w(c) / pat_w(c);
  D.N = D.gf * D.gh * D.gw;
  D.n = D.N + 1;
  D.P = c->channels * c->image_patch_size * pat_w(c) * c->frame_patch_size;
  D.Ppad = (int)align_up(D.P, 8);
  D.d = c->dim; D.heads = c->heads; D.dh = c->dim_head; D.inner = c->heads * c->dim_head; D.m = c->mlp_dim;
  D.L = c->depth; D.C = c->num_classes;
  D.M = B * D.n; D.T = B * D.N;
  D.pool_mean = c->pool_mean != 0;
  return NV_OK;
}

// ---- parameter arena layout (reference ViT.state_dict() order, vit_3d.py:91-110) -------------------
struct LayerP { long n1g, n1b, wqkv, wo, bo, n2g, n2b, w1, b1, w2, b2; };
struct ParamTab {
  long pos, cls, pe_g, pe_b, pe_w, pe_bias, pe_g2, pe_b2, hg, hb, hw, hbias, total;
  std::vector<LayerP> layer;
  std::vector<long> offsets, numels;
};

void make_params(const Dims& D, ParamTab& T) {
  long cur = 0;
  auto add = [&](long numel) { const long o = cur; T.offsets.push_back(o); T.numels.push_back(numel); cur = align_up(cur + numel, 8); return o; };
  T.pos = add((long)D.n * D.d); T.cls = add(D.d);
  T.pe_g = add(D.P); T.pe_b = add(D.P); T.pe_w = add((long)D.d * D.P); T.pe_bias = add(D.d); T.pe_g2 = add(D.d); T.pe_b2 = add(D.d);
  T.layer.resize(D.L);
  for (int l = 0; l < D.L; ++l) {
    LayerP& p = T.layer[l];
    p.n1g = add(D.d); p.n1b = add(D.d); p.wqkv = add(3L * D.inner * D.d); p.wo = add((long)D.d * D.inner); p.bo = add(D.d);
    p.n2g = add(D.d); p.n2b = add(D.d); p.w1 = add((long)D.m * D.d); p.b1 = add(D.m); p.w2 = add((long)D.d * D.m); p.b2 = add(D.d);
  }
  T.hg = add(D.d); T.hb = add(D.d); T.hw = add((long)D.C * D.d); T.hbias = add(D.C);
  T.total = cur;
}

// ---- workspace layout -------------------------------------------------------------------------------
struct LayerW { long xn1, st1, qkv, lse, ao, x1, xn2, st2, u, h, x2; };
struct WS {
  long xp, pst, t, est, x0, xh, hst, wpe16, xm;
  long fst1, fst2; // partial row statistics of the LayerNorm-folded inference forward (nv_gemm_resid_ln -> nv_gemm_lnfold): of a block's input / of x1
  long f8x, f8h;   // training layout: transient e4m3 operands of an fp8 training forward (LayerNorm output [M, d], GELU output [M, m])
  std::vector<LayerW> layer;
  // backward scratch
  long g, g16, g16b, dxn, hookg, du, dao, dqkv, delta, dt, dt16, dxp, dwpe, red, red2, red3, cs1;
  long alt[8];   // second copy (odd layers) of g16, g16b, du, dqkv, red, red2, red3, cs1: the auxiliary stream works one layer behind
  long red_bytes, red2_bytes, total;
};

void make_ws(const Dims& D, int training, WS& W) {
  long cur = 0;
  auto add = [&](long bytes) { const long o = cur; cur = align_up(cur + bytes, 256); return o; };
  const long M = D.M, T = D.T, d = D.d;
  W.xp = add(T * D.Ppad * 2); W.pst = add(T * 2 * 4); W.t = add(T * d * 4); W.est = add(T * 2 * 4); W.x0 = add(M * d * 4);
  W.xh = add((long)D.B * d * 4); W.hst = add((long)D.B * 2 * 4);
  W.xm = D.pool_mean ? add((long)D.B * d * 4) : -1;      // pool='mean': token mean of the last block's output
  W.wpe16 = (D.P != D.Ppad) ? add(d * D.Ppad * 2) : -1;
  W.fst1 = add(nv_ln_fold_stats_floats((int)M, (int)d) * 4); W.fst2 = add(nv_ln_fold_stats_floats((int)M, (int)d) * 4);
  const int nl = training ? D.L : 1;     // inference: every layer reuses one set of buffers (x ping-pongs x1 <-> x2/x0)
  W.layer.resize(D.L);
  for (int l = 0; l < nl; ++l) {
    LayerW& w = W.layer[l];
    w.xn1 = add(M * d * 2); w.st1 = add(M * 2 * 4); w.qkv = add(M * 3 * D.inner * 2); w.lse = add((long)D.B * D.heads * D.n * 4);
    w.ao = add(M * D.inner * 2); w.x1 = add(M * d * 4); w.xn2 = add(M * d * 2); w.st2 = add(M * 2 * 4);
    w.u = add(M * D.m * 2); w.h = add(M * D.m * 2); w.x2 = add(M * d * 4);
  }
  for (int l = nl; l < D.L; ++l) W.layer[l] = W.layer[0];
  W.f8x = W.f8h = -1;
  if (training) {
    W.g = add(M * d * 4); W.g16 = add(M * d * 2); W.dxn = add(M * d * 4); W.hookg = add(M * d * 4); W.du = add(M * D.m * 2);
    W.dao = add(M * D.inner * 2); W.dqkv = add(M * 3 * D.inner * 2); W.delta = add((long)D.B * D.heads * D.n * 4);
    W.dt = add(T * d * 4); W.dt16 = add(T * d * 2); W.dxp = add(T * D.Ppad * 4);
    W.dwpe = (D.P != D.Ppad) ? add(d * D.Ppad * 4) : -1;
    long r = nv_ln_bwd_workspace_bytes(D.M, D.d);
    const long r2 = nv_patch_ln_bwd_workspace_bytes(D.T, D.P), r3 = nv_head_step_workspace_bytes(D.B, D.d),
               r4 = nv_colsum_workspace_bytes(D.M, D.m);
    r = r > r2 ? r : r2; r = r > r3 ? r : r3; r = r > r4 ? r : r4;
    W.red_bytes = r; W.red = add(r);
    W.g16b = add(M * d * 2);
    W.red2_bytes = r4 > r2 ? r4 : r2; W.red2 = add(W.red2_bytes);       // reduction scratch of the auxiliary stream (column sums, patch-LN backward)
    W.red3 = add(nv_ln_bwd_workspace_bytes(D.M, D.d));   // LN1-backward partials (reduced on the auxiliary stream one layer late)
    W.alt[0] = add(M * d * 2); W.alt[1] = add(M * d * 2); W.alt[2] = add(M * D.m * 2); W.alt[3] = add(M * 3 * D.inner * 2);
    W.alt[4] = add(W.red_bytes); W.alt[5] = add(W.red2_bytes); W.alt[6] = add(nv_ln_bwd_workspace_bytes(D.M, D.d));
    const long cs1_bytes = (long)((M + 63) / 64) * D.m * 4;          // per-tile column sums of dU (bias gradient of FC1), <= M / 64 tile rows
    W.cs1 = add(cs1_bytes); W.alt[7] = add(cs1_bytes);
    W.f8x = add(M * d); W.f8h = add(M * (long)D.m);
  } else {
    W.g = W.g16 = W.dxn = W.hookg = W.du = W.dao = W.dqkv = W.delta = W.dt = W.dt16 = W.dxp = W.dwpe = W.red = W.g16b = W.red2 = W.red3 = W.cs1 = -1;
    for (int i = 0; i < 8; ++i) W.alt[i] = -1;
    W.red_bytes = W.red2_bytes = 0;
  }
  W.total = cur;
}

// ---- workspace of the fp32 inference path (precise.hip; `training` = 2 in the workspace queries): every activation fp32, one set of
// layer buffers shared by all layers (x ping-pongs x1 <-> x2 / x0 as in the bf16 inference layout)
struct WSF { long xp, pst, t, est, x0, xh, hst, xm, xn1, qkv, ao, x1, xn2, h, x2, total; };
void make_wsf(const Dims& D, WSF& W) {
  long cur = 0;
  auto add = [&](long bytes) { const long o = cur; cur = align_up(cur + bytes, 256); return o; };
  const long M = D.M, T = D.T, d = D.d;
  W.xp = add(T * D.P * 4); W.pst = add(T * 2 * 4); W.t = add(T * d * 4); W.est = add(T * 2 * 4); W.x0 = add(M * d * 4);
  W.xh = add((long)D.B * d * 4); W.hst = add((long)D.B * 2 * 4);
  W.xm = D.pool_mean ? add((long)D.B * d * 4) : -1;
  W.xn1 = add(M * d * 4); W.qkv = add(M * 3 * D.inner * 4); W.ao = add(M * D.inner * 4); W.x1 = add(M * d * 4);
  W.xn2 = add(M * d * 4); W.h = add(M * D.m * 4); W.x2 = add(M * d * 4);
  W.total = cur;
}

// per-site dropout seeds: site = 4*layer + {0 attention probs, 1 to_out, 2 FF hidden, 3 FF out}; 4*depth = embedding
inline unsigned long site_seed(unsigned long seed, int site) { return seed ^ (0x9E3779B97F4A7C15ul * (unsigned long)(site + 1)); }

// Fork/join between the main stream and the auxiliary stream that runs the weight-gradient GEMMs (they depend only on
// buffers the data-gradient chain has already produced, and the chain's kernels - 198-tile GEMMs, attention backward,
// LayerNorm backward - leave LDS and CUs idle).  Events are pooled (created once, outside any capture).
inline int stream_sync(hipStream_t from, hipStream_t to) { return nv_stream_sync((void*)from, (void*)to); }
// deferred join: record a point on the auxiliary stream now, make the main stream wait for it later
inline hipEvent_t deferred_event() {
  static std::vector<hipEvent_t> pool;
  static size_t next = 0;
  if (pool.size() < 32) {
    hipEvent_t e;
    if (hipEventCreateWithFlags(&e, nv_sync_event_flags()) != hipSuccess) return nullptr;
    pool.push_back(e);
    return e;
  }
  return pool[next++ % pool.size()];
}

// nv_vit_backward_stages(join_aux = 0) leaves the auxiliary stream running; the next call on the same workspace must order
// its first buffer reuse after that work: the event recorded at the end of the unjoined call is carried over, PER WORKSPACE
// (two encoders whose staged backward calls interleave each keep their own dependency), on an event of its own (never one of
// the pooled events above, which another call may have re-recorded by then).
struct Carry { hipEvent_t ev = nullptr; bool pending = false; };
std::unordered_map<void*, Carry>& carry_map() { static std::unordered_map<void*, Carry> m; return m; }
hipEvent_t carry_take(void* workspace) {                 // the pending dependency of `workspace`, consumed
  auto it = carry_map().find(workspace);
  if (it == carry_map().end() || !it->second.pending) return nullptr;
  it->second.pending = false;
  return it->second.ev;
}
int carry_record(void* workspace, hipStream_t on) {
  Carry& c = carry_map()[workspace];
  if (!c.ev && hipEventCreateWithFlags(&c.ev, nv_sync_event_flags()) != hipSuccess) { nv_set_error("nv_vit_backward: event create failed"); return NV_ERR_HIP; }
  if (hipEventRecord(c.ev, on) != hipSuccess) { nv_set_error("nv_vit_backward: event record failed"); return NV_ERR_HIP; }
  c.pending = true;
  return NV_OK;
}

#define RUN(call)            \
  do {                       \
    const int rc__ = (call); \
    if (rc__) return rc__;   \
  } while (0)

}  // namespace

extern "C" long nv_vit_param_count(const nv_vit_config* cfg) {
  Dims D; if (make_dims(cfg, 1, D)) return -1;
  ParamTab T; make_params(D, T);
  return T.total;
}

extern "C" int nv_vit_param_table(const nv_vit_config* cfg, long* offsets, long* numels, int max_entries) {
  Dims D; if (make_dims(cfg, 1, D)) return NV_ERR_ARG;
  ParamTab T; make_params(D, T);
  const int cnt = (int)T.offsets.size();
  for (int i = 0; i < cnt && i < max_entries; ++i) { offsets[i] = T.offsets[i]; numels[i] = T.numels[i]; }
  return cnt;
}

extern "C" long nv_vit_workspace_bytes(const nv_vit_config* cfg, int B, int training) {
  Dims D; if (make_dims(cfg, B, D)) return -1;
  if (training == 2) { WSF F; make_wsf(D, F); return F.total; }
  WS W; make_ws(D, training, W);
  return W.total;
}

extern "C" long nv_vit_workspace_offset(const nv_vit_config* cfg, int B, int training, const char* name, int layer) {
  Dims D; if (make_dims(cfg, B, D)) return -1;
  if (training == 2) {        // fp32 inference layout: one set of layer buffers (they hold the LAST layer's values after a forward)
    WSF F; make_wsf(D, F);
    if (layer >= D.L) return -1;
    if (!strcmp(name, "xn1")) return F.xn1; if (!strcmp(name, "qkv")) return F.qkv; if (!strcmp(name, "ao")) return F.ao;
    if (!strcmp(name, "x1")) return F.x1; if (!strcmp(name, "xn2")) return F.xn2; if (!strcmp(name, "h")) return F.h;
    if (!strcmp(name, "x2")) return (layer >= 0 && (layer & 1)) ? F.x0 : F.x2;
    if (!strcmp(name, "xp")) return F.xp; if (!strcmp(name, "t")) return F.t; if (!strcmp(name, "x0")) return F.x0; if (!strcmp(name, "xh")) return F.xh;
    return -1;
  }
  WS W; make_ws(D, training, W);
  if (layer >= 0) {
    if (layer >= D.L) return -1;
    const LayerW& w = W.layer[layer];
    if (!strcmp(name, "xn1")) return w.xn1; if (!strcmp(name, "st1")) return w.st1; if (!strcmp(name, "qkv")) return w.qkv;
    if (!strcmp(name, "lse")) return w.lse; if (!strcmp(name, "ao")) return w.ao; if (!strcmp(name, "x1")) return w.x1;
    if (!strcmp(name, "xn2")) return w.xn2; if (!strcmp(name, "st2")) return w.st2; if (!strcmp(name, "u")) return w.u;
    if (!strcmp(name, "h")) return w.h; if (!strcmp(name, "x2")) return w.x2;
    return -1;
  }
  if (!strcmp(name, "xp")) return W.xp; if (!strcmp(name, "pst")) return W.pst; if (!strcmp(name, "t")) return W.t;
  if (!strcmp(name, "est")) return W.est; if (!strcmp(name, "x0")) return W.x0; if (!strcmp(name, "xh")) return W.xh;
  if (!strcmp(name, "g")) return W.g; if (!strcmp(name, "hookg")) return W.hookg; if (!strcmp(name, "dqkv")) return W.dqkv;
  if (!strcmp(name, "dt")) return W.dt; if (!strcmp(name, "dxp")) return W.dxp;
  return -1;
}

// Patch front end shared by the bf16 and fp8 forwards: the three input forms of nv_vit_input (see the header).
static int patch_front(const nv_vit_config* cfg, const Dims& D, const ParamTab& T, int B, const float* video, const long* strides5, const nv_vit_input* in,
                       const float* p, float eps, void* xp, float* pst, void* stream) {
  const float* sigma = in ? in->vol_sigma : nullptr;
  if (in && in->time_points > 0) {
    NV_CHECK_ARG(B % in->time_points == 0 && cfg->channels == 1, "nv_vit_forward: time_points=%d must divide B=%d (channels = 1)", in->time_points, B);
    // video = contiguous [B / T, H, W, D, T]: (H, W, D) of the dataset = (image, image, frames) of the ViT (NeuroEncoder.py:200-202)
    return nv_patch_ln_fwd_4d(video, B / in->time_points, cfg->image_size, img_w(cfg), cfg->frames, in->time_points, cfg->image_patch_size,
                              pat_w(cfg), cfg->frame_patch_size, p + T.pe_g, p + T.pe_b, eps, xp, D.Ppad, pst, pst + D.T, sigma, stream);
  }
  return nv_patch_ln_fwd(video, strides5, B, cfg->channels, cfg->frames, cfg->image_size, img_w(cfg), cfg->image_patch_size,
                         pat_w(cfg), cfg->frame_patch_size, p + T.pe_g, p + T.pe_b, eps, xp, D.Ppad, pst, pst + D.T, sigma, stream);
}

extern "C" int nv_vit_forward(const nv_vit_config* cfg, int B, const float* video, const long* shape5, const long* strides5,
                              const float* params, const void* params16, void* workspace, long ws_bytes, int training, float drop_p, float emb_drop_p,
                              unsigned long drop_seed, float* logits, void* stream) {
  return nv_vit_forward_in(cfg, B, video, shape5, strides5, nullptr, params, params16, workspace, ws_bytes, training, drop_p, emb_drop_p, drop_seed, logits, stream);
}

static int forward_in_impl(const nv_vit_config* cfg, int B, const float* video, const long* shape5, const long* strides5, const nv_vit_input* in,
                           const float* params, const void* params16, void* workspace, long ws_bytes, int training, float drop_p, float emb_drop_p,
                           unsigned long drop_seed, float* logits, void* stream, bool skip_head, const void* fold16 = nullptr, const float* fold32 = nullptr);

// ---- LayerNorm folded into the GEMMs around it (inference forwards; gemm_common.h EPI_BIAS_RESID_LN / EPI_LNFOLD_*).
// fold16: 16-bit arena with the parameter arena's element offsets holding W_qkv diag(gamma1) and W_1 diag(gamma2) of every block; fold32: per block
// [colsum qkv (3 inner) | folded bias qkv (3 inner) | colsum FC1 (m) | folded bias FC1 (m)].  Prepared once per parameter state (nv_vit_lnfold_prepare).
extern "C" long nv_vit_lnfold_floats(const nv_vit_config* cfg) {
  Dims D; if (make_dims(cfg, 1, D)) return -1;
  return (long)D.L * (6L * D.inner + 2L * D.m);
}
extern "C" int nv_vit_lnfold_prepare(const nv_vit_config* cfg, const float* params, void* fold16, float* fold32, void* stream) {
  Dims D; RUN(make_dims(cfg, 1, D));
  ParamTab T; make_params(D, T);
  NV_CHECK_ARG(params && fold16 && fold32 && nv_aligned16(params) && nv_aligned16(fold16) && nv_aligned16(fold32), "nv_vit_lnfold_prepare: null / unaligned arena");
  r16* f16 = (r16*)fold16;
  const long per = 6L * D.inner + 2L * D.m;
  for (int l = 0; l < D.L; ++l) {
    const LayerP& q = T.layer[l];
    float* f = fold32 + l * per;
    RUN(nv_ln_fold_weight(params + q.wqkv, D.d, 3 * D.inner, D.d, params + q.n1g, params + q.n1b, nullptr, f16 + q.wqkv, D.d, f, f + 3L * D.inner, stream));
    RUN(nv_ln_fold_weight(params + q.w1, D.d, D.m, D.d, params + q.n2g, params + q.n2b, params + q.b1, f16 + q.w1, D.d, f + 6L * D.inner, f + 6L * D.inner + D.m, stream));
  }
  return NV_OK;
}
extern "C" int nv_vit_forward_lnfold(const nv_vit_config* cfg, int B, const float* video, const long* shape5, const long* strides5, const nv_vit_input* in,
                                     const float* params, const void* params16, const void* fold16, const float* fold32, void* workspace, long ws_bytes,
                                     float* logits, void* stream) {
  NV_CHECK_ARG(fold16 && fold32 && nv_aligned16(fold16) && nv_aligned16(fold32), "nv_vit_forward_lnfold: null / unaligned fold arenas (nv_vit_lnfold_prepare)");
  return forward_in_impl(cfg, B, video, shape5, strides5, in, params, params16, workspace, ws_bytes, 0, 0.f, 0.f, 0, logits, stream, false, fold16, fold32);
}

extern "C" int nv_vit_forward_in(const nv_vit_config* cfg, int B, const float* video, const long* shape5, const long* strides5, const nv_vit_input* in,
                                 const float* params, const void* params16, void* workspace, long ws_bytes, int training, float drop_p, float emb_drop_p,
                                 unsigned long drop_seed, float* logits, void* stream) {
  return forward_in_impl(cfg, B, video, shape5, strides5, in, params, params16, workspace, ws_bytes, training, drop_p, emb_drop_p, drop_seed, logits, stream, false);
}

// skip_head: everything up to the last block's output; the caller runs the head itself (nv_vit_train_step: nv_head_step)
static int forward_in_impl(const nv_vit_config* cfg, int B, const float* video, const long* shape5, const long* strides5, const nv_vit_input* in,
                           const float* params, const void* params16, void* workspace, long ws_bytes, int training, float drop_p, float emb_drop_p,
                           unsigned long drop_seed, float* logits, void* stream, bool skip_head, const void* fold16, const float* fold32) {
  Dims D; RUN(make_dims(cfg, B, D));
  ParamTab T; make_params(D, T);
  WS W; make_ws(D, training, W);
  NV_CHECK_ARG(video && shape5 && strides5 && params && params16 && workspace && logits, "nv_vit_forward: null pointer");
  if (in && in->time_points > 0)
    NV_CHECK_ARG(shape5[0] * shape5[4] == B && shape5[4] == in->time_points && shape5[1] == cfg->image_size && shape5[2] == img_w(cfg) && shape5[3] == cfg->frames,
                 "nv_vit_forward: 4D input is [%ld,%ld,%ld,%ld,%ld], expected [B/T, %d, %d, %d, T=%d] with B = %d", shape5[0], shape5[1], shape5[2], shape5[3],
                 shape5[4], cfg->image_size, img_w(cfg), cfg->frames, in->time_points, B);
  else
  NV_CHECK_ARG(shape5[0] == B && shape5[1] == cfg->channels && shape5[2] == cfg->frames && shape5[3] == cfg->image_size && shape5[4] == img_w(cfg),
               "nv_vit_forward: video is [%ld,%ld,%ld,%ld,%ld], the model was built for [%d,%d,%d,%d,%d] (B, channels, frames, height, width)",
               shape5[0], shape5[1], shape5[2], shape5[3], shape5[4], B, cfg->channels, cfg->frames, cfg->image_size, img_w(cfg));
  NV_CHECK_ARG(ws_bytes >= W.total, "nv_vit_forward: workspace too small (%ld < %ld)", ws_bytes, W.total);
  NV_CHECK_ARG(((uintptr_t)workspace & 255) == 0 && nv_aligned16(params) && nv_aligned16(params16), "nv_vit_forward: alignment");
  char* ws = (char*)workspace;
  const float* p = params;
  const r16* p16 = (const r16*)params16;
  const float eps = cfg->ln_eps;
  const int M = D.M, d = D.d;

  // A1+A2: gather + LayerNorm(patch_dim) -> bf16
  float* pst = (float*)(ws + W.pst);
  RUN(patch_front(cfg, D, T, B, video, strides5, in, p, eps, ws + W.xp, pst, stream));
  // A3: Linear(patch_dim, dim)
  const void* wpe = p16 + T.pe_w;
  if (D.P != D.Ppad) {
    RUN(nv_cast_bf16_2d(p + T.pe_w, D.P, d, D.P, ws + W.wpe16, D.Ppad, stream));
    wpe = ws + W.wpe16;
  }
  RUN(nv_gemm_bf16(0, 2, D.T, d, D.Ppad, ws + W.xp, D.Ppad, wpe, D.Ppad, ws + W.t, d, p + T.pe_bias, nullptr, 0, nullptr, 0, 0, 1.f, 0, 0.f, stream));
  // A4+A5: LayerNorm(dim) + cls + pos
  float* est = (float*)(ws + W.est);
  RUN(nv_embed_finish_fwd((float*)(ws + W.t), d, B, D.N, d, p + T.pe_g2, p + T.pe_b2, eps, p + T.pos, p + T.cls, (float*)(ws + W.x0), d, est,
                          est + D.T, site_seed(drop_seed, 4 * D.L), emb_drop_p, stream));

  const float scale = 1.0f / sqrtf((float)D.dh);
  const float* xin = (float*)(ws + W.x0);
  const bool tail = cls_tail_wanted(D, training, drop_p, in ? in->rows_form : 0);
  // LayerNorm folded into the GEMMs (inference, no dropout, every one of a block's four shapes on an LDS-epilogue kernel).  Not folded: LN1 of block 0 (its
  // input comes out of embed_finish, not out of a GEMM epilogue), LN1 of the LAST block (its output is the Grad-CAM hook tensor, NeuroEncoder.py:70-75: it
  // must exist), and a last block that runs on its cls rows.  ViT3D-base: 21 of the 24 LayerNorm launches of a forward.
  // (measured, ViT3D-base forward-only, same box, folded / plain: batch 4 +6.5 %, 8 +2.9 %, 20 +1.9 %, 64 +0.6 %; ViT3D-large (16 388 rows: its qkv would leave the
  //  256 x 256 kernel) -0.6 %: beyond 12 288 rows the plain launches stay)
  const bool fold = fold16 && fold32 && !training && drop_p == 0.f && M <= 12288 && nv_gemm_lnfold_supported(M, 3 * D.inner, d) && nv_gemm_lnfold_supported(M, D.m, d) &&
                    nv_gemm_lnfold_supported(M, d, D.inner) && nv_gemm_lnfold_supported(M, d, D.m);
  const r16* f16 = (const r16*)fold16;
  const long fper = 6L * D.inner + 2L * D.m;
  float* fst1 = (float*)(ws + W.fst1);
  float* fst2 = (float*)(ws + W.fst2);
  bool have_in16 = false;            // the previous block's FC2 left this block's input in the operand format (in xn1) with its statistics (fst1)
  for (int l = 0; l < D.L; ++l) {
    const LayerP& q = T.layer[l];
    const LayerW& w = W.layer[l];
    const float* ff = fold ? fold32 + l * fper : nullptr;
    // inference ping-pong: layer output goes to x2, except that x2 would alias the next layer's output -> alternate x0/x2
    float* x1 = (float*)(ws + w.x1);
    float* x2 = (float*)(ws + ((!training && (l & 1)) ? W.x0 : w.x2));
    float* st1 = (float*)(ws + w.st1);
    float* st2 = (float*)(ws + w.st2);
    if (have_in16) {
      RUN(nv_gemm_lnfold(0, M, 3 * D.inner, d, ws + w.xn1, d, f16 + q.wqkv, d, fst1, ff, ff + 3L * D.inner, eps, ws + w.qkv, 3 * D.inner, stream));
    } else {
      RUN(nv_ln_fwd(xin, d, M, d, p + q.n1g, p + q.n1b, eps, ws + w.xn1, d, st1, st1 + M, stream));
      RUN(nv_gemm_bf16(0, 0, M, 3 * D.inner, d, ws + w.xn1, d, p16 + q.wqkv, d, ws + w.qkv, 3 * D.inner, nullptr, nullptr, 0, nullptr, 0, 0, 1.f, 0, 0.f, stream));
    }
    have_in16 = false;
    RUN(nv_attn_fwd(ws + w.qkv, 3 * D.inner, B, D.n, D.heads, D.dh, scale, ws + w.ao, D.inner, (float*)(ws + w.lse), site_seed(drop_seed, 4 * l + 0), drop_p, stream));
    if (tail && l == D.L - 1) {
      // cls rows only (row b of the small problem = row b * n of the buffers); LN2 statistics land at st2[0 .. B) / st2[M .. M + B)
      const long rs = D.n;
      RUN(nv_skinny_nt(0, B, d, D.inner, ws + w.ao, D.inner * rs, p16 + q.wo, D.inner, p + q.bo, xin, d * rs, x1, d * rs, nullptr, 0, site_seed(drop_seed, 4 * l + 1), proj_drop_p(cfg, drop_p), stream));
      RUN(nv_ln_fwd(x1, d * rs, B, d, p + q.n2g, p + q.n2b, eps, ws + w.xn2, d * rs, st2, st2 + M, stream));
      RUN(nv_skinny_nt(1, B, D.m, d, ws + w.xn2, d * rs, p16 + q.w1, d, p + q.b1, nullptr, 0, ws + w.h, D.m * rs, training ? ws + w.u : nullptr, D.m * rs, site_seed(drop_seed, 4 * l + 2), drop_p, stream));
      RUN(nv_skinny_nt(0, B, d, D.m, ws + w.h, D.m * rs, p16 + q.w2, D.m, p + q.b2, x1, d * rs, x2, d * rs, nullptr, 0, site_seed(drop_seed, 4 * l + 3), drop_p, stream));
      xin = x2;
      continue;
    }
    if (fold) {
      // out-projection writes x1 (f32), x1 in the operand format (into the xn2 buffer) and its row statistics; FC1 contracts those rows with W1 diag(gamma2)
      RUN(nv_gemm_resid_ln(M, d, D.inner, ws + w.ao, D.inner, p16 + q.wo, D.inner, p + q.bo, xin, d, x1, d, ws + w.xn2, d, fst2, stream));
      RUN(nv_gemm_lnfold(1, M, D.m, d, ws + w.xn2, d, f16 + q.w1, d, fst2, ff + 6L * D.inner, ff + 6L * D.inner + D.m, eps, ws + w.h, D.m, stream));
      const bool next_folds = l + 1 <= D.L - 2;          // the next block's LN1 (blocks 1 .. L-2)
      if (next_folds) {
        RUN(nv_gemm_resid_ln(M, d, D.m, ws + w.h, D.m, p16 + q.w2, D.m, p + q.b2, x1, d, x2, d, ws + W.layer[l + 1].xn1, d, fst1, stream));
        have_in16 = true;
      } else {
        RUN(nv_gemm_bf16(0, 4, M, d, D.m, ws + w.h, D.m, p16 + q.w2, D.m, x2, d, p + q.b2, x1, d, nullptr, 0, 0, 1.f, 0, 0.f, stream));
      }
      xin = x2;
      continue;
    }
    RUN(nv_gemm_bf16(0, 4, M, d, D.inner, ws + w.ao, D.inner, p16 + q.wo, D.inner, x1, d, p + q.bo, xin, d, nullptr, 0, 0, 1.f, site_seed(drop_seed, 4 * l + 1), proj_drop_p(cfg, drop_p), stream));
    RUN(nv_ln_fwd(x1, d, M, d, p + q.n2g, p + q.n2b, eps, ws + w.xn2, d, st2, st2 + M, stream));
    RUN(nv_gemm_bf16(0, 3, M, D.m, d, ws + w.xn2, d, p16 + q.w1, d, ws + w.h, D.m, p + q.b1, nullptr, 0, training ? ws + w.u : nullptr, D.m, 0, 1.f, site_seed(drop_seed, 4 * l + 2), drop_p, stream));
    RUN(nv_gemm_bf16(0, 4, M, d, D.m, ws + w.h, D.m, p16 + q.w2, D.m, x2, d, p + q.b2, x1, d, nullptr, 0, 0, 1.f, site_seed(drop_seed, 4 * l + 3), drop_p, stream));
    xin = x2;
  }
  if (skip_head) return NV_OK;
  // A9: cls pooling + LayerNorm + Linear(dim, C)
  const float* pooled = xin;
  long pooled_stride = (long)D.n * d;
  if (D.pool_mean) {
    RUN(nv_token_mean(xin, B, D.n, d, (float*)(ws + W.xm), stream));
    pooled = (float*)(ws + W.xm); pooled_stride = d;
  }
  RUN(nv_head_fwd(pooled, pooled_stride, B, d, p + T.hg, p + T.hb, eps, p + T.hw, p + T.hbias, D.C, (float*)(ws + W.xh), (float*)(ws + W.hst),
                  logits, stream));
  return NV_OK;
}

// ---- fp32 inference forward (precise.hip): what the reference's fp32 validate computes (Trainer.py:101-118), every operand fp32,
// contractions on the fp32 MFMA.  Weights come from the fp32 parameter arena itself; no shadow arena, no dropout (eval mode).
extern "C" int nv_vit_forward_f32(const nv_vit_config* cfg, int B, const float* video, const long* shape5, const long* strides5, const nv_vit_input* in,
                                  const float* params, void* workspace, long ws_bytes, float* logits, void* stream) {
  Dims D; RUN(make_dims(cfg, B, D));
  ParamTab T; make_params(D, T);
  WSF W; make_wsf(D, W);
  NV_CHECK_ARG(video && shape5 && strides5 && params && workspace && logits, "nv_vit_forward_f32: null pointer");
  if (in && in->time_points > 0)
    NV_CHECK_ARG(shape5[0] * shape5[4] == B && shape5[4] == in->time_points && shape5[1] == cfg->image_size && shape5[2] == img_w(cfg) && shape5[3] == cfg->frames &&
                     B % in->time_points == 0 && cfg->channels == 1,
                 "nv_vit_forward_f32: 4D input is [%ld,%ld,%ld,%ld,%ld], expected [B/T, %d, %d, %d, T=%d] with B = %d", shape5[0], shape5[1], shape5[2], shape5[3],
                 shape5[4], cfg->image_size, img_w(cfg), cfg->frames, in->time_points, B);
  else
    NV_CHECK_ARG(shape5[0] == B && shape5[1] == cfg->channels && shape5[2] == cfg->frames && shape5[3] == cfg->image_size && shape5[4] == img_w(cfg),
                 "nv_vit_forward_f32: video is [%ld,%ld,%ld,%ld,%ld], the model was built for [%d,%d,%d,%d,%d] (B, channels, frames, height, width)",
                 shape5[0], shape5[1], shape5[2], shape5[3], shape5[4], B, cfg->channels, cfg->frames, cfg->image_size, img_w(cfg));
  NV_CHECK_ARG(ws_bytes >= W.total, "nv_vit_forward_f32: workspace too small (%ld < %ld)", ws_bytes, W.total);
  NV_CHECK_ARG(((uintptr_t)workspace & 255) == 0 && nv_aligned16(params), "nv_vit_forward_f32: alignment");
  char* ws = (char*)workspace;
  const float* p = params;
  const float eps = cfg->ln_eps;
  const int M = D.M, d = D.d;
  auto F32 = [&](long off) -> float* { return (float*)(ws + off); };

  // A1 + A2: gather + LayerNorm(patch_dim) -> fp32 tokens [T, P]
  float* pst = F32(W.pst);
  const float* sigma = in ? in->vol_sigma : nullptr;
  if (in && in->time_points > 0)
    RUN(nv_patch_ln_fwd_4d_f32(video, B / in->time_points, cfg->image_size, img_w(cfg), cfg->frames, in->time_points, cfg->image_patch_size,
                               pat_w(cfg), cfg->frame_patch_size, p + T.pe_g, p + T.pe_b, eps, F32(W.xp), D.P, pst, pst + D.T, sigma, stream));
  else
    RUN(nv_patch_ln_fwd_f32(video, strides5, B, cfg->channels, cfg->frames, cfg->image_size, img_w(cfg), cfg->image_patch_size,
                            pat_w(cfg), cfg->frame_patch_size, p + T.pe_g, p + T.pe_b, eps, F32(W.xp), D.P, pst, pst + D.T, sigma, stream));
  // A3: Linear(patch_dim, dim)
  RUN(nv_gemm_f32(2, D.T, d, D.P, F32(W.xp), D.P, p + T.pe_w, D.P, F32(W.t), d, p + T.pe_bias, nullptr, 0, stream));
  // A4 + A5: LayerNorm(dim) + cls + pos
  float* est = F32(W.est);
  RUN(nv_embed_finish_fwd(F32(W.t), d, B, D.N, d, p + T.pe_g2, p + T.pe_b2, eps, p + T.pos, p + T.cls, F32(W.x0), d, est, est + D.T, 0, 0.f, stream));
  const float scale = 1.0f / sqrtf((float)D.dh);
  const float* xin = F32(W.x0);
  for (int l = 0; l < D.L; ++l) {
    const LayerP& q = T.layer[l];
    float* x1 = F32(W.x1);
    float* x2 = F32((l & 1) ? W.x0 : W.x2);
    RUN(nv_ln_fwd_f32(xin, d, M, d, p + q.n1g, p + q.n1b, eps, F32(W.xn1), d, nullptr, nullptr, stream));
    RUN(nv_gemm_f32(0, M, 3 * D.inner, d, F32(W.xn1), d, p + q.wqkv, d, F32(W.qkv), 3 * D.inner, nullptr, nullptr, 0, stream));
    RUN(nv_attn_fwd_f32(F32(W.qkv), 3 * D.inner, B, D.n, D.heads, D.dh, scale, F32(W.ao), D.inner, stream));
    if (l == D.L - 1 && cls_tail_wanted(D, 0, 0.f, in ? in->rows_form : 0)) {
      // pool = 'cls' (NeuroEncoder.py:194): behind the last attention only the B cls rows reach the head - the last block's
      // out-projection, LayerNorm and FeedForward run on those rows as strided views (row stride n); same values for the logits
      const long rs = D.n;
      RUN(nv_gemm_f32(4, B, d, D.inner, F32(W.ao), D.inner * rs, p + q.wo, D.inner, x1, d * rs, p + q.bo, xin, d * rs, stream));
      RUN(nv_ln_fwd_f32(x1, d * rs, B, d, p + q.n2g, p + q.n2b, eps, F32(W.xn2), d * rs, nullptr, nullptr, stream));
      RUN(nv_gemm_f32(3, B, D.m, d, F32(W.xn2), d * rs, p + q.w1, d, F32(W.h), D.m * rs, p + q.b1, nullptr, 0, stream));
      RUN(nv_gemm_f32(4, B, d, D.m, F32(W.h), D.m * rs, p + q.w2, D.m, x2, d * rs, p + q.b2, x1, d * rs, stream));
      xin = x2;
      continue;
    }
    RUN(nv_gemm_f32(4, M, d, D.inner, F32(W.ao), D.inner, p + q.wo, D.inner, x1, d, p + q.bo, xin, d, stream));
    RUN(nv_ln_fwd_f32(x1, d, M, d, p + q.n2g, p + q.n2b, eps, F32(W.xn2), d, nullptr, nullptr, stream));
    RUN(nv_gemm_f32(3, M, D.m, d, F32(W.xn2), d, p + q.w1, d, F32(W.h), D.m, p + q.b1, nullptr, 0, stream));
    RUN(nv_gemm_f32(4, M, d, D.m, F32(W.h), D.m, p + q.w2, D.m, x2, d, p + q.b2, x1, d, stream));
    xin = x2;
  }
  const float* pooled = xin;
  long pooled_stride = (long)D.n * d;
  if (D.pool_mean) {
    RUN(nv_token_mean(xin, B, D.n, d, F32(W.xm), stream));
    pooled = F32(W.xm); pooled_stride = d;
  }
  RUN(nv_head_fwd(pooled, pooled_stride, B, d, p + T.hg, p + T.hb, eps, p + T.hw, p + T.hbias, D.C, F32(W.xh), F32(W.hst), logits, stream));
  return NV_OK;
}

// ---- fp8 inference path (BASELINE.json configs[4]): qkv, FC1 and FC2 of every block on OCP e4m3 operands (92 % of the linear
// FLOPs; the output projection, whose input comes head by head out of the attention kernel, stays bf16, as do the patch embedding,
// attention and the head).  Weights: per-output-row scales; activations (LN outputs, GELU output): one calibrated scale per
// tensor, applied by the kernel that produces them.  Layout of the scale arena: layer l at l * (3*inner + m + d):
// [colscale qkv (3*inner) | colscale FC1 (m) | colscale FC2 (d) | colscale out-projection (d)].  act_scales: HOST array [depth][4] = xn1, xn2, h, ao.
extern "C" long nv_vit_fp8_scale_count(const nv_vit_config* cfg) {
  Dims D; if (make_dims(cfg, 1, D)) return -1;
  return (long)D.L * (3L * D.inner + D.m + 2L * D.d);
}

extern "C" int nv_vit_quantize_fp8(const nv_vit_config* cfg, const float* params, const float* act_scales, void* params8, float* colscales, void* stream) {
  Dims D; RUN(make_dims(cfg, 1, D));
  ParamTab T; make_params(D, T);
  NV_CHECK_ARG(params && act_scales && params8 && colscales, "nv_vit_quantize_fp8: null pointer");
  // heads * dim_head is the K of the out-projection only: it matters when some block runs that linear in e4m3 (act scale > 0)
  bool out_proj8 = false;
  for (int l = 0; l < D.L; ++l) out_proj8 = out_proj8 || act_scales[4 * l + 3] > 0.f;
  NV_CHECK_ARG(D.d % 128 == 0 && D.m % 128 == 0 && (!out_proj8 || D.inner % 128 == 0),
               "nv_vit_quantize_fp8: the fp8 GEMM needs dim and mlp_dim - and heads * dim_head when the out-projection runs in e4m3 - to be multiples of 128 (got %d, %d, %d)", D.d, D.m, D.inner);
  char* p8 = (char*)params8;
  const long per = 3L * D.inner + D.m + 2L * D.d;
  for (int l = 0; l < D.L; ++l) {
    const LayerP& q = T.layer[l];
    float* cs = colscales + l * per;
    RUN(nv_quant_rows_f8(params + q.wqkv, D.d, 3 * D.inner, D.d, p8 + q.wqkv, D.d, act_scales[4 * l + 0], cs, stream));
    RUN(nv_quant_rows_f8(params + q.w1, D.d, D.m, D.d, p8 + q.w1, D.d, act_scales[4 * l + 1], cs + 3L * D.inner, stream));
    RUN(nv_quant_rows_f8(params + q.w2, D.m, D.d, D.m, p8 + q.w2, D.m, act_scales[4 * l + 2], cs + 3L * D.inner + D.m, stream));
    if (act_scales[4 * l + 3] > 0.f)                      // <= 0: this block's out-projection stays on bf16 operands
      RUN(nv_quant_rows_f8(params + q.wo, D.inner, D.d, D.inner, p8 + q.wo, D.inner, act_scales[4 * l + 3], cs + 3L * D.inner + D.m + D.d, stream));
  }
  return NV_OK;
}

extern "C" int nv_vit_forward_fp8(const nv_vit_config* cfg, int B, const float* video, const long* shape5, const long* strides5, const nv_vit_input* in,
                                  const float* params, const void* params16, const void* params8, const float* colscales, const float* act_scales, void* workspace,
                                  long ws_bytes, float* logits, void* stream) {
  Dims D; RUN(make_dims(cfg, B, D));
  ParamTab T; make_params(D, T);
  WS W; make_ws(D, 0, W);
  NV_CHECK_ARG(nv_operand_format() == NV_OPERAND_BF16, "nv_vit_forward_fp8: the fp8 path is built beside bf16 operands (nv_set_operand_format(NV_OPERAND_BF16))");
  NV_CHECK_ARG(video && shape5 && strides5 && params && params16 && params8 && colscales && act_scales && workspace && logits, "nv_vit_forward_fp8: null pointer");
  NV_CHECK_ARG((in && in->time_points > 0) || (shape5[0] == B && shape5[1] == cfg->channels && shape5[2] == cfg->frames && shape5[3] == cfg->image_size && shape5[4] == img_w(cfg)),
               "nv_vit_forward_fp8: video is [%ld,%ld,%ld,%ld,%ld], the model was built for [%d,%d,%d,%d,%d] (B, channels, frames, height, width)",
               shape5[0], shape5[1], shape5[2], shape5[3], shape5[4], B, cfg->channels, cfg->frames, cfg->image_size, img_w(cfg));
  NV_CHECK_ARG(ws_bytes >= W.total, "nv_vit_forward_fp8: workspace too small (%ld < %ld)", ws_bytes, W.total);
  NV_CHECK_ARG(D.d % 128 == 0 && D.m % 128 == 0, "nv_vit_forward_fp8: dim and mlp_dim must be multiples of 128");
  NV_CHECK_ARG(((uintptr_t)workspace & 255) == 0 && nv_aligned16(params) && nv_aligned16(params16) && nv_aligned16(params8), "nv_vit_forward_fp8: alignment");
  char* ws = (char*)workspace;
  const float* p = params;
  const r16* p16 = (const r16*)params16;
  const char* p8 = (const char*)params8;
  const float eps = cfg->ln_eps;
  const int M = D.M, d = D.d;
  const long per = 3L * D.inner + D.m + 2L * D.d;

  float* pst = (float*)(ws + W.pst);
  RUN(patch_front(cfg, D, T, B, video, strides5, in, p, eps, ws + W.xp, pst, stream));
  const void* wpe = p16 + T.pe_w;
  if (D.P != D.Ppad) {
    RUN(nv_cast_bf16_2d(p + T.pe_w, D.P, d, D.P, ws + W.wpe16, D.Ppad, stream));
    wpe = ws + W.wpe16;
  }
  RUN(nv_gemm_bf16(0, 2, D.T, d, D.Ppad, ws + W.xp, D.Ppad, wpe, D.Ppad, ws + W.t, d, p + T.pe_bias, nullptr, 0, nullptr, 0, 0, 1.f, 0, 0.f, stream));
  float* est = (float*)(ws + W.est);
  RUN(nv_embed_finish_fwd((float*)(ws + W.t), d, B, D.N, d, p + T.pe_g2, p + T.pe_b2, eps, p + T.pos, p + T.cls, (float*)(ws + W.x0), d, est,
                          est + D.T, 0, 0.f, stream));
  const float scale = 1.0f / sqrtf((float)D.dh);
  const float* xin = (float*)(ws + W.x0);
  const bool tail8 = cls_tail_wanted(D, 0, 0.f, in ? in->rows_form : 0);
  for (int l = 0; l < D.L; ++l) {
    const LayerP& q = T.layer[l];
    const LayerW& w = W.layer[l];
    const float* cs = colscales + l * per;
    const float s_xn1 = act_scales[4 * l], s_xn2 = act_scales[4 * l + 1], s_h = act_scales[4 * l + 2], s_ao = act_scales[4 * l + 3];
    const bool tail_here = tail8 && l == D.L - 1;
    const bool ao8 = !tail_here && D.dh == 64 && s_ao > 0.f && D.inner % 128 == 0;   // the MFMA attention kernels write e4m3 themselves; the cls-rows tail, the generic
                                                                                     // kernels and a heads * dim_head the fp8 GEMM cannot take as K stay bf16
    float* x1 = (float*)(ws + w.x1);
    float* x2 = (float*)(ws + ((l & 1) ? W.x0 : w.x2));
    RUN(nv_ln_fwd_f8(xin, d, M, d, p + q.n1g, p + q.n1b, eps, s_xn1, ws + w.xn1, d, stream));
    RUN(nv_gemm_f8(0, M, 3 * D.inner, d, ws + w.xn1, d, p8 + q.wqkv, d, ws + w.qkv, 3 * D.inner, cs, nullptr, nullptr, 0, 1.f, stream));
    if (ao8) RUN(nv_attn_fwd_o8(ws + w.qkv, 3 * D.inner, B, D.n, D.heads, D.dh, scale, ws + w.ao, D.inner, s_ao, stream));
    else RUN(nv_attn_fwd(ws + w.qkv, 3 * D.inner, B, D.n, D.heads, D.dh, scale, ws + w.ao, D.inner, (float*)(ws + w.lse), 0, 0.f, stream));
    if (tail_here) {
      // the last block's out-projection / LayerNorm / FeedForward on the B cls rows (see g_cls_tail): bf16 operands through the
      // weight-streaming kernels - these few rows gain nothing from fp8 and lose nothing by staying in bf16
      const long rs = D.n;
      float* st = (float*)(ws + w.st2);
      RUN(nv_skinny_nt(0, B, d, D.inner, ws + w.ao, D.inner * rs, p16 + q.wo, D.inner, p + q.bo, xin, d * rs, x1, d * rs, nullptr, 0, 0, 0.f, stream));
      RUN(nv_ln_fwd(x1, d * rs, B, d, p + q.n2g, p + q.n2b, eps, ws + w.xn2, d * rs, st, st + M, stream));
      RUN(nv_skinny_nt(1, B, D.m, d, ws + w.xn2, d * rs, p16 + q.w1, d, p + q.b1, nullptr, 0, ws + w.h, D.m * rs, nullptr, 0, 0, 0.f, stream));
      RUN(nv_skinny_nt(0, B, d, D.m, ws + w.h, D.m * rs, p16 + q.w2, D.m, p + q.b2, x1, d * rs, x2, d * rs, nullptr, 0, 0, 0.f, stream));
      xin = x2;
      continue;
    }
    if (ao8) RUN(nv_gemm_f8(4, M, d, D.inner, ws + w.ao, D.inner, p8 + q.wo, D.inner, x1, d, cs + 3L * D.inner + D.m + D.d, p + q.bo, xin, d, 1.f, stream));
    else RUN(nv_gemm_bf16(0, 4, M, d, D.inner, ws + w.ao, D.inner, p16 + q.wo, D.inner, x1, d, p + q.bo, xin, d, nullptr, 0, 0, 1.f, 0, 0.f, stream));
    RUN(nv_ln_fwd_f8(x1, d, M, d, p + q.n2g, p + q.n2b, eps, s_xn2, ws + w.xn2, d, stream));
    RUN(nv_gemm_f8(7, M, D.m, d, ws + w.xn2, d, p8 + q.w1, d, ws + w.h, D.m, cs + 3L * D.inner, p + q.b1, nullptr, 0, s_h, stream));
    RUN(nv_gemm_f8(4, M, d, D.m, ws + w.h, D.m, p8 + q.w2, D.m, x2, d, cs + 3L * D.inner + D.m, p + q.b2, x1, d, 1.f, stream));
    xin = x2;
  }
  const float* pooled = xin;
  long pooled_stride = (long)D.n * d;
  if (D.pool_mean) {
    RUN(nv_token_mean(xin, B, D.n, d, (float*)(ws + W.xm), stream));
    pooled = (float*)(ws + W.xm); pooled_stride = d;
  }
  RUN(nv_head_fwd(pooled, pooled_stride, B, d, p + T.hg, p + T.hb, eps, p + T.hw, p + T.hbias, D.C, (float*)(ws + W.xh), (float*)(ws + W.hst),
                  logits, stream));
  return NV_OK;
}

// ---- fp8 TRAINING forward (BASELINE.json configs[4] is quoted "fwd / fwd+bwd"): the forward of the train step with qkv / FC1 / FC2 of
// every block on e4m3 operands, writing every buffer of the training layout the (bf16) backward pass reads: LayerNorm outputs and
// statistics (nv_ln_fwd_f8_train: e4m3 + bf16 + stats in one pass), qkv bf16 (the fp8 GEMM's plain store), attention in bf16 with
// its lse, u / h in bf16 beside h in e4m3 (nv_gemm_f8_gelu_train).  The out-projection stays on bf16 operands (its A operand comes
// head by head out of the attention kernel, which would have to write a second copy; 8 % of the linear FLOPs).
extern "C" int nv_vit_forward_fp8_train(const nv_vit_config* cfg, int B, const float* video, const long* shape5, const long* strides5, const nv_vit_input* in,
                                        const float* params, const void* params16, const void* params8, const float* colscales, const float* act_scales,
                                        void* workspace, long ws_bytes, float drop_p, float emb_drop_p, unsigned long drop_seed, float* logits, void* stream) {
  Dims D; RUN(make_dims(cfg, B, D));
  ParamTab T; make_params(D, T);
  WS W; make_ws(D, 1, W);
  NV_CHECK_ARG(video && shape5 && strides5 && params && params16 && params8 && colscales && act_scales && workspace && logits, "nv_vit_forward_fp8_train: null pointer");
  NV_CHECK_ARG(!(in && in->time_points > 0), "nv_vit_forward_fp8_train: the fused 4D input form is forward-only");
  NV_CHECK_ARG(shape5[0] == B && shape5[1] == cfg->channels && shape5[2] == cfg->frames && shape5[3] == cfg->image_size && shape5[4] == img_w(cfg),
               "nv_vit_forward_fp8_train: video is [%ld,%ld,%ld,%ld,%ld], the model was built for [%d,%d,%d,%d,%d] (B, channels, frames, height, width)",
               shape5[0], shape5[1], shape5[2], shape5[3], shape5[4], B, cfg->channels, cfg->frames, cfg->image_size, img_w(cfg));
  NV_CHECK_ARG(ws_bytes >= W.total, "nv_vit_forward_fp8_train: workspace too small (%ld < %ld): the training layout is needed", ws_bytes, W.total);
  NV_CHECK_ARG(D.d % 128 == 0 && D.m % 128 == 0, "nv_vit_forward_fp8_train: dim and mlp_dim must be multiples of 128 (got %d, %d)", D.d, D.m);
  NV_CHECK_ARG(((uintptr_t)workspace & 255) == 0 && nv_aligned16(params) && nv_aligned16(params16) && nv_aligned16(params8), "nv_vit_forward_fp8_train: alignment");
  char* ws = (char*)workspace;
  const float* p = params;
  const r16* p16 = (const r16*)params16;
  const char* p8 = (const char*)params8;
  const float eps = cfg->ln_eps;
  const int M = D.M, d = D.d;
  const long per = 3L * D.inner + D.m + 2L * D.d;

  float* pst = (float*)(ws + W.pst);
  RUN(patch_front(cfg, D, T, B, video, strides5, in, p, eps, ws + W.xp, pst, stream));
  const void* wpe = p16 + T.pe_w;
  if (D.P != D.Ppad) {
    RUN(nv_cast_bf16_2d(p + T.pe_w, D.P, d, D.P, ws + W.wpe16, D.Ppad, stream));
    wpe = ws + W.wpe16;
  }
  RUN(nv_gemm_bf16(0, 2, D.T, d, D.Ppad, ws + W.xp, D.Ppad, wpe, D.Ppad, ws + W.t, d, p + T.pe_bias, nullptr, 0, nullptr, 0, 0, 1.f, 0, 0.f, stream));
  float* est = (float*)(ws + W.est);
  RUN(nv_embed_finish_fwd((float*)(ws + W.t), d, B, D.N, d, p + T.pe_g2, p + T.pe_b2, eps, p + T.pos, p + T.cls, (float*)(ws + W.x0), d, est,
                          est + D.T, site_seed(drop_seed, 4 * D.L), emb_drop_p, stream));
  const float scale = 1.0f / sqrtf((float)D.dh);
  const float* xin = (float*)(ws + W.x0);
  const bool tail = cls_tail_wanted(D, 1, drop_p, in ? in->rows_form : 0);
  void* x8 = ws + W.f8x;
  void* h8 = ws + W.f8h;
  for (int l = 0; l < D.L; ++l) {
    const LayerP& q = T.layer[l];
    const LayerW& w = W.layer[l];
    const float* cs = colscales + l * per;
    const float s_xn1 = act_scales[4 * l], s_xn2 = act_scales[4 * l + 1], s_h = act_scales[4 * l + 2];
    NV_CHECK_ARG(s_xn1 > 0.f && s_xn2 > 0.f && s_h > 0.f, "nv_vit_forward_fp8_train: activation scales of layer %d must be positive", l);
    float* x1 = (float*)(ws + w.x1);
    float* x2 = (float*)(ws + w.x2);
    float* st1 = (float*)(ws + w.st1);
    float* st2 = (float*)(ws + w.st2);
    RUN(nv_ln_fwd_f8_train(xin, d, M, d, p + q.n1g, p + q.n1b, eps, s_xn1, x8, d, ws + w.xn1, d, st1, st1 + M, stream));
    RUN(nv_gemm_f8(0, M, 3 * D.inner, d, x8, d, p8 + q.wqkv, d, ws + w.qkv, 3 * D.inner, cs, nullptr, nullptr, 0, 1.f, stream));
    RUN(nv_attn_fwd(ws + w.qkv, 3 * D.inner, B, D.n, D.heads, D.dh, scale, ws + w.ao, D.inner, (float*)(ws + w.lse), site_seed(drop_seed, 4 * l + 0), drop_p, stream));
    if (tail && l == D.L - 1) {       // the last block on its B cls rows: the bf16 weight-streaming kernels, as in nv_vit_forward_in
      const long rs = D.n;
      RUN(nv_skinny_nt(0, B, d, D.inner, ws + w.ao, D.inner * rs, p16 + q.wo, D.inner, p + q.bo, xin, d * rs, x1, d * rs, nullptr, 0, site_seed(drop_seed, 4 * l + 1), proj_drop_p(cfg, drop_p), stream));
      RUN(nv_ln_fwd(x1, d * rs, B, d, p + q.n2g, p + q.n2b, eps, ws + w.xn2, d * rs, st2, st2 + M, stream));
      RUN(nv_skinny_nt(1, B, D.m, d, ws + w.xn2, d * rs, p16 + q.w1, d, p + q.b1, nullptr, 0, ws + w.h, D.m * rs, ws + w.u, D.m * rs, site_seed(drop_seed, 4 * l + 2), drop_p, stream));
      RUN(nv_skinny_nt(0, B, d, D.m, ws + w.h, D.m * rs, p16 + q.w2, D.m, p + q.b2, x1, d * rs, x2, d * rs, nullptr, 0, site_seed(drop_seed, 4 * l + 3), drop_p, stream));
      xin = x2;
      continue;
    }
    RUN(nv_gemm_bf16(0, 4, M, d, D.inner, ws + w.ao, D.inner, p16 + q.wo, D.inner, x1, d, p + q.bo, xin, d, nullptr, 0, 0, 1.f, site_seed(drop_seed, 4 * l + 1), proj_drop_p(cfg, drop_p), stream));
    RUN(nv_ln_fwd_f8_train(x1, d, M, d, p + q.n2g, p + q.n2b, eps, s_xn2, x8, d, ws + w.xn2, d, st2, st2 + M, stream));
    RUN(nv_gemm_f8_gelu_train(M, D.m, d, x8, d, p8 + q.w1, d, cs + 3L * D.inner, p + q.b1, s_h, h8, D.m, ws + w.h, D.m, ws + w.u, D.m,
                              site_seed(drop_seed, 4 * l + 2), drop_p, stream));
    RUN(nv_gemm_f8_resid_drop(M, d, D.m, h8, D.m, p8 + q.w2, D.m, x2, d, cs + 3L * D.inner + D.m, p + q.b2, x1, d, site_seed(drop_seed, 4 * l + 3), drop_p, stream));
    xin = x2;
  }
  const float* pooled = xin;
  long pooled_stride = (long)D.n * d;
  if (D.pool_mean) {
    RUN(nv_token_mean(xin, B, D.n, d, (float*)(ws + W.xm), stream));
    pooled = (float*)(ws + W.xm); pooled_stride = d;
  }
  RUN(nv_head_fwd(pooled, pooled_stride, B, d, p + T.hg, p + T.hb, eps, p + T.hw, p + T.hbias, D.C, (float*)(ws + W.xh), (float*)(ws + W.hst),
                  logits, stream));
  return NV_OK;
}

// Backward in stages so the caller can overlap the data-parallel gradient all-reduce with it:
//   stage 0 = classification head, stage 1+k = transformer layer (depth-1-k), stage depth+1 = patch embedding.
// Stages must be run in increasing order over [0, depth+1]; the running residual gradient lives in the workspace.
// When stage s has run, the gradient-arena range of its parameters is final (see nv_vit_stage_param_range).
extern "C" int nv_vit_backward_stages(const nv_vit_config* cfg, int B, const float* video, const long* strides5, const float* params,
                                      const void* params16, void* workspace, long ws_bytes, const float* dlogits, float* grads,
                                      int accumulate, int first_stage, int last_stage, float drop_p, float emb_drop_p,
                                      unsigned long drop_seed, void* stream, void* aux_stream, int join_aux) {
  return nv_vit_backward_stages16(cfg, B, video, strides5, params, params16, workspace, ws_bytes, dlogits, grads, nullptr, accumulate, first_stage,
                                  last_stage, drop_p, emb_drop_p, drop_seed, stream, aux_stream, join_aux, 0);
}

// fuse != null (nv_vit_train_step, fuse_update): the four Linear weights of every layer are updated on the auxiliary stream while the
// backward pass is still running - fuse_mode 3: by an AdamW launch over those weights queued behind the layer's weight-gradient
// GEMMs; fuse_mode 1 / 2: by those GEMMs themselves (nv_gemm_bf16_grouped_adamw).  Either rewrites the bf16 shadow of W_qkv, which
// the layer's last data-gradient GEMM (dxn1 = dqkv W_qkv) reads: in these modes that GEMM is queued BEFORE the main stream signals
// the auxiliary one.
static int backward_impl(const nv_vit_config* cfg, int B, const float* video, const long* strides5, const float* params,
                         const void* params16, void* workspace, long ws_bytes, const float* dlogits, float* grads, void* grads16,
                         int accumulate, int first_stage, int last_stage, float drop_p, float emb_drop_p,
                         unsigned long drop_seed, void* stream, void* aux_stream, int join_aux, int rows_form, const nv_adamw_arena* fuse, int fuse_mode);

extern "C" int nv_vit_backward_stages16(const nv_vit_config* cfg, int B, const float* video, const long* strides5, const float* params,
                                        const void* params16, void* workspace, long ws_bytes, const float* dlogits, float* grads, void* grads16,
                                        int accumulate, int first_stage, int last_stage, float drop_p, float emb_drop_p,
                                        unsigned long drop_seed, void* stream, void* aux_stream, int join_aux, int rows_form) {
  return backward_impl(cfg, B, video, strides5, params, params16, workspace, ws_bytes, dlogits, grads, grads16, accumulate, first_stage, last_stage,
                       drop_p, emb_drop_p, drop_seed, stream, aux_stream, join_aux, rows_form, nullptr, 0);
}

static int backward_impl(const nv_vit_config* cfg, int B, const float* video, const long* strides5, const float* params,
                         const void* params16, void* workspace, long ws_bytes, const float* dlogits, float* grads, void* grads16,
                         int accumulate, int first_stage, int last_stage, float drop_p, float emb_drop_p,
                         unsigned long drop_seed, void* stream, void* aux_stream, int join_aux, int rows_form, const nv_adamw_arena* fuse, int fuse_mode) {
  Dims D; RUN(make_dims(cfg, B, D));
  NV_CHECK_ARG(!fuse || (!accumulate && !grads16 && fuse->grads == grads && first_stage <= 1 && last_stage == D.L + 1),
               "nv_vit_backward: the optimizer update during the backward pass needs accumulate = 0, no bf16 mirror, its own gradient arena and every stage in one call");
  ParamTab T; make_params(D, T);
  WS W; make_ws(D, 1, W);
  NV_CHECK_ARG(video && strides5 && params && params16 && workspace && dlogits && grads, "nv_vit_backward: null pointer");
  NV_CHECK_ARG(ws_bytes >= W.total, "nv_vit_backward: workspace too small (%ld < %ld) - forward must run with training=1", ws_bytes, W.total);
  NV_CHECK_ARG(nv_aligned16(grads) && nv_aligned16(grads16), "nv_vit_backward: grads / grads16 must be 16-byte aligned");
  const bool tail_fwd = cls_tail_wanted(D, 1, drop_p, rows_form);     // the form the (training) forward took, given the same arguments
  r16* gr16 = (r16*)grads16;                     // optional bf16 mirror of the Linear weight gradients (data-parallel messages)
  auto M16 = [&](long off) -> void* { return gr16 ? (void*)(gr16 + off) : nullptr; };
  char* ws = (char*)workspace;
  const float* p = params;
  const r16* p16 = (const r16*)params16;
  float* gr = grads;
  const int M = D.M, d = D.d, acc = accumulate;
  float* g = (float*)(ws + W.g);
  // Buffers the auxiliary stream reads exist twice (even / odd layers): layer l's weight-gradient work runs while the main
  // stream is already in layer l-1, which writes the other copy.
  auto G16 = [&](int l) -> void* { return ws + ((l & 1) ? W.alt[0] : W.g16); };     // bf16 residual gradient entering layer l
  auto G16B = [&](int l) -> void* { return ws + ((l & 1) ? W.alt[1] : W.g16b); };   // ... after LN2 backward
  auto DU = [&](int l) -> void* { return ws + ((l & 1) ? W.alt[2] : W.du); };
  auto DQKV = [&](int l) -> void* { return ws + ((l & 1) ? W.alt[3] : W.dqkv); };
  auto RED = [&](int l) -> void* { return ws + ((l & 1) ? W.alt[4] : W.red); };
  auto RED2 = [&](int l) -> void* { return ws + ((l & 1) ? W.alt[5] : W.red2); };
  auto RED3 = [&](int l) -> void* { return ws + ((l & 1) ? W.alt[6] : W.red3); };
  auto CS1 = [&](int l) -> float* { return (float*)(ws + ((l & 1) ? W.alt[7] : W.cs1)); };
  // rows of the GEMM tile that will compute dU: > 0 when the fused column-sum epilogue (bias gradient of FC1) is available
  const int du_tile_rows = nv_gemm_tile_rows(1, M, D.m, d, d, D.m);
  const int ln_rows = nv_ln_bwd_partial_rows(M);
  void* red = ws + W.red;
  const float scale = 1.0f / sqrtf((float)D.dh);
  hipStream_t S = (hipStream_t)stream;
  hipStream_t A = aux_stream ? (hipStream_t)aux_stream : S;       // weight-gradient stream (== S: fully serial)
  void* sA = (void*)A;
  const bool forked = (A != S);
  // (no S -> A ordering here: everything the auxiliary stream does in this call is queued behind a signal of the main stream below)

  NV_CHECK_ARG(first_stage >= 0 && last_stage <= D.L + 1 && first_stage <= last_stage, "nv_vit_backward_stages: bad stage range [%d, %d]", first_stage, last_stage);
  // head: writes g (zeros + cls rows) and the last layer's FC2 bias gradient (colsum of g)
  const float* xlast = (float*)(ws + W.layer[D.L - 1].x2);
  if (first_stage == 0)
  RUN(nv_head_bwd(dlogits, B, D.C, p + T.hw, D.pool_mean ? (const float*)(ws + W.xm) : xlast, D.pool_mean ? (long)d : (long)D.n * d,
                  (float*)(ws + W.hst), (float*)(ws + W.xh), p + T.hg, d, D.n, g, d, G16(D.L - 1), d,
                  gr + T.hg, gr + T.hb, gr + T.hw, gr + T.hbias, gr + T.layer[D.L - 1].b2, acc, red, W.red_bytes,
                  site_seed(drop_seed, 4 * (D.L - 1) + 3), drop_p, D.pool_mean, stream));

  // One cross-stream event per layer in each direction (an event record costs several microseconds of queue time): the main
  // stream signals once, after the attention backward; the auxiliary stream then runs, one layer behind the main stream,
  // [db1 column sum, LN2 reduction, LN1 reduction of the layer above, grouped weight-gradient GEMMs] and signals back once.
  hipEvent_t prev_done = carry_take(workspace);     // everything the previous (higher) layer queued on [A] - in the previous, unjoined call
  bool layers_here = false;
  int pending_ln1 = -1;               // layer whose LN1-backward partials still wait for their reduction
  int pending_adam = -1;              // fuse_mode 3: layer whose weights are updated at the NEXT signal - the main stream orders a layer's dxn1 GEMM (last
                                      // reader of its bf16 weights) behind the signal of that layer, so the update waits for the one after it
  auto layer_update = [&](int lu) -> int {
    const LayerP& qu = T.layer[lu];
    const long b[3] = {qu.wqkv, qu.w1, qu.w2};
    const long n[3] = {align_up(3L * D.inner * d, 8) + (long)d * D.inner, (long)D.m * d, (long)d * D.m};
    return nv_adamw_ranges(fuse, b, n, 3, sA);
  };
  auto reduce_ln1 = [&](int lp) -> int {
    const LayerP& qp = T.layer[lp];
    return nv_ln_bwd_reduce(RED3(lp), M, d, gr + qp.n1g, gr + qp.n1b, (lp > 0) ? gr + T.layer[lp - 1].b2 : nullptr, acc, sA);
  };
  void* const ln_reduce = NV_LN_NO_REDUCE;   // every parameter-gradient reduction of a layer goes into ONE nv_reduce_multi launch
  for (int l = D.L - 1; l >= 0; --l) {
    const int stage = D.L - l;
    if (stage < first_stage || stage > last_stage) continue;
    const LayerP& q = T.layer[l];
    const LayerW& w = W.layer[l];
    const float* xin = (l == 0) ? (float*)(ws + W.x0) : (float*)(ws + W.layer[l - 1].x2);
    float* st1 = (float*)(ws + w.st1);
    float* st2 = (float*)(ws + w.st2);
    float* dxn = (float*)(ws + W.dxn);
    void* g16 = G16(l);
    void* g16b = G16B(l);
    void* du = DU(l);
    void* dqkv = DQKV(l);
    // last block in the cls-rows form (see g_cls_tail): the residual gradient is zero outside the B cls rows until the attention
    // backward mixes the rows, so dU, dxn2, the LN2 backward, dAO and three of the four weight gradients are products of B rows
    const bool tail = tail_fwd && l == D.L - 1;
    const int Mr = tail ? B : M;                 // rows that carry a gradient; row r of the small problem is row r * rs of the buffers
    const long rs = tail ? (long)D.n : 1;
    // ---- FeedForward backward (vit_3d.py:16-26)
    if (tail) {
      RUN(nv_skinny_nn(0, B, D.m, d, g16, d * rs, p16 + q.w2, D.m, ws + w.u, D.m * rs, du, D.m * rs, gr + q.b1, acc, site_seed(drop_seed, 4 * l + 2), drop_p, stream));   // dU = (g W2 * mask) * gelu'(u), db1 = column sums
      RUN(nv_skinny_nn(1, B, d, D.m, du, D.m * rs, p16 + q.w1, d, nullptr, 0, dxn, d * rs, nullptr, 0, 0, 0.f, stream));             // dxn2 = dU W1
    } else {
    RUN(nv_gemm_bf16(1, du_tile_rows ? 6 : 5, M, D.m, d, g16, d, p16 + q.w2, D.m, du, D.m, nullptr, ws + w.u, D.m, du_tile_rows ? CS1(l) : nullptr, D.m, 0, 1.f,
                     site_seed(drop_seed, 4 * l + 2), drop_p, stream));   // dU = (g W2 * mask) * gelu'(u)  [+ per-tile column sums -> db1]
    RUN(nv_gemm_bf16(1, 1, M, d, D.m, du, D.m, p16 + q.w1, d, dxn, d, nullptr, nullptr, 0, nullptr, 0, 0, 1.f, 0, 0.f, stream));                        // dxn2 = dU W1
    }
    RUN(nv_ln_bwd(dxn, d * rs, (float*)(ws + w.x1), d * rs, st2, st2 + M, p + q.n2g, Mr, d, g, g, d * rs, g16b, d * rs, gr + q.n2g, gr + q.n2b, gr + q.bo, acc, RED(l),
                  W.red_bytes, site_seed(drop_seed, 4 * l + 1), proj_drop_p(cfg, drop_p), stream, ln_reduce));                                   // g += dLN2 -> g16b
    // ---- Attention backward (vit_3d.py:48-60)
    if (tail) {
      RUN(nv_skinny_nn_sparse(B, D.inner, d, g16b, d * rs, p16 + q.wo, D.inner, ws + W.dao, M, D.n, stream));                       // dAO = g Wo on the cls rows, zeros elsewhere
    } else
    RUN(nv_gemm_bf16(1, 0, M, D.inner, d, g16b, d, p16 + q.wo, D.inner, ws + W.dao, D.inner, nullptr, nullptr, 0, nullptr, 0, 0, 1.f, 0, 0.f, stream));  // dAO = g Wo
    RUN(nv_attn_bwd(ws + w.qkv, 3 * D.inner, ws + w.ao, ws + W.dao, D.inner, (float*)(ws + w.lse), B, D.n, D.heads, D.dh, scale,
                    (float*)(ws + W.delta), dqkv, 3 * D.inner, site_seed(drop_seed, 4 * l + 0), drop_p, stream));
    float* dxn1 = (l == D.L - 1) ? (float*)(ws + W.hookg) : dxn;    // gradient of the last block's attention-LN output is kept (Grad-CAM hook)
    const bool dxn1_first = fuse && fuse_mode != 3;
    if (dxn1_first)      // the last reader of this layer's bf16 weights, ahead of the launch that rewrites them (see above)
      RUN(nv_gemm_bf16(1, 1, M, d, 3 * D.inner, dqkv, 3 * D.inner, p16 + q.wqkv, d, dxn1, d, nullptr, nullptr, 0, nullptr, 0, 0, 1.f, 0, 0.f, stream));
    // ---- [A] everything of this layer that only finishes parameter gradients
    if (forked) RUN(stream_sync(S, A));                                                                                        // dU, g16b, dqkv (and the LN partials) ready
    if (pending_adam >= 0) { RUN(layer_update(pending_adam)); pending_adam = -1; }      // [A] fuse_mode 3: the layer above, whose last reader (its dxn1 GEMM) ran before this signal
    {
      // [A] ONE launch for every small parameter gradient that is final by now: db1 (column sums of dU), dLN2 affine + dbo
      // (= colsum(g)), and dLN1 affine + db2 of the layer above (whose partials were written after that layer's block ran)
      nv_reduce_job jobs[3];
      int nj = 0;
      if (tail) {}                                                  // db1 came out of the dU kernel
      else if (du_tile_rows) jobs[nj++] = {CS1(l), (M + du_tile_rows - 1) / du_tile_rows, D.m, 1, {gr + q.b1, nullptr, nullptr}, acc};
      else RUN(nv_colsum_bf16(du, D.m, M, D.m, gr + q.b1, acc, RED2(l), W.red2_bytes, sA));
      jobs[nj++] = {(const float*)RED(l), nv_ln_bwd_partial_rows(Mr), d, 3, {gr + q.n2g, gr + q.n2b, gr + q.bo}, acc};
      if (pending_ln1 >= 0) {
        const LayerP& qp = T.layer[pending_ln1];
        jobs[nj++] = {(const float*)RED3(pending_ln1), ln_rows, d, 3, {gr + qp.n1g, gr + qp.n1b, (pending_ln1 > 0) ? gr + T.layer[pending_ln1 - 1].b2 : nullptr}, acc};
        pending_ln1 = -1;
      }
      RUN(nv_reduce_multi(jobs, nj, sA));
    }
    {
      // the four weight gradients of the layer in ONE grouped launch (864 tiles keep two workgroups resident on every CU;
      // launched one by one their 72-288 tiles leave the CUs half empty and latency bound)
      nv_gemm_problem pr[4];
      pr[0] = {d, D.m, Mr, g16, d * rs, ws + w.h, D.m * rs, gr + q.w2, D.m, acc, M16(q.w2), D.m};                     // dW2 = g^T h
      pr[1] = {D.m, d, Mr, du, D.m * rs, ws + w.xn2, d * rs, gr + q.w1, d, acc, M16(q.w1), d};                        // dW1 = dU^T xn2
      pr[2] = {d, D.inner, Mr, g16b, d * rs, ws + w.ao, D.inner * rs, gr + q.wo, D.inner, acc, M16(q.wo), D.inner};   // dWo = g^T ao
      pr[3] = {3 * D.inner, d, M, dqkv, 3 * D.inner, ws + w.xn1, d, gr + q.wqkv, d, acc, M16(q.wqkv), d};  // dWqkv = dqkv^T xn1
      if (fuse && fuse_mode == 3) {      // gradients stored as ever; the layer's update follows as a launch of its own, one signal later
        RUN(nv_gemm_bf16_grouped(2, 1, 4, pr, sA));
        pending_adam = l;
      } else if (fuse) RUN(nv_gemm_bf16_grouped_adamw(4, pr, fuse, sA));
      else RUN(nv_gemm_bf16_grouped(2, 1, 4, pr, sA));
    }
    hipEvent_t done = nullptr;
    if (forked) { done = deferred_event(); if (!done || hipEventRecord(done, A) != hipSuccess) { nv_set_error("nv_vit_backward: event record failed"); return NV_ERR_HIP; } }
    if (!dxn1_first)
      RUN(nv_gemm_bf16(1, 1, M, d, 3 * D.inner, dqkv, 3 * D.inner, p16 + q.wqkv, d, dxn1, d, nullptr, nullptr, 0, nullptr, 0, 0, 1.f, 0, 0.f, stream));
    // LN1 backward writes the residual gradient of layer l-1 into the buffer copy layer l+1 used (and layer l-1 then rewrites
    // the rest of that copy): the auxiliary work of layer l+1 - a whole layer behind by now - must have finished with it.
    if (forked && prev_done && hipStreamWaitEvent(S, prev_done, 0) != hipSuccess) { nv_set_error("nv_vit_backward: event wait failed"); return NV_ERR_HIP; }
    RUN(nv_ln_bwd(dxn1, d, xin, d, st1, st1 + M, p + q.n1g, M, d, g, g, d, G16(l - 1), d, gr + q.n1g, gr + q.n1b, (l > 0) ? gr + T.layer[l - 1].b2 : nullptr,
                  acc, RED3(l), nv_ln_bwd_workspace_bytes(M, d), site_seed(drop_seed, 4 * (l - 1) + 3), (l > 0) ? drop_p : 0.f, stream, ln_reduce));
    pending_ln1 = l;
    prev_done = done;
    layers_here = true;
  }
  if (last_stage < D.L + 1) {
    if (pending_ln1 >= 0) { if (forked) RUN(stream_sync(S, A)); RUN(reduce_ln1(pending_ln1)); }
    if (forked) {
      if (join_aux) {
        RUN(stream_sync(A, S));   // every gradient written on [A] (weight GEMMs, reductions) is ordered before what follows on the main stream
      } else {                    // the caller orders the consumer of this range after BOTH streams; the next call inherits the dependency
        RUN(carry_record(workspace, A));
      }
    }
    return NV_OK;
  }
  // a call that starts at the embedding stage behind an unjoined call: its scratch may still be in use over there
  if (forked && !layers_here && prev_done && hipStreamWaitEvent(S, prev_done, 0) != hipSuccess) { nv_set_error("nv_vit_backward: event wait failed"); return NV_ERR_HIP; }
  // ---- patch embedding backward (vit_3d.py:91-96,116-118).  The main stream does NOT join the auxiliary one first: layer 0's
  // grouped weight gradients keep running beside it.  Its reduction scratch is the odd copy, which the auxiliary stream
  // released before the main stream was allowed into layer 0's LN1 backward.
  float* est = (float*)(ws + W.est);
  float* pst = (float*)(ws + W.pst);
  RUN(nv_embed_finish_bwd(g, d, (float*)(ws + W.t), d, est, est + D.T, p + T.pe_g2, B, D.N, d, (float*)(ws + W.dt), d, ws + W.dt16, d, gr + T.pe_g2,
                          gr + T.pe_b2, gr + T.pe_bias, gr + T.pos, gr + T.cls, acc, RED(1), W.red_bytes, site_seed(drop_seed, 4 * D.L), emb_drop_p, stream));
  // patch_dim not a multiple of 8 (reference default 90^3 / p 9 -> P = 729): operands are zero padded to Ppad columns;
  // the weight gradient is produced in a padded scratch matrix and its valid columns copied / added into the arena.
  const void* wpe = (D.P != D.Ppad) ? (const void*)(ws + W.wpe16) : (const void*)(p16 + T.pe_w);
  void* redA = forked ? (void*)(ws + W.red2) : red;
  const long redA_bytes = forked ? W.red2_bytes : W.red_bytes;
  // [A] gradient of the patch LayerNorm's affine parameters (needs dxp = dt Wpe and a second gather of the volume);
  // the main stream meanwhile produces the patch-embedding weight gradient
  if (forked) RUN(stream_sync(S, A));                                                                                          // dt16 and layer 0's LN1 partials ready
  if (pending_ln1 >= 0) RUN(reduce_ln1(pending_ln1));
  if (pending_adam >= 0) { RUN(layer_update(pending_adam)); pending_adam = -1; }                                                // [A] layer 0's weights
  RUN(nv_gemm_bf16(1, 1, D.T, D.Ppad, d, ws + W.dt16, d, wpe, D.Ppad, ws + W.dxp, D.Ppad, nullptr, nullptr, 0, nullptr, 0, 0, 1.f, 0, 0.f, sA));              // [A] dxp = dt Wpe
  RUN(nv_patch_ln_bwd(video, strides5, B, cfg->channels, cfg->frames, cfg->image_size, img_w(cfg), cfg->image_patch_size,
                      pat_w(cfg), cfg->frame_patch_size, (float*)(ws + W.dxp), D.Ppad, pst, pst + D.T, gr + T.pe_g, gr + T.pe_b, acc, redA,
                      redA_bytes, sA));                                                                                         // [A]
  if (D.P != D.Ppad) {
    RUN(nv_gemm_bf16(2, 1, d, D.Ppad, D.T, ws + W.dt16, d, ws + W.xp, D.Ppad, ws + W.dwpe, D.Ppad, nullptr, nullptr, 0, nullptr, 0, 0, 1.f, 0, 0.f, stream));
    RUN(nv_copy_2d_f32((float*)(ws + W.dwpe), D.Ppad, d, D.P, gr + T.pe_w, D.P, acc, stream));
  } else {
    RUN(nv_gemm_bf16(2, 1, d, D.P, D.T, ws + W.dt16, d, ws + W.xp, D.Ppad, gr + T.pe_w, D.P, nullptr, nullptr, 0, M16(T.pe_w), D.P, acc, 1.f, 0, 0.f, stream));   // dWpe = dt^T xp
  }
  if (forked) RUN(stream_sync(A, S));
  return NV_OK;
}

extern "C" int nv_vit_backward(const nv_vit_config* cfg, int B, const float* video, const long* strides5, const float* params,
                               const void* params16, void* workspace, long ws_bytes, const float* dlogits, float* grads,
                               int accumulate, float drop_p, float emb_drop_p, unsigned long drop_seed, void* stream, void* aux_stream) {
  return nv_vit_backward_stages(cfg, B, video, strides5, params, params16, workspace, ws_bytes, dlogits, grads, accumulate, 0,
                                cfg ? cfg->depth + 1 : 0, drop_p, emb_drop_p, drop_seed, stream, aux_stream, 1);
}

// Element range [begin, end) of the parameter / gradient arena that is FINAL once backward stage `stage` has run.
// (stage 0: head; stage 1+k: layer depth-1-k - its FC2 bias was already written by the stage before; last stage: embedding.)
extern "C" int nv_vit_stage_param_range(const nv_vit_config* cfg, int stage, long* begin, long* end) {
  Dims D; RUN(make_dims(cfg, 1, D));
  ParamTab T; make_params(D, T);
  NV_CHECK_ARG(stage >= 0 && stage <= D.L + 1 && begin && end, "nv_vit_stage_param_range: bad stage %d", stage);
  if (stage == 0) { *begin = T.hg; *end = T.total; return NV_OK; }
  if (stage == D.L + 1) { *begin = 0; *end = T.layer[0].n1g; return NV_OK; }
  const int l = D.L - stage;
  *begin = T.layer[l].n1g;
  *end = (l + 1 < D.L) ? T.layer[l + 1].n1g : T.hg;
  return NV_OK;
}

// ---- data-parallel backward + update of nv_vit_train_step (nv_dp_plan): the backward pass in groups of stages; behind each group the
// communication stream all-reduces the group's gradient range (RCCL, comm.cpp) - and, with update_per_bucket, applies AdamW to it -
// while the main stream is already in the next group.  The same launches as the single-process step otherwise.
static int dp_backward_update(const nv_vit_config* cfg, int B, const float* video, const long* strides5, const nv_vit_input* in, float* params, void* params16,
                              float* grads, float* adam_m, float* adam_v, void* workspace, long ws_bytes, const float* dlogits, const nv_train_hparams* hp,
                              const nv_dp_plan* dp, bool head_fused, float lscale, float drop_p, float emb_drop_p, unsigned long drop_seed, void* stream,
                              void* aux_stream) {
  Dims D; RUN(make_dims(cfg, B, D));
  ParamTab T; make_params(D, T);
  const int n_stages = D.L + 2, last_stage = D.L + 1;
  const int nb = dp->n_buckets < n_stages ? dp->n_buckets : n_stages;
  const bool forked = aux_stream && aux_stream != stream;
  void* C = dp->comm_stream;
  r16* msg = (r16*)dp->grads16;
  const float gs = hp->grad_scale / lscale / (float)dp->world;
  // gradient ranges the GEMMs write into the 16-bit message arena themselves (nv_vit_backward_stages16): everything else of a bucket is converted here
  std::vector<std::pair<long, long>> mirrored;
  if (msg) {
    if (D.P % 8 == 0) mirrored.push_back({T.pe_w, T.pe_w + (long)D.d * D.P});
    for (int l = 0; l < D.L; ++l) {
      const LayerP& q = T.layer[l];
      mirrored.push_back({q.wqkv, q.wqkv + 3L * D.inner * D.d}); mirrored.push_back({q.wo, q.wo + (long)D.d * D.inner});
      mirrored.push_back({q.w1, q.w1 + (long)D.m * D.d}); mirrored.push_back({q.w2, q.w2 + (long)D.d * D.m});
    }
  }
  // update_per_bucket: 1 = AdamW of a bucket on the communication stream right behind its all-reduce; 2 = on the AUXILIARY stream, one bucket late
  // (queued in front of the next bucket's weight-gradient work once the all-reduce has finished: the HBM-bound update then shares the chip
  // with the main stream's chain only, never with the weight-gradient GEMMs - the placement of the single-process step's fuse_update = 3)
  const int upd = (dp->update_per_bucket == 2 && !forked) ? 1 : dp->update_per_bucket;
  auto adamw_range = [&](long begin, long end, void* on) -> int {
    return nv_adamw_step_scaled(params + begin, msg ? (const void*)(msg + begin) : (const void*)(grads + begin), msg ? 1 : 0, adam_m + begin, adam_v + begin,
                                (r16*)params16 + begin, end - begin, hp->step, hp->lr, hp->beta1, hp->beta2, hp->eps, hp->weight_decay, gs, 0, nullptr, on);
  };
  long late_begin = -1, late_end = -1;
  int s0 = 0;
  for (int b = 0; b < nb; ++b) {
    const int cnt = n_stages / nb + (b < n_stages % nb ? 1 : 0), s1 = s0 + cnt - 1;       // stages [s0, s1]: parallel.py::bucket_stages
    if (late_begin >= 0) {                                                                // [A] the previous bucket's update, behind its all-reduce
      RUN(stream_sync((hipStream_t)C, (hipStream_t)aux_stream));
      RUN(adamw_range(late_begin, late_end, aux_stream));
      late_begin = -1;
    }
    const int first = (head_fused && s0 == 0) ? 1 : s0;                                  // (the fused head step has run stage 0 already)
    const bool joined = (s1 == last_stage);
    if (first <= s1)
      RUN(nv_vit_backward_stages16(cfg, B, video, strides5, params, params16, workspace, ws_bytes, dlogits, grads, msg, hp->accumulate ? 1 : 0, first, s1, drop_p,
                                   emb_drop_p, drop_seed, stream, aux_stream, joined ? 1 : 0, in ? in->rows_form : 0));
    long begin = -1, end = -1;
    for (int s = s0; s <= s1; ++s) {
      long lo, hi;
      RUN(nv_vit_stage_param_range(cfg, s, &lo, &hi));
      begin = (begin < 0 || lo < begin) ? lo : begin; end = hi > end ? hi : end;
    }
    RUN(stream_sync((hipStream_t)stream, (hipStream_t)C));                                  // the bucket's gradients: complete on the main stream ...
    if (forked && !joined) RUN(stream_sync((hipStream_t)aux_stream, (hipStream_t)C));      // ... and on the auxiliary one (not joined into the main stream yet)
    if (msg) {
      std::vector<long> rb, rl;
      long cur = begin;
      for (const auto& mr : mirrored) {
        if (mr.second <= begin || mr.first >= end) continue;
        if (mr.first > cur) { rb.push_back(cur); rl.push_back(mr.first - cur); }
        cur = mr.second > cur ? mr.second : cur;
      }
      if (cur < end) { rb.push_back(cur); rl.push_back(end - cur); }
      RUN(nv_cast_ranges_bf16(grads, msg, rb.data(), rl.data(), (int)rb.size(), C));
      RUN(nv_comm_all_reduce(dp->comm, msg + begin, end - begin, 1, C));
    } else {
      RUN(nv_comm_all_reduce(dp->comm, grads + begin, end - begin, 0, C));
    }
    if (upd == 1) RUN(adamw_range(begin, end, C));
    else if (upd == 2) { late_begin = begin; late_end = end; }
    s0 = s1 + 1;
  }
  RUN(stream_sync((hipStream_t)C, (hipStream_t)stream));                                    // every bucket reduced (and updated) before what follows on the main stream
  if (late_begin >= 0) RUN(adamw_range(late_begin, late_end, stream));                       // the last bucket (layer 0 + embedding): nothing left to overlap with
  if (!upd) {
    if (hp->loss_scale_state) {
      RUN(nv_loss_scale_check(grads, T.total, hp->loss_scale_state, stream));
      RUN(nv_loss_scale_update(hp->loss_scale_state, hp->lr, hp->beta1, hp->beta2, stream));
    }
    RUN(nv_adamw_step_scaled(params, msg ? (const void*)msg : (const void*)grads, msg ? 1 : 0, adam_m, adam_v, params16, T.total, hp->step, hp->lr, hp->beta1, hp->beta2,
                             hp->eps, hp->weight_decay, gs, 0, hp->loss_scale_state, stream));
  }
  return NV_OK;
}

// ---- the whole train step in one call (Trainer.py:65-79): forward -> CrossEntropyLoss -> backward -> AdamW.  Host-side sequencing
// only: the same launches, in the same order, on the same streams as the four separate calls - enqueued without a Python
// interpreter (or an autograd graph walk) between them.
extern "C" int nv_vit_train_step(const nv_vit_config* cfg, int B, const float* video, const long* shape5, const long* strides5, const nv_vit_input* in,
                                 float* params, void* params16, float* grads, float* adam_m, float* adam_v, void* workspace, long ws_bytes,
                                 const long* labels, float* logits, float* loss, float* dlogits, const nv_train_hparams* hp,
                                 float drop_p, float emb_drop_p, unsigned long drop_seed, void* stream, void* aux_stream) {
  NV_CHECK_ARG(cfg && hp && hp->struct_size == (int)sizeof(nv_train_hparams), "nv_vit_train_step: nv_train_hparams.struct_size = %d, this library expects %d (ABI revision %d)",
               hp ? hp->struct_size : -1, (int)sizeof(nv_train_hparams), NV_ABI_VERSION);
  NV_CHECK_ARG(labels && loss && dlogits && grads && (!hp->update || (adam_m && adam_v && hp->step >= 1)), "nv_vit_train_step: null pointer (labels / loss / dlogits / grads / optimizer state) or step < 1");
  NV_CHECK_ARG(!(in && in->time_points > 0), "nv_vit_train_step: the fused 4D input form is forward-only (the 4D model's encoder is frozen, NeuroEncoder.py:34-36)");
  // every argument check ahead of the first launch (a refused call must not leave half a step in the queue)
  NV_CHECK_ARG(hp->fuse_update >= 0 && hp->fuse_update <= 3, "nv_vit_train_step: fuse_update = %d (0 .. 3)", hp->fuse_update);
  NV_CHECK_ARG(hp->loss_scale >= 0.f && !(hp->loss_scale_state && hp->loss_scale > 0.f && hp->loss_scale != 1.f),
               "nv_vit_train_step: loss_scale must be >= 0 (0 = 1 = none) and is not combined with a dynamic loss_scale_state");
  const bool fused = hp->update && !hp->accumulate && hp->fuse_update;
  NV_CHECK_ARG(!(fused && hp->loss_scale_state), "nv_vit_train_step: a dynamic loss scale decides AFTER the backward pass whether the step is applied: fuse_update must be 0");
  const nv_dp_plan* dp = hp->dp;
  if (dp) {
    NV_CHECK_ARG(dp->struct_size == (int)sizeof(nv_dp_plan) && dp->comm && dp->comm_stream && dp->world >= 1 && dp->n_buckets >= 1,
                 "nv_vit_train_step: nv_dp_plan.struct_size = %d (expected %d), or null communicator / stream, world < 1, n_buckets < 1", dp->struct_size, (int)sizeof(nv_dp_plan));
    NV_CHECK_ARG(!fused, "nv_vit_train_step: with a data-parallel plan the optimizer update follows the all-reduce (nv_dp_plan.update_per_bucket): fuse_update must be 0");
    NV_CHECK_ARG(dp->update_per_bucket >= 0 && dp->update_per_bucket <= 2, "nv_vit_train_step: nv_dp_plan.update_per_bucket = %d (0, 1, 2)", dp->update_per_bucket);
    NV_CHECK_ARG(!hp->loss_scale_state || (!dp->grads16 && !dp->update_per_bucket),
                 "nv_vit_train_step: a dynamic loss scale checks the REDUCED fp32 gradients before any update: fp32 messages, update_per_bucket = 0");
    NV_CHECK_ARG(dp->comm_stream != stream && dp->comm_stream != aux_stream, "nv_vit_train_step: nv_dp_plan.comm_stream must be a stream of its own");
  }
  const float lscale = hp->loss_scale > 0.f ? hp->loss_scale : 1.f;      // static loss scale: folded into d(loss)/d(logits), undone by the update's grad_scale
  // the head's forward, the loss and the head's backward as two launches instead of five (nv_head_step: bit-identical to the three calls;
  // it needs num_classes <= dim and dim % 8 == 0 - other heads take the five-launch path, which has no such limit)
  const bool head_fused = g_head_step && !cfg->pool_mean && cfg->num_classes <= cfg->dim && cfg->dim % 8 == 0;
  RUN(forward_in_impl(cfg, B, video, shape5, strides5, in, params, params16, workspace, ws_bytes, 1, drop_p, emb_drop_p, drop_seed, logits, stream, head_fused));
  if (head_fused) {
    Dims D; RUN(make_dims(cfg, B, D));
    ParamTab T; make_params(D, T);
    WS W; make_ws(D, 1, W);
    char* ws = (char*)workspace;
    const int Ll = D.L - 1;
    RUN(nv_head_step_scaled((const float*)(ws + W.layer[Ll].x2), (long)D.n * D.d, B, D.d, params + T.hg, params + T.hb, cfg->ln_eps, params + T.hw, params + T.hbias, D.C,
                            labels, lscale, hp->loss_scale_state, (float*)(ws + W.xh), (float*)(ws + W.hst), logits, loss, dlogits, D.n, (float*)(ws + W.g), D.d,
                            ws + ((Ll & 1) ? W.alt[0] : W.g16), D.d, grads + T.hg, grads + T.hb, grads + T.hw, grads + T.hbias, grads + T.layer[Ll].b2,
                            hp->accumulate ? 1 : 0, ws + W.red, W.red_bytes, site_seed(drop_seed, 4 * Ll + 3), drop_p, stream));
  } else {
    RUN(nv_ce_loss_scaled(logits, labels, B, cfg->num_classes, lscale, hp->loss_scale_state, loss, dlogits, stream));
  }
  nv_adamw_arena opt;
  opt.struct_size = (int)sizeof(opt); opt.step = hp->step; opt.lr = hp->lr; opt.beta1 = hp->beta1; opt.beta2 = hp->beta2; opt.eps = hp->eps;
  opt.weight_decay = hp->weight_decay; opt.grad_scale = hp->grad_scale / lscale; opt.keep_grads = hp->fuse_update == 2;
  opt.params = params; opt.grads = grads; opt.adam_m = adam_m; opt.adam_v = adam_v; opt.params16 = params16;
  if (dp && hp->update)
    return dp_backward_update(cfg, B, video, strides5, in, params, params16, grads, adam_m, adam_v, workspace, ws_bytes, dlogits, hp, dp, head_fused, lscale,
                              drop_p, emb_drop_p, drop_seed, stream, aux_stream);
  RUN(backward_impl(cfg, B, video, strides5, params, params16, workspace, ws_bytes, dlogits, grads, nullptr, hp->accumulate ? 1 : 0, head_fused ? 1 : 0, cfg->depth + 1,
                    drop_p, emb_drop_p, drop_seed, stream, aux_stream, 1, in ? in->rows_form : 0, fused ? &opt : nullptr, hp->fuse_update));
  if (hp->update && !fused) {
    const long total = nv_vit_param_count(cfg);
    if (hp->loss_scale_state) {      // GradScaler.step / .update (Trainer.py:75-76) on the device: any inf / NaN gradient skips the update and halves the scale
      RUN(nv_loss_scale_check(grads, total, hp->loss_scale_state, stream));
      RUN(nv_loss_scale_update(hp->loss_scale_state, hp->lr, hp->beta1, hp->beta2, stream));
    }
    RUN(nv_adamw_step_scaled(params, grads, 0, adam_m, adam_v, params16, total, hp->step, hp->lr, hp->beta1, hp->beta2, hp->eps, hp->weight_decay, hp->grad_scale / lscale, 0,
                             hp->loss_scale_state, stream));
  } else if (fused) {
    // what was not updated during the backward pass: the arena minus the four Linear weights of every layer (arena order:
    // ... n1b | wqkv | wo | bo n2g n2b | w1 | b1 | w2 | b2 n1g' ...), one launch
    Dims D; RUN(make_dims(cfg, B, D));
    ParamTab T; make_params(D, T);
    std::vector<long> begins, lens;
    long cur = 0;
    auto skip = [&](long off, long numel) { if (off > cur) { begins.push_back(cur); lens.push_back(off - cur); } cur = align_up(off + numel, 8); };
    for (int l = 0; l < D.L; ++l) {
      const LayerP& q = T.layer[l];
      skip(q.wqkv, 3L * D.inner * D.d); skip(q.wo, (long)D.d * D.inner); skip(q.w1, (long)D.m * D.d); skip(q.w2, (long)D.d * D.m);
    }
    if (T.total > cur) { begins.push_back(cur); lens.push_back(T.total - cur); }
    RUN(nv_adamw_ranges(&opt, begins.data(), lens.data(), (int)begins.size(), stream));
  }
  return NV_OK;
}
