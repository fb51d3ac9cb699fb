/* neurovit_hip.h - C ABI of libneurovit_hip.so: the MI355X (gfx950) hot path of NeuroViT.
 *
 * The reference (gillet-thomas/NeuroViT) has NO native/FFI interface: its seam is the Python
 * nn.Module contract.  This header is the C-ABI that sits beneath our drop-in
 * nn.Modules; every entry point names the reference code it replaces (paths relative to the
 * reference checkout).  The Python binding a maintainer adds is a ctypes.CDLL stub - see
 * INTEGRATION.md.
 *
 * Conventions (all entry points):
 *   - plain pointers + sizes, no torch types; `void*` buffers marked bf16 hold bfloat16;
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch owns all memory);
 *   - `stream` is a hipStream_t; nothing here allocates, frees or synchronises;
 *   - returns 0 on success, <0 on error (NV_ERR_*); nv_last_error() gives the message;
 *   - row-major, `ld*` = leading dimension in ELEMENTS;
 *   - arithmetic: bf16 MFMA operands, fp32 accumulate, fp32 residual stream / LayerNorm /
 *     softmax / optimizer state ("dtype": "bf16" in bench.py).
 */
#ifndef NEUROVIT_HIP_H
#define NEUROVIT_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define NV_OK 0
#define NV_ERR_ARG (-1)
#define NV_ERR_HIP (-2)
#define NV_ERR_ARCH (-3)

/* ---- housekeeping ------------------------------------------------------------------------- */
int nv_version(void);
int nv_arch_ok(void);                 /* 1 iff the current HIP device is gfx950 */
const char* nv_last_error(void);

/* ---- 16-bit operand format (process-wide, like the nv_*_set_* tuning switches): what every `void*` buffer this header calls
 * "bf16" holds, and which MFMA instruction contracts it.  NV_OPERAND_BF16 (default; BASELINE.json's dtype) or NV_OPERAND_FP16 - the
 * reference's own training arithmetic (torch.autocast(float16) + GradScaler, src/Trainer.py:29,68,74-76): same MFMA rate, 11 instead
 * of 8 significand bits (logits within 1e-3 of the reference's fp32 CPU forward), 5 instead of 8 exponent bits (train with a loss
 * scale: nv_train_hparams.loss_scale).  The fp8 entry points require NV_OPERAND_BF16.  Set it before the calls of a model; buffers
 * written under one format must be read under the same one. */
#define NV_OPERAND_BF16 0
#define NV_OPERAND_FP16 1
int nv_set_operand_format(int fmt);
int nv_operand_format(void);

/* ---- optional per-launch hipEvent profiler (bench.py roofline leg).  kind: warp-specialised GEMM kernels 0 NT, 1 NN, 2 TN;
 * 3 attention fwd, 4 attention bwd; 5 fp8 GEMM; eight-wave 256 x 128 GEMM kernel 10 NT, 11 NN, 12 TN, 13 grouped TN; 256 x 256 kernel 20 NT,
 * 21 NN, 22 TN; fp32 path: 30 GEMM, 31 attention.  nv_prof_summary synchronises; call it outside timed regions. */
int nv_prof_enable(int on);
int nv_prof_summary(int kind, double* ms, double* work, long* count);
int nv_prof_summary_bytes(int kind, double* bytes);   /* algorithmic bytes (operands read once + outputs written once) of the same records */

/* ---- GEMM with fused epilogues (replaces every nn.Linear on the path: vit_3d.py:19,22,41,44,94)
 * layout 0 (NT): C[M,N] = A[M,K] . B[N,K]^T      forward  y = x W^T
 * layout 1 (NN): C[M,N] = A[M,K] . B[K,N]        dgrad    dx = dy W
 * layout 2 (TN): C[M,N] = A[K,M]^T . B[K,N]      wgrad    dW = dy^T x
 * epi 0 STORE_BF16, 1 STORE_F32 (+= if accumulate), 2 BIAS_F32, 3 BIAS_GELU (aux_out = pre-activation bf16,
 * C = exact-erf GELU bf16; vit_3d.py:19-20), 4 BIAS_RESID (C f32 = aux_in f32 + acc + bias; vit_3d.py:73-74),
 * 5 DGELU (C bf16 = acc * gelu'(aux_in bf16)), 6 DGELU_COLSUM (5, and aux_out f32 [ceil(M / tile rows), ld_aux_out] receives the
 * per-tile column sums of the stored values: the bias gradient of the Linear in front of the GELU without a second pass over C;
 * sum its rows with nv_reduce_multi; tile rows from nv_gemm_tile_rows).  A, B bf16.
 * Dropout (nn.Dropout of vit_3d.py:21,23,45; drop_p = 0 disables): applied by epilogue 3 to the GELU output, by 4 to
 * (acc + bias) before the residual add, by 5 to acc; element (m, n) is kept iff hash(drop_seed, m*N + n) >= p*2^32 and
 * scaled by 1/(1-p) - the same mask is recomputed wherever the backward pass needs it. */
/* epilogue 1 (fp32 store / accumulate): aux_out, when given, receives a bf16 copy of the stored values (ld_aux_out elements per row) */
int nv_gemm_bf16(int layout, int epi, int M, int N, int K, const void* A, long lda, const void* B, long ldb, void* C,
                 long ldc, const float* bias, const void* aux_in, long ld_aux_in, void* aux_out, long ld_aux_out,
                 int accumulate, float alpha, unsigned long drop_seed, float drop_p, void* stream);

int nv_gemm_tile_rows(int layout, int M, int N, int K, long lda, long ldb);   /* 0: epilogue 6 not available for this shape */

/* ---- fp8 inference path (BASELINE.json configs[4]: ViT3D-large, "fp8 MFMA"): OCP e4m3 operands, fp32 accumulate on
 * v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales), dequantisation per output column in the epilogue.
 * nv_quant_rows_f8: out8[r, :] = sat(W[r, :] * sw[r]) with sw[r] = 448 / max|W[r, :]|; colscale[r] = 1 / (act_scale * sw[r]).
 * nv_ln_fwd_f8: LayerNorm(d) (vit_3d.py:18,37) -> sat(y * out_scale) as e4m3 (inference: no statistics saved).
 * nv_gemm_f8 (NT): C = epi((A8 . B8^T) * colscale[n]); epi 0 bf16 store, 1 f32 store, 4 f32 = aux_in + acc + bias,
 * 7 e4m3 = sat(gelu(acc + bias) * out_scale) (FC1 feeding FC2).  K % 128 == 0. */
int nv_quant_rows_f8(const float* W, long ldw, int rows, int cols, void* out8, long ld8, float act_scale, float* colscale, void* stream);
int nv_ln_fwd_f8(const float* x, long ldx, int M, int d, const float* gamma, const float* beta, float eps, float out_scale, void* y8,
                 long ldy, void* stream);
int nv_gemm_f8(int epi, int M, int N, int K, const void* A8, long lda, const void* B8, long ldb, void* C, long ldc,
               const float* colscale, const float* bias, const void* aux_in, long ld_aux_in, float out_scale, void* stream);
/* Training forward on fp8 operands (BASELINE.json configs[4], "fwd / fwd+bwd"): the forward linears run on e4m3 operands, the backward
 * pass stays on bf16 operands and reads bf16 copies that the SAME forward kernels write.
 * nv_ln_fwd_f8_train: nv_ln_fwd_f8 + the bf16 output and the row statistics of nv_ln_fwd (bit for bit) in one pass over x.
 * nv_gemm_f8_gelu_train (FC1): h8 e4m3 = sat(gelu(u) * out_scale), h16 bf16 = gelu(u), u16 bf16 (optional) = u = acc * colscale + bias. */
int nv_ln_fwd_f8_train(const float* x, long ldx, int M, int d, const float* gamma, const float* beta, float eps, float out_scale, void* y8, long ldy8,
                       void* y16, long ldy16, float* mean, float* rstd, void* stream);
int nv_gemm_f8_gelu_train(int M, int N, int K, const void* A8, long lda, const void* B8, long ldb, const float* colscale, const float* bias,
                          float out_scale, void* h8, long ldh8, void* h16, long ldh16, void* u16, long ldu16, unsigned long drop_seed,
                          float drop_p, void* stream);      /* dropout (vit_3d.py:21) on gelu(u): h8 and h16 carry the same mask */
/* nv_gemm_f8 epilogue 4 with the nn.Dropout of vit_3d.py:23 on (acc * colscale + bias) before the residual add (mask of nv_gemm_bf16 epilogue 4) */
int nv_gemm_f8_resid_drop(int M, int N, int K, const void* A8, long lda, const void* B8, long ldb, void* C, long ldc, const float* colscale,
                          const float* bias, const void* aux_in, long ld_aux_in, unsigned long drop_seed, float drop_p, void* stream);

/* ---- fp32 inference path ("precise" mode).  The reference validates in fp32 without autocast (src/Trainer.py:101-118) and the logits
 * are to match its CPU forward to 1e-3; bf16 MFMA operands cannot (weights rounded to bf16 alone cost 1e-3 ... 6e-3), so these entry
 * points keep every operand fp32 and contract on v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains, 1/16 of the bf16 MFMA rate).
 * nv_gemm_f32 (NT only: every nn.Linear forward, vit_3d.py:19,22,41,44,94): C[M,N] = epi(A[M,K] . B[N,K]^T), all fp32, B = the
 * [out, in] weight as the state_dict holds it.  epi 0 store, 2 + bias, 3 exact-erf GELU(+ bias), 4 resid + (+ bias).  N % 4 == 0.
 * Any K / lda / ldb (float4 operand loads when they are multiples of 4 and 16-byte aligned, scalar loads otherwise).
 * nv_attn_fwd_f32 (vit_3d.py:53-59): qkv f32 [B, n, 3*inner] -> out f32 [B, n, inner]; dim_head a multiple of 4 up to 128. */
int nv_gemm_f32(int epi, int M, int N, int K, const float* A, long lda, const float* B, long ldb, float* C, long ldc,
                const float* bias, const float* resid, long ldr, void* stream);
int nv_gemm_f32_set_tile(int wm, int wn);     /* tuning aid: wave tile (16 wm) x (16 wn), wm, wn in {2, 4}; (0, 0) = heuristic; (-1, 2 | 4 | 0): waves per workgroup of nv_attn_fwd_f32 */
int nv_attn_fwd_f32(const float* qkv, long ld_qkv, int B, int n, int heads, int dim_head, float scale, float* out, long ld_out,
                    void* stream);
int nv_ln_fwd_f32(const float* x, long ldx, int M, int d, const float* gamma, const float* beta, float eps, float* y, long ldy,
                  float* mean, float* rstd, void* stream);      /* as nv_ln_fwd with an fp32 output; mean / rstd may both be NULL */

/* Up to four independent problems of one layout / epilogue in ONE launch (provided for layout 2 / TN with epilogue 1: the
 * four weight-gradient GEMMs of a transformer layer, vit_3d.py:19,22,41,44 backward).  Same arithmetic per tile as
 * nv_gemm_bf16; the point is occupancy: 864 tiles together instead of 72-288 at a time. */
typedef struct nv_gemm_problem {
  int M, N, K;
  const void* A; long lda;     /* A[K, M] bf16 (layout 2) */
  const void* B; long ldb;     /* B[K, N] bf16 */
  void* C; long ldc;           /* C[M, N] f32 */
  int accumulate;              /* C += instead of C = */
  void* C16; long ldc16;       /* optional (NULL): bf16 mirror of the stored C (data-parallel gradient message) */
} nv_gemm_problem;
int nv_gemm_bf16_grouped(int layout, int epi, int count, const nv_gemm_problem* problems, void* stream);

/* ---- AdamW applied where the gradient is produced (Trainer.py:75 optimizer.step() folded into loss.backward() for the Linear
 * weights: torch's "optimizer in backward" pattern).  The five arenas share element offsets: element i of `grads` is the gradient of
 * params[i], whose optimizer state is adam_m[i] / adam_v[i] and whose bf16 shadow is params16[i].
 * nv_gemm_bf16_grouped_adamw: the TN problems of nv_gemm_bf16_grouped (every C inside `grads`, accumulate = 0, no C16) whose epilogue
 *   runs the update of nv_adamw_step on the tile it holds instead of storing it - the same arithmetic on the same fp32 gradient, so
 *   parameters, state and shadow come out bit-identical to nv_gemm_bf16_grouped + nv_adamw_step; the gradient itself is stored only
 *   when keep_grads = 1.  26 B/param of memory traffic instead of 34, and no second pass over these weights.
 *   The caller orders every reader of the weights' bf16 shadow (the data-gradient GEMMs of the same layer) BEFORE this launch.
 * nv_adamw_ranges: the plain update (nv_adamw_step arithmetic, fp32 gradients) over `count` element ranges (HOST arrays; begins and
 *   lens multiples of 4) in one launch - the rest of the arena (biases, LayerNorm, embeddings, head) behind a fused backward. */
typedef struct nv_adamw_arena {
  int struct_size;            /* sizeof(nv_adamw_arena) */
  int step;                   /* >= 1 */
  double lr, beta1, beta2, eps, weight_decay;
  float grad_scale;           /* the update reads grad * grad_scale */
  int keep_grads;             /* nv_gemm_bf16_grouped_adamw: 1 = also store the gradient to C */
  float* params; float* grads; float* adam_m; float* adam_v; void* params16;
} nv_adamw_arena;
int nv_gemm_bf16_grouped_adamw(int count, const nv_gemm_problem* problems, const nv_adamw_arena* opt, void* stream);
int nv_adamw_ranges(const nv_adamw_arena* opt, const long* begins, const long* lens, int count, void* stream);

/* tuning aid: force the workgroup tile ((64,64), (64,128), (128,128): the general small-tile kernel); bm = 0 restores the built-in
 * heuristic; bm = 1 / 3 forces the warp-specialised 128 x 128 / 64 x 128 tile (bn: ring, 0 = heuristic, 1 = 3 x 64-deep, (3,3) =
 * 3 x 128-deep), bm = 4 the eight-wave 256 x 128 ping-pong kernel, bm = 5 forbids it, bm = 9 the 256 x 256 kernel; (6, n) sets the ping-pong
 * kernel's minimum tile count, (7, 0|1) switches the grouped weight-gradient launch between the two kernel families, (11, 0|1) runs
 * the NT problems of the 256 x 128 kernel on v_mfma_f32_32x32x16_bf16 instead of 16x16x32, (12, n) = workgroups of nv_gemm_bf16_grouped_adamw (walking its
 * tiles; 0 = one per tile; default 128), (13, n) = at most n workgroups per nv_adamw_ranges launch (0 = one per 2048-element chunk) */
int nv_gemm_set_tile(int bm, int bn);

/* ---- LayerNorm folded into the GEMMs around it - inference forwards ("fused LayerNorm" as GEMM prologue; vit_3d.py:18-19,37-41):
 *   LN(x) W^T + b  =  rstd (x Wg^T) - rstd mu colsum(Wg) + (W beta + b),   Wg = W diag(gamma)
 * so a block's two LayerNorm launches (and their normalised copies) disappear: the GEMM that PRODUCES the residual stream (out-projection, FC2:
 * nv_gemm_resid_ln = nv_gemm_bf16 epilogue 4 without dropout) also writes the rows in the operand format and, per row and 128-column tile, (mean, sum of squared
 * deviations); the GEMM that CONSUMES them (to_qkv, FC1: nv_gemm_lnfold) contracts the un-normalised rows with Wg and applies mu / rstd - merged from the tile
 * partials by Chan's update, no E[x^2] - E[x]^2 - in its epilogue.  nv_ln_fold_weight prepares Wg (operand format), colsum (of the rounded Wg) and the folded bias.
 * Kernels with an LDS epilogue only (nv_gemm_lnfold_supported: every ViT3D-base shape from batch 1 up); same MFMA mainloops as nv_gemm_bf16. */
int nv_gemm_lnfold_supported(int M, int N, int K);
long nv_ln_fold_stats_floats(int M, int d);
int nv_ln_fold_weight(const float* W, long ldw, int N, int K, const float* gamma, const float* beta, const float* bias, void* Wg16, long ldg,
                      float* colsum, float* fbias, void* stream);
int nv_gemm_resid_ln(int M, int N, int K, const void* A, long lda, const void* W, long ldw, const float* bias, const float* resid, long ldr, float* out,
                     long ldo, void* out16, long ldo16, float* stats, void* stream);
int nv_gemm_lnfold(int gelu, int M, int N, int K, const void* X16, long ldx, const void* Wg16, long ldw, const float* stats, const float* colsum,
                   const float* fbias, float eps, void* out16, long ldo, void* stream);

/* ---- Linear layers on a few rows (the cls rows of the last block under pool='cls'): weight-streaming kernels, rows addressed through
 * leading dimensions (a [B, n, d] tensor's cls rows: ld = n * d).  bf16 operands, fp32 accumulation, cast points of nv_gemm_bf16.
 * nv_skinny_nt: W [N, K] row-major.  epi 0: out f32 [R, N] = resid + (bias + A W^T)   epi 1: u = bias + A W^T (bf16, optional), out bf16 = gelu(u)
 * nv_skinny_nn: W [K, N] row-major.  epi 0: out bf16 = (A W) * gelu'(u), dcol[n] (+)= column sums of the stored values (optional, R <= 4)
 *                                    epi 1: out f32 = A W     epi 2: out bf16 = A W */
/* (revision 5) drop_seed / drop_p: the nn.Dropout of the site (vit_3d.py:21,23,45), as in nv_gemm_bf16 - nt epi 0 on (bias + A W^T), epi 1 on gelu(u),
 * nn epi 0 on A W (the mask of the GELU output the gradient flows back through).  The mask is the one of the DENSE [M, N] tensor `out` is a
 * row-strided view of (ldo a multiple of N): element (r, n) of the view hashes at its offset r * ldo + n. */
int nv_skinny_nt(int epi, int R, int N, int K, const void* A, long lda, const void* W, long ldw, const float* bias, const float* resid,
                 long ldr, void* out, long ldo, void* u_out, long ldu, unsigned long drop_seed, float drop_p, void* stream);
int nv_skinny_nn(int epi, int R, int N, int K, const void* A, long lda, const void* W, long ldw, const void* u, long ldu, void* out, long ldo,
                 float* dcol, int accumulate, unsigned long drop_seed, float drop_p, void* stream);
/* out bf16 [total_rows, N] dense = zeros, except rows r * keep_every (r < R) = A[r, :] W: one launch (dAO of the last block under
 * pool = 'cls', whose incoming gradient lives on the cls rows only - clearing the other rows used to be a memset node) */
int nv_skinny_nn_sparse(int R, int N, int K, const void* A, long lda, const void* W, long ldw, void* out, long total_rows, int keep_every,
                        void* stream);

/* ---- LayerNorm of the residual stream (vit_3d.py:18,37): x f32 [M,d] -> y bf16, saves mean / rstd */
int nv_ln_fwd(const float* x, long ldx, int M, int d, const float* gamma, const float* beta, float eps, void* y, long ldy,
              float* mean, float* rstd, void* stream);
long nv_ln_bwd_workspace_bytes(int M, int d);
/* g_out = g_in + dLN(dy); g16 = bf16(g_out * mask); dgamma / dbeta / dcolsum(= column sums of g_out * mask) optional.
 * mask = the dropout mask (drop_seed, drop_p) of the Linear output that was added to this residual stream (1 if p = 0). */
int nv_ln_bwd(const float* dy, long lddy, const float* x, long ldx, const float* mean, const float* rstd, const float* gamma,
              int M, int d, const float* g_in, float* g_out, long ldg, void* g16, long ldg16, float* dgamma, float* dbeta,
              float* dcolsum, int accumulate, void* workspace, long ws_bytes, unsigned long drop_seed, float drop_p,
              void* stream, void* reduce_stream);
/* reduce_stream (null = stream): where the dgamma / dbeta / dcolsum reduction of the per-workgroup partials runs; it is
 * ordered after the main kernel by an event, and `workspace` must stay untouched until that stream has executed it.
 * NV_LN_NO_REDUCE leaves the reduction to a later nv_ln_bwd_reduce (same M, d, workspace) on a stream the caller has
 * ordered after this call - lets several reductions share one cross-stream event. */
#define NV_LN_NO_REDUCE ((void*)(-1L))
int nv_ln_bwd_reduce(const void* workspace, int M, int d, float* dgamma, float* dbeta, float* dcolsum, int accumulate, void* stream);

/* Several reductions of per-workgroup partial sums in ONE launch: out[s][c] (+)= sum_r partials[r][s * width + c].  A layer's bias
 * and LayerNorm-affine gradients (nv_ln_bwd with NV_LN_NO_REDUCE: rows = nv_ln_bwd_partial_rows(M), nseg = 3, width = d;
 * nv_gemm_bf16 epilogue 6: rows = ceil(M / nv_gemm_tile_rows), nseg = 1, width = N) all become final at the same point of the
 * backward pass; one launch instead of one per tensor.  count <= 8.  Deterministic (fixed summation order). */
typedef struct nv_reduce_job {
  const float* partials;
  int rows, width, nseg;
  float* out[3];           /* NULL = segment skipped */
  int accumulate;
} nv_reduce_job;
int nv_reduce_multi(const nv_reduce_job* jobs, int count, void* stream);
int nv_ln_bwd_partial_rows(int M);

/* ---- input contract (src/data/DatasetADNI.py:212-213, DatasetADNI_4D.py:86-87): crop of the raw volume + z-score
 * (x - mean) / (std + eps), population std over the whole cropped sample, statistics accumulated in double.
 * raw [B,X,Y,Z,T] with element strides (T = 1 for 3D), dtype 0 = float32 / 1 = int16; crop8 = {x0,y0,z0,t0,Sx,Sy,Sz,St};
 * out dense float32 [B,Sx,Sy,Sz,St]; stats (optional) [B,2] = mean, std. */
long nv_zscore_crop_workspace_bytes(int B);
/* statistics only: sigma[b] = population std of cropped volume b + eps, mean[b] optional (workspace as nv_zscore_crop) */
int nv_volume_sigma(const void* raw, int dtype, const long* strides5, int B, const int* crop8, float eps, float* sigma, float* mean,
                    void* workspace, long ws_bytes, void* stream);
int nv_zscore_crop(const void* raw, int dtype, const long* strides5, int B, const int* crop8, float eps, float* out,
                   float* stats, void* workspace, long ws_bytes, void* stream);

/* ---- patch embedding front end (vit_3d.py:92-93 + the permute of NeuroEncoder.py:200-202)
 * video [B,C,F,H,W] f32 with arbitrary element strides (pass the strides of the permuted VIEW of the
 * [B,H,W,D] dataset tensor - no copy); out bf16 [B*N, ldo] = LayerNorm(patch_dim)(patches). */
int nv_patch_ln_fwd(const float* video, const long* strides5, int B, int C, int F, int H, int W, int p1, int p2, int pf,
                    const float* gamma, const float* beta, float eps, void* out, long ldo, float* mean, float* rstd,
                    const float* vol_sigma, void* stream);
/* Two forms of the same arithmetic (bit-identical rows): a wave per token gathering its 64-byte runs itself (default: measured 14.5 us at
 * ViT3D-base batch 4), or a workgroup per patch COLUMN that stages its p1 * p2 rows of F contiguous floats through LDS with fully
 * coalesced reads and serves its F / pf tokens from there (frame stride 1, channels 1, the slab within 156 KiB of LDS - what a [B, H, W, D]
 * volume gives; 21 us: load, then compute, one workgroup per CU).  nv_patch_set_mode: 0 = the first, 2 = the second where it applies. */
int nv_patch_set_mode(int mode);
/* vol_sigma (may be NULL): [B] = std + 1e-8 of each RAW volume (nv_volume_sigma).  `video` is then the un-normalised (cropped
 * view of the) scanner volume: LayerNorm over a patch of (x - mu) / sigma equals LayerNorm over the patch of x with eps * sigma^2,
 * so the dataset's z-score (src/data/DatasetADNI.py:213) is folded into this kernel's epsilon - no normalised copy is written.
 * nv_patch_ln_fwd_4d: all T timepoints of a 4D sample x [Bo, H, W, D, T] (contiguous, T % 4 == 0; src/data/DatasetADNI_4D.py:86-96)
 * in one pass - replaces the strided regroup copy of NeuroEncoder.py:54-56; token rows (bo*T + t)*N + n. */
int nv_patch_ln_fwd_4d(const float* x, int Bo, int H, int W, int D, int T, int p1, int p2, int pf, const float* gamma,
                       const float* beta, float eps, void* out, long ldo, float* mean, float* rstd, const float* vol_sigma,
                       void* stream);
/* fp32 tokens (fp32 inference path): out f32 [B*N, ldo], ldo >= patch_dim */
int nv_patch_ln_fwd_f32(const float* video, const long* strides5, int B, int C, int F, int H, int W, int p1, int p2, int pf,
                        const float* gamma, const float* beta, float eps, float* out, long ldo, float* mean, float* rstd,
                        const float* vol_sigma, void* stream);
int nv_patch_ln_fwd_4d_f32(const float* x, int Bo, int H, int W, int D, int T, int p1, int p2, int pf, const float* gamma,
                           const float* beta, float eps, float* out, long ldo, float* mean, float* rstd, const float* vol_sigma,
                           void* stream);
long nv_patch_ln_bwd_workspace_bytes(int tokens, int P);
int nv_patch_ln_bwd(const float* video, const long* strides5, int B, int C, int F, int H, int W, int p1, int p2, int pf,
                    const float* dxp, long ldd, const float* mean, const float* rstd, float* dgamma, float* dbeta,
                    int accumulate, void* workspace, long ws_bytes, void* stream);

/* ---- LayerNorm(dim) + cls token + positional embedding (vit_3d.py:95,116-118): t [B*N,d] -> x [B,N+1,d] */
int nv_embed_finish_fwd(const float* t, long ldt, int B, int N, int d, const float* gamma, const float* beta, float eps,
                        const float* pos, const float* cls, float* x, long ldx, float* mean, float* rstd,
                        unsigned long drop_seed, float drop_p, void* stream);
long nv_embed_finish_bwd_workspace_bytes(int B, int N, int d);
int nv_embed_finish_bwd(const float* g, long ldg, const float* t, long ldt, const float* mean, const float* rstd,
                        const float* gamma, int B, int N, int d, float* dt, long lddt, void* dt16, long lddt16, float* dgamma,
                        float* dbeta, float* dbias_pe, float* dpos, float* dcls, int accumulate, void* workspace, long ws_bytes,
                        unsigned long drop_seed, float drop_p, void* stream);

/* ---- multi-head attention core (vit_3d.py:51-59): qkv bf16 [B,n,3*inner] -> out bf16 [B,n,inner], lse f32 [B,heads,n] */
int nv_attn_set_mode(int mode);   /* testing aid: 0 = heuristic, 1 = streaming kernels, 2 = LDS-resident kernels (n <= 576), 3 = wide streaming forward;
                                    + 20: resident forward with two partner waves per row group (key range split, merged through LDS);
                                    + 100: resident backward as ONE launch whose dK / dV workgroups compute delta themselves, instead of two
                                    dependent launches (dQ, then dK / dV reading its delta): same results bit for bit, measured slower */
int nv_attn_fwd(const void* qkv, long ld_qkv, int B, int n, int heads, int dim_head, float scale, void* out, long ld_out,
                float* lse, unsigned long drop_seed, float drop_p, void* stream);
/* the same forward with the output as OCP e4m3 bytes of (value * out_scale): out = byte buffer [B*n, ld_out]; dim_head 64, no dropout, no lse */
int nv_attn_fwd_o8(const void* qkv, long ld_qkv, int B, int n, int heads, int dim_head, float scale, void* out, long ld_out,
                   float out_scale, void* stream);
int nv_attn_bwd(const void* qkv, long ld_qkv, const void* out, const void* dout, long ld_out, const float* lse, int B, int n,
                int heads, int dim_head, float scale, float* delta, void* dqkv, long ld_dqkv, unsigned long drop_seed,
                float drop_p, void* stream);
int nv_stream_sync(void* from, void* to);   /* stream `to` waits for everything enqueued so far on `from` (pooled events) */
int nv_spin_us(int microseconds, void* stream);   /* one wave that keeps `stream` busy for the given time (<= 50 ms): stream-placement probes */

/* ---- classification head (vit_3d.py:107-110,123-126): cls row -> LayerNorm -> Linear(dim, C), fp32 */
int nv_head_fwd(const float* x, long row_stride, int B, int d, const float* gamma, const float* beta, float eps,
                const float* W, const float* bias, int C, float* xh, float* stats, float* logits, void* stream);
long nv_head_bwd_workspace_bytes(int B, int d);
int nv_head_bwd(const float* dlogits, int B, int C, const float* W, const float* x, long row_stride, const float* stats,
                const float* xh, const float* gamma, int d, int n, float* g, long ldg, void* g16, long ldg16, float* dgamma,
                float* dbeta, float* dW, float* dbias, float* dcolsum, int accumulate, void* workspace, long ws_bytes,
                unsigned long drop_seed, float drop_p, int pool_mean, void* stream);
/* nv_head_fwd + nv_ce_loss + nv_head_bwd (pool = 'cls') in TWO launches instead of five, every output bit-identical to the three calls
 * (launch 1: one workgroup per volume runs the bodies of their per-volume kernels back to back, the other workgroups clear g / g16
 * outside the cls rows; launch 2: every sum over the volumes): mlp_head forward, nn.CrossEntropyLoss (mean) and their backward,
 * vit_3d.py:118-123,128-130 + Trainer.py:70,74.  workspace: nv_head_step_workspace_bytes(B, d). */
long nv_head_step_workspace_bytes(int B, int d);
int nv_head_step(const float* x, long row_stride, int B, int d, const float* gamma, const float* beta, float eps, const float* W,
                 const float* bias, int C, const long* labels, float grad_scale, float* xh, float* stats, float* logits, float* loss,
                 float* dlogits, int n, float* g, long ldg, void* g16, long ldg16, float* dgamma, float* dbeta, float* dW, float* dbias,
                 float* dcolsum, int accumulate, void* workspace, long ws_bytes, unsigned long drop_seed, float drop_p, void* stream);
/* pool='mean' (vit_3d.py:127): out[b,:] = mean_t x[b,t,:]; feed it to nv_head_fwd / nv_head_bwd with row_stride = d
   and pool_mean = 1 (every row of g then receives dx / n) */
int nv_token_mean(const float* x, int B, int n, int d, float* out, void* stream);

/* ---- bias gradients: out[c] (+)= sum_r X[r,c], X bf16 */
long nv_colsum_workspace_bytes(int M, int N);
int nv_colsum_bf16(const void* X, long ld, int M, int N, float* out, int accumulate, void* workspace, long ws_bytes,
                   void* stream);

/* ---- loss / optimizer (Trainer.py:30-31,70,75): nn.CrossEntropyLoss (mean) and torch.optim.AdamW */
int nv_ce_loss(const float* logits, const long* target, int B, int C, float grad_scale, float* loss, float* dlogits,
               void* stream);
int nv_adamw_step(float* p, const void* grad, int grad_bf16, float* m, float* v, void* p16, long count, int step, double lr,
                  double beta1, double beta2, double eps, double weight_decay, float grad_scale, int max_blocks, void* stream);
/* ---- dynamic loss scale for NV_OPERAND_FP16 training: torch.amp.GradScaler (src/Trainer.py:29,74-76) kept on the device - no
 * found_inf read-back per step.  `state`: NV_LOSS_SCALE_FLOATS floats of device memory owned by the caller; [0] = current scale,
 * [5] = optimizer updates applied so far, [11] = updates skipped (the rest: common.h LS_*).  Per optimizer step:
 *   nv_ce_loss_scaled / nv_head_step_scaled   d(loss)/d(logits) is multiplied by state[0] (the reported loss is not)
 *   nv_loss_scale_check(grads, count, state)  found_inf |= any inf / NaN in grads[0 .. count)   (after the backward pass / all-reduce)
 *   nv_loss_scale_update(state, lr, b1, b2)   decides skip or step: backoff / growth of the scale, AdamW's step count and bias
 *                                             corrections (double arithmetic, as torch forms them), 1 / scale for the update
 *   nv_adamw_step_scaled(..., state, stream)  nv_adamw_step that does nothing when the step is skipped and un-scales the gradients
 *                                             (its `step` argument is ignored: the state's own count of applied updates is used)
 * GradScaler defaults: init_scale 65536, growth_factor 2, backoff_factor 0.5, growth_interval 2000.  start_step = updates already
 * applied to the optimizer state. */
#define NV_LOSS_SCALE_FLOATS 16
int nv_loss_scale_init(float* state, float init_scale, float growth_factor, float backoff_factor, int growth_interval, int start_step, void* stream);
int nv_loss_scale_check(const float* grads, long count, float* state, void* stream);
int nv_loss_scale_update(float* state, double lr, double beta1, double beta2, void* stream);
int nv_ce_loss_scaled(const float* logits, const long* target, int B, int C, float grad_scale, const float* scale_state, float* loss,
                      float* dlogits, void* stream);
int nv_adamw_step_scaled(float* p, const void* grad, int grad_bf16, float* m, float* v, void* p16, long count, int step, double lr,
                         double beta1, double beta2, double eps, double weight_decay, float grad_scale, int max_blocks,
                         const float* scale_state, void* stream);
int nv_head_step_scaled(const float* x, long row_stride, int B, int d, const float* gamma, const float* beta, float eps, const float* W,
                        const float* bias, int C, const long* labels, float grad_scale, const float* scale_state, float* xh, float* stats,
                        float* logits, float* loss, float* dlogits, int n, float* g, long ldg, void* g16, long ldg16, float* dgamma,
                        float* dbeta, float* dW, float* dbias, float* dcolsum, int accumulate, void* workspace, long ws_bytes,
                        unsigned long drop_seed, float drop_p, void* stream);
/* grad_bf16 = 1: `grad` is a bf16 buffer (the gradient all-reduce ran on bf16 messages): no cast back to fp32 is needed. */
/* max_blocks > 0 caps the grid (256-thread workgroups, grid-stride): used when the update of one gradient bucket runs on a
   side stream beside the backward pass, so that it takes a slice of the chip instead of queueing ahead of the GEMMs. */
int nv_cast_bf16_2d(const float* src, long ld_src, int rows, int cols, void* dst, long ld_dst, void* stream);
/* out16 (bf16) / out32 (f32), either may be NULL: x[M,N] f32 times the nn.Dropout mask of one site (same mask as the GEMM
 * epilogues, element index m*N + n; drop_p = 0: plain cast / copy).  Standalone Attention / FeedForward modules (vit_3d.py:23,45). */
int nv_dropout_apply(const float* x, long ldx, int M, int N, unsigned long drop_seed, float drop_p, void* out16, long ld16,
                     float* out32, long ld32, void* stream);
int nv_copy_2d_f32(const float* src, long ld_src, int rows, int cols, float* dst, long ld_dst, int accumulate, void* stream);

/* ---- Grad-CAM reduction (src/models/NeuroEncoder.py:101-116): act bf16 [B,n,d] = output of the last block's attention
 * LayerNorm, grad f32 [B,n,d] = its gradient -> cam f32 [B, n-1]: relu(mean_d(grad) * sum_d(act)) of the patch tokens
 * (cls dropped), min-max normalised over the whole map ((x - min) / (max - min + 1e-8)); minmax (optional) [2] = the raw
 * min / max.  One launch; the percentile threshold and the trilinear upsampling of the G^3 map stay with the caller. */
long nv_gradcam_workspace_bytes(int B, int n);
int nv_gradcam_reduce(const void* act, const float* grad, int B, int n, int d, float* cam, float* minmax, void* workspace,
                      long ws_bytes, void* stream);

/* ---- the 4D model's temporal head (src/models/NeuroEncoder.py:60-66: temporal_transformer -> mean over time -> projection_head;
 * :207-217 TemporalTransformer = one nn.TransformerEncoderLayer(d_model 2, nhead 2, batch_first, post-norm, ReLU, dim_feedforward ff,
 * dropout p at its four sites); :219-230 ProjectionHead = nn.Linear(2, 2)) - ONE launch per direction.
 * x [B, T, 2] f32 contiguous (the frozen encoder's logits per timepoint), T <= 64, ff <= 2048; out [B, 2].
 * params / grads: one flat fp32 arena of nv_temporal_head_param_count(ff) = 40 + 5 ff floats in named_parameters() order:
 *   in_proj_weight [6,2] | in_proj_bias [6] | out_proj.weight [2,2] | out_proj.bias [2] | linear1.weight [ff,2] | linear1.bias [ff] |
 *   linear2.weight [2,ff] | linear2.bias [2] | norm1.weight | norm1.bias | norm2.weight | norm2.bias [2 each] |
 *   projection_head.weight [2,2] | projection_head.bias [2].
 * drop_p > 0 (training): counter-based masks from drop_seed; the backward call recomputes the forward from x (nothing else is saved)
 * and must be given the forward's seed and p.  accumulate != 0 adds into grads; dx ([B, T, 2], may be NULL) receives the input gradient. */
long nv_temporal_head_param_count(int ff);
int nv_temporal_head_fwd(const float* x, int B, int T, int ff, const float* params, float eps, unsigned long drop_seed, float drop_p,
                         float* out, void* stream);
int nv_temporal_head_bwd(const float* x, int B, int T, int ff, const float* params, float eps, unsigned long drop_seed, float drop_p,
                         const float* dout, float* grads, int accumulate, float* dx, void* stream);

/* ---- whole-encoder engine: ViT.forward / its backward as ONE call each (vit_3d.py:112-126)
 * Parameters live in one flat fp32 arena (+ a bf16 shadow with identical element offsets) laid out by
 * nv_vit_param_table in the reference's state_dict order; gradients go to an arena of the same layout. */
typedef struct nv_vit_config {
  int image_size, image_patch_size, frames, frame_patch_size;
  int channels, num_classes, dim, depth, heads, dim_head, mlp_dim;
  float ln_eps;
  int pool_mean;   /* 0: pool='cls' (NeuroEncoder.py:194), 1: pool='mean' (vit_3d.py:127) */
  int image_width, patch_width;   /* vit_3d.py:80-81 takes (height, width) pairs: image_size / image_patch_size are the HEIGHTS,
                                     these the widths; 0 = square (the NeuroEncoder path is cubic: NeuroEncoder.py:183-186) */
  int no_proj_dropout;            /* (revision 6) 1: no nn.Dropout behind the output projection - the heads == 1 && dim_head == dim geometry, whose
                                     to_out is nn.Identity() (vit_3d.py:32,43-46); the other three dropout sites of a block are unaffected */
} nv_vit_config;

/* heads == 1 && dim_head == dim: the reference has no output projection (vit_3d.py:32,43-46) but the table still carries every
 * block's to_out weight / bias slot: fill them with the identity / zeros and leave them out of the optimizer (x + I ao + 0 == x + ao
 * exactly, and the data gradient g I == g); neurovit_amd.ViT does that. */
long nv_vit_param_count(const nv_vit_config* cfg);
int nv_vit_param_table(const nv_vit_config* cfg, long* offsets, long* numels, int max_entries);
/* training: 0 = bf16 / fp8 inference layout, 1 = training layout (every layer's activations kept), 2 = fp32 inference layout */
long nv_vit_workspace_bytes(const nv_vit_config* cfg, int B, int training);
/* byte offset of a named activation inside the workspace (-1 if unknown); layer < 0 for global buffers */
long nv_vit_workspace_offset(const nv_vit_config* cfg, int B, int training, const char* name, int layer);
/* shape5 = {B, C, F, H, W} of `video`: must equal {B, cfg.channels, cfg.frames, cfg.image_size, width (cfg.image_width or cfg.image_size)} (the reference
 * fails in einops / the pos_embedding add for any other volume, vit_3d.py:92,118; here a wrong extent would be gathered out of
 * bounds, so it is rejected with NV_ERR_ARG).
 * drop_p / emb_drop_p / drop_seed: nn.Dropout of the blocks (vit_3d.py:21,23,39,45) and of the embedding (:100); both 0
 * in eval mode.  backward must be given the forward's values. */
/* Optional input forms, nv_vit_forward_in / nv_vit_forward_fp8:
 *   vol_sigma   != NULL: `video` holds RAW volumes; [B] (or [B / time_points]) = std + 1e-8 per sample (nv_volume_sigma): the z-score
 *                        is folded into the patch LayerNorm (see nv_patch_ln_fwd); crop = the strides / base pointer of the view;
 *   time_points  > 0   : `video` is a contiguous 4D batch [B / T, H, W, D, T] (shape5 = that shape) and volume b*T + t is timepoint t
 *                        of sample b - no regroup copy (nv_patch_ln_fwd_4d; T % 4 == 0, channels = 1); strides5 is ignored;
 *   rows_form          : which rows of the LAST block's out-projection / LayerNorm / FeedForward are computed - see nv_vit_set_cls_tail.
 *                        A backward must be given the rows_form (and dropout) of its forward. */
typedef struct nv_vit_input {
  const float* vol_sigma;
  int time_points;
  int rows_form;   /* last block under pool='cls': 0 = process default (nv_vit_set_cls_tail), 1 = every row, 2 = cls rows when eligible */
} nv_vit_input;
int nv_vit_forward_in(const nv_vit_config* cfg, int B, const float* video, const long* shape5, const long* strides5,
                      const nv_vit_input* in, const float* params, const void* params16, void* workspace, long ws_bytes,
                      int training, float drop_p, float emb_drop_p, unsigned long drop_seed, float* logits, void* stream);
int nv_vit_forward(const nv_vit_config* cfg, int B, const float* video, const long* shape5, const long* strides5,
                   const float* params, const void* params16, void* workspace, long ws_bytes, int training, float drop_p, float emb_drop_p,
                   unsigned long drop_seed, float* logits, void* stream);
/* Inference forward with the blocks' LayerNorms folded into the GEMMs around them (nv_gemm_resid_ln / nv_gemm_lnfold above): nv_vit_forward_in(training = 0)
 * without 21 of ViT3D-base's 24 LayerNorm launches (not folded: LN1 of block 0, LN1 of the last block - the Grad-CAM hook tensor - and a last block on its cls
 * rows).  fold16: 16-bit arena (operand format) with the parameter arena's element offsets; fold32: nv_vit_lnfold_floats floats; both filled by
 * nv_vit_lnfold_prepare whenever the parameters have changed.  Falls back to the unfolded launches for shapes outside the LDS-epilogue kernels. */
long nv_vit_lnfold_floats(const nv_vit_config* cfg);
int nv_vit_lnfold_prepare(const nv_vit_config* cfg, const float* params, void* fold16, float* fold32, void* stream);
int nv_vit_forward_lnfold(const nv_vit_config* cfg, int B, const float* video, const long* shape5, const long* strides5, const nv_vit_input* in,
                          const float* params, const void* params16, const void* fold16, const float* fold32, void* workspace, long ws_bytes,
                          float* logits, void* stream);
/* fp32 inference forward: ViT.forward (vit_3d.py:112-126) as the reference's fp32 validate computes it (Trainer.py:101-118) - every
 * operand fp32 (weights straight from `params`, no shadow arena), contractions on the fp32 MFMA, eval mode (no dropout).
 * Workspace: nv_vit_workspace_bytes(cfg, B, 2).  Input forms as nv_vit_forward_in. */
int nv_vit_forward_f32(const nv_vit_config* cfg, int B, const float* video, const long* shape5, const long* strides5,
                       const nv_vit_input* in, const float* params, void* workspace, long ws_bytes, float* logits, void* stream);
/* fp8 inference forward (BASELINE.json configs[4] "ViT3D-large ... fp8 MFMA"): all four linears of every block - qkv, the
 * out-projection (its operand written as e4m3 by the attention kernel itself: nv_attn_fwd_o8), FC1, FC2 - on e4m3 operands.
 * act_scales: HOST array [depth][4] (LN1 output, LN2 output, GELU output, attention output - <= 0 keeps that block's out-projection on
 * bf16 operands; calibrated: 448 / (headroom * amax)); params8: byte arena
 * with the element offsets of the parameter arena; colscales: f32 [nv_vit_fp8_scale_count].  Workspace: training = 0 layout. */
long nv_vit_fp8_scale_count(const nv_vit_config* cfg);
int nv_vit_quantize_fp8(const nv_vit_config* cfg, const float* params, const float* act_scales, void* params8, float* colscales, void* stream);
int nv_vit_forward_fp8(const nv_vit_config* cfg, int B, const float* video, const long* shape5, const long* strides5,
                       const nv_vit_input* in, const float* params, const void* params16, const void* params8, const float* colscales,
                       const float* act_scales, void* workspace, long ws_bytes, float* logits, void* stream);
/* fp8 TRAINING forward: as nv_vit_forward_in(training = 1) - every activation the backward pass reads is written, in bf16 / fp32, where
 * the bf16 forward writes it - with qkv, FC1 and FC2 of every block on e4m3 operands (params8 / colscales / act_scales as for
 * nv_vit_forward_fp8; act_scales[4 l + 3], the out-projection's, is ignored: that linear stays on bf16 operands, as do attention, the
 * patch embedding, the head and a last block in the cls-rows form).  drop_p / emb_drop_p / drop_seed as for nv_vit_forward_in (the masks
 * of the four block sites are those of the bf16 forward: the backward pass recomputes them from the same arguments).
 * Follow it with nv_vit_backward[_stages16] exactly as after nv_vit_forward_in; re-quantise the weights (nv_vit_quantize_fp8) after
 * every optimizer step.  Workspace: the training layout (nv_vit_workspace_bytes(cfg, B, 1)). */
int nv_vit_forward_fp8_train(const nv_vit_config* cfg, int B, const float* video, const long* shape5, const long* strides5,
                             const nv_vit_input* in, const float* params, const void* params16, const void* params8, const float* colscales,
                             const float* act_scales, void* workspace, long ws_bytes, float drop_p, float emb_drop_p, unsigned long drop_seed,
                             float* logits, void* stream);
int nv_vit_backward(const nv_vit_config* cfg, int B, const float* video, const long* strides5, const float* params,
                    const void* params16, void* workspace, long ws_bytes, const float* dlogits, float* grads,
                    int accumulate, float drop_p, float emb_drop_p, unsigned long drop_seed, void* stream, void* aux_stream);

/* aux_stream (may be NULL): a second hipStream_t on which the weight-gradient GEMMs run concurrently with the data-gradient
 * chain; the engine forks / joins with pooled events.
 * Backward split into stages (0 = head, 1+k = layer depth-1-k, depth+1 = patch embedding) so the caller can start the
 * data-parallel all-reduce of a stage's gradient range (nv_vit_stage_param_range) while later stages still run.
 * join_aux = 1: the call returns with `stream` ordered after all of its work on both streams.  join_aux = 0 (allowed for
 * ranges that do not contain the last stage): `stream` is NOT made to wait for the auxiliary stream - the range's gradients
 * are complete once BOTH streams have executed what this call enqueued, so the consumer (the all-reduce) must be ordered
 * after both; the next call on the same workspace picks the dependency up where this one left it. */
int nv_vit_backward_stages(const nv_vit_config* cfg, int B, const float* video, const long* strides5, const float* params,
                           const void* params16, void* workspace, long ws_bytes, const float* dlogits, float* grads,
                           int accumulate, int first_stage, int last_stage, float drop_p, float emb_drop_p,
                           unsigned long drop_seed, void* stream, void* aux_stream, int join_aux);
/* as nv_vit_backward_stages; grads16 (may be NULL): bf16 arena with the element offsets of `grads` - the weight gradients of the
 * Linear layers (to_qkv, to_out, FC1, FC2 of every block; the patch embedding's when patch_dim % 8 == 0) are ALSO written there,
 * rounded, by the GEMMs that produce them: a data-parallel caller sends them without a cast pass and converts only the small
 * remaining ranges (nv_cast_ranges_bf16). */
int nv_vit_backward_stages16(const nv_vit_config* cfg, int B, const float* video, const long* strides5, const float* params,
                             const void* params16, void* workspace, long ws_bytes, const float* dlogits, float* grads, void* grads16,
                             int accumulate, int first_stage, int last_stage, float drop_p, float emb_drop_p,
                             unsigned long drop_seed, void* stream, void* aux_stream, int join_aux, int rows_form);
int nv_vit_stage_param_range(const nv_vit_config* cfg, int stage, long* begin, long* end);
/* pool='cls': the last block's out-projection / LayerNorm / FeedForward, forward and backward, on the B cls rows only (whenever the
 * block dropout is off; training additionally B <= 4).  Logits and every gradient are unchanged (the other rows never reach the
 * head and receive exact zeros in the backward pass), but rows 1..n-1 of the last block's x1 / xn2 / u / h / x2 workspace buffers
 * are not produced.  The form is chosen PER CALL by rows_form (nv_vit_input / nv_vit_backward_stages16); this sets the process
 * default used by rows_form = 0 and by the entry points without that argument: 1 (initial) = cls rows, 0 = every row. */
int nv_vit_set_cls_tail(int on);
int nv_vit_set_head_step(int on);   /* A/B aid: 0 = nv_vit_train_step runs the head as nv_head_fwd + nv_ce_loss + nv_head_bwd (same bits), 1 (default) = nv_head_step */

/* ---- the reference's whole train step (src/Trainer.py:65-79) as ONE call: ViT forward (training layout) -> nn.CrossEntropyLoss
 * (mean) -> backward of every stage -> torch.optim.AdamW update of the whole arena (+ bf16 shadow refresh).  ~225 kernel launches
 * enqueued from native code with no interpreter in between; nothing synchronises, `loss` and `logits` stay on the device.
 * hp: see the struct (struct_size = sizeof(nv_train_hparams): checked, NV_ERR_ARG on mismatch).  labels: int64 [B].
 * logits / dlogits: f32 [B, num_classes] (dlogits is scratch the backward reads); loss: f32 [1].
 * accumulate = 1: gradients are added to `grads` (micro-steps 2.. of an accumulation window); update = 0: no optimizer update
 * (every micro-step but the last).  The arguments a separate backward would need (rows_form of `in`, dropout) are the forward's
 * by construction. */
/* ---- data-parallel train step from native code (no counterpart in the reference, which is single-device: main.py:41-46).
 * RCCL is bound at run time (dlopen; nv_comm_load(path) names the copy to use - the one torch's "nccl" backend has loaded - or NULL for
 * the default search).  Communicator: rank 0 calls nv_comm_unique_id (128 bytes), every rank receives those bytes over a channel of its
 * own (one torch.distributed broadcast) and calls nv_comm_init.  nv_comm_all_reduce: in-place SUM on `stream`; dtype 0 = f32, 1 = the
 * 16-bit operand format.
 * nv_dp_plan (nv_train_hparams.dp): the backward pass of nv_vit_train_step runs as n_buckets groups of stages; as soon as a group has
 * been enqueued, comm_stream waits for it and all-reduces the group's (contiguous) gradient range while the main stream continues -
 * fp32 in `grads` itself, or, with grads16 != NULL, as 16-bit messages in that arena (the Linear weight gradients are written there by
 * their GEMMs, the small ranges are converted on comm_stream).  update_per_bucket = 1 queues AdamW of that range (grad * grad_scale /
 * world) behind its all-reduce on comm_stream; 2 = on aux_stream, one bucket late (in front of the next bucket's weight-gradient work,
 * once the all-reduce has finished: the update then never runs beside the weight-gradient GEMMs; the last bucket's on `stream`);
 * 0 = one update over the arena when every bucket is in.  With grads16 the update reads
 * the reduced messages and `grads` keeps the LOCAL gradients.  Micro-steps with update = 0 run no collective. */
typedef struct nv_dp_plan {
  int struct_size;
  int world;               /* ranks: the update scales the summed gradients by 1 / world */
  void* comm;              /* nv_comm_init */
  void* comm_stream;       /* hipStream_t of the collectives; must not share a hardware queue with `stream` / `aux_stream` for overlap */
  int n_buckets;           /* 1 .. depth + 2 */
  int update_per_bucket;
  void* grads16;           /* NULL = fp32 messages */
} nv_dp_plan;
int nv_comm_load(const char* path);
int nv_comm_unique_id(void* id128);
int nv_comm_init(const void* id128, int world, int rank, void** comm);
int nv_comm_destroy(void* comm);
int nv_comm_all_reduce(void* comm, void* buf, long count, int dtype, void* stream);

typedef struct nv_train_hparams {
  int struct_size;
  int step;                 /* AdamW step count (>= 1) of this update: bias corrections (ignored when update = 0) */
  double lr, beta1, beta2, eps, weight_decay;
  float grad_scale;         /* the update reads grad * grad_scale */
  int accumulate, update;
  int fuse_update;          /* (revision 5) with update = 1, accumulate = 0: where the Linear weights of the transformer layers (96 % of the
                             * parameters) are updated.  0 = with everything else, one nv_adamw_step behind the backward pass.
                             * 3 = per layer, by an AdamW launch on the auxiliary stream behind that layer's weight-gradient GEMMs, while
                             * the main stream is already in the next layer (gradients stay in `grads`).  1 = by the weight-gradient GEMMs
                             * themselves (nv_gemm_bf16_grouped_adamw; the gradients of those weights are then NOT left in `grads`),
                             * 2 = the same and they are.  In 1 .. 3 the rest of the arena is updated by one nv_adamw_ranges launch.
                             * Parameters, optimizer state and losses are bit-identical in all four */
  float loss_scale;         /* (revision 6) static loss scale: d(loss)/d(logits) is multiplied by it and the update divides it out again
                             * (0 or 1 = none) - what keeps the 16-bit gradient tensors of NV_OPERAND_FP16 away from the subnormals; a power
                             * of two changes no bit of a finite result.  No overflow check: use loss_scale_state for GradScaler semantics */
  const struct nv_dp_plan* dp;   /* (revision 6) data-parallel step: NULL, or the plan below */
  float* loss_scale_state;  /* (revision 6) device block of nv_loss_scale_init or NULL: dynamic loss scale (torch.amp.GradScaler, Trainer.py:29,
                             * 74-76) - the step is scaled by its current value, every gradient is checked for inf / NaN after the backward
                             * pass, and the update is applied or skipped on the device (requires fuse_update = 0; with accumulate / update
                             * = 0 micro-steps only the scaling happens) */
} nv_train_hparams;
int nv_vit_train_step(const nv_vit_config* cfg, int B, const float* video, const long* shape5, const long* strides5, const nv_vit_input* in,
                      float* params, void* params16, float* grads, float* adam_m, float* adam_v, void* workspace, long ws_bytes,
                      const long* labels, float* logits, float* loss, float* dlogits, const nv_train_hparams* hp,
                      float drop_p, float emb_drop_p, unsigned long drop_seed, void* stream, void* aux_stream);

/* diagnostic: where do the workgroups of a grid run?  out u32 [blocks][2] = (HW_REG_HW_ID, HW_REG_XCC_ID) of each workgroup, which then
 * holds its CU for hold_us microseconds (threads per workgroup / dynamic LDS bytes shape its footprint).  Used to read the CU set of a
 * CU-masked stream (hipExtStreamCreateWithCUMask) and the XCD placement the 1-D grids rely on for speed. */
int nv_cu_census(unsigned* out, int blocks, int threads, int lds_bytes, int hold_us, void* stream);

/* ABI revision of this header: bumped whenever a struct gains a field or an entry point changes its argument list (the list is in
 * INTEGRATION.md "ABI revisions").  A caller built against revision R must refuse a library whose nv_abi_version() != R. */
#define NV_ABI_VERSION 6
int nv_abi_version(void);
/* dst[b .. b + len) = bf16(src[b .. b + len)) for `count` element ranges (HOST arrays begins / lens; any count) */
int nv_cast_ranges_bf16(const float* src, void* dst, const long* begins, const long* lens, int count, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NEUROVIT_HIP_H */
