// Flash-style multi-head attention for the ViT3D encoder (vit_3d.py:48-59), gfx950, dim_head = 64.
//
// Input is the packed projection qkv [B, n, 3*inner] bf16 ('b n (h d)' per chunk); output is
// 'b n (h d)' bf16 - the rearranges of the reference are pure addressing here.  n = N+1 is never
// tile aligned (513, 65, 1001): keys beyond n are zero-filled and masked to -inf, query rows
// beyond n are computed on clamped loads and never stored.
//
// MFMA orientation (v_mfma_f32_16x16x32_bf16, one wave = 16 query rows):
//   S^T = K . Q^T      -> accumulator lane (r,g) holds S[q = r][key = 16t + 4g + reg]: the softmax row index
//                         is the LANE, so row max/sum need only two shuffles (xor 16, 32), and
//   O^T = V^T . P^T    -> the P accumulators ARE the next MFMA's B operand (no LDS round trip); V^T comes
//                         from the row-major V tile through ds_read_b64_tr_b16.
// The k index of the second product is permuted (element j<4 <- key tile 2ks, j>=4 <- key tile 2ks+1) and the
// transposed V reads use the same permutation (row groups 4g and 16+4g).
// Backward = two deterministic kernels (no atomics): dQ per query tile, dK/dV per key tile, both
// recomputing P from the saved log-sum-exp.
#include "common.h"

// attention_generic.hip: any other head dim (multiples of 8 up to 128), scalar kernels
bool attn_generic_supported(int dim_head);
int launch_attn_generic_fwd(const void* qkv, long ld, int B, int n, int heads, int dh, float scale, void* out, long ldo, float* lse, DropCfg drop,
                            hipStream_t s);
int launch_attn_generic_bwd(const void* qkv, long ld, const void* out, const void* dout, long ldo, const float* lse, int B, int n, int heads, int dh,
                            float scale, float* delta, void* dqkv, long ldd, DropCfg drop, hipStream_t s);

constexpr int DH = 64;            // dim_head (reference default, vit_3d.py:29)
constexpr int TQ = 64;            // rows per workgroup (4 waves x 16)
constexpr int TK = 64;            // keys per LDS tile
constexpr int IMG = TK * DH * 2;  // 8 KiB image

// stage one [64 rows][64 bf16] tile: rows row0.. of X (row stride ld), zero-fill rows >= nrows
__device__ __forceinline__ void tile_gload(const r16* X, long ld, int row0, int nrows, int tid, uint4 (&reg)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = tid + 256 * i, row = row0 + (c >> 3);
    const bool ok = row < nrows;
    const uint4 v = *reinterpret_cast<const uint4*>(X + (ok ? (long)row * ld + ((c & 7) << 3) : 0));
    reg[i] = ok ? v : make_uint4(0, 0, 0, 0);
  }
}
__device__ __forceinline__ void tile_swrite(char* img, int tid, const uint4 (&reg)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = tid + 256 * i;
    *reinterpret_cast<uint4*>(img + img128_off(c >> 3, c & 7)) = reg[i];
  }
}
// row-read fragment: lane (r,g) <- X[16t + r][32ks + 8g .. +7]
__device__ __forceinline__ r16x8 frag_row(const char* img, int t, int ks, int r, int g) {
  return *reinterpret_cast<const r16x8*>(img + img128_off(16 * t + r, 4 * ks + g));
}
// transposed fragment: lane (r,g) <- { X[32ks + 4g + j][16t + r] (j<4), X[32ks + 16 + 4g + j-4][16t + r] (j>=4) }
__device__ __forceinline__ r16x8 frag_tr(const char* img, int t, int ks, int r, int g) {
  const int q = r >> 2, p = r & 3;
  const int ch = 2 * t + (p >> 1), sub = (p & 1) << 3;
  const r16x4 lo = lds_read_tr(img + img128_off(32 * ks + 4 * g + q, ch) + sub);
  const r16x4 hi = lds_read_tr(img + img128_off(32 * ks + 16 + 4 * g + q, ch) + sub);
  return cat4(lo, hi);
}
// bare v_exp_f32: arguments here are <= 0 and results in [0, 1]; results below 2^-126 flush to zero (they vanish in the sums)
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
// Exchange with the lanes l ^ 16 and l ^ 32 (the four lane groups that share a score row) on the VALU: v_permlane16_swap /
// v_permlane32_swap (gfx950) swap the odd 16- / 32-lane rows of one register with the even rows of another, so two copies of v
// come back as (v[l & ~16], v[l | 16]) - their max / sum is the pair's in every lane.  __shfl_xor compiles to ds_bpermute_b32:
// an LDS-crossbar round trip the wave waits for (lgkmcnt(0)) twice per key tile, in a loop with two waves per SIMD to hide it.
// (asm: the builtin form lost the combining fmaxf at -O3.)
__device__ __forceinline__ void lane_pair16(float v, float& a, float& b) {
  a = v; b = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void lane_pair32(float v, float& a, float& b) {
  a = v; b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
// max of two / three (v_max_f32 / v_max3_f32).  Written as inline asm these lost the IEEE canonicalisation (v_max_f32 x, x, x) the
// compiler puts in front of fmaxf operands, but an asm consumer placed right behind an MFMA is outside the compiler's hazard
// bookkeeping (a NaN showed up in a test): plain builtins.  Scalar f32 helpers in the same style were tried for the softmax
// arithmetic and measured 10 % slower than the packed forms.
__device__ __forceinline__ float vmax(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ float vmax3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
// ... and the bare instruction for operands that do NOT come out of an MFMA (the cross-lane maximum behind the lane swaps, the running
// maximum): __builtin_fmaxf is llvm.maxnum, which must quiet signalling NaNs, so hipcc puts a canonicalising `v_max_f32 x, x, x` in
// front of every operand it cannot prove canonical - six of the ~75 VALU instructions of a 16-score softmax step, in VALU-bound kernels.
// No NaN reaches these (-inf for masked keys is handled by the instruction as by fmaxf): bit-identical results.
__device__ __forceinline__ float vmax_bare(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float group_max(float v) {   // over the 4 lane groups sharing r
  float a, b;
  lane_pair16(v, a, b);
  v = vmax_bare(a, b);
  lane_pair32(v, a, b);
  return vmax_bare(a, b);
}
__device__ __forceinline__ float group_sum(float v) {
  float a, b;
  lane_pair16(v, a, b);
  v = a + b;
  lane_pair32(v, a, b);
  return a + b;
}

// ------------------------------------------------------------------------------------------------ forward
// One 64-key tile of the forward pass for one wave (16 query rows), in pieces so that the resident kernel can software-
// pipeline the LDS fragment reads of the NEXT products under the softmax arithmetic; the streaming kernel runs the same
// pieces back to back (identical arithmetic, bit-identical results).
// The softmax is VALU-bound (head dim 64: 16 MFMAs per tile against ~16 scores per lane), so it is written for the
// packed fp32 pipe (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32), v_max3_f32 and the bare v_exp_f32.
struct RowFrags { r16x8 f[2][4]; };    // [ks][t]: fragments of one 64-row operand tile
__device__ __forceinline__ void load_row_frags(const char* img, int r, int g, RowFrags& F) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int t = 0; t < 4; ++t) F.f[ks][t] = frag_row(img, t, ks, r, g);
}
__device__ __forceinline__ void load_tr_frags(const char* img, int r, int g, RowFrags& F) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int t = 0; t < 4; ++t) F.f[ks][t] = frag_tr(img, t, ks, r, g);
}
// acc[t] (+)= X_tile[16t.., :] . y   (ks outer: four independent accumulators between dependent MFMAs)
template <typename T>
__device__ __forceinline__ void mfma_rows(const RowFrags& F, const r16x8 (&y)[2], f32x4 (&acc)[4], bool zero) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (zero) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = mfma16<T>(F.f[ks][t], y[ks], acc[t]);
}
// online softmax of one tile's raw scores s (in place -> probabilities, dropout applied), running max m / sum l, rescale of o
template <bool DROP>
__device__ __forceinline__ void fwd_softmax(f32x4 (&s)[4], int kt, bool last, int n, f32x4 (&o)[4], float& m, float& l, float scale_log2e,
                                            const DropCfg& drop, int bh, int qabs, int g) {
    // bookkeeping on the RAW scores (scale_log2e > 0, so max commutes with the scaling); the scaling itself is
    // folded into one fma in front of exp2.  Only the last key tile can hold out-of-range keys.
    if (last) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (kt * TK + 16 * t + 4 * g + j >= n) s[t][j] = -INFINITY;
    }
    float mloc = vmax3(s[0][0], s[0][1], s[0][2]);
    mloc = vmax3(mloc, s[0][3], s[1][0]);
#pragma unroll
    for (int t = 1; t < 4; ++t) {
      if (t > 1) mloc = vmax3(mloc, s[t - 1][3], s[t][0]);
      mloc = vmax3(mloc, s[t][1], s[t][2]);
    }
    mloc = vmax(mloc, s[3][3]);
    const float mnew = vmax_bare(m, group_max(mloc) * scale_log2e);
    const float alpha = (mnew != m) ? fast_exp2(m - mnew) : 1.0f;      // m = mnew = -inf cannot make a NaN this way
    m = mnew;
    const f32x2 c2 = {scale_log2e, scale_log2e}, m2 = {-mnew, -mnew};
    f32x2 ps = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        f32x2 x = {s[t][2 * hh], s[t][2 * hh + 1]};
        x = __builtin_elementwise_fma(x, c2, m2);
        x[0] = fast_exp2(x[0]);
        x[1] = fast_exp2(x[1]);
        ps += x;
        s[t][2 * hh] = x[0];
        s[t][2 * hh + 1] = x[1];
      }
    const float psum = ps[0] + ps[1];
    // branch-free: a factor of exactly 1 where the maximum stood still (bit-identical to skipping the multiply; the skipping
    // form made the compiler copy all sixteen accumulators on the not-taken side to rejoin the two paths)
    l = __builtin_fmaf(l, alpha, psum);        // explicit shape: the same rounding in every kernel that inlines this
    // (round 3: skipping the multiply under a wave-uniform ballot when no lane's maximum moved - no accumulator copies this time -
    //  measured 1.5 % SLOWER on the ViT3D-large forward and 0.7 % on the batch-64 base forward: the branch costs more than 8 packed multiplies)
#pragma unroll
    for (int t = 0; t < 4; ++t) o[t] *= alpha;
    if constexpr (DROP) {      // attention dropout (vit_3d.py:56): mask the probabilities that enter P.V, not the normaliser
      // element (bh, q, key) has index (bh * n + q) * npad + key with npad = n rounded up to 4: every lane's four keys share one hash
      const unsigned long long base = (((unsigned long long)bh * n + qabs) * ((n + 3) & ~3)) + (unsigned long long)kt * TK;
#pragma unroll
      for (int t = 0; t < 4; ++t) s[t] *= drop_factor4(drop, base + 16 * t + 4 * g);
    }
}
// o[t] += V_tile^T[16t.., :] . P^T   (the probability accumulators are the MFMA B operand)
template <typename T>
__device__ __forceinline__ void fwd_pv(const RowFrags& V, const f32x4 (&s)[4], f32x4 (&o)[4]) {
  const r16x8 pf[2] = {cvt8<T>(s[0], s[1]), cvt8<T>(s[2], s[3])};
  mfma_rows<T>(V, pf, o, false);
}
template <bool DROP, typename T>
__device__ __forceinline__ void fwd_tile(const char* sK, const char* sV, int kt, bool last, int n, const r16x8 (&qf)[2], f32x4 (&o)[4],
                                         float& m, float& l, float scale_log2e, const DropCfg& drop, int bh, int qabs, int r, int g) {
  RowFrags F;
  f32x4 s[4];
  load_row_frags(sK, r, g, F);
  mfma_rows<T>(F, qf, s, true);
  fwd_softmax<DROP>(s, kt, last, n, o, m, l, scale_log2e, drop, bh, qabs, g);
  load_tr_frags(sV, r, g, F);
  fwd_pv<T>(F, s, o);
}

// merge of two online-softmax states of the same rows (key range split in two): shared by the streaming kernel (which keeps
// both states in one wave) and the resident kernel (partner waves), so the two stay bit-identical
__device__ __forceinline__ void softmax_merge(float& m, float& l, f32x4 (&o)[4], float m1, float l1, const f32x4 (&o1)[4]) {
  const float mn = fmaxf(m, m1);
  const float a0 = fast_exp2(m - mn), a1 = fast_exp2(m1 - mn);
  m = mn;
  l = __builtin_fmaf(l1, a1, l * a0);          // explicit fma shapes: identical rounding in every kernel that inlines this
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) o[t][j] = __builtin_fmaf(o1[t][j], a1, o[t][j] * a0);
}
// first key tile of the second half of the key range
__device__ __forceinline__ int attn_half_tiles(int nkt) { return (nkt + 1) >> 1; }

// One query row's 64 outputs of this lane (4 x 4 consecutive columns): bf16, or - o8 > 0, the fp8 inference path's out-projection
// operand - OCP e4m3 bytes of (value * o8) at the same ELEMENT offsets of a byte buffer.
template <typename T>
__device__ __forceinline__ void store_out_row(r16* out, long off, const f32x4 (&o)[4], float inv, float o8, int g) {
  if (o8 > 0.f) {
    unsigned char* row = reinterpret_cast<unsigned char*>(out) + off;
    const float sc = inv * o8;
#pragma unroll
    for (int t = 0; t < 4; ++t) *reinterpret_cast<unsigned*>(row + 16 * t + 4 * g) = pack_fp8x4(o[t] * sc);
  } else {
    r16* row = out + off;
#pragma unroll
    for (int t = 0; t < 4; ++t) *reinterpret_cast<r16x4*>(row + 16 * t + 4 * g) = cvt4<T>(o[t][0] * inv, o[t][1] * inv, o[t][2] * inv, o[t][3] * inv);
  }
}

template <bool DROP, typename T>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const r16* __restrict__ qkv, long ld, int n, int heads, float scale_log2e,
                                                       r16* __restrict__ out, long ldo, float* __restrict__ lse, DropCfg drop, float o8) {
  __shared__ __attribute__((aligned(16))) char smem[2 * IMG];
  char* sK = smem;
  char* sV = smem + IMG;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, r = lane & 15, g = lane >> 4;
  const Grid2 gb = grid2d_xcd((n + TQ - 1) / TQ);                  // 1-D launch: whole heads per XCD
  const int b = gb.by / heads, h = gb.by - b * heads, inner = heads * DH;
  const r16* Q = qkv + (long)b * n * ld + h * DH;
  const r16* K = Q + inner;
  const r16* V = Q + 2 * inner;
  const int q0 = gb.bx * TQ + 16 * wid;
  const int qrow = min(q0 + r, n - 1);

  r16x8 qf[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) qf[ks] = *reinterpret_cast<const r16x8*>(Q + (long)qrow * ld + 32 * ks + 8 * g);

  f32x4 o[4], o0[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) o[t] = o0[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, l = 0.f, m0 = -INFINITY, l0 = 0.f;

  const int nkt = (n + TK - 1) / TK;
  const int nh = attn_half_tiles(nkt);   // the key range is processed as two independent online-softmax states merged at the end,
                                         // exactly as the resident kernel's partner waves do (bit-identical results)
  uint4 rk[2], rv[2];
  tile_gload(K, ld, 0, n, tid, rk);
  tile_gload(V, ld, 0, n, tid, rv);
  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();
    tile_swrite(sK, tid, rk);
    tile_swrite(sV, tid, rv);
    __syncthreads();
    if (kt + 1 < nkt) {
      tile_gload(K, ld, (kt + 1) * TK, n, tid, rk);
      tile_gload(V, ld, (kt + 1) * TK, n, tid, rv);
    }
    if (kt == nh) {                      // second half starts: park the first state
      m0 = m; l0 = l; m = -INFINITY; l = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) { o0[t] = o[t]; o[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    }
    fwd_tile<DROP, T>(sK, sV, kt, kt == nkt - 1, n, qf, o, m, l, scale_log2e, drop, gb.by, q0 + r, r, g);
  }
  if (nkt <= nh) {                       // single tile: the (empty) second state is merged all the same
    m0 = m; l0 = l; m = -INFINITY; l = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) { o0[t] = o[t]; o[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  }
  softmax_merge(m0, l0, o0, m, l, o);
  m = m0; l = l0;
#pragma unroll
  for (int t = 0; t < 4; ++t) o[t] = o0[t];
  const float ltot = group_sum(l);
  const float inv = 1.0f / ltot;
  const int q = q0 + r;
  if (q < n) {
    store_out_row<T>(out, ((long)b * n + q) * ldo + h * DH, o, inv, o8, g);
    if (g == 0 && lse) lse[((long)b * heads + h) * n + q] = (m + log2f(ltot)) * 0.69314718055994530942f;
  }
}

// ================================================================================================ resident variants
// For n <= 576 (ViT3D-base: n = 513) the whole K and V of one (batch, head) - or Q and dO for the dK/dV pass - fit in
// the CU's LDS (2 x 9 tiles x 8 KiB = 144 KiB).  One 512-thread workgroup streams them in ONCE with LDS-DMA (buffer
// bounds zero-fill the rows beyond n), then its eight waves each own one 16-row group and run the same per-tile steps as
// the streaming kernels with no further global loads and no barriers: the streaming kernels expose one L2 round trip
// and two barriers per 64-key tile (9 of them at n = 513), which is where their time went.  Row groups are spread evenly
// over ceil(groups / 8) workgroups per (batch, head): n = 513 -> 33 groups -> 5 workgroups of 7,7,7,6,6.
// Results are bit-identical to the streaming kernels (same tile order, same arithmetic).
typedef __attribute__((address_space(3))) void lds_void_t;
constexpr int RES_THREADS = 512;
constexpr int RES_MAX_TILES = 9;

// (LDS-DMA helpers - uniform_rsrc / lds_dma, issued from inline asm - are in common.h)
// all eight waves: DMA rows 0 .. 64*nt-1 of X (row stride ld elements, 64 bf16 per row) into nt swizzled 8 KiB images
__device__ __forceinline__ void res_dma(const r16* X, long ld, int n, int nt, char* img, int wid, int lane) {
  const unsigned bytes = (unsigned)((((long)n - 1) * ld + DH) * 2);
  const dma_desc rs = uniform_rsrc(X, bytes);
  const int row = lane >> 3;                                          // 8 rows x 8 chunks per wave-instruction
  const int voff = (int)((((long)(wid * 8 + row)) * ld + ((((lane & 7) ^ (row & 7))) << 3)) * 2);   // img128_off inverse
  const int step = (int)(64 * ld * 2);
  for (int t = 0; t < nt; ++t)
    lds_dma<16>(rs, img + t * IMG + wid * 1024, voff, t * step);
}
// this wave's 16-row group, or -1
__device__ __forceinline__ int res_group(int n, int wid, int bx, int nblk) {
  const int groups = (n + 15) >> 4;
  const int g0 = (int)((long)bx * groups / nblk), g1 = (int)((long)(bx + 1) * groups / nblk);
  return (g0 + wid < g1) ? g0 + wid : -1;
}
// workgroups per (batch, head) of the resident kernels
__host__ __device__ __forceinline__ int attn_res_blocks(int n) { return (((n + 15) >> 4) + 7) / 8; }
// The resident kernels run on a 1-D grid of nblk x (batch x heads) workgroups: consecutive block ids are dealt round-robin over the
// eight XCDs, so a 2-D (nblk, batch x heads) grid put the nblk workgroups of ONE head on nblk different XCDs and every XCD's L2
// fetched that head's K / V (or Q / dO) images separately (rocprofv3 FETCH_SIZE: 35-42 MB per launch against 10-20 MB of operands).
// xcd_remap gives each XCD a contiguous run of logical ids, i.e. whole heads.
struct ResBlock { int bx, by, nblk; };
__device__ __forceinline__ ResBlock res_block(int n) {
  ResBlock rb;
  rb.nblk = attn_res_blocks(n);
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  rb.by = lid / rb.nblk;
  rb.bx = lid - rb.by * rb.nblk;
  return rb;
}

// SPLIT = 2: sixteen waves; waves w and w + 8 share a row group and take one half of the key tiles each (four waves per
// SIMD instead of two hide the softmax's dependent-instruction latency), then merge (m, l, o) through LDS.
template <int SPLIT, bool DROP, typename T>
__global__ __launch_bounds__(RES_THREADS * SPLIT) void attn_fwd_res_kernel(const r16* __restrict__ qkv, long ld, int n, int heads,
                                                                           float scale_log2e, r16* __restrict__ out, long ldo,
                                                                           float* __restrict__ lse, DropCfg drop, float o8) {
  extern __shared__ __attribute__((aligned(16))) char rsmem[];
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, g = lane >> 4;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int slot = wid & 7, half = wid >> 3;          // half is 0 when SPLIT == 1
  const ResBlock rb = res_block(n);
  const int b = rb.by / heads, h = rb.by - b * heads, inner = heads * DH;
  const r16* Q = qkv + (long)b * n * ld + h * DH;
  const int nkt = (n + TK - 1) / TK;
  char* sK = rsmem;
  char* sV = rsmem + nkt * IMG;
  const int grp = res_group(n, slot, rb.bx, rb.nblk);
  const int q0 = (grp < 0 ? 0 : grp) * 16;
  const int qrow = min(q0 + r, n - 1);
  r16x8 qf[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) qf[ks] = *reinterpret_cast<const r16x8*>(Q + (long)qrow * ld + 32 * ks + 8 * g);
  if (SPLIT == 1 || half == 0) res_dma(Q + inner, ld, n, nkt, sK, slot, lane);
  if (SPLIT == 1 || half == 1) res_dma(Q + 2 * inner, ld, n, nkt, sV, slot, lane);
  // (taking the tiles one by one as they land - a counted wait and a barrier per tile - measured 16.1 us against 14.5: the
  // per-tile barriers keep the two waves of a SIMD in step, and in step their MFMA and softmax phases collide instead of overlapping)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                   // the DMA of every wave has landed
  if (SPLIT == 1 && grp < 0) return;
  f32x4 o[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) o[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, l = 0.f;
  const int nh = attn_half_tiles(nkt);
  // both halves exist for every n (the streaming kernel uses the same split point), a one-wave block walks them in turn
  for (int part = 0; part < 2; ++part) {
    if (SPLIT == 2 && part != half) continue;
    const int k0 = part ? nh : 0, k1 = part ? nkt : nh;
    f32x4 o1[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) o1[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m1 = -INFINITY, l1 = 0.f;
    if (SPLIT == 2) {
      // four waves per SIMD hide the LDS latency by themselves; the plain tile step keeps the kernel within 128 VGPRs
      if (grp >= 0)
        for (int kt = k0; kt < k1; ++kt)
          fwd_tile<DROP, T>(sK + kt * IMG, sV + kt * IMG, kt, kt == nkt - 1, n, qf, o1, m1, l1, scale_log2e, drop, rb.by, q0 + r, r, g);
    } else if (grp >= 0 && k0 < k1) {
      // software pipeline: the V fragments of this tile and the K fragments of the next one are requested from LDS before the
      // softmax arithmetic, so the ds_read latency sits under ~130 VALU instructions instead of in front of the MFMAs
      RowFrags KF, VF;
      f32x4 sc[4];
      load_row_frags(sK + k0 * IMG, r, g, KF);
      for (int kt = k0; kt < k1; ++kt) {
        mfma_rows<T>(KF, qf, sc, true);
        __builtin_amdgcn_sched_barrier(0);
        load_tr_frags(sV + kt * IMG, r, g, VF);
        load_row_frags(sK + (kt + 1 < k1 ? kt + 1 : kt) * IMG, r, g, KF);
        __builtin_amdgcn_sched_barrier(0);
        fwd_softmax<DROP>(sc, kt, kt == nkt - 1, n, o1, m1, l1, scale_log2e, drop, rb.by, q0 + r, g);
        __builtin_amdgcn_sched_barrier(0);
        fwd_pv<T>(VF, sc, o1);
      }
    }
    if (SPLIT == 1) {
      if (part == 0) {
        m = m1; l = l1;
#pragma unroll
        for (int t = 0; t < 4; ++t) o[t] = o1[t];
      } else {
        softmax_merge(m, l, o, m1, l1, o1);
      }
    } else {
      // partner exchange through LDS (the K/V images are dead after the barrier): [slot][18][64 lanes] floats
      __syncthreads();
      float* xch = reinterpret_cast<float*>(rsmem) + slot * 18 * 64 + lane;
      if (half == 1) {
        xch[0] = m1; xch[64] = l1;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int j = 0; j < 4; ++j) xch[(2 + 4 * t + j) * 64] = o1[t][j];
      }
      __syncthreads();
      if (half == 1 || grp < 0) return;
      m = m1; l = l1;
#pragma unroll
      for (int t = 0; t < 4; ++t) o[t] = o1[t];
      float m2 = xch[0], l2 = xch[64];
      f32x4 o2[4];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) o2[t][j] = xch[(2 + 4 * t + j) * 64];
      softmax_merge(m, l, o, m2, l2, o2);
    }
  }
  const float ltot = group_sum(l);
  const float inv = 1.0f / ltot;
  const int q = q0 + r;
  if (q < n) {
    store_out_row<T>(out, ((long)b * n + q) * ldo + h * DH, o, inv, o8, g);
    if (g == 0 && lse) lse[((long)b * heads + h) * n + q] = (m + log2f(ltot)) * 0.69314718055994530942f;
  }
}

// ================================================================================================ wide streaming forward
// Long sequences / large batches (ViT3D-large: n = 4097; the 4D path: B*T = 20 volumes).  Against attn_fwd_kernel:
//   * every wave owns TWO 16-row groups (32 query rows): one set of K / V fragment reads from LDS (8 ds_read_b128 +
//     16 ds_read_b64_tr_b16 per 64-key tile) now feeds 32 MFMAs instead of 16 - the fragment reads were the streaming
//     kernel's largest per-tile cost after the softmax;
//   * K / V tiles arrive by LDS-DMA into a two-stage ring (no staging registers, no ds_write pass), ONE barrier per tile:
//     the DMA of tile kt+1 is in flight while tile kt is computed.
// Same tile order, same two-state online softmax, same per-group arithmetic as the other forward kernels: bit-identical.
constexpr int WIDE_ROWS = 128;     // query rows per workgroup (4 waves x 32)
#ifndef NV_WIDE_FWD_BLOCKS
#define NV_WIDE_FWD_BLOCKS 3       // workgroups per CU the register budget is set for.  3 = 168 VGPRs (10 spilled, 44 B of scratch) against 178 at 2:
                                   // same-box A/B, ViT3D-large forward 161.9 / 162.4 -> 167.3 / 167.5 volumes/s, base batch 64 6113 / 6119 -> 6181 / 6174
#endif
template <bool DROP, typename T>
__global__ __launch_bounds__(256, NV_WIDE_FWD_BLOCKS) void attn_fwd_wide_kernel(const r16* __restrict__ qkv, long ld, int n, int heads, float scale_log2e,
                                                               r16* __restrict__ out, long ldo, float* __restrict__ lse, DropCfg drop, float o8) {
  __shared__ __attribute__((aligned(16))) char wsmem[2 * 2 * IMG];      // [stage][K image | V image]
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, g = lane >> 4;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const Grid2 gb = grid2d_xcd((n + WIDE_ROWS - 1) / WIDE_ROWS);                  // 1-D launch: whole heads per XCD
  const int b = gb.by / heads, h = gb.by - b * heads, inner = heads * DH;
  const r16* Q = qkv + (long)b * n * ld + h * DH;
  const int q0 = gb.bx * WIDE_ROWS + 32 * wid;

  r16x8 qf[2][2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int qrow = min(q0 + 16 * u + r, n - 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf[u][ks] = *reinterpret_cast<const r16x8*>(Q + (long)qrow * ld + 32 * ks + 8 * g);
  }

  // LDS-DMA of one K and one V tile: 8 + 8 pieces of 1 KiB (8 rows x 128 B), pieces wid and wid + 4 of each per wave
  const unsigned bytes = (unsigned)((((long)n - 1) * ld + DH) * 2);
  const dma_desc rsK = uniform_rsrc((Q + inner), bytes);
  const dma_desc rsV = uniform_rsrc((Q + 2 * inner), bytes);
  int voff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (wid + 4 * i) * 8 + (lane >> 3);
    voff[i] = (int)(((long)row * ld + (((lane & 7) ^ (row & 7)) << 3)) * 2);            // img128_off inverse
  }
  const int step = (int)(64 * ld * 2);
  auto issue = [&](int t, char* stage) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      lds_dma<16>(rsK, stage + (wid + 4 * i) * 1024, voff[i], t * step);
      lds_dma<16>(rsV, stage + IMG + (wid + 4 * i) * 1024, voff[i], t * step);
    }
  };

  f32x4 o[2][4], o0[2][4];
  float m[2], l[2], m0[2], l0[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    m[u] = m0[u] = -INFINITY; l[u] = l0[u] = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) o[u][t] = o0[u][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int nkt = (n + TK - 1) / TK;
  const int nh = attn_half_tiles(nkt);
  issue(0, wsmem);
  // the Q fragments' loads retire HERE as far as the compiler's wait bookkeeping goes: left to their first use inside the loop it
  // would wait for them there, on every pass, with counts that ignore the DMA issued in between (vmcnt(3..0): a full drain)
  asm volatile("" ::"v"(qf[0][0]), "v"(qf[0][1]), "v"(qf[1][0]), "v"(qf[1][1]));
  // wave-uniform: does the wave's first / second 16-row group hold any query row?  (Rows beyond n are computed on clamped loads and never stored: skipping them
  // changes no result; at n = 513 the fifth workgroup of every (batch, head) holds ONE row in 128.)
  const bool act0 = q0 < n, act1 = q0 + 16 < n;
  auto tile = [&](int kt) {
    char* cur = wsmem + (kt & 1) * 2 * IMG;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of tile kt have landed
    __builtin_amdgcn_s_barrier();                         // ... everybody's; and every read of the other stage (tile kt-1) is done
    if (kt + 1 < nkt) issue(kt + 1, wsmem + ((kt + 1) & 1) * 2 * IMG);
    if (!act0) return;                                    // no query row of this wave exists (the last workgroup of n = 512 + 1: three of its four waves)
    RowFrags F;
    f32x4 s[2][4];
    load_row_frags(cur, r, g, F);
    mfma_rows<T>(F, qf[0], s[0], true);
    if (act1) mfma_rows<T>(F, qf[1], s[1], true);
    __builtin_amdgcn_sched_barrier(0);
    load_tr_frags(cur + IMG, r, g, F);                    // V fragments requested before the softmax arithmetic
    __builtin_amdgcn_sched_barrier(0);
    fwd_softmax<DROP>(s[0], kt, kt == nkt - 1, n, o[0], m[0], l[0], scale_log2e, drop, gb.by, q0 + r, g);
    if (act1) fwd_softmax<DROP>(s[1], kt, kt == nkt - 1, n, o[1], m[1], l[1], scale_log2e, drop, gb.by, q0 + 16 + r, g);
    __builtin_amdgcn_sched_barrier(0);
    fwd_pv<T>(F, s[0], o[0]);
    if (act1) fwd_pv<T>(F, s[1], o[1]);
  };
  // two loops, one per half of the key range (the split point every forward kernel uses), the first state parked in between:
  // as ONE loop with the parking under `if (kt == nh)` the compiler copied all 34 state registers on every pass
  const int nfirst = nh < nkt ? nh : nkt;
  for (int kt = 0; kt < nfirst; ++kt) tile(kt);
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    m0[u] = m[u]; l0[u] = l[u]; m[u] = -INFINITY; l[u] = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) { o0[u][t] = o[u][t]; o[u][t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  }
  for (int kt = nfirst; kt < nkt; ++kt) tile(kt);
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    softmax_merge(m0[u], l0[u], o0[u], m[u], l[u], o[u]);      // a single tile leaves the second state empty: merged all the same
    const float ltot = group_sum(l0[u]);
    const float inv = 1.0f / ltot;
    const int q = q0 + 16 * u + r;
    if (q < n) {
      store_out_row<T>(out, ((long)b * n + q) * ldo + h * DH, o0[u], inv, o8, g);
      if (g == 0 && lse) lse[((long)b * heads + h) * n + q] = (m0[u] + log2f(ltot)) * 0.69314718055994530942f;
    }
  }
}

static int g_attn_mode = 0;    // 0 = heuristic, 1 = streaming kernels, 2 = resident kernels (tests compare the two bit for bit)
static int g_attn_split = 1;   // resident forward: waves per row group.  Two (partner waves split the key range, four waves per
                               // SIMD) measured the same 13.5 us as one at n = 513: the kernel is VALU/MFMA-issue bound, not latency bound
static int g_attn_bwd_merged = 0;   // resident backward: mode + 100 = dQ and dK / dV as ONE launch (attn_bwd_res_kernel) instead of two.  Measured
                                    // SLOWER in the train step (same-box A/B, ViT3D-base batch 4: 36.3 us per layer against 16.3 + 15.5, step
                                    // 1122 -> 1088 volumes/s; profiles/r04_negative_results.log): 480 workgroups of 144 KiB on 256 CUs run as two
                                    // uneven rounds, which costs more than the dependent launch boundary and the 16 idle CUs it removes
extern "C" int nv_attn_set_mode(int mode) {
  g_attn_bwd_merged = (mode >= 100) ? 1 : 0;
  mode %= 100;
  g_attn_mode = mode % 10;
  g_attn_split = (mode / 10 == 2) ? 2 : 1;
  return 0;
}
static bool attn_resident(int n) {
  const int nt = (n + TK - 1) / TK;
  return g_attn_mode != 1 && nt <= RES_MAX_TILES;
}
template <typename Kern>
static void attn_res_attr(Kern kern, int lds) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
}

// One launch of kernel family K for the run-time (dropout, operand format) pair: K(D, T) names the instantiation (two per kernel and
// format: the dropout code under a runtime branch cost register copies - and, in one kernel, spills - on the path without it).
#define ATTN_LAUNCH(K, ...)                                                                                      \
  do {                                                                                                           \
    if (fp16) { if (dropping) hipLaunchKernelGGL((K(true, fp16_t)), __VA_ARGS__); else hipLaunchKernelGGL((K(false, fp16_t)), __VA_ARGS__); } \
    else { if (dropping) hipLaunchKernelGGL((K(true, bf16_t)), __VA_ARGS__); else hipLaunchKernelGGL((K(false, bf16_t)), __VA_ARGS__); }       \
  } while (0)
#define ATTN_ATTR(K, lds)                                                                                        \
  do { attn_res_attr(K(false, bf16_t), lds); attn_res_attr(K(true, bf16_t), lds); attn_res_attr(K(false, fp16_t), lds); attn_res_attr(K(true, fp16_t), lds); } while (0)
#define K_FWD_RES1(D, T) attn_fwd_res_kernel<1, D, T>
#define K_FWD_RES2(D, T) attn_fwd_res_kernel<2, D, T>
#define K_FWD_WIDE(D, T) attn_fwd_wide_kernel<D, T>
#define K_FWD(D, T) attn_fwd_kernel<D, T>
#define K_DQ_RES(D, T) attn_bwd_dq_res_kernel<D, T>
#define K_DKV_RES(D, T) attn_bwd_dkv_res_kernel<D, T>
#define K_BWD_RES(D, T) attn_bwd_res_kernel<D, T>
#define K_DQ_WIDE(D, T) attn_bwd_dq_wide_kernel<D, T>
#define K_DKV_WIDE(D, T) attn_bwd_dkv_wide_kernel<D, T>
#define K_DQ(D, T) attn_bwd_dq_kernel<D, T>
#define K_DKV(D, T) attn_bwd_dkv_kernel<D, T>

static int attn_fwd_impl(const void* qkv, long ld_qkv, int B, int n, int heads, int dim_head, float scale, void* out, long ld_out,
                         float* lse, unsigned long drop_seed, float drop_p, float o8, void* stream) {
  NV_CHECK_ARG(attn_generic_supported(dim_head), "nv_attn_fwd: dim_head=%d unsupported (multiples of 8 up to 128)", dim_head);
  NV_CHECK_ARG(B > 0 && n > 0 && heads > 0 && ld_qkv >= 3 * heads * dim_head && ld_out >= heads * dim_head && (ld_qkv % 8) == 0 && (ld_out % 4) == 0,
               "nv_attn_fwd: bad dims");
  NV_CHECK_ARG(nv_aligned16(qkv) && nv_aligned16(out), "nv_attn_fwd: alignment");
  if (dim_head != DH)      // the MFMA kernels below are built for the reference's default head dim (vit_3d.py:29); any other one: attention_generic.hip
    return launch_attn_generic_fwd(qkv, ld_qkv, B, n, heads, dim_head, scale, out, ld_out, lse, make_drop(drop_seed, drop_p), (hipStream_t)stream);
  const DropCfg drop = make_drop(drop_seed, drop_p);
  const bool dropping = drop.thresh != 0, fp16 = nv_operand_format() == NV_OPERAND_FP16;
  const float sl2 = scale * 1.44269504088896340736f;
  hipStream_t s = (hipStream_t)stream;
  const int slot = nv_prof_begin(3, 4.0 * B * heads * (double)n * n * DH, stream);
  // LDS-resident K / V (one 144 KiB workgroup per CU) pays when there are few row groups (ViT3D-base at batch 4: 240 workgroups);
  // with thousands of row groups the wide streaming kernel keeps several workgroups per CU and reads half the fragments
  const long wide_groups = (long)B * heads * ((n + WIDE_ROWS - 1) / WIDE_ROWS);
  const bool use_res = g_attn_mode != 3 && attn_resident(n) && (g_attn_mode == 2 || wide_groups < 768);
  if (use_res) {
    NV_CHECK_ARG((long)n * ld_qkv < (1L << 30), "nv_attn_fwd: operand too large for 32-bit buffer offsets");
    const int lds = 2 * ((n + TK - 1) / TK) * IMG;
    static bool attr = false;
    if (!attr) {
      ATTN_ATTR(K_FWD_RES1, 2 * RES_MAX_TILES * IMG);
      ATTN_ATTR(K_FWD_RES2, 2 * RES_MAX_TILES * IMG);
      attr = true;
    }
    const dim3 grid(attn_res_blocks(n) * B * heads);
    if (g_attn_split == 2) ATTN_LAUNCH(K_FWD_RES2, grid, dim3(2 * RES_THREADS), lds, s, (const r16*)qkv, ld_qkv, n, heads, sl2, (r16*)out, ld_out, lse, drop, o8);
    else ATTN_LAUNCH(K_FWD_RES1, grid, dim3(RES_THREADS), lds, s, (const r16*)qkv, ld_qkv, n, heads, sl2, (r16*)out, ld_out, lse, drop, o8);
  } else if (g_attn_mode != 1 && (long)n * ld_qkv < (1L << 30)) {
    ATTN_LAUNCH(K_FWD_WIDE, dim3(((n + WIDE_ROWS - 1) / WIDE_ROWS) * B * heads), dim3(256), 0, s, (const r16*)qkv, ld_qkv, n, heads, sl2, (r16*)out, ld_out, lse, drop, o8);
  } else {
    ATTN_LAUNCH(K_FWD, dim3(((n + TQ - 1) / TQ) * B * heads), dim3(256), 0, s, (const r16*)qkv, ld_qkv, n, heads, sl2, (r16*)out, ld_out, lse, drop, o8);
  }
  nv_prof_end(slot, stream);
  NV_CHECK_LAUNCH("nv_attn_fwd");
  return NV_OK;
}

extern "C" int nv_attn_fwd(const void* qkv, long ld_qkv, int B, int n, int heads, int dim_head, float scale, void* out, long ld_out,
                           float* lse, unsigned long drop_seed, float drop_p, void* stream) {
  return attn_fwd_impl(qkv, ld_qkv, B, n, heads, dim_head, scale, out, ld_out, lse, drop_seed, drop_p, 0.f, stream);
}

// fp8 inference path: the attention output as OCP e4m3 bytes of (value * out_scale) - the operand of the fp8 out-projection;
// out is a byte buffer [B * n, ld_out] (ld_out in elements = bytes); dim_head 64, no dropout.
extern "C" int nv_attn_fwd_o8(const void* qkv, long ld_qkv, int B, int n, int heads, int dim_head, float scale, void* out, long ld_out,
                              float out_scale, void* stream) {
  NV_CHECK_ARG(dim_head == DH && out_scale > 0.f && (ld_out % 16) == 0, "nv_attn_fwd_o8: dim_head must be %d, out_scale > 0, ld_out a multiple of 16", DH);
  return attn_fwd_impl(qkv, ld_qkv, B, n, heads, dim_head, scale, out, ld_out, nullptr, 0, 0.f, out_scale, stream);
}

// ------------------------------------------------------------------------------------------------ backward: dQ (+ delta)
// one 64-key tile of the dQ pass for one wave (16 query rows), in pieces (see fwd_tile)
template <bool DROP>
__device__ __forceinline__ void dq_softmax_grad(const f32x4 (&sc)[4], f32x4 (&dp)[4], f32x4 (&ds)[4], int kt, int n, float dl, float lse2,
                                                float scale_log2e, const DropCfg& drop, int bh, int qabs, int g) {
    const f32x2 c2 = {scale_log2e, scale_log2e}, l2 = {-lse2, -lse2}, d2 = {-dl, -dl};
    const bool ragged = kt * TK + TK > n;         // only the last key tile can hold out-of-range (zero-filled) keys
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if constexpr (DROP) {
        dp[t] *= drop_factor4(drop, (((unsigned long long)bh * n + qabs) * ((n + 3) & ~3)) + (kt * TK + 16 * t + 4 * g));
      }
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        f32x2 x = {sc[t][2 * hh], sc[t][2 * hh + 1]};
        x = __builtin_elementwise_fma(x, c2, l2);
        f32x2 pv = {fast_exp2(x[0]), fast_exp2(x[1])};
        const f32x2 dpv = {dp[t][2 * hh], dp[t][2 * hh + 1]};
        const f32x2 dsv = pv * (dpv + d2);
        ds[t][2 * hh] = dsv[0];
        ds[t][2 * hh + 1] = dsv[1];
      }
      if (ragged) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (kt * TK + 16 * t + 4 * g + j >= n) ds[t][j] = 0.f;
      }
    }
}
template <bool DROP, typename T>
__device__ __forceinline__ void dq_tile(const char* sK, const char* sV, int kt, int n, const r16x8 (&qf)[2], const r16x8 (&dof)[2],
                                        f32x4 (&dq)[4], float dl, float lse2, float scale_log2e, const DropCfg& drop, int bh, int qabs,
                                        int r, int g) {
  RowFrags F;
  f32x4 sc[4], dp[4], ds[4];
  load_row_frags(sK, r, g, F);
  mfma_rows<T>(F, qf, sc, true);
  load_row_frags(sV, r, g, F);
  mfma_rows<T>(F, dof, dp, true);
  dq_softmax_grad<DROP>(sc, dp, ds, kt, n, dl, lse2, scale_log2e, drop, bh, qabs, g);
  load_tr_frags(sK, r, g, F);
  const r16x8 dsf[2] = {cvt8<T>(ds[0], ds[1]), cvt8<T>(ds[2], ds[3])};
  mfma_rows<T>(F, dsf, dq, false);
}

template <bool DROP, typename T>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const r16* __restrict__ qkv, long ld, const r16* __restrict__ out,
                                                          const r16* __restrict__ dout, long ldo, const float* __restrict__ lse, int n,
                                                          int heads, float scale, float* __restrict__ delta, r16* __restrict__ dqkv,
                                                          long ldd, DropCfg drop) {
  __shared__ __attribute__((aligned(16))) char smem[2 * IMG];
  char* sK = smem;
  char* sV = smem + IMG;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, r = lane & 15, g = lane >> 4;
  const Grid2 gb = grid2d_xcd((n + TQ - 1) / TQ);                  // 1-D launch: whole heads per XCD
  const int b = gb.by / heads, h = gb.by - b * heads, inner = heads * DH;
  const r16* Q = qkv + (long)b * n * ld + h * DH;
  const r16* K = Q + inner;
  const r16* V = Q + 2 * inner;
  const int q0 = gb.bx * TQ + 16 * wid;
  const int qrow = min(q0 + r, n - 1);
  const float scale_log2e = scale * 1.44269504088896340736f;

  r16x8 qf[2], dof[2];
  float dl = 0.f;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    qf[ks] = *reinterpret_cast<const r16x8*>(Q + (long)qrow * ld + 32 * ks + 8 * g);
    const long off = ((long)b * n + qrow) * ldo + h * DH + 32 * ks + 8 * g;
    dof[ks] = *reinterpret_cast<const r16x8*>(dout + off);
    const r16x8 of = *reinterpret_cast<const r16x8*>(out + off);
#pragma unroll
    for (int j = 0; j < 8; ++j) dl += dec1<T>(dof[ks][j]) * dec1<T>(of[j]);
  }
  dl = group_sum(dl);
  const float lse2 = lse[((long)b * heads + h) * n + qrow] * 1.44269504088896340736f;
  if (g == 0 && q0 + r < n) delta[((long)b * heads + h) * n + q0 + r] = dl;

  f32x4 dq[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) dq[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nkt = (n + TK - 1) / TK;
  uint4 rk[2], rv[2];
  tile_gload(K, ld, 0, n, tid, rk);
  tile_gload(V, ld, 0, n, tid, rv);
  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();
    tile_swrite(sK, tid, rk);
    tile_swrite(sV, tid, rv);
    __syncthreads();
    if (kt + 1 < nkt) {
      tile_gload(K, ld, (kt + 1) * TK, n, tid, rk);
      tile_gload(V, ld, (kt + 1) * TK, n, tid, rv);
    }
    dq_tile<DROP, T>(sK, sV, kt, n, qf, dof, dq, dl, lse2, scale_log2e, drop, gb.by, q0 + r, r, g);
  }
  const int q = q0 + r;
  if (q < n) {
    r16* drow = dqkv + ((long)b * n + q) * ldd + h * DH;
#pragma unroll
    for (int t = 0; t < 4; ++t)
      *reinterpret_cast<r16x4*>(drow + 16 * t + 4 * g) = cvt4<T>(dq[t][0] * scale, dq[t][1] * scale, dq[t][2] * scale, dq[t][3] * scale);
  }
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV
// Dropout factors of THIS lane's key for the four query rows q0 .. q0 + 3 (the dK/dV pass holds S^T: a lane's four values are four QUERY rows of one
// key, while the mask is hashed in groups of four consecutive KEYS of one query row).  The four lanes of a DPP quad hold the keys 4a .. 4a + 3
// (key0 is a multiple of 16, r = lane & 15): one hash group per query row - so lane i of the quad hashes row q0 + i only, and the quad exchanges the
// 16-bit fields with quad broadcasts: one 64-bit hash (three 64-bit multiplies) + 8 DPP moves per four values instead of four hashes.
// (ViT3D-base batch 4, dropout 0.1: the dK/dV kernel 29.7 us with four hashes per lane against 15.0 us without dropout.)
template <int J>
__device__ __forceinline__ unsigned quad_bcast(unsigned v) {
  return (unsigned)__builtin_amdgcn_mov_dpp((int)v, J * 0x55, 0xF, 0xF, true);
}
__device__ __forceinline__ f32x4 drop_factor_rows4(const DropCfg& d, unsigned long long row0_idx, unsigned long long row_stride) {
  const unsigned c = (unsigned)(row0_idx & 3);              // this lane's key within its group of four = its place in the quad
  const uint64_t h = nv_hash64(d.seed, (row0_idx + c * row_stride) >> 2);
  const unsigned lo = (unsigned)h, hi = (unsigned)(h >> 32);
  const unsigned l0 = quad_bcast<0>(lo), l1 = quad_bcast<1>(lo), l2 = quad_bcast<2>(lo), l3 = quad_bcast<3>(lo);
  const unsigned h0 = quad_bcast<0>(hi), h1 = quad_bcast<1>(hi), h2 = quad_bcast<2>(hi), h3 = quad_bcast<3>(hi);
  const unsigned w[4] = {(c & 2) ? h0 : l0, (c & 2) ? h1 : l1, (c & 2) ? h2 : l2, (c & 2) ? h3 : l3};
  f32x4 f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned field = (c & 1) ? (w[j] >> 16) : (w[j] & 0xFFFFu);
    f[j] = (field >= d.thresh) ? d.scale : 0.f;
  }
  return f;
}

// one 64-query tile of the dK/dV pass for one wave (16 keys); sL / sDl hold this tile's 64 log2-domain lse and delta values
template <bool DROP>
__device__ __forceinline__ void dkv_softmax_grad(const f32x4 (&sc)[4], const f32x4 (&dp)[4], f32x4 (&p)[4], f32x4 (&ds)[4], const float* sL,
                                                 const float* sDl, int qt, int n, float scale_log2e, const DropCfg& drop, int bh, int keyabs,
                                                 int g) {
    const f32x2 c2 = {scale_log2e, scale_log2e};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const f32x4 l4 = *reinterpret_cast<const f32x4*>(sL + 16 * t + 4 * g);
      const f32x4 d4 = *reinterpret_cast<const f32x4*>(sDl + 16 * t + 4 * g);
      f32x4 f = {1.f, 1.f, 1.f, 1.f};
      if constexpr (DROP)
        f = drop_factor_rows4(drop, (((unsigned long long)bh * n + (qt * TQ + 16 * t + 4 * g)) * ((n + 3) & ~3)) + keyabs, (unsigned long long)((n + 3) & ~3));
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        f32x2 x = {sc[t][2 * hh], sc[t][2 * hh + 1]};
        const f32x2 lv = {-l4[2 * hh], -l4[2 * hh + 1]}, dlv = {d4[2 * hh], d4[2 * hh + 1]};
        const f32x2 fv = {f[2 * hh], f[2 * hh + 1]}, dpv = {dp[t][2 * hh], dp[t][2 * hh + 1]};
        x = __builtin_elementwise_fma(x, c2, lv);
        const f32x2 pv = {fast_exp2(x[0]), fast_exp2(x[1])};
        const f32x2 pd = pv * fv;
        const f32x2 dsv = pv * (dpv * fv - dlv);
        p[t][2 * hh] = pd[0]; p[t][2 * hh + 1] = pd[1];
        ds[t][2 * hh] = dsv[0]; ds[t][2 * hh + 1] = dsv[1];
      }
    }
}
template <bool DROP, typename T>
__device__ __forceinline__ void dkv_tile(const char* sQ, const char* sD, const float* sL, const float* sDl, int qt, int n,
                                         const r16x8 (&kf)[2], const r16x8 (&vf)[2], f32x4 (&dk)[4], f32x4 (&dv)[4], float scale_log2e,
                                         const DropCfg& drop, int bh, int keyabs, int r, int g) {
  RowFrags F, G;
  f32x4 sc[4], dp[4], p[4], ds[4];
  load_row_frags(sQ, r, g, F);
  load_row_frags(sD, r, g, G);
  mfma_rows<T>(F, kf, sc, true);
  mfma_rows<T>(G, vf, dp, true);
  dkv_softmax_grad<DROP>(sc, dp, p, ds, sL, sDl, qt, n, scale_log2e, drop, bh, keyabs, g);
  load_tr_frags(sD, r, g, G);
  load_tr_frags(sQ, r, g, F);
  const r16x8 pf[2] = {cvt8<T>(p[0], p[1]), cvt8<T>(p[2], p[3])};
  const r16x8 dsf[2] = {cvt8<T>(ds[0], ds[1]), cvt8<T>(ds[2], ds[3])};
  mfma_rows<T>(G, pf, dv, false);
  mfma_rows<T>(F, dsf, dk, false);
}

template <bool DROP, typename T>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const r16* __restrict__ qkv, long ld, const r16* __restrict__ dout, long ldo,
                                                           const float* __restrict__ lse, const float* __restrict__ delta, int n,
                                                           int heads, float scale, r16* __restrict__ dqkv, long ldd, DropCfg drop) {
  __shared__ __attribute__((aligned(16))) char smem[2 * IMG + 2 * TQ * 4];
  char* sQ = smem;
  char* sD = smem + IMG;
  float* sL = reinterpret_cast<float*>(smem + 2 * IMG);
  float* sDl = sL + TQ;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, r = lane & 15, g = lane >> 4;
  const Grid2 gb = grid2d_xcd((n + TK - 1) / TK);                  // 1-D launch: whole heads per XCD
  const int b = gb.by / heads, h = gb.by - b * heads, inner = heads * DH;
  const r16* Q = qkv + (long)b * n * ld + h * DH;
  const r16* K = Q + inner;
  const r16* V = Q + 2 * inner;
  const r16* dO = dout + (long)b * n * ldo + h * DH;
  const float* L = lse + ((long)b * heads + h) * n;
  const float* Dl = delta + ((long)b * heads + h) * n;
  const int key0 = gb.bx * TK + 16 * wid;
  const int krow = min(key0 + r, n - 1);
  const float scale_log2e = scale * 1.44269504088896340736f;

  r16x8 kf[2], vf[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    kf[ks] = *reinterpret_cast<const r16x8*>(K + (long)krow * ld + 32 * ks + 8 * g);
    vf[ks] = *reinterpret_cast<const r16x8*>(V + (long)krow * ld + 32 * ks + 8 * g);
  }
  f32x4 dk[4], dv[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) dk[t] = dv[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nqt = (n + TQ - 1) / TQ;
  uint4 rq[2], rd[2];
  float rl = 0.f, rdl = 0.f;
  auto stats_load = [&](int qt) {
    if (tid < TQ) {
      const int q = qt * TQ + tid;
      rl = (q < n) ? L[q] * 1.44269504088896340736f : INFINITY;   // exp2(x - inf) = 0 masks padded query rows
      rdl = (q < n) ? Dl[q] : 0.f;
    }
  };
  tile_gload(Q, ld, 0, n, tid, rq);
  tile_gload(dO, ldo, 0, n, tid, rd);
  stats_load(0);
  for (int qt = 0; qt < nqt; ++qt) {
    __syncthreads();
    tile_swrite(sQ, tid, rq);
    tile_swrite(sD, tid, rd);
    if (tid < TQ) { sL[tid] = rl; sDl[tid] = rdl; }
    __syncthreads();
    if (qt + 1 < nqt) {
      tile_gload(Q, ld, (qt + 1) * TQ, n, tid, rq);
      tile_gload(dO, ldo, (qt + 1) * TQ, n, tid, rd);
      stats_load(qt + 1);
    }
    dkv_tile<DROP, T>(sQ, sD, sL, sDl, qt, n, kf, vf, dk, dv, scale_log2e, drop, gb.by, key0 + r, r, g);
  }
  const int key = key0 + r;
  if (key < n) {
    r16* drow = dqkv + ((long)b * n + key) * ldd + h * DH;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      *reinterpret_cast<r16x4*>(drow + inner + 16 * t + 4 * g) = cvt4<T>(dk[t][0] * scale, dk[t][1] * scale, dk[t][2] * scale, dk[t][3] * scale);
      *reinterpret_cast<r16x4*>(drow + 2 * inner + 16 * t + 4 * g) = cvt4<T>(dv[t][0], dv[t][1], dv[t][2], dv[t][3]);
    }
  }
}

// ------------------------------------------------------------------------------------------------ resident backward kernels
// delta of one query row (16 lanes r, 4 lane groups g): rowsum(dO . O) over the head's 64 columns.  ONE function for every place that
// needs it, so the value is the same to the last bit wherever it is computed (the dQ pass, or - merged launch - the dK/dV pass itself).
template <typename T>
__device__ __forceinline__ float row_delta(const r16* __restrict__ dout, const r16* __restrict__ out, long off, int g, r16x8 (&dof)[2]) {
  float dl = 0.f;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    dof[ks] = *reinterpret_cast<const r16x8*>(dout + off + 32 * ks + 8 * g);
    const r16x8 of = *reinterpret_cast<const r16x8*>(out + off + 32 * ks + 8 * g);
#pragma unroll
    for (int j = 0; j < 8; ++j) dl += dec1<T>(dof[ks][j]) * dec1<T>(of[j]);
  }
  return group_sum(dl);
}

// body of the dQ pass for workgroup (bx of nblk) of (batch, head) pair `by`
template <bool DROP, typename T>
__device__ __forceinline__ void attn_bwd_dq_res_body(char* rsmem, int by, int bx, int nblk, const r16* __restrict__ qkv, long ld,
                                                     const r16* __restrict__ out, const r16* __restrict__ dout, long ldo,
                                                     const float* __restrict__ lse, int n, int heads, float scale, float* __restrict__ delta,
                                                     r16* __restrict__ dqkv, long ldd, const DropCfg& drop) {
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, g = lane >> 4;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = by / heads, h = by - b * heads, inner = heads * DH;
  const r16* Q = qkv + (long)b * n * ld + h * DH;
  const int nkt = (n + TK - 1) / TK;
  char* sK = rsmem;
  char* sV = rsmem + nkt * IMG;
  res_dma(Q + inner, ld, n, nkt, sK, wid, lane);
  res_dma(Q + 2 * inner, ld, n, nkt, sV, wid, lane);
  const int grp = res_group(n, wid, bx, nblk);
  const int q0 = (grp < 0 ? 0 : grp) * 16;
  const int qrow = min(q0 + r, n - 1);
  const float scale_log2e = scale * 1.44269504088896340736f;

  r16x8 qf[2], dof[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) qf[ks] = *reinterpret_cast<const r16x8*>(Q + (long)qrow * ld + 32 * ks + 8 * g);
  const float dl = row_delta<T>(dout, out, ((long)b * n + qrow) * ldo + h * DH, g, dof);
  const float lse2 = lse[((long)b * heads + h) * n + qrow] * 1.44269504088896340736f;
  __syncthreads();                                   // drains the DMA (vmcnt(0)) of every wave
  if (grp < 0) return;
  if (g == 0 && q0 + r < n) delta[((long)b * heads + h) * n + q0 + r] = dl;

  f32x4 dq[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) dq[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  // (software-pipelining the fragment reads as in the forward kernel measured slower inside the train step: 901 vs 923 volumes/s)
  for (int kt = 0; kt < nkt; ++kt)
    dq_tile<DROP, T>(sK + kt * IMG, sV + kt * IMG, kt, n, qf, dof, dq, dl, lse2, scale_log2e, drop, by, q0 + r, r, g);
  const int q = q0 + r;
  if (q < n) {
    r16* drow = dqkv + ((long)b * n + q) * ldd + h * DH;
#pragma unroll
    for (int t = 0; t < 4; ++t)
      *reinterpret_cast<r16x4*>(drow + 16 * t + 4 * g) = cvt4<T>(dq[t][0] * scale, dq[t][1] * scale, dq[t][2] * scale, dq[t][3] * scale);
  }
}

// body of the dK / dV pass.  OWN_DELTA: delta = rowsum(dO . O) of every query row of the head is computed HERE (from `out`), with
// row_delta - the dQ pass's own function - instead of being read from what a preceding dQ launch wrote: that dependency was the only
// reason for two launches.  48 KiB of dO / O re-read per workgroup out of L2, under the wait for the resident tiles.
template <bool DROP, bool OWN_DELTA, typename T>
__device__ __forceinline__ void attn_bwd_dkv_res_body(char* rsmem, int by, int bx, int nblk, const r16* __restrict__ qkv, long ld,
                                                      const r16* __restrict__ out, const r16* __restrict__ dout, long ldo,
                                                      const float* __restrict__ lse, const float* __restrict__ delta, int n, int heads,
                                                      float scale, r16* __restrict__ dqkv, long ldd, const DropCfg& drop) {
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, g = lane >> 4;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = by / heads, h = by - b * heads, inner = heads * DH;
  const r16* Q = qkv + (long)b * n * ld + h * DH;
  const r16* K = Q + inner;
  const r16* V = Q + 2 * inner;
  const r16* dO = dout + (long)b * n * ldo + h * DH;
  const float* L = lse + ((long)b * heads + h) * n;
  const int nqt = (n + TQ - 1) / TQ;
  char* sQ = rsmem;
  char* sD = rsmem + nqt * IMG;
  float* sL = reinterpret_cast<float*>(rsmem + 2 * nqt * IMG);
  float* sDl = sL + nqt * TQ;
  res_dma(Q, ld, n, nqt, sQ, wid, lane);
  res_dma(dO, ldo, n, nqt, sD, wid, lane);
  if constexpr (OWN_DELTA) {
    for (int q = tid; q < nqt * TQ; q += RES_THREADS) sL[q] = (q < n) ? L[q] * 1.44269504088896340736f : INFINITY;   // exp2(x - inf) = 0 masks padded query rows
    const r16* O = out + (long)b * n * ldo + h * DH;
    for (int q0 = 16 * wid; q0 < nqt * TQ; q0 += 16 * (RES_THREADS / 64)) {
      r16x8 scratch[2];
      const int q = q0 + r;
      const float dl = row_delta<T>(dO, O, (long)min(q, n - 1) * ldo, g, scratch);
      if (g == 0) sDl[q] = (q < n) ? dl : 0.f;
    }
  } else {
    const float* Dl = delta + ((long)b * heads + h) * n;
    for (int q = tid; q < nqt * TQ; q += RES_THREADS) {
      sL[q] = (q < n) ? L[q] * 1.44269504088896340736f : INFINITY;
      sDl[q] = (q < n) ? Dl[q] : 0.f;
    }
  }
  const int grp = res_group(n, wid, bx, nblk);
  const int key0 = (grp < 0 ? 0 : grp) * 16;
  const int krow = min(key0 + r, n - 1);
  const float scale_log2e = scale * 1.44269504088896340736f;
  r16x8 kf[2], vf[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    kf[ks] = *reinterpret_cast<const r16x8*>(K + (long)krow * ld + 32 * ks + 8 * g);
    vf[ks] = *reinterpret_cast<const r16x8*>(V + (long)krow * ld + 32 * ks + 8 * g);
  }
  __syncthreads();                                   // DMA drained, sL / sDl written
  if (grp < 0) return;

  f32x4 dk[4], dv[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) dk[t] = dv[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int qt = 0; qt < nqt; ++qt)
    dkv_tile<DROP, T>(sQ + qt * IMG, sD + qt * IMG, sL + qt * TQ, sDl + qt * TQ, qt, n, kf, vf, dk, dv, scale_log2e, drop, by, key0 + r, r, g);
  const int key = key0 + r;
  if (key < n) {
    r16* drow = dqkv + ((long)b * n + key) * ldd + h * DH;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      *reinterpret_cast<r16x4*>(drow + inner + 16 * t + 4 * g) = cvt4<T>(dk[t][0] * scale, dk[t][1] * scale, dk[t][2] * scale, dk[t][3] * scale);
      *reinterpret_cast<r16x4*>(drow + 2 * inner + 16 * t + 4 * g) = cvt4<T>(dv[t][0], dv[t][1], dv[t][2], dv[t][3]);
    }
  }
}

template <bool DROP, typename T>
__global__ __launch_bounds__(RES_THREADS) void attn_bwd_dq_res_kernel(const r16* __restrict__ qkv, long ld, const r16* __restrict__ out,
                                                                      const r16* __restrict__ dout, long ldo, const float* __restrict__ lse,
                                                                      int n, int heads, float scale, float* __restrict__ delta,
                                                                      r16* __restrict__ dqkv, long ldd, DropCfg drop) {
  extern __shared__ __attribute__((aligned(16))) char rsmem[];
  const ResBlock rb = res_block(n);
  attn_bwd_dq_res_body<DROP, T>(rsmem, rb.by, rb.bx, rb.nblk, qkv, ld, out, dout, ldo, lse, n, heads, scale, delta, dqkv, ldd, drop);
}

template <bool DROP, typename T>
__global__ __launch_bounds__(RES_THREADS) void attn_bwd_dkv_res_kernel(const r16* __restrict__ qkv, long ld, const r16* __restrict__ dout,
                                                                       long ldo, const float* __restrict__ lse, const float* __restrict__ delta,
                                                                       int n, int heads, float scale, r16* __restrict__ dqkv, long ldd,
                                                                       DropCfg drop) {
  extern __shared__ __attribute__((aligned(16))) char rsmem[];
  const ResBlock rb = res_block(n);
  attn_bwd_dkv_res_body<DROP, false, T>(rsmem, rb.by, rb.bx, rb.nblk, qkv, ld, nullptr, dout, ldo, lse, delta, n, heads, scale, dqkv, ldd, drop);
}

// dQ and dK / dV of the resident form as ONE grid of 2 x nblk workgroups per (batch, head): logical ids [0, nblk)
// of a head run the dK / dV pass (computing delta themselves), [nblk, 2 nblk) the dQ pass; the ten workgroups of a head are
// consecutive logical ids, i.e. one XCD (xcd_remap).  Results are those of the two launches bit for bit (same bodies, same delta
// arithmetic: tests/test_kernels_gpu.py).  NOT the default: ViT3D-base at batch 4 is 480 workgroups of 144 KiB on 256 CUs - two
// uneven rounds - and measured 36.3 us per layer against 16.3 + 15.5 for the two launches (nv_attn_set_mode(+100) selects it).
template <bool DROP, typename T>
__global__ __launch_bounds__(RES_THREADS) void attn_bwd_res_kernel(const r16* __restrict__ qkv, long ld, const r16* __restrict__ out,
                                                                   const r16* __restrict__ dout, long ldo, const float* __restrict__ lse,
                                                                   int n, int heads, float scale, float* __restrict__ delta,
                                                                   r16* __restrict__ dqkv, long ldd, DropCfg drop) {
  extern __shared__ __attribute__((aligned(16))) char rsmem[];
  const int nblk = attn_res_blocks(n);
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int by = lid / (2 * nblk), rem = lid - by * 2 * nblk;
  if (rem < nblk) attn_bwd_dkv_res_body<DROP, true, T>(rsmem, by, rem, nblk, qkv, ld, out, dout, ldo, lse, nullptr, n, heads, scale, dqkv, ldd, drop);
  else attn_bwd_dq_res_body<DROP, T>(rsmem, by, rem - nblk, nblk, qkv, ld, out, dout, ldo, lse, n, heads, scale, delta, dqkv, ldd, drop);
}

// ------------------------------------------------------------------------------------------------ wide streaming backward
// Long sequences (ViT3D-large, n = 4097): the same restructuring as attn_fwd_wide_kernel.  Every wave owns two 16-row groups
// (query rows in the dQ pass, keys in the dK / dV pass), so one set of LDS fragment reads feeds twice the MFMAs; the streamed
// operand tiles arrive by LDS-DMA into a two-stage ring with one barrier per tile.  Per-group arithmetic is that of the
// streaming kernels (shared __device__ pieces).
template <bool DROP, typename T>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_wide_kernel(const r16* __restrict__ qkv, long ld, const r16* __restrict__ out,
                                                                  const r16* __restrict__ dout, long ldo, const float* __restrict__ lse, int n,
                                                                  int heads, float scale, float* __restrict__ delta, r16* __restrict__ dqkv,
                                                                  long ldd, DropCfg drop) {
  __shared__ __attribute__((aligned(16))) char wsmem[2 * 2 * IMG];      // [stage][K image | V image]
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, g = lane >> 4;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const Grid2 gb = grid2d_xcd((n + WIDE_ROWS - 1) / WIDE_ROWS);                  // 1-D launch: whole heads per XCD
  const int b = gb.by / heads, h = gb.by - b * heads, inner = heads * DH;
  const r16* Q = qkv + (long)b * n * ld + h * DH;
  const int q0 = gb.bx * WIDE_ROWS + 32 * wid;
  const float scale_log2e = scale * 1.44269504088896340736f;

  r16x8 qf[2][2], dof[2][2];
  float dl[2], lse2[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int qrow = min(q0 + 16 * u + r, n - 1);
    float d = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      qf[u][ks] = *reinterpret_cast<const r16x8*>(Q + (long)qrow * ld + 32 * ks + 8 * g);
      const long off = ((long)b * n + qrow) * ldo + h * DH + 32 * ks + 8 * g;
      dof[u][ks] = *reinterpret_cast<const r16x8*>(dout + off);
      const r16x8 of = *reinterpret_cast<const r16x8*>(out + off);
#pragma unroll
      for (int j = 0; j < 8; ++j) d += dec1<T>(dof[u][ks][j]) * dec1<T>(of[j]);
    }
    dl[u] = group_sum(d);
    lse2[u] = lse[((long)b * heads + h) * n + qrow] * 1.44269504088896340736f;
    if (g == 0 && q0 + 16 * u + r < n) delta[((long)b * heads + h) * n + q0 + 16 * u + r] = dl[u];
  }

  const unsigned bytes = (unsigned)((((long)n - 1) * ld + DH) * 2);
  const dma_desc rsK = uniform_rsrc((Q + inner), bytes);
  const dma_desc rsV = uniform_rsrc((Q + 2 * inner), bytes);
  int voff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (wid + 4 * i) * 8 + (lane >> 3);
    voff[i] = (int)(((long)row * ld + (((lane & 7) ^ (row & 7)) << 3)) * 2);            // img128_off inverse
  }
  const int step = (int)(64 * ld * 2);
  auto issue = [&](int t, char* stage) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      lds_dma<16>(rsK, stage + (wid + 4 * i) * 1024, voff[i], t * step);
      lds_dma<16>(rsV, stage + IMG + (wid + 4 * i) * 1024, voff[i], t * step);
    }
  };

  f32x4 dq[2][4];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int t = 0; t < 4; ++t) dq[u][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nkt = (n + TK - 1) / TK;
  issue(0, wsmem);
  for (int kt = 0; kt < nkt; ++kt) {
    const char* cur = wsmem + (kt & 1) * 2 * IMG;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 1 < nkt) issue(kt + 1, wsmem + ((kt + 1) & 1) * 2 * IMG);
    RowFrags F;
    f32x4 sc[2][4], dp[2][4], ds[2][4];
    load_row_frags(cur, r, g, F);
    mfma_rows<T>(F, qf[0], sc[0], true);
    mfma_rows<T>(F, qf[1], sc[1], true);
    load_row_frags(cur + IMG, r, g, F);
    mfma_rows<T>(F, dof[0], dp[0], true);
    mfma_rows<T>(F, dof[1], dp[1], true);
    dq_softmax_grad<DROP>(sc[0], dp[0], ds[0], kt, n, dl[0], lse2[0], scale_log2e, drop, gb.by, q0 + r, g);
    dq_softmax_grad<DROP>(sc[1], dp[1], ds[1], kt, n, dl[1], lse2[1], scale_log2e, drop, gb.by, q0 + 16 + r, g);
    load_tr_frags(cur, r, g, F);                          // (after the exponentials: requesting them earlier spills registers)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const r16x8 dsf[2] = {cvt8<T>(ds[u][0], ds[u][1]), cvt8<T>(ds[u][2], ds[u][3])};
      mfma_rows<T>(F, dsf, dq[u], false);
    }
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int q = q0 + 16 * u + r;
    if (q < n) {
      r16* drow = dqkv + ((long)b * n + q) * ldd + h * DH;
#pragma unroll
      for (int t = 0; t < 4; ++t)
        *reinterpret_cast<r16x4*>(drow + 16 * t + 4 * g) = cvt4<T>(dq[u][t][0] * scale, dq[u][t][1] * scale, dq[u][t][2] * scale, dq[u][t][3] * scale);
    }
  }
}

// in-place form of dkv_softmax_grad for the wide kernel (p overwrites sc, ds overwrites dp: identical arithmetic, fewer live
// registers); sLraw holds the raw natural-log lse of this query tile (converted to the log2 domain here: the same one rounding)
template <bool DROP>
__device__ __forceinline__ void dkv_softmax_grad_inplace(f32x4 (&sc)[4], f32x4 (&dp)[4], const float* sLraw, const float* sDl, int qt, int n,
                                                         float scale_log2e, const DropCfg& drop, int bh, int keyabs, int g) {
    const f32x2 c2 = {scale_log2e, scale_log2e};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const f32x4 l4 = *reinterpret_cast<const f32x4*>(sLraw + 16 * t + 4 * g) * 1.44269504088896340736f;
      const f32x4 d4 = *reinterpret_cast<const f32x4*>(sDl + 16 * t + 4 * g);
      f32x4 f = {1.f, 1.f, 1.f, 1.f};
      if constexpr (DROP)
        f = drop_factor_rows4(drop, (((unsigned long long)bh * n + (qt * TQ + 16 * t + 4 * g)) * ((n + 3) & ~3)) + keyabs, (unsigned long long)((n + 3) & ~3));
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        f32x2 x = {sc[t][2 * hh], sc[t][2 * hh + 1]};
        const f32x2 lv = {-l4[2 * hh], -l4[2 * hh + 1]}, dlv = {d4[2 * hh], d4[2 * hh + 1]};
        const f32x2 fv = {f[2 * hh], f[2 * hh + 1]}, dpv = {dp[t][2 * hh], dp[t][2 * hh + 1]};
        x = __builtin_elementwise_fma(x, c2, lv);
        const f32x2 pv = {fast_exp2(x[0]), fast_exp2(x[1])};
        const f32x2 pd = pv * fv;
        const f32x2 dsv = pv * (dpv * fv - dlv);
        sc[t][2 * hh] = pd[0]; sc[t][2 * hh + 1] = pd[1];
        dp[t][2 * hh] = dsv[0]; dp[t][2 * hh + 1] = dsv[1];
      }
    }
}

// (two workgroups per CU: at three - 168 VGPRs, 38 spilled - the ViT3D-large train step drops from 50.5 to 45.6 volumes/s, same box)
template <bool DROP, typename T>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_wide_kernel(const r16* __restrict__ qkv, long ld, const r16* __restrict__ dout, long ldo,
                                                                   const float* __restrict__ lse, const float* __restrict__ delta, int n,
                                                                   int heads, float scale, r16* __restrict__ dqkv, long ldd, DropCfg drop) {
  // [stage][Q image | dO image | 64 lse | 64 delta].  Query rows beyond n arrive as zeros everywhere (buffer bounds): Q = dO = 0 makes
  // their scores, dP and delta zero, so their (unmasked) probabilities multiply zeros - the same exact zeros the streaming kernel adds
  constexpr int STG = 2 * IMG + 2 * TQ * 4;
  __shared__ __attribute__((aligned(16))) char wsmem[2 * STG];
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, g = lane >> 4;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const Grid2 gb = grid2d_xcd((n + WIDE_ROWS - 1) / WIDE_ROWS);                  // 1-D launch: whole heads per XCD
  const int b = gb.by / heads, h = gb.by - b * heads, inner = heads * DH;
  const r16* Q = qkv + (long)b * n * ld + h * DH;
  const r16* K = Q + inner;
  const r16* V = Q + 2 * inner;
  const r16* dO = dout + (long)b * n * ldo + h * DH;
  const int key0 = gb.bx * WIDE_ROWS + 32 * wid;
  const float scale_log2e = scale * 1.44269504088896340736f;

  r16x8 kf[2][2], vf[2][2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int krow = min(key0 + 16 * u + r, n - 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      kf[u][ks] = *reinterpret_cast<const r16x8*>(K + (long)krow * ld + 32 * ks + 8 * g);
      vf[u][ks] = *reinterpret_cast<const r16x8*>(V + (long)krow * ld + 32 * ks + 8 * g);
    }
  }
  const dma_desc rsQ = uniform_rsrc(Q, (unsigned)((((long)n - 1) * ld + DH) * 2));
  const dma_desc rsD = uniform_rsrc(dO, (unsigned)((((long)n - 1) * ldo + DH) * 2));
  const dma_desc rsL = uniform_rsrc((lse + ((long)b * heads + h) * n), (unsigned)(n * 4));
  const dma_desc rsDl = uniform_rsrc((delta + ((long)b * heads + h) * n), (unsigned)(n * 4));
  const int stepq = (int)(64 * ld * 2), stepd = (int)(64 * ldo * 2);
  auto issue = [&](int t, char* stage) {
    // piece (wid + 4 i) = rows 8 (wid + 4 i) .. + 7: the piece's first row goes into the scalar offset, the lane's row inside the
    // piece and its swizzled chunk (img128_off inverse) are recomputed here from a fresh lane id - this kernel is at the register
    // limit, and offsets held across the loop were spilled (their reload drains the DMA in flight: see fresh_lane)
    const int ln = fresh_lane(), lrow = ln >> 3, ch = ((ln & 7) ^ lrow) << 3;
    const int vq = (lrow * (int)ld + ch) * 2, vd = (lrow * (int)ldo + ch) * 2;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      lds_dma<16>(rsQ, stage + (wid + 4 * i) * 1024, vq, t * stepq + (wid + 4 * i) * 8 * (int)ld * 2);
      lds_dma<16>(rsD, stage + IMG + (wid + 4 * i) * 1024, vd, t * stepd + (wid + 4 * i) * 8 * (int)ldo * 2);
    }
    if (wid == 0) {                                       // 64 lse and 64 delta values of the tile: one dword per lane
      lds_dma<4>(rsL, stage + 2 * IMG, ln * 4, t * TQ * 4);
      lds_dma<4>(rsDl, stage + 2 * IMG + TQ * 4, ln * 4, t * TQ * 4);
    }
  };

  f32x4 dk[2][4], dv[2][4];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int t = 0; t < 4; ++t) dk[u][t] = dv[u][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nqt = (n + TQ - 1) / TQ;
  issue(0, wsmem);
  for (int qt = 0; qt < nqt; ++qt) {
    const char* cur = wsmem + (qt & 1) * STG;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (qt + 1 < nqt) issue(qt + 1, wsmem + ((qt + 1) & 1) * STG);
    const float* sL = reinterpret_cast<const float*>(cur + 2 * IMG);
    const float* sDl = sL + TQ;
    RowFrags F;
    if constexpr (DROP) {
      // under dropout the mask hashes need ~40 more registers than the budget of two workgroups per CU leaves (29 VGPRs went to
      // scratch inside the MFMA loop): the two 16-key groups run one after the other instead, so only one group's scores / dP are
      // live next to the hash temporaries; the price is reading the four fragment sets twice from LDS
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        f32x4 sc1[4], dp1[4];
        load_row_frags(cur, r, g, F);
        mfma_rows<T>(F, kf[u], sc1, true);
        __builtin_amdgcn_sched_barrier(0);
        load_row_frags(cur + IMG, r, g, F);
        mfma_rows<T>(F, vf[u], dp1, true);
        __builtin_amdgcn_sched_barrier(0);
        dkv_softmax_grad_inplace<DROP>(sc1, dp1, sL, sDl, qt, n, scale_log2e, drop, gb.by, key0 + 16 * u + r, g);
        r16x8 pf1[2], dsf1[2];
        pf1[0] = cvt8<T>(sc1[0], sc1[1]); pf1[1] = cvt8<T>(sc1[2], sc1[3]);
        dsf1[0] = cvt8<T>(dp1[0], dp1[1]); dsf1[1] = cvt8<T>(dp1[2], dp1[3]);
        __builtin_amdgcn_sched_barrier(0);
        load_tr_frags(cur + IMG, r, g, F);
        mfma_rows<T>(F, pf1, dv[u], false);
        __builtin_amdgcn_sched_barrier(0);
        load_tr_frags(cur, r, g, F);
        mfma_rows<T>(F, dsf1, dk[u], false);
        __builtin_amdgcn_sched_barrier(0);
      }
      continue;
    }
    f32x4 sc[2][4], dp[2][4];
    load_row_frags(cur, r, g, F);
    mfma_rows<T>(F, kf[0], sc[0], true);
    mfma_rows<T>(F, kf[1], sc[1], true);
    __builtin_amdgcn_sched_barrier(0);                    // (keeps one fragment set live at a time: the kernel is at the register limit)
    load_row_frags(cur + IMG, r, g, F);
    mfma_rows<T>(F, vf[0], dp[0], true);
    mfma_rows<T>(F, vf[1], dp[1], true);
    __builtin_amdgcn_sched_barrier(0);
    r16x8 pf[2][2], dsf[2][2];                           // rounded at once: 16 registers per group instead of 32
    dkv_softmax_grad_inplace<DROP>(sc[0], dp[0], sL, sDl, qt, n, scale_log2e, drop, gb.by, key0 + r, g);
    pf[0][0] = cvt8<T>(sc[0][0], sc[0][1]); pf[0][1] = cvt8<T>(sc[0][2], sc[0][3]);
    dsf[0][0] = cvt8<T>(dp[0][0], dp[0][1]); dsf[0][1] = cvt8<T>(dp[0][2], dp[0][3]);
    dkv_softmax_grad_inplace<DROP>(sc[1], dp[1], sL, sDl, qt, n, scale_log2e, drop, gb.by, key0 + 16 + r, g);
    pf[1][0] = cvt8<T>(sc[1][0], sc[1][1]); pf[1][1] = cvt8<T>(sc[1][2], sc[1][3]);
    dsf[1][0] = cvt8<T>(dp[1][0], dp[1][1]); dsf[1][1] = cvt8<T>(dp[1][2], dp[1][3]);
    __builtin_amdgcn_sched_barrier(0);
    load_tr_frags(cur + IMG, r, g, F);
    mfma_rows<T>(F, pf[0], dv[0], false);
    mfma_rows<T>(F, pf[1], dv[1], false);
    __builtin_amdgcn_sched_barrier(0);
    load_tr_frags(cur, r, g, F);
    mfma_rows<T>(F, dsf[0], dk[0], false);
    mfma_rows<T>(F, dsf[1], dk[1], false);
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int key = key0 + 16 * u + r;
    if (key < n) {
      r16* drow = dqkv + ((long)b * n + key) * ldd + h * DH;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        *reinterpret_cast<r16x4*>(drow + inner + 16 * t + 4 * g) = cvt4<T>(dk[u][t][0] * scale, dk[u][t][1] * scale, dk[u][t][2] * scale, dk[u][t][3] * scale);
        *reinterpret_cast<r16x4*>(drow + 2 * inner + 16 * t + 4 * g) = cvt4<T>(dv[u][t][0], dv[u][t][1], dv[u][t][2], dv[u][t][3]);
      }
    }
  }
}

// delta: [B, heads, n] fp32 scratch (written by the dQ kernel, read by the dK/dV kernel).
extern "C" int nv_attn_bwd(const void* qkv, long ld_qkv, const void* out, const void* dout, long ld_out, const float* lse, int B, int n,
                           int heads, int dim_head, float scale, float* delta, void* dqkv, long ld_dqkv, unsigned long drop_seed, float drop_p,
                           void* stream) {
  NV_CHECK_ARG(attn_generic_supported(dim_head), "nv_attn_bwd: dim_head=%d unsupported (multiples of 8 up to 128)", dim_head);
  NV_CHECK_ARG(B > 0 && n > 0 && heads > 0 && ld_qkv >= 3 * heads * dim_head && ld_dqkv >= 3 * heads * dim_head && ld_out >= heads * dim_head &&
                   (ld_qkv % 8) == 0 && (ld_out % 8) == 0 && (ld_dqkv % 4) == 0,
               "nv_attn_bwd: bad dims");
  NV_CHECK_ARG(nv_aligned16(qkv) && nv_aligned16(out) && nv_aligned16(dout) && nv_aligned16(dqkv), "nv_attn_bwd: alignment");
  hipStream_t s = (hipStream_t)stream;
  const DropCfg drop = make_drop(drop_seed, drop_p);
  if (dim_head != DH)
    return launch_attn_generic_bwd(qkv, ld_qkv, out, dout, ld_out, lse, B, n, heads, dim_head, scale, delta, dqkv, ld_dqkv, drop, s);
  const dim3 grid(((n + TQ - 1) / TQ) * B * heads);      // TQ == TK: the same 1-D grid serves the dQ and the dK / dV kernel
  const bool dropping = drop.thresh != 0, fp16 = nv_operand_format() == NV_OPERAND_FP16;
  const r16* q16 = (const r16*)qkv; const r16* o16 = (const r16*)out; const r16* do16 = (const r16*)dout; r16* dq16 = (r16*)dqkv;
  const int slot = nv_prof_begin(4, 10.0 * B * heads * (double)n * n * DH, stream);   // algorithmic: 5 products
  if (attn_resident(n) && g_attn_mode != 3) {
    NV_CHECK_ARG((long)n * ld_qkv < (1L << 30) && (long)n * ld_out < (1L << 30), "nv_attn_bwd: operand too large for 32-bit buffer offsets");
    const int nt = (n + TK - 1) / TK;
    static bool attr = false;
    if (!attr) {
      ATTN_ATTR(K_DQ_RES, 2 * RES_MAX_TILES * IMG);
      ATTN_ATTR(K_DKV_RES, 2 * RES_MAX_TILES * IMG + 2 * RES_MAX_TILES * TQ * 4);
      ATTN_ATTR(K_BWD_RES, 2 * RES_MAX_TILES * IMG + 2 * RES_MAX_TILES * TQ * 4);
      attr = true;
    }
    const dim3 rgrid(attn_res_blocks(n) * B * heads);       // 1-D: whole heads per XCD (res_block)
    if (g_attn_bwd_merged) {
      ATTN_LAUNCH(K_BWD_RES, dim3(2 * rgrid.x), dim3(RES_THREADS), 2 * nt * IMG + 2 * nt * TQ * 4, s, q16, ld_qkv, o16, do16, ld_out, lse, n, heads, scale, delta, dq16, ld_dqkv, drop);
      nv_prof_end(slot, stream);
      NV_CHECK_LAUNCH("nv_attn_bwd(resident, one launch)");
      return NV_OK;
    }
    ATTN_LAUNCH(K_DQ_RES, rgrid, dim3(RES_THREADS), 2 * nt * IMG, s, q16, ld_qkv, o16, do16, ld_out, lse, n, heads, scale, delta, dq16, ld_dqkv, drop);
    NV_CHECK_LAUNCH("nv_attn_bwd/dq(resident)");
    ATTN_LAUNCH(K_DKV_RES, rgrid, dim3(RES_THREADS), 2 * nt * IMG + 2 * nt * TQ * 4, s, q16, ld_qkv, do16, ld_out, lse, delta, n, heads, scale, dq16, ld_dqkv, drop);
    nv_prof_end(slot, stream);
    NV_CHECK_LAUNCH("nv_attn_bwd/dkv(resident)");
    return NV_OK;
  }
  // wide kernels (32 rows / keys per wave) once the grid is large enough to fill the chip with them; small grids keep 16
  const bool wide = g_attn_mode != 1 && (long)n * ld_qkv < (1L << 30) && (long)n * ld_out < (1L << 30) &&
                    (g_attn_mode >= 3 || (long)B * heads * ((n + WIDE_ROWS - 1) / WIDE_ROWS) >= 512);
  const dim3 wgrid(((n + WIDE_ROWS - 1) / WIDE_ROWS) * B * heads);
  if (wide) ATTN_LAUNCH(K_DQ_WIDE, wgrid, dim3(256), 0, s, q16, ld_qkv, o16, do16, ld_out, lse, n, heads, scale, delta, dq16, ld_dqkv, drop);
  else ATTN_LAUNCH(K_DQ, grid, dim3(256), 0, s, q16, ld_qkv, o16, do16, ld_out, lse, n, heads, scale, delta, dq16, ld_dqkv, drop);
  NV_CHECK_LAUNCH("nv_attn_bwd/dq");
  if (wide) ATTN_LAUNCH(K_DKV_WIDE, wgrid, dim3(256), 0, s, q16, ld_qkv, do16, ld_out, lse, delta, n, heads, scale, dq16, ld_dqkv, drop);
  else ATTN_LAUNCH(K_DKV, grid, dim3(256), 0, s, q16, ld_qkv, do16, ld_out, lse, delta, n, heads, scale, dq16, ld_dqkv, drop);
  nv_prof_end(slot, stream);
  NV_CHECK_LAUNCH("nv_attn_bwd/dkv");
  return NV_OK;
}
