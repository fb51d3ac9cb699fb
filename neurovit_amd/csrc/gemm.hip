// bf16 MFMA GEMM for gfx950 with fused epilogues - the workhorse of the ViT3D hot path.
//
//   C[M,N] = op(A) . op(B)   (fp32 accumulate, v_mfma_f32_16x16x32_bf16)
//     layout NT: A [M,K] row-major, B [N,K] row-major   (y = x W^T      : forward linears, vit_3d.py:19,22,41,44)
//     layout NN: A [M,K] row-major, B [K,N] row-major   (dx = dy W      : data gradients)
//     layout TN: A [K,M] row-major, B [K,N] row-major   (dW = dy^T x    : weight gradients)
//
// Workgroup tile BM x BN x 64 with 2x2 waves; BM, BN in {64, 128} chosen per problem so that the grid fills the
// 256 CUs several times over (the ViT3D-base shapes have only M = 2052 rows: 128x128 tiles would leave most CUs
// with a single wave per SIMD and nothing to hide latency behind).  Operands are staged by LDS-DMA
// (global_load_lds_dwordx4: no staging VGPRs, no ds_write pass; the swizzle is applied on the per-lane SOURCE
// address because the LDS destination of one wave-instruction is linear), two LDS buffers, one barrier per
// K tile; only a ragged K tail goes through registers.  K-contiguous operands use a
// [rows][64] image + ds_read_b128; K-strided ("T") operands keep their memory layout in a [64][cols] image and
// are transposed for free by ds_read_b64_tr_b16.  All images are XOR-swizzled: zero LDS bank conflicts measured
// (SQ_LDS_BANK_CONFLICT = 0).  The MFMA is issued with swapped operands (D = B_frag x A_frag) so every lane ends
// up holding four CONSECUTIVE output columns of one row: epilogue loads/stores are 8-16 bytes per lane.
// Ragged M/N need no masking on the load side (out-of-range rows/columns are clamped to valid memory and their
// results never stored); only a ragged K tail is zero-filled, in a peeled last iteration.
#include "gemm_common.h"
#include "gemm_pp.h"

template <typename T, int BM, int BN, bool A_T, bool B_T, int EPI>
__global__ __launch_bounds__(NTHREADS) void gemm_bf16_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
  constexpr int TM = BM / 2, TN = BN / 2;        // wave tile (2 x 2 waves)
  constexpr int MI = TM / 16, NI = TN / 16;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (g.col_order ? bid % tiles_m : bid / tiles_n) * BM, n0 = (g.col_order ? bid / tiles_m : bid % tiles_n) * BN;
  const int nk = (g.K + BK - 1) / BK;

  char* sA0 = smem;
  char* sB0 = smem + A_BYTES;
  char* sA1 = smem + A_BYTES + B_BYTES;
  char* sB1 = smem + 2 * A_BYTES + B_BYTES;

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // LDS-DMA source pointers (advance by one K tile per issue)
  const int uwid = __builtin_amdgcn_readfirstlane(wid);
  const r16* pa[BM / 32];
  const r16* pb[BN / 32];
  if constexpr (A_T) dma_init_kmajor<BM>(g.A, g.lda, g.M, m0, uwid, lane, pa); else dma_init_rowmajor<BM>(g.A, g.lda, g.M, m0, uwid, lane, pa);
  if constexpr (B_T) dma_init_kmajor<BN>(g.B, g.ldb, g.N, n0, uwid, lane, pb); else dma_init_rowmajor<BN>(g.B, g.ldb, g.N, n0, uwid, lane, pb);
  const long stepA = A_T ? (long)BK * g.lda : BK, stepB = B_T ? (long)BK * g.ldb : BK;
  const int nfull = g.K / BK;          // K tiles that need no zero fill

  // ragged K tail (only the weight-gradient GEMMs, K = tokens): staged through registers with zero fill
  uint4 ra[BM / 32], rb[BN / 32];
  auto gload_tail = [&](int kt) {
    const int k0 = kt * BK;
    if constexpr (A_T) gload_kmajor<BM, true>(g.A, g.lda, g.M, g.K, m0, k0, tid, ra);
    else gload_rowmajor<BM, true>(g.A, g.lda, g.M, g.K, m0, k0, tid, ra);
    if constexpr (B_T) gload_kmajor<BN, true>(g.B, g.ldb, g.N, g.K, n0, k0, tid, rb);
    else gload_rowmajor<BN, true>(g.B, g.ldb, g.N, g.K, n0, k0, tid, rb);
  };
  auto swrite = [&](char* a, char* b) {
    if constexpr (A_T) swrite_kmajor<BM>(a, tid, ra); else swrite_rowmajor<BM>(a, tid, ra);
    if constexpr (B_T) swrite_kmajor<BN>(b, tid, rb); else swrite_rowmajor<BN>(b, tid, rb);
  };

  if (nfull > 0) {
    dma_issue<BM>(pa, stepA, sA0, uwid);
    dma_issue<BN>(pb, stepB, sB0, uwid);
  } else {
    gload_tail(0);
    swrite(sA0, sB0);
  }
  __syncthreads();   // hipcc drains the outstanding LDS-DMA (vmcnt(0)) in front of the barrier

  for (int kt = 0; kt < nk; ++kt) {
    const char* a = (kt & 1) ? sA1 : sA0;
    const char* b = (kt & 1) ? sB1 : sB0;
    char* na = (kt & 1) ? sA0 : sA1;
    char* nb = (kt & 1) ? sB0 : sB1;
    const bool next_dma = kt + 1 < nfull, next_tail = (kt + 1 < nk) && !next_dma;
    if (next_dma) {              // the buffer was last read in iteration kt-1; every wave has passed that barrier
      dma_issue<BM>(pa, stepA, na, uwid);
      dma_issue<BN>(pb, stepB, nb, uwid);
    } else if (next_tail) {
      gload_tail(kt + 1);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      r16x8 af[MI], bfr[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = read_frag<A_T, BM>(a, wm * TM + 16 * i, ks, lane);
#pragma unroll
      for (int j = 0; j < NI; ++j) bfr[j] = read_frag<B_T, BN>(b, wn * TN + 16 * j, ks, lane);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = mfma16<T>(bfr[j], af[i], acc[i][j]);
    }
    if (next_tail) swrite(na, nb);
    __syncthreads();
  }

  epilogue<EPI, T, MI, NI>(acc, g, m0 + wm * TM, n0 + wn * TN, lane);
}

// =====================================================================================================
// Warp-specialised BM x BN x 64 kernel (512 threads): waves 0-3 are CONSUMERS (2x2, MFMA + ds_read only), waves
// 4-7 are LOADERS (LDS-DMA only).  Measured on MI355X: waves that do nothing but issue buffer_load...lds stream
// L2 -> LDS at 122-134 GB/s per CU (tools/l2_stream_bench.hip), three times what a wave that also issues MFMAs
// sustains, because every DMA issue stalls the in-order instruction stream behind it.  A 128x128x64 step moves
// 32 KiB (~550 cycles at that rate) for 512 cycles of MFMA per SIMD: with the two roles on different waves of
// each SIMD both pipes run concurrently.
//   * three LDS stages; loaders run two K tiles ahead (counted vmcnt), one raw s_barrier per K tile joins all
//     eight waves: after barrier kt tile kt+1 is complete in LDS and the stage of tile kt-1 is free;
//   * consumers double-buffer the MFMA operand fragments so the ds_reads of one 32-deep half step overlap the
//     MFMAs of the other;
//   * buffer descriptors return zeros for out-of-range rows (ragged M / N, ragged token count K of the
//     weight-gradient GEMMs): no masks, no clamps.  Requires K % 64 == 0 for K-contiguous operands.
// =====================================================================================================
constexpr int WS_THREADS = 512;

template <bool T, int BX>
__device__ __forceinline__ void ws_offsets(long ld, int rc0, int lw, int lane, int (&voff)[BX / 32]) {
#pragma unroll
  for (int i = 0; i < BX / 32; ++i) {
    const int I = lw + 4 * i;
    if constexpr (!T) {
      const int row = I * 8 + (lane >> 3), ch = (lane & 7) ^ ((lane >> 3) & 7);           // img128_off inverse
      voff[i] = (int)((((long)(rc0 + row)) * ld + (ch << 3)) * 2);
    } else if constexpr (BX == 128) {
      const int krow = I * 4 + (lane >> 4);
      const int ch = (lane & 15) ^ (((krow & 3) << 2) | ((krow >> 2) & 3));               // img256_off inverse
      voff[i] = (int)(((long)krow * ld + rc0 + (ch << 3)) * 2);
    } else {
      const int krow = I * 8 + (lane >> 3);
      const int ch = (lane & 7) ^ ((((krow >> 1) & 1) | (((krow >> 3) & 1) << 1)) << 1);  // img128t_off inverse
      voff[i] = (int)(((long)krow * ld + rc0 + (ch << 3)) * 2);
    }
  }
}
template <int N>
__device__ __forceinline__ void ws_issue(__amdgpu_buffer_rsrc_t rsrc, const int* voff, int soff, char* img, int lw) {
#pragma unroll
  for (int i = 0; i < N; ++i)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)(img + (lw + 4 * i) * 1024), 16, voff[i], soff, 0, 0);
}
// wait until at most `tiles` K tiles (PIECES DMA instructions each) of this wave are still in flight
#define WS_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
template <int PIECES>
__device__ __forceinline__ void ws_wait_tiles_in_flight(int tiles) {
  static_assert(PIECES == 4 || PIECES == 6 || PIECES == 8 || PIECES == 12 || PIECES == 16, "unsupported DMA piece count");
  if constexpr (PIECES == 16) {
    switch (tiles) { case 0: WS_VM(0); break; case 1: WS_VM(16); break; case 2: WS_VM(32); break; default: WS_VM(48); }
  } else if constexpr (PIECES == 12) {
    switch (tiles) { case 0: WS_VM(0); break; case 1: WS_VM(12); break; case 2: WS_VM(24); break; case 3: WS_VM(36); break; default: WS_VM(48); }
  } else if constexpr (PIECES == 8) {
    switch (tiles) { case 0: WS_VM(0); break; case 1: WS_VM(8); break; case 2: WS_VM(16); break; case 3: WS_VM(24); break; case 4: WS_VM(32); break; default: WS_VM(40); }
  } else if constexpr (PIECES == 6) {
    switch (tiles) { case 0: WS_VM(0); break; case 1: WS_VM(6); break; case 2: WS_VM(12); break; case 3: WS_VM(18); break; case 4: WS_VM(24); break; default: WS_VM(30); }
  } else {
    switch (tiles) { case 0: WS_VM(0); break; case 1: WS_VM(4); break; case 2: WS_VM(8); break; case 3: WS_VM(12); break; case 4: WS_VM(16); break; default: WS_VM(20); }
  }
}
#undef WS_VM

template <typename T, int BM, int BN, int S, int KS, bool A_T, bool B_T, int EPI>   // S = LDS ring stages, KS = 64-deep sub-tiles per stage
__device__ __forceinline__ void gemm_ws_body(const GemmArgs& g, const int bid) {   // bid = logical tile of this workgroup
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int A_BYTES = BM * BK * 2, SUB = (BM + BN) * BK * 2, STAGE = KS * SUB;
  constexpr int TM = BM / 2, TN = BN / 2, MI = TM / 16, NI = TN / 16;
  constexpr int PIECES = KS * (BM / 32 + BN / 32);      // DMA instructions per loader wave per stage
  constexpr int D = S - 1;                              // loaders run S-1 stages ahead; barrier kt sits at the END of stage kt
  static_assert(S >= 3 && S <= 7 && (KS == 1 || KS == 2), "ring geometry");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM;
  const int m0 = (g.col_order ? bid % tiles_m : bid / tiles_n) * BM, n0 = (g.col_order ? bid / tiles_m : bid % tiles_n) * BN;
  const int nk = ((g.K + BK - 1) / BK + KS - 1) / KS;   // ring stages to process

  if (wid >= 4) {
    // ------------------------------------------------------------------ loader waves
    const int lw = wid - 4;
    const unsigned bytesA = (unsigned)((((long)(A_T ? g.K : g.M) - 1) * g.lda + (A_T ? g.M : g.K)) * 2);
    const unsigned bytesB = (unsigned)((((long)(B_T ? g.K : g.N) - 1) * g.ldb + (B_T ? g.N : g.K)) * 2);
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, bytesA, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)g.B, 0, bytesB, 0x00020000);
    int voA[BM / 32], voB[BN / 32];
    ws_offsets<A_T, BM>(g.lda, m0, lw, lane, voA);
    ws_offsets<B_T, BN>(g.ldb, n0, lw, lane, voB);
    const int stepA = (int)((A_T ? (long)BK * g.lda : BK) * 2), stepB = (int)((B_T ? (long)BK * g.ldb : BK) * 2);
    auto issue_stage = [&](int t, char* dst) {
#pragma unroll
      for (int u = 0; u < KS; ++u) {
        ws_issue<BM / 32>(rA, voA, (t * KS + u) * stepA, dst + u * SUB, lw);
        ws_issue<BN / 32>(rB, voB, (t * KS + u) * stepB, dst + u * SUB + A_BYTES, lw);
      }
    };
    const int pre = nk < D ? nk : D;
    for (int t = 0; t < pre; ++t) issue_stage(t, smem + t * STAGE);
    ws_wait_tiles_in_flight<PIECES>(pre - 1);
    __builtin_amdgcn_s_barrier();                       // barrier -1: stage 0 is in LDS
    int fs = D % S;                                     // ring slot of stage kt + D
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + D < nk) issue_stage(kt + D, smem + fs * STAGE);   // slot of stage kt-1: every consumer read retired at barrier kt-1
      const int last = (kt + D < nk) ? kt + D : nk - 1; // newest stage issued so far
      ws_wait_tiles_in_flight<PIECES>(last - (kt + 1) > 0 ? last - (kt + 1) : 0);   // stage kt+1 landed (this wave's share)
      __builtin_amdgcn_s_barrier();                     // barrier kt
      fs = (fs + 1 == S) ? 0 : fs + 1;
    }
    __builtin_amdgcn_s_barrier();                       // barrier E: the consumers have parked the C tile in LDS
    epilogue_lds<EPI, T, BM, BN, WS_THREADS>(smem, g, m0, n0, tid, reinterpret_cast<const float*>(smem + S * STAGE));
    return;
  }

  // -------------------------------------------------------------------- consumer waves
  const int wm = wid >> 1, wn = wid & 1;
  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  r16x8 fa0[MI], fb0[NI], fa1[MI], fb1[NI];
  int ci = 0;                                           // ring slot of the current stage
  char* cur = smem;
  char* nxt = smem + STAGE;

// half step HS of a stage = sub-tile HS/2, 32-deep k half HS%2
#define WS_READ(FA, FB, BUF, HS)                                                                                   \
  _Pragma("unroll") for (int i = 0; i < MI; ++i) FA[i] = read_frag<A_T, BM>(BUF + ((HS) >> 1) * SUB, wm * TM + 16 * i, (HS) & 1, lane); \
  _Pragma("unroll") for (int j = 0; j < NI; ++j) FB[j] = read_frag<B_T, BN>(BUF + ((HS) >> 1) * SUB + A_BYTES, wn * TN + 16 * j, (HS) & 1, lane);
#define WS_MFMA(FA, FB)                                                       \
  _Pragma("unroll") for (int i = 0; i < MI; ++i)                             \
    _Pragma("unroll") for (int j = 0; j < NI; ++j)                           \
      acc[i][j] = mfma16<T>(FB[j], FA[i], acc[i][j]);
#define SB __builtin_amdgcn_sched_barrier(0);   // pin the phase order: [reads of a later half step][MFMAs of this one]
#define WS_ADVANCE ci = (ci + 1 == S) ? 0 : ci + 1; cur = nxt; nxt = smem + ((ci + 1 == S) ? 0 : ci + 1) * STAGE;

  if constexpr (epi_is_fold<EPI>()) {                  // row statistics of the folded LayerNorm, while the first stage is on its way (gemm_common.h)
    ln_rows_to_lds(g, m0, BM, reinterpret_cast<float*>(smem + S * STAGE), tid, 256);
    ln_cols_to_lds<BM, BN>(g, n0, reinterpret_cast<float*>(smem + S * STAGE), 255 - tid);      // (from the far end: the rows are 64 or 128 threads' work)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();                         // barrier -1
  if constexpr (KS == 1) {
    WS_READ(fa0, fb0, cur, 0)
    for (int kt = 0; kt + 1 < nk; ++kt) {
      WS_READ(fa1, fb1, cur, 1)
      SB WS_MFMA(fa0, fb0) SB
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // all reads of `cur` have returned before it can be refilled
      __builtin_amdgcn_s_barrier();                     // barrier kt: stage kt+1 complete
      SB WS_READ(fa0, fb0, nxt, 0)
      SB WS_MFMA(fa1, fb1) SB
      WS_ADVANCE
    }
    WS_READ(fa1, fb1, cur, 1)                           // last stage
    SB WS_MFMA(fa0, fb0) SB
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // every fragment read has returned: the ring may be overwritten
    __builtin_amdgcn_s_barrier();                       // barrier nk-1 (pairs with the loaders' last one; no DMA in flight any more)
    WS_MFMA(fa1, fb1)
  } else {
    // Four half steps per stage and three fragment register sets rotating once per stage ((a,b,c) -> (b,c,a)): three of the
    // four half steps get their fragments TWO half steps ahead (the 8-16 MFMAs of one half step, 128-256 cycles, alone do
    // not cover an LDS read under load); only the first half step of a stage, whose reads must follow the barrier, is one
    // ahead.  Every stage body is branch free (the last stage prefetches two unused half steps from a valid slot), so hipcc
    // derives exact lgkmcnt values.
    r16x8 fa2[MI], fb2[NI];
#define WS_STAGE(A0, B0, A1, B1, A2, B2)                                      \
    WS_READ(A2, B2, cur, 2)                                                   \
    SB WS_MFMA(A0, B0) SB                                                     \
    WS_READ(A0, B0, cur, 3)                                                   \
    SB WS_MFMA(A1, B1) SB                                                     \
    WS_MFMA(A2, B2) SB                                                        \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                        \
    __builtin_amdgcn_s_barrier();                                             \
    SB WS_READ(A1, B1, nxt, 0)                                                \
    WS_READ(A2, B2, nxt, 1)                                                   \
    SB WS_MFMA(A0, B0) SB                                                     \
    WS_ADVANCE
    WS_READ(fa0, fb0, cur, 0)
    WS_READ(fa1, fb1, cur, 1)
    int kt = 0;
    for (; kt + 3 <= nk; kt += 3) {
      WS_STAGE(fa0, fb0, fa1, fb1, fa2, fb2)
      WS_STAGE(fa1, fb1, fa2, fb2, fa0, fb0)
      WS_STAGE(fa2, fb2, fa0, fb0, fa1, fb1)
    }
    if (kt + 1 <= nk) {
      WS_STAGE(fa0, fb0, fa1, fb1, fa2, fb2)
      if (kt + 2 <= nk) { WS_STAGE(fa1, fb1, fa2, fb2, fa0, fb0) }
    }
#undef WS_STAGE
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the stray prefetches of the last stage have returned
  }
#undef WS_READ
#undef WS_MFMA
#undef SB
#undef WS_ADVANCE
  static_assert(BM * cpitch<BN>() + colsum_scratch_bytes<BM, BN, WS_THREADS>() <= S * STAGE, "C tile (+ column-sum scratch) must fit in the ring");
  park_acc<MI, NI, BN>(acc, smem, wm * TM, wn * TN, lane);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the raw barrier carries no wait: the parked tile must be written first
  __builtin_amdgcn_s_barrier();                         // barrier E
  epilogue_lds<EPI, T, BM, BN, WS_THREADS>(smem, g, m0, n0, tid, reinterpret_cast<const float*>(smem + S * STAGE));
}

template <typename T, int BM, int BN, int S, int KS, bool A_T, bool B_T, int EPI>
__global__ __launch_bounds__(WS_THREADS) void gemm_ws_kernel(const GemmArgs g) {
  gemm_ws_body<T, BM, BN, S, KS, A_T, B_T, EPI>(g, xcd_remap(blockIdx.x, gridDim.x));
}

template <typename T, int BM, int BN, int S, int KS, bool A_T, bool B_T, int EPI>
__global__ __launch_bounds__(WS_THREADS) void gemm_ws_grouped_kernel(const GemmGroup G) {
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  int p = 0;
  while (p + 1 < G.count && bid >= G.tile_end[p]) ++p;      // workgroup-uniform
  gemm_ws_body<T, BM, BN, S, KS, A_T, B_T, EPI>(G.p[p], bid - (p ? G.tile_end[p - 1] : 0));
}

// ---- host side ---------------------------------------------------------------------------------------
static int g_ring_override = 0;   // 0 heuristic, 1 = 3 x 64, 3 = 3 x 128
static int g_tile_override = 0;   // 0 = heuristic; 1 / 3 = warp-specialised 128x128 / 64x128; 4 = ping-pong 256x128; 5 = never ping-pong;
                                  // else BM*1000 + BN (small-tile kernel)
static int g_pp_grouped = 1;      // grouped weight-gradient launch on ping-pong tiles
static int g_pp_min_tiles = 128;
static int g_pq_min_tiles = 768;       // problems with at least this many 256 x 256 tiles go to gemm_pq.hip: measured +5-8 % on 8192^2 x 4096
                                       // (1.14 vs 1.05-1.10 PFLOP/s), equal on ViT3D-large's FC1, worse below (tile quantisation on 256 CUs)   // heuristic: problems with at least this many 256 x 128 tiles go to the ping-pong kernel
extern "C" int nv_gemm_set_tile(int bm, int bn) {   // tuning aid (tools/gemm_bench.py)
  if (bm == 6) { g_pp_min_tiles = bn; return 0; }
  if (bm == 7) { g_pp_grouped = bn; return 0; }
  if (bm == 8) { g_pp_dbg = bn; return 0; }
  if (bm == 10) { g_pq_min_tiles = bn; return 0; }
  if (bm == 11) { g_pp_w32 = bn ? 1 : 0; return 0; }     // NT problems of the 256 x 128 kernel on 32 x 32 x 16 MFMAs
  if (bm == 12) { g_pp_adamw_wgs = bn > 0 ? bn : 0; return 0; }     // workgroups of nv_gemm_bf16_grouped_adamw (0 = one per tile)
  if (bm == 14) { g_pp_wgrad_wgs = bn > 0 ? bn : 0; return 0; }      // workgroups of the grouped weight-gradient launch (0 = one per tile)
  if (bm == 13) { extern int g_adamw_ranges_cap; g_adamw_ranges_cap = bn > 0 ? bn : 0; return 0; }
  const bool small = (bm == 64 && (bn == 64 || bn == 128)) || (bm == 128 && bn == 128);
  NV_CHECK_ARG(bm == 0 || bm == 1 || (bm >= 3 && bm <= 5) || bm == 9 || small, "nv_gemm_set_tile: (%d, %d) is not a compiled tile", bm, bn);
  NV_CHECK_ARG(!(bm == 1 || bm == 3) || bn == 0 || bn == 1 || (bm == 3 && bn == 3), "nv_gemm_set_tile: ring %d is not compiled for tile %d", bn, bm);
  g_tile_override = (bm == 0) ? 0 : ((bm <= 5 || bm == 9) ? bm : bm * 1000 + bn);
  g_ring_override = (bm == 1 || bm == 3) ? bn : 0;      // for the warp-specialised tiles bn selects the ring: 1 = 3 x 64-deep, 3 = 3 x 128-deep (64 x 128 only)
  return 0;
}

template <typename T, int BM, int BN, bool A_T, bool B_T, int EPI>
static int launch_tile(const GemmArgs& a, hipStream_t s) {
  constexpr int LDS = 2 * (BM + BN) * BK * 2;
  const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  auto kern = gemm_bf16_kernel<T, BM, BN, A_T, B_T, EPI>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set = true;
  }
  const int slot = nv_prof_begin((A_T ? 2 : (B_T ? 1 : 0)), 2.0 * a.M * a.N * a.K, s);
  nv_prof_bytes(slot, gemm_algo_bytes(a, EPI, 2));
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(NTHREADS), LDS, s, a);
  nv_prof_end(slot, s);
  NV_CHECK_LAUNCH("nv_gemm_bf16");
  return NV_OK;
}

template <typename T, int BM, int BN, int S, int KS, bool A_T, bool B_T, int EPI>
static int launch_ws(const GemmArgs& a, hipStream_t s) {
  constexpr int LDS = S * KS * (BM + BN) * BK * 2 + ln_rows_bytes<EPI, BM, BN>();
  static_assert(LDS <= 160 * 1024, "LDS ring too large");
  const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  auto kern = gemm_ws_kernel<T, BM, BN, S, KS, A_T, B_T, EPI>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set = true;
  }
  const int slot = nv_prof_begin((A_T ? 2 : (B_T ? 1 : 0)), 2.0 * a.M * a.N * a.K, s);
  nv_prof_bytes(slot, gemm_algo_bytes(a, EPI, 2));
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(WS_THREADS), LDS, s, a);
  nv_prof_end(slot, s);
  NV_CHECK_LAUNCH("nv_gemm_bf16/ws");
  return NV_OK;
}

// Kernel choice for one problem (shared by the launcher and by nv_gemm_tile_rows, which tells the caller how many partial rows a
// fused column-sum epilogue produces).  family: 0 small-tile register-epilogue kernel, 1 warp-specialised (ws = 1: 128x128,
// 2: 128x64, 3: 64x128; ring as below), 2 eight-wave ping-pong 256 x 128.
struct GemmPlan { int family, ws, ring, sel, bm; };
static GemmPlan plan_gemm(bool A_T, bool B_T, int epi, int M, int N, int K, long lda, long ldb) {
  GemmPlan p{0, 0, 0, 0, 0};
  // large-tile kernels: need whole K tiles for K-contiguous operands (buffer bounds zero-fill a ragged K only when K is the row
  // index, i.e. for "T" operands) and 31-bit byte offsets; tiny problems keep the small-tile kernel.
  const bool k_ok = (K % BK == 0) || (A_T && B_T);
  const long rowsA = A_T ? K : M, rowsB = B_T ? K : N;
  const bool fits = rowsA * lda < (1L << 30) && rowsB * ldb < (1L << 30);
  const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
  const long tpp = (long)((M + PP_BM - 1) / PP_BM) * ((N + PP_BN - 1) / PP_BN);
  const long tpq = (long)((M + PQ_BM - 1) / PQ_BM) * ((N + PQ_BN - 1) / PQ_BN);
  // 256 x 256: by a same-box A/B it wins on square problems and on the large preset's qkv (126 vs 133 us), ties on FC1 + GELU and loses
  // to 256 x 128 wherever the epilogue or a transposed operand weighs (dU + GELU' 283 vs 270 us, N = 1024 data gradients, TN): plain
  // stores of NT problems only
  const bool pq_shape = !A_T && !B_T && (epi == EPI_STORE_BF16 || epi == EPI_STORE_F32 || epi == EPI_BIAS_F32);
  if (k_ok && fits && (g_tile_override == 9 || (g_tile_override == 0 && pq_shape && tpq >= g_pq_min_tiles))) {
    p.family = 3; p.bm = 128;              // 256 x 256 tiles whose epilogue runs in two passes of 128 rows
    return p;
  }
  if (k_ok && fits && (g_tile_override == 4 || (g_tile_override == 0 && tpp >= g_pp_min_tiles))) {
    p.family = 2; p.bm = PP_BM;
    return p;
  }
  if (k_ok && fits && g_tile_override < 1000) {
    int ws = (g_tile_override >= 4) ? 0 : g_tile_override;                 // 1: 128x128, 2: 128x64, 3: 64x128 (forced); 0: heuristic
    // measured on the ViT3D-base shapes (M = 2052): the 64x128 tile wins or ties everywhere (two workgroups per CU, so
    // one block's epilogue overlaps the other's MFMA phase); very large problems prefer 128x128 (less LDS / L2 traffic)
    // (round 5: the gate was t128 >= 32 and left the N = 768 problems of ONE volume - M = 513: 5 x 6 tiles of 128 x 128 - on the small-tile kernel:
    //  FC2 25.6 us against 16.5 on 54 tiles of 64 x 128 with the 128-deep ring, dxn2 31.2 against 15.5, dxn1 22.5 against 12.0)
    const long t64x128w = (long)((M + 63) / 64) * ((N + 127) / 128);
    if (ws == 0 && (t128 >= 32 || t64x128w >= 48)) ws = (t128 >= 1024) ? 1 : 3;
    // ring geometry: 1 = 3 stages x 64-deep, 2 = deep ring (6 / 4 stages x 64), 3 = 3 stages x 128-deep (one barrier per 128 of K;
    // needs whole 128-deep steps unless both operands are K-strided, where the buffer bounds zero-fill)
    int ring = g_ring_override;
    const bool k128_ok = ((K % (2 * BK)) == 0) || (A_T && B_T);
    // measured (profiles/r01_gemm_shapes_tiles.log): 128-deep steps win when the grid leaves one workgroup per CU anyway and
    // K is long; otherwise two co-resident 3 x 64 workgroups per CU (72 KiB each) overlap each other's epilogue and waits
    const long t64x128 = (long)((M + 63) / 64) * ((N + 127) / 128);
    if (ring == 0) ring = (k128_ok && t64x128 <= 256 && K >= 1536) ? 3 : 1;
    if (ring == 3 && !k128_ok) ring = 1;
    if (ws >= 1 && ws <= 3) {
      p.family = 1; p.ws = ws; p.ring = ring; p.bm = (ws == 3) ? 64 : 128;
      return p;
    }
  }
  p.sel = g_tile_override >= 1000 ? g_tile_override : 0;
  if (!p.sel) {
    // enough workgroups to give every one of the 256 CUs several co-resident blocks; prefer the larger tile
    // (less LDS traffic per MFMA) when the problem is big enough.
    const long t64x128 = (long)((M + 63) / 64) * ((N + 127) / 128);
    p.sel = (t128 >= 1024) ? 128128 : (t64x128 >= 768 ? 64128 : 64064);
  }
  return p;
}

template <typename T, bool A_T, bool B_T, int EPI>
static int launch_fmt(const GemmArgs& a, hipStream_t s) {
  const GemmPlan p = plan_gemm(A_T, B_T, EPI, a.M, a.N, a.K, a.lda, a.ldb);
  if (p.family == 3) return launch_pq(A_T ? 2 : (B_T ? 1 : 0), EPI, a, s);
  if (p.family == 2) return launch_pp(A_T ? 2 : (B_T ? 1 : 0), EPI, a, s);
  if (p.family == 1) {
    const int ws = p.ws, ring = p.ring;
    // (the 128 x 64 tile and the deep 6 / 4-stage rings, reachable only through nv_gemm_set_tile, lost every comparison of
    // profiles/r01_gemm_shapes_tiles.log and are no longer compiled: 50 instantiations)
    if (ws == 1) return launch_ws<T, 128, 128, 3, 1, A_T, B_T, EPI>(a, s);
    return ring == 3 ? launch_ws<T, 64, 128, 3, 2, A_T, B_T, EPI>(a, s) : launch_ws<T, 64, 128, 3, 1, A_T, B_T, EPI>(a, s);
  }
  if constexpr (EPI == EPI_DGELU_COLSUM) {
    nv_set_error("nv_gemm_bf16: the fused column-sum epilogue needs a large-tile kernel for this shape (ask nv_gemm_tile_rows first)");
    return NV_ERR_ARG;
  } else {
    switch (p.sel) {
      case 128128: return launch_tile<T, 128, 128, A_T, B_T, EPI>(a, s);
      case 64128: return launch_tile<T, 64, 128, A_T, B_T, EPI>(a, s);
      default: return launch_tile<T, 64, 64, A_T, B_T, EPI>(a, s);
    }
  }
}

template <bool A_T, bool B_T, int EPI>
static int launch(const GemmArgs& a, hipStream_t s) {
  NV_DISPATCH_OPERAND(T, return launch_fmt<T, A_T, B_T, EPI>(a, s));
}

// ---- LayerNorm folded into the GEMMs around it (gemm_common.h EPI_BIAS_RESID_LN / EPI_LNFOLD_*): NT problems on the kernels with an LDS epilogue
template <typename T, int EPI>
static int launch_fold_fmt(const GemmArgs& a, hipStream_t s) {
  const GemmPlan p = plan_gemm(false, false, EPI, a.M, a.N, a.K, a.lda, a.ldb);
  if (p.family == 2) return launch_pp(0, EPI, a, s);
  if (p.family == 1) {
    if (p.ws == 1) return launch_ws<T, 128, 128, 3, 1, false, false, EPI>(a, s);
    return p.ring == 3 ? launch_ws<T, 64, 128, 3, 2, false, false, EPI>(a, s) : launch_ws<T, 64, 128, 3, 1, false, false, EPI>(a, s);
  }
  nv_set_error("nv_gemm_lnfold: no LDS-epilogue kernel runs this shape (M=%d N=%d K=%d): ask nv_gemm_lnfold_supported first", a.M, a.N, a.K);
  return NV_ERR_ARG;
}
extern "C" int nv_gemm_lnfold_supported(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0 || (N % 8) || (K % 8)) return 0;
  const GemmPlan p = plan_gemm(false, false, EPI_LNFOLD_STORE, M, N, K, K, K);
  return (p.family == 1 || p.family == 2) ? 1 : 0;
}
extern "C" long nv_ln_fold_stats_floats(int M, int d) { return 2L * M * ((d + 127) / 128); }

static void fold_args(GemmArgs& a, int M, int N, int K, const void* A, long lda, const void* W, long ldw, const float* bias) {
  a.A = (const r16*)A; a.B = (const r16*)W; a.bias = bias; a.aux_in = nullptr; a.aux_out = nullptr; a.aux_out2 = nullptr;
  a.lda = lda; a.ldb = ldw; a.ld_aux_in = 0; a.ld_aux_out = 0; a.ld_aux_out2 = 0;
  a.M = M; a.N = N; a.K = K; a.accumulate = 0; a.alpha = 1.f; a.drop = make_drop(0, 0.f); a.colscale = nullptr;
  a.col_order = (N > M) ? 1 : 0;
}

// out f32 [M, N] = resid + (A W^T + bias)  (vit_3d.py:73-74)  +  out16 = the same rows in the operand format  +  stats [ceil(N / 128)][M][2]
extern "C" int nv_gemm_resid_ln(int M, int N, int K, const void* A, long lda, const void* W, long ldw, const float* bias, const float* resid, long ldr, float* out,
                                long ldo, void* out16, long ldo16, float* stats, void* stream) {
  NV_CHECK_ARG(M > 0 && N > 0 && K > 0 && A && W && bias && resid && out && out16 && stats, "nv_gemm_resid_ln: null pointer / empty problem");
  NV_CHECK_ARG(nv_aligned16(A) && nv_aligned16(W) && nv_aligned16(bias) && nv_aligned16(resid) && nv_aligned16(out) && ((uintptr_t)out16 & 7) == 0 && nv_aligned16(stats) &&
                   (lda % 8) == 0 && (ldw % 8) == 0 && (ldr % 4) == 0 && (ldo % 4) == 0 && (ldo16 % 4) == 0 && (N % 8) == 0 && (K % 8) == 0 && lda >= K && ldw >= K && ldo >= N && ldo16 >= N,
               "nv_gemm_resid_ln: alignment / leading dimensions");
  GemmArgs a;
  fold_args(a, M, N, K, A, lda, W, ldw, bias);
  a.C = out; a.ldc = ldo; a.aux_in = resid; a.ld_aux_in = ldr; a.aux_out = out16; a.ld_aux_out = ldo16; a.aux_out2 = stats;
  NV_DISPATCH_OPERAND(T, return launch_fold_fmt<T, EPI_BIAS_RESID_LN>(a, (hipStream_t)stream));
}

// out16 [M, N] = (gelu)(LayerNorm_K(x) W^T + b) computed as rstd (X16 Wg16^T - mu colsum) + fbias: X16 the UN-normalised rows, Wg16 / colsum / fbias from
// nv_ln_fold_weight, stats from the nv_gemm_resid_ln launch that produced X16 (K = the LayerNorm width)
extern "C" int nv_gemm_lnfold(int gelu, int M, int N, int K, const void* X16, long ldx, const void* Wg16, long ldw, const float* stats, const float* colsum,
                              const float* fbias, float eps, void* out16, long ldo, void* stream) {
  NV_CHECK_ARG(M > 0 && N > 0 && K > 0 && X16 && Wg16 && stats && colsum && fbias && out16, "nv_gemm_lnfold: null pointer / empty problem");
  NV_CHECK_ARG(nv_aligned16(X16) && nv_aligned16(Wg16) && nv_aligned16(colsum) && nv_aligned16(fbias) && nv_aligned16(out16) && (ldx % 8) == 0 && (ldw % 8) == 0 &&
                   (ldo % 4) == 0 && (N % 8) == 0 && (K % 8) == 0 && ldx >= K && ldw >= K && ldo >= N,
               "nv_gemm_lnfold: alignment / leading dimensions");
  GemmArgs a;
  fold_args(a, M, N, K, X16, ldx, Wg16, ldw, fbias);
  a.C = out16; a.ldc = ldo; a.ln_stats = stats; a.ln_cs = colsum; a.ln_tiles = (K + 127) / 128; a.ln_eps = eps;
  if (gelu) NV_DISPATCH_OPERAND(T, return launch_fold_fmt<T, EPI_LNFOLD_GELU>(a, (hipStream_t)stream));
  NV_DISPATCH_OPERAND(T, return launch_fold_fmt<T, EPI_LNFOLD_STORE>(a, (hipStream_t)stream));
}

// Rows of the workgroup tile nv_gemm_bf16 will use for this problem when asked for the fused column-sum epilogue (6): the
// epilogue writes ceil(M / rows) partial rows.  0 = that epilogue is not available for the shape (use epilogue 5 + nv_colsum_bf16).
extern "C" int nv_gemm_tile_rows(int layout, int M, int N, int K, long lda, long ldb) {
  if (layout < 0 || layout > 2 || M <= 0 || N <= 0 || K <= 0) return 0;
  const GemmPlan p = plan_gemm(layout == 2, layout >= 1, EPI_DGELU_COLSUM, M, N, K, lda, ldb);
  return p.family == 0 ? 0 : p.bm;
}

extern "C" int nv_gemm_bf16(int layout, int epi, int M, int N, int K, const void* A, long lda, const void* B, long ldb,
                            void* C, long ldc, const float* bias, const void* aux_in, long ld_aux_in, void* aux_out,
                            long ld_aux_out, int accumulate, float alpha, unsigned long drop_seed, float drop_p, void* stream) {
  NV_CHECK_ARG(M > 0 && N > 0 && K > 0, "nv_gemm_bf16: empty problem M=%d N=%d K=%d", M, N, K);
  NV_CHECK_ARG(A && B && C, "nv_gemm_bf16: null operand");
  NV_CHECK_ARG(nv_aligned16(A) && nv_aligned16(B) && nv_aligned16(C), "nv_gemm_bf16: operands must be 16-byte aligned");
  NV_CHECK_ARG((lda % 8) == 0 && (ldb % 8) == 0 && (ldc % 4) == 0, "nv_gemm_bf16: lda/ldb must be multiples of 8, ldc of 4");
  NV_CHECK_ARG((N % 8) == 0, "nv_gemm_bf16: N=%d must be a multiple of 8", N);
  if (layout == 0) NV_CHECK_ARG((K % 8) == 0 && lda >= K && ldb >= K, "nv_gemm_bf16[NT]: K=%d must be a multiple of 8 and <= lda, ldb", K);
  if (layout == 1) NV_CHECK_ARG((K % 8) == 0 && lda >= K && ldb >= N, "nv_gemm_bf16[NN]: bad K/lda/ldb");
  if (layout == 2) NV_CHECK_ARG((M % 8) == 0 && lda >= M && ldb >= N, "nv_gemm_bf16[TN]: M=%d must be a multiple of 8; lda>=M, ldb>=N", M);
  NV_CHECK_ARG(ldc >= N, "nv_gemm_bf16: ldc < N");
  GemmArgs a;
  a.A = (const r16*)A; a.B = (const r16*)B; a.C = C; a.bias = bias; a.aux_in = aux_in; a.aux_out = aux_out;
  a.aux_out2 = nullptr; a.ld_aux_out2 = 0;
  a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ld_aux_in = ld_aux_in; a.ld_aux_out = ld_aux_out;
  a.M = M; a.N = N; a.K = K; a.accumulate = accumulate; a.alpha = alpha;
  a.drop = make_drop(drop_seed, drop_p);
  a.colscale = nullptr;
  // each XCD (private 4 MiB L2) gets a contiguous run of tiles: run along the dimension of the SMALLER operand so the
  // larger operand's panel is the one that stays resident
  a.col_order = (N > M) ? 1 : 0;
  hipStream_t s = (hipStream_t)stream;
  const bool need_bias = (epi == EPI_BIAS_F32 || epi == EPI_BIAS_GELU || epi == EPI_BIAS_RESID);
  NV_CHECK_ARG(!need_bias || (bias && nv_aligned16(bias)), "nv_gemm_bf16: epilogue %d needs a 16-byte aligned bias", epi);
  NV_CHECK_ARG(epi != EPI_DGELU_COLSUM || (aux_out && ld_aux_out >= N), "nv_gemm_bf16: epilogue 6 needs aux_out = f32 [ceil(M / tile rows), ld_aux_out >= N]");
  NV_CHECK_ARG(!(epi == EPI_BIAS_RESID || epi == EPI_DGELU || epi == EPI_DGELU_COLSUM) || (aux_in && nv_aligned16(aux_in) && (ld_aux_in % 4) == 0),
               "nv_gemm_bf16: epilogue %d needs aux_in", epi);
  NV_CHECK_ARG(epi != EPI_STORE_F32 || !aux_out || (((uintptr_t)aux_out & 7) == 0 && (ld_aux_out % 4) == 0 && ld_aux_out >= N),
               "nv_gemm_bf16: epilogue 1: the optional bf16 mirror (aux_out) must be 8-byte aligned with ld_aux_out >= N, a multiple of 4");
  NV_CHECK_ARG(epi != EPI_BIAS_GELU || !aux_out || (nv_aligned16(aux_out) && (ld_aux_out % 4) == 0),
               "nv_gemm_bf16: EPI_BIAS_GELU: aux_out must be 16-byte aligned (or null: the pre-activation is not stored)");
  switch (layout * 16 + epi) {
    case 0 * 16 + EPI_STORE_BF16: return launch<false, false, EPI_STORE_BF16>(a, s);
    case 0 * 16 + EPI_STORE_F32: return launch<false, false, EPI_STORE_F32>(a, s);
    case 0 * 16 + EPI_BIAS_F32: return launch<false, false, EPI_BIAS_F32>(a, s);
    case 0 * 16 + EPI_BIAS_GELU: return launch<false, false, EPI_BIAS_GELU>(a, s);
    case 0 * 16 + EPI_BIAS_RESID: return launch<false, false, EPI_BIAS_RESID>(a, s);
    case 1 * 16 + EPI_STORE_BF16: return launch<false, true, EPI_STORE_BF16>(a, s);
    case 1 * 16 + EPI_STORE_F32: return launch<false, true, EPI_STORE_F32>(a, s);
    case 1 * 16 + EPI_DGELU: return launch<false, true, EPI_DGELU>(a, s);
    case 1 * 16 + EPI_DGELU_COLSUM: return launch<false, true, EPI_DGELU_COLSUM>(a, s);
    case 2 * 16 + EPI_STORE_F32: return launch<true, true, EPI_STORE_F32>(a, s);
    default: break;
  }
  nv_set_error("nv_gemm_bf16: unsupported layout/epilogue combination (%d, %d)", layout, epi);
  return NV_ERR_ARG;
}

// fp8 (OCP e4m3) operands, NT layout: C = epilogue((A8 . B8^T) * colscale[n]).  Eight-wave 256 x 128 kernel only (the path exists for
// the large-M inference shapes of ViT3D-large); K % 128 == 0.
static int gemm_f8_impl(int epi, int M, int N, int K, const void* A8, long lda, const void* B8, long ldb, void* C, long ldc, const float* colscale,
                        const float* bias, const void* aux_in, long ld_aux_in, float out_scale, void* u16, long ldu16, void* h16, long ldh16, void* stream,
                        unsigned long drop_seed = 0, float drop_p = 0.f);

extern "C" int nv_gemm_f8(int epi, int M, int N, int K, const void* A8, long lda, const void* B8, long ldb, void* C, long ldc, const float* colscale,
                          const float* bias, const void* aux_in, long ld_aux_in, float out_scale, void* stream) {
  NV_CHECK_ARG(epi != EPI_BIAS_GELU_F8T, "nv_gemm_f8: epilogue 8 has two more outputs: call nv_gemm_f8_gelu_train");
  return gemm_f8_impl(epi, M, N, K, A8, lda, B8, ldb, C, ldc, colscale, bias, aux_in, ld_aux_in, out_scale, nullptr, 0, nullptr, 0, stream);
}

// FC1 of a TRAINING forward on fp8 operands (vit_3d.py:19-20): h8 (e4m3) = sat(gelu(u) * out_scale) feeds the fp8 FC2; u16 = the
// pre-activation (optional) and h16 = gelu(u), both bf16, are what the bf16 backward pass reads (GELU' and the FC2 weight gradient).
extern "C" int nv_gemm_f8_gelu_train(int M, int N, int K, const void* A8, long lda, const void* B8, long ldb, const float* colscale, const float* bias,
                                     float out_scale, void* h8, long ldh8, void* h16, long ldh16, void* u16, long ldu16, unsigned long drop_seed,
                                     float drop_p, void* stream) {
  NV_CHECK_ARG(h16 && nv_aligned16(h16) && (ldh16 % 4) == 0 && ldh16 >= N && (!u16 || (nv_aligned16(u16) && (ldu16 % 4) == 0 && ldu16 >= N)),
               "nv_gemm_f8_gelu_train: h16 (required) / u16 (optional) must be 16-byte aligned bf16 [M, ld >= N], ld a multiple of 4");
  return gemm_f8_impl(EPI_BIAS_GELU_F8T, M, N, K, A8, lda, B8, ldb, h8, ldh8, colscale, bias, nullptr, 0, out_scale, u16, ldu16, h16, ldh16, stream, drop_seed, drop_p);
}

// nv_gemm_f8 epilogue 4 (f32 = aux_in + (acc * colscale + bias) * mask) with the nn.Dropout of a training forward (vit_3d.py:23): FC2 of
// nv_vit_forward_fp8_train; the mask is the one nv_gemm_bf16's epilogue 4 applies for the same (seed, p) - the backward pass recomputes it.
extern "C" int nv_gemm_f8_resid_drop(int M, int N, int K, const void* A8, long lda, const void* B8, long ldb, void* C, long ldc, const float* colscale,
                                     const float* bias, const void* aux_in, long ld_aux_in, unsigned long drop_seed, float drop_p, void* stream) {
  return gemm_f8_impl(EPI_BIAS_RESID, M, N, K, A8, lda, B8, ldb, C, ldc, colscale, bias, aux_in, ld_aux_in, 1.f, nullptr, 0, nullptr, 0, stream, drop_seed, drop_p);
}

static int gemm_f8_impl(int epi, int M, int N, int K, const void* A8, long lda, const void* B8, long ldb, void* C, long ldc, const float* colscale,
                        const float* bias, const void* aux_in, long ld_aux_in, float out_scale, void* u16, long ldu16, void* h16, long ldh16, void* stream,
                        unsigned long drop_seed, float drop_p) {
  NV_CHECK_ARG(nv_operand_format() == NV_OPERAND_BF16, "nv_gemm_f8: the fp8 path is built beside bf16 operands (nv_set_operand_format(NV_OPERAND_BF16))");
  NV_CHECK_ARG(M > 0 && N > 0 && K > 0 && A8 && B8 && C && colscale, "nv_gemm_f8: null operand / empty problem");
  NV_CHECK_ARG((K % 128) == 0 && (N % 8) == 0 && (lda % 16) == 0 && (ldb % 16) == 0 && lda >= K && ldb >= K && ldc >= N && (ldc % 4) == 0,
               "nv_gemm_f8: K must be a multiple of 128, N of 8, lda / ldb of 16, ldc of 4");
  NV_CHECK_ARG(nv_aligned16(A8) && nv_aligned16(B8) && nv_aligned16(C) && nv_aligned16(colscale), "nv_gemm_f8: 16-byte alignment");
  NV_CHECK_ARG((long)M * lda < (1L << 31) && (long)N * ldb < (1L << 31), "nv_gemm_f8: operand too large for 32-bit offsets");
  const bool need_bias = (epi == EPI_BIAS_RESID || epi == EPI_BIAS_GELU_F8 || epi == EPI_BIAS_GELU_F8T);
  NV_CHECK_ARG(!need_bias || (bias && nv_aligned16(bias)), "nv_gemm_f8: epilogue %d needs a bias", epi);
  NV_CHECK_ARG(epi != EPI_BIAS_RESID || (aux_in && nv_aligned16(aux_in) && (ld_aux_in % 4) == 0), "nv_gemm_f8: epilogue 4 needs aux_in");
  GemmArgs a;
  a.A = (const r16*)A8; a.B = (const r16*)B8; a.C = C; a.bias = bias; a.aux_in = aux_in; a.aux_out = u16;
  a.aux_out2 = h16; a.ld_aux_out2 = ldh16;
  a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ld_aux_in = ld_aux_in; a.ld_aux_out = ldu16;
  a.M = M; a.N = N; a.K = K; a.accumulate = 0; a.alpha = out_scale;
  a.drop = make_drop(drop_seed, drop_p);
  a.colscale = colscale;
  a.col_order = 0;
  const long tpq = (long)((M + PQ_BM - 1) / PQ_BM) * ((N + PQ_BN - 1) / PQ_BN);
  if (g_tile_override == 9 || (g_tile_override == 0 && tpq >= g_pq_min_tiles)) return launch_pq_f8(epi, a, (hipStream_t)stream);
  return launch_pp_f8(epi, a, (hipStream_t)stream);
}

// Grouped weight-gradient GEMMs (layout 2 / TN, fp32 store or accumulate): see gemm_ws_grouped_kernel.
static int grouped_tn_impl(int count, const nv_gemm_problem* pr, const nv_adamw_arena* opt, void* stream) {
  NV_CHECK_ARG(pr && count >= 1 && count <= GROUP_MAX, "nv_gemm_bf16_grouped: 1..%d problems", GROUP_MAX);
  AdamArgs adam = {};
  if (opt) {
    NV_CHECK_ARG(opt->struct_size == (int)sizeof(nv_adamw_arena), "nv_gemm_bf16_grouped_adamw: nv_adamw_arena.struct_size = %d, this library expects %d (ABI revision %d)",
                 opt->struct_size, (int)sizeof(nv_adamw_arena), NV_ABI_VERSION);
    NV_CHECK_ARG(opt->params && opt->grads && opt->adam_m && opt->adam_v && opt->params16 && opt->step >= 1, "nv_gemm_bf16_grouped_adamw: null arena or step < 1");
    NV_CHECK_ARG(nv_aligned16(opt->params) && nv_aligned16(opt->grads) && nv_aligned16(opt->adam_m) && nv_aligned16(opt->adam_v) && nv_aligned16(opt->params16),
                 "nv_gemm_bf16_grouped_adamw: arenas must be 16-byte aligned");
    adam = make_adam_args(opt->step, opt->lr, opt->beta1, opt->beta2, opt->eps, opt->weight_decay, opt->grad_scale);
  }
  constexpr int BM = 64, BN = 128;
  GemmGroup G;
  G.count = count;
  int tiles = 0;
  double flops = 0.0;
  for (int i = 0; i < count; ++i) {
    const nv_gemm_problem& q = pr[i];
    NV_CHECK_ARG(q.M > 0 && q.N > 0 && q.K > 0 && q.A && q.B && q.C, "nv_gemm_bf16_grouped: empty problem %d", i);
    NV_CHECK_ARG(nv_aligned16(q.A) && nv_aligned16(q.B) && nv_aligned16(q.C) && (q.lda % 8) == 0 && (q.ldb % 8) == 0 && (q.ldc % 4) == 0 &&
                     (q.M % 8) == 0 && (q.N % 8) == 0 && q.lda >= q.M && q.ldb >= q.N && q.ldc >= q.N,
                 "nv_gemm_bf16_grouped: problem %d: alignment / leading dimensions", i);
    NV_CHECK_ARG((long)q.K * q.lda < (1L << 30) && (long)q.K * q.ldb < (1L << 30), "nv_gemm_bf16_grouped: problem %d too large for 32-bit offsets", i);
    GemmArgs& a = G.p[i];
    NV_CHECK_ARG(!q.C16 || (((uintptr_t)q.C16 & 7) == 0 && (q.ldc16 % 4) == 0 && q.ldc16 >= q.N), "nv_gemm_bf16_grouped: problem %d: bf16 mirror alignment / leading dimension", i);
    a.A = (const r16*)q.A; a.B = (const r16*)q.B; a.C = q.C; a.bias = nullptr; a.aux_in = nullptr; a.aux_out = q.C16;
    a.aux_out2 = nullptr; a.ld_aux_out2 = 0;
    a.lda = q.lda; a.ldb = q.ldb; a.ldc = q.ldc; a.ld_aux_in = 0; a.ld_aux_out = q.ldc16;
    a.M = q.M; a.N = q.N; a.K = q.K; a.accumulate = q.accumulate; a.alpha = 1.f;
    a.drop = make_drop(0, 0.f);
    a.colscale = nullptr;
    a.col_order = (q.N > q.M) ? 1 : 0;
    if (opt) {
      const long off = (const float*)q.C - opt->grads;
      NV_CHECK_ARG(!q.accumulate && !q.C16 && off >= 0 && (off % 4) == 0, "nv_gemm_bf16_grouped_adamw: problem %d: C must lie in opt->grads (16-byte aligned offset), accumulate = 0, no C16", i);
      a.opt.p = opt->params + off; a.opt.m = opt->adam_m + off; a.opt.v = opt->adam_v + off; a.opt.p16 = (r16*)opt->params16 + off;
      a.opt.a = adam; a.opt.keep_grad = opt->keep_grads ? 1 : 0;
    }
    tiles += ((q.M + BM - 1) / BM) * ((q.N + BN - 1) / BN);
    G.tile_end[i] = tiles;
    flops += 2.0 * q.M * q.N * q.K;
  }
  for (int i = count; i < GROUP_MAX; ++i) { G.p[i] = G.p[0]; G.tile_end[i] = tiles; }
  if (opt || g_tile_override == 4 || (g_tile_override == 0 && g_pp_grouped)) {      // (the fused update exists on this kernel only)
    // 256 x 128 ping-pong tiles: the four problems of a ViT3D-base layer are 216 tiles - one round on 256 CUs
    int tpp = 0;
    for (int i = 0; i < count; ++i) {
      tpp += ((pr[i].M + PP_BM - 1) / PP_BM) * ((pr[i].N + PP_BN - 1) / PP_BN);
      G.tile_end[i] = tpp;
    }
    for (int i = count; i < GROUP_MAX; ++i) G.tile_end[i] = tpp;
    return launch_pp_grouped_tn(G, tpp, flops, (hipStream_t)stream, opt != nullptr);
  }
  constexpr int LDS = 3 * (BM + BN) * BK * 2;
  const bool fp16 = nv_operand_format() == NV_OPERAND_FP16;
  auto kern = fp16 ? gemm_ws_grouped_kernel<fp16_t, BM, BN, 3, 1, true, true, EPI_STORE_F32> : gemm_ws_grouped_kernel<bf16_t, BM, BN, 3, 1, true, true, EPI_STORE_F32>;
  static bool attr_set[2] = {false, false};
  if (!attr_set[fp16]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set[fp16] = true;
  }
  hipStream_t s = (hipStream_t)stream;
  const int slot = nv_prof_begin(2, flops, s);
  if (slot >= 0) {
    double bytes = 0.0;
    for (int i = 0; i < G.count; ++i) bytes += gemm_algo_bytes(G.p[i], EPI_STORE_F32, 2);
    nv_prof_bytes(slot, bytes);
  }
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(WS_THREADS), LDS, s, G);
  nv_prof_end(slot, s);
  NV_CHECK_LAUNCH("nv_gemm_bf16_grouped");
  return NV_OK;
}

extern "C" int nv_gemm_bf16_grouped(int layout, int epi, int count, const nv_gemm_problem* pr, void* stream) {
  NV_CHECK_ARG(layout == 2 && epi == EPI_STORE_F32, "nv_gemm_bf16_grouped: only layout 2 (TN) with epilogue 1 (fp32 store) is provided");
  return grouped_tn_impl(count, pr, nullptr, stream);
}

extern "C" int nv_gemm_bf16_grouped_adamw(int count, const nv_gemm_problem* pr, const nv_adamw_arena* opt, void* stream) {
  NV_CHECK_ARG(opt, "nv_gemm_bf16_grouped_adamw: null optimizer block");
  return grouped_tn_impl(count, pr, opt, stream);
}
