// DIAGNOSTIC, not part of the library (built by tools/qw_probe.sh, run by tools/qw_probe.py): a 256 x 256-tile NT GEMM on FOUR
// waves - one per SIMD, each owning a 128 x 128 quadrant (256 accumulator AGPRs) - with in-kernel clock stamps and ablation modes.
// It answers what bounds the large-tile kernels of csrc/gemm_pp.hip / gemm_pq.hip (1.05-1.16 PFLOP/s on 8192^2 x 4096):
//
//   * LDS traffic is not it.  This kernel needs 192 KiB of LDS traffic per 2048 matrix cycles against gemm_pp's 352 and gemm_pq's
//     256 (a 128 x 128 wave tile reads half of a 64 x 64 tile's operand bytes per FLOP) and, with its operand staging removed,
//     keeps the matrix pipe 90 % busy (1132 cycles per 64 MFMAs of 16 cycles) with ONE wave per SIMD.
//   * Operand STAGING is: every way of putting 1 KiB into LDS holds the issuing wave for 80-90 cycles - `buffer_load ... lds`
//     ~90 (first version of this file), `buffer_load` to registers 25 + `ds_write_b128` 54 (this version) - during which an
//     in-order wave issues no MFMA: +540-730 cycles on the 1132 of a 32-deep stage, whatever the ring depth (3, 4, 5 slots), with
//     the loads killed (no memory traffic) or real, on 32 or on 256 CUs.  gemm_pp.hip hides that behind loader waves, which a
//     400-register wave leaves no room for (registers are allocated per kernel, not per wave).
//   * And the chip is POWER limited on top: the more of the pipe a variant keeps busy, the lower the clock it is given (full kernel
//     55 % busy at 2183 MHz, 61 % at 2043, 66 % at 1924 - same 1.12-1.16 PFLOP/s; variants without memory traffic run at
//     2370-2400 MHz): profiles/r03_gemm_qw_power_probe.log.
// Correct (tools/qw_probe.py checks it against the library's result), never faster than gemm_pq.hip, slower on the model's
// K = 768 / 1024 shapes (one wave per SIMD: nothing overlaps a tile's prologue and epilogue) - so it is not shipped.
//
// Structure: a ring stage is 32 deep (A 256 x 32 + B 256 x 32 = 32 KiB): ONE ds_read_b128 per 16-row block is the whole K of a
// 16 x 16 x 32 MFMA.  The fragments of stage t live in registers; beside its 64 MFMAs the wave stores its eight pieces of stage
// t+2 (loaded two steps earlier) into the slot stage t-1 left, reloads those registers with stage t+4 and reads the fragments of
// stage t+1.  One raw s_barrier per stage.  Image of a stage: [512 rows][64 bytes], chunk c of row r at
// r * 64 + ((c ^ F[(r >> 2) & 3]) << 4), F = {0,3,2,1}: the four 16-lane groups of a ds_read_b128 each touch 16 different 16-byte
// columns.  K % 64 == 0; ragged M / N come back as zeros from the buffer bounds; stages beyond K are "killed" (offsets pushed out
// of range: zeros, no traffic) so that every step runs the same instructions.
#include "gemm_common.h"
constexpr int QW_BM = 256, QW_BN = 256;

namespace {

constexpr int QW_THREADS = 256;
constexpr int QW_BK = 32;
constexpr int QW_S = 3;                                      // ring slots: being read, being written, free next
constexpr int QW_HALF = QW_BM * QW_BK * 2;                   // bytes of the A (or B) part of a stage: 16 KiB
constexpr int QW_STAGE = 2 * QW_HALF;
constexpr int QW_LDS = QW_S * QW_STAGE;                      // 160 KiB
constexpr int QW_PIECES = 8;                                 // DMA pieces per wave per stage (4 of A, 4 of B)
constexpr int QW_GM = 4;                                     // tile rows per group of the tile order (gemm_pq.hip)
constexpr int QW_LDS_EPI = 128 * (QW_BN * 4 + 16) + 4 * QW_BN * 4;       // epilogue pass: fp32 [128][256] tile + column-sum scratch
static_assert(QW_LDS <= 160 * 1024 && 128 * cpitch<QW_BN>() + colsum_scratch_bytes<128, QW_BN, QW_THREADS>() <= QW_LDS_EPI, "LDS budget");

#define QW_SB __builtin_amdgcn_sched_barrier(0)

#ifdef QW_PROBE
// diagnostic build only (tools/qw_probe.sh): shader-clock and real-time (100 MHz) stamps around the K loop of every workgroup -
// the clock the chip actually runs this kernel at (MI355X_MICROARCH.md, profiling recipe (6)).  QW_DBG: 1 = no DMA inside the loop,
// 2 = no fragment reads inside the loop either (matrix instructions only); results are wrong by design.
__device__ unsigned long long g_qw_probe[4 * 4096];
#ifndef QW_DBG
#define QW_DBG 0
#endif
#endif

typedef unsigned qw_u32x4 __attribute__((ext_vector_type(4)));

template <int EPI>
__global__ __launch_bounds__(QW_THREADS, 1) void gemm_qw_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BM = QW_BM, BN = QW_BN, MI = 8, NI = 8;
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, gq = lane >> 4;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wid >> 1, wn = wid & 1;                    // quadrant: rows grp * 128, columns wn * 128
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM;
  const int per_group = QW_GM * tiles_n;
  const int gid = bid / per_group, first_m = gid * QW_GM;
  const int gsz = (tiles_m - first_m < QW_GM) ? tiles_m - first_m : QW_GM;
  const int rin = bid - gid * per_group;
  const int m0 = (first_m + rin % gsz) * BM, n0 = (rin / gsz) * BN;
  const int nk = g.K / QW_BK;

  // ---- staging: wave w moves pieces w, w + 4, w + 8, w + 12 (16 rows x 64 bytes each) of the A part and of the B part of a stage;
  // lane l of a piece lands at piece + 16 l, so the image's swizzle is applied to the SOURCE chunk
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, (unsigned)((((long)g.M - 1) * g.lda + g.K) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)g.B, 0, (unsigned)((((long)g.N - 1) * g.ldb + g.K) * 2), 0x00020000);
  int vo[QW_PIECES];
  {
    const int prow = lane >> 2, src_chunk = (lane & 3) ^ ((4 - ((lane >> 4) & 3)) & 3);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = 16 * (wid + 4 * q) + prow;
      vo[q] = (int)(((long)(m0 + row) * g.lda) * 2) + (src_chunk << 4);
      vo[4 + q] = (int)(((long)(n0 + row) * g.ldb) * 2) + (src_chunk << 4);
    }
  }
  // piece i of this wave's eight (0-3: A, 4-7: B) of stage t: global -> registers.  `kill` (0 or 2^30, wave-uniform) pushes the
  // offsets of a stage beyond K out of the buffer's range: the load returns zeros without touching memory, and they are stored
  // into a slot nobody reads again - every step runs the same instructions from the first to the last.
  auto load_piece = [&](int i, int t, int kill) -> qw_u32x4 {
    return __builtin_amdgcn_raw_buffer_load_b128(i < 4 ? rA : rB, vo[i] + kill, t * (QW_BK * 2), 0);
  };
  const int st_off = ((lane << 4)) + wid * 1024;             // + (A: 0 | B: QW_HALF) + 4096 q
  auto store_piece = [&](int i, char* slot, qw_u32x4 v) {
#ifdef QW_ST64
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    char* p = slot + ((i >> 2) ? QW_HALF : 0) + (i & 3) * 4096 + st_off;
    *reinterpret_cast<u32x2*>(p) = u32x2{v[0], v[1]};
    asm volatile("" ::: "memory");
    *reinterpret_cast<u32x2*>(p + 8) = u32x2{v[2], v[3]};
#else
    *reinterpret_cast<qw_u32x4*>(slot + ((i >> 2) ? QW_HALF : 0) + (i & 3) * 4096 + st_off) = v;
#endif
  };

  // ---- fragments: block i of this wave's A rows / B rows = one ds_read_b128 at a fixed per-lane offset + i KiB
  const int lane_off = r * 64 + ((gq ^ ((4 - ((r >> 2) & 3)) & 3)) << 4);
  const int a_base = grp * 8192 + lane_off, b_base = QW_HALF + wn * 8192 + lane_off;
  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 fa0[MI], fb0[NI], fa1[MI], fb1[NI];
  qw_u32x4 rg0[QW_PIECES], rg1[QW_PIECES];                  // two stages on their way from global memory

  // ---- prologue: stages 0, 1 into ring slots 0, 1; stages 2, 3 in flight in the registers; fragments of stage 0
#pragma unroll
  for (int i = 0; i < QW_PIECES; ++i) { rg0[i] = load_piece(i, 0, 0); rg1[i] = load_piece(i, 1, 1 < nk ? 0 : (1 << 30)); }
#pragma unroll
  for (int i = 0; i < QW_PIECES; ++i) { store_piece(i, smem, rg0[i]); store_piece(i, smem + QW_STAGE, rg1[i]); }
#pragma unroll
  for (int i = 0; i < QW_PIECES; ++i) { rg0[i] = load_piece(i, 2, 2 < nk ? 0 : (1 << 30)); rg1[i] = load_piece(i, 3, 3 < nk ? 0 : (1 << 30)); }
  __builtin_amdgcn_s_waitcnt(0xc07f);                       // lgkmcnt(0): this wave's stores are in LDS
  __builtin_amdgcn_s_barrier();
  QW_SB;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    fa0[i] = *reinterpret_cast<const bf16x8*>(smem + a_base + i * 1024);
    fb0[i] = *reinterpret_cast<const bf16x8*>(smem + b_base + i * 1024);
#if defined(QW_PROBE) && QW_DBG == 2
    fa1[i] = fa0[i]; fb1[i] = fb0[i];
#endif
  }

  // Step T: 64 MFMAs on (CA, CB) = stage T.  Beside them, per block row i (8 MFMAs): store piece i of stage T+2 (RG, loaded two
  // steps ago) into the slot stage T-1 left, reload RG with piece i of stage T+4, read the fragments of block i of stage T+1 into
  // (NA, NB).  Straight-line code, ONE loop body of two steps (with branches in it, or with peeled tail loops, the register
  // allocator put the FRAGMENTS into AGPRs and shuttled the accumulators through v_accvgpr moves and scratch).  The reads of the
  // last step fetch a stale slot and are not used.
#if defined(QW_PROBE) && QW_DBG == 2
#define QW_READS(NA, NB)
#else
#define QW_READS(NA, NB)                                                                                               \
      NA[i] = *reinterpret_cast<const bf16x8*>(rd + a_base + i * 1024);                                                \
      NB[i] = *reinterpret_cast<const bf16x8*>(rd + b_base + i * 1024);
#endif
#if defined(QW_PROBE) && QW_DBG == 5                      /* stores only (the registers are never reloaded) */
#define QW_FEED(RG, T) store_piece(i, wr, RG[i]); (void)kill;
#elif defined(QW_PROBE) && QW_DBG == 6                    /* loads only (never stored; kept alive by the asm below) */
#define QW_FEED(RG, T) asm volatile("" ::"v"(RG[i])); RG[i] = load_piece(i, (T) + 4, kill);
#elif defined(QW_PROBE) && QW_DBG == 3                    /* every load killed: issued, no memory traffic */
#define QW_FEED(RG, T) store_piece(i, wr, RG[i]); RG[i] = load_piece(i, (T) + 4, 1 << 30);
#elif defined(QW_PROBE) && QW_DBG >= 1 && QW_DBG != 4
#define QW_FEED(RG, T) (void)kill;
#else
#define QW_FEED(RG, T) store_piece(i, wr, RG[i]); RG[i] = load_piece(i, (T) + 4, kill);
#endif
#if defined(QW_PROBE) && QW_DBG == 4                      /* data movement only: staging + fragment reads, one MFMA per block row */
#define QW_MFMAS(CA, CB) acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(CB[i], CA[i], acc[i][0], 0, 0, 0);
#else
#define QW_MFMAS(CA, CB)                                                                                               \
      _Pragma("unroll") for (int j = 0; j < NI; ++j)                                                                   \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(CB[j], CA[i], acc[i][j], 0, 0, 0);
#endif
#define QW_STEP(CA, CB, NA, NB, RG, T)                                                                                 \
  {                                                                                                                    \
    __builtin_amdgcn_s_waitcnt(0xc07f);                     /* lgkmcnt(0): last step's fragment reads and stores are done */ \
    __builtin_amdgcn_s_barrier();                                                                                      \
    QW_SB;                                                                                                             \
    const char* rd = smem + slot_rd * QW_STAGE;                                                                        \
    char* wr = smem + slot_wr * QW_STAGE;                                                                              \
    const int kill = ((T) + 4 < nk) ? 0 : (1 << 30);                                                                   \
    _Pragma("unroll") for (int i = 0; i < MI; ++i) {                                                                   \
      QW_FEED(RG, T)                                                                                                   \
      QW_READS(NA, NB)                                                                                                 \
      QW_MFMAS(CA, CB)                                                                                                 \
      QW_SB;                                                                                                           \
    }                                                                                                                  \
    slot_rd = slot_wr;                                                                                                 \
    slot_wr = (slot_wr + 1 == QW_S) ? 0 : slot_wr + 1;                                                                 \
  }

  int slot_rd = 1, slot_wr = 2;
#ifdef QW_PROBE
  const unsigned long long pc0 = __builtin_amdgcn_s_memtime(), pr0 = __builtin_amdgcn_s_memrealtime();
  __builtin_amdgcn_s_waitcnt(0xc07f);
#endif
  for (int t = 0; t < nk; t += 2) {                               // K % 64 == 0: an even number of stages
    QW_STEP(fa0, fb0, fa1, fb1, rg0, t)
    QW_STEP(fa1, fb1, fa0, fb0, rg1, t + 1)
  }
#undef QW_STEP
#undef QW_READS
#undef QW_FEED
#undef QW_MFMAS
#ifdef QW_PROBE
  {
    const unsigned long long pc1 = __builtin_amdgcn_s_memtime(), pr1 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    if (tid == 0 && blockIdx.x < 4096) {
      g_qw_probe[4 * blockIdx.x] = pc1 - pc0; g_qw_probe[4 * blockIdx.x + 1] = pr1 - pr0;
      g_qw_probe[4 * blockIdx.x + 2] = pr0; g_qw_probe[4 * blockIdx.x + 3] = nk;
    }
  }
#endif
  asm volatile("" ::"v"(rg0[0]), "v"(rg1[0]));              // (the loads of the killed tail stages are simply dropped)

  // ---- fused epilogue, 128 rows (one pair of waves) at a time through an fp32 [128][256] LDS tile, all four waves
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();                           // the ring / the previous pass's tile is dead
    if (grp == pass) park_acc<MI, NI, BN>(acc, smem, 0, wn * 128, lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    epilogue_lds<EPI, 128, BN, QW_THREADS>(smem, g, m0 + 128 * pass, n0, tid);
  }
}

}  // namespace

// C[M, N] (bf16) = A[M, K] B[N, K]^T, both bf16, K % 64 == 0
extern "C" int nv_debug_gemm_qw(int M, int N, int K, const void* A, long lda, const void* B, long ldb, void* C, long ldc, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || (K % 64) || (N % 8) || (lda % 8) || (ldb % 8) || (ldc % 4)) return -1;
  GemmArgs a{};
  a.A = (const bf16*)A; a.B = (const bf16*)B; a.C = C; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.M = M; a.N = N; a.K = K; a.alpha = 1.f;
  a.drop = make_drop(0, 0.f);
  const int tiles = ((M + QW_BM - 1) / QW_BM) * ((N + QW_BN - 1) / QW_BN);
  auto kern = gemm_qw_kernel<EPI_STORE_BF16>;
  constexpr int lds = QW_LDS > QW_LDS_EPI ? QW_LDS : QW_LDS_EPI;
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr_set = true; }
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(QW_THREADS), lds, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int nv_debug_qw_probe(unsigned long long* out, int blocks) {
#ifdef QW_PROBE
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_qw_probe), (size_t)blocks * 4 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
#else
  (void)out; (void)blocks; return -1;
#endif
}
