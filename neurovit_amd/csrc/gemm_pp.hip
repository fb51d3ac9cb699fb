// 256 x 128 x 64 tiles: eight compute waves + four LDS-DMA loader waves per workgroup - bf16 and fp8 MFMA GEMM for gfx950.
//
// Why a second kernel family.  The warp-specialised 128 x 128 / 64 x 128 kernels of gemm.hip give each of FOUR consumer waves a
// 64 x 64 tile: per 64-deep K step a workgroup pulls 32 KiB through the CU's L2 -> LDS path (~64 B/clk) for 512 MFMA cycles per
// SIMD - the path's limit - and they top out at ~0.86 PFLOP/s (profiles/r01_gemm_vs_vendor_blas.log).  A 256 x 128 tile needs
// 48 KiB per 1024 MFMA cycles and, with EIGHT waves computing 64 x 64 sub-tiles, the same fragment bytes per MFMA.
//
// Structure:
//   * 768 threads: waves 0-7 compute (4 (M) x 2 (N), wave tile 64 x 64 = 4 x 4 MFMA tiles of 16 x 16 x 32, 64 accumulator VGPRs,
//     <= 168 VGPRs in all: three waves per SIMD); waves 8-11, one per SIMD, only issue LDS-DMA (buffer_load ... lds, 12 x 1 KiB
//     pieces per K tile each) behind counted s_waitcnt vmcnt;
//   * three-stage LDS ring (3 x 48 KiB), ONE raw s_barrier per K tile executed by all twelve waves: after barrier k tile k is
//     complete (the loaders waited for it) and the slot of tile k-1 is free; the loaders then issue tile k+2 and wait for k+1;
//   * the two compute waves of a SIMD (w and w + 4) run HALF A TILE out of step: right after the barrier group 1 still has 16 MFMAs
//     of tile k-1 (fragments held in registers) while group 0's fragment reads are in flight, and group 1 reads while group 0
//     computes - the matrix pipe never waits for LDS.  (First built as the CDNA4 guide's phase-by-phase ping-pong - two barriers
//     per 16 MFMAs, DMA issued by the compute waves: 0.93-1.05 PFLOP/s.  Timing ablations showed the loop running at 1.43-1.48
//     without DMA and 1.46-1.51 MFMA-only, the fragment reads free, and the DMA's cost on the CU independent of where the data
//     comes from or how far ahead it is issued; what paid was fewer barriers and DMA off the compute waves: 1.05-1.10.);
//   * L2-aware tile order (groups of four tile rows, column by column: a 4 x 8 block of tiles per XCD wave, 82 % L2 hits);
//   * operand images, swizzles, transposed reads (ds_read_b64_tr_b16) and the fused epilogues are those of gemm.hip
//     (gemm_common.h); 128-wide sub-images make every layout (NT / NN / TN) a composition of the same 16 KiB pieces;
//   * epilogue: accumulators parked as an fp32 [256][128] tile in the (now idle) ring, then all twelve waves run the row-wise
//     fused epilogue (gemm_common.h::epilogue_lds);
//   * F8: OCP e4m3 operands on v_mfma_scale_f32_16x16x128_f8f6f4 (see gemm_pp_body).
#include <type_traits>

#include "gemm_common.h"
#include "gemm_pp.h"

namespace {

constexpr int PP_CWAVES = 8;                       // compute waves (two staggered groups of four)
constexpr int PP_LWAVES = 4;                       // loader waves (one per SIMD)
constexpr int PP_THREADS = 64 * (PP_CWAVES + PP_LWAVES);
constexpr int PP_SUB = 16384;   // bytes of one 128-wide sub-image of a 64-deep K tile
constexpr int PP_S = 3;         // ring stages
constexpr int PP_GM = 4;        // tile rows per group of the tile order (4 x 256 rows beside 8 x 128 columns per XCD wave)

// per-lane DMA source offsets (bytes) of pieces I = lw, lw + 4, lw + 8, lw + 12 of one 128-wide sub-image (four loader waves);
// rc0 = first matrix row (K-contiguous operand) / first matrix column (K-strided operand) of the sub-image
// W32: the K-contiguous image for 32-row fragments (v_mfma_f32_32x32x16_bf16): chunk XOR ((row >> 1) & 7) instead of (row & 7).  A
// ds_read_b128 serves 16 lanes per LDS cycle - lanes {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 - which for a 32-row
// fragment (lane l: row l & 31, chunk 2 ks + (l >> 5)) are 16 different rows at ONE chunk: with 128-byte rows the even rows share one
// half of the 256-byte bank line and the odd rows the other, so the eight even (odd) rows of a group need eight different chunk
// positions - (row >> 1) & 7 gives {0,1,6,7,2,3,4,5} and {2,3,4,5,0,1,6,7} for the two groups (row & 7 puts rows 12 and 20 on one slot).
__device__ __forceinline__ int img128w_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <bool T, int ESZ = 2, bool W32 = false>
__device__ __forceinline__ void pp_offsets(long ld, int rc0, int lw, int lane, int (&voff)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int I = lw + 4 * i;
    if constexpr (!T) {
      const int row = I * 8 + (lane >> 3);
      const int ch = W32 ? ((lane & 7) ^ ((row >> 1) & 7)) : ((lane & 7) ^ ((lane >> 3) & 7));   // img128w_off / img128_off inverse
      voff[i] = (int)(((long)(rc0 + row)) * ld * ESZ + (ch << 4));
    } else {
      static_assert(ESZ == 2, "K-strided operands are bf16 only");
      const int krow = I * 4 + (lane >> 4);
      const int ch = (lane & 15) ^ (((krow & 3) << 2) | ((krow >> 2) & 3));                // img256_off inverse
      voff[i] = (int)(((long)krow * ld + rc0 + (ch << 3)) * 2);
    }
  }
}

typedef int i32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ i32x8 f8cat(r16x8 lo, r16x8 hi) {      // 2 x 16 bytes -> the 32-byte fp8 operand of one lane
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  const i32x4 a = __builtin_bit_cast(i32x4, lo), b = __builtin_bit_cast(i32x4, hi);
  return i32x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

#define PP_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define PP_SB __builtin_amdgcn_sched_barrier(0)
#define PP_FENCE asm volatile("" ::: "memory")

// One s_barrier per 64-deep K tile, executed by all twelve waves: after barrier k, tile k is complete in LDS (the loaders
// waited for it) and the slot of tile k-1 is free (every fragment read of it has returned).
//   loaders:  B_k | issue tile k+2 (into the slot of tile k-1), wait for tile k+1 | B_k+1
//   group 0:  B_k | read all fragments of tile k, 32 MFMAs | B_k+1
//   group 1:  B_k | 16 MFMAs (lower half of tile k-1, fragments held in registers), read B + upper-half A fragments of tile k,
//                   16 MFMAs (upper half of tile k), read the lower-half A fragments of tile k | B_k+1
// The two waves of a SIMD (w, w + 4) are thereby half a tile out of step: right after a barrier group 1 still has matrix work
// while group 0's reads are in flight, and group 1 reads while group 0 computes - the matrix pipe never waits for LDS, with a
// quarter of the barriers of a phase-by-phase ping-pong (measured: the barrier-per-phase form lost 25 % of the MFMA-only rate
// to barrier overhead and to loaders that arrive late at a 256-cycle barrier interval).
// F8: both operands are OCP e4m3 bytes, K-contiguous; a ring stage is 128 elements deep (the same 128-byte rows, images, DMA pieces
// and fragment reads as bf16) and each 16 x 16 output block takes ONE v_mfma_scale_f32_16x16x128_f8f6f4 per stage (unit block
// scales; 32 cycles, i.e. the cycles of the two bf16 MFMAs it replaces at twice their K: double the FLOPs per byte and per cycle).
// The k order inside a lane's 32 bytes is [chunk g | chunk 4 + g] of the row for BOTH operands - a dot product does not care.
// W32: the same structure on v_mfma_f32_32x32x16_bf16 (NT / bf16 only): a 64 x 64 wave tile is 2 x 2 blocks of 32 x 32, four 16-deep
// MFMAs per block and K tile - the same fragment bytes, accumulator registers and matrix cycles as 4 x 4 blocks of 16 x 16 x 32, at
// half the operand register reads per FLOP.  A bare loop of either shape on random data (tools/mfma_shape_bench.hip,
// profiles/r04_mfma_shape_microbench.log): 1.27 against 1.08 PFLOP/s at one wave per SIMD with the fragments re-read from LDS,
// 1.82 against 1.43 from registers - the chip holds a higher clock on the wider shape.
template <typename T, int BM, int BN, int WM, int WN, bool A_T, bool B_T, int EPI, int DBG = 0, bool F8 = false, bool W32 = false>
__device__ __forceinline__ void gemm_pp_body(const GemmArgs& g, const int bid) {
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass only needs the declaration (it rejects the TN instantiation of this body)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NSA = BM / 128, NSB = BN / 128, NSUB = NSA + NSB, STAGE = NSUB * PP_SUB;
  constexpr int MB = W32 ? 32 : 16;                                       // rows / columns of one MFMA block
  constexpr int TM = BM / WM, TN = BN / WN, MI = TM / MB, NI = TN / MB;
  static_assert(WM * WN == PP_CWAVES && MI % 2 == 0, "wave layout");
  static_assert(!W32 || (!A_T && !B_T && !F8), "the 32 x 32 x 16 form is built for K-contiguous bf16 operands");
  static_assert(128 % TM == 0 && 128 % TN == 0, "a wave tile must not straddle two sub-images");
  constexpr int MH = MI / 2;                  // m-tiles per half
  constexpr int NP = 4 * NSUB;                // DMA pieces per loader wave per K tile
  static_assert(2 * NP <= 63, "vmcnt range");
  static_assert(PP_S * STAGE + ln_rows_bytes<EPI, BM, BN>() <= 160 * 1024 && BM * cpitch<BN>() + colsum_scratch_bytes<BM, BN, PP_THREADS>() <= PP_S * STAGE, "LDS budget");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Tile order: each XCD (private 4 MiB L2) owns a contiguous run of logical ids (xcd_remap) and its 32 CUs hold 32 consecutive
  // ids at a time.  Ids walk GROUPS of PP_GM tile rows column by column, so those 32 tiles form a PP_GM x (32 / PP_GM) block:
  // per K step they request 32 x 48 KiB but only PP_GM A slices + 32/PP_GM B slices are distinct (82 % L2 hits measured for
  // 4 x 8; a plain row-major order shares ONE A slice and fetches 32 different B slices: 64 % measured).
  const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM;
  const int per_group = PP_GM * tiles_n;
  const int gid = bid / per_group, first_m = gid * PP_GM;
  const int gsz = (tiles_m - first_m < PP_GM) ? tiles_m - first_m : PP_GM;
  const int rin = bid - gid * per_group;
  const int m0 = (first_m + rin % gsz) * BM, n0 = (rin / gsz) * BN;
  constexpr int ESZ = F8 ? 1 : 2, KT = F8 ? 2 * BK : BK;      // element bytes, elements per ring stage
  static_assert(!F8 || (!A_T && !B_T), "fp8 operands are K-contiguous");
  const int nk = (g.K + KT - 1) / KT;

  if (wid >= PP_CWAVES) {
    // ------------------------------------------------------------------ loader waves: LDS-DMA only
    const int lw = wid - PP_CWAVES;
    const unsigned bytesA = (unsigned)((((long)(A_T ? g.K : g.M) - 1) * g.lda + (A_T ? g.M : g.K)) * ESZ);
    const unsigned bytesB = (unsigned)((((long)(B_T ? g.K : g.N) - 1) * g.ldb + (B_T ? g.N : g.K)) * ESZ);
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, bytesA, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)g.B, 0, bytesB, 0x00020000);
    int voA[NSA][4], voB[NSB][4];
#pragma unroll
    for (int s = 0; s < NSA; ++s) pp_offsets<A_T, ESZ, W32>(g.lda, ((DBG & 2) ? 0 : m0) + 128 * s, lw, lane, voA[s]);
#pragma unroll
    for (int s = 0; s < NSB; ++s) pp_offsets<B_T, ESZ, W32>(g.ldb, ((DBG & 2) ? 0 : n0) + 128 * s, lw, lane, voB[s]);
    const int stepA = (DBG & 2) ? 0 : (int)((A_T ? (long)BK * g.lda : KT) * ESZ), stepB = (DBG & 2) ? 0 : (int)((B_T ? (long)BK * g.ldb : KT) * ESZ);
    auto issue_tile = [&](int t, char* dst) {
#pragma unroll
      for (int s = 0; s < NSA; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_void_t*)(dst + s * PP_SUB + (lw + 4 * i) * 1024), 16, voA[s][i], t * stepA, 0, 0);
#pragma unroll
      for (int s = 0; s < NSB; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_void_t*)(dst + (NSA + s) * PP_SUB + (lw + 4 * i) * 1024), 16, voB[s][i], t * stepB, 0, 0);
    };
    issue_tile(0, smem);
    if (nk > 1) {
      issue_tile(1, smem + STAGE);
      PP_VMCNT(NP);
    } else {
      PP_VMCNT(0);
    }
    int slot2 = 2;
    for (int kt = 0; kt < nk; ++kt) {
      __builtin_amdgcn_s_barrier();                                // B_kt
      if (!(DBG & 1)) {
        if (kt + 2 < nk) issue_tile(kt + 2, smem + slot2 * STAGE);
        if (kt + 1 < nk) { if (kt + 2 < nk) PP_VMCNT(NP); else PP_VMCNT(0); }     // K tile kt+1 landed (this wave's share)
      }
      slot2 = (slot2 + 1 == PP_S) ? 0 : slot2 + 1;
    }
    __builtin_amdgcn_s_barrier();                                  // B_nk: every fragment read has returned
    __builtin_amdgcn_s_barrier();                                  // accumulators parked
    epilogue_lds<EPI, T, BM, BN, PP_THREADS>(smem, g, m0, n0, tid, reinterpret_cast<const float*>(smem + PP_S * STAGE));
    return;
  }

  // -------------------------------------------------------------------- compute waves
  const int grp = wid >> 2;                   // waves w and w + 4 share a SIMD: group 1 runs half a tile behind group 0
  const int wm = wid / WN, wn = wid % WN;
  typedef typename std::conditional<W32, f32x16, f32x4>::type acc_t;
  acc_t acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < (W32 ? 16 : 4); ++e) acc[i][j][e] = 0.f;
  // sub-image and row / column inside it of this wave's tile
  const int a_off = ((wm * TM) / 128) * PP_SUB, a_rc = (wm * TM) % 128;
  const int b_off = (NSA + (wn * TN) / 128) * PP_SUB, b_rc = (wn * TN) % 128;
  constexpr int KS = W32 ? 4 : 2;                    // fragments per block and 64-deep K tile (16- or 32-deep MFMAs)
  r16x8 fb[NI][KS], fa[MI][KS];
  // 32-row fragment: lane l holds row l & 31, k = 16 ks + 8 (l >> 5) .. + 7: ONE ds_read_b128 at chunk 2 ks + (l >> 5)
  auto frag32 = [&](const char* img, int rc0, int ks) -> r16x8 {
    return *reinterpret_cast<const r16x8*>(img + img128w_off(rc0 + (lane & 31), 2 * ks + (lane >> 5)));
  };

#define PP_READ_B(img)                                                          \
  _Pragma("unroll") for (int j = 0; j < NI; ++j) {                             \
    if constexpr (W32) {                                                        \
      _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) fb[j][ks] = frag32((img) + b_off, b_rc + 32 * j, ks);   \
    } else {                                                                    \
    fb[j][0] = read_frag<B_T, 128>((img) + b_off, b_rc + 16 * j, 0, lane);     \
    fb[j][1] = read_frag<B_T, 128>((img) + b_off, b_rc + 16 * j, 1, lane);     \
    }                                                                           \
  }
#define PP_READ_A(img, I0)                                                      \
  _Pragma("unroll") for (int i = (I0); i < (I0) + MH; ++i) {                   \
    if constexpr (W32) {                                                        \
      _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) fa[i][ks] = frag32((img) + a_off, a_rc + 32 * i, ks);   \
    } else {                                                                    \
    fa[i][0] = read_frag<A_T, 128>((img) + a_off, a_rc + 16 * i, 0, lane);     \
    fa[i][1] = read_frag<A_T, 128>((img) + a_off, a_rc + 16 * i, 1, lane);     \
    }                                                                           \
  }
#define PP_MFMA(I0)                                                             \
  if constexpr (W32) {                                                          \
    _Pragma("unroll") for (int ks = 0; ks < KS; ++ks)                          \
      _Pragma("unroll") for (int i = (I0); i < (I0) + MH; ++i)                 \
        _Pragma("unroll") for (int j = 0; j < NI; ++j)                         \
          acc[i][j] = mfma32<T>(fb[j][ks], fa[i][ks], acc[i][j]);  \
  } else if constexpr (!F8) {                                                   \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                           \
      _Pragma("unroll") for (int i = (I0); i < (I0) + MH; ++i)                 \
        _Pragma("unroll") for (int j = 0; j < NI; ++j)                         \
          acc[i][j] = mfma16<T>(fb[j][ks], fa[i][ks], acc[i][j]);  \
  } else {                                                                      \
    _Pragma("unroll") for (int i = (I0); i < (I0) + MH; ++i)                   \
      _Pragma("unroll") for (int j = 0; j < NI; ++j)                           \
        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(f8cat(fb[j][0], fb[j][1]), f8cat(fa[i][0], fa[i][1]), acc[i][j], \
                                                                     0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);                         \
  }

  if constexpr (epi_is_fold<EPI>()) {                  // row statistics of the folded LayerNorm, while the first K tile is on its way (gemm_common.h)
    ln_rows_to_lds(g, m0, BM, reinterpret_cast<float*>(smem + PP_S * STAGE), tid, 64 * PP_CWAVES);
    ln_cols_to_lds<BM, BN>(g, n0, reinterpret_cast<float*>(smem + PP_S * STAGE), 64 * PP_CWAVES - 1 - tid);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  int slot = 0;
  if (grp == 0) {
    for (int kt = 0; kt < nk; ++kt) {
      const char* cur = smem + slot * STAGE;
      __builtin_amdgcn_s_barrier();                     // B_kt
      PP_FENCE; PP_SB;
      if (!(DBG & 4) || kt == 0) { PP_READ_B(cur) PP_READ_A(cur, 0) PP_READ_A(cur, MH) }
      PP_FENCE; PP_SB;
      __builtin_amdgcn_s_setprio(1);
      PP_MFMA(0)
      PP_MFMA(MH)
      __builtin_amdgcn_s_setprio(0);
      PP_FENCE; PP_SB;
      slot = (slot + 1 == PP_S) ? 0 : slot + 1;
    }
  } else {
    for (int kt = 0; kt < nk; ++kt) {
      const char* cur = smem + slot * STAGE;
      __builtin_amdgcn_s_barrier();                     // B_kt
      PP_FENCE; PP_SB;
      if (kt > 0) {                                     // lower half of tile kt-1: B fragments and lower A fragments are in registers
        __builtin_amdgcn_s_setprio(1);
        PP_MFMA(MH)
        __builtin_amdgcn_s_setprio(0);
      }
      PP_FENCE; PP_SB;
      if (!(DBG & 4) || kt == 0) { PP_READ_B(cur) PP_READ_A(cur, 0) }
      PP_FENCE; PP_SB;
      __builtin_amdgcn_s_setprio(1);
      PP_MFMA(0)
      __builtin_amdgcn_s_setprio(0);
      PP_FENCE; PP_SB;
      if (!(DBG & 4) || kt == 0) { PP_READ_A(cur, MH) }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the reads of this slot have returned before the barrier that frees it
      PP_FENCE; PP_SB;
      slot = (slot + 1 == PP_S) ? 0 : slot + 1;
    }
  }
  __builtin_amdgcn_s_barrier();                         // B_nk
  if (grp == 1) { PP_MFMA(MH) }                         // lower half of the last tile
#undef PP_READ_A
#undef PP_READ_B
#undef PP_MFMA

  // every fragment read of the ring has returned: park the accumulators and run the fused epilogue with all waves
  if constexpr (W32) {
    // D = B_frag x A_frag (operands swapped as in the 16 x 16 form): lane l holds row l & 31 of the block and, in acc[4 q .. 4 q + 3],
    // the four consecutive columns 8 q + 4 (l >> 5) .. + 3
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<f32x4*>(smem + (wm * TM + 32 * i + (lane & 31)) * cpitch<BN>() + (wn * TN + 32 * j + 8 * q + 4 * (lane >> 5)) * 4) =
              f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
  } else {
    park_acc<MI, NI, BN>(acc, smem, wm * TM, wn * TN, lane);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  epilogue_lds<EPI, T, BM, BN, PP_THREADS>(smem, g, m0, n0, tid, reinterpret_cast<const float*>(smem + PP_S * STAGE));
#endif
}

template <typename T, int BM, int BN, int WM, int WN, bool A_T, bool B_T, int EPI>
__global__ __launch_bounds__(PP_THREADS, 3) void gemm_pp_kernel(const GemmArgs g) {
  gemm_pp_body<T, BM, BN, WM, WN, A_T, B_T, EPI>(g, xcd_remap(blockIdx.x, gridDim.x));
}
template <typename T, int EPI>
__global__ __launch_bounds__(PP_THREADS, 3) void gemm_pp_w32_kernel(const GemmArgs g) {      // NT, 32 x 32 x 16 MFMAs
  gemm_pp_body<T, PP_BM, PP_BN, 4, 2, false, false, EPI, 0, false, true>(g, xcd_remap(blockIdx.x, gridDim.x));
}
template <int EPI>
__global__ __launch_bounds__(PP_THREADS, 3) void gemm_pp_f8_kernel(const GemmArgs g) {      // (16-bit outputs of the fp8 path are bf16)
  gemm_pp_body<bf16_t, PP_BM, PP_BN, 4, 2, false, false, EPI, 0, true>(g, xcd_remap(blockIdx.x, gridDim.x));
}
template <int DBG>
__global__ __launch_bounds__(PP_THREADS, 3) void gemm_pp_dbg_kernel(const GemmArgs g) {
  gemm_pp_body<bf16_t, PP_BM, PP_BN, 4, 2, false, false, EPI_STORE_BF16, DBG>(g, xcd_remap(blockIdx.x, gridDim.x));
}

// grouped weight gradients (TN, fp32 store / accumulate)
template <typename T>
__global__ __launch_bounds__(PP_THREADS, 3) void gemm_pp_grouped_tn_kernel(const GemmGroup G) {
  // a workgroup walks tiles bid, bid + grid, ...: one tile each by default; with fewer workgroups than tiles (g_pp_wgrad_wgs) the launch holds that many
  // CUs - each of them completely: 12 waves x <= 168 VGPRs fill a CU's register file - and leaves the rest to the other stream's kernels
  const int total = G.tile_end[GROUP_MAX - 1];
  for (int bid = xcd_remap(blockIdx.x, gridDim.x); bid < total; bid += gridDim.x) {
    int p = 0;
    while (p + 1 < G.count && bid >= G.tile_end[p]) ++p;      // workgroup-uniform
    gemm_pp_body<T, PP_BM, PP_BN, 4, 2, true, true, EPI_STORE_F32>(G.p[p], bid - (p ? G.tile_end[p - 1] : 0));
    if (bid + (int)gridDim.x < total) __syncthreads();          // the parked tile has been consumed before the next tile's first DMA lands on it
  }
}

// the same with the AdamW update of the differentiated weights in the epilogue (EPI_ADAMW)
// A workgroup walks tiles bid, bid + grid, ...: with fewer workgroups than tiles (g_pp_adamw_wgs) the HBM-bound epilogues - 26 B per
// weight, during which the workgroup's CU computes nothing - occupy that many CUs instead of one per tile, and the rest of the
// chip stays with the other stream's kernels.
template <typename T>
__global__ __launch_bounds__(PP_THREADS, 3) void gemm_pp_grouped_tn_adamw_kernel(const GemmGroup G) {
  const int total = G.tile_end[GROUP_MAX - 1];
  for (int bid = xcd_remap(blockIdx.x, gridDim.x); bid < total; bid += gridDim.x) {
    int p = 0;
    while (p + 1 < G.count && bid >= G.tile_end[p]) ++p;
    gemm_pp_body<T, PP_BM, PP_BN, 4, 2, true, true, EPI_ADAMW>(G.p[p], bid - (p ? G.tile_end[p - 1] : 0));
    __syncthreads();      // the parked tile has been consumed before the next tile's first DMA lands on it
  }
}

template <typename T, bool A_T, bool B_T, int EPI>
int launch_pp_t(const GemmArgs& a, hipStream_t s) {
  constexpr int BM = PP_BM, BN = PP_BN;
  constexpr int LDS = PP_S * (BM / 128 + BN / 128) * PP_SUB + ln_rows_bytes<EPI, BM, BN>();
  const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  auto kern = gemm_pp_kernel<T, BM, BN, 4, 2, A_T, B_T, EPI>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set = true;
  }
  const int slot = nv_prof_begin(10 + (A_T ? 2 : (B_T ? 1 : 0)), 2.0 * a.M * a.N * a.K, s);
  nv_prof_bytes(slot, gemm_algo_bytes(a, EPI, 2));
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(PP_THREADS), LDS, s, a);
  nv_prof_end(slot, s);
  NV_CHECK_LAUNCH("nv_gemm_bf16/pp");
  return NV_OK;
}

}  // namespace

// workgroups of the weight-gradient launch with the AdamW epilogue (0 = one per tile); nv_gemm_set_tile(12, n).  Half the chip: ViT3D-base,
// batch 4, same box, volumes/s of the train step - one per tile (216) 1107, 144: 1082, 136: 1108, 128: 1122, 120: 1123, 112: 1124,
// 108: 1133 against 1140 for 128 on another box, 96: 1055, 72: 1078; update unfused 1094 (profiles/r04_adamw_in_wgrad_epilogue.log)
int g_pp_adamw_wgs = 128;
int g_pp_wgrad_wgs = 0;      // workgroups of the plain grouped weight-gradient launch (0 = one per tile); nv_gemm_set_tile(14, n)
int g_pp_w32 = 0;   // NT problems on the 32 x 32 x 16 MFMA form of the kernel (nv_gemm_set_tile(11, 0 | 1))
template <typename T, int EPI>
static int launch_pp_w32_t(const GemmArgs& a, hipStream_t s) {
  constexpr int LDS = PP_S * (PP_BM / 128 + PP_BN / 128) * PP_SUB;
  const int tiles = ((a.M + PP_BM - 1) / PP_BM) * ((a.N + PP_BN - 1) / PP_BN);
  auto kern = gemm_pp_w32_kernel<T, EPI>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set = true;
  }
  const int slot = nv_prof_begin(10, 2.0 * a.M * a.N * a.K, s);
  nv_prof_bytes(slot, gemm_algo_bytes(a, EPI, 2));
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(PP_THREADS), LDS, s, a);
  nv_prof_end(slot, s);
  NV_CHECK_LAUNCH("nv_gemm_bf16/pp32");
  return NV_OK;
}

int g_pp_dbg = 0;   // timing-only ablations of the NT / bf16-store kernel (tools/gemm_bench.py --dbg; results are wrong by design):
                    // 1 no DMA after the prologue, 2 every DMA from one L2-hot 48 KiB region, 4 fragments read once (5 = 1 + 4: MFMA only)
template <int DBG>
static int launch_pp_dbg(const GemmArgs& a, hipStream_t s) {
  constexpr int LDS = PP_S * (PP_BM / 128 + PP_BN / 128) * PP_SUB;
  const int tiles = ((a.M + PP_BM - 1) / PP_BM) * ((a.N + PP_BN - 1) / PP_BN);
  auto kern = gemm_pp_dbg_kernel<DBG>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(PP_THREADS), LDS, s, a);
  return NV_OK;
}

template <typename T>
static int launch_pp_fmt(int layout, int epi, const GemmArgs& a, hipStream_t s) {
  if (g_pp_dbg && layout == 0 && epi == EPI_STORE_BF16) {
    switch (g_pp_dbg) {
      case 1: return launch_pp_dbg<1>(a, s);
      case 2: return launch_pp_dbg<2>(a, s);
      case 4: return launch_pp_dbg<4>(a, s);
      case 5: return launch_pp_dbg<5>(a, s);
      default: break;
    }
  }
  if (g_pp_w32 && layout == 0) {
    switch (epi) {
      case EPI_STORE_BF16: return launch_pp_w32_t<T, EPI_STORE_BF16>(a, s);
      case EPI_STORE_F32: return launch_pp_w32_t<T, EPI_STORE_F32>(a, s);
      case EPI_BIAS_F32: return launch_pp_w32_t<T, EPI_BIAS_F32>(a, s);
      case EPI_BIAS_GELU: return launch_pp_w32_t<T, EPI_BIAS_GELU>(a, s);
      case EPI_BIAS_RESID: return launch_pp_w32_t<T, EPI_BIAS_RESID>(a, s);
      default: break;
    }
  }
  switch (layout * 16 + epi) {
    case 0 * 16 + EPI_STORE_BF16: return launch_pp_t<T, false, false, EPI_STORE_BF16>(a, s);
    case 0 * 16 + EPI_STORE_F32: return launch_pp_t<T, false, false, EPI_STORE_F32>(a, s);
    case 0 * 16 + EPI_BIAS_F32: return launch_pp_t<T, false, false, EPI_BIAS_F32>(a, s);
    case 0 * 16 + EPI_BIAS_GELU: return launch_pp_t<T, false, false, EPI_BIAS_GELU>(a, s);
    case 0 * 16 + EPI_BIAS_RESID: return launch_pp_t<T, false, false, EPI_BIAS_RESID>(a, s);
    case 0 * 16 + EPI_BIAS_RESID_LN: return launch_pp_t<T, false, false, EPI_BIAS_RESID_LN>(a, s);
    case 0 * 16 + EPI_LNFOLD_STORE: return launch_pp_t<T, false, false, EPI_LNFOLD_STORE>(a, s);
    case 0 * 16 + EPI_LNFOLD_GELU: return launch_pp_t<T, false, false, EPI_LNFOLD_GELU>(a, s);
    case 1 * 16 + EPI_STORE_BF16: return launch_pp_t<T, false, true, EPI_STORE_BF16>(a, s);
    case 1 * 16 + EPI_STORE_F32: return launch_pp_t<T, false, true, EPI_STORE_F32>(a, s);
    case 1 * 16 + EPI_DGELU: return launch_pp_t<T, false, true, EPI_DGELU>(a, s);
    case 1 * 16 + EPI_DGELU_COLSUM: return launch_pp_t<T, false, true, EPI_DGELU_COLSUM>(a, s);
    case 2 * 16 + EPI_STORE_F32: return launch_pp_t<T, true, true, EPI_STORE_F32>(a, s);
    default: break;
  }
  nv_set_error("nv_gemm_bf16/pp: unsupported layout/epilogue combination (%d, %d)", layout, epi);
  return NV_ERR_ARG;
}
int launch_pp(int layout, int epi, const GemmArgs& a, hipStream_t s) {
  NV_DISPATCH_OPERAND(T, return launch_pp_fmt<T>(layout, epi, a, s));
}

template <int EPI>
static int launch_pp_f8_t(const GemmArgs& a, hipStream_t s) {
  constexpr int LDS = PP_S * (PP_BM / 128 + PP_BN / 128) * PP_SUB;
  const int tiles = ((a.M + PP_BM - 1) / PP_BM) * ((a.N + PP_BN - 1) / PP_BN);
  auto kern = gemm_pp_f8_kernel<EPI>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set = true;
  }
  const int slot = nv_prof_begin(5, 2.0 * a.M * a.N * a.K, s);
  nv_prof_bytes(slot, gemm_algo_bytes(a, EPI, 1));
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(PP_THREADS), LDS, s, a);
  nv_prof_end(slot, s);
  NV_CHECK_LAUNCH("nv_gemm_f8");
  return NV_OK;
}

// fp8 (OCP e4m3) x fp8, NT layout, fp32 accumulate, per-column dequantisation in the epilogue
int launch_pp_f8(int epi, const GemmArgs& a, hipStream_t s) {
  switch (epi) {
    case EPI_STORE_BF16: return launch_pp_f8_t<EPI_STORE_BF16>(a, s);
    case EPI_STORE_F32: return launch_pp_f8_t<EPI_STORE_F32>(a, s);
    case EPI_BIAS_RESID: return launch_pp_f8_t<EPI_BIAS_RESID>(a, s);
    case EPI_BIAS_GELU_F8: return launch_pp_f8_t<EPI_BIAS_GELU_F8>(a, s);
    case EPI_BIAS_GELU_F8T: return launch_pp_f8_t<EPI_BIAS_GELU_F8T>(a, s);
    default: break;
  }
  nv_set_error("nv_gemm_f8: unsupported epilogue %d (0 bf16 store, 1 f32 store, 4 bias + residual, 7 bias + GELU -> fp8, 8 the same + bf16 copies)", epi);
  return NV_ERR_ARG;
}

// grouped weight-gradient GEMMs (TN, fp32 store / accumulate) on 256 x 128 tiles
template <typename T>
static int launch_pp_grouped_tn_fmt(const GemmGroup& G, int tiles, double flops, hipStream_t s, bool adamw) {
  constexpr int BM = PP_BM, BN = PP_BN;
  constexpr int LDS = PP_S * (BM / 128 + BN / 128) * PP_SUB;
  auto kern = adamw ? gemm_pp_grouped_tn_adamw_kernel<T> : gemm_pp_grouped_tn_kernel<T>;
  static bool attr_set[2] = {false, false};
  if (!attr_set[adamw]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set[adamw] = true;
  }
  const int slot = nv_prof_begin(adamw ? 14 : 13, flops, s);
  if (slot >= 0) {
    double bytes = 0.0;
    for (int i = 0; i < G.count; ++i) bytes += gemm_algo_bytes(G.p[i], adamw ? EPI_ADAMW : EPI_STORE_F32, 2);
    nv_prof_bytes(slot, bytes);
  }
  const int cap = adamw ? g_pp_adamw_wgs : g_pp_wgrad_wgs;
  const int wgs = (cap > 0 && cap < tiles) ? cap : tiles;
  hipLaunchKernelGGL(kern, dim3(wgs), dim3(PP_THREADS), LDS, s, G);
  nv_prof_end(slot, s);
  NV_CHECK_LAUNCH("nv_gemm_bf16_grouped/pp");
  return NV_OK;
}
int launch_pp_grouped_tn(const GemmGroup& G, int tiles, double flops, hipStream_t s, bool adamw) {
  NV_DISPATCH_OPERAND(T, return launch_pp_grouped_tn_fmt<T>(G, tiles, flops, s, adamw));
}
