// 256 x 256 x 64 tiles: the large-problem member of the GEMM family (bf16 and fp8 operands) on gfx950.
//
// gemm_pp.hip's 256 x 128 tile moves 48 KiB through the CU's L2 -> LDS path per 1024 MFMA cycles, and what limits it is that
// DMA's presence on the CU (its ablations: 1.05-1.10 PFLOP/s with the DMA, 1.43-1.48 without).  A 256 x 256 tile moves 64 KiB
// per 2048 MFMA cycles: two thirds of the bytes per FLOP.  The price is registers - a wave's 128 x 64 tile is 128 accumulator
// VGPRs, ~220 in all - so there is no room for loader waves beside two compute waves per SIMD: the eight compute waves issue
// the LDS-DMA themselves (eight 1 KiB pieces per wave per K tile), and the ring has two stages (2 x 64 KiB).
//
// Schedule (one raw s_barrier per K tile, as in gemm_pp.hip): after barrier k tile k is complete in LDS and the slot of tile k-1
// is free.  Waves w and w + 4 share a SIMD and run half a tile out of step:
//   group 0 (rows 0-127):    B_k | issue DMA of tile k+1 | read B + upper A fragments, 32 MFMAs | read lower A fragments, 32 MFMAs
//   group 1 (rows 128-255):  B_k | 32 MFMAs (lower half of tile k-1, fragments held) | issue DMA of tile k+1 | read B + upper A
//                                  fragments, 32 MFMAs | read lower A fragments (held across the barrier)
// so one wave's DMA issue and fragment reads sit beside its partner's MFMAs.  Every wave waits for its own DMA pieces
// (s_waitcnt vmcnt(0)) right before the next barrier - a full K tile after issuing them.
// The fp32 C tile (256 KiB) does not fit in LDS: the fused epilogue runs in two passes of 128 rows (the rows of one wave group).
// Operand images, fragment reads, tile order and epilogues are those of gemm_pp.hip / gemm_common.h.
#include "gemm_common.h"
#include "gemm_pp.h"

namespace {

constexpr int PQ_THREADS = 512;
constexpr int PQ_SUB = 16384;       // one 128-wide sub-image of a 64-deep (bf16) / 128-deep (fp8) K tile
constexpr int PQ_NSUB = 4;          // two A + two B sub-images per stage
constexpr int PQ_STAGE = PQ_NSUB * PQ_SUB;
constexpr int PQ_GM = 4;            // tile rows per group of the tile order
constexpr int PQ_LDS = 128 * (PQ_BN * 4 + 16) + 8 * PQ_BN * 4;   // epilogue pass: fp32 [128][256] tile + column-sum scratch (> 2 stages)
static_assert(PQ_LDS >= 2 * PQ_STAGE && PQ_LDS <= 160 * 1024, "LDS budget");

// per-lane DMA source offsets (bytes) of pieces I = w, w + 8 of one 128-wide sub-image (eight issuing waves)
template <bool T, int ESZ>
__device__ __forceinline__ void pq_offsets(long ld, int rc0, int w, int lane, int (&voff)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int I = w + 8 * i;
    if constexpr (!T) {
      const int row = I * 8 + (lane >> 3), ch = (lane & 7) ^ ((lane >> 3) & 7);            // img128_off inverse
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
__device__ __forceinline__ i32x8 pq_f8cat(r16x8 lo, r16x8 hi) {
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  const i32x4 a = __builtin_bit_cast(i32x4, lo), b = __builtin_bit_cast(i32x4, hi);
  return i32x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

#define PQ_SB __builtin_amdgcn_sched_barrier(0)
#define PQ_FENCE asm volatile("" ::: "memory")

template <bool A_T, bool B_T, int EPI, bool F8, typename T>
__device__ __forceinline__ void gemm_pq_body(const GemmArgs& g, const int bid) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BM = PQ_BM, BN = PQ_BN, MI = 8, NI = 4, MH = 4;
  constexpr int ESZ = F8 ? 1 : 2, KT = F8 ? 2 * BK : BK;
  static_assert(!F8 || (!A_T && !B_T), "fp8 operands are K-contiguous");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wid >> 2, wn = wid & 3;                  // group = M half of the tile; waves w and w + 4 share a SIMD
  const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM;
  const int per_group = PQ_GM * tiles_n;
  const int gid = bid / per_group, first_m = gid * PQ_GM;
  const int gsz = (tiles_m - first_m < PQ_GM) ? tiles_m - first_m : PQ_GM;
  const int rin = bid - gid * per_group;
  const int m0 = (first_m + rin % gsz) * BM, n0 = (rin / gsz) * BN;
  const int nk = (g.K + KT - 1) / KT;

  const unsigned bytesA = (unsigned)((((long)(A_T ? g.K : g.M) - 1) * g.lda + (A_T ? g.M : g.K)) * ESZ);
  const unsigned bytesB = (unsigned)((((long)(B_T ? g.K : g.N) - 1) * g.ldb + (B_T ? g.N : g.K)) * ESZ);
  const dma_desc rA = uniform_rsrc(g.A, bytesA);
  const dma_desc rB = uniform_rsrc(g.B, bytesB);
  int voA[2][2], voB[2][2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    pq_offsets<A_T, ESZ>(g.lda, m0 + 128 * s, wid, lane, voA[s]);
    pq_offsets<B_T, ESZ>(g.ldb, n0 + 128 * s, wid, lane, voB[s]);
  }
  const int stepA = (int)((A_T ? (long)BK * g.lda : KT) * ESZ), stepB = (int)((B_T ? (long)BK * g.ldb : KT) * ESZ);
  auto issue_tile = [&](int t, char* dst) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        lds_dma<16>(rA, dst + s * PQ_SUB + (wid + 8 * i) * 1024, voA[s][i], t * stepA);
        lds_dma<16>(rB, dst + (2 + s) * PQ_SUB + (wid + 8 * i) * 1024, voB[s][i], t * stepB);
      }
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int a_off = grp * PQ_SUB;                                   // this wave's 128 rows = A sub-image `grp`
  const int b_off = (2 + (wn >> 1)) * PQ_SUB, b_rc = (wn & 1) * 64;
  r16x8 fb[NI][2], fa[MH][2];

#define PQ_READ_B(img)                                                          \
  _Pragma("unroll") for (int j = 0; j < NI; ++j) {                             \
    fb[j][0] = read_frag<B_T, 128>((img) + b_off, b_rc + 16 * j, 0, lane);     \
    fb[j][1] = read_frag<B_T, 128>((img) + b_off, b_rc + 16 * j, 1, lane);     \
  }
#define PQ_READ_A(img, I0)                                                      \
  _Pragma("unroll") for (int i = 0; i < MH; ++i) {                             \
    fa[i][0] = read_frag<A_T, 128>((img) + a_off, 16 * ((I0) + i), 0, lane);   \
    fa[i][1] = read_frag<A_T, 128>((img) + a_off, 16 * ((I0) + i), 1, lane);   \
  }
#define PQ_MFMA(I0)                                                             \
  if constexpr (!F8) {                                                          \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                           \
      _Pragma("unroll") for (int i = 0; i < MH; ++i)                           \
        _Pragma("unroll") for (int j = 0; j < NI; ++j)                         \
          acc[(I0) + i][j] = mfma16<T>(fb[j][ks], fa[i][ks], acc[(I0) + i][j]);  \
  } else {                                                                      \
    _Pragma("unroll") for (int i = 0; i < MH; ++i)                             \
      _Pragma("unroll") for (int j = 0; j < NI; ++j)                           \
        acc[(I0) + i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(pq_f8cat(fb[j][0], fb[j][1]), pq_f8cat(fa[i][0], fa[i][1]), \
                                                                            acc[(I0) + i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);         \
  }

  issue_tile(0, smem);
  if (grp == 0) {
    for (int kt = 0; kt < nk; ++kt) {
      const char* cur = smem + (kt & 1) * PQ_STAGE;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of tile kt
      __builtin_amdgcn_s_barrier();                         // B_kt
      PQ_FENCE; PQ_SB;
      if (kt + 1 < nk) issue_tile(kt + 1, smem + ((kt + 1) & 1) * PQ_STAGE);
      PQ_READ_B(cur) PQ_READ_A(cur, 0)
      PQ_FENCE; PQ_SB;
      __builtin_amdgcn_s_setprio(1);
      PQ_MFMA(0)
      __builtin_amdgcn_s_setprio(0);
      PQ_FENCE; PQ_SB;
      PQ_READ_A(cur, MH)
      PQ_FENCE; PQ_SB;
      __builtin_amdgcn_s_setprio(1);
      PQ_MFMA(MH)
      __builtin_amdgcn_s_setprio(0);
      PQ_FENCE; PQ_SB;
    }
  } else {
    for (int kt = 0; kt < nk; ++kt) {
      const char* cur = smem + (kt & 1) * PQ_STAGE;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                         // B_kt
      PQ_FENCE; PQ_SB;
      if (kt > 0) {                                         // lower half of tile kt-1: fragments are in registers
        __builtin_amdgcn_s_setprio(1);
        PQ_MFMA(MH)
        __builtin_amdgcn_s_setprio(0);
      }
      PQ_FENCE; PQ_SB;
      if (kt + 1 < nk) issue_tile(kt + 1, smem + ((kt + 1) & 1) * PQ_STAGE);
      PQ_READ_B(cur) PQ_READ_A(cur, 0)
      PQ_FENCE; PQ_SB;
      __builtin_amdgcn_s_setprio(1);
      PQ_MFMA(0)
      __builtin_amdgcn_s_setprio(0);
      PQ_FENCE; PQ_SB;
      PQ_READ_A(cur, MH)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads of this slot have returned before the barrier that frees it
      PQ_FENCE; PQ_SB;
    }
  }
  __builtin_amdgcn_s_barrier();                             // B_nk
  if (grp == 1) { PQ_MFMA(MH) }
#undef PQ_READ_A
#undef PQ_READ_B
#undef PQ_MFMA

  // fused epilogue, 128 rows (one wave group) at a time through an fp32 [128][256] LDS tile, all eight waves
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    __builtin_amdgcn_s_barrier();                           // the ring / the previous pass's tile is dead
    if (grp == pass) park_acc<MI, NI, BN>(acc, smem, 0, wn * 64, lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    epilogue_lds<EPI, T, 128, BN, PQ_THREADS>(smem, g, m0 + 128 * pass, n0, tid);
  }
#endif
}

template <bool A_T, bool B_T, int EPI, bool F8, typename T>
__global__ __launch_bounds__(PQ_THREADS, 2) void gemm_pq_kernel(const GemmArgs g) {
  gemm_pq_body<A_T, B_T, EPI, F8, T>(g, xcd_remap(blockIdx.x, gridDim.x));
}

template <bool A_T, bool B_T, int EPI, bool F8, typename T>
int launch_pq_t(const GemmArgs& a, hipStream_t s) {
  const int tiles = ((a.M + PQ_BM - 1) / PQ_BM) * ((a.N + PQ_BN - 1) / PQ_BN);
  auto kern = gemm_pq_kernel<A_T, B_T, EPI, F8, T>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, PQ_LDS);
    attr_set = true;
  }
  const int slot = nv_prof_begin(F8 ? 5 : 20 + (A_T ? 2 : (B_T ? 1 : 0)), 2.0 * a.M * a.N * a.K, s);
  nv_prof_bytes(slot, gemm_algo_bytes(a, EPI, (F8 ? 1 : 2)));
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(PQ_THREADS), PQ_LDS, s, a);
  nv_prof_end(slot, s);
  NV_CHECK_LAUNCH("nv_gemm/pq");
  return NV_OK;
}

template <typename T>
int launch_pq_fmt(int layout, int epi, const GemmArgs& a, hipStream_t s) {
  switch (layout * 16 + epi) {
    case 0 * 16 + EPI_STORE_BF16: return launch_pq_t<false, false, EPI_STORE_BF16, false, T>(a, s);
    case 0 * 16 + EPI_STORE_F32: return launch_pq_t<false, false, EPI_STORE_F32, false, T>(a, s);
    case 0 * 16 + EPI_BIAS_F32: return launch_pq_t<false, false, EPI_BIAS_F32, false, T>(a, s);
    case 0 * 16 + EPI_BIAS_GELU: return launch_pq_t<false, false, EPI_BIAS_GELU, false, T>(a, s);
    case 0 * 16 + EPI_BIAS_RESID: return launch_pq_t<false, false, EPI_BIAS_RESID, false, T>(a, s);
    case 1 * 16 + EPI_STORE_BF16: return launch_pq_t<false, true, EPI_STORE_BF16, false, T>(a, s);
    case 1 * 16 + EPI_STORE_F32: return launch_pq_t<false, true, EPI_STORE_F32, false, T>(a, s);
    case 1 * 16 + EPI_DGELU: return launch_pq_t<false, true, EPI_DGELU, false, T>(a, s);
    case 1 * 16 + EPI_DGELU_COLSUM: return launch_pq_t<false, true, EPI_DGELU_COLSUM, false, T>(a, s);
    case 2 * 16 + EPI_STORE_F32: return launch_pq_t<true, true, EPI_STORE_F32, false, T>(a, s);
    default: break;
  }
  nv_set_error("nv_gemm_bf16/pq: unsupported layout/epilogue combination (%d, %d)", layout, epi);
  return NV_ERR_ARG;
}

}  // namespace

int launch_pq(int layout, int epi, const GemmArgs& a, hipStream_t s) {
  NV_DISPATCH_OPERAND(T, return launch_pq_fmt<T>(layout, epi, a, s));
}

int launch_pq_f8(int epi, const GemmArgs& a, hipStream_t s) {
  switch (epi) {
    case EPI_STORE_BF16: return launch_pq_t<false, false, EPI_STORE_BF16, true, bf16_t>(a, s);
    case EPI_STORE_F32: return launch_pq_t<false, false, EPI_STORE_F32, true, bf16_t>(a, s);
    case EPI_BIAS_RESID: return launch_pq_t<false, false, EPI_BIAS_RESID, true, bf16_t>(a, s);
    case EPI_BIAS_GELU_F8: return launch_pq_t<false, false, EPI_BIAS_GELU_F8, true, bf16_t>(a, s);
    case EPI_BIAS_GELU_F8T: return launch_pq_t<false, false, EPI_BIAS_GELU_F8T, true, bf16_t>(a, s);
    default: break;
  }
  nv_set_error("nv_gemm_f8/pq: unsupported epilogue %d", epi);
  return NV_ERR_ARG;
}
