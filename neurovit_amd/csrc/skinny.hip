// Linear layers on a handful of rows (R <= 4 per launch slice): out[r, :] = epilogue(a[r, :] . W), bf16 operands, fp32 accumulation.
//
// Where they run: under pool = 'cls' (NeuroEncoder.py:194) only the B cls rows of the LAST block leave the attention towards the head,
// and in the backward pass the residual gradient entering that block is exactly zero in every other row - its out-projection and
// FeedForward (forward and backward) are products of B rows.  The tiled MFMA kernels would spend 12-48 workgroups walking K serially
// on such a problem (31.8 us for 4 x 768 x 3072 against 16.7 us for the dense 2052-row GEMM); these are weight-streaming kernels:
// every weight element is read once, by whichever lane layout makes that read whole 128-byte lines.
//   NT (forward, W [N, K] row-major): one wave per output column, lanes across K in 16-byte pieces, wave_sum per row.
//   NN (backward, W [K, N] row-major): one 1024-thread workgroup per 16 output columns (48-192 workgroups); a wave reads 32 k-rows
//      x 16 columns per instruction, sixteen waves split K, partial sums meet in LDS in a fixed order.
// Rows are addressed through leading dimensions, so the cls rows of a [B, n, d] tensor are a view (ld = n * d), never a copy.
// Cast points and epilogue arithmetic are those of the tiled kernels (gemm_common.h::epilogue4), nn.Dropout masks included: the mask of the
// dense tensor, hashed at the element offset of the strided view.
#include "common.h"

namespace {

constexpr int SK_R = 4;       // rows per launch slice (blockIdx.y walks slices): ViT3D-base trains at batch 4

enum { SK_NT_RESID = 0, SK_NT_GELU = 1 };
enum { SK_NN_DGELU = 0, SK_NN_F32 = 1, SK_NN_BF16 = 2 };

template <int EPI, typename T>
__global__ __launch_bounds__(256) void skinny_nt_kernel(const r16* __restrict__ A, long lda, int R, const r16* __restrict__ W, long ldw, int N, int K,
                                                        const float* __restrict__ bias, const float* __restrict__ resid, long ldr,
                                                        void* __restrict__ out, long ldo, r16* __restrict__ u_out, long ldu, DropCfg drop) {
  const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int r0 = blockIdx.y * SK_R, rows = (R - r0 < SK_R) ? R - r0 : SK_R;
  if (n >= N) return;
  float acc[SK_R];
#pragma unroll
  for (int r = 0; r < SK_R; ++r) acc[r] = 0.f;
  const r16* w = W + (long)n * ldw;
  // four 512-element steps per pass, every load of a pass issued before its arithmetic: the loop is a chain of memory round trips
  // (K = 3072 is six steps per lane - two passes)
  for (int k = lane * 8; k < K; k += 4 * 512) {
    r16x8 wq[4], aq[4][SK_R];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int kq = k + 512 * q;
      if (kq < K) {
        wq[q] = *reinterpret_cast<const r16x8*>(w + kq);
#pragma unroll
        for (int r = 0; r < SK_R; ++r)
          if (r < rows) aq[q][r] = *reinterpret_cast<const r16x8*>(A + (long)(r0 + r) * lda + kq);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (k + 512 * q < K) {
#pragma unroll
        for (int r = 0; r < SK_R; ++r) {
          if (r < rows) {
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[r] = __builtin_fmaf(dec1<T>(aq[q][r][e]), dec1<T>(wq[q][e]), acc[r]);
          }
        }
      }
    }
  }
  float mine = 0.f;
#pragma unroll
  for (int r = 0; r < SK_R; ++r) {
    if (r < rows) {
      const float t = wave_sum(acc[r]);
      if (lane == r) mine = t;
    }
  }
  if (lane < rows) {
    const long row = r0 + lane;
    const float v = mine + bias[n];
    // nn.Dropout of the site (vit_3d.py:21,23,45): the mask of the DENSE [M, N] tensor this launch writes a few rows of - `out` is a strided view of it
    // (ldo = N x row spacing), so the element offset row * ldo + n IS the dense element index the tiled epilogues hash (gemm_common.h::epilogue4)
    const float keep = drop.thresh ? drop_factor(drop, (unsigned long long)(row * ldo + n)) : 1.f;
    if constexpr (EPI == SK_NT_RESID) {
      ((float*)out)[row * ldo + n] = v * keep + resid[row * ldr + n];
    } else {
      if (u_out) u_out[row * ldu + n] = cvt1<T>(v);
      ((r16*)out)[row * ldo + n] = cvt1<T>(gelu_f(v) * keep);
    }
  }
}

constexpr int NN_COLS = 16;    // output columns per workgroup (two 16-byte lanes per k-row)
constexpr int NN_KK = 64 / (NN_COLS / 8);   // k-rows per wave-instruction: 32

template <int EPI, typename T>
__global__ __launch_bounds__(1024) void skinny_nn_kernel(const r16* __restrict__ A, long lda, int R, const r16* __restrict__ W, long ldw, int N, int K,
                                                         const r16* __restrict__ u, long ldu, void* __restrict__ out, long ldo,
                                                         float* __restrict__ dcol, int accumulate, long fill_rows, int keep_every, DropCfg drop) {
  // LDS: every lane's partial sums [16 waves][32 k-row lanes][SK_R][16 columns] (128 KiB), then per-wave sums [16][SK_R][16]
  extern __shared__ __attribute__((aligned(16))) float sk_lds[];
  if (blockIdx.z == 1) {
    // nv_skinny_nn_sparse: `out` is a dense bf16 [fill_rows, N] matrix of which this launch computes the rows r * keep_every; the
    // z = 1 half of the grid clears all the others (was a hipMemsetAsync node in front of the launch)
    zero_rows_except(reinterpret_cast<char*>(out), fill_rows, (long)N * 2, keep_every, blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    return;
  }
  float (*part)[NN_KK][SK_R][NN_COLS] = reinterpret_cast<float (*)[NN_KK][SK_R][NN_COLS]>(sk_lds);
  float (*red)[SK_R][NN_COLS] = reinterpret_cast<float (*)[SK_R][NN_COLS]>(sk_lds + 16 * NN_KK * SK_R * NN_COLS);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, ng = lane & (NN_COLS / 8 - 1), kk = lane / (NN_COLS / 8);
  const int n0 = blockIdx.x * NN_COLS, nc = n0 + ng * 8;
  const int r0 = blockIdx.y * SK_R, rows = (R - r0 < SK_R) ? R - r0 : SK_R;
  float acc[SK_R][8];
#pragma unroll
  for (int r = 0; r < SK_R; ++r)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[r][e] = 0.f;
  if (nc < N) {                                                  // N % 8 == 0: a column group is inside or outside as a whole
    constexpr int STEP = 16 * NN_KK;                             // k-rows per workgroup step: 512
    // four steps per pass, loads first: a thread walks K / 512 rows (six at K = 3072) and each is a memory round trip
    for (int k = wv * NN_KK + kk; k < K; k += 4 * STEP) {
      r16x8 wq[4];
      float aq[4][SK_R];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int kq = k + STEP * q;
        if (kq < K) {
          wq[q] = *reinterpret_cast<const r16x8*>(W + (long)kq * ldw + nc);
#pragma unroll
          for (int r = 0; r < SK_R; ++r)
            if (r < rows) aq[q][r] = dec1<T>(A[(long)(r0 + r) * lda + kq]);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (k + STEP * q < K) {
#pragma unroll
          for (int r = 0; r < SK_R; ++r) {
            if (r < rows) {
#pragma unroll
              for (int e = 0; e < 8; ++e) acc[r][e] = __builtin_fmaf(aq[q][r], dec1<T>(wq[q][e]), acc[r][e]);
            }
          }
        }
      }
    }
  }
  // through LDS in a fixed order (cross-lane shuffles here are 160 ds_bpermute round trips per lane): every lane parks its partial
  // sums, thread (wave w, row r, column c) adds the 32 k-row lanes of its wave, then 64 threads add the sixteen waves
#pragma unroll
  for (int r = 0; r < SK_R; ++r) {
    *reinterpret_cast<f32x4*>(&part[wv][kk][r][ng * 8]) = f32x4{acc[r][0], acc[r][1], acc[r][2], acc[r][3]};
    *reinterpret_cast<f32x4*>(&part[wv][kk][r][ng * 8 + 4]) = f32x4{acc[r][4], acc[r][5], acc[r][6], acc[r][7]};
  }
  __syncthreads();
  {
    const int w = tid / (SK_R * NN_COLS), rc = tid % (SK_R * NN_COLS);      // 16 x 64 = 1024 threads
    float v = 0.f;
#pragma unroll
    for (int q = 0; q < NN_KK; ++q) v += part[w][q][rc / NN_COLS][rc % NN_COLS];
    red[w][rc / NN_COLS][rc % NN_COLS] = v;
  }
  __syncthreads();
  const int c = tid % NN_COLS, r = tid / NN_COLS;                // thread (r, c) finishes output (r0 + r, n0 + c); r < SK_R used
  float stored = 0.f;
  const bool live = r < rows && n0 + c < N;
  if (live) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) v += red[w][r][c];
    const long row = r0 + r;
    const int n = n0 + c;
    if constexpr (EPI == SK_NN_DGELU) {
      const float keep = drop.thresh ? drop_factor(drop, (unsigned long long)(row * ldo + n)) : 1.f;      // the mask of the GELU output this gradient flows back through
      const r16 o = cvt1<T>(v * keep * gelu_grad_f(dec1<T>(u[row * ldu + n])));
      ((r16*)out)[row * ldo + n] = o;
      stored = dec1<T>(o);                                         // what the weight-gradient product will read: summed as stored
    } else if constexpr (EPI == SK_NN_F32) {
      ((float*)out)[row * ldo + n] = v;
    } else {
      ((r16*)out)[row * ldo + n] = cvt1<T>(v);
    }
  }
  if constexpr (EPI == SK_NN_DGELU) {
    if (dcol) {                                                  // bias gradient of the Linear in front of the GELU: column sums over the rows
      __syncthreads();                                           // every partial sum has been read
      if (r < SK_R) red[0][r][c] = live ? stored : 0.f;
      __syncthreads();
      if (r == 0 && n0 + c < N) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < SK_R; ++q) s += red[0][q][c];
        dcol[n0 + c] = (accumulate ? dcol[n0 + c] : 0.f) + s;    // (the host keeps R <= SK_R when column sums are asked for: one slice)
      }
    }
  }
}

constexpr int NN_LDS = (16 * NN_KK * SK_R * NN_COLS + 16 * SK_R * NN_COLS) * (int)sizeof(float);
template <typename T>
void nn_attrs_fmt() {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(skinny_nn_kernel<SK_NN_DGELU, T>), hipFuncAttributeMaxDynamicSharedMemorySize, NN_LDS);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(skinny_nn_kernel<SK_NN_F32, T>), hipFuncAttributeMaxDynamicSharedMemorySize, NN_LDS);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(skinny_nn_kernel<SK_NN_BF16, T>), hipFuncAttributeMaxDynamicSharedMemorySize, NN_LDS);
}
void nn_attrs() {
  static bool attr = false;
  if (attr) return;
  nn_attrs_fmt<bf16_t>(); nn_attrs_fmt<fp16_t>();
  attr = true;
}

}  // namespace

// out[r, n] = resid[r, n] + (bias[n] + sum_k A[r, k] W[n, k])   (epi 0, out f32)   |   u = bias + sum; out = gelu(u) (epi 1, out / u bf16)
extern "C" int nv_skinny_nt(int epi, int R, int N, int K, const void* A, long lda, const void* W, long ldw, const float* bias, const float* resid,
                            long ldr, void* out, long ldo, void* u_out, long ldu, unsigned long drop_seed, float drop_p, void* stream) {
  NV_CHECK_ARG(drop_p == 0.f || (ldo % N) == 0, "nv_skinny_nt: with dropout `out` must be a row-strided view of a dense [M, N] tensor (ldo a multiple of N)");
  const DropCfg drop = make_drop(drop_seed, drop_p);
  NV_CHECK_ARG(R > 0 && N > 0 && K > 0 && (K % 8) == 0 && (lda % 8) == 0 && (ldw % 8) == 0 && A && W && bias && out && nv_aligned16(A) && nv_aligned16(W),
               "nv_skinny_nt: K, lda, ldw must be multiples of 8, operands 16-byte aligned");
  NV_CHECK_ARG(epi == SK_NT_GELU || (epi == SK_NT_RESID && resid), "nv_skinny_nt: epilogue 0 needs resid; epilogues are 0 (bias + residual, f32) and 1 (bias + GELU, bf16)");
  const dim3 grid((N + 3) / 4, (R + SK_R - 1) / SK_R), block(256);
  hipStream_t s = (hipStream_t)stream;
  NV_DISPATCH_OPERAND(T,
    if (epi == SK_NT_RESID)
      hipLaunchKernelGGL((skinny_nt_kernel<SK_NT_RESID, T>), grid, block, 0, s, (const r16*)A, lda, R, (const r16*)W, ldw, N, K, bias, resid, ldr, out, ldo, (r16*)nullptr, 0L, drop);
    else
      hipLaunchKernelGGL((skinny_nt_kernel<SK_NT_GELU, T>), grid, block, 0, s, (const r16*)A, lda, R, (const r16*)W, ldw, N, K, bias, resid, ldr, out, ldo, (r16*)u_out, ldu, drop));
  NV_CHECK_LAUNCH("nv_skinny_nt");
  return NV_OK;
}

// out[r, n] = epilogue(sum_k A[r, k] W[k, n]): epi 0 bf16 out = sum * gelu'(u[r, n]) (+ dcol[n] (+)= column sums of the stored values, R <= 4),
// 1 f32 store, 2 bf16 store.  N % 8 == 0.
extern "C" int nv_skinny_nn(int epi, int R, int N, int K, const void* A, long lda, const void* W, long ldw, const void* u, long ldu, void* out, long ldo,
                            float* dcol, int accumulate, unsigned long drop_seed, float drop_p, void* stream) {
  NV_CHECK_ARG(drop_p == 0.f || (epi == SK_NN_DGELU && (ldo % N) == 0), "nv_skinny_nn: dropout only with epilogue 0, `out` a row-strided view of a dense [M, N] tensor");
  const DropCfg drop = make_drop(drop_seed, drop_p);
  NV_CHECK_ARG(R > 0 && N > 0 && K > 0 && (N % 8) == 0 && (ldw % 8) == 0 && A && W && out && nv_aligned16(W), "nv_skinny_nn: N, ldw must be multiples of 8, W 16-byte aligned");
  NV_CHECK_ARG(epi >= 0 && epi <= 2 && (epi != SK_NN_DGELU || u) && (!dcol || (epi == SK_NN_DGELU && R <= SK_R)), "nv_skinny_nn: epilogue 0 needs u; column sums only with epilogue 0 and R <= 4");
  const dim3 grid((N + NN_COLS - 1) / NN_COLS, (R + SK_R - 1) / SK_R), block(1024);
  hipStream_t s = (hipStream_t)stream;
  nn_attrs();
#define SK_NN(E) hipLaunchKernelGGL((skinny_nn_kernel<E, T>), grid, block, NN_LDS, s, (const r16*)A, lda, R, (const r16*)W, ldw, N, K, (const r16*)u, ldu, out, ldo, dcol, accumulate, 0L, 0, drop)
  NV_DISPATCH_OPERAND(T, if (epi == SK_NN_DGELU) SK_NN(SK_NN_DGELU); else if (epi == SK_NN_F32) SK_NN(SK_NN_F32); else SK_NN(SK_NN_BF16));
#undef SK_NN
  NV_CHECK_LAUNCH("nv_skinny_nn");
  return NV_OK;
}

// out bf16 [total_rows, N] dense: rows r * keep_every (r < R) = A[r, :] W, every other row zero - ONE launch (the data gradient of a
// Linear whose incoming gradient is non-zero on the cls rows only: dAO of the last block under pool = 'cls').  N % 8 == 0.
extern "C" int nv_skinny_nn_sparse(int R, int N, int K, const void* A, long lda, const void* W, long ldw, void* out, long total_rows, int keep_every,
                                   void* stream) {
  NV_CHECK_ARG(R > 0 && N > 0 && K > 0 && (N % 8) == 0 && (ldw % 8) == 0 && A && W && out && nv_aligned16(W) && nv_aligned16(out) && keep_every >= 1 &&
                   (long)(R - 1) * keep_every < total_rows && (long)R * keep_every >= total_rows,
               "nv_skinny_nn_sparse: N, ldw multiples of 8; W, out 16-byte aligned; R = ceil(total_rows / keep_every) (every kept row is computed: the fill skips them all)");
  const dim3 grid((N + NN_COLS - 1) / NN_COLS, (R + SK_R - 1) / SK_R, 2), block(1024);
  nn_attrs();
  NV_DISPATCH_OPERAND(T, hipLaunchKernelGGL((skinny_nn_kernel<SK_NN_BF16, T>), grid, block, NN_LDS, (hipStream_t)stream, (const r16*)A, lda, R, (const r16*)W, ldw, N, K,
                                            (const r16*)nullptr, 0L, out, (long)N * keep_every, (float*)nullptr, 0, total_rows, keep_every, make_drop(0, 0.f)));
  NV_CHECK_LAUNCH("nv_skinny_nn_sparse");
  return NV_OK;
}
