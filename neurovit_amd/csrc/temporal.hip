// The 4D model's temporal head (src/models/NeuroEncoder.py:60-66, 207-230) as ONE launch per direction:
//
//   per_volume [B, T, 2]  ->  nn.TransformerEncoderLayer(d_model = 2, nhead = 2, dim_feedforward = F, post-norm, ReLU, dropout p)
//                         ->  mean over T  ->  nn.Linear(2, 2)                                          -> [B, 2]
//
// 10 280 parameters and a few hundred kFLOP per sample: on stock modules this is ~60 tiny launches per train micro-step (forward,
// autograd backward, a multi-tensor AdamW) - 0.38 ms beside a 3.4 ms frozen-encoder forward of the 20 volumes.  Here one
// 512-thread workgroup walks the samples:
//   * token-sized work (qkv, the two heads' T x T softmax with head_dim 1, out-projection, the two LayerNorms over 2 elements, the
//     mean and the projection) runs on the first T or 2 T threads out of LDS;
//   * the FeedForward is thread-per-hidden-unit: thread k keeps W1[k,:], b1[k], W2[:,k] in registers for the whole launch, walks
//     the T tokens, and the two output sums per token are reduced wave-wide then across the 8 waves in a fixed order;
//   * the backward pass recomputes the forward (nothing is saved but the input), keeps every hidden unit's weight gradients in
//     the registers of its thread across all samples and the small parameters' gradients in per-token registers reduced once at
//     the end: no atomics, bit-reproducible.
// Dropout uses the counter-based masks of common.h (four sites: attention probabilities, after attention, inside the FeedForward,
// after it), recomputed in backward from the seed.  Arithmetic is fp32 throughout, like the reference's.
//
// Parameter arena (floats; nn.Module.named_parameters() order of temporal_transformer then projection_head):
//   in_proj_weight [6,2] | in_proj_bias [6] | out_proj.weight [2,2] | out_proj.bias [2] | linear1.weight [F,2] | linear1.bias [F] |
//   linear2.weight [2,F] | linear2.bias [2] | norm1.weight [2] | norm1.bias [2] | norm2.weight [2] | norm2.bias [2] |
//   projection_head.weight [2,2] | projection_head.bias [2]                                             = 40 + 5 F floats
#include "common.h"

namespace {

constexpr int TH_THREADS = 512, TH_WAVES = TH_THREADS / 64, TH_MAXT = 64, TH_KPT = 4;   // 8 waves: 256 VGPRs per lane (16 waves spilled 70 in backward)

struct THOff {
  int win, bin, wo, bo, w1, b1, w2, b2, g1, be1, g2, be2, wp, bp, total;
};
__host__ __device__ inline THOff th_offsets(int F) {
  THOff o;
  o.win = 0; o.bin = 12; o.wo = 18; o.bo = 22; o.w1 = 24; o.b1 = 24 + 2 * F; o.w2 = 24 + 3 * F; o.b2 = 24 + 5 * F;
  o.g1 = 26 + 5 * F; o.be1 = o.g1 + 2; o.g2 = o.g1 + 4; o.be2 = o.g1 + 6; o.wp = 34 + 5 * F; o.bp = o.wp + 4; o.total = 40 + 5 * F;
  return o;
}

struct THArgs {
  const float* x;      // [B, T, 2]
  const float* p;      // parameter arena
  float* out;          // [B, 2]
  const float* dout;   // [B, 2]            (backward)
  float* grads;        // arena layout       (backward)
  float* dx;           // [B, T, 2] or null  (backward)
  int B, T, F, accumulate;
  float eps;
  DropCfg d_attn, d_sa, d_ff, d_out;
};

__device__ __forceinline__ float dropf(const DropCfg& d, unsigned long long idx) { return d.thresh ? drop_factor(d, idx) : 1.0f; }

// LayerNorm over the TWO elements of a token (biased variance, like nn.LayerNorm), in closed form: with c = (z0 - z1) / 2 the centred
// values are (+c, -c) and the variance is c^2, so h0 = c / sqrt(c^2 + eps) = -h1.  One rounding in front of the square root instead of
// the generic form's mean -> subtract -> square -> average chain.
__device__ __forceinline__ void ln2(float z0, float z1, float eps, float& h0, float& h1, float& rstd) {
  const float c = 0.5f * (z0 - z1);
  rstd = 1.0f / sqrtf(c * c + eps);
  h0 = c * rstd; h1 = -h0;
}
// its backward: dh = gradient w.r.t. the normalised values -> gradient w.r.t. z.  The generic form rstd * (dh - mean(dh) - h * mean(dh h))
// collapses, for two features, to  dz0 = -dz1 = (dh0 - dh1) / 2 * (1 - h0^2) * rstd  with  1 - h0^2 = eps / (c^2 + eps) = eps * rstd^2:
// a product, no cancellation.  The generic form subtracts two nearly equal numbers whenever the row is saturated (|c| >> sqrt(eps):
// every row at fixture-scale parameters), leaving fp32 noise of a few 1e-2 of the result - torch's own kernels have it too
// (profiles/r03_parity_report.txt); this form is exact to rounding against a float64 evaluation.
__device__ __forceinline__ void ln2_bwd(float dh0, float dh1, float rstd, float eps, float& dz0, float& dz1) {
  dz0 = 0.5f * (dh0 - dh1) * (eps * rstd * rstd) * rstd;
  dz1 = -dz0;
}

template <bool BWD>
__global__ __launch_bounds__(TH_THREADS) void temporal_head_kernel(const THArgs a) {
  __shared__ float sx[TH_MAXT][2], sqkv[TH_MAXT][6], sctx[TH_MAXT][2], sy1[TH_MAXT][2], sh1[TH_MAXT][2], sh2[TH_MAXT][2], sy2[TH_MAXT][2];
  __shared__ float srs1[TH_MAXT], srs2[TH_MAXT];
  __shared__ float spart[TH_MAXT][TH_WAVES][2];
  __shared__ float sp[BWD ? 2 * TH_MAXT * TH_MAXT : 1];            // attention probabilities (before dropout), later dS in place
  __shared__ float sdy1[BWD ? TH_MAXT : 1][2], sdf[BWD ? TH_MAXT : 1][2], sdctx[BWD ? TH_MAXT : 1][2], sdz1[BWD ? TH_MAXT : 1][2];
  __shared__ float sdqkv[BWD ? TH_MAXT : 1][6];
  __shared__ float spool[2], sdpool[2];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int T = a.T, F = a.F;
  const THOff o = th_offsets(F);
  const float* P = a.p;
  const float invT = 1.0f / (float)T;

  // this thread's hidden units: k = tid + 512 u
  float w1a[TH_KPT], w1b[TH_KPT], b1k[TH_KPT], w2a[TH_KPT], w2b[TH_KPT];
  float gw1a[TH_KPT], gw1b[TH_KPT], gb1[TH_KPT], gw2a[TH_KPT], gw2b[TH_KPT];
#pragma unroll
  for (int u = 0; u < TH_KPT; ++u) {
    const int k = tid + TH_THREADS * u;
    const bool on = k < F;
    w1a[u] = on ? P[o.w1 + 2 * k] : 0.f; w1b[u] = on ? P[o.w1 + 2 * k + 1] : 0.f; b1k[u] = on ? P[o.b1 + k] : 0.f;
    w2a[u] = on ? P[o.w2 + k] : 0.f; w2b[u] = on ? P[o.w2 + F + k] : 0.f;
    gw1a[u] = gw1b[u] = gb1[u] = gw2a[u] = gw2b[u] = 0.f;
  }
  // small parameters (uniform): read once
  float win[6][2], bin[6], wo[2][2], bo[2], b2[2], g1[2], be1[2], g2[2], be2[2], wp[2][2], bp[2];
#pragma unroll
  for (int r = 0; r < 6; ++r) { win[r][0] = P[o.win + 2 * r]; win[r][1] = P[o.win + 2 * r + 1]; bin[r] = P[o.bin + r]; }
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    wo[c][0] = P[o.wo + 2 * c]; wo[c][1] = P[o.wo + 2 * c + 1]; bo[c] = P[o.bo + c]; b2[c] = P[o.b2 + c];
    g1[c] = P[o.g1 + c]; be1[c] = P[o.be1 + c]; g2[c] = P[o.g2 + c]; be2[c] = P[o.be2 + c];
    wp[c][0] = P[o.wp + 2 * c]; wp[c][1] = P[o.wp + 2 * c + 1]; bp[c] = P[o.bp + c];
  }
  // per-token gradient accumulators of the small parameters (threads < T; reduced over wave 0 at the end) - backward only
  float gwin[6][2], gbin[6], gwo[2][2], gbo[2], gb2[2], gg1[2], gbe1[2], gg2[2], gbe2[2], gwp[2][2], gbp[2];
#pragma unroll
  for (int r = 0; r < 6; ++r) { gwin[r][0] = gwin[r][1] = gbin[r] = 0.f; }
#pragma unroll
  for (int c = 0; c < 2; ++c) { gwo[c][0] = gwo[c][1] = gbo[c] = gb2[c] = gg1[c] = gbe1[c] = gg2[c] = gbe2[c] = gwp[c][0] = gwp[c][1] = gbp[c] = 0.f; }

  for (int b = 0; b < a.B; ++b) {
    const unsigned long long tok0 = (unsigned long long)b * T;
    // ---------------------------------------------------------------- A: in-projection
    if (tid < T) {
      const float x0 = a.x[(tok0 + tid) * 2], x1 = a.x[(tok0 + tid) * 2 + 1];
      sx[tid][0] = x0; sx[tid][1] = x1;
#pragma unroll
      for (int r = 0; r < 6; ++r) sqkv[tid][r] = win[r][0] * x0 + win[r][1] * x1 + bin[r];
    }
    __syncthreads();
    // ---------------------------------------------------------------- B: two heads of dimension 1 (scale 1/sqrt(1)): row softmax
    if (tid < 2 * T) {
      const int h = tid / T, i = tid - h * T;
      const float q = sqkv[i][h];
      float m = -INFINITY;
      for (int j = 0; j < T; ++j) m = fmaxf(m, q * sqkv[j][2 + h]);
      float l = 0.f;
      for (int j = 0; j < T; ++j) l += expf(q * sqkv[j][2 + h] - m);
      const float inv = 1.0f / l;
      float ctx = 0.f;
      for (int j = 0; j < T; ++j) {
        const float p = expf(q * sqkv[j][2 + h] - m) * inv;
        if constexpr (BWD) sp[(h * T + i) * T + j] = p;
        ctx += p * dropf(a.d_attn, ((tok0 * 2 + (unsigned long long)h * T + i) * T) + j) * sqkv[j][4 + h];
      }
      sctx[i][h] = ctx;
    }
    __syncthreads();
    // ---------------------------------------------------------------- C: out-projection, residual, LayerNorm 1
    if (tid < T) {
      const float c0 = sctx[tid][0], c1 = sctx[tid][1];
      const float a0 = wo[0][0] * c0 + wo[0][1] * c1 + bo[0], a1 = wo[1][0] * c0 + wo[1][1] * c1 + bo[1];
      const float z0 = sx[tid][0] + a0 * dropf(a.d_sa, (tok0 + tid) * 2), z1 = sx[tid][1] + a1 * dropf(a.d_sa, (tok0 + tid) * 2 + 1);
      float h0, h1, rs;
      ln2(z0, z1, a.eps, h0, h1, rs);
      sh1[tid][0] = h0; sh1[tid][1] = h1; srs1[tid] = rs;
      sy1[tid][0] = g1[0] * h0 + be1[0]; sy1[tid][1] = g1[1] * h1 + be1[1];
    }
    __syncthreads();
    // ---------------------------------------------------------------- D: FeedForward, one thread per hidden unit
    for (int i = 0; i < T; ++i) {
      const float y0 = sy1[i][0], y1 = sy1[i][1];
      float f0 = 0.f, f1 = 0.f;
#pragma unroll
      for (int u = 0; u < TH_KPT; ++u) {
        const int k = tid + TH_THREADS * u;
        const float pre = w1a[u] * y0 + w1b[u] * y1 + b1k[u];
        const float hd = fmaxf(pre, 0.f) * dropf(a.d_ff, (tok0 + i) * F + k);
        f0 += w2a[u] * hd; f1 += w2b[u] * hd;
      }
      f0 = wave_sum(f0); f1 = wave_sum(f1);
      if (lane == 0) { spart[i][wid][0] = f0; spart[i][wid][1] = f1; }
    }
    __syncthreads();
    // ---------------------------------------------------------------- E: residual, LayerNorm 2
    if (tid < T) {
      float f0 = 0.f, f1 = 0.f;
#pragma unroll
      for (int w = 0; w < TH_WAVES; ++w) { f0 += spart[tid][w][0]; f1 += spart[tid][w][1]; }
      f0 += b2[0]; f1 += b2[1];
      const float z0 = sy1[tid][0] + f0 * dropf(a.d_out, (tok0 + tid) * 2), z1 = sy1[tid][1] + f1 * dropf(a.d_out, (tok0 + tid) * 2 + 1);
      float h0, h1, rs;
      ln2(z0, z1, a.eps, h0, h1, rs);
      sh2[tid][0] = h0; sh2[tid][1] = h1; srs2[tid] = rs;
      sy2[tid][0] = g2[0] * h0 + be2[0]; sy2[tid][1] = g2[1] * h1 + be2[1];
    }
    __syncthreads();
    // ---------------------------------------------------------------- F: mean over time, projection
    if (tid < 2) {
      float s = 0.f;
      for (int i = 0; i < T; ++i) s += sy2[i][tid];
      spool[tid] = s * invT;
    }
    __syncthreads();
    if (tid < 2 && !BWD) a.out[b * 2 + tid] = wp[tid][0] * spool[0] + wp[tid][1] * spool[1] + bp[tid];
    if constexpr (BWD) {
      // -------------------------------------------------------------- G: projection backward
      if (tid == 0) {
        const float d0 = a.dout[b * 2], d1 = a.dout[b * 2 + 1];
        gwp[0][0] += d0 * spool[0]; gwp[0][1] += d0 * spool[1]; gwp[1][0] += d1 * spool[0]; gwp[1][1] += d1 * spool[1];
        gbp[0] += d0; gbp[1] += d1;
        sdpool[0] = wp[0][0] * d0 + wp[1][0] * d1; sdpool[1] = wp[0][1] * d0 + wp[1][1] * d1;
      }
      __syncthreads();
      // -------------------------------------------------------------- H: LayerNorm 2 backward
      if (tid < T) {
        const float dy0 = sdpool[0] * invT, dy1 = sdpool[1] * invT;
        const float h0 = sh2[tid][0], h1 = sh2[tid][1];
        gg2[0] += dy0 * h0; gg2[1] += dy1 * h1; gbe2[0] += dy0; gbe2[1] += dy1;
        float dz0, dz1;
        ln2_bwd(dy0 * g2[0], dy1 * g2[1], srs2[tid], a.eps, dz0, dz1);
        sdy1[tid][0] = dz0; sdy1[tid][1] = dz1;
        const float df0 = dz0 * dropf(a.d_out, (tok0 + tid) * 2), df1 = dz1 * dropf(a.d_out, (tok0 + tid) * 2 + 1);
        sdf[tid][0] = df0; sdf[tid][1] = df1;
        gb2[0] += df0; gb2[1] += df1;
      }
      __syncthreads();
      // -------------------------------------------------------------- I: FeedForward backward
      for (int i = 0; i < T; ++i) {
        const float y0 = sy1[i][0], y1 = sy1[i][1], df0 = sdf[i][0], df1 = sdf[i][1];
        float e0 = 0.f, e1 = 0.f;
#pragma unroll
        for (int u = 0; u < TH_KPT; ++u) {
          const int k = tid + TH_THREADS * u;
          const float pre = w1a[u] * y0 + w1b[u] * y1 + b1k[u];
          const float fk = dropf(a.d_ff, (tok0 + i) * F + k);
          const float hd = fmaxf(pre, 0.f) * fk;
          gw2a[u] += df0 * hd; gw2b[u] += df1 * hd;
          const float dpre = (pre > 0.f) ? (df0 * w2a[u] + df1 * w2b[u]) * fk : 0.f;
          gw1a[u] += dpre * y0; gw1b[u] += dpre * y1; gb1[u] += dpre;
          e0 += dpre * w1a[u]; e1 += dpre * w1b[u];
        }
        e0 = wave_sum(e0); e1 = wave_sum(e1);
        if (lane == 0) { spart[i][wid][0] = e0; spart[i][wid][1] = e1; }
      }
      __syncthreads();
      // -------------------------------------------------------------- J: LayerNorm 1 backward, out-projection backward
      if (tid < T) {
        float dy0 = sdy1[tid][0], dy1 = sdy1[tid][1];
#pragma unroll
        for (int w = 0; w < TH_WAVES; ++w) { dy0 += spart[tid][w][0]; dy1 += spart[tid][w][1]; }
        const float h0 = sh1[tid][0], h1 = sh1[tid][1];
        gg1[0] += dy0 * h0; gg1[1] += dy1 * h1; gbe1[0] += dy0; gbe1[1] += dy1;
        float dz0, dz1;
        ln2_bwd(dy0 * g1[0], dy1 * g1[1], srs1[tid], a.eps, dz0, dz1);
        sdz1[tid][0] = dz0; sdz1[tid][1] = dz1;
        const float da0 = dz0 * dropf(a.d_sa, (tok0 + tid) * 2), da1 = dz1 * dropf(a.d_sa, (tok0 + tid) * 2 + 1);
        const float c0 = sctx[tid][0], c1 = sctx[tid][1];
        gbo[0] += da0; gbo[1] += da1;
        gwo[0][0] += da0 * c0; gwo[0][1] += da0 * c1; gwo[1][0] += da1 * c0; gwo[1][1] += da1 * c1;
        sdctx[tid][0] = wo[0][0] * da0 + wo[1][0] * da1; sdctx[tid][1] = wo[0][1] * da0 + wo[1][1] * da1;
      }
      __syncthreads();
      // -------------------------------------------------------------- K: dV (column sums over the queries)
      if (tid < 2 * T) {
        const int h = tid / T, j = tid - h * T;
        float dv = 0.f;
        for (int i = 0; i < T; ++i)
          dv += sp[(h * T + i) * T + j] * dropf(a.d_attn, ((tok0 * 2 + (unsigned long long)h * T + i) * T) + j) * sdctx[i][h];
        sdqkv[j][4 + h] = dv;
      }
      __syncthreads();
      // -------------------------------------------------------------- L: softmax backward per row -> dS in place, dQ
      if (tid < 2 * T) {
        const int h = tid / T, i = tid - h * T;
        const float dc = sdctx[i][h];
        float* prow = sp + (h * T + i) * T;
        const unsigned long long base = (tok0 * 2 + (unsigned long long)h * T + i) * T;
        float dot = 0.f;
        for (int j = 0; j < T; ++j) dot += prow[j] * (dropf(a.d_attn, base + j) * dc * sqkv[j][4 + h]);
        float dq = 0.f;
        for (int j = 0; j < T; ++j) {
          const float ds = prow[j] * (dropf(a.d_attn, base + j) * dc * sqkv[j][4 + h] - dot);
          prow[j] = ds;
          dq += ds * sqkv[j][2 + h];
        }
        sdqkv[i][h] = dq;
      }
      __syncthreads();
      // -------------------------------------------------------------- M: dK (column sums)
      if (tid < 2 * T) {
        const int h = tid / T, j = tid - h * T;
        float dk = 0.f;
        for (int i = 0; i < T; ++i) dk += sp[(h * T + i) * T + j] * sqkv[i][h];
        sdqkv[j][2 + h] = dk;
      }
      __syncthreads();
      // -------------------------------------------------------------- N: in-projection backward
      if (tid < T) {
        const float x0 = sx[tid][0], x1 = sx[tid][1];
        float dx0 = sdz1[tid][0], dx1 = sdz1[tid][1];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          const float d = sdqkv[tid][r];
          gwin[r][0] += d * x0; gwin[r][1] += d * x1; gbin[r] += d;
          dx0 += win[r][0] * d; dx1 += win[r][1] * d;
        }
        if (a.dx) { a.dx[(tok0 + tid) * 2] = dx0; a.dx[(tok0 + tid) * 2 + 1] = dx1; }
      }
    }
    __syncthreads();       // the token buffers are reused by the next sample
  }

  if constexpr (BWD) {
    float* G = a.grads;
    const bool acc = a.accumulate != 0;
    auto put = [&](int idx, float v) { G[idx] = acc ? G[idx] + v : v; };
#pragma unroll
    for (int u = 0; u < TH_KPT; ++u) {
      const int k = tid + TH_THREADS * u;
      if (k < F) {
        put(o.w1 + 2 * k, gw1a[u]); put(o.w1 + 2 * k + 1, gw1b[u]); put(o.b1 + k, gb1[u]);
        put(o.w2 + k, gw2a[u]); put(o.w2 + F + k, gw2b[u]);
      }
    }
    if (wid == 0) {        // T <= 64: every per-token accumulator lives in wave 0 (lanes >= T hold zeros)
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        const float s0 = wave_sum(gwin[r][0]), s1 = wave_sum(gwin[r][1]), sb = wave_sum(gbin[r]);
        if (lane == 0) { put(o.win + 2 * r, s0); put(o.win + 2 * r + 1, s1); put(o.bin + r, sb); }
      }
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const float s0 = wave_sum(gwo[c][0]), s1 = wave_sum(gwo[c][1]), s2 = wave_sum(gbo[c]), s3 = wave_sum(gb2[c]);
        const float s4 = wave_sum(gg1[c]), s5 = wave_sum(gbe1[c]), s6 = wave_sum(gg2[c]), s7 = wave_sum(gbe2[c]);
        if (lane == 0) {
          put(o.wo + 2 * c, s0); put(o.wo + 2 * c + 1, s1); put(o.bo + c, s2); put(o.b2 + c, s3);
          put(o.g1 + c, s4); put(o.be1 + c, s5); put(o.g2 + c, s6); put(o.be2 + c, s7);
          put(o.wp + 2 * c, gwp[c][0]); put(o.wp + 2 * c + 1, gwp[c][1]); put(o.bp + c, gbp[c]);      // thread 0's own sums
        }
      }
    }
  }
}

int th_check(const char* fn, const float* x, int B, int T, int ff, const float* params, float drop_p) {
  NV_CHECK_ARG(x && params, "%s: null pointer", fn);
  NV_CHECK_ARG(B >= 1 && T >= 1 && T <= TH_MAXT, "%s: B = %d, T = %d: up to %d timepoints per sample", fn, B, T, TH_MAXT);
  NV_CHECK_ARG(ff >= 1 && ff <= TH_THREADS * TH_KPT, "%s: dim_feedforward = %d: up to %d", fn, ff, TH_THREADS * TH_KPT);
  NV_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "%s: dropout %g out of [0, 1)", fn, (double)drop_p);
  return NV_OK;
}

void th_drop(THArgs& a, unsigned long seed, float p) {
  a.d_attn = make_drop(seed ^ 0x9E3779B97F4A7C15ul, p);
  a.d_sa = make_drop(seed ^ (0x9E3779B97F4A7C15ul * 2), p);
  a.d_ff = make_drop(seed ^ (0x9E3779B97F4A7C15ul * 3), p);
  a.d_out = make_drop(seed ^ (0x9E3779B97F4A7C15ul * 4), p);
}

}  // namespace

extern "C" long nv_temporal_head_param_count(int ff) { return ff >= 1 ? (long)th_offsets(ff).total : -1; }

extern "C" int nv_temporal_head_fwd(const float* x, int B, int T, int ff, const float* params, float eps, unsigned long drop_seed,
                                    float drop_p, float* out, void* stream) {
  if (int rc = th_check("nv_temporal_head_fwd", x, B, T, ff, params, drop_p)) return rc;
  NV_CHECK_ARG(out, "nv_temporal_head_fwd: null output");
  THArgs a{};
  a.x = x; a.p = params; a.out = out; a.B = B; a.T = T; a.F = ff; a.eps = eps;
  th_drop(a, drop_seed, drop_p);
  hipLaunchKernelGGL(temporal_head_kernel<false>, dim3(1), dim3(TH_THREADS), 0, (hipStream_t)stream, a);
  NV_CHECK_LAUNCH("nv_temporal_head_fwd");
  return NV_OK;
}

extern "C" int nv_temporal_head_bwd(const float* x, int B, int T, int ff, const float* params, float eps, unsigned long drop_seed,
                                    float drop_p, const float* dout, float* grads, int accumulate, float* dx, void* stream) {
  if (int rc = th_check("nv_temporal_head_bwd", x, B, T, ff, params, drop_p)) return rc;
  NV_CHECK_ARG(dout && grads, "nv_temporal_head_bwd: null pointer");
  THArgs a{};
  a.x = x; a.p = params; a.dout = dout; a.grads = grads; a.dx = dx; a.accumulate = accumulate; a.B = B; a.T = T; a.F = ff; a.eps = eps;
  th_drop(a, drop_seed, drop_p);
  hipLaunchKernelGGL(temporal_head_kernel<true>, dim3(1), dim3(TH_THREADS), 0, (hipStream_t)stream, a);
  NV_CHECK_LAUNCH("nv_temporal_head_bwd");
  return NV_OK;
}
