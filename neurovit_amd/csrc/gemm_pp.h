// Interface between gemm.hip (host dispatch) and gemm_pp.hip (eight-wave ping-pong kernel, 256 x 128 tiles).
#pragma once
#include "gemm_common.h"

constexpr int PP_BM = 256, PP_BN = 128;
constexpr int PQ_BM = 256, PQ_BN = 256;     // gemm_pq.hip
int launch_pq(int layout, int epi, const GemmArgs& a, hipStream_t s);
int launch_pq_f8(int epi, const GemmArgs& a, hipStream_t s);
int launch_pp(int layout, int epi, const GemmArgs& a, hipStream_t s);
int launch_pp_f8(int epi, const GemmArgs& a, hipStream_t s);
int launch_pp_grouped_tn(const GemmGroup& G, int tiles, double flops, hipStream_t s, bool adamw = false);
extern int g_pp_dbg;
extern int g_pp_w32;
extern int g_pp_adamw_wgs;
extern int g_pp_wgrad_wgs;
