// Internal C++ view of the C ABI structs (include/tfx.h) with device bf16 pointer types.
#pragma once
#include "tfx_common.h"
#define TFX_BF16_DEFINED
typedef __bf16 tfx_bf16;
#include "../../include/tfx.h"

namespace tfx {
using GemmNT = tfx_gemm_nt_args;
using GemmTN = tfx_gemm_tn_args;
enum { EPI_BF16 = TFX_EPI_BF16, EPI_F32 = TFX_EPI_F32, EPI_SILU = TFX_EPI_SILU, EPI_RESID = TFX_EPI_RESID,
       EPI_GEGLU = TFX_EPI_GEGLU, EPI_GEGLU_BWD = TFX_EPI_GEGLU_BWD, EPI_QKNR = TFX_EPI_QKV_NORM_ROPE };
int gemm_nt(const GemmNT& p, hipStream_t s);
int gemm_tn(const GemmTN& p, hipStream_t s);
int gemm_nt_plan(const GemmNT& p, int* kind, int* grid);
int gemm_tn_plan(const GemmTN& p, int* kind, int* tiles, int* splits, int* grid);
int attn_fwd(const tfx_attn_args& p, hipStream_t s);
int attn_bwd(const tfx_attn_args& p, hipStream_t s);
}  // namespace tfx
