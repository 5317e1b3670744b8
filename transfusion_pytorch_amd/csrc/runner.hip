// Launch-list replay (include/tfx.h "launch lists"): the training / decode step is a static list of kernel launches over persistent
// buffers, so the host hands the whole list to the library once per step instead of paying one FFI round trip per kernel.
// Host code only; every case forwards to the public entry point of the same name.
#include "../../include/tfx.h"

namespace {

inline int run_one(const tfx_launch& l, void* s) {
  const void* a = l.args;
  const tfx_raw_args* r = static_cast<const tfx_raw_args*>(a);
  switch (l.op) {
    case TFX_OP_GEMM_NT:          return tfx_gemm_nt(static_cast<const tfx_gemm_nt_args*>(a), s);
    case TFX_OP_GEMM_TN:          return tfx_gemm_tn(static_cast<const tfx_gemm_tn_args*>(a), s);
    case TFX_OP_ATTN_FWD:         return tfx_attn_fwd(static_cast<const tfx_attn_args*>(a), s);
    case TFX_OP_ATTN_BWD:         return tfx_attn_bwd(static_cast<const tfx_attn_args*>(a), s);
    case TFX_OP_ADALN_PRE_FWD:    return tfx_adaln_pre_fwd(static_cast<const tfx_adaln_pre_args*>(a), s);
    case TFX_OP_ADALN_PRE_BWD:    return tfx_adaln_pre_bwd(static_cast<const tfx_adaln_pre_args*>(a), s);
    case TFX_OP_ADALN_POST_FWD:   return tfx_adaln_post_fwd(static_cast<const tfx_adaln_post_args*>(a), s);
    case TFX_OP_ADALN_POST_BWD:   return tfx_adaln_post_bwd(static_cast<const tfx_adaln_post_args*>(a), s);
    case TFX_OP_QK_NORM_ROPE_FWD: return tfx_qk_norm_rope_fwd(static_cast<const tfx_qk_norm_rope_args*>(a), s);
    case TFX_OP_QK_NORM_ROPE_BWD: return tfx_qk_norm_rope_bwd(static_cast<const tfx_qk_norm_rope_args*>(a), s);
    case TFX_OP_ATTNRES_FWD:      return tfx_attnres_fwd(static_cast<const tfx_attnres_args*>(a), s);
    case TFX_OP_ATTNRES_BWD:      return tfx_attnres_bwd(static_cast<const tfx_attnres_args*>(a), s);
    case TFX_OP_RMSNORM_FWD:      return tfx_rmsnorm_fwd(static_cast<const tfx_rmsnorm_args*>(a), s);
    case TFX_OP_RMSNORM_BWD:      return tfx_rmsnorm_bwd(static_cast<const tfx_rmsnorm_args*>(a), s);
    case TFX_OP_EMBED_FWD:        return tfx_embed_fwd(static_cast<const tfx_embed_args*>(a), s);
    case TFX_OP_EMBED_BWD:        return tfx_embed_bwd(static_cast<const tfx_embed_args*>(a), s);
    case TFX_OP_NOISE_MIX:        return tfx_noise_mix(static_cast<const tfx_noise_mix_args*>(a), s);
    case TFX_OP_FOURIER:          return tfx_fourier(static_cast<const tfx_fourier_args*>(a), s);
    case TFX_OP_CE_FWD_BWD:       return tfx_ce_fwd_bwd(static_cast<const tfx_ce_args*>(a), s);
    case TFX_OP_MSE_FWD_BWD:      return tfx_mse_fwd_bwd(static_cast<const tfx_mse_args*>(a), s);
    case TFX_OP_CAST_ROWS:        return tfx_cast_rows(static_cast<const tfx_cast_args*>(a), s);
    case TFX_OP_CAST_ROWS_T:      return tfx_cast_rows_t(static_cast<const tfx_cast_args*>(a), s);
    case TFX_OP_ADAM_STEP:        return tfx_adam_step(static_cast<const tfx_adam_args*>(a), s);
    // positional entry points: pointers p0.., integers i0.., floats f0 in declaration order
    case TFX_OP_OUTPUT_TO_FLOW:
      return tfx_output_to_flow((float*)r->p0, (const float*)r->p1, (const float*)r->p2, (const int32_t*)r->p3, (const float*)r->p4,
                                (int32_t)r->i0, (int32_t)r->i1, r->f0, s);
    case TFX_OP_GATHER_F32:
      return tfx_gather_f32((const float*)r->p0, (const int32_t*)r->p1, (float*)r->p2, (int32_t)r->i0, s);
    case TFX_OP_ONEHOT_BF16:
      return tfx_onehot_bf16((const int32_t*)r->p0, (const int32_t*)r->p1, (tfx_bf16*)r->p2, (int32_t)r->i0, (int32_t)r->i1, s);
    case TFX_OP_SCATTER_ROWS_BF16:
      return tfx_scatter_rows_bf16((const tfx_bf16*)r->p0, (int32_t)r->i0, (int32_t)r->i1, (tfx_bf16*)r->p1, (int32_t)r->i2,
                                   (const int32_t*)r->p2, (int32_t)r->i3, s);
    case TFX_OP_F32_TO_BF16:
      return tfx_f32_to_bf16((const float*)r->p0, (tfx_bf16*)r->p1, r->i0, s);
    case TFX_OP_SILU_BWD:
      return tfx_silu_bwd((const tfx_bf16*)r->p0, (const tfx_bf16*)r->p1, (tfx_bf16*)r->p2, r->i0, s);
    case TFX_OP_COLSUM_BF16:
      return tfx_colsum_bf16((const tfx_bf16*)r->p0, (int32_t)r->i0, (int32_t)r->i1, (int32_t)r->i2, (const int32_t*)r->p1,
                             (const int32_t*)r->p2, (float*)r->p3, s);
    case TFX_OP_COLSUM_F32:
      return tfx_colsum_f32((const float*)r->p0, (int32_t)r->i0, (int32_t)r->i1, (int32_t)r->i2, (float*)r->p1, s);
    case TFX_OP_ADD_BF16:
      return tfx_add_bf16((const tfx_bf16*)r->p0, (const tfx_bf16*)r->p1, (tfx_bf16*)r->p2, r->i0, s);
    default: return -100;          // unknown op
  }
}

}  // namespace

extern "C" int tfx_run_list(const tfx_launch* list, int32_t n, void* stream, int32_t* failed_at) {
  if (n < 0 || (n > 0 && !list)) return -1;
  for (int32_t i = 0; i < n; ++i) {
    if (!list[i].args) { if (failed_at) *failed_at = i; return -2; }
    int rc = run_one(list[i], stream);
    if (rc != 0) { if (failed_at) *failed_at = i; return rc; }
  }
  return 0;
}
