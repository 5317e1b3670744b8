/* tfx.h — C ABI of libtfx_hip.so: the MI355X (gfx950) Transfusion hot path.
 *
 * Plain C: raw device pointers, sizes, a hipStream_t passed as void*.  No torch types.
 * Every entry point enqueues work on `stream` and returns immediately (no host sync);
 * return value 0 = ok, non-zero = argument / launch error (hipError_t or negative code).
 * bf16 tensors are `uint16_t*` storage here (`__bf16` on the device).
 *
 * Reference interfaces replaced (file:line relative to /root/reference/transfusion_pytorch;
 * T = transfusion.py, MP = modality_processing.py):
 *
 *   tfx_gemm_nt / tfx_gemm_tn    every nn.Linear on the path and its backward: to_qk/to_v/to_gates/to_out
 *                                T:877-916, FeedForward T:845-853 (GEGLU epilogue T:831-834), skip_proj
 *                                T:1083,1214-1219, to_time_cond T:1068-1072, to_film/to_ada_ln_zero T:664-665,
 *                                latent_to_model / model_to_latent T:1478-1479, to_text_logits T:1507
 *   tfx_attn_fwd/_bwd_*          Attention.forward score pipeline T:998-1027 (softclamp T:280-281, mask
 *                                naive_attn_mask T:452-470 in prefix-extension form, sigmoid value gate T:1027)
 *   tfx_qk_norm_rope_*           q_norm/k_norm RMSNorm T:950-952,779-786 + apply_rotary_emb T:965 + q*scale T:998
 *   tfx_adaln_pre_* / _post_*    AdaptiveWrapper.forward T:721-775
 *   tfx_attnres_*                AttentionResidual.forward T:807-829
 *   tfx_rmsnorm_*                RMSNorm T:779-786 (final norm T:1250)
 *   tfx_embed_*                  text_embed gather + einx.where select T:3173-3184
 *   tfx_noise_mix                process_type_flat noising MP:654-656
 *   tfx_fourier                  RandomFourierEmbed T:617-635
 *   tfx_ce_* / tfx_mse_*         loss block T:3320-3376
 *   tfx_adam_*                   train_toy.py:55-57 (clip_grad_norm_ + Adam)
 */
#ifndef TFX_H
#define TFX_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#ifndef TFX_BF16_DEFINED
typedef uint16_t tfx_bf16;
#endif

/* ---- GEMM ------------------------------------------------------------------------------------ */
enum { TFX_EPI_BF16 = 0, TFX_EPI_F32 = 1, TFX_EPI_SILU = 2, TFX_EPI_RESID = 3, TFX_EPI_GEGLU = 4, TFX_EPI_GEGLU_BWD = 5, TFX_EPI_QKV_NORM_ROPE = 6 };

/* C[M,N] = A[M,K] . B[N,K]^T (+ epilogue).  K % 64 == 0; lda/ldb % 8 == 0; all bases 16-byte aligned.
 * Optional second A source A2 for k >= K1 (skip-proj concat without the cat copy, T:1214-1219).
 * a_rowmap: gather A rows (row index = a_rowmap[m]); rowmap: scatter output rows (negative = drop).
 * GEGLU layout: physical column c of the 2*dip-wide [value|gate] buffer: block j = c/64, c%64 < 32 is
 * value feature j*32 + c%32, else gate feature j*32 + c%64-32 (weights/bias shadows use the same order).
 * TFX_EPI_GEGLU (FeedForward T:845-853, GEGLU T:831-834): with a | g = acc + bias in that layout, C2 (ldc2 = dip) receives h = a gelu(g) and C (ldc = 2 dip)
 * receives what the backward needs, in the same layout: u = gelu(g) in the value slots and v = a gelu'(g) in the gate slots (round 5; the round-1..4 form
 * stored the pre-activations a | g).  TFX_EPI_GEGLU_BWD: acc = dh (N = dip), aux = that saved [u|v]; C (ldc = 2 dip) receives d[a|g] = dh u | dh v. */
typedef struct {
  const tfx_bf16* A; int32_t lda;
  const tfx_bf16* A2; int32_t lda2; int32_t K1;
  const tfx_bf16* B; int32_t ldb;
  int32_t M, N, K;
  int32_t epi;
  void* C; int32_t ldc;
  void* C2; int32_t ldc2;
  const float* bias;
  const tfx_bf16* R; int32_t ldr; int32_t resid_mapped;   /* residual row = scattered row if set */
  const int32_t* rowmap;
  const int32_t* a_rowmap;
  const tfx_bf16* aux; int32_t ldaux;
  /* TFX_EPI_QKV_NORM_ROPE (round 4; SURVEY K4, reference T:946-965): the fused [q | k | v | gates] projection.  C (bf16, ldc) receives the raw
   * projection as with TFX_EPI_BF16 (the backward of the norm reads it); C2 (bf16, ldc2 >= 2 qk_heads 64) receives q~ | k~ = QK-RMSNorm + RoPE
   * (+ the q scale) of the first 2 qk_heads 64 columns - exactly what tfx_qk_norm_rope_fwd computes from C, bit for bit, without the launch and
   * without reading q, k back.  The qk_* fields mirror tfx_qk_norm_rope_args (gammas [64], rot_pos [M], cos / sin tables [P, 32], scales, optional
   * soft-cap plan).  Decode steps: qk_cache (bf16 rows of ld_cache = [k~ (qk_heads 64) | v (qk_heads 64)]) + qk_cache_pos [M] append this step's
   * k~ and v rows at row qk_cache_pos[m] (< 0: skip) - the KV-cache append of tfx_qk_norm_rope_args.cache in the same epilogue (T:1005-1016).
   * Fused in the 256 x 256 kernel (training shapes) and in the decode-step kernel (M <= 1024); every other shape runs as the plain projection
   * followed by tfx_qk_norm_rope_fwd inside the call - same results bit for bit. */
  int32_t qk_heads;
  const float* qk_gamma_q; const float* qk_gamma_k;
  const int32_t* qk_rot_pos; const float* qk_cos; const float* qk_sin;
  float qk_q_scale, qk_norm_scale;
  float* qk_plan; float qk_softcap;
  tfx_bf16* qk_cache; int32_t qk_ld_cache; const int32_t* qk_cache_pos;
  /* Cold-operand prefetch (round 6, decode plans): `prefetch_bytes` bytes at `prefetch` - the weights of the NEXT GEMM of a launch list - are touched by
   * spare blocks of THIS launch (one load per 64 bytes, nothing written), so that they wait in the 256 MB Infinity Cache when their launch comes.  A decode
   * forward streams 25 MB of weights per layer through launches of a few dozen blocks whose time is a chain of memory round trips (tools/decode_cold_probe.py:
   * 10.1 us cold against 6.7 us warm for the out-projection at 64 rows).  Honoured by the small-M kernels (fewer than 512 tiles of 256 x 256), ignored when
   * null / 0.  Results do not depend on it. */
  const void* prefetch; int64_t prefetch_bytes;
} tfx_gemm_nt_args;
int tfx_gemm_nt(const tfx_gemm_nt_args* a, void* stream);
/* which kernel tfx_gemm_nt would launch for these arguments and on how many blocks, without launching (host logic only, no device needed):
 * 0 register-staged fallback, 1 LDS-DMA 128 x 128, 2 "mid" (<= one 128 x 128 tile per CU), 3 ping-pong 256 x 256 (8 waves: the fused epilogues, row-gathered /
 * split A), 4 skinny (M <= 512 ... 1024), 5 decode (M <= 1024, K split across the waves), 6 one wave per SIMD 256 x 256, one tile per block (fp32 outputs from
 * K = 1024, bf16 outputs with K < 192), 7 one wave per SIMD, persistent (bf16 outputs, K >= 192).  Kinds 6 / 7 since round 6 (they reported 3 before). */
int tfx_gemm_nt_plan(const tfx_gemm_nt_args* a, int32_t* kind, int32_t* grid);

/* Measurement aid (bench.py `roofline.measured_peak`; no reference counterpart): a register-resident loop of `iters` x 4 v_mfma_f32_32x32x16_bf16 per wave on
 * `blocks` blocks of 4 waves, operands = the 128 x 16 bytes at `ops` (lane-indexed; random bf16 for the power the data toggles cost, zeros read ~20 % higher),
 * no memory traffic inside the loop.  flops = 2 x 32 x 32 x 16 x 4 x iters x blocks x 4; the caller times the launch.  `out`: 2 floats of scratch. */
int tfx_mfma_peak_probe(const void* ops, float* out, int32_t iters, int32_t blocks, void* stream);

/* C[rowmap[n]][k] += alpha * sum_m A[m][n] * B[m][k]   (fp32 C, ALWAYS accumulates: split-M partial sums are
 * added with fp32 atomics, the caller zeroes C when it wants a plain product; `accumulate` is ignored).
 * a_cols/b_cols: number of readable columns of A/B (multiples of 8); k_valid: columns of the product that are written.
 * splits: number of M chunks (of a multiple of 64 rows) summed through the atomics; 1 = one block per output tile over all rows;
 * 0 = the library picks the count that fills the chip with the tiles of the kernel it launches. */
typedef struct {
  const tfx_bf16* A; int32_t lda; int32_t a_cols;
  const tfx_bf16* B; int32_t ldb; int32_t b_cols;
  int32_t M, N, K;
  float* C; int32_t ldc;
  const int32_t* rowmap;
  int32_t k_valid;
  int32_t splits;
  int32_t accumulate;
  float alpha;
  const int32_t* a_rowmap;    /* gather rows of A / B (row index = map[m]); NULL = identity */
  const int32_t* b_rowmap;
  float* colsum;              /* optional: colsum[rowmap[n]] += sum_m A[m][n] - the bias gradient that belongs to this weight gradient,
                                 folded into the GEMM (one extra MFMA against a ones operand in the blocks of the first K tile) */
  int32_t k_group;            /* 0 = off; else the K columns come in groups of 64 of which the first k_group are written, compacted:
                                 C column = (k / 64) * k_group + k % 64  (per-head padded operand -> unpadded weight gradient) */
  /* grouped launch (round 5): HOST pointer to the tfx_gemm_tn_args of another product over the same M rows (its own group_next continues the chain; at most 6
   * products; `splits` of the chain's head applies).  The weight-gradient products of a transformer layer - the FeedForward pair, to_out, to_qk/v/gates, the skip
   * projection's halves - then run as ONE launch whose output tiles fill the chip at 4 row chunks instead of 11-20 per product: half the fp32 atomics of the
   * split-M sums (which the chip retires at ~1.25 TB/s: 27-42 % of the ungrouped kernels) and 256 x 256 tiles for the 512 x 512 products.  Products the one-wave
   * kernel does not take (row-gathered operands, M % 64 != 0, chunks under 192 rows) make the library run the chain one by one - same results.  NULL = single. */
  const void* group_next;
} tfx_gemm_tn_args;
int tfx_gemm_tn(const tfx_gemm_tn_args* a, void* stream);
/* what tfx_gemm_tn would launch for these arguments, without launching (host logic only, no device needed): kernel form (-1 register-staged
 * fallback, 0 = 128 x 128 tiles / 4 waves, 2 = 256 x 256 / 8 waves, 3 = 256 x 256 / 4 waves of 128 x 128 (one wave per SIMD, round 5)), output tiles,
 * row chunks (`splits`, chosen when a->splits == 0), grid.  For the head of a `group_next` chain that runs as one launch: kind 3 with the chain's summed tiles, its
 * chunk count and grid; a chain that runs product by product reports the head's own plan. */
int tfx_gemm_tn_plan(const tfx_gemm_tn_args* a, int32_t* kind, int32_t* tiles, int32_t* splits, int32_t* grid);

/* ---- attention ------------------------------------------------------------------------------- */
/* q,k,v: bf16, token-major; element (token t, head h, i) at ptr[t*ld + h*64 + i]; dim_head == 64.
 * kv_end[t] (per token, relative to its sample): key j visible iff j < kv_end.  q_start[j]: first query
 * position that sees key j.  gate: logits at gate[t*ld_gate + h].  out: gated output og (bf16). */
typedef struct {
  const tfx_bf16 *q, *k, *v; int32_t ld_q, ld_k, ld_v;
  const tfx_bf16* gate; int32_t ld_gate;
  const int32_t* kv_end;
  const int32_t* q_start;
  tfx_bf16* out; int32_t ld_out;
  float* lse;                 /* [b, h, n] */
  int32_t b, h, n;
  float softcap;
  int32_t n_kv;               /* forward only: keys/values have n_kv rows per sample (KV-cache decode); 0 = n */
  /* backward */
  const tfx_bf16* dout; int32_t ld_dout;     /* grad wrt gated output */
  tfx_bf16* do_eff; int32_t ld_do;           /* scratch: dout * sigmoid(gate) */
  float* delta;                               /* [b, h, n] */
  tfx_bf16* dgate; int32_t ld_dgate;         /* grad wrt gate logits */
  tfx_bf16 *dq, *dk, *dv; int32_t ld_dq, ld_dk, ld_dv;
  int32_t order;              /* set by the library (block order of the launch); callers leave it 0 */
  const float* sc_plan;       /* optional, forward and backward: the layer's soft-cap plan (tfx_qk_norm_rope_args.sc_plan); NULL = decide from the scores.
                               * PRECONDITION: a plan of mode 0 / 1 is a polynomial fitted on [-B, B] and the kernels do NOT check the scores
                               * against B - every q and k row of this call (KV-cache rows included) must have been produced by tfx_qk_norm_rope_fwd (or the
                               * fused projection epilogue) under the SAME gammas / scales the plan was written from.  Rows from other gains, or without
                               * QK-norm, make the polynomial diverge silently outside [-B, B]: pass NULL for such inputs.  tfx_decode_attn (the
                               * matrix-core-free decode kernel) ignores the plan and evaluates the data-dependent form. */
  /* forward with a KV cache, optional (compacted decode steps): sample s of the launch owns the query rows q_row0[s] .. q_row0[s] + q_cnt[s] - 1 of the
   * token arrays (q, gate, kv_end, out) instead of rows s * n .. (`n` stays the per-sample maximum and sizes the grid; q_cnt[s] = 0: nothing to do);
   * keys / values are still the sample's n_kv cache rows.  Device arrays of `b` int32 each; NULL = the dense layout. */
  const int32_t* q_row0; const int32_t* q_cnt;
  /* backward, optional (round 5): the backward of QK-RMSNorm + RoPE (+ the q scale; reference T:950-952, T:965, T:998 backwards) in the epilogues of the dQ and
   * dK/dV kernels.  With nr_qkv set, the kernels do not write d q~ / d k~ to dq / dk: each wave takes its 32 x 64 block through the staging image it stores
   * from anyway - 8 lanes per row, 8 contiguous columns each, the token-wise kernel's own shape - loads the raw (pre-norm) q / k chunk, applies
   * tfx_qk_norm_rope_bwd's arithmetic to the bf16-rounded d q~ / d k~ and writes d q | d k (raw) to nr_dqkv (q at column h*64, k at column h_total*64 + h*64,
   * the layout of tfx_qk_norm_rope_args.dqkv); the gain gradients accumulate per block and leave as 64 atomics per block and kernel.  One write + one read of
   * the [T, 2 h 64] d q~ | d k~ matrix and the tfx_qk_norm_rope_bwd launch drop out of a layer's backward.  All nr_* leading dimensions % 8 == 0, 16-byte
   * aligned bases; nr_norm_scale 0 = 8. */
  const tfx_bf16* nr_qkv; int32_t nr_ld_qkv;
  tfx_bf16* nr_dqkv; int32_t nr_ld_dqkv;
  const float* nr_gamma_q; const float* nr_gamma_k;
  const int32_t* nr_rot_pos; const float* nr_cos; const float* nr_sin;
  float nr_q_scale, nr_norm_scale;
  float* nr_dgamma_q; float* nr_dgamma_k;
} tfx_attn_args;
int tfx_attn_fwd(const tfx_attn_args* a, void* stream);
int tfx_attn_bwd(const tfx_attn_args* a, void* stream);   /* prep + dK/dV kernel + dQ kernel */

/* decode step (reference T:2279-2349, T:2409-2419): the forward kernel addressed against a KV cache - `n` NEW query rows per
 * sample, keys / values in a per-sample buffer of `n_kv` > 0 rows, kv_end[t] = visible cache length of query row t (own prefix + the block
 * being decoded).  Fails with -10 when n_kv == 0. */
int tfx_decode_attn(const tfx_attn_args* a, void* stream);

/* ---- token-wise kernels ---------------------------------------------------------------------- */
typedef struct {
  int32_t T, d;
  const tfx_bf16* x; tfx_bf16* u;            /* [T,d] */
  const int32_t* tok_inst;                    /* [T] instance id or -1 (text) */
  const float* table; int32_t ld_table;       /* fp32 [I, ld]; this wrapper's gamma at +0, beta at +d, z at +2d */
  const float* gamma_text;                    /* layernorm_gamma [d] */
  float* mean; float* rstd;                   /* [T] saved stats */
  /* backward */
  const tfx_bf16* du; tfx_bf16* dx;           /* dx += LN backward (in place accumulate) */
  float* dtable;                              /* fp32 [I, ld] (atomic accumulate) */
  float* dgamma_text;                         /* [d] (atomic accumulate) */
  /* optional segment mode (backward): runs of consecutive tokens sharing one tok_inst value; one wave owns a
   * segment, so an instance's FiLM gradients are reduced in registers and STORED (no atomics).  NULL or n_seg == 0 = per-token atomics */
  const int32_t* seg_start; const int32_t* seg_len; int32_t n_seg;
  const tfx_bf16* dx_add;                     /* backward, optional [T,d]: one more addend of dx (the gradient a U-Net skip hands to this layer's input) */
} tfx_adaln_pre_args;
int tfx_adaln_pre_fwd(const tfx_adaln_pre_args* a, void* stream);
int tfx_adaln_pre_bwd(const tfx_adaln_pre_args* a, void* stream);

typedef struct {
  int32_t T, d;
  const tfx_bf16* x; const tfx_bf16* y; tfx_bf16* out;   /* out = x + y * scale */
  const int32_t* tok_inst;
  const float* table; int32_t ld_table;       /* z logits at table[inst*ld + 2d + col] */
  const float* layerscale;                    /* [d] */
  /* backward: g = grad wrt out (also the residual grad, passed through untouched) */
  const tfx_bf16* g; tfx_bf16* dy;
  float* dtable; float* dlayerscale;
  const int32_t* seg_start; const int32_t* seg_len; int32_t n_seg;   /* optional segment mode, see tfx_adaln_pre_args */
  float* dbias;                               /* optional [d]: += column sums of dy (the bias gradient of the Linear that produced y) */
} tfx_adaln_post_args;
int tfx_adaln_post_fwd(const tfx_adaln_post_args* a, void* stream);
int tfx_adaln_post_bwd(const tfx_adaln_post_args* a, void* stream);

typedef struct {
  int32_t T, H;                               /* kernel layout: 64 columns per head; dim_head < 64 = zero columns past dim_head */
  const tfx_bf16* qkv; int32_t ld_qkv;        /* q at col h*64, k at col H*64 + h*64 (pre-norm) */
  tfx_bf16* qk; int32_t ld_qk;                /* post norm+rope(+q scale), same column layout */
  const float* gamma_q; const float* gamma_k; /* [64] */
  const int32_t* rot_pos;                     /* [T] */
  const float* cos_tab; const float* sin_tab; /* [P, 32] */
  float q_scale;
  /* backward */
  const tfx_bf16* dqk; int32_t ld_dqk;        /* grad wrt post-norm q,k */
  tfx_bf16* dqkv; int32_t ld_dqkv;            /* grad wrt pre-norm q,k (written) */
  float* dgamma_q; float* dgamma_k;           /* atomic accumulate */
  float norm_scale;                           /* sqrt(dim_head), the RMSNorm scale T:779-786; 0 = 8 (dim_head 64) */
  /* forward, decode steps (optional): KV-cache append fused in (T:1005-1016) - token t's k~ (H*64 columns) and v (the H*64 columns behind k in
   * `qkv`) are also written to cache row cache_pos[t] ([k~ | v], ld_cache elements per row; cache_pos < 0 = skip).  A decode step is launch-bound
   * (~4.5 us per kernel whatever its size): this removes two launches per layer */
  tfx_bf16* cache; int32_t ld_cache; const int32_t* cache_pos;
  /* forward, optional: the layer's soft-cap plan for the attention kernels (`tfx_attn_args.sc_plan`), 8 floats written by block 0.  QK-RMSNorm bounds
   * the scores, |q~ . k~| <= B = norm_scale^2 q_scale max|1 + gamma_q| max|1 + gamma_k| (x 1.02 for the bf16 rounding of q~, k~), so the degree of the
   * polynomial that replaces cap tanh(s / cap) is a property of the LAYER, not of the data: the attention kernels need not look at their scores to pick
   * it.  plan = {mode, p1, p3, p5, d1, d3, d5, B}: mode 0 (B / cap <= 0.2) cubic, mode 1 (<= 0.35) quintic - Chebyshev-economised on [-B, B],
   * s2 = s (p1 + p3 s^2 + p5 s^4) in the log2 domain and d tanh / dx = d1 + d3 s^2 + d5 s^4 its exact derivative; mode 2: the kernels decide
   * per wave from the scores as without a plan.  `softcap` must be the attention's. */
  float* sc_plan; float softcap;
} tfx_qk_norm_rope_args;
int tfx_qk_norm_rope_fwd(const tfx_qk_norm_rope_args* a, void* stream);
int tfx_qk_norm_rope_bwd(const tfx_qk_norm_rope_args* a, void* stream);
/* decode steps: AdaLN output side of one wrapper + AdaLN input side of the next in ONE launch (out = x + y * scale is written, then normalised and
 * modulated into `pre->u` exactly as tfx_adaln_pre_fwd would read it back: same bf16 rounding of `out`).  post->T/d/tok_inst must equal pre's. */
int tfx_adaln_post_pre_fwd(const tfx_adaln_post_args* post, const tfx_adaln_pre_args* pre, void* stream);

typedef struct {
  int32_t T, d, L;                            /* L = number of hiddens h_0..h_{L-1} */
  const tfx_bf16* hiddens; int64_t stride_h;  /* hiddens + l*stride_h -> [T,d] */
  const float* gamma; const float* pq;        /* norm_keys.gamma, pseudo_queries [d] */
  tfx_bf16* out;
  /* backward */
  const tfx_bf16* g; const tfx_bf16* g2;      /* grad wrt out (+ optional second addend) */
  tfx_bf16* dhiddens; int64_t stride_dh;      /* accumulate (or store if `first`) */
  int32_t first;
  float* dgamma; float* dpq;                  /* atomic accumulate */
  /* forward, optional (training plans): the depth softmax of every token is kept for the pull-form backward (tfx_attnres_pull_bwd):
   * save[(t * L + l) * 4 + {0, 1, 2}] = softmax weight a_l, 1 / |h_l|, score s_l = <h_l, w> / |h_l| */
  float* save;
  tfx_bf16* err;                              /* with `save`: [T,d] what the bf16 rounding of `out` dropped (exact mix - stored out), for <g, out> in the backward */
} tfx_attnres_args;
int tfx_attnres_fwd(const tfx_attnres_args* a, void* stream);
int tfx_attnres_bwd(const tfx_attnres_args* a, void* stream);
/* decode steps: the END of a layer in one launch - feed-forward AdaLN output side (writes hidden L - 1 = post->out), the layer's AttentionResidual
 * over hiddens 0 .. L - 1 (writes ar->out), and - when `pre` is non-NULL - the input side of the NEXT layer's attention wrapper on that output.
 * Same roundings as the three separate launches (every written row is re-read as bf16).  post->out must be hidden L - 1 of `ar`, pre->x == ar->out. */
int tfx_layer_end_fwd(const tfx_adaln_post_args* post, const tfx_attnres_args* ar, const tfx_adaln_pre_args* pre, void* stream);

/* AttentionResidual backward in PULL form (T:807-829; the training plans' form).  Hidden l feeds the AttentionResidual of every layer
 * j >= l - 1, so its gradient is   dh_l = sum_j [ a_jl g_j + k1_jl w_j - k2_jl h_l ],   k1 = ds inv_l, k2 = k1 s_jl inv_l,
 * ds_jl = a_jl (<g_j, h_l> - <g_j, out_j>)   (sum_l a_jl <g_j, h_l> = <g_j, out_j>: the forward output is the softmax mix of the hiddens).
 * The push form (tfx_attnres_bwd, one launch per layer) reads every earlier hidden and read-modify-writes every earlier gradient once per
 * layer: (3L + 2) passes of [T, d] per layer.  The pull form computes dh_l ONCE, when the backward reaches layer l - 1: it reads h_l and the
 * n_src output gradients g_j (all final by then), plus three saved scalars per (j, token) - (n_src + 2) passes, about 0.6x the bytes at
 * depth 8 and 0.5x at depth 24.  d w_j = sum_t sum_l k1_jl h_l accumulates in registers (n_src <= 8 at d <= 512) or in LDS, per launch, into
 * src[j].dw; tfx_attnres_finish turns it into d gamma / d pseudo_queries.
 * With `post` the feed-forward wrapper's output side (tfx_adaln_post_bwd on g = the dh_l just written) runs in the same launch. */
typedef struct {
  const tfx_bf16* g;                          /* [T,d] gradient wrt the output of this layer's AttentionResidual (final) */
  float* save;                                /* [T, L, 4] its saved forward state (tfx_attnres_args.save); entry [t][0][3] receives <g, out> in the backward */
  float* dsum;                                /* [T] <g, out>: written by the launch that lists this layer first with `out_own`, read by later ones */
  float* w; float* dw;                        /* [d] fp32: (1 + gamma) * pseudo_queries (tfx_attnres_prep); its gradient accumulator (atomics) */
  const float* gamma; const float* pq;        /* norm_keys.gamma, pseudo_queries [d] */
  float* dgamma; float* dpq;                  /* their gradients (tfx_attnres_finish, accumulate) */
  int32_t L; int32_t reserved;                /* hiddens this layer mixes */
} tfx_attnres_src;
typedef struct {
  int32_t T, d, l, n_src;                     /* hidden index l; 1 <= n_src <= 32 layers read it */
  const tfx_bf16* h;                          /* [T,d] hidden l */
  const tfx_attnres_src* src;                 /* DEVICE array [n_src], lowest layer first */
  const tfx_bf16* out_own;                    /* optional [T,d]: the forward output of src[0]'s AttentionResidual - its dsum is formed here */
  const tfx_bf16* out_err;                    /* optional [T,d]: what rounding that output to bf16 dropped (tfx_attnres_args.err) */
  const tfx_bf16* add;                        /* optional [T,d] addend (gradient that reaches the hidden directly) */
  tfx_bf16* dh;                               /* [T,d] result (stored) */
  const int32_t* seg_start; const int32_t* seg_len; int32_t n_seg;   /* token segments (tfx_adaln_pre_args); n_seg == 0: one token per wave */
  /* d w of the sources: with k1 == NULL (allowed for n_src <= 8, d <= 512) it accumulates inside the launch into src[j].dw; otherwise the launch
   * only writes k1[t * ld_k1 + j] = the per-token coefficient of source j (bf16, ld_k1 >= n_src, a multiple of 8) and the caller adds
   * dw[j] += sum_t k1[t][j] h[t]  with one tfx_gemm_tn (A = k1, B = h, C = src[0].dw, ldc = d: the dw rows of consecutive sources are contiguous) */
  tfx_bf16* k1; int32_t ld_k1;
} tfx_attnres_pull_args;
int tfx_attnres_prep(const tfx_attnres_src* src_dev, int32_t n, int32_t d, void* stream);     /* w = (1 + gamma) pq ; dw = 0, for n layers */
int tfx_attnres_pull_bwd(const tfx_attnres_pull_args* a, const tfx_adaln_post_args* post, void* stream);
int tfx_attnres_finish(const tfx_attnres_src* src_dev, int32_t n, int32_t d, void* stream);   /* dgamma += dw pq ; dpq += dw (1 + gamma) */
/* backward of two adjacent wrapper sides in one launch: tfx_adaln_pre_bwd(pre) (dx += LN backward, in place) followed by
 * tfx_adaln_post_bwd(post) with post->g == pre->dx: the residual-gradient row is written once and not read back.  Segment mode only
 * (pre->n_seg > 0, same segments); bit-identical to the two launches. */
int tfx_adaln_pre_post_bwd(const tfx_adaln_pre_args* pre, const tfx_adaln_post_args* post, void* stream);

typedef struct {
  int32_t T, d;
  const tfx_bf16* x; tfx_bf16* y; const float* gamma;
  const tfx_bf16* dy; tfx_bf16* dx; float* dgamma;
} tfx_rmsnorm_args;
int tfx_rmsnorm_fwd(const tfx_rmsnorm_args* a, void* stream);
int tfx_rmsnorm_bwd(const tfx_rmsnorm_args* a, void* stream);

/* text embedding gather into the token buffer (only tokens with tok_inst < 0), and its backward */
typedef struct {
  int32_t T, d;
  const int32_t* text_ids; const int32_t* tok_inst;
  const tfx_bf16* table;                      /* bf16 shadow of text_embed.weight [V, d] */
  tfx_bf16* x;
  const tfx_bf16* dx; float* dtable;          /* backward: fp32 [V, d] atomic accumulate */
} tfx_embed_args;
int tfx_embed_fwd(const tfx_embed_args* a, void* stream);
int tfx_embed_bwd(const tfx_embed_args* a, void* stream);

/* x_t = t*x + (1-t)*eps (bf16, padded to ld_xt with zeros), flow = x - eps (fp32)   MP:654-656 */
typedef struct {
  int32_t R, dl;
  const float* x; const float* eps;           /* [R, dl] */
  const int32_t* row_inst; const float* inst_time;
  tfx_bf16* xt; int32_t ld_xt;
  float* flow;                                /* [R, dl] (may be NULL) */
} tfx_noise_mix_args;
int tfx_noise_mix(const tfx_noise_mix_args* a, void* stream);

/* e[i,:] = [t, sin(2 pi t w), cos(2 pi t w)] bf16, zero padded to ld   T:617-635 */
typedef struct { int32_t I, half; const float* times; const float* w; tfx_bf16* out; int32_t ld; } tfx_fourier_args;
int tfx_fourier(const tfx_fourier_args* a, void* stream);

/* fused cross-entropy forward + backward over fp32 logits [T, ld] (V valid columns)   T:3320-3331 */
typedef struct {
  int32_t T, V; const float* logits; int32_t ld;
  const int32_t* labels;                      /* -1 = ignore */
  float grad_scale;                           /* text_loss_weight / total_tokens */
  tfx_bf16* dlogits; int32_t ld_d;            /* bf16 [T, ld_d], pad columns zeroed */
  float* acc;                                 /* acc[0] += sum of token CE, acc[1] += #valid */
} tfx_ce_args;
int tfx_ce_fwd_bwd(const tfx_ce_args* a, void* stream);

/* fused MSE forward + backward: pred/flow fp32 [R, dl]   T:3359-3362 */
typedef struct {
  int32_t R, dl; const float* pred; int32_t ld_pred; const float* flow;
  float grad_scale;                           /* 2 * weight / (R*dl) */
  tfx_bf16* dpred; int32_t ld_d;              /* bf16 [R, ld_d], pad columns zeroed */
  float* acc;                                 /* acc[0] += sum of squared error */
  int32_t accumulate;                         /* != 0: dpred += gradient (a second target on the same prediction: the
                                                 velocity-consistency term T:3394-3418), else dpred = gradient */
  /* model_output_clean (T:1297, MP:100-126): `pred` already holds (out - noised) / max(1 - t, clean_eps) (tfx_output_to_flow);
   * the gradient wrt the model output carries the same 1 / max(1 - t, clean_eps).  row_inst NULL = off */
  const int32_t* row_inst; const float* inst_time; float clean_eps;
  /* reconstruction loss (`reconstruction_loss_weight`, MP:177-200 / T:2840-2853): recon_w non-NULL switches the residual to
   *   (1 - t) pred - c flow,   t = recon_time[recon_inst[row]],   c = t (recon_mode 0: target = the noised latent, interleaved forward)
   *                                                               c = 1 (recon_mode 1: target = the clean latent, forward_modality)
   * - algebraically noised - (noise + pred (1 - t)) resp. clean - (noise + pred (1 - t)) with flow = clean - noise - and weights every row by
   * recon_w[row] (1 / (instances of the type x rows of the instance): the reference averages per-instance means); acc += sum w r^2,
   * d pred (+)= grad_scale w (1 - t) r */
  const float* recon_w; const int32_t* recon_inst; const float* recon_time; int32_t recon_mode;
} tfx_mse_args;
int tfx_mse_fwd_bwd(const tfx_mse_args* a, void* stream);
/* model_output_clean: pred[r][c] <- (pred[r][c] - noised[r][c]) / max(1 - t_r, clean_eps), noised = eps ? x*t + eps*(1-t) : x,
 * t_r = inst_time[row_inst[r]]   (get_model_output_to_flow_fn, MP:100-126) */
int tfx_output_to_flow(float* pred, const float* x, const float* eps, const int32_t* row_inst, const float* inst_time,
                       int32_t R, int32_t dl, float clean_eps, void* stream);

/* ---- sampling (sample_many, T:2082-2583) ------------------------------------------------------ */
/* sample_text_token (T:597-605) with min_p_filter (T:591-595), one row of fp32 logits [B, ld] (V valid columns) per sample, on the device:
 * temperature == 0 -> argmax (first index on ties); else p = softmax(logits / temperature), tokens with p < min_p * max p are removed and
 * the survivor whose cumulative mass (index order) crosses uniforms[row] * (surviving mass) is drawn.  Rows with active[row] == 0 are left
 * untouched (active NULL = all rows).  out_ids: int32 [B]. */
int tfx_sample_tokens(const float* logits, int32_t ld, int32_t B, int32_t V, float temperature, float min_p, const float* uniforms,
                      const int32_t* active, int32_t* out_ids, void* stream);
/* the same with the DRAW restricted to the first V_draw columns while the maximum (hence the min-p threshold, and the argmax at temperature 0)
 * still runs over all V: `generate_text_only` (T:2690-2698) filters over every logit, then masks everything but the text tokens */
int tfx_sample_tokens_range(const float* logits, int32_t ld, int32_t B, int32_t V, int32_t V_draw, float temperature, float min_p, const float* uniforms,
                            const int32_t* active, int32_t* out_ids, void* stream);
/* ODE state update of the fixed-grid midpoint solver (torchdiffeq semantics; T:2468-2525) fused with classifier-free
 * guidance (T:2516-2521): f = f_uncond ? f_uncond + cfg_scale * (f_cond - f_uncond) : f_cond;  out = y + a * f   (fp32, n elements) */
int tfx_ode_axpy(const float* y, const float* f_cond, const float* f_uncond, float cfg_scale, float a, float* out, int64_t n, void* stream);
/* The same solver as a per-sample state machine (continuous decode schedule: in one forward every sample is at its OWN evaluation).
 * y, ym: fp32 [B][Lc][dmax] - the state at the start of a solver step and its midpoint; ctl: fp32 [2][B] = per sample (mode, coefficient a):
 *   mode 1  first evaluation of a step:  input y,  result ym = y + a f        mode 3  a finished block re-encoded at t = 1: input y, no update
 *   mode 2  second evaluation:           input ym, result y  = y + a f        mode 0  the sample is not inside a modality
 * tfx_ode_stage writes the evaluation inputs into the plan's latent rows: x[(h B + i) Lq + j][c] = input_i[j][c] (h < H halves, j < Lc, c < dl; x: [H B Lq][dl]).
 * tfx_ode_update applies the results: f = pred[i Lq + j] (H == 1) or, with guidance (H == 2, T:2516-2521), pu + cfg_scale (pc - pu) with
 * pc = pred[i Lq + j], pu = pred[(B + i) Lq + j]; samples with sel[i] == 0 are skipped (sel NULL = all: one call per modality type).
 * rows0 (optional, compacted decode steps): int32 [H B], the first latent row of the block of (half h, sample i) - rows0[h B + i] + j replaces
 * (h B + i) Lq + j in both kernels; a negative entry = that half carries no block this step (nothing written / read there). */
int tfx_ode_stage(const float* y, const float* ym, const float* ctl, int32_t B, int32_t Lc, int32_t dmax, float* x, int32_t H, int32_t Lq, int32_t dl,
                  const int32_t* rows0, void* stream);
int tfx_ode_update(float* y, float* ym, const float* ctl, int32_t B, int32_t Lc, int32_t dmax, const float* pred, int32_t H, int32_t Lq, int32_t dl,
                   float cfg_scale, const float* sel, const int32_t* rows0, void* stream);

/* ---- parameter plumbing ---------------------------------------------------------------------- */
/* dst[r][c] (bf16, ld_dst, Rd rows, Cd cols) = src[rowmap ? rowmap[r] : r][c] or 0 when out of range / map < 0 */
typedef struct { const float* src; int32_t ld_src, Rs, Cs; const int32_t* rowmap; tfx_bf16* dst; int32_t ld_dst, Rd, Cd; } tfx_cast_args;
int tfx_cast_rows(const tfx_cast_args* a, void* stream);
/* dst[c][r] = src[rowmap ? rowmap[r] : r][c]  (transposed shadow for the dX GEMMs) */
int tfx_cast_rows_t(const tfx_cast_args* a, void* stream);
/* all shadows in one launch: `jobs_dev` is a DEVICE array of n_jobs jobs sorted by first_block; job j owns blocks
 * [first_block, first_block + nb) with nb = ceil(Rd*ld_dst/2048) (plain) or ceil(ld_dst/64)*ceil(Rd/64) (transposed);
 * n_blocks is the total; ld_dst % 8 == 0.  Fields as in tfx_cast_args. */
typedef struct { const float* src; int32_t ld_src, Rs, Cs; const int32_t* rowmap; tfx_bf16* dst; int32_t ld_dst, Rd, Cd; int32_t transposed, first_block; } tfx_cast_job;
int tfx_cast_batch(const tfx_cast_job* jobs_dev, int32_t n_jobs, int32_t n_blocks, void* stream);
/* out_f32[i] = gather of fp32 vector through a map (bias shadows): dst[i] = map[i] >= 0 ? src[map[i]] : 0 */
int tfx_gather_f32(const float* src, const int32_t* map, float* dst, int32_t n, void* stream);
/* out[t][c] (bf16, ld % 8 == 0) = 1 if token t is text and max(ids[t],0) == c else 0  (embedding-gradient GEMM operand) */
int tfx_onehot_bf16(const int32_t* ids, const int32_t* tok_inst, tfx_bf16* out, int32_t T, int32_t ld, void* stream);
/* dst[rowmap[r]][0:cols] = src[r][0:cols] for rowmap[r] >= 0   (KV-cache append; bf16, cols % 8 == 0) */
int tfx_scatter_rows_bf16(const tfx_bf16* src, int32_t ld_src, int32_t cols, tfx_bf16* dst, int32_t ld_dst, const int32_t* rowmap, int32_t R, void* stream);
int tfx_f32_to_bf16(const float* src, tfx_bf16* dst, int64_t n, void* stream);
/* dst[r][c] (bf16, row stride ld_dst) = src[r][c] (fp32, row stride ld_src) for an R x C block inside larger matrices (C, both strides and both base
 * offsets multiples of 8): the per-layer slice of the AdaLN table gradients, cast as soon as that layer's backward has finished */
int tfx_cast_block_bf16(const float* src, int32_t ld_src, tfx_bf16* dst, int32_t ld_dst, int32_t R, int32_t C, void* stream);
/* dst(bf16) = a(bf16) * silu'(pre(bf16))  (time-MLP backward) */
int tfx_silu_bwd(const tfx_bf16* dy, const tfx_bf16* pre, tfx_bf16* dx, int64_t n, void* stream);
/* column sums: out[c] += sum_r src[r][c]  (bias gradients); src bf16 or fp32 */
int tfx_colsum_bf16(const tfx_bf16* src, int32_t ld, int32_t R, int32_t C, const int32_t* colmap, const int32_t* rowmap, float* out, void* stream);
int tfx_colsum_f32(const float* src, int32_t ld, int32_t R, int32_t C, float* out, void* stream);
int tfx_add_bf16(const tfx_bf16* a, const tfx_bf16* b, tfx_bf16* out, int64_t n, void* stream);
/* x[i] *= *scale (in place; bf16 values, the fp32 scalar lives ON THE DEVICE so that no host sync is needed to read it).  A scale of
 * exactly 1.0 makes every block return before touching memory: `loss.backward()` with the default upstream gradient costs a launch,
 * not a pass over the loss seeds (autograd hands d total / d loss over as a device tensor, T: train_toy.py:53) */
int tfx_scale_bf16_dev(tfx_bf16* x, int64_t n, const float* scale, void* stream);
/* dst[i] = scale * src[i] (bf16, host scalar): the negated loss seed of the model-space model_output_clean backward */
int tfx_scale_bf16_copy(const tfx_bf16* src, tfx_bf16* dst, int64_t n, float scale, void* stream);

/* fused global-norm clip + Adam over flat fp32 buffers   train_toy.py:55-57
 * sumsq[0] must hold the sum of squared gradients (tfx_sumsq accumulates into it; zero it first). */
int tfx_sumsq(const float* g, int64_t n, float* sumsq, void* stream);
typedef struct {
  float* p; const float* g; float* m; float* v; int64_t n;
  float lr, beta1, beta2, eps, weight_decay, max_norm, grad_scale;
  int32_t step; const float* sumsq;
} tfx_adam_args;
int tfx_adam_step(const tfx_adam_args* a, void* stream);
/* exponential moving average of a flat parameter buffer: ema = decay * ema + (1 - decay) * online   (ema_pytorch EMA.update, used by
 * Transfusion.create_ema T:1681-1699) */
int tfx_ema_update(float* ema, const float* online, int64_t n, float decay, void* stream);

/* ---- data-parallel gradient exchange -----------------------------------
 * ONE all-reduce(sum) per range of the flat fp32 gradient buffer over RCCL / xGMI, one process per GPU (reference practice: DDP under
 * `accelerate`, train_mnist.py:114-126).  RCCL is bound at run time (dlopen), so a host that already carries a copy keeps using it.
 *   rank 0: tfx_allreduce_unique_id(id);  broadcast the 128 bytes by any host channel;  every rank: tfx_allreduce_init(rank, world, id)
 *   per step: tfx_allreduce_run(grad + offset, count, stream)  (in place, enqueued on `stream`; scale by 1 / world in tfx_adam_step.grad_scale) */
int tfx_allreduce_unique_id(void* out128);
int tfx_allreduce_init(int32_t rank, int32_t world, const void* unique_id128);
int tfx_allreduce_run(float* buf, int64_t count, void* stream);
int tfx_allreduce_destroy(void);

/* ---- launch lists ------------------------------------------------------------------------------
 * The step is a STATIC list of launches over persistent buffers (engine.Plan), so the host replays it with ONE call instead of one
 * FFI round trip per kernel: `tfx_run_list` walks `n` items in order on `stream` (or the side stream, per item) and stops at the first non-zero return code
 * (its index is written to *failed_at).  `args` of a struct entry point is that entry point's args struct; entry points with
 * positional arguments take a `tfx_raw_args`: pointers fill p0.., integers i0.., floats f0 in declaration order. */
typedef struct { const void *p0, *p1, *p2, *p3, *p4; int64_t i0, i1, i2, i3; float f0; int32_t reserved; } tfx_raw_args;
enum { TFX_OP_GEMM_NT = 0, TFX_OP_GEMM_TN = 1, TFX_OP_ATTN_FWD = 2, TFX_OP_ATTN_BWD = 3, TFX_OP_ADALN_PRE_FWD = 4, TFX_OP_ADALN_PRE_BWD = 5,
       TFX_OP_ADALN_POST_FWD = 6, TFX_OP_ADALN_POST_BWD = 7, TFX_OP_QK_NORM_ROPE_FWD = 8, TFX_OP_QK_NORM_ROPE_BWD = 9, TFX_OP_ATTNRES_FWD = 10,
       TFX_OP_ATTNRES_BWD = 11, TFX_OP_RMSNORM_FWD = 12, TFX_OP_RMSNORM_BWD = 13, TFX_OP_EMBED_FWD = 14, TFX_OP_EMBED_BWD = 15,
       TFX_OP_NOISE_MIX = 16, TFX_OP_FOURIER = 17, TFX_OP_CE_FWD_BWD = 18, TFX_OP_MSE_FWD_BWD = 19, TFX_OP_CAST_ROWS = 20, TFX_OP_CAST_ROWS_T = 21,
       TFX_OP_ADAM_STEP = 22, TFX_OP_DECODE_ATTN = 23,
       /* positional entry points (args = tfx_raw_args) */
       TFX_OP_OUTPUT_TO_FLOW = 32, TFX_OP_GATHER_F32 = 33, TFX_OP_ONEHOT_BF16 = 34, TFX_OP_SCATTER_ROWS_BF16 = 35, TFX_OP_F32_TO_BF16 = 36,
       TFX_OP_SILU_BWD = 37, TFX_OP_COLSUM_BF16 = 38, TFX_OP_COLSUM_F32 = 39, TFX_OP_ADD_BF16 = 40, TFX_OP_SCALE_BF16_DEV = 41, TFX_OP_CAST_BLOCK_BF16 = 42, TFX_OP_SCALE_BF16_COPY = 43, TFX_OP_ADALN_POST_PRE_FWD = 44, TFX_OP_LAYER_END_FWD = 45,
       TFX_OP_ATTNRES_PREP = 46, TFX_OP_ATTNRES_PULL_BWD = 47, TFX_OP_ATTNRES_FINISH = 52, TFX_OP_ADALN_PRE_POST_BWD = 53,
       /* stream control (args = any non-NULL pointer; `stream` = event slot 0..63):
          FORK: the library's side stream waits for everything enqueued so far on the caller's stream;
          JOIN_RECORD: mark "everything enqueued so far on the side stream";  JOIN_WAIT: the caller's stream waits for that mark;
          JOIN = JOIN_RECORD + JOIN_WAIT */
       TFX_OP_FORK = 48, TFX_OP_JOIN = 49, TFX_OP_JOIN_RECORD = 50, TFX_OP_JOIN_WAIT = 51 };
/* `stream`: 0 = the caller's stream, 1 = the library's side stream (a second HIP stream the weight-gradient GEMMs run on, next to
 * the data-gradient chain); for FORK / JOIN the event slot. */
typedef struct { int32_t op; int32_t stream; const void* args; } tfx_launch;
int tfx_run_list(const tfx_launch* list, int32_t n, void* stream, int32_t* failed_at);
/* hipGraph form of a launch list whose arguments live in DEVICE memory between replays (decode plans: token ids, cache positions, visible
 * lengths, rotary positions, times and latents are device arrays the host overwrites in place; every kernel argument is a fixed pointer or
 * size).  `tfx_graph_create` captures the list on a library-owned stream (nothing executes) and instantiates it; `tfx_graph_launch`
 * replays it on `stream`.  Kernel arguments are frozen at capture time: re-create the graph when an args struct of the list changes. */
int tfx_graph_create(const tfx_launch* list, int32_t n, void** graph_out);
/* the same capture with every item on ONE stream (FORK / JOIN items become no-ops, side-stream items run in list order): the chain form the training
 * lists are captured in (engine.replay_auto, TFX_TRAIN_GRAPH=1).  The single-stream mode holds for this call on this thread only - it does not touch
 * the process-wide switch of tfx_set_single_stream, so other threads' replays and fingerprints are unaffected. */
int tfx_graph_create_single(const tfx_launch* list, int32_t n, void** graph_out);
int tfx_graph_launch(void* graph, void* stream);
int tfx_graph_destroy(void* graph);
/* 64-bit fingerprint of everything a capture of the list would freeze: ops, stream tags, the bytes of every args struct (and of the host structs
 * the two-struct entry points point at), the single-stream switch.  A caller that replays a TRAINING list as a graph compares it with the
 * fingerprint the graph was captured under and re-captures (or replays the list) when a scalar / pointer of the step has changed. */
int tfx_list_fingerprint(const tfx_launch* list, int32_t n, int64_t* out);
/* on != 0: replay every item on the caller's stream (FORK / JOIN become no-ops) - same results, kernels one at a time (used to time a
 * kernel family without its side-stream neighbours); returns the previous setting.  PROCESS-wide: `loss.backward()` replays its list on PyTorch's
 * autograd thread, which must see what the main thread set. */
int tfx_set_single_stream(int32_t on);

const char* tfx_version(void);

#ifdef __cplusplus
}
#endif
#endif
