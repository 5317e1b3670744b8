// Common device helpers for the Transfusion MI355X (gfx950 / CDNA4) kernels.
// wave = 64 lanes; MFMA fragment layouts verified on hardware by tools/probe_layouts.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define TFX_DEV __device__ __forceinline__

TFX_DEV float bf2f(bf16 v) { return (float)v; }
TFX_DEV bf16 f2bf(float v) { return (bf16)v; }   // round-to-nearest-even

// Wave-wide reductions on the VALU's DPP path (no LDS): `__shfl_xor` compiles to ds_bpermute_b32 - an LDS round trip of ~50+ cycles per step,
// six steps per reduction, and the token-wise kernels chain several reductions per token.  Quad swaps, half-row / row mirrors, then
// row_bcast15 / row_bcast31 (gfx9 DPP controls) leave the total in lane 63; v_readlane makes it wave-uniform.
// -DTFX_BPERMUTE_REDUCE keeps the shuffle form (A/B).
#ifndef TFX_BPERMUTE_REDUCE
#define TFX_DPP_STEP(OP, V, CTRL, ROWMASK, IDENT)                                                                     \
  V = OP(V, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (float)(IDENT)),            \
                                                                    __builtin_bit_cast(int, V), CTRL, ROWMASK, 0xf, false)))
TFX_DEV float dpp_add_(float a, float b) { return a + b; }
TFX_DEV float dpp_max_(float a, float b) { return fmaxf(a, b); }
TFX_DEV float wave_sum(float v) {
  TFX_DPP_STEP(dpp_add_, v, 0xB1, 0xf, 0.f);      // quad_perm [1,0,3,2]
  TFX_DPP_STEP(dpp_add_, v, 0x4E, 0xf, 0.f);      // quad_perm [2,3,0,1]
  TFX_DPP_STEP(dpp_add_, v, 0x141, 0xf, 0.f);     // row_half_mirror
  TFX_DPP_STEP(dpp_add_, v, 0x140, 0xf, 0.f);     // row_mirror: every lane of a 16-lane row holds the row sum
  TFX_DPP_STEP(dpp_add_, v, 0x142, 0xa, 0.f);     // row_bcast15 into rows 1, 3
  TFX_DPP_STEP(dpp_add_, v, 0x143, 0xc, 0.f);     // row_bcast31 into rows 2, 3: lane 63 holds the total
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
TFX_DEV float wave_max(float v) {
  TFX_DPP_STEP(dpp_max_, v, 0xB1, 0xf, -INFINITY);
  TFX_DPP_STEP(dpp_max_, v, 0x4E, 0xf, -INFINITY);
  TFX_DPP_STEP(dpp_max_, v, 0x141, 0xf, -INFINITY);
  TFX_DPP_STEP(dpp_max_, v, 0x140, 0xf, -INFINITY);
  TFX_DPP_STEP(dpp_max_, v, 0x142, 0xa, -INFINITY);
  TFX_DPP_STEP(dpp_max_, v, 0x143, 0xc, -INFINITY);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// sum over each aligned group of 8 lanes, in all 8 of them (per-head reductions: 8 lanes x 8 elements = one 64-wide head)
TFX_DEV float group8_sum(float v) {
  TFX_DPP_STEP(dpp_add_, v, 0xB1, 0xf, 0.f);
  TFX_DPP_STEP(dpp_add_, v, 0x4E, 0xf, 0.f);
  TFX_DPP_STEP(dpp_add_, v, 0x141, 0xf, 0.f);
  return v;
}
#undef TFX_DPP_STEP
#else
TFX_DEV float group8_sum(float v) { v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); return v; }
TFX_DEV float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
TFX_DEV float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
#endif

// XCD-aware, bijective block remap (cdna_hip_programming.md T1): each of the 8 XCDs (private L2)
// gets a contiguous chunk of tile ids so neighbouring tiles share operand panels in L2.
TFX_DEV int xcd_remap(int bid, int nblk) {
  const int nx = 8;
  int q = nblk / nx, r = nblk % nx;
  int xcd = bid % nx, idx = bid / nx;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// LDS transpose read (gfx950 ds_read_b64_tr_b16).  Within each 16-lane group, lane q' supplies the
// address of 4 contiguous bf16 (row q'/4, 4-element chunk q'%4 of a [4][16] block); lane q receives
// column q of that block (rows 0..3).  Verified by tools/probe_layouts.hip.
TFX_DEV s16x4 lds_tr4(const bf16* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
}

// 8-element MFMA operand (for lane l: fixed column c0 + (l&31), 8 contraction rows) gathered from a
// row-major [row][col] LDS tile with `stride` elements per row.  rowA / rowB give this lane-half's first
// row of the first / second group of 4 contraction rows.
TFX_DEV bf16x8 lds_tr8(const bf16* tile, int stride, int rowA, int rowB, int c0) {
  const int l = threadIdx.x & 63;
  const int q = l & 15;
  const int col = c0 + 16 * ((l >> 4) & 1) + 4 * (q & 3);
  s16x4 lo = lds_tr4(tile + (rowA + (q >> 2)) * stride + col);
  s16x4 hi = lds_tr4(tile + (rowB + (q >> 2)) * stride + col);
  const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
  const u32x4 v = {l2[0], l2[1], h2[0], h2[1]};                                   // dword-granular register sequence
  return __builtin_bit_cast(bf16x8, v);
}

// QK-RMSNorm + RoPE (+ the q scale) of one 8-column chunk - 4 rotary pairs - of a 64-wide head vector held by 8 adjacent lanes (T:950-952, T:965,
// T:998): ONE spelling for the token-wise kernel and for the fused epilogue of the projection GEMM, every multiply-add an explicit fma and compiler
// contraction off - left to itself hipcc contracts `a cs - b sn` differently in the two kernels, one bf16 ulp apart where the products cancel.
// rs = norm_scale x (q ? q_scale : 1); gm = the chunk's 8 gains; cs / sn = the row's cos / sin for the chunk's 4 pairs.
TFX_DEV bf16x8 qk_norm_rope_chunk(const bf16x8 x, float ns, float qs, const float* gm, const f32x4 cs, const f32x4 sn) {
#pragma clang fp contract(off)
  float v[8], q = 0.f;
#pragma unroll
  for (int e = 0; e < 8; e++) { v[e] = bf2f(x[e]); q = __builtin_fmaf(v[e], v[e], q); }
  q = group8_sum(q);
  const float r = ns / fmaxf(sqrtf(q), 1e-12f) * qs;
  bf16x8 o;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const float a = v[2 * i] * r * (1.f + gm[2 * i]), b = v[2 * i + 1] * r * (1.f + gm[2 * i + 1]);
    o[2 * i] = f2bf(__builtin_fmaf(a, cs[i], -(b * sn[i])));
    o[2 * i + 1] = f2bf(__builtin_fmaf(b, cs[i], a * sn[i]));
  }
  return o;
}

// The layer's soft-cap plan (tfx.h tfx_qk_norm_rope_args.sc_plan), written by ONE wave (all 64 lanes call; lane e holds gain e).
TFX_DEV void softcap_plan_write(const float* gamma_q, const float* gamma_k, float norm_scale, float q_scale, float cap, float* o) {
  const int lane = threadIdx.x & 63;
  const float mq = wave_max(fabsf(1.f + gamma_q[lane])), mk = wave_max(fabsf(1.f + gamma_k[lane]));
  if (lane == 0) {
    const float ns = norm_scale > 0.f ? norm_scale : 8.f;
    const float B = 1.02f * ns * ns * q_scale * mq * mk, L2E = 1.4426950408889634f;
    const float bx = B / cap, b2 = bx * bx, ic2 = 1.f / (cap * cap);
    float mode = 2.f, a1 = 1.f, a3 = -1.f / 3.f, a5 = 0.f;
    if (bx <= 0.2f) { mode = 0.f; a1 = 1.f - b2 * b2 / 24.f; a3 = -1.f / 3.f + b2 / 6.f; }      // x^5 economised: (2/15)((5/4) b^2 x^3 - (5/16) b^4 x)
    else if (bx <= 0.35f) {                                                                        // x^7 economised: -(17/315)((7/4) b^2 x^5 - (7/8) b^4 x^3 + (7/64) b^6 x)
      const float c7 = 17.f / 315.f;
      mode = 1.f; a1 = 1.f - c7 * (7.f / 64.f) * b2 * b2 * b2; a3 = -1.f / 3.f + c7 * (7.f / 8.f) * b2 * b2; a5 = 2.f / 15.f - c7 * (7.f / 4.f) * b2;
    }
    o[0] = mode; o[1] = L2E * a1; o[2] = L2E * a3 * ic2; o[3] = L2E * a5 * ic2 * ic2;
    o[4] = a1; o[5] = 3.f * a3 * ic2; o[6] = 5.f * a5 * ic2 * ic2; o[7] = B;
  }
}

// LDS-DMA (global_load_lds_dwordx4) issued from inline asm: lane i of the wave writes 16 bytes at lds_wave_base + 16*i
// (lane-linear 1 KiB piece), M0 = wave-uniform LDS byte address.  hipcc models the builtin form as a pending LDS write
// and drains it with vmcnt(0); the asm form is invisible to that bookkeeping, so kernels count their own DMAs with
// `s_waitcnt vmcnt(N)` (cdna_hip_programming.md 5.7).
typedef __attribute__((address_space(3))) void lds_void_t;
TFX_DEV void glds16_asm(const bf16* g, const bf16* lds_wave_base) {
  const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void_t*)lds_wave_base);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_addr) : "memory");
}

// Phi(g) = 0.5 (1 + erf(g / sqrt 2)) from the exponential the GELU derivative needs anyway: erf(x) = 1 - poly(t) exp(-x^2),
// t = 1 / (1 + 0.3275911 x), x >= 0 (Abramowitz-Stegun 7.1.26, |error| <= 1.5e-7) - with x = |g| / sqrt 2 the exponential is
// E = exp(-g^2 / 2), the Gaussian of phi(g) = E / sqrt(2 pi).  ~12 VALU slots against ~30 for erff + the separate exp.
TFX_DEV float gelu_cdf(float g, float& E) {
  E = __expf(-0.5f * g * g);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, fabsf(g), 1.f));
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  const float half_erf = 0.5f - 0.5f * poly * E;                     // 0.5 erf(|g| / sqrt 2)
  return 0.5f + __builtin_copysignf(half_erf, g);
}
TFX_DEV float gelu_erf(float x) { float E; return x * gelu_cdf(x, E); }
// What the GEGLU forward SAVES for its backward (round 5): u = gelu(g) and v = a gelu'(g), in the value / gate slots of the interleaved buffer, instead of the
// pre-activations a and g.  d(a gelu(g)) = dh u da' + dh v dg', so the backward epilogue is two multiplies per element pair - no exponential, no table,
// no conversion of a second stream - and the forward pays 3 more FMAs on values it holds in fp32 anyway (u, v come from the fp32 accumulator: one
// rounding to bf16 each, where the round-4 form rounded a and g first and evaluated gelu / gelu' on the rounded g).  h = a u is the product itself.
struct GegluUVH { float u, v, h; };
TFX_DEV GegluUVH geglu_uvh(float a, float g) {
  float E; const float cdf = gelu_cdf(g, E);
  GegluUVH r; r.u = g * cdf; r.h = a * r.u;
  r.v = a * fmaf(g * 0.39894228040143267794f, E, cdf);
  return r;
}
TFX_DEV float gelu_erf_grad(float x) {
  return 0.5f * (1.f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
}
TFX_DEV float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }   // v_rcp_f32: 1 ulp, no IEEE divide sequence

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE setting: one flag bit per device (a process may drive several), set under a lock so that
// a second host thread cannot launch between another thread's flag write and its attribute call.  `done_mask`: a static of the launch site.
inline std::mutex& smem_attr_mutex() { static std::mutex mu; return mu; }
inline void ensure_smem_attr(const void* fn, int bytes, uint32_t& done_mask) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  const uint32_t bit = 1u << (dev & 31);
  std::lock_guard<std::mutex> lock(smem_attr_mutex());
  if (!(done_mask & bit)) { (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); done_mask |= bit; }
}
