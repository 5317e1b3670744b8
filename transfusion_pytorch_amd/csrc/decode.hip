// Decode-side kernels of `sample_many` (reference T:2082-2583): text-token sampling on the device and the ODE state update.
// Latency-bound (a handful of rows per step): one wave per row, no host round trip for the arithmetic.
#include "tfx_kernels.h"

namespace tfx {

// sample_text_token (T:597-605) + min_p_filter (T:591-595) for one row of fp32 logits per wave.
//   temperature == 0 : argmax over the V logits (first index on ties, like torch.argmax)
//   else             : p = softmax(logits / temperature); tokens with p < min_p * max(p) are removed; the survivor with
//                      cumulative mass crossing u * (total surviving mass) is drawn (inverse CDF in index order - the same
//                      distribution torch.multinomial samples from)
__global__ __launch_bounds__(256) void sample_tokens_k(const float* logits, int ld, int B, int V, float temperature, float min_p, const float* uniforms,
                                                       const int* active, int* out_ids, int V_draw) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= B) return;
  if (active && !active[row]) return;
  const float* lg = logits + (size_t)row * ld;
  // pass 1: maximum and its first index
  float best = -__builtin_inff(); int bi = 0x7fffffff;
  for (int c = lane; c < V; c += 64) { const float v = lg[c]; if (v > best) { best = v; bi = c; } }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (temperature == 0.f) { if (lane == 0) out_ids[row] = bi; return; }
  // pass 2: surviving mass.  p_c / p_max = exp((l_c - l_max) / T): the min-p test needs no normaliser
  const float it = 1.f / temperature;
  float mass = 0.f;
  for (int c = lane; c < V_draw; c += 64) { const float r = __expf((lg[c] - best) * it); mass += r >= min_p ? r : 0.f; }
  mass = wave_sum(mass);
  // pass 3: inverse CDF in index order, 64 columns per step with a wave prefix sum
  const float target = uniforms[row] * mass;
  float run = 0.f; int pick = bi;                       // falls back to the mode if rounding leaves the target unreached
  bool done = false;
  for (int c0 = 0; c0 < V_draw && !done; c0 += 64) {
    const int c = c0 + lane;
    float r = 0.f;
    if (c < V_draw) { r = __expf((lg[c] - best) * it); r = r >= min_p ? r : 0.f; }
    float pre = r;                                      // inclusive prefix sum across the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const float t = __shfl_up(pre, o, 64); if (lane >= o) pre += t; }
    const bool hit = r > 0.f && run + pre > target;
    const unsigned long long m = __ballot(hit);
    if (m) { pick = c0 + __builtin_ctzll(m); done = true; }
    run += __shfl(pre, 63, 64);
  }
  if (lane == 0) out_ids[row] = pick;
}

// out = y + a * f with f = f_cond, or f_uncond + cfg * (f_cond - f_uncond) when f_uncond is given (classifier-free guidance, T:2516-2521)
__global__ void ode_axpy_k(const float* y, const float* fc, const float* fu, float cfg, float a, float* out, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float f = fc[i];
  if (fu) { const float u = fu[i]; f = u + cfg * (f - u); }
  out[i] = y[i] + a * f;
}

// per-sample solver state machine (tfx.h: tfx_ode_stage / tfx_ode_update): one thread per (sample, row, column < dl)
__global__ void ode_stage_k(const float* y, const float* ym, const float* ctl, int B, int Lc, int dmax, float* x, int H, int Lq, int dl, const int32_t* rows0) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long long)B * Lc * dl) return;
  const int c = (int)(e % dl), j = (int)((e / dl) % Lc), i = (int)(e / ((long long)dl * Lc));
  const int mode = (int)ctl[i];
  if (mode == 0) return;
  const size_t src = ((size_t)i * Lc + j) * dmax + c;
  const float v = mode == 2 ? ym[src] : y[src];
  for (int h = 0; h < H; h++) {
    const int r0 = rows0 ? rows0[h * B + i] : (h * B + i) * Lq;
    if (r0 >= 0) x[((size_t)r0 + j) * dl + c] = v;
  }
}

__global__ void ode_update_k(float* y, float* ym, const float* ctl, int B, int Lc, int dmax, const float* pred, int H, int Lq, int dl, float cfg,
                             const float* sel, const int32_t* rows0) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long long)B * Lc * dl) return;
  const int c = (int)(e % dl), j = (int)((e / dl) % Lc), i = (int)(e / ((long long)dl * Lc));
  const int mode = (int)ctl[i];
  if ((mode != 1 && mode != 2) || (sel && sel[i] == 0.f)) return;
  const int rc = rows0 ? rows0[i] : i * Lq;
  if (rc < 0) return;
  float f = pred[((size_t)rc + j) * dl + c];
  if (H == 2) {
    const int ru = rows0 ? rows0[B + i] : (B + i) * Lq;
    if (ru < 0) return;
    const float u = pred[((size_t)ru + j) * dl + c]; f = u + cfg * (f - u);
  }
  const size_t at = ((size_t)i * Lc + j) * dmax + c;
  const float r = y[at] + ctl[B + i] * f;
  if (mode == 1) ym[at] = r; else y[at] = r;
}

// R x C block of a fp32 matrix -> bf16 block of another matrix (8 elements per thread)
__global__ void cast_block_k(const float* src, int ld_src, bf16* dst, int ld_dst, int R, int C8) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)R * C8) return;
  const int r = (int)(i / C8), c = (int)(i % C8) * 8;
  const f32x4 a = *(const f32x4*)(src + (size_t)r * ld_src + c), b = *(const f32x4*)(src + (size_t)r * ld_src + c + 4);
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 4; e++) { o[e] = f2bf(a[e]); o[4 + e] = f2bf(b[e]); }
  *(bf16x8*)(dst + (size_t)r * ld_dst + c) = o;
}

__global__ void scale_copy_k(const bf16* src, bf16* dst, long long n, float sc) {
  const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (i + 7 < n) {
    bf16x8 v = *(const bf16x8*)(src + i);
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = f2bf(bf2f(v[e]) * sc);
    *(bf16x8*)(dst + i) = v;
  } else for (long long j = i; j < n; j++) dst[j] = f2bf(bf2f(src[j]) * sc);
}


// Decode attention (reference score pipeline T:998-1027 against the KV cache T:1005-1016): one BLOCK per (sample, head) for up to R = 8 NEW query
// rows of the sample (instantiated for R = 1, 2: text steps), its four waves walk the visible keys 32 at
// a time.  Lane = (key group g = lane / 8, dim slice sl = lane % 8): per trip a wave takes 8 keys, every lane 8 of the 64 dims of its key (16-byte
// loads: a key row is 8 lanes x 16 B, contiguous) - loaded ONCE for all rows - and per row the partial dots meet with three DPP steps, then soft-cap
// (exact tanh - a handful of scores per lane), exp2 against the FIXED reference 0 (|cap * log2e * tanh| <= 72: no running maximum needed, as in
// the training kernel), and the value row joins the row's per-lane accumulator.  The 32 key groups are summed through LDS; out = o / l *
// sigmoid(gate).  A row's arithmetic does not depend on R or on the other rows of its block (keys past its own visible length add exact zeros in
// the same order), so a token comes out the same whichever plan carried it.
template <int R>
__global__ __launch_bounds__(256) void decode_attn_rows_k(tfx_attn_args p) {
  extern __shared__ float red_raw[];                       // [R][wave * 8 + key group][dim slice][8 dims + l]
  float (*red)[32][8][9] = (float (*)[32][8][9])red_raw;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int g = lane >> 3, sl = lane & 7;
  const int b = blockIdx.x / p.h, h = blockIdx.x % p.h;
  const int t0 = b * p.n;
  int kve[R], kmax = 1;
  float q8[R][8], o8[R][8], l[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    kve[r] = r < p.n ? min(p.kv_end[t0 + r], p.n_kv) : 0;
    kmax = max(kmax, kve[r]);
    bf16x8 qv;
    if (r < p.n) qv = *(const bf16x8*)(p.q + (size_t)(t0 + r) * p.ld_q + h * 64 + sl * 8);
#pragma unroll
    for (int e = 0; e < 8; e++) { q8[r][e] = r < p.n ? bf2f(qv[e]) : 0.f; o8[r][e] = 0.f; }
    l[r] = 0.f;
  }
  const bf16* kb = p.k + (size_t)b * p.n_kv * p.ld_k + h * 64 + sl * 8;
  const bf16* vb = p.v + (size_t)b * p.n_kv * p.ld_v + h * 64 + sl * 8;
  const float icap = 1.f / p.softcap, cap2 = p.softcap * 1.4426950408889634f;
#pragma unroll 2
  for (int j0 = wv * 8; j0 < kmax; j0 += 32) {
    const int j = j0 + g;
    const int jc = min(j, kmax - 1);
    const bf16x8 kv = *(const bf16x8*)(kb + (size_t)jc * p.ld_k);
    const bf16x8 vv = *(const bf16x8*)(vb + (size_t)jc * p.ld_v);
    float kf[8], vf[8];
#pragma unroll
    for (int e = 0; e < 8; e++) { kf[e] = bf2f(kv[e]); vf[e] = bf2f(vv[e]); }
#pragma unroll
    for (int r = 0; r < R; r++) {
      float dot = 0.f;
#pragma unroll
      for (int e = 0; e < 8; e++) dot = __builtin_fmaf(q8[r][e], kf[e], dot);
      dot = group8_sum(dot);
      const float x = dot * icap;
      const float th = 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * (2.f * 1.4426950408889634f)));   // tanh(x)
      const float pj = j < kve[r] ? __builtin_amdgcn_exp2f(cap2 * th) : 0.f;
      l[r] += pj;
#pragma unroll
      for (int e = 0; e < 8; e++) o8[r][e] = __builtin_fmaf(pj, vf[e], o8[r][e]);
    }
  }
#pragma unroll
  for (int r = 0; r < R; r++) {
#pragma unroll
    for (int e = 0; e < 8; e++) red[r][wv * 8 + g][sl][e] = o8[r][e];
    red[r][wv * 8 + g][sl][8] = l[r];
  }
  __syncthreads();
  if (threadIdx.x < 8 * R && (int)(threadIdx.x >> 3) < p.n) {
    const int r = threadIdx.x >> 3, s2 = threadIdx.x & 7, t = t0 + r;
    float o[8], lt = 0.f;
#pragma unroll
    for (int e = 0; e < 8; e++) o[e] = 0.f;
    for (int gg = 0; gg < 32; gg++) {
#pragma unroll
      for (int e = 0; e < 8; e++) o[e] += red[r][gg][s2][e];
      lt += red[r][gg][s2][8];
    }
    const float gate = bf2f(p.gate[(size_t)t * p.ld_gate + h]);
    const float sc = __builtin_amdgcn_rcpf(1.f + __expf(-gate)) / lt;
    bf16x8 ov;
#pragma unroll
    for (int e = 0; e < 8; e++) ov[e] = f2bf(o[e] * sc);
    *(bf16x8*)(p.out + (size_t)t * p.ld_out + h * 64 + s2 * 8) = ov;
    if (p.lse && s2 == 0) p.lse[((size_t)b * p.h + h) * p.n + r] = __logf(lt);
  }
}

template <int R> static int launch_decode_rows(const tfx_attn_args& a, hipStream_t s) {
  static uint32_t attr = 0;
  const int smem = R * 32 * 8 * 9 * 4;
  ensure_smem_attr((const void*)decode_attn_rows_k<R>, smem, attr);
  hipLaunchKernelGGL(decode_attn_rows_k<R>, dim3((unsigned)(a.b * a.h)), dim3(256), smem, s, a);
  return (int)hipGetLastError();
}
}  // namespace tfx
using namespace tfx;

extern "C" {
int tfx_sample_tokens(const float* logits, int32_t ld, int32_t B, int32_t V, float temperature, float min_p, const float* uniforms,
                      const int32_t* active, int32_t* out_ids, void* s) {
  if (B <= 0) return 0;
  if (!logits || !out_ids || V <= 0 || ld < V) return -1;
  if (temperature != 0.f && !uniforms) return -2;
  if (temperature < 0.f) return -3;
  hipLaunchKernelGGL(sample_tokens_k, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)s, logits, ld, B, V, temperature, min_p, uniforms, active, out_ids, V);
  return (int)hipGetLastError();
}
int tfx_sample_tokens_range(const float* logits, int32_t ld, int32_t B, int32_t V, int32_t V_draw, float temperature, float min_p, const float* uniforms,
                            const int32_t* active, int32_t* out_ids, void* s) {
  if (B <= 0) return 0;
  if (!logits || !out_ids || V <= 0 || V_draw <= 0 || V_draw > V || ld < V || (temperature != 0.f && !uniforms)) return -1;
  hipLaunchKernelGGL(sample_tokens_k, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)s, logits, ld, B, V, temperature, min_p, uniforms, active, out_ids, V_draw);
  return (int)hipGetLastError();
}
int tfx_ode_axpy(const float* y, const float* f_cond, const float* f_uncond, float cfg_scale, float a, float* out, int64_t n, void* s) {
  if (n <= 0) return 0;
  if (!y || !f_cond || !out) return -1;
  hipLaunchKernelGGL(ode_axpy_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)s, y, f_cond, f_uncond, cfg_scale, a, out, (long long)n);
  return (int)hipGetLastError();
}
int tfx_ode_stage(const float* y, const float* ym, const float* ctl, int32_t B, int32_t Lc, int32_t dmax, float* x, int32_t H, int32_t Lq, int32_t dl,
                  const int32_t* rows0, void* s) {
  if (B <= 0 || Lc <= 0 || dl <= 0) return 0;
  if (!y || !ym || !ctl || !x || H < 1 || H > 2 || Lq < Lc || dl > dmax) return -1;
  const long long n = (long long)B * Lc * dl;
  hipLaunchKernelGGL(ode_stage_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)s, y, ym, ctl, B, Lc, dmax, x, H, Lq, dl, rows0);
  return (int)hipGetLastError();
}
int tfx_ode_update(float* y, float* ym, const float* ctl, int32_t B, int32_t Lc, int32_t dmax, const float* pred, int32_t H, int32_t Lq, int32_t dl,
                   float cfg_scale, const float* sel, const int32_t* rows0, void* s) {
  if (B <= 0 || Lc <= 0 || dl <= 0) return 0;
  if (!y || !ym || !ctl || !pred || H < 1 || H > 2 || Lq < Lc || dl > dmax) return -1;
  const long long n = (long long)B * Lc * dl;
  hipLaunchKernelGGL(ode_update_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)s, y, ym, ctl, B, Lc, dmax, pred, H, Lq, dl, cfg_scale, sel, rows0);
  return (int)hipGetLastError();
}
int tfx_scale_bf16_copy(const tfx_bf16* src, tfx_bf16* dst, int64_t n, float scale, void* s) {
  if (n <= 0) return 0;
  if (!src || !dst) return -1;
  hipLaunchKernelGGL(scale_copy_k, dim3((unsigned)((n / 8 + 256) / 256)), dim3(256), 0, (hipStream_t)s, src, dst, (long long)n, scale);
  return (int)hipGetLastError();
}
int tfx_cast_block_bf16(const float* src, int32_t ld_src, tfx_bf16* dst, int32_t ld_dst, int32_t R, int32_t C, void* s) {
  if (R <= 0 || C <= 0) return 0;
  if (!src || !dst || (C | ld_src | ld_dst) % 8 || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return -1;
  const long long n = (long long)R * (C / 8);
  hipLaunchKernelGGL(cast_block_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)s, src, ld_src, dst, ld_dst, R, C / 8);
  return (int)hipGetLastError();
}
/* the K13 name of SURVEY 8(b): attention of a short block of NEW query rows per sample against keys / values that live in a KV cache
 * (`n_kv` > 0 rows per sample, per-row visible length in `kv_end`; the rows of a sample need not be ordered by visible length). */
int tfx_decode_attn(const tfx_attn_args* a, void* s) {
  if (!a || a->n_kv <= 0) return -10;
  if ((long long)a->b * a->h == 0 || a->n <= 0) return 0;
  // one or two rows per sample (text steps): the multi-row kernel; from there on the tiled forward kernel with its cache addressing is the faster
  // one - both read the layer's whole KV cache and are HBM-bound (86 MB per launch at 128 cache rows x 330 keys: 21.6 us for five rows per sample
  // here against 18.9 us there, rocprofv3 inside the continuous decode loop)
  if (a->n == 1) return tfx::launch_decode_rows<1>(*a, (hipStream_t)s);
  if (a->n == 2) return tfx::launch_decode_rows<2>(*a, (hipStream_t)s);
  return tfx::attn_fwd(*a, (hipStream_t)s);
}
}
