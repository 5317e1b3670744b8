// Token-wise (HBM-bound) kernels: one 64-lane wave per token row, 16-byte bf16x8 accesses, wavefront
// shuffle reductions.  A lane owns chunks c = lane + 64*i (8 contiguous elements each), i < NC.
#include "tfx_kernels.h"
#include <map>
#include <mutex>
#include <utility>
#include <cstdlib>

namespace tfx {

constexpr int WAVES = 4;          // waves per block
constexpr int MAXB = 1024;        // grid cap for kernels that end in parameter-gradient atomics

template <int NC> struct Row {
  float v[NC][8];
};

template <int NC> TFX_DEV void load_row(Row<NC>& r, const bf16* p, int d, int lane) {
#pragma unroll
  for (int i = 0; i < NC; i++) {
    int c = lane + 64 * i;
    if (c * 8 < d) {
      bf16x8 x = *(const bf16x8*)(p + c * 8);
#pragma unroll
      for (int e = 0; e < 8; e++) r.v[i][e] = bf2f(x[e]);
    } else {
#pragma unroll
      for (int e = 0; e < 8; e++) r.v[i][e] = 0.f;
    }
  }
}
template <int NC> TFX_DEV void store_row(const Row<NC>& r, bf16* p, int d, int lane) {
#pragma unroll
  for (int i = 0; i < NC; i++) {
    int c = lane + 64 * i;
    if (c * 8 < d) {
      bf16x8 x;
#pragma unroll
      for (int e = 0; e < 8; e++) x[e] = f2bf(r.v[i][e]);
      *(bf16x8*)(p + c * 8) = x;
    }
  }
}
template <int NC> TFX_DEV void load_vec(Row<NC>& r, const float* p, int d, int lane) {
#pragma unroll
  for (int i = 0; i < NC; i++) {
    int c = lane + 64 * i;
    if (c * 8 < d) {
      f32x4 a = *(const f32x4*)(p + c * 8), b = *(const f32x4*)(p + c * 8 + 4);
#pragma unroll
      for (int e = 0; e < 4; e++) { r.v[i][e] = a[e]; r.v[i][4 + e] = b[e]; }
    } else {
#pragma unroll
      for (int e = 0; e < 8; e++) r.v[i][e] = 0.f;
    }
  }
}
// raw (unconverted) row: 4 VGPRs per 8 elements - used where several rows are prefetched before their first use
template <int NC> struct RowRaw { bf16x8 v[NC]; };
template <int NC> TFX_DEV void load_raw(RowRaw<NC>& r, const bf16* p, int d, int lane) {
#pragma unroll
  for (int i = 0; i < NC; i++) {
    const int c = lane + 64 * i;
    if (c * 8 < d) r.v[i] = *(const bf16x8*)(p + c * 8);
    else {
#pragma unroll
      for (int e = 0; e < 8; e++) r.v[i][e] = f2bf(0.f);
    }
  }
}
// no instruction: what is computed from the row afterwards cannot be scheduled above this point (keeps prefetched rows in their 4-register
// raw form until their turn instead of widening all of them as soon as they land)
template <int NC> TFX_DEV void pin_raw(RowRaw<NC>& r) {
#pragma unroll
  for (int i = 0; i < NC; i++) asm volatile("" : "+v"(r.v[i]));
}
template <int NC> TFX_DEV void widen(Row<NC>& o, const RowRaw<NC>& r) {
#pragma unroll
  for (int i = 0; i < NC; i++)
#pragma unroll
    for (int e = 0; e < 8; e++) o.v[i][e] = bf2f(r.v[i][e]);
}
// block-level reduction of per-lane column partials over the block's waves, then global atomics
template <int NC> TFX_DEV void flush_col_partials(const Row<NC>& part, float* out, int d, float* smem /* [WAVES][NC*512] */) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NC; i++)
#pragma unroll
    for (int e = 0; e < 8; e++) smem[w * NC * 512 + (lane + 64 * i) * 8 + e] = part.v[i][e];
  __syncthreads();
  for (int c = threadIdx.x; c < d; c += blockDim.x) {
    float s = 0.f;
    const int nw = blockDim.x >> 6;
    for (int ww = 0; ww < nw; ww++) s += smem[ww * NC * 512 + c];
    if (s != 0.f) atomicAdd(out + c, s);
  }
  __syncthreads();
}

#define DISPATCH_NC(d, CALL)                     \
  do {                                           \
    if ((d) % 8 != 0 || (d) > 2048) return -1;   \
    if ((d) <= 512) { constexpr int NC = 1; CALL; } \
    else if ((d) <= 1024) { constexpr int NC = 2; CALL; } \
    else { constexpr int NC = 4; CALL; }         \
  } while (0)

#define DISPATCH_NC_LE1024(d, CALL)              \
  do {                                           \
    if ((d) % 8 != 0 || (d) > 1024) return -1;   \
    if ((d) <= 512) { constexpr int NC = 1; CALL; } \
    else { constexpr int NC = 2; CALL; }         \
  } while (0)

static inline int grid_tokens(int T) { return (T + WAVES - 1) / WAVES; }
static inline int grid_capped(int T) { int g = grid_tokens(T); return g < MAXB ? g : MAXB; }

// ------------------------------------------------------------------------------------------------
// AdaptiveWrapper input side: non-affine LayerNorm + (text gamma | FiLM gamma,beta)        T:747-755
// ------------------------------------------------------------------------------------------------
template <int NC> __global__ __launch_bounds__(256) void adaln_pre_fwd_k(tfx_adaln_pre_args p) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * WAVES + (threadIdx.x >> 6);
  if (t >= p.T) return;
  const int d = p.d;
  Row<NC> x; load_row(x, p.x + (size_t)t * d, d, lane);
  // the instance id and the modulation rows are requested BEFORE the statistics: three dependent round trips (row, id, table) become two
  // (a grid-stride form with next-row prefetch measured 4 % slower than this one-row-per-wave form)
  const int inst = p.tok_inst[t];
  Row<NC> g, b;
  if (inst < 0) load_vec(g, p.gamma_text, d, lane);
  else { load_vec(g, p.table + (size_t)inst * p.ld_table, d, lane); load_vec(b, p.table + (size_t)inst * p.ld_table + d, d, lane); }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NC; i++)
#pragma unroll
    for (int e = 0; e < 8; e++) s += x.v[i][e];
  const float mean = wave_sum(s) / d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NC; i++) {
    int c = lane + 64 * i;
    if (c * 8 < d)
#pragma unroll
      for (int e = 0; e < 8; e++) { float dv = x.v[i][e] - mean; q += dv * dv; }
  }
  const float rstd = rsqrtf(wave_sum(q) / d + 1e-5f);
  if (lane == 0) { p.mean[t] = mean; p.rstd[t] = rstd; }
#pragma unroll
  for (int i = 0; i < NC; i++)
#pragma unroll
    for (int e = 0; e < 8; e++) {
      float xh = (x.v[i][e] - mean) * rstd;
      x.v[i][e] = xh * (1.f + g.v[i][e]) + (inst < 0 ? 0.f : b.v[i][e]);
    }
  store_row(x, p.u + (size_t)t * d, d, lane);
}

template <int NC> __global__ __launch_bounds__(256) void adaln_pre_bwd_k(tfx_adaln_pre_args p) {
  __shared__ float smem[WAVES * NC * 512];
  const int lane = threadIdx.x & 63;
  const int d = p.d;
  Row<NC> pg;
#pragma unroll
  for (int i = 0; i < NC; i++)
#pragma unroll
    for (int e = 0; e < 8; e++) pg.v[i][e] = 0.f;
  for (int t = blockIdx.x * WAVES + (threadIdx.x >> 6); t < p.T; t += gridDim.x * WAVES) {
    Row<NC> x, du, g;
    load_row(x, p.x + (size_t)t * d, d, lane);
    load_row(du, p.du + (size_t)t * d, d, lane);
    const float mean = p.mean[t], rstd = p.rstd[t];
    const int inst = p.tok_inst[t];
    if (inst < 0) load_vec(g, p.gamma_text, d, lane);
    else load_vec(g, p.table + (size_t)inst * p.ld_table, d, lane);
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int i = 0; i < NC; i++) {
      int c = lane + 64 * i;
      if (c * 8 >= d) continue;
#pragma unroll
      for (int e = 0; e < 8; e++) {
        float xh = (x.v[i][e] - mean) * rstd;
        float dg = du.v[i][e] * xh;
        if (inst < 0) pg.v[i][e] += dg;
        else {
          atomicAdd(p.dtable + (size_t)inst * p.ld_table + c * 8 + e, dg);
          atomicAdd(p.dtable + (size_t)inst * p.ld_table + d + c * 8 + e, du.v[i][e]);
        }
        float dxh = du.v[i][e] * (1.f + g.v[i][e]);
        c1 += dxh; c2 += dxh * xh;
        x.v[i][e] = xh; du.v[i][e] = dxh;
      }
    }
    c1 = wave_sum(c1) / d; c2 = wave_sum(c2) / d;
    Row<NC> dx; load_row(dx, p.dx + (size_t)t * d, d, lane);
    if (p.dx_add) {
      Row<NC> da; load_row(da, p.dx_add + (size_t)t * d, d, lane);
#pragma unroll
      for (int i = 0; i < NC; i++)
#pragma unroll
        for (int e = 0; e < 8; e++) dx.v[i][e] += da.v[i][e];
    }
#pragma unroll
    for (int i = 0; i < NC; i++)
#pragma unroll
      for (int e = 0; e < 8; e++) dx.v[i][e] += rstd * (du.v[i][e] - c1 - x.v[i][e] * c2);
    store_row(dx, p.dx + (size_t)t * d, d, lane);
  }
  flush_col_partials<NC>(pg, p.dgamma_text, d, smem);
}

// ------------------------------------------------------------------------------------------------
// AdaptiveWrapper output side + residual: out = x + y * (text: layerscale+1 | modality: sigmoid(z))   T:763-769
// ------------------------------------------------------------------------------------------------
template <int NC> __global__ __launch_bounds__(256) void adaln_post_fwd_k(tfx_adaln_post_args p) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * WAVES + (threadIdx.x >> 6);
  if (t >= p.T) return;
  const int d = p.d;
  Row<NC> x, y, s;
  load_row(x, p.x + (size_t)t * d, d, lane);
  load_row(y, p.y + (size_t)t * d, d, lane);
  const int inst = p.tok_inst[t];
  if (inst < 0) load_vec(s, p.layerscale, d, lane);
  else load_vec(s, p.table + (size_t)inst * p.ld_table + 2 * d, d, lane);
#pragma unroll
  for (int i = 0; i < NC; i++)
#pragma unroll
    for (int e = 0; e < 8; e++) {
      float sc = inst < 0 ? 1.f + s.v[i][e] : sigmoidf_(s.v[i][e]);
      x.v[i][e] += y.v[i][e] * sc;
    }
  store_row(x, p.out + (size_t)t * d, d, lane);
}

// decode steps: output side of one wrapper + input side of the next, one token per wave (see include/tfx.h tfx_adaln_post_pre_fwd)
template <int NC> __global__ __launch_bounds__(256) void adaln_post_pre_fwd_k(tfx_adaln_post_args p, tfx_adaln_pre_args q) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * WAVES + (threadIdx.x >> 6);
  if (t >= p.T) return;
  const int d = p.d;
  Row<NC> x, y, s, g, b;
  load_row(x, p.x + (size_t)t * d, d, lane);
  load_row(y, p.y + (size_t)t * d, d, lane);
  const int inst = p.tok_inst[t];
  if (inst < 0) { load_vec(s, p.layerscale, d, lane); load_vec(g, q.gamma_text, d, lane); }
  else {
    load_vec(s, p.table + (size_t)inst * p.ld_table + 2 * d, d, lane);
    load_vec(g, q.table + (size_t)inst * q.ld_table, d, lane); load_vec(b, q.table + (size_t)inst * q.ld_table + d, d, lane);
  }
  float sm = 0.f;
#pragma unroll
  for (int i = 0; i < NC; i++)
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const float sc = inst < 0 ? 1.f + s.v[i][e] : sigmoidf_(s.v[i][e]);
      x.v[i][e] = bf2f(f2bf(x.v[i][e] + y.v[i][e] * sc));            // what tfx_adaln_pre_fwd would read back
      sm += x.v[i][e];
    }
  store_row(x, p.out + (size_t)t * d, d, lane);
  const float mean = wave_sum(sm) / d;
  float qq = 0.f;
#pragma unroll
  for (int i = 0; i < NC; i++) {
    int c = lane + 64 * i;
    if (c * 8 < d)
#pragma unroll
      for (int e = 0; e < 8; e++) { float dv = x.v[i][e] - mean; qq += dv * dv; }
  }
  const float rstd = rsqrtf(wave_sum(qq) / d + 1e-5f);
  if (lane == 0) { q.mean[t] = mean; q.rstd[t] = rstd; }
#pragma unroll
  for (int i = 0; i < NC; i++)
#pragma unroll
    for (int e = 0; e < 8; e++) {
      float xh = (x.v[i][e] - mean) * rstd;
      x.v[i][e] = xh * (1.f + g.v[i][e]) + (inst < 0 ? 0.f : b.v[i][e]);
    }
  store_row(x, q.u + (size_t)t * d, d, lane);
}

template <int NC> __global__ __launch_bounds__(256) void adaln_post_bwd_k(tfx_adaln_post_args p) {
  __shared__ float smem[WAVES * NC * 512];
  const int lane = threadIdx.x & 63;
  const int d = p.d;
  Row<NC> pl, pb;
#pragma unroll
  for (int i = 0; i < NC; i++)
#pragma unroll
    for (int e = 0; e < 8; e++) { pl.v[i][e] = 0.f; pb.v[i][e] = 0.f; }
  for (int t = blockIdx.x * WAVES + (threadIdx.x >> 6); t < p.T; t += gridDim.x * WAVES) {
    Row<NC> g, y, s;
    load_row(g, p.g + (size_t)t * d, d, lane);
    load_row(y, p.y + (size_t)t * d, d, lane);
    const int inst = p.tok_inst[t];
    if (inst < 0) load_vec(s, p.layerscale, d, lane);
    else load_vec(s, p.table + (size_t)inst * p.ld_table + 2 * d, d, lane);
#pragma unroll
    for (int i = 0; i < NC; i++) {
      int c = lane + 64 * i;
      if (c * 8 >= d) continue;
#pragma unroll
      for (int e = 0; e < 8; e++) {
        float gy = g.v[i][e] * y.v[i][e];
        float sc;
        if (inst < 0) { sc = 1.f + s.v[i][e]; pl.v[i][e] += gy; }
        else { sc = sigmoidf_(s.v[i][e]); atomicAdd(p.dtable + (size_t)inst * p.ld_table + 2 * d + c * 8 + e, gy * sc * (1.f - sc)); }
        g.v[i][e] *= sc; pb.v[i][e] += g.v[i][e];
      }
    }
    store_row(g, p.dy + (size_t)t * d, d, lane);
  }
  flush_col_partials<NC>(pl, p.dlayerscale, d, smem);
  if (p.dbias) flush_col_partials<NC>(pb, p.dbias, d, smem);
}

// ------------------------------------------------------------------------------------------------
// segment-mode backward of the two AdaLN kernels: a wave owns a run of consecutive tokens with one tok_inst
// value (a whole modality instance, or a short chunk of text), so the per-instance FiLM / ada-ln-zero
// gradients are reduced in registers and written with plain stores - no atomics on the table gradients.
// ------------------------------------------------------------------------------------------------
constexpr int SEG_WAVES = 8;      // 512-thread blocks: half as many closing atomics per parameter as 4-wave blocks
template <int NC> __global__ __launch_bounds__(512) void adaln_pre_bwd_seg_k(tfx_adaln_pre_args p) {
  __shared__ float smem[SEG_WAVES * NC * 512];
  const int lane = threadIdx.x & 63;
  const int d = p.d;
  Row<NC> pg;
#pragma unroll
  for (int i = 0; i < NC; i++)
#pragma unroll
    for (int e = 0; e < 8; e++) pg.v[i][e] = 0.f;
  for (int s = blockIdx.x * SEG_WAVES + (threadIdx.x >> 6); s < p.n_seg; s += gridDim.x * SEG_WAVES) {
    const int t0 = p.seg_start[s], len = p.seg_len[s];
    const int inst = p.tok_inst[t0];
    Row<NC> g, ag, ab;
    if (inst < 0) load_vec(g, p.gamma_text, d, lane);
    else load_vec(g, p.table + (size_t)inst * p.ld_table, d, lane);
#pragma unroll
    for (int i = 0; i < NC; i++)
#pragma unroll
      for (int e = 0; e < 8; e++) { ag.v[i][e] = 0.f; ab.v[i][e] = 0.f; }
    // U tokens per trip: all 3U row loads are issued before the first dependent use (a wave walks its segment
    // serially, so without this every token pays two full memory round trips)
    constexpr int U = NC == 1 ? 4 : (NC == 2 ? 2 : 1);
    const int tend = t0 + len;
    for (int tb = t0; tb < tend; tb += U) {
      RowRaw<NC> xs[U], dus[U], dxs[U], das[U];             // raw bf16: 12 prefetched rows cost 48 VGPRs, not 96
      float means[U], rstds[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int t = min(tb + u, tend - 1);
        load_raw(xs[u], p.x + (size_t)t * d, d, lane);
        load_raw(dus[u], p.du + (size_t)t * d, d, lane);
        load_raw(dxs[u], p.dx + (size_t)t * d, d, lane);
        if (p.dx_add) load_raw(das[u], p.dx_add + (size_t)t * d, d, lane);
        means[u] = p.mean[t]; rstds[u] = p.rstd[t];
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int t = tb + u;
        if (t >= tend) break;
        Row<NC> x, du, dx;
        widen(x, xs[u]); widen(du, dus[u]); widen(dx, dxs[u]);
        if (p.dx_add) {
          Row<NC> da; widen(da, das[u]);
#pragma unroll
          for (int i = 0; i < NC; i++)
#pragma unroll
            for (int e = 0; e < 8; e++) dx.v[i][e] += da.v[i][e];
        }
        const float mean = means[u], rstd = rstds[u];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int i = 0; i < NC; i++) {
          int c = lane + 64 * i;
          if (c * 8 >= d) continue;
#pragma unroll
          for (int e = 0; e < 8; e++) {
            float xh = (x.v[i][e] - mean) * rstd;
            ag.v[i][e] += du.v[i][e] * xh; ab.v[i][e] += du.v[i][e];
            float dxh = du.v[i][e] * (1.f + g.v[i][e]);
            c1 += dxh; c2 += dxh * xh;
            x.v[i][e] = xh; du.v[i][e] = dxh;
          }
        }
        c1 = wave_sum(c1) / d; c2 = wave_sum(c2) / d;
#pragma unroll
        for (int i = 0; i < NC; i++)
#pragma unroll
          for (int e = 0; e < 8; e++) dx.v[i][e] += rstd * (du.v[i][e] - c1 - x.v[i][e] * c2);
        store_row(dx, p.dx + (size_t)t * d, d, lane);
      }
    }
    if (inst < 0) {
#pragma unroll
      for (int i = 0; i < NC; i++)
#pragma unroll
        for (int e = 0; e < 8; e++) pg.v[i][e] += ag.v[i][e];
    } else {
      float* dt = p.dtable + (size_t)inst * p.ld_table;
#pragma unroll
      for (int i = 0; i < NC; i++) {
        int c = lane + 64 * i;
        if (c * 8 >= d) continue;
        f32x4 a0, a1, b0, b1;
#pragma unroll
        for (int e = 0; e < 4; e++) { a0[e] = ag.v[i][e]; a1[e] = ag.v[i][4 + e]; b0[e] = ab.v[i][e]; b1[e] = ab.v[i][4 + e]; }
        *(f32x4*)(dt + c * 8) = a0; *(f32x4*)(dt + c * 8 + 4) = a1;
        *(f32x4*)(dt + d + c * 8) = b0; *(f32x4*)(dt + d + c * 8 + 4) = b1;
      }
    }
  }
  flush_col_partials<NC>(pg, p.dgamma_text, d, smem);
}

template <int NC> __global__ __launch_bounds__(512) void adaln_post_bwd_seg_k(tfx_adaln_post_args p) {
  __shared__ float smem[SEG_WAVES * NC * 512];
  const int lane = threadIdx.x & 63;
  const int d = p.d;
  Row<NC> pl, pb;                                           // per-lane partials: d layerscale, d bias (column sums of dy)
#pragma unroll
  for (int i = 0; i < NC; i++)
#pragma unroll
    for (int e = 0; e < 8; e++) { pl.v[i][e] = 0.f; pb.v[i][e] = 0.f; }
  for (int s = blockIdx.x * SEG_WAVES + (threadIdx.x >> 6); s < p.n_seg; s += gridDim.x * SEG_WAVES) {
    const int t0 = p.seg_start[s], len = p.seg_len[s];
    const int inst = p.tok_inst[t0];
    Row<NC> sc, az;
    if (inst < 0) load_vec(sc, p.layerscale, d, lane);
    else load_vec(sc, p.table + (size_t)inst * p.ld_table + 2 * d, d, lane);
#pragma unroll
    for (int i = 0; i < NC; i++)
#pragma unroll
      for (int e = 0; e < 8; e++) { sc.v[i][e] = inst < 0 ? 1.f + sc.v[i][e] : sigmoidf_(sc.v[i][e]); az.v[i][e] = 0.f; }
    constexpr int U = NC == 1 ? 4 : (NC == 2 ? 2 : 1);      // U tokens per trip, loads first (see adaln_pre_bwd_seg_k)
    const int tend = t0 + len;
    for (int tb = t0; tb < tend; tb += U) {
      RowRaw<NC> gs[U], ys[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int t = min(tb + u, tend - 1);
        load_raw(gs[u], p.g + (size_t)t * d, d, lane);
        load_raw(ys[u], p.y + (size_t)t * d, d, lane);
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int t = tb + u;
        if (t >= tend) break;
        Row<NC> gg, yy;
        widen(gg, gs[u]); widen(yy, ys[u]);
#pragma unroll
        for (int i = 0; i < NC; i++)
#pragma unroll
          for (int e = 0; e < 8; e++) { az.v[i][e] += gg.v[i][e] * yy.v[i][e]; gg.v[i][e] *= sc.v[i][e]; pb.v[i][e] += gg.v[i][e]; }
        store_row(gg, p.dy + (size_t)t * d, d, lane);
      }
    }
    if (inst < 0) {
#pragma unroll
      for (int i = 0; i < NC; i++)
#pragma unroll
        for (int e = 0; e < 8; e++) pl.v[i][e] += az.v[i][e];
    } else {
      float* dt = p.dtable + (size_t)inst * p.ld_table + 2 * d;
#pragma unroll
      for (int i = 0; i < NC; i++) {
        int c = lane + 64 * i;
        if (c * 8 >= d) continue;
        f32x4 a0, a1;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          a0[e] = az.v[i][e] * sc.v[i][e] * (1.f - sc.v[i][e]);
          a1[e] = az.v[i][4 + e] * sc.v[i][4 + e] * (1.f - sc.v[i][4 + e]);
        }
        *(f32x4*)(dt + c * 8) = a0; *(f32x4*)(dt + c * 8 + 4) = a1;
      }
    }
  }
  flush_col_partials<NC>(pl, p.dlayerscale, d, smem);
  if (p.dbias) flush_col_partials<NC>(pb, p.dbias, d, smem);
}

// ------------------------------------------------------------------------------------------------
// RMSNorm (final norm): y = x / max(|x|, 1e-12) * sqrt(d) * (gamma + 1)                T:779-786
// ------------------------------------------------------------------------------------------------
template <int NC> __global__ __launch_bounds__(256) void rmsnorm_fwd_k(tfx_rmsnorm_args p) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * WAVES + (threadIdx.x >> 6);
  if (t >= p.T) return;
  const int d = p.d;
  Row<NC> x, g;
  load_row(x, p.x + (size_t)t * d, d, lane);
  load_vec(g, p.gamma, d, lane);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NC; i++)
#pragma unroll
    for (int e = 0; e < 8; e++) q += x.v[i][e] * x.v[i][e];
  const float r = sqrtf((float)d) / fmaxf(sqrtf(wave_sum(q)), 1e-12f);
#pragma unroll
  for (int i = 0; i < NC; i++)
#pragma unroll
    for (int e = 0; e < 8; e++) x.v[i][e] *= r * (1.f + g.v[i][e]);
  store_row(x, p.y + (size_t)t * d, d, lane);
}

template <int NC> __global__ __launch_bounds__(256) void rmsnorm_bwd_k(tfx_rmsnorm_args p) {
  __shared__ float smem[WAVES * NC * 512];
  const int lane = threadIdx.x & 63;
  const int d = p.d;
  Row<NC> pg, g;
  load_vec(g, p.gamma, d, lane);
#pragma unroll
  for (int i = 0; i < NC; i++)
#pragma unroll
    for (int e = 0; e < 8; e++) pg.v[i][e] = 0.f;
  for (int t = blockIdx.x * WAVES + (threadIdx.x >> 6); t < p.T; t += gridDim.x * WAVES) {
    Row<NC> x, dy;
    load_row(x, p.x + (size_t)t * d, d, lane);
    load_row(dy, p.dy + (size_t)t * d, d, lane);
    float q = 0.f, S = 0.f;
#pragma unroll
    for (int i = 0; i < NC; i++)
#pragma unroll
      for (int e = 0; e < 8; e++) { q += x.v[i][e] * x.v[i][e]; S += (1.f + g.v[i][e]) * dy.v[i][e] * x.v[i][e]; }
    q = wave_sum(q); S = wave_sum(S);
    const float r = sqrtf((float)d) / fmaxf(sqrtf(q), 1e-12f);
    const float k2 = r * r * r / d * S;
#pragma unroll
    for (int i = 0; i < NC; i++)
#pragma unroll
      for (int e = 0; e < 8; e++) {
        pg.v[i][e] += dy.v[i][e] * x.v[i][e] * r;
        x.v[i][e] = r * (1.f + g.v[i][e]) * dy.v[i][e] - x.v[i][e] * k2;
      }
    store_row(x, p.dx + (size_t)t * d, d, lane);
  }
  flush_col_partials<NC>(pg, p.dgamma, d, smem);
}

// ------------------------------------------------------------------------------------------------
// AttentionResidual: out = sum_l softmax_l(<h_l, w> / |h_l|) h_l,  w = (gamma+1) * pseudo_query   T:807-829
// ------------------------------------------------------------------------------------------------
// one step of the depth softmax (online form): fold hidden h into (o, m, den).  Shared by attnres_fwd_k and layer_end_fwd_k so that both
// evaluate the very same expressions (the decode fusion must reproduce the separate kernels bit for bit)
template <int NC> TFX_DEV void attnres_mix(const Row<NC>& h, const Row<NC>& w, Row<NC>& o, float& m, float& den, float& s_out, float& inv_out) {
  // every multiply-add is spelled as an explicit fma: the compiler's own contraction choices depend on the surrounding code
  float nsq = 0.f, dt = 0.f;
#pragma unroll
  for (int i = 0; i < NC; i++)
#pragma unroll
    for (int e = 0; e < 8; e++) { nsq = __builtin_fmaf(h.v[i][e], h.v[i][e], nsq); dt = __builtin_fmaf(h.v[i][e], w.v[i][e], dt); }
  nsq = wave_sum(nsq); dt = wave_sum(dt);
  const float nrm = fmaxf(sqrtf(nsq), 1e-12f);
  const float s = dt / nrm;
  s_out = s; inv_out = 1.f / nrm;
  const float mn = fmaxf(m, s);
  const float al = __expf(m - mn), ex = __expf(s - mn);
  den = __builtin_fmaf(den, al, ex); m = mn;
#pragma unroll
  for (int i = 0; i < NC; i++)
#pragma unroll
    for (int e = 0; e < 8; e++) { const float t = ex * h.v[i][e]; o.v[i][e] = __builtin_fmaf(o.v[i][e], al, t); }
}

template <int NC> TFX_DEV void attnres_mix(const Row<NC>& h, const Row<NC>& w, Row<NC>& o, float& m, float& den) {
  float s, inv; attnres_mix<NC>(h, w, o, m, den, s, inv);
}
// the depth softmax of one token, kept for the pull-form backward: lane l holds (s_l, 1 / |h_l|) of hidden l
TFX_DEV void attnres_save(float* save, int t, int L, int lane, float s_l, float inv_l, float m, float den) {
  if (lane < L) {
    f32x4 v = {__expf(s_l - m) / den, inv_l, s_l, 0.f};
    *(f32x4*)(save + ((size_t)t * L + lane) * 4) = v;
  }
}

template <int NC> __global__ __launch_bounds__(256) void attnres_fwd_k(tfx_attnres_args p) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * WAVES + (threadIdx.x >> 6);
  if (t >= p.T) return;
  const int d = p.d;
  Row<NC> w, pq, o;
  load_vec(w, p.gamma, d, lane); load_vec(pq, p.pq, d, lane);
#pragma unroll
  for (int i = 0; i < NC; i++)
#pragma unroll
    for (int e = 0; e < 8; e++) { w.v[i][e] = (1.f + w.v[i][e]) * pq.v[i][e]; o.v[i][e] = 0.f; }
  float m = -INFINITY, den = 0.f;
  float s_l = 0.f, inv_l = 0.f;
  constexpr int PF = NC == 1 ? 4 : 2;          // rows requested before the first of them is reduced (see attnres_bwd_k)
  for (int l0 = 0; l0 < p.L; l0 += PF) {
    RowRaw<NC> hr[PF];
#pragma unroll
    for (int j = 0; j < PF; j++)
      if (l0 + j < p.L) load_raw(hr[j], p.hiddens + (size_t)(l0 + j) * p.stride_h + (size_t)t * d, d, lane);
#pragma unroll
    for (int j = 0; j < PF; j++) {
      if (l0 + j < p.L) {
        Row<NC> h; widen(h, hr[j]);
        float sc, iv; attnres_mix<NC>(h, w, o, m, den, sc, iv);
        if (lane == l0 + j) { s_l = sc; inv_l = iv; }
      }
    }
  }
  if (p.save) attnres_save(p.save, t, p.L, lane, s_l, inv_l, m, den);
  const float inv = 1.f / den;
#pragma unroll
  for (int i = 0; i < NC; i++)
#pragma unroll
    for (int e = 0; e < 8; e++) o.v[i][e] *= inv;
  store_row(o, p.out + (size_t)t * d, d, lane);
  if (p.err) {
#pragma unroll
    for (int i = 0; i < NC; i++)
#pragma unroll
      for (int e = 0; e < 8; e++) o.v[i][e] -= bf2f(f2bf(o.v[i][e]));
    store_row(o, p.err + (size_t)t * d, d, lane);
  }
}

// decode steps: end of a layer in one launch (include/tfx.h tfx_layer_end_fwd): h = x + y * scale ; out = AttentionResidual(h_0 .. h_{L-2}, h) ;
// u = AdaLN-pre(out) for the next layer.  One token per wave; every stored row is rounded to bf16 before it is used again, as a re-read would.
template <int NC> __global__ __launch_bounds__(256) void layer_end_fwd_k(tfx_adaln_post_args p, tfx_attnres_args a, tfx_adaln_pre_args q, int has_pre) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * WAVES + (threadIdx.x >> 6);
  if (t >= p.T) return;
  const int d = p.d;
  const int inst = p.tok_inst[t];
  Row<NC> hn, w, o;
  {
    Row<NC> y, s;
    load_row(hn, p.x + (size_t)t * d, d, lane);
    load_row(y, p.y + (size_t)t * d, d, lane);
    if (inst < 0) load_vec(s, p.layerscale, d, lane);
    else load_vec(s, p.table + (size_t)inst * p.ld_table + 2 * d, d, lane);
#pragma unroll
    for (int i = 0; i < NC; i++)
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const float sc = inst < 0 ? 1.f + s.v[i][e] : sigmoidf_(s.v[i][e]);
        hn.v[i][e] = bf2f(f2bf(hn.v[i][e] + y.v[i][e] * sc));
      }
    store_row(hn, p.out + (size_t)t * d, d, lane);
  }
  {
    Row<NC> pq; load_vec(w, a.gamma, d, lane); load_vec(pq, a.pq, d, lane);
#pragma unroll
    for (int i = 0; i < NC; i++)
#pragma unroll
      for (int e = 0; e < 8; e++) { w.v[i][e] = (1.f + w.v[i][e]) * pq.v[i][e]; o.v[i][e] = 0.f; }
  }
  float m = -INFINITY, den = 0.f;
  float s_l = 0.f, inv_l = 0.f;
  // the L - 1 stored hiddens are requested four at a time (raw bf16) before the first of the group is reduced: with up to 25 hiddens and a
  // handful of tokens per CU the loop is a chain of memory round trips otherwise (measured 17.5 us per launch at depth 24)
  for (int l0 = 0; l0 + 1 < a.L; l0 += 4) {
    RowRaw<NC> hr[4];
#pragma unroll
    for (int j = 0; j < 4; j++)
      if (l0 + j + 1 < a.L) load_raw(hr[j], a.hiddens + (size_t)(l0 + j) * a.stride_h + (size_t)t * d, d, lane);
#pragma unroll
    for (int j = 0; j < 4; j++)
      if (l0 + j + 1 < a.L) {
        Row<NC> h; widen(h, hr[j]);
        float sc, iv; attnres_mix<NC>(h, w, o, m, den, sc, iv);
        if (lane == l0 + j) { s_l = sc; inv_l = iv; }
      }
  }
  {
    float sc, iv; attnres_mix<NC>(hn, w, o, m, den, sc, iv);
    if (lane == a.L - 1) { s_l = sc; inv_l = iv; }
  }
  if (a.save) attnres_save(a.save, t, a.L, lane, s_l, inv_l, m, den);
  const float inv = 1.f / den;
  float sm = 0.f;
  if (a.err) {                                               // what the rounding of the output drops (training plans; see tfx_attnres_args.err)
    Row<NC> er;
#pragma unroll
    for (int i = 0; i < NC; i++)
#pragma unroll
      for (int e = 0; e < 8; e++) { const float ex = o.v[i][e] * inv; er.v[i][e] = ex - bf2f(f2bf(ex)); }
    store_row(er, a.err + (size_t)t * d, d, lane);
  }
#pragma unroll
  for (int i = 0; i < NC; i++)
#pragma unroll
    for (int e = 0; e < 8; e++) { o.v[i][e] = bf2f(f2bf(o.v[i][e] * inv)); sm += o.v[i][e]; }
  store_row(o, a.out + (size_t)t * d, d, lane);
  if (!has_pre) return;
  Row<NC> g, b;
  if (inst < 0) load_vec(g, q.gamma_text, d, lane);
  else { load_vec(g, q.table + (size_t)inst * q.ld_table, d, lane); load_vec(b, q.table + (size_t)inst * q.ld_table + d, d, lane); }
  const float mean = wave_sum(sm) / d;
  float qq = 0.f;
#pragma unroll
  for (int i = 0; i < NC; i++) {
    int c = lane + 64 * i;
    if (c * 8 < d)
#pragma unroll
      for (int e = 0; e < 8; e++) { float dv = o.v[i][e] - mean; qq += dv * dv; }
  }
  const float rstd = rsqrtf(wave_sum(qq) / d + 1e-5f);
  if (lane == 0) { q.mean[t] = mean; q.rstd[t] = rstd; }
#pragma unroll
  for (int i = 0; i < NC; i++)
#pragma unroll
    for (int e = 0; e < 8; e++) {
      float xh = (o.v[i][e] - mean) * rstd;
      o.v[i][e] = xh * (1.f + g.v[i][e]) + (inst < 0 ? 0.f : b.v[i][e]);
    }
  store_row(o, q.u + (size_t)t * d, d, lane);
}

template <int NC> __global__ __launch_bounds__(256) void attnres_bwd_k(tfx_attnres_args p) {
  __shared__ float smem[WAVES * NC * 512];
  const int lane = threadIdx.x & 63;
  const int d = p.d;
  Row<NC> w, pw;     // w = (1+gamma)*pq ; pw = per-lane partial of d(w)
  {
    Row<NC> pq; load_vec(w, p.gamma, d, lane); load_vec(pq, p.pq, d, lane);
#pragma unroll
    for (int i = 0; i < NC; i++)
#pragma unroll
      for (int e = 0; e < 8; e++) { w.v[i][e] = (1.f + w.v[i][e]) * pq.v[i][e]; pw.v[i][e] = 0.f; }
  }
  for (int t = blockIdx.x * WAVES + (threadIdx.x >> 6); t < p.T; t += gridDim.x * WAVES) {
    Row<NC> g; load_row(g, p.g + (size_t)t * d, d, lane);
    if (p.g2) {
      Row<NC> g2; load_row(g2, p.g2 + (size_t)t * d, d, lane);
#pragma unroll
      for (int i = 0; i < NC; i++)
#pragma unroll
        for (int e = 0; e < 8; e++) g.v[i][e] += g2.v[i][e];
    }
    // pass 1: lane l keeps (score, inverse norm, <g,h_l>) of hidden l   (L <= 64).  Rows are requested PF at a time (raw bf16) before the
    // first of them is reduced: one wave then keeps PF rows in flight instead of one (the kernel is request-latency bound otherwise)
    float s_l = -INFINITY, inv_l = 0.f, da_l = 0.f;
    constexpr int PF = NC == 1 ? 4 : 2;
    for (int l0 = 0; l0 < p.L; l0 += PF) {
      RowRaw<NC> hr[PF];
#pragma unroll
      for (int j = 0; j < PF; j++)
        if (l0 + j < p.L) load_raw(hr[j], p.hiddens + (size_t)(l0 + j) * p.stride_h + (size_t)t * d, d, lane);
#pragma unroll
      for (int j = 0; j < PF; j++) {
        if (l0 + j < p.L) {
          Row<NC> h; widen(h, hr[j]);
          float nsq = 0.f, dt = 0.f, da = 0.f;
#pragma unroll
          for (int i = 0; i < NC; i++)
#pragma unroll
            for (int e = 0; e < 8; e++) { nsq += h.v[i][e] * h.v[i][e]; dt += h.v[i][e] * w.v[i][e]; da += h.v[i][e] * g.v[i][e]; }
          nsq = wave_sum(nsq); dt = wave_sum(dt); da = wave_sum(da);
          const float inv = 1.f / fmaxf(sqrtf(nsq), 1e-12f);
          if (lane == l0 + j) { s_l = dt * inv; inv_l = inv; da_l = da; }
        }
      }
    }
    const float mx = wave_max(s_l);
    const float ex = (lane < p.L) ? __expf(s_l - mx) : 0.f;
    const float a_l = ex / wave_sum(ex);
    const float dsum = wave_sum(a_l * da_l);
    const float ds_l = a_l * (da_l - dsum);
    // pass 2: the rows of hidden l + 1 (and its gradient, when accumulating) are requested before hidden l is updated
    RowRaw<NC> hq[2], dq[2];
    load_raw(hq[0], p.hiddens + (size_t)t * d, d, lane);
    if (!p.first) load_raw(dq[0], p.dhiddens + (size_t)t * d, d, lane);
    for (int l = 0; l < p.L; l += 2) {
#pragma unroll
      for (int b = 0; b < 2; b++) {
        const int ll = l + b;
        if (ll < p.L) {
          if (ll + 1 < p.L) {
            load_raw(hq[b ^ 1], p.hiddens + (size_t)(ll + 1) * p.stride_h + (size_t)t * d, d, lane);
            if (!p.first) load_raw(dq[b ^ 1], p.dhiddens + (size_t)(ll + 1) * p.stride_dh + (size_t)t * d, d, lane);
          }
          const float a = __shfl(a_l, ll, 64), ds = __shfl(ds_l, ll, 64), inv = __shfl(inv_l, ll, 64), sc = __shfl(s_l, ll, 64);
          Row<NC> h, dh; widen(h, hq[b]);
          if (!p.first) widen(dh, dq[b]);
          const float k1 = ds * inv, k2 = ds * sc * inv * inv;
#pragma unroll
          for (int i = 0; i < NC; i++)
#pragma unroll
            for (int e = 0; e < 8; e++) {
              float v = a * g.v[i][e] + k1 * w.v[i][e] - k2 * h.v[i][e];
              dh.v[i][e] = p.first ? v : dh.v[i][e] + v;
              pw.v[i][e] += k1 * h.v[i][e];
            }
          store_row(dh, p.dhiddens + (size_t)ll * p.stride_dh + (size_t)t * d, d, lane);
        }
      }
    }
  }
  // d gamma = dw * pq ; d pq = dw * (1 + gamma)
  Row<NC> gm, pq, tmp; load_vec(gm, p.gamma, d, lane); load_vec(pq, p.pq, d, lane);
#pragma unroll
  for (int i = 0; i < NC; i++)
#pragma unroll
    for (int e = 0; e < 8; e++) tmp.v[i][e] = pw.v[i][e] * pq.v[i][e];
  flush_col_partials<NC>(tmp, p.dgamma, d, smem);
#pragma unroll
  for (int i = 0; i < NC; i++)
#pragma unroll
    for (int e = 0; e < 8; e++) tmp.v[i][e] = pw.v[i][e] * (1.f + gm.v[i][e]);
  flush_col_partials<NC>(tmp, p.dpq, d, smem);
}

// ------------------------------------------------------------------------------------------------
// Backward of two adjacent wrapper sides in one launch (include/tfx.h tfx_adaln_pre_post_bwd): adaln_pre_bwd_seg_k followed by
// adaln_post_bwd_seg_k on the residual-gradient row it just produced (rounded to bf16 exactly as the second launch would read it back).
// ------------------------------------------------------------------------------------------------
template <int NC> __global__ __launch_bounds__(512) void adaln_pre_post_bwd_seg_k(tfx_adaln_pre_args p, tfx_adaln_post_args q) {
  __shared__ float smem[SEG_WAVES * NC * 512];
  const int lane = threadIdx.x & 63;
  const int d = p.d;
  Row<NC> pg, pl;                                           // text partials: d layernorm_gamma (pre), d layerscale (post)
#pragma unroll
  for (int i = 0; i < NC; i++)
#pragma unroll
    for (int e = 0; e < 8; e++) { pg.v[i][e] = 0.f; pl.v[i][e] = 0.f; }
  for (int s = blockIdx.x * SEG_WAVES + (threadIdx.x >> 6); s < p.n_seg; s += gridDim.x * SEG_WAVES) {
    const int t0 = p.seg_start[s], len = p.seg_len[s];
    const int inst = p.tok_inst[t0];
    Row<NC> g, ag, ab, sc, az;
    if (inst < 0) { load_vec(g, p.gamma_text, d, lane); load_vec(sc, q.layerscale, d, lane); }
    else { load_vec(g, p.table + (size_t)inst * p.ld_table, d, lane); load_vec(sc, q.table + (size_t)inst * q.ld_table + 2 * d, d, lane); }
#pragma unroll
    for (int i = 0; i < NC; i++)
#pragma unroll
      for (int e = 0; e < 8; e++) {
        ag.v[i][e] = 0.f; ab.v[i][e] = 0.f; az.v[i][e] = 0.f;
        sc.v[i][e] = inst < 0 ? 1.f + sc.v[i][e] : sigmoidf_(sc.v[i][e]);
      }
    constexpr int U = NC == 1 ? 4 : (NC == 2 ? 2 : 1);
    const int tend = t0 + len;
    for (int tb = t0; tb < tend; tb += U) {
      RowRaw<NC> xs[U], dus[U], dxs[U], ys[U];
      float means[U], rstds[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int t = min(tb + u, tend - 1);
        load_raw(xs[u], p.x + (size_t)t * d, d, lane);
        load_raw(dus[u], p.du + (size_t)t * d, d, lane);
        load_raw(dxs[u], p.dx + (size_t)t * d, d, lane);
        load_raw(ys[u], q.y + (size_t)t * d, d, lane);
        means[u] = p.mean[t]; rstds[u] = p.rstd[t];
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int t = tb + u;
        if (t >= tend) break;
        Row<NC> x, du, dx;
        widen(x, xs[u]); widen(du, dus[u]); widen(dx, dxs[u]);
        const float mean = means[u], rstd = rstds[u];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int i = 0; i < NC; i++) {
          int c = lane + 64 * i;
          if (c * 8 >= d) continue;
#pragma unroll
          for (int e = 0; e < 8; e++) {
            float xh = (x.v[i][e] - mean) * rstd;
            ag.v[i][e] += du.v[i][e] * xh; ab.v[i][e] += du.v[i][e];
            float dxh = du.v[i][e] * (1.f + g.v[i][e]);
            c1 += dxh; c2 += dxh * xh;
            x.v[i][e] = xh; du.v[i][e] = dxh;
          }
        }
        c1 = wave_sum(c1) / d; c2 = wave_sum(c2) / d;
#pragma unroll
        for (int i = 0; i < NC; i++)
#pragma unroll
          for (int e = 0; e < 8; e++) dx.v[i][e] = bf2f(f2bf(dx.v[i][e] + rstd * (du.v[i][e] - c1 - x.v[i][e] * c2)));
        store_row(dx, p.dx + (size_t)t * d, d, lane);
        // output side of the wrapper below: g = the row just written
        Row<NC> yy; widen(yy, ys[u]);
#pragma unroll
        for (int i = 0; i < NC; i++)
#pragma unroll
          for (int e = 0; e < 8; e++) { az.v[i][e] += dx.v[i][e] * yy.v[i][e]; dx.v[i][e] *= sc.v[i][e]; }
        store_row(dx, q.dy + (size_t)t * d, d, lane);
      }
    }
    if (inst < 0) {
#pragma unroll
      for (int i = 0; i < NC; i++)
#pragma unroll
        for (int e = 0; e < 8; e++) { pg.v[i][e] += ag.v[i][e]; pl.v[i][e] += az.v[i][e]; }
    } else {
      float* dt = p.dtable + (size_t)inst * p.ld_table;
      float* dz = q.dtable + (size_t)inst * q.ld_table + 2 * d;
#pragma unroll
      for (int i = 0; i < NC; i++) {
        int c = lane + 64 * i;
        if (c * 8 >= d) continue;
        f32x4 a0, a1, b0, b1, z0, z1;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          a0[e] = ag.v[i][e]; a1[e] = ag.v[i][4 + e]; b0[e] = ab.v[i][e]; b1[e] = ab.v[i][4 + e];
          z0[e] = az.v[i][e] * sc.v[i][e] * (1.f - sc.v[i][e]);
          z1[e] = az.v[i][4 + e] * sc.v[i][4 + e] * (1.f - sc.v[i][4 + e]);
        }
        *(f32x4*)(dt + c * 8) = a0; *(f32x4*)(dt + c * 8 + 4) = a1;
        *(f32x4*)(dt + d + c * 8) = b0; *(f32x4*)(dt + d + c * 8 + 4) = b1;
        *(f32x4*)(dz + c * 8) = z0; *(f32x4*)(dz + c * 8 + 4) = z1;
      }
    }
  }
  flush_col_partials<NC>(pg, p.dgamma_text, d, smem);
  flush_col_partials<NC>(pl, q.dlayerscale, d, smem);
}

// ------------------------------------------------------------------------------------------------
// AttentionResidual backward, pull form (include/tfx.h tfx_attnres_pull_bwd): the gradient of hidden l from every layer that mixed it,
// optionally followed by the output side of the feed-forward wrapper that produced the hidden.
//   NJ > 0: at most NJ sources, their d w partials live in registers and all NJ gradient rows of a token are requested at once;
//   NJ == 0: any number of sources (<= 32), processed JC rows at a time, d w partials accumulate in LDS (ds_add_f32).
// Dynamic LDS: [n_src][d] fp32 = the w rows (NJ > 0) or the d w accumulators (NJ == 0).
// ------------------------------------------------------------------------------------------------
constexpr int PULL_MAX_SRC = 32;
template <int NC> TFX_DEV float row_dot(const Row<NC>& a, const Row<NC>& b) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NC; i++)
#pragma unroll
    for (int e = 0; e < 8; e++) s += a.v[i][e] * b.v[i][e];
  return wave_sum(s);
}
TFX_DEV float lane_bcast(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }

// register form, kept for A/B (TFX_PULL_VARIANT=1): at most NJ sources, d w partials in registers, w rows in LDS, every row of a token requested
// at once and held in registers until used (two waves per SIMD, bytes in flight only while a wave waits).  DB: the wrapper's bias gradient (column sums
// of dy) is accumulated here (8 more registers per lane; instantiated both ways so that a caller without a bias does not carry them).  Round 4's form
// spilled 9 registers into the token loop at NJ = 8; with the segment's scale row in LDS (below) it is 246 (DB) / 238 registers, no spills
template <int NC, int NJ, bool DB> __global__ __launch_bounds__(512) void attnres_pull_reg_k(tfx_attnres_pull_args p, tfx_adaln_post_args q, int has_post) {
  extern __shared__ float dyn[];                            // [n_src][d]
  __shared__ float smem[SEG_WAVES * NC * 512];
  __shared__ tfx_attnres_src srcs[PULL_MAX_SRC];
  constexpr bool REG = NJ > 0;
  constexpr int JC = REG ? NJ : 4;
  const int lane = threadIdx.x & 63;
  const int d = p.d, ns = p.n_src;
  for (int i = threadIdx.x; i < ns * (int)(sizeof(tfx_attnres_src) / 8); i += blockDim.x)
    ((unsigned long long*)srcs)[i] = ((const unsigned long long*)p.src)[i];
  __syncthreads();
  // w rows in LDS, register form: a lane reads its 8 columns c 8 .. c 8 + 7 as two 16-byte halves - stored as [half][c][4] so that consecutive lanes are
  // 16 bytes apart in each ds_read_b128 (the natural [c][8] image puts lanes 32 bytes apart: 2-way bank conflicts on every read, 30 % of the
  // kernel's LDS cycles in profiles/r03_f_pmc_sq_attn_tokenwise.txt)
  for (int i = threadIdx.x; i < ns * d; i += blockDim.x) {
    const int j = i / d, c = i - j * d;
    if (REG) dyn[(size_t)j * d + ((c >> 2) & 1) * (d >> 1) + (c >> 3) * 4 + (c & 3)] = srcs[j].w[c];
    else dyn[i] = 0.f;
  }
  __syncthreads();
  Row<NC> pl, pb;
  Row<NC> pw[REG ? NJ : 1];
#pragma unroll
  for (int i = 0; i < NC; i++)
#pragma unroll
    for (int e = 0; e < 8; e++) {
      pl.v[i][e] = 0.f; if constexpr (DB) pb.v[i][e] = 0.f;
#pragma unroll
      for (int j = 0; j < (REG ? NJ : 1); j++) pw[j].v[i][e] = 0.f;
    }
  const bool tok_mode = p.n_seg <= 0;                        // no segments: one token per wave, per-token atomics for the table gradients
  const int n_items = tok_mode ? p.T : p.n_seg;
  for (int s = blockIdx.x * SEG_WAVES + (threadIdx.x >> 6); s < n_items; s += gridDim.x * SEG_WAVES) {
    const int t0 = tok_mode ? s : p.seg_start[s], len = tok_mode ? 1 : p.seg_len[s];
    const int inst = has_post ? q.tok_inst[t0] : -1;
    // the segment's scale row (1 + layerscale | sigmoid z) lives in the wave's own 2 KiB of `smem` (free until the closing flush, which every wave enters
    // only after its last token), as [half][lane][4] so that the two 16-byte reads per token are conflict-free: 8 registers less across the token loop,
    // where the NJ = 8 form had 9 / 3 spilled registers
    float* scl = smem + (threadIdx.x >> 6) * NC * 512;
    Row<NC> az;
    if (has_post) {
      Row<NC> sc;
      if (inst < 0) load_vec(sc, q.layerscale, d, lane);
      else load_vec(sc, q.table + (size_t)inst * q.ld_table + 2 * d, d, lane);
#pragma unroll
      for (int i = 0; i < NC; i++) {
        f32x4 s0, s1;
#pragma unroll
        for (int e = 0; e < 8; e++) { const float v = inst < 0 ? 1.f + sc.v[i][e] : sigmoidf_(sc.v[i][e]); if (e < 4) s0[e] = v; else s1[e - 4] = v; az.v[i][e] = 0.f; }
        *(f32x4*)(scl + (i * 2) * 256 + lane * 4) = s0; *(f32x4*)(scl + (i * 2 + 1) * 256 + lane * 4) = s1;
      }
    }
    const int tend = t0 + len;
    for (int t = t0; t < tend; t++) {
      // every row this token needs is requested before the first use (raw bf16: 4 * NC registers per row)
      RowRaw<NC> hr, xo, xe, ad, yr, gr[JC];
      load_raw(hr, p.h + (size_t)t * d, d, lane);
#pragma unroll
      for (int jj = 0; jj < JC; jj++)
        if (jj < ns) load_raw(gr[jj], srcs[jj].g + (size_t)t * d, d, lane);
      if (p.out_own) { load_raw(xo, p.out_own + (size_t)t * d, d, lane); if (p.out_err) load_raw(xe, p.out_err + (size_t)t * d, d, lane); }
      if (p.add) load_raw(ad, p.add + (size_t)t * d, d, lane);
      if (has_post) load_raw(yr, q.y + (size_t)t * d, d, lane);
      // lane k: the saved softmax state of source k at (t, l), and its <g, out>
      f32x4 sv = {0.f, 0.f, 0.f, 0.f};
      float dsl = 0.f;
      if (lane < ns) {
        sv = *(const f32x4*)(srcs[lane].save + ((size_t)t * srcs[lane].L + p.l) * 4);
        if (lane > 0 || !p.out_own) dsl = srcs[lane].dsum[t];
      }
      Row<NC> h, G;
      widen(h, hr);
      if (p.add) widen(G, ad);
      else {
#pragma unroll
        for (int i = 0; i < NC; i++)
#pragma unroll
          for (int e = 0; e < 8; e++) G.v[i][e] = 0.f;
      }
      float k2sum = 0.f;                                        // sum_j k2_jl: the - k2 h_l terms of all sources leave as one row pass
      for (int j0 = 0; j0 < ns; j0 += JC) {
        if (j0 > 0) {
#pragma unroll
          for (int jj = 0; jj < JC; jj++)
            if (j0 + jj < ns) load_raw(gr[jj], srcs[j0 + jj].g + (size_t)t * d, d, lane);
        }
#pragma unroll
        for (int jj = 0; jj < JC; jj++) {
          const int j = j0 + jj;
          if (j >= ns) break;
          Row<NC> gj; widen(gj, gr[jj]);
          const float da = row_dot<NC>(gj, h);
          float dsum;
          if (j == 0 && p.out_own) {
            Row<NC> o; widen(o, xo);
            if (p.out_err) {
              Row<NC> oe; widen(oe, xe);
#pragma unroll
              for (int i = 0; i < NC; i++)
#pragma unroll
                for (int e = 0; e < 8; e++) o.v[i][e] += oe.v[i][e];
            }
            dsum = row_dot<NC>(gj, o);
            if (lane == 0) { srcs[0].dsum[t] = dsum; srcs[0].save[(size_t)t * srcs[0].L * 4 + 3] = dsum; }   // (the ring form reads it from the saved state)
          } else dsum = lane_bcast(dsl, j);
          const float a = lane_bcast(sv[0], j), inv = lane_bcast(sv[1], j), sj = lane_bcast(sv[2], j);
          const float ds = a * (da - dsum);
          const float k1 = ds * inv, k2 = k1 * sj * inv;
          k2sum += k2;
          const float* wj = REG ? (dyn + (size_t)j * d) : srcs[j].w;
#pragma unroll
          for (int i = 0; i < NC; i++) {
            const int c = lane + 64 * i;
            if (c * 8 >= d) continue;
            const f32x4 w0 = *(const f32x4*)(wj + (REG ? c * 4 : c * 8)), w1 = *(const f32x4*)(wj + (REG ? (d >> 1) + c * 4 : c * 8 + 4));
#pragma unroll
            for (int e = 0; e < 8; e++) {
              const float we = e < 4 ? w0[e & 3] : w1[e & 3];
              G.v[i][e] += a * gj.v[i][e] + k1 * we;             // (- k2 h: one pass behind the sources with the summed k2)
              const float ph = k1 * h.v[i][e];
              if constexpr (REG) pw[jj].v[i][e] += ph;
              else atomicAdd(dyn + (size_t)j * d + c * 8 + e, ph);
            }
          }
        }
      }
#pragma unroll
      for (int i = 0; i < NC; i++)
#pragma unroll
        for (int e = 0; e < 8; e++) G.v[i][e] -= k2sum * h.v[i][e];
#pragma unroll
      for (int i = 0; i < NC; i++)
#pragma unroll
        for (int e = 0; e < 8; e++) G.v[i][e] = bf2f(f2bf(G.v[i][e]));     // what the next kernel reads back
      store_row(G, p.dh + (size_t)t * d, d, lane);
      if (has_post) {
        Row<NC> yy; widen(yy, yr);
        asm volatile("" ::: "memory");                            // (keeps the scale-row reads inside the loop: hoisted, they are 8 live registers again)
#pragma unroll
        for (int i = 0; i < NC; i++) {
          const int c = lane + 64 * i;
          if (c * 8 >= d) continue;
          const f32x4 s0 = *(const f32x4*)(scl + (i * 2) * 256 + lane * 4), s1 = *(const f32x4*)(scl + (i * 2 + 1) * 256 + lane * 4);
#pragma unroll
          for (int e = 0; e < 8; e++) {
            const float sce = e < 4 ? s0[e & 3] : s1[e & 3];
            const float gy = G.v[i][e] * yy.v[i][e];
            if (tok_mode && inst >= 0) atomicAdd(q.dtable + (size_t)inst * q.ld_table + 2 * d + c * 8 + e, gy * sce * (1.f - sce));
            else az.v[i][e] += gy;
            G.v[i][e] *= sce; if constexpr (DB) pb.v[i][e] += G.v[i][e];
          }
        }
        store_row(G, q.dy + (size_t)t * d, d, lane);
      }
    }
    if (has_post) {
      if (inst < 0) {
#pragma unroll
        for (int i = 0; i < NC; i++)
#pragma unroll
          for (int e = 0; e < 8; e++) pl.v[i][e] += az.v[i][e];
      } else if (!tok_mode) {
        float* dt = q.dtable + (size_t)inst * q.ld_table + 2 * d;
#pragma unroll
        for (int i = 0; i < NC; i++) {
          int c = lane + 64 * i;
          if (c * 8 >= d) continue;
          const f32x4 s0 = *(const f32x4*)(scl + (i * 2) * 256 + lane * 4), s1 = *(const f32x4*)(scl + (i * 2 + 1) * 256 + lane * 4);
          f32x4 a0, a1;
#pragma unroll
          for (int e = 0; e < 4; e++) {
            a0[e] = az.v[i][e] * s0[e] * (1.f - s0[e]);
            a1[e] = az.v[i][4 + e] * s1[e] * (1.f - s1[e]);
          }
          *(f32x4*)(dt + c * 8) = a0; *(f32x4*)(dt + c * 8 + 4) = a1;
        }
      }
    }
  }
  if (has_post) {
    flush_col_partials<NC>(pl, q.dlayerscale, d, smem);
    if constexpr (DB) { if (q.dbias) flush_col_partials<NC>(pb, q.dbias, d, smem); }
  }
  if constexpr (REG) {
#pragma unroll
    for (int j = 0; j < NJ; j++)
      if (j < ns) flush_col_partials<NC>(pw[j], srcs[j].dw, d, smem);
  } else {
    __syncthreads();
    for (int i = threadIdx.x; i < ns * d; i += blockDim.x) {
      const int j = i / d, c = i - j * d;
      const float v = dyn[i];
      if (v != 0.f) atomicAdd(srcs[j].dw + c, v);
    }
  }
}


// ------------------------------------------------------------------------------------------------
// Pull form, LDS-DMA ring (the product form).  The kernels above keep the rows of a token in registers between request and use, so a wave has
// bytes in flight only while it waits (and the register forms run at two waves per SIMD): 3.2 TB/s.  Here every row a wave will need - the
// hidden, the n_src gradient rows, the forward output and its rounding residual, the wrapper's y, the per-source scalars, the segment's scale
// row - is one entry of a fixed ORDER known in advance, and the wave streams that order through a private ring of LDS slots with
// global_load_lds_dwordx4 (no registers, no waiting): the DMA of row r + AHEAD is issued when row r is consumed, `s_waitcnt vmcnt(AHEAD * NC)`
// certifies row r (vmcnt retires in order; any other memory operation issued in between only makes the wait conservative).
//   stream of a wave:  per segment [scale row: 2 slots, fp32] (with the wrapper's output side), then per token
//                      [scalars][h][out][out residual][addend][y] (those present) [g_0] .. [g_{ns-1}]
//   d w = sum_t sum_l k1 h:  NJ > 0 - at most NJ sources, per-lane partials in registers (the j loop is unrolled);
//                            NJ == 0 - any number of sources: the k1 of every (token, source) are written out as a bf16 [T, ld_k1] matrix and
//                            d w += K1^T H is one weight-gradient GEMM behind the launch (tfx_gemm_tn; ds_add_f32 accumulators in LDS were
//                            measured at ~200 clocks per wave instruction - 4.5x the whole kernel)
//   w rows: fp32 in LDS, or bf16 (W16) when fp32 would leave the rings too few slots
// ------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) unsigned char lds_u8;
template <int NC> TFX_DEV void ring_raw(RowRaw<NC>& r, const lds_u8* slot, int lane, int d) {
#pragma unroll
  for (int i = 0; i < NC; i++) {
    if ((lane + 64 * i) * 8 < d) r.v[i] = *(const __attribute__((address_space(3))) bf16x8*)(slot + i * 1024 + lane * 16);
    else {                                                      // past the end of a row narrower than NC * 512: zeros (the DMA fetched a dummy there)
#pragma unroll
      for (int e = 0; e < 8; e++) r.v[i][e] = f2bf(0.f);
    }
  }
}
template <int NC, int AHEAD, bool W16, int NJ>
__global__ __launch_bounds__(512) void attnres_pull_dma_k(tfx_attnres_pull_args p, tfx_adaln_post_args q, int has_post) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char dynb[];
  __shared__ tfx_attnres_src srcs[PULL_MAX_SRC];
  __shared__ unsigned long long rbase[8 + PULL_MAX_SRC];          // row k >= 1 of a token: base pointer (every row advances d * 2 bytes per token)
  constexpr int RING = AHEAD + 1, SLOT = NC * 1024;
  constexpr bool REG = NJ > 0;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int d = p.d, ns = p.n_src;
  const size_t wbytes = (((size_t)ns * d * (W16 ? 2 : 4)) + 1023) & ~(size_t)1023;
  lds_u8* ring = (lds_u8*)dynb + wbytes + (size_t)wave * RING * SLOT;
  for (int i = threadIdx.x; i < ns * (int)(sizeof(tfx_attnres_src) / 8); i += blockDim.x)
    ((unsigned long long*)srcs)[i] = ((const unsigned long long*)p.src)[i];
  __syncthreads();
  // row table of a token: k = 0 scalars (special), 1 h, then the optional fixed rows, then the sources
  const int own = p.out_own != nullptr, has_err = own && p.out_err != nullptr, has_add = p.add != nullptr;
  const int k_out = 2, k_err = k_out + own, k_add = k_err + has_err, k_y = k_add + has_add, k_src = k_y + (has_post ? 1 : 0);
  const int nr = k_src + ns;
  if (threadIdx.x == 0) {
    rbase[1] = (unsigned long long)p.h;
    if (own) rbase[k_out] = (unsigned long long)p.out_own;
    if (has_err) rbase[k_err] = (unsigned long long)p.out_err;
    if (has_add) rbase[k_add] = (unsigned long long)p.add;
    if (has_post) rbase[k_y] = (unsigned long long)q.y;
  }
  for (int j = threadIdx.x; j < ns; j += blockDim.x) rbase[k_src + j] = (unsigned long long)srcs[j].g;
  for (int i = threadIdx.x; i < ns * d; i += blockDim.x) {
    const int j = i / d, c = i - j * d;
    if (W16) ((bf16*)dynb)[i] = f2bf(srcs[j].w[c]); else ((float*)dynb)[i] = srcs[j].w[c];
  }
  __syncthreads();

  const bool tok_mode = p.n_seg <= 0;
  const int n_items = tok_mode ? p.T : p.n_seg;
  const int stride_items = gridDim.x * SEG_WAVES;
  const int hdr = has_post ? -2 : 0;                            // header rows of a segment: the scale row (fp32 d floats = 2 slots)
  // ---- issue side: walks the same (segment, token, row) order AHEAD rows in front of the consumer
  int is_ = blockIdx.x * SEG_WAVES + wave, it_ = 0, itend = 0, ik = 0, islot = 0;
  const float* iscp = nullptr;                                  // scale row of the issue side's segment
  bool ilive = is_ < n_items;
  auto seg_open = [&](int s, int& t0, int& tend) {
    t0 = tok_mode ? s : p.seg_start[s];
    tend = t0 + (tok_mode ? 1 : p.seg_len[s]);
  };
  auto scale_row = [&](int t0) -> const float* {
    const int inst = q.tok_inst[t0];
    return inst < 0 ? q.layerscale : q.table + (size_t)inst * q.ld_table + 2 * d;
  };
  if (ilive) { seg_open(is_, it_, itend); ik = hdr; if (has_post) iscp = scale_row(it_); }
  auto issue_one = [&]() {
    lds_u8* dst = ring + islot * SLOT;
    islot = islot + 1 == RING ? 0 : islot + 1;
#pragma unroll
    for (int pc = 0; pc < NC; pc++) {
      const bf16* gp = p.h;                                     // (any valid address: what lands is never read)
      if (ilive) {
        if (ik < 0) {                                           // scale row, half (ik + 2): floats [half * d/2, ...)
          const int f = pc * 256 + lane * 4;
          if (f < (d >> 1)) gp = (const bf16*)(iscp + (ik + 2) * (d >> 1) + f);
        } else if (ik == 0) {                                   // lane k: saved state of source k at (t, l); lane 32 + k: its entry 0 (carries <g, out>)
          if (pc == 0 && lane < ns) gp = (const bf16*)(srcs[lane].save + ((size_t)it_ * srcs[lane].L + p.l) * 4);
          else if (pc == 0 && lane >= 32 && lane - 32 < ns) gp = (const bf16*)(srcs[lane - 32].save + (size_t)it_ * srcs[lane - 32].L * 4);
        } else {
          const int col = (pc * 64 + lane) * 8;
          if (col < d) gp = (const bf16*)rbase[ik] + (size_t)it_ * d + col;
        }
      }
      glds16_asm(gp, (const bf16*)(dst + pc * 1024));
    }
    if (ilive) {
      ik++;
      if (ik == nr) {
        it_++;
        if (it_ < itend) ik = 0;
        else {
          is_ += stride_items;
          ilive = is_ < n_items;
          if (ilive) { seg_open(is_, it_, itend); ik = hdr; if (has_post) iscp = scale_row(it_); }
        }
      }
    }
  };
  int cslot = 0;
  auto acquire = [&]() -> const lds_u8* {
    issue_one();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AHEAD * NC) : "memory");
    const lds_u8* sp = ring + cslot * SLOT;
    cslot = cslot + 1 == RING ? 0 : cslot + 1;
    return sp;
  };
#pragma unroll 1
  for (int i = 0; i < AHEAD; i++) issue_one();

  Row<NC> pl, pb;
  Row<NC> pw[REG ? NJ : 1];
#pragma unroll
  for (int i = 0; i < NC; i++)
#pragma unroll
    for (int e = 0; e < 8; e++) {
      pl.v[i][e] = 0.f; pb.v[i][e] = 0.f;
#pragma unroll
      for (int j = 0; j < (REG ? NJ : 1); j++) pw[j].v[i][e] = 0.f;
    }
  for (int s = blockIdx.x * SEG_WAVES + wave; s < n_items; s += stride_items) {
    int t0, tend; seg_open(s, t0, tend);
    const int inst = has_post ? q.tok_inst[t0] : -1;
    Row<NC> sc, az;
    if (has_post) {
      // (a slot is valid until the NEXT acquire - that one re-issues into it: each half is read right behind its own acquire)
#pragma unroll
      for (int half = 0; half < 2; half++) {
        const lds_u8* sh = acquire();
#pragma unroll
        for (int i = 0; i < NC; i++) {
          const int f0 = (lane + 64 * i) * 8 - half * (d >> 1);   // this lane's first column, relative to the half
          if (f0 >= 0 && f0 < (d >> 1)) {
            const f32x4 a = *(const __attribute__((address_space(3))) f32x4*)(sh + f0 * 4), b = *(const __attribute__((address_space(3))) f32x4*)(sh + f0 * 4 + 16);
#pragma unroll
            for (int e = 0; e < 4; e++) { sc.v[i][e] = a[e]; sc.v[i][4 + e] = b[e]; }
          }
        }
      }
#pragma unroll
      for (int i = 0; i < NC; i++)
#pragma unroll
        for (int e = 0; e < 8; e++) {
          const float z = (lane + 64 * i) * 8 < d ? sc.v[i][e] : 0.f;
          sc.v[i][e] = inst < 0 ? 1.f + z : sigmoidf_(z); az.v[i][e] = 0.f;
        }
    }
    for (int t = t0; t < tend; t++) {
      const f32x4 sv = *(const __attribute__((address_space(3))) f32x4*)(acquire() + lane * 16);
      RowRaw<NC> rr;
      Row<NC> h, o, G, yy;
      ring_raw(rr, acquire(), lane, d); widen(h, rr);
      if (own) {
        ring_raw(rr, acquire(), lane, d); widen(o, rr);
        if (has_err) {
          Row<NC> oe; ring_raw(rr, acquire(), lane, d); widen(oe, rr);
#pragma unroll
          for (int i = 0; i < NC; i++)
#pragma unroll
            for (int e = 0; e < 8; e++) o.v[i][e] += oe.v[i][e];
        }
      }
      if (has_add) { ring_raw(rr, acquire(), lane, d); widen(G, rr); }
      else {
#pragma unroll
        for (int i = 0; i < NC; i++)
#pragma unroll
          for (int e = 0; e < 8; e++) G.v[i][e] = 0.f;
      }
      if (has_post) { ring_raw(rr, acquire(), lane, d); widen(yy, rr); }
      float k1_l = 0.f;                                         // lane j: k1 of source j (exported when NJ == 0)
      float k2sum = 0.f;                                        // sum_j k2_jl: the - k2 h_l terms of all sources leave as one row pass
      auto one_source = [&](int j, Row<NC>& pwj) {
        Row<NC> gj; ring_raw(rr, acquire(), lane, d); widen(gj, rr);
        const float da = row_dot<NC>(gj, h);
        float dsum;
        if (j == 0 && own) {
          // <g, out> with the forward output as it was BEFORE rounding to bf16 (stored row + stored residual): the score gradient
          // a_l (<g, h_l> - <g, out>) cancels to the difference of nearby hiddens, which the rounding of `out` alone would swamp
          dsum = row_dot<NC>(gj, o);
          if (lane == 0) { srcs[0].dsum[t] = dsum; srcs[0].save[(size_t)t * srcs[0].L * 4 + 3] = dsum; }
        } else dsum = lane_bcast(sv[3], 32 + j);
        const float a = lane_bcast(sv[0], j), inv = lane_bcast(sv[1], j), sj = lane_bcast(sv[2], j);
        const float ds = a * (da - dsum);
        const float k1 = ds * inv, k2 = k1 * sj * inv;
        k2sum += k2;
        if (!REG && lane == j) k1_l = k1;
#pragma unroll
        for (int i = 0; i < NC; i++) {
          const int c = lane + 64 * i;
          if (c * 8 >= d) continue;
          float w[8];
          if (W16) {
            const bf16x8 w8 = *(const bf16x8*)((const bf16*)dynb + (size_t)j * d + c * 8);
#pragma unroll
            for (int e = 0; e < 8; e++) w[e] = bf2f(w8[e]);
          } else {
            const float* wj = (const float*)dynb + (size_t)j * d + c * 8;
            const f32x4 w0 = *(const f32x4*)wj, w1 = *(const f32x4*)(wj + 4);
#pragma unroll
            for (int e = 0; e < 4; e++) { w[e] = w0[e]; w[4 + e] = w1[e]; }
          }
#pragma unroll
          for (int e = 0; e < 8; e++) {
            G.v[i][e] += a * gj.v[i][e] + k1 * w[e];             // (- k2 h: one pass behind the sources with the summed k2)
            if (REG) pwj.v[i][e] += k1 * h.v[i][e];
          }
        }
      };
      if constexpr (REG) {
#pragma unroll
        for (int j = 0; j < NJ; j++)
          if (j < ns) one_source(j, pw[j]);
      } else {
#pragma unroll 1
        for (int j = 0; j < ns; j++) one_source(j, pw[0]);
        if (lane < ns) p.k1[(size_t)t * p.ld_k1 + lane] = f2bf(k1_l);
      }
#pragma unroll
      for (int i = 0; i < NC; i++)
#pragma unroll
        for (int e = 0; e < 8; e++) G.v[i][e] -= k2sum * h.v[i][e];
#pragma unroll
      for (int i = 0; i < NC; i++)
#pragma unroll
        for (int e = 0; e < 8; e++) G.v[i][e] = bf2f(f2bf(G.v[i][e]));     // what the next kernel reads back
      store_row(G, p.dh + (size_t)t * d, d, lane);
      if (has_post) {
#pragma unroll
        for (int i = 0; i < NC; i++) {
          const int c = lane + 64 * i;
          if (c * 8 >= d) continue;
#pragma unroll
          for (int e = 0; e < 8; e++) {
            const float gy = G.v[i][e] * yy.v[i][e];
            if (tok_mode && inst >= 0) atomicAdd(q.dtable + (size_t)inst * q.ld_table + 2 * d + c * 8 + e, gy * sc.v[i][e] * (1.f - sc.v[i][e]));
            else az.v[i][e] += gy;
            G.v[i][e] *= sc.v[i][e]; pb.v[i][e] += G.v[i][e];
          }
        }
        store_row(G, q.dy + (size_t)t * d, d, lane);
      }
    }
    if (has_post) {
      if (inst < 0) {
#pragma unroll
        for (int i = 0; i < NC; i++)
#pragma unroll
          for (int e = 0; e < 8; e++) pl.v[i][e] += az.v[i][e];
      } else if (!tok_mode) {
        float* dt = q.dtable + (size_t)inst * q.ld_table + 2 * d;
#pragma unroll
        for (int i = 0; i < NC; i++) {
          int c = lane + 64 * i;
          if (c * 8 >= d) continue;
          f32x4 a0, a1;
#pragma unroll
          for (int e = 0; e < 4; e++) {
            a0[e] = az.v[i][e] * sc.v[i][e] * (1.f - sc.v[i][e]);
            a1[e] = az.v[i][4 + e] * sc.v[i][4 + e] * (1.f - sc.v[i][4 + e]);
          }
          *(f32x4*)(dt + c * 8) = a0; *(f32x4*)(dt + c * 8 + 4) = a1;
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the trailing (unread) DMAs land before the ring area is reused below
  __syncthreads();
  float* smem = (float*)(dynb + wbytes);                       // the rings are idle now: staging area of the closing reductions (8 * NC * 2 KiB <= 8 rings)
  if (has_post) {
    flush_col_partials<NC>(pl, q.dlayerscale, d, smem);
    if (q.dbias) flush_col_partials<NC>(pb, q.dbias, d, smem);
  }
  if constexpr (REG) {
#pragma unroll
    for (int j = 0; j < NJ; j++)
      if (j < ns) flush_col_partials<NC>(pw[j], srcs[j].dw, d, smem);
  }
}

__global__ __launch_bounds__(256) void attnres_prep_k(const tfx_attnres_src* src, int d) {
  const tfx_attnres_src sj = src[blockIdx.y];
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < d) { sj.w[c] = (1.f + sj.gamma[c]) * sj.pq[c]; sj.dw[c] = 0.f; }
}
__global__ __launch_bounds__(256) void attnres_finish_k(const tfx_attnres_src* src, int d) {
  const tfx_attnres_src sj = src[blockIdx.y];
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < d) { const float dw = sj.dw[c]; sj.dgamma[c] += dw * sj.pq[c]; sj.dpq[c] += dw * (1.f + sj.gamma[c]); }
}

// ------------------------------------------------------------------------------------------------
// text embedding gather / scatter-add                                                 T:3173-3184
// ------------------------------------------------------------------------------------------------
template <int NC> __global__ __launch_bounds__(256) void embed_fwd_k(tfx_embed_args p) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * WAVES + (threadIdx.x >> 6);
  if (t >= p.T || p.tok_inst[t] >= 0) return;
  const int id = max(p.text_ids[t], 0);
#pragma unroll
  for (int i = 0; i < NC; i++) {
    int c = lane + 64 * i;
    if (c * 8 < p.d) *(bf16x8*)(p.x + (size_t)t * p.d + c * 8) = *(const bf16x8*)(p.table + (size_t)id * p.d + c * 8);
  }
}
template <int NC> __global__ __launch_bounds__(256) void embed_bwd_k(tfx_embed_args p) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * WAVES + (threadIdx.x >> 6);
  if (t >= p.T || p.tok_inst[t] >= 0) return;
  const int id = max(p.text_ids[t], 0);
  Row<NC> g; load_row(g, p.dx + (size_t)t * p.d, p.d, lane);
#pragma unroll
  for (int i = 0; i < NC; i++) {
    int c = lane + 64 * i;
    if (c * 8 < p.d)
#pragma unroll
      for (int e = 0; e < 8; e++) atomicAdd(p.dtable + (size_t)id * p.d + c * 8 + e, g.v[i][e]);
  }
}

// ------------------------------------------------------------------------------------------------
// qk RMSNorm + RoPE (+ q scale)                                       T:950-952, T:965, T:998
// 8 threads per 64-wide head vector, 8 contiguous elements (4 rotary pairs) per thread
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void qk_norm_rope_fwd_k(tfx_qk_norm_rope_args p) {
  // 32-bit index math (the launcher checks T * 2H * 8 < 2^31): a 64-bit divide by a run-time value costs ~100 instructions
  const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned vid = gid >> 3;
  const int sub = gid & 7;
  const unsigned twoH = (p.cache ? 3u : 2u) * p.H, nvec = (unsigned)p.T * twoH;      // with a cache the v vectors ride along (copied, not normalised)
  if (p.sc_plan && blockIdx.x == 0 && threadIdx.x < 64)                              // the layer's soft-cap plan (tfx.h), one wave
    softcap_plan_write(p.gamma_q, p.gamma_k, p.norm_scale, p.q_scale, p.softcap, p.sc_plan);
  if (vid >= nvec) return;
  const int t = (int)(vid / twoH), rem = (int)(vid - (unsigned)t * twoH);
  const int which = rem >= p.H;
  const int col = rem * 64 + sub * 8;          // which*H*64 + h*64 + sub*8
  bf16x8 x = *(const bf16x8*)(p.qkv + (size_t)t * p.ld_qkv + col);
  if (rem >= 2 * p.H) {                        // v: cache append only
    const int cp = p.cache_pos[t];
    if (cp >= 0) *(bf16x8*)(p.cache + (size_t)cp * p.ld_cache + (col - p.H * 64)) = x;
    return;
  }
  const float* gm = (which == 0 ? p.gamma_q : p.gamma_k) + sub * 8;
  const int pos = p.rot_pos[t];
  const f32x4 cs = *(const f32x4*)(p.cos_tab + (size_t)pos * 32 + sub * 4);
  const f32x4 sn = *(const f32x4*)(p.sin_tab + (size_t)pos * 32 + sub * 4);
  const bf16x8 o = qk_norm_rope_chunk(x, p.norm_scale > 0.f ? p.norm_scale : 8.f, which == 0 ? p.q_scale : 1.f, gm, cs, sn);
  *(bf16x8*)(p.qk + (size_t)t * p.ld_qk + col) = o;
  if (p.cache && which) {
    const int cp = p.cache_pos[t];
    if (cp >= 0) *(bf16x8*)(p.cache + (size_t)cp * p.ld_cache + (col - p.H * 64)) = o;
  }
}

__global__ __launch_bounds__(1024) void qk_norm_rope_bwd_k(tfx_qk_norm_rope_args p) {
  __shared__ float sg[16][2][64];                      // per wave: the gamma gradients of q / k, one value per head column
  const int sub = threadIdx.x & 7;      // constant per thread (grid stride is a multiple of 8 threads)
  float pq[8], pk[8];
#pragma unroll
  for (int e = 0; e < 8; e++) { pq[e] = 0.f; pk[e] = 0.f; }
  // incremental (t, rem) instead of a divide per trip; 32-bit (the launcher checks T * 2H * 8 < 2^31)
  const int twoH = 2 * p.H, nvec_i = p.T * twoH;
  const int stride = (int)((gridDim.x * blockDim.x) >> 3), st_t = stride / twoH, st_r = stride - st_t * twoH;
  int vid = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 3);
  int t = vid / twoH, rem = vid - t * twoH;
  for (; vid < nvec_i; vid += stride, t += st_t, rem += st_r) {
    if (rem >= twoH) { rem -= twoH; t++; }
    const int which = rem >= p.H;
    const int col = rem * 64 + sub * 8;
    bf16x8 x = *(const bf16x8*)(p.qkv + (size_t)t * p.ld_qkv + col);
    bf16x8 dy8 = *(const bf16x8*)(p.dqk + (size_t)t * p.ld_dqk + col);
    float v[8], q = 0.f;
#pragma unroll
    for (int e = 0; e < 8; e++) { v[e] = bf2f(x[e]); q += v[e] * v[e]; }
    q = group8_sum(q);
    const float nrm = fmaxf(sqrtf(q), 1e-12f), inv = 1.f / nrm;
    const float sc = (which == 0 ? p.q_scale : 1.f) * (p.norm_scale > 0.f ? p.norm_scale : 8.f);
    const float* gm = (which == 0 ? p.gamma_q : p.gamma_k) + sub * 8;
    const int pos = p.rot_pos[t];
    const f32x4 cs = *(const f32x4*)(p.cos_tab + (size_t)pos * 32 + sub * 4);
    const f32x4 sn = *(const f32x4*)(p.sin_tab + (size_t)pos * 32 + sub * 4);
    float dyn[8], S = 0.f;      // dyn = grad wrt (v/nrm), i.e. includes c = sc*(1+gamma)
#pragma unroll
    for (int i = 0; i < 4; i++) {
      float da = bf2f(dy8[2 * i]), db = bf2f(dy8[2 * i + 1]);
      float ga = da * cs[i] + db * sn[i];          // inverse rotation
      float gb = db * cs[i] - da * sn[i];
      float ca = sc * (1.f + gm[2 * i]), cb = sc * (1.f + gm[2 * i + 1]);
      float dga = ga * v[2 * i] * (inv * sc), dgb = gb * v[2 * i + 1] * (inv * sc);
      pq[2 * i] += which == 0 ? dga : 0.f; pq[2 * i + 1] += which == 0 ? dgb : 0.f;
      pk[2 * i] += which == 0 ? 0.f : dga; pk[2 * i + 1] += which == 0 ? 0.f : dgb;
      dyn[2 * i] = ga * ca; dyn[2 * i + 1] = gb * cb;
      S += dyn[2 * i] * v[2 * i] + dyn[2 * i + 1] * v[2 * i + 1];
    }
    S = group8_sum(S);
    const float k = S * inv * inv * inv;
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; e++) o[e] = f2bf(dyn[e] * inv - v[e] * k);
    *(bf16x8*)(p.dqkv + (size_t)t * p.ld_dqkv + col) = o;
  }
  // the 8 lanes of a wave that own the same 8 columns (lane & 7) are summed with three cross-lane exchanges, the four waves through plain LDS
  // stores: 16 LDS float atomics per thread with 8 lanes on every address were a ~10 us tail on a 100 us launch (ds_add_f32 with conflicts
  // runs at hundreds of clocks per instruction - measured for the AttentionResidual backward earlier this round)
#pragma unroll
  for (int e = 0; e < 8; e++) {
#pragma unroll
    for (int m = 8; m < 64; m <<= 1) { pq[e] += __shfl_xor(pq[e], m, 64); pk[e] += __shfl_xor(pk[e], m, 64); }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane < 8) {
#pragma unroll
    for (int e = 0; e < 8; e++) { sg[wv][0][lane * 8 + e] = pq[e]; sg[wv][1][lane * 8 + e] = pk[e]; }
  }
  __syncthreads();
  if (threadIdx.x < 128) {
    const int which = threadIdx.x >> 6, c = threadIdx.x & 63;
    float s = 0.f;
    const int nw = blockDim.x >> 6;
    for (int i = 0; i < nw; i++) s += sg[i][which][c];
    if (s != 0.f) atomicAdd((which == 0 ? p.dgamma_q : p.dgamma_k) + c, s);
  }
}

// ------------------------------------------------------------------------------------------------
// small elementwise / reduction kernels
// ------------------------------------------------------------------------------------------------
__global__ void noise_mix_k(tfx_noise_mix_args p) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long n = (long long)p.R * p.ld_xt;
  if (i >= n) return;
  const int r = (int)(i / p.ld_xt), c = (int)(i % p.ld_xt);
  if (c >= p.dl) { p.xt[i] = f2bf(0.f); return; }
  const float x = p.x[(size_t)r * p.dl + c];
  if (!p.eps) { p.xt[i] = f2bf(x); return; }
  const float e = p.eps[(size_t)r * p.dl + c];
  const float t = p.inst_time[p.row_inst[r]];
  p.xt[i] = f2bf(x * t + e * (1.f - t));
  if (p.flow) p.flow[(size_t)r * p.dl + c] = x - e;
}

__global__ void fourier_k(tfx_fourier_args p) {
  const int i = blockIdx.x, c = threadIdx.x;
  const float t = p.times[i];
  for (int col = c; col < p.ld; col += blockDim.x) {
    float v = 0.f;
    if (col == 0) v = t;
    else if (col <= p.half) v = sinf(t * p.w[col - 1] * 6.283185307179586f);
    else if (col <= 2 * p.half) v = cosf(t * p.w[col - 1 - p.half] * 6.283185307179586f);
    p.out[(size_t)i * p.ld + col] = f2bf(v);
  }
}

__global__ __launch_bounds__(256) void ce_k(tfx_ce_args p) {
  // one wave per token, grid-stride over tokens so that each block issues ONE pair of atomics (the per-4-token
  // atomics of the first version serialised in L2 and set the kernel's duration)
  __shared__ float sacc[2][WAVES];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float loss = 0.f, cnt = 0.f;
  // rows of at most 512 logits (the text vocabulary is 256 + specials): the row is read ONCE, 16 bytes per lane, and stays in registers for the
  // maximum, the exponentials and the gradient - the three strided passes below read it three times, 4 bytes per lane (101 us -> under 56 us per launch at 65536 x 264)
  const bool fast = p.ld <= 512 && p.ld_d <= 512 && (p.ld & 3) == 0 && (p.ld_d & 3) == 0 && (((uintptr_t)p.logits) & 15) == 0 && (((uintptr_t)p.dlogits) & 7) == 0;
  if (fast) {
    for (int t = blockIdx.x * WAVES + w; t < p.T; t += gridDim.x * WAVES) {
      const int lab = p.labels[t];
      const float* lg = p.logits + (size_t)t * p.ld;
      bf16* dl = p.dlogits + (size_t)t * p.ld_d;
      f32x4 v[2];
      float mx = -INFINITY;
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const int c = k * 256 + lane * 4;
        const f32x4 ninf = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        v[k] = ninf;
        if (lab >= 0 && c < p.ld) v[k] = *(const f32x4*)(lg + c);
#pragma unroll
        for (int e = 0; e < 4; e++) { if (c + e >= p.V) v[k][e] = -INFINITY; mx = fmaxf(mx, v[k][e]); }
      }
      float lse = 0.f;
      if (lab >= 0) {
        mx = wave_max(mx);
        float se = 0.f;
#pragma unroll
        for (int k = 0; k < 2; k++)
#pragma unroll
          for (int e = 0; e < 4; e++) se += __expf(v[k][e] - mx);                 // exp(-inf) = 0 for the columns past V
        se = wave_sum(se);
        lse = mx + __logf(se);
        loss += lse - lg[lab]; cnt += 1.f;
      }
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const int c = k * 256 + lane * 4;
        if (c < p.ld_d) {
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; e++) {
            float g = 0.f;
            if (lab >= 0 && c + e < p.V) g = (__expf(v[k][e] - lse) - (c + e == lab ? 1.f : 0.f)) * p.grad_scale;
            o[e] = f2bf(g);
          }
          *(bf16x4*)(dl + c) = o;
        }
      }
    }
  } else
  for (int t = blockIdx.x * WAVES + w; t < p.T; t += gridDim.x * WAVES) {
    const int lab = p.labels[t];
    const float* lg = p.logits + (size_t)t * p.ld;
    bf16* dl = p.dlogits + (size_t)t * p.ld_d;
    if (lab < 0) {
      for (int c = lane; c < p.ld_d; c += 64) dl[c] = f2bf(0.f);
      continue;
    }
    float mx = -INFINITY;
    for (int c = lane; c < p.V; c += 64) mx = fmaxf(mx, lg[c]);
    mx = wave_max(mx);
    float se = 0.f;
    for (int c = lane; c < p.V; c += 64) se += __expf(lg[c] - mx);
    se = wave_sum(se);
    const float lse = mx + __logf(se);
    for (int c = lane; c < p.ld_d; c += 64) {
      float g = 0.f;
      if (c < p.V) g = (__expf(lg[c] - lse) - (c == lab ? 1.f : 0.f)) * p.grad_scale;
      dl[c] = f2bf(g);
    }
    loss += lse - lg[lab]; cnt += 1.f;
  }
  if (lane == 0) { sacc[0][w] = loss; sacc[1][w] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.f, b = 0.f;
    for (int i = 0; i < WAVES; i++) { a += sacc[0][i]; b += sacc[1][i]; }
    if (b > 0.f) { atomicAdd(p.acc, a); atomicAdd(p.acc + 1, b); }
  }
}

__global__ __launch_bounds__(256) void mse_k(tfx_mse_args p) {
  __shared__ float sacc[WAVES];
  const long long n = (long long)p.R * p.ld_d;
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / p.ld_d), c = (int)(i % p.ld_d);
    float g = 0.f;
    if (c < p.dl) {
      const float pr = p.pred[(size_t)r * p.ld_pred + c], fl = p.flow[(size_t)r * p.dl + c];
      if (p.recon_w) {
        const float t = p.recon_time[p.recon_inst[r]], w = p.recon_w[r];
        const float df = (1.f - t) * pr - (p.recon_mode ? 1.f : t) * fl;
        s += w * df * df; g = df * (p.grad_scale * w * (1.f - t));
      } else {
        const float df = pr - fl;
        s += df * df; g = df * p.grad_scale;
      }
      if (p.row_inst) g *= 1.f / fmaxf(1.f - p.inst_time[p.row_inst[r]], p.clean_eps);
    }
    p.dpred[i] = f2bf(p.accumulate ? g + bf2f(p.dpred[i]) : g);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sacc[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) { float a = 0.f; for (int i = 0; i < WAVES; i++) a += sacc[i]; atomicAdd(p.acc, a); }
}

TFX_DEV void cast_rows_body(const tfx_cast_args& p, long long blk) {
  const long long i = blk * 256 + threadIdx.x;
  if (i >= (long long)p.Rd * p.ld_dst) return;
  const int r = (int)(i / p.ld_dst), c = (int)(i % p.ld_dst);
  float v = 0.f;
  if (c < p.Cs && c < p.Cd) {
    int rs = p.rowmap ? p.rowmap[r] : r;
    if (rs >= 0 && rs < p.Rs) v = p.src[(size_t)rs * p.ld_src + c];
  }
  p.dst[i] = f2bf(v);
}
// dst[c][r] = src[map(r)][c];  dst has Rd (= padded Cs) rows and ld_dst >= Cd (= padded #r) columns
TFX_DEV void cast_rows_t_body(const tfx_cast_args& p, int bx, int by, float (*tile)[33]) {
  const int r0 = bx * 32, c0 = by * 32;                     // r: dst column index, c: dst row index
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: 32 x 8
  for (int j = ty; j < 32; j += 8) {
    int r = r0 + j, c = c0 + tx;
    float v = 0.f;
    if (r < p.Cd && c < p.Cs) {
      int rs = p.rowmap ? p.rowmap[r] : r;
      if (rs >= 0 && rs < p.Rs) v = p.src[(size_t)rs * p.ld_src + c];
    }
    tile[j][tx] = v;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    int c = c0 + j, r = r0 + tx;
    if (c < p.Rd && r < p.ld_dst) p.dst[(size_t)c * p.ld_dst + r] = f2bf(tile[tx][j]);
  }
}
__global__ __launch_bounds__(256) void cast_rows_k(tfx_cast_args p) { cast_rows_body(p, blockIdx.x); }
__global__ __launch_bounds__(256) void cast_rows_t_k(tfx_cast_args p) {
  __shared__ float tile[32][33];
  cast_rows_t_body(p, blockIdx.x, blockIdx.y, tile);
}
// every shadow of the parameter set in ONE launch: block -> job by binary search over the jobs' first blocks.
// plain jobs: 8 elements per thread (2 x 16-B loads, one 16-B store); transposed jobs: 64x64 tiles through LDS
// (256-B read segments, 128-B write segments).
TFX_DEV void cast8_body(const tfx_cast_args& p, long long blk) {
  const long long i = (blk * 256 + threadIdx.x) * 8;
  if (i >= (long long)p.Rd * p.ld_dst) return;
  const int r = (int)(i / p.ld_dst), c = (int)(i % p.ld_dst);          // ld_dst % 8 == 0: the 8 elements share a row
  const int rs = p.rowmap ? p.rowmap[r] : r;
  const int cmax = min(p.Cs, p.Cd);
  bf16x8 o;
  if (rs >= 0 && rs < p.Rs) {
    const float* sp = p.src + (size_t)rs * p.ld_src + c;
    if (c + 8 <= cmax && (((uintptr_t)sp) & 15) == 0) {
      const f32x4 a = *(const f32x4*)sp, b = *(const f32x4*)(sp + 4);
#pragma unroll
      for (int e = 0; e < 4; e++) { o[e] = f2bf(a[e]); o[4 + e] = f2bf(b[e]); }
    } else {
#pragma unroll
      for (int e = 0; e < 8; e++) o[e] = f2bf(c + e < cmax ? sp[e] : 0.f);
    }
  } else {
#pragma unroll
    for (int e = 0; e < 8; e++) o[e] = f2bf(0.f);
  }
  *(bf16x8*)(p.dst + i) = o;
}
TFX_DEV void cast64_t_body(const tfx_cast_args& p, int bx, int by, float (*tile)[65]) {
  // dst[c][r] = src[map(r)][c] over a 64 (r) x 64 (c) tile.  Loads: every thread owns one 4-column group of four rows (r = rr + 16 k) - the four
  // row-map entries first, then four 16-byte loads back to back (scalar where a row start is not 16-byte aligned or the tile hangs over Cs);
  // stores: 16 bytes = 8 consecutive r of one dst row per thread and pass.  (Round 2's form read 4 bytes per lane with the row-map load inside
  // the loop - two dependent round trips per iteration - and wrote 4 bytes per lane: 2.2 ms for the 604 M parameters of dim 1024 / depth 24.)
  const int r0 = bx * 64, c0 = by * 64;                     // r: dst column index, c: dst row index
  {
    const int cq = threadIdx.x & 15, rr = threadIdx.x >> 4;
    const int c = c0 + 4 * cq;
    int rs[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int r = r0 + rr + 16 * k;
      rs[k] = r < p.Cd ? (p.rowmap ? p.rowmap[r] : r) : -1;
      if (rs[k] >= p.Rs) rs[k] = -1;
    }
    f32x4 v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      v[k] = z;
      if (rs[k] >= 0 && c < p.Cs) {
        const float* sp = p.src + (size_t)rs[k] * p.ld_src + c;
        if (c + 4 <= p.Cs && (((uintptr_t)sp) & 15) == 0) v[k] = *(const f32x4*)sp;
        else {
#pragma unroll
          for (int e = 0; e < 4; e++) if (c + e < p.Cs) v[k][e] = sp[e];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
      for (int e = 0; e < 4; e++) tile[rr + 16 * k][4 * cq + e] = v[k][e];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const int q = threadIdx.x + 256 * k, row = q >> 3, ch = q & 7;
    const int c = c0 + row, r = r0 + 8 * ch;
    if (c < p.Rd && r < p.ld_dst) {                          // ld_dst % 8 == 0: the 8 elements stay inside the row
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; e++) o[e] = f2bf(tile[8 * ch + e][row]);
      *(bf16x8*)(p.dst + (size_t)c * p.ld_dst + r) = o;
    }
  }
}
__global__ __launch_bounds__(256) void cast_batch_k(const tfx_cast_job* jobs, int n_jobs) {
  __shared__ float tile[64][65];
  int lo = 0, hi = n_jobs - 1;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (jobs[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1; }
  const tfx_cast_job j = jobs[lo];
  const tfx_cast_args p = {j.src, j.ld_src, j.Rs, j.Cs, j.rowmap, j.dst, j.ld_dst, j.Rd, j.Cd};
  const int lb = blockIdx.x - j.first_block;
  if (j.transposed) { const int tx = (p.ld_dst + 63) / 64; cast64_t_body(p, lb % tx, lb / tx, tile); }
  else cast8_body(p, lb);
}
// one-hot rows of the text tokens (zero rows for modality tokens): the embedding gradient becomes a TN GEMM
__global__ void onehot_k(const int* ids, const int* tok_inst, bf16* out, int T, int ld) {
  const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (i >= (long long)T * ld) return;
  const int t = (int)(i / ld), c = (int)(i % ld);
  const int id = tok_inst[t] < 0 ? max(ids[t], 0) : -1;
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; e++) o[e] = f2bf(c + e == id ? 1.f : 0.f);
  *(bf16x8*)(out + i) = o;
}
__global__ void scatter_rows_k(const bf16* src, int ld_src, int cols, bf16* dst, int ld_dst, const int* rowmap, int R) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int cpr = cols / 8;
  if (i >= (long long)R * cpr) return;
  const int r = (int)(i / cpr), c = (int)(i % cpr) * 8;
  const int ro = rowmap[r];
  if (ro >= 0) *(bf16x8*)(dst + (size_t)ro * ld_dst + c) = *(const bf16x8*)(src + (size_t)r * ld_src + c);
}
__global__ void gather_f32_k(const float* src, const int* map, float* dst, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = map[i] >= 0 ? src[map[i]] : 0.f;
}
__global__ void f32_to_bf16_k(const float* src, bf16* dst, long long n) {
  long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < n) {
    f32x4 v = *(const f32x4*)(src + i);
    bf16x4 o; o[0] = f2bf(v[0]); o[1] = f2bf(v[1]); o[2] = f2bf(v[2]); o[3] = f2bf(v[3]);
    *(bf16x4*)(dst + i) = o;
  } else for (; i < n; i++) dst[i] = f2bf(src[i]);
}
__global__ void silu_bwd_k(const bf16* dy, const bf16* pre, bf16* dx, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float x = bf2f(pre[i]), s = sigmoidf_(x);
  dx[i] = f2bf(bf2f(dy[i]) * s * (1.f + x * (1.f - s)));
}
__global__ void add_bf16_k(const bf16* a, const bf16* b, bf16* o, long long n) {
  long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (i + 7 < n) {
    bf16x8 x = *(const bf16x8*)(a + i), y = *(const bf16x8*)(b + i), r;
#pragma unroll
    for (int e = 0; e < 8; e++) r[e] = f2bf(bf2f(x[e]) + bf2f(y[e]));
    *(bf16x8*)(o + i) = r;
  } else for (; i < n; i++) o[i] = f2bf(bf2f(a[i]) + bf2f(b[i]));
}

__global__ void scale_bf16_dev_k(bf16* x, long long n, const float* scale) {
  const float sc = *scale;
  if (sc == 1.f) return;                                   // the default upstream gradient: nothing to do
  long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (i + 7 < n) {
    bf16x8 v = *(const bf16x8*)(x + i);
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = f2bf(bf2f(v[e]) * sc);
    *(bf16x8*)(x + i) = v;
  } else for (; i < n; i++) x[i] = f2bf(bf2f(x[i]) * sc);
}

template <typename T> __global__ __launch_bounds__(256) void colsum_k(const T* src, int ld, int R, int C, const int* colmap, const int* rowmap, float* out, int rows_per_block) {
  __shared__ float s[4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cx;
  const int rbeg = blockIdx.y * rows_per_block, rend = min(R, rbeg + rows_per_block);
  float a = 0.f;
  if (c < C) for (int r = rbeg + ry; r < rend; r += 4) a += (float)src[(size_t)(rowmap ? rowmap[r] : r) * ld + c];
  s[ry][cx] = a;
  __syncthreads();
  if (ry == 0 && c < C) {
    float v = s[0][cx] + s[1][cx] + s[2][cx] + s[3][cx];
    int co = colmap ? colmap[c] : c;
    if (co >= 0 && v != 0.f) atomicAdd(out + co, v);
  }
}

// bf16 column sums, 16-byte loads: a thread owns 8 consecutive columns, 8 row lanes per 512-thread block, 8 independent
// loads in flight per thread (HBM latency x bandwidth needs >= 16 MB outstanding across the chip)
__global__ __launch_bounds__(512) void colsum8_k(const bf16* src, int ld, int R, int C, const int* colmap, const int* rowmap, float* out, int rows_per_block) {
  __shared__ float s[8][64][9];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int c0 = (blockIdx.x * 64 + cx) * 8;
  const int rbeg = blockIdx.y * rows_per_block, rend = min(R, rbeg + rows_per_block);
  float a[8];
#pragma unroll
  for (int e = 0; e < 8; e++) a[e] = 0.f;
  if (c0 < C) {
    int r = rbeg + ry;
    for (; r + 56 < rend; r += 64) {
      bf16x8 v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = *(const bf16x8*)(src + (size_t)(rowmap ? rowmap[r + 8 * u] : r + 8 * u) * ld + c0);
#pragma unroll
      for (int u = 0; u < 8; u++)
#pragma unroll
        for (int e = 0; e < 8; e++) a[e] += bf2f(v[u][e]);
    }
    for (; r < rend; r += 8) {
      bf16x8 v = *(const bf16x8*)(src + (size_t)(rowmap ? rowmap[r] : r) * ld + c0);
#pragma unroll
      for (int e = 0; e < 8; e++) a[e] += bf2f(v[e]);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; e++) s[ry][cx][e] = a[e];
  __syncthreads();
  // 512 threads close 512 columns: thread t sums the 8 row lanes of column t
  {
    const int cc = threadIdx.x >> 3, e = threadIdx.x & 7;
    const int col = (blockIdx.x * 64 + cc) * 8 + e;
    if (col < C) {
      float v = 0.f;
#pragma unroll
      for (int y = 0; y < 8; y++) v += s[y][cc][e];
      const int co = colmap ? colmap[col] : col;
      if (co >= 0 && v != 0.f) atomicAdd(out + co, v);
    }
  }
}

__global__ __launch_bounds__(256) void sumsq_k(const float* g, long long n, float* out) {
  __shared__ float sacc[WAVES];
  float s = 0.f;
  for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (long long)gridDim.x * blockDim.x * 4) {
    if (i + 3 < n) { f32x4 v = *(const f32x4*)(g + i); s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]; }
    else for (long long j = i; j < n; j++) s += g[j] * g[j];
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sacc[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) { float a = 0.f; for (int i = 0; i < WAVES; i++) a += sacc[i]; atomicAdd(out, a); }
}

// clip_grad_norm_(max_norm) + Adam, fused: coef = min(1, max_norm / (norm + 1e-6))   train_toy.py:55-57
__global__ __launch_bounds__(256) void adam_k(tfx_adam_args p, float step_size, float inv_sqrt_bc2) {
  // 4 parameters per thread (16-B accesses); bias corrections are computed once on the host side of the launch
  const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= p.n) return;
  float coef = p.grad_scale;
  if (p.max_norm > 0.f) {
    float norm = sqrtf(p.sumsq[0]) * p.grad_scale;
    coef *= fminf(1.f, p.max_norm / (norm + 1e-6f));
  }
  if (i + 4 <= p.n) {
    const f32x4 g4 = *(const f32x4*)(p.g + i), w4 = *(const f32x4*)(p.p + i);
    f32x4 m4 = *(const f32x4*)(p.m + i), v4 = *(const f32x4*)(p.v + i), o4;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      float g = g4[e] * coef;
      if (p.weight_decay != 0.f) g += p.weight_decay * w4[e];
      m4[e] = p.beta1 * m4[e] + (1.f - p.beta1) * g;
      v4[e] = p.beta2 * v4[e] + (1.f - p.beta2) * g * g;
      o4[e] = w4[e] - step_size * m4[e] / (sqrtf(v4[e]) * inv_sqrt_bc2 + p.eps);
    }
    *(f32x4*)(p.m + i) = m4; *(f32x4*)(p.v + i) = v4; *(f32x4*)(p.p + i) = o4;
  } else {
    for (long long j = i; j < p.n; j++) {
      float g = p.g[j] * coef;
      const float w = p.p[j];
      if (p.weight_decay != 0.f) g += p.weight_decay * w;
      const float m = p.beta1 * p.m[j] + (1.f - p.beta1) * g;
      const float v = p.beta2 * p.v[j] + (1.f - p.beta2) * g * g;
      p.m[j] = m; p.v[j] = v;
      p.p[j] = w - step_size * m / (sqrtf(v) * inv_sqrt_bc2 + p.eps);
    }
  }
}

__global__ __launch_bounds__(256) void output_to_flow_k(float* pred, const float* x, const float* eps, const int* row_inst, const float* inst_time,
                                                        int R, int dl, float clean_eps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R * dl) return;
  const int r = i / dl;
  const float t = inst_time[row_inst[r]];
  const float noised = eps ? x[i] * t + eps[i] * (1.f - t) : x[i];
  pred[i] = (pred[i] - noised) / fmaxf(1.f - t, clean_eps);
}

__global__ __launch_bounds__(256) void ema_k(float* ema, const float* online, long long n, float decay) {
  const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 4 <= n) {
    f32x4 e = *(const f32x4*)(ema + i); const f32x4 o = *(const f32x4*)(online + i);
#pragma unroll
    for (int k = 0; k < 4; k++) e[k] = o[k] + decay * (e[k] - o[k]);
    *(f32x4*)(ema + i) = e;
  } else {
    for (long long j = i; j < n; j++) ema[j] = online[j] + decay * (ema[j] - online[j]);
  }
}

}  // namespace tfx

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
using namespace tfx;
#define ST(s) ((hipStream_t)(s))
#define RET() return (int)hipGetLastError()

// blocks of `threads` threads that are RESIDENT at once on the chip for kernel `fn` (registers / LDS): segment kernels walk their items with a
// grid stride from exactly that many blocks, so that every wave gets the same share of the (length-sorted) segments instead of the
// hardware back-filling whole blocks whose waves finish at different times
template <typename F> static int resident_blocks(F fn, int threads, size_t dyn) {
  // (the occupancy query costs microseconds of host time: asked once per kernel and LDS size)
  static std::map<std::pair<const void*, size_t>, int> cache;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  const auto key = std::make_pair((const void*)fn, dyn);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, dyn) != hipSuccess || per_cu < 1) per_cu = 1;
  static int cus = 0;
  if (!cus) { hipDeviceProp_t pr; int dev = 0; (void)hipGetDevice(&dev); cus = (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256; }
  return cache[key] = per_cu * cus;
}
// grid of a segment kernel (8 waves = 8 segments per 512-thread block): every block resident at once, each wave walking its share of the
// length-sorted segments with a grid stride (packing.token_segments balance=True)
template <typename F> static int seg_grid(F fn, int n_seg) {
  int g = (n_seg + 7) / 8;
  const int cap = resident_blocks(fn, 512, 0);
  if (g > cap) g = cap;
  return g < 1 ? 1 : g;
}
template <int NC, int AHEAD, bool W16, int NJ> static int launch_pull_dma(const tfx_attnres_pull_args& a, const tfx_adaln_post_args& b, int has_post, size_t dyn, hipStream_t s) {
  auto fn = attnres_pull_dma_k<NC, AHEAD, W16, NJ>;
  static uint32_t attr = 0;
  ensure_smem_attr((const void*)fn, 156 * 1024, attr);
  const int items = a.n_seg > 0 ? a.n_seg : a.T;
  int grid = (items + 7) / 8;
  const int cap = resident_blocks(fn, 512, dyn);
  if (grid > cap) grid = cap;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(fn, dim3(grid), dim3(512), dyn, s, a, b, has_post);
  return (int)hipGetLastError();
}
template <int NC, int NJ> static int pull_dma(const tfx_attnres_pull_args& a, const tfx_adaln_post_args& b, int has_post, hipStream_t s) {
  // LDS: [w rows: fp32, or bf16 when fp32 would leave the rings fewer than 6 slots][8 rings of (AHEAD + 1) row slots]; 160 KiB per CU, ~4 KiB static
  const size_t budget = 154 * 1024, slot8 = (size_t)8 * NC * 1024;
  const size_t w32 = (((size_t)a.n_src * a.d * 4) + 1023) & ~(size_t)1023, w16 = (((size_t)a.n_src * a.d * 2) + 1023) & ~(size_t)1023;
  const bool w_bf16 = w32 + 6 * slot8 > budget;
  const size_t fixed = w_bf16 ? w16 : w32;
  if (fixed + 3 * slot8 > budget) return -4;
  const size_t slots = (budget - fixed) / slot8;
#define TFX_PULL_GO(AH) do { const size_t dyn = fixed + ((AH) + 1) * slot8; \
    return w_bf16 ? launch_pull_dma<NC, AH, true, NJ>(a, b, has_post, dyn, s) : launch_pull_dma<NC, AH, false, NJ>(a, b, has_post, dyn, s); } while (0)
  if (slots >= 14 && NC == 1) TFX_PULL_GO(13);
  if (slots >= 10) TFX_PULL_GO(9);
  if (slots >= 6) TFX_PULL_GO(5);
  if (slots >= 4) TFX_PULL_GO(3);
  TFX_PULL_GO(2);
#undef TFX_PULL_GO
}

// register-resident MFMA loop (tfx.h tfx_mfma_peak_probe; tools/mfma_peak.hip is the stand-alone form with the mixed MFMA + VALU variants)
__global__ __launch_bounds__(256) void mfma_peak_k(const bf16x8* ops, float* out, int iters) {
  bf16x8 a = ops[threadIdx.x & 63], b = ops[64 + (threadIdx.x & 63)];
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  for (int i = 0; i < iters; i++) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b));
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c2) : "v"(a), "v"(b));
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c3) : "v"(a), "v"(b));
  }
  asm volatile("s_nop 15\n\ts_nop 15" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
  float sm = 0.f;
  for (int i = 0; i < 16; i++) sm += c0[i] + c1[i] + c2[i] + c3[i];
  if (sm == 1.2345f) out[1] = sm;
}

extern "C" {

int tfx_adaln_pre_fwd(const tfx_adaln_pre_args* a, void* s) { DISPATCH_NC(a->d, hipLaunchKernelGGL(adaln_pre_fwd_k<NC>, dim3(grid_tokens(a->T)), dim3(256), 0, ST(s), *a)); RET(); }
// d > 1024 (NC = 4, outside every BASELINE width): the segment forms of the input side keep a 32-column register tile per lane plus its partials and
// spilled 52 / 233 registers (tools/check_spills.sh, flagged three rounds running) - such widths take the per-token forms, which do not spill (the
// table gradients then arrive by atomics on the zeroed buffer instead of one plain store per segment: same sums)
int tfx_adaln_pre_bwd(const tfx_adaln_pre_args* a, void* s) {
  if (a->seg_start && a->n_seg > 0 && a->d <= 1024) { DISPATCH_NC_LE1024(a->d, hipLaunchKernelGGL(adaln_pre_bwd_seg_k<NC>, dim3(seg_grid(adaln_pre_bwd_seg_k<NC>, a->n_seg)), dim3(512), 0, ST(s), *a)); }
  else { DISPATCH_NC(a->d, hipLaunchKernelGGL(adaln_pre_bwd_k<NC>, dim3(grid_capped(a->T)), dim3(256), 0, ST(s), *a)); }
  RET();
}
int tfx_adaln_post_fwd(const tfx_adaln_post_args* a, void* s) { DISPATCH_NC(a->d, hipLaunchKernelGGL(adaln_post_fwd_k<NC>, dim3(grid_tokens(a->T)), dim3(256), 0, ST(s), *a)); RET(); }
int tfx_adaln_post_bwd(const tfx_adaln_post_args* a, void* s) {
  if (a->seg_start && a->n_seg > 0) { DISPATCH_NC(a->d, hipLaunchKernelGGL(adaln_post_bwd_seg_k<NC>, dim3(seg_grid(adaln_post_bwd_seg_k<NC>, a->n_seg)), dim3(512), 0, ST(s), *a)); }
  else { DISPATCH_NC(a->d, hipLaunchKernelGGL(adaln_post_bwd_k<NC>, dim3(grid_capped(a->T)), dim3(256), 0, ST(s), *a)); }
  RET();
}
int tfx_adaln_post_pre_fwd(const tfx_adaln_post_args* a, const tfx_adaln_pre_args* b, void* s) {
  if (a->T != b->T || a->d != b->d || a->tok_inst != b->tok_inst || (const void*)a->out != (const void*)b->x) return -2;
  DISPATCH_NC(a->d, hipLaunchKernelGGL(adaln_post_pre_fwd_k<NC>, dim3(grid_tokens(a->T)), dim3(256), 0, ST(s), *a, *b)); RET();
}
int tfx_layer_end_fwd(const tfx_adaln_post_args* a, const tfx_attnres_args* r, const tfx_adaln_pre_args* b, void* s) {
  if (a->T != r->T || a->d != r->d || r->L < 1 || r->L > 64) return -2;
  if ((const void*)a->out != (const void*)(r->hiddens + (size_t)(r->L - 1) * r->stride_h)) return -3;
  if (b && (b->T != a->T || b->d != a->d || b->tok_inst != a->tok_inst || (const void*)b->x != (const void*)r->out)) return -4;
  tfx_adaln_pre_args none = {};
  DISPATCH_NC(a->d, hipLaunchKernelGGL(layer_end_fwd_k<NC>, dim3(grid_tokens(a->T)), dim3(256), 0, ST(s), *a, *r, b ? *b : none, b ? 1 : 0)); RET();
}
int tfx_rmsnorm_fwd(const tfx_rmsnorm_args* a, void* s) { DISPATCH_NC(a->d, hipLaunchKernelGGL(rmsnorm_fwd_k<NC>, dim3(grid_tokens(a->T)), dim3(256), 0, ST(s), *a)); RET(); }
int tfx_rmsnorm_bwd(const tfx_rmsnorm_args* a, void* s) { DISPATCH_NC(a->d, hipLaunchKernelGGL(rmsnorm_bwd_k<NC>, dim3(grid_capped(a->T)), dim3(256), 0, ST(s), *a)); RET(); }
int tfx_attnres_fwd(const tfx_attnres_args* a, void* s) { if (a->L > 64) return -2; DISPATCH_NC(a->d, hipLaunchKernelGGL(attnres_fwd_k<NC>, dim3(grid_tokens(a->T)), dim3(256), 0, ST(s), *a)); RET(); }
int tfx_attnres_bwd(const tfx_attnres_args* a, void* s) {
  if (a->L > 64) return -2;
  DISPATCH_NC(a->d, hipLaunchKernelGGL(attnres_bwd_k<NC>, dim3(grid_capped(a->T)), dim3(256), 0, ST(s), *a)); RET();
}
int tfx_adaln_pre_post_bwd(const tfx_adaln_pre_args* a, const tfx_adaln_post_args* b, void* s) {
  if (!a || !b || a->T != b->T || a->d != b->d || a->tok_inst != b->tok_inst || (const void*)a->dx != (const void*)b->g) return -2;
  if (!a->seg_start || a->n_seg <= 0 || a->d > 1024) {   // no segments (or a width whose fused segment form would spill): the two launches
    int rc = tfx_adaln_pre_bwd(a, s);
    return rc ? rc : tfx_adaln_post_bwd(b, s);
  }
  if (a->dx_add || b->dbias) return -3;
  DISPATCH_NC_LE1024(a->d, hipLaunchKernelGGL(adaln_pre_post_bwd_seg_k<NC>, dim3(seg_grid(adaln_pre_post_bwd_seg_k<NC>, a->n_seg)), dim3(512), 0, ST(s), *a, *b)); RET();
}
int tfx_attnres_prep(const tfx_attnres_src* src, int32_t n, int32_t d, void* s) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(attnres_prep_k, dim3((d + 255) / 256, n), dim3(256), 0, ST(s), src, d); RET();
}
int tfx_attnres_finish(const tfx_attnres_src* src, int32_t n, int32_t d, void* s) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(attnres_finish_k, dim3((d + 255) / 256, n), dim3(256), 0, ST(s), src, d); RET();
}
int tfx_attnres_pull_bwd(const tfx_attnres_pull_args* a, const tfx_adaln_post_args* b, void* s) {
  if (!a || a->n_src < 1 || a->n_src > PULL_MAX_SRC || (a->d & 7) || a->d > 2048) return -2;
  if (b && (b->T != a->T || b->d != a->d || (const void*)b->g != (const void*)a->dh)) return -3;
  tfx_adaln_post_args none = {};
  const tfx_adaln_post_args& bb = b ? *b : none;
  const int hp = b ? 1 : 0;
  const size_t acc = (size_t)a->n_src * a->d * sizeof(float);
  // few narrow sources (depth <= 8 at d <= 512): the register form (1.79 ms / step at depth 8 against 1.97 for the ring form, whose per-row issue
  // code is the larger part of its instruction stream there); everything else: the LDS-DMA ring form (depth 24 / d 1024: 18.5 ms / step against
  // 32.3 for the push form + the separate wrapper launch).
  const bool small = a->d <= 512 && a->n_src <= 8 && !a->k1;
  if (!small) {
    if (a->d > 1024) return -6;
    // d w partials exported (a->k1) for one weight-gradient GEMM behind this launch
    if (!a->k1 || a->ld_k1 < a->n_src) return -5;
    return a->d <= 512 ? pull_dma<1, 0>(*a, bb, hp, ST(s)) : pull_dma<2, 0>(*a, bb, hp, ST(s));
  }
  if (small) {
    // few sources, narrow rows: d w partials in registers, every row of the token in flight at once
    auto fn = (b && b->dbias) ? attnres_pull_reg_k<1, 8, true> : attnres_pull_reg_k<1, 8, false>;
    const int items = a->n_seg > 0 ? a->n_seg : a->T;
    int grid = (items + 7) / 8; const int cap = resident_blocks(fn, 512, acc);          // (512 / 1024 blocks measured 180 / 197 against 173 us, round 3)
    if (grid > cap) grid = cap; if (grid < 1) grid = 1;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(512), acc, ST(s), *a, bb, hp); RET();
  }
  return -6;                                               // no other form for this size
}
int tfx_embed_fwd(const tfx_embed_args* a, void* s) { DISPATCH_NC(a->d, hipLaunchKernelGGL(embed_fwd_k<NC>, dim3(grid_tokens(a->T)), dim3(256), 0, ST(s), *a)); RET(); }
int tfx_embed_bwd(const tfx_embed_args* a, void* s) { DISPATCH_NC(a->d, hipLaunchKernelGGL(embed_bwd_k<NC>, dim3(grid_tokens(a->T)), dim3(256), 0, ST(s), *a)); RET(); }

int tfx_qk_norm_rope_fwd(const tfx_qk_norm_rope_args* a, void* s) {
  long long nthreads = (long long)a->T * (a->cache ? 3 : 2) * a->H * 8;
  if (nthreads >= (1ll << 31)) return -3;
  hipLaunchKernelGGL(qk_norm_rope_fwd_k, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, ST(s), *a); RET();
}
int tfx_qk_norm_rope_bwd(const tfx_qk_norm_rope_args* a, void* s) {
  long long nthreads = (long long)a->T * 2 * a->H * 8;
  if (nthreads >= (1ll << 31)) return -3;
  // 1024-thread blocks, at most one per CU: every block ends in 128 global atomics onto the SAME 128 gamma-gradient addresses, and same-address
  // atomics retire at about 15 ns apiece - 1024 blocks of 256 threads spent 15 us of a 94 us launch there, 4096 blocks 61 us of 150
  // (gpurun_out/r03qkb*_call.log).
  long long g = (nthreads + 1023) / 1024;
  const long long cap = 256;
  if (g > cap) g = cap;
  hipLaunchKernelGGL(qk_norm_rope_bwd_k, dim3((unsigned)g), dim3(1024), 0, ST(s), *a); RET();
}
int tfx_noise_mix(const tfx_noise_mix_args* a, void* s) {
  long long n = (long long)a->R * a->ld_xt; if (n == 0) return 0;
  hipLaunchKernelGGL(noise_mix_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ST(s), *a); RET();
}
int tfx_fourier(const tfx_fourier_args* a, void* s) { if (a->I == 0) return 0; hipLaunchKernelGGL(fourier_k, dim3(a->I), dim3(256), 0, ST(s), *a); RET(); }
int tfx_ce_fwd_bwd(const tfx_ce_args* a, void* s) {
  int g = grid_tokens(a->T); if (g > 1024) g = 1024;
  hipLaunchKernelGGL(ce_k, dim3(g), dim3(256), 0, ST(s), *a); RET();
}
int tfx_mse_fwd_bwd(const tfx_mse_args* a, void* s) {
  long long n = (long long)a->R * a->ld_d; if (n == 0) return 0;
  long long g = (n + 255) / 256; if (g > 2048) g = 2048;
  hipLaunchKernelGGL(mse_k, dim3((unsigned)g), dim3(256), 0, ST(s), *a); RET();
}
int tfx_cast_rows(const tfx_cast_args* a, void* s) {
  long long n = (long long)a->Rd * a->ld_dst; if (n == 0) return 0;
  hipLaunchKernelGGL(cast_rows_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ST(s), *a); RET();
}
int tfx_cast_rows_t(const tfx_cast_args* a, void* s) {
  hipLaunchKernelGGL(cast_rows_t_k, dim3((a->ld_dst + 31) / 32, (a->Rd + 31) / 32), dim3(256), 0, ST(s), *a); RET();
}
int tfx_cast_batch(const tfx_cast_job* jobs_dev, int32_t n_jobs, int32_t n_blocks, void* s) {
  if (n_jobs <= 0 || n_blocks <= 0) return 0;
  hipLaunchKernelGGL(cast_batch_k, dim3(n_blocks), dim3(256), 0, ST(s), jobs_dev, n_jobs); RET();
}
int tfx_onehot_bf16(const int32_t* ids, const int32_t* tok_inst, tfx_bf16* out, int32_t T, int32_t ld, void* s) {
  if (T == 0) return 0; if (ld % 8) return -1;
  long long n = (long long)T * ld / 8;
  hipLaunchKernelGGL(onehot_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ST(s), ids, tok_inst, out, T, ld); RET();
}
int tfx_scatter_rows_bf16(const tfx_bf16* src, int32_t ld_src, int32_t cols, tfx_bf16* dst, int32_t ld_dst, const int32_t* rowmap, int32_t R, void* s) {
  if (R == 0) return 0; if (cols % 8 || ld_src % 8 || ld_dst % 8) return -1;
  long long n = (long long)R * (cols / 8);
  hipLaunchKernelGGL(scatter_rows_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ST(s), src, ld_src, cols, dst, ld_dst, rowmap, R); RET();
}
int tfx_gather_f32(const float* src, const int32_t* map, float* dst, int32_t n, void* s) {
  if (n == 0) return 0; hipLaunchKernelGGL(gather_f32_k, dim3((n + 255) / 256), dim3(256), 0, ST(s), src, map, dst, n); RET();
}
int tfx_f32_to_bf16(const float* src, tfx_bf16* dst, int64_t n, void* s) {
  if (n == 0) return 0; hipLaunchKernelGGL(f32_to_bf16_k, dim3((unsigned)((n / 4 + 256) / 256)), dim3(256), 0, ST(s), src, dst, (long long)n); RET();
}
int tfx_silu_bwd(const tfx_bf16* dy, const tfx_bf16* pre, tfx_bf16* dx, int64_t n, void* s) {
  if (n == 0) return 0; hipLaunchKernelGGL(silu_bwd_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ST(s), dy, pre, dx, (long long)n); RET();
}
int tfx_add_bf16(const tfx_bf16* a, const tfx_bf16* b, tfx_bf16* o, int64_t n, void* s) {
  if (n == 0) return 0; hipLaunchKernelGGL(add_bf16_k, dim3((unsigned)((n / 8 + 256) / 256)), dim3(256), 0, ST(s), a, b, o, (long long)n); RET();
}
int tfx_scale_bf16_dev(tfx_bf16* x, int64_t n, const float* scale, void* s) {
  if (n == 0) return 0;
  if (!x || !scale) return -1;
  hipLaunchKernelGGL(scale_bf16_dev_k, dim3((unsigned)((n / 8 + 256) / 256)), dim3(256), 0, ST(s), x, (long long)n, scale); RET();
}
static inline int colsum_rows_per_block(int R) { int rpb = (R + 63) / 64; return rpb < 4 ? 4 : rpb; }
int tfx_colsum_bf16(const tfx_bf16* src, int32_t ld, int32_t R, int32_t C, const int32_t* colmap, const int32_t* rowmap, float* out, void* s) {
  if (R == 0 || C == 0) return 0;
  if (C % 8 == 0 && ld % 8 == 0) {
    const int gx = (C + 511) / 512;
    int gy = 768 / gx; if (gy < 1) gy = 1;          // ~3 blocks of 512 threads per CU; one closing atomic per thread
    int rpb8 = (R + gy - 1) / gy; if (rpb8 < 64) rpb8 = 64;
    hipLaunchKernelGGL(colsum8_k, dim3(gx, (R + rpb8 - 1) / rpb8), dim3(512), 0, ST(s), src, ld, R, C, colmap, rowmap, out, rpb8); RET();
  }
  int rpb = colsum_rows_per_block(R);
  hipLaunchKernelGGL(colsum_k<bf16>, dim3((C + 63) / 64, (R + rpb - 1) / rpb), dim3(256), 0, ST(s), src, ld, R, C, colmap, rowmap, out, rpb); RET();
}
int tfx_colsum_f32(const float* src, int32_t ld, int32_t R, int32_t C, float* out, void* s) {
  if (R == 0 || C == 0) return 0;
  int rpb = colsum_rows_per_block(R);
  hipLaunchKernelGGL(colsum_k<float>, dim3((C + 63) / 64, (R + rpb - 1) / rpb), dim3(256), 0, ST(s), src, ld, R, C, (const int*)nullptr, (const int*)nullptr, out, rpb); RET();
}
int tfx_sumsq(const float* g, int64_t n, float* out, void* s) {
  if (n == 0) return 0;
  long long blocks = (n / 4 + 255) / 256; if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(sumsq_k, dim3((unsigned)blocks), dim3(256), 0, ST(s), g, (long long)n, out); RET();
}
int tfx_adam_step(const tfx_adam_args* a, void* s) {
  if (a->n == 0) return 0;
  if (((uintptr_t)a->p | (uintptr_t)a->g | (uintptr_t)a->m | (uintptr_t)a->v) & 15) return -1;
  const double bc1 = 1.0 - pow((double)a->beta1, (double)a->step), bc2 = 1.0 - pow((double)a->beta2, (double)a->step);
  const long long nthr = (a->n + 3) / 4;
  hipLaunchKernelGGL(adam_k, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, ST(s), *a, (float)(a->lr / bc1), (float)(1.0 / sqrt(bc2))); RET();
}
int tfx_output_to_flow(float* pred, const float* x, const float* eps, const int32_t* row_inst, const float* inst_time,
                       int32_t R, int32_t dl, float clean_eps, void* s) {
  if (R <= 0 || dl <= 0) return 0;
  if ((long long)R * dl >= (1ll << 31)) return -1;
  hipLaunchKernelGGL(output_to_flow_k, dim3((unsigned)(((long long)R * dl + 255) / 256)), dim3(256), 0, ST(s), pred, x, eps, row_inst, inst_time, R, dl, clean_eps); RET();
}
int tfx_ema_update(float* ema, const float* online, int64_t n, float decay, void* s) {
  if (n <= 0) return 0;
  if (((uintptr_t)ema | (uintptr_t)online) & 15) return -1;
  hipLaunchKernelGGL(ema_k, dim3((unsigned)(((n + 3) / 4 + 255) / 256)), dim3(256), 0, ST(s), ema, online, (long long)n, decay); RET();
}
int tfx_gemm_nt(const tfx_gemm_nt_args* a, void* s) { return gemm_nt(*a, ST(s)); }
int tfx_gemm_tn(const tfx_gemm_tn_args* a, void* s) { return gemm_tn(*a, ST(s)); }
int tfx_gemm_nt_plan(const tfx_gemm_nt_args* a, int32_t* kind, int32_t* grid) { return gemm_nt_plan(*a, kind, grid); }
int tfx_gemm_tn_plan(const tfx_gemm_tn_args* a, int32_t* kind, int32_t* tiles, int32_t* splits, int32_t* grid) { return gemm_tn_plan(*a, kind, tiles, splits, grid); }
int tfx_attn_fwd(const tfx_attn_args* a, void* s) { return attn_fwd(*a, ST(s)); }
int tfx_attn_bwd(const tfx_attn_args* a, void* s) { return attn_bwd(*a, ST(s)); }
int tfx_mfma_peak_probe(const void* ops, float* out, int32_t iters, int32_t blocks, void* s) {
  if (!ops || !out || iters <= 0 || blocks <= 0) return -1;
  hipLaunchKernelGGL(mfma_peak_k, dim3(blocks), dim3(256), 0, ST(s), (const bf16x8*)ops, out, iters); RET();
}
const char* tfx_version(void) { return "tfx-hip gfx950 r1"; }

}  // extern "C"
