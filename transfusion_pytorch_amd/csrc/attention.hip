// Flash attention for the Transfusion score pipeline (reference T:998-1027), dim_head = 64, gfx950 MFMA.
//
//   S = q~ k~^T ; S <- cap*tanh(S/cap) ; key j visible to query i iff j < kv_end[i] (prefix-extension
//   form of naive_attn_mask T:452-470) ; P = softmax(S) ; O = P V ; out = O * sigmoid(gate)
//
// All matmuls are v_mfma_f32_32x32x16_bf16.  Layout trick: scores are computed TRANSPOSED
// (S^T = K.Q^T, lane = query row) so every per-row softmax quantity is per-lane, and P^T feeds the
// second MFMA straight from registers; the operand that is contracted over its row index (V, or K / Q /
// dO in the backward) is gathered from row-major LDS tiles with ds_read_b64_tr_b16 (lds_tr8).
// MFMA contraction "slots" e=0..7 of lane-half hi map to rows 16*tt + 8*(e>>2) + 4*hi + (e&3) of a
// 32-row block: both operands use the same map, so the sum is unchanged.
#include "tfx_kernels.h"
#include "attn_asm_clobbers.inc"

namespace tfx {

constexpr int DH = 64;
constexpr int LDT = 72;            // LDS row stride (elements): 64 + 8 pad -> conflict-free ds_read_b128

// cooperative [64][64] bf16 tile: global (row stride ld, rows clamped to [0, nrows)) -> registers -> LDS
struct TileRegs { u32x4 r[2]; };
TFX_DEV void tile_gload(TileRegs& tr, const bf16* base, int ld, int row0, int nrows) {
#pragma unroll
  for (int i = 0; i < 2; i++) {
    int c = threadIdx.x + 256 * i;
    int row = min(row0 + (c >> 3), nrows - 1);
    tr.r[i] = *(const u32x4*)(base + (size_t)row * ld + (c & 7) * 8);
  }
}
TFX_DEV void tile_sstore(const TileRegs& tr, bf16* lds) {
#pragma unroll
  for (int i = 0; i < 2; i++) {
    int c = threadIdx.x + 256 * i;
    *(u32x4*)(lds + (c >> 3) * LDT + (c & 7) * 8) = tr.r[i];
  }
}
// row-operand fragment (lane: row r0 + (l&31), 8 contiguous columns 16*ks + 8*hi) from an LDS tile
TFX_DEV bf16x8 lds_rowfrag(const bf16* lds, int r0, int ks) {
  const int l = threadIdx.x & 63;
  return *(const bf16x8*)(lds + (r0 + (l & 31)) * LDT + 16 * ks + 8 * (l >> 5));
}
// same fragment straight from global memory (row clamped)
// (lane: callers inside a per-block TILE LOOP hand in an opaque copy of the thread index, so that hipcc does not hoist the lane-derived addresses out of that loop -
//  where they are spilled next to an asm statement's fixed registers; -1 = threadIdx.x.  The same parameter on tile_dma, dma_rowfrag, dma_tr8, the block stores.)
TFX_DEV bf16x8 g_rowfrag(const bf16* base, int ld, int row, int nrows, int ks, int lane = -1) {
  const int l = (lane >= 0 ? lane : (int)threadIdx.x) & 63;
  row = min(row, nrows - 1);
  return *(const bf16x8*)(base + (size_t)row * ld + 16 * ks + 8 * (l >> 5));
}
// ---- LDS-DMA tiles: unpadded [64][64] bf16 (128-byte rows), 16-byte chunk index XOR swz_f(row).  swz_f is a bijection of
// (row >> 1) & 7 whose bit 2 is bit 1 of the row: any 16 consecutive rows of a ds_read_b128 lane group land on distinct
// banks (row parity picks the 128-byte half of the bank window, swz_f the chunk), and the 4 rows x 64 bytes of a
// ds_read_b64_tr_b16 group do too (rows r, r+2 share a half but differ in chunk bit 2).  The DMA writes lane-linear
// 1 KiB pieces (8 rows), so the same XOR is applied to the SOURCE chunk each lane fetches.
TFX_DEV int swz_f(int row) { const int v = (row >> 1) & 7; return ((v & 1) << 2) | (v >> 1); }
TFX_DEV void tile_dma(const bf16* base, int ld, int row0, int nrows, bf16* lds_tile, int t = -1) {   // 256 threads: wave w brings rows [16w, 16w+16)
  const int tx = t >= 0 ? t : (int)threadIdx.x, l = tx & 63, w = tx >> 6;
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int r = w * 16 + j * 8 + (l >> 3);
    const int c = (l & 7) ^ swz_f(r);
    glds16_asm(base + (size_t)min(row0 + r, nrows - 1) * ld + c * 8, lds_tile + (w * 16 + j * 8) * 64);
  }
}
TFX_DEV bf16x8 dma_rowfrag(const bf16* lds, int r0, int ks, int lane = -1) {
  const int l = (lane >= 0 ? lane : (int)threadIdx.x) & 63, r = r0 + (l & 31);
  return *(const bf16x8*)(lds + r * 64 + (((2 * ks + (l >> 5)) ^ swz_f(r)) << 3));
}
TFX_DEV bf16x8 dma_tr8(const bf16* tile, int rowA, int rowB, int c0, int lane = -1) {
  const int l = (lane >= 0 ? lane : (int)threadIdx.x) & 63, q = l & 15;
  const int col = c0 + 16 * ((l >> 4) & 1) + 4 * (q & 3);
  const int ra = rowA + (q >> 2), rb = rowB + (q >> 2);
  s16x4 lo = lds_tr4(tile + ra * 64 + (((col >> 3) ^ swz_f(ra)) << 3) + (col & 7));
  s16x4 hi = lds_tr4(tile + rb * 64 + (((col >> 3) ^ swz_f(rb)) << 3) + (col & 7));
  const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
  const u32x4 v = {l2[0], l2[1], h2[0], h2[1]};
  return __builtin_bit_cast(bf16x8, v);
}
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
typedef float f32x2 __attribute__((ext_vector_type(2)));
// Soft-cap in the log2 domain: s2 = cap*log2e*tanh(s/cap).  Scores are q~.k~ with |q~|,|k~| bounded by the QK-RMSNorm
// gains, so x = s/cap is small: the odd Taylor polynomial to x^9 is exact to 1.3e-6 for |x| <= 0.45.  It is evaluated
// directly in s (the 1/cap powers and log2e are folded into the coefficients) on PAIRS of scores so that every step
// is a packed v_pk_mul/v_pk_fma (a wave64 VALU op costs 4 cycles; the packed forms halve that per score).  If ANY lane
// of the wave exceeds the bound the whole wave takes the exact exp2/rcp form (wave-uniform branch, no divergence).
// Because |s2| <= cap*log2e (72 for cap 50), exp2(s2) can never overflow or vanish in fp32/bf16: the softmax uses the
// FIXED reference 0 instead of a running maximum, which removes the max/rescale work of online softmax entirely.
struct SoftCap { float cap, g2; int mode; float p1, p3, p5, d1, d3, d5; };
struct SoftCapLegacy { float k1, k3, k5, k7, k9, icap, cap2, smax, smid, slo; };
// `plan` (tfx.h tfx_qk_norm_rope_args.sc_plan): the layer's polynomial, chosen from the QK-RMSNorm bound on the scores - modes 0 / 1 need no look at
// the scores (no running |s| maximum, no wave vote, no degree branch); NULL or mode 2 = the data-dependent choice of softcap16's second half
TFX_DEV SoftCap make_softcap(float cap, const float* plan = nullptr) {
  SoftCap c;
  c.cap = cap; c.mode = 2; c.p1 = c.p3 = c.p5 = c.d1 = c.d3 = c.d5 = 0.f;
  if (plan) {                                                  // kernel argument: uniform; scalar loads
    c.mode = (int)plan[0]; c.p1 = plan[1]; c.p3 = plan[2]; c.p5 = plan[3]; c.d1 = plan[4]; c.d3 = plan[5]; c.d5 = plan[6];
  }
  const float cap2 = cap * LOG2E;
  c.g2 = 1.f / (cap2 * cap2);
  return c;
}
// The constants of the data-dependent form are derived where that form runs, behind an opaque copy of `cap`: VALU instructions take one scalar
// operand, so hipcc keeps loop-invariant coefficients in VECTOR registers for the whole kernel - ten of them pushed the pipelined forward from
// 168 to 176 allocated registers (two waves per SIMD instead of three) once the plan's coefficients joined them.
TFX_DEV SoftCapLegacy make_softcap_legacy(float cap) {
  asm volatile("" : "+s"(cap));
  SoftCapLegacy c;
  const float ic = 1.f / cap, i2 = ic * ic;
  c.k1 = LOG2E; c.k3 = -LOG2E * i2 * 0.33333334f; c.k5 = LOG2E * i2 * i2 * 0.13333334f;
  c.k7 = -LOG2E * i2 * i2 * i2 * 0.053968254f; c.k9 = LOG2E * i2 * i2 * i2 * i2 * 0.021869488f;
  c.icap = ic; c.cap2 = cap * LOG2E; c.smax = 0.45f * cap; c.smid = 0.28f * cap; c.slo = 0.12f * cap;
  return c;
}
TFX_DEV float tanh_exact(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * (2.f * LOG2E))); }
TFX_DEV f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
TFX_DEV f32x2 bc2(float v) { f32x2 r = {v, v}; return r; }
// in place: s <- cap*log2e*tanh(s/cap) for one 32x32 accumulator block
TFX_DEV void softcap16(f32x16& s, const SoftCap& cp) {
  if (cp.mode == 0) {                                           // (scalar branch) the layer's cubic
#pragma unroll
    for (int r = 0; r < 16; r++) s[r] = s[r] * fmaf(s[r] * s[r], cp.p3, cp.p1);
    return;
  }
  if (cp.mode == 1) {                                          // the layer's quintic
#pragma unroll
    for (int r = 0; r < 16; r++) { const float u = s[r] * s[r]; s[r] = s[r] * fmaf(u, fmaf(u, cp.p5, cp.p3), cp.p1); }
    return;
  }
  const SoftCapLegacy c = make_softcap_legacy(cp.cap);
  float amax = 0.f;
#pragma unroll
  for (int r = 0; r < 16; r++) amax = fmaxf(amax, fabsf(s[r]));
  if (__any(amax > c.smax)) {
#pragma unroll
    for (int r = 0; r < 16; r++) s[r] = c.cap2 * tanh_exact(s[r] * c.icap);
    return;
  }
  // the polynomial degree follows the wave's largest |s/cap|: x^3 below 0.12 (error < 4e-6), x^5 below 0.28 (< 8e-6), else x^9.
  // SCALAR fma forms: a wave64 v_fma_f32 issues in 2.7 clocks, v_pk_fma_f32 / v_pk_mul_f32 in 6.5 (tools/valu_probe.hip) - the packed forms
  // are slower than the two scalar instructions they replace (the file is built with -fno-slp-vectorize so hipcc does not re-pack them)
  if (!__any(amax > c.slo)) {
#pragma unroll
    for (int r = 0; r < 16; r++) s[r] = s[r] * fmaf(s[r] * s[r], c.k3, c.k1);
  } else if (!__any(amax > c.smid)) {
#pragma unroll
    for (int r = 0; r < 16; r++) { const float u = s[r] * s[r]; s[r] = s[r] * fmaf(u, fmaf(u, c.k5, c.k3), c.k1); }
  } else {
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const float u = s[r] * s[r];
      s[r] = s[r] * fmaf(u, fmaf(u, fmaf(u, fmaf(u, c.k9, c.k7), c.k5), c.k3), c.k1);
    }
  }
}
TFX_DEV int wave_min_i(int v) {                  // wave-uniform result, returned in an SGPR so that tests on it are scalar branches
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
  return __builtin_amdgcn_readfirstlane(v);
}
TFX_DEV int wave_max_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
  return __builtin_amdgcn_readfirstlane(v);
}
// Number of keys a 128-row query tile has to walk.  Training layouts: kv_end is non-decreasing in the query index (prefix-extension mask), the
// last row of the tile has the largest.  Decode steps against a KV cache: a sample's new rows need not be ordered by visible length (a text row
// next to a modality block in the mixed steps of the continuous schedule) - take the true maximum over the tile's rows.
TFX_DEV int tile_kv_limit(const tfx_attn_args& p, size_t tok0, int q0, int n, int kve, bool row_valid) {
  int lim = p.kv_end[tok0 + min(q0 + 127, n - 1)];
  if (p.n_kv > 0) {                                              // block-uniform
    __shared__ int kvl[4];
    int m = row_valid ? kve : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) kvl[threadIdx.x >> 6] = m;
    __syncthreads();
    lim = max(max(kvl[0], kvl[1]), max(kvl[2], kvl[3]));
  }
  return __builtin_amdgcn_readfirstlane(lim);
}
TFX_DEV bf16x8 pack8(const f32x16& v, int tt) {
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; e++) o[e] = f2bf(v[tt * 8 + e]);
  return o;
}
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

// ------------------------------------------------------------------------------------------------
// forward: block = 128 query rows (4 waves x 32), loop over 64-key tiles
// ------------------------------------------------------------------------------------------------
// Block order.  A block's work grows with its query tile (forward, dQ: keys 0 .. kv_end) or shrinks with its key block (dK/dV:
// queries q_start .. n), 1 : 8 across a 1024-token sample.  The grid is (heads, samples, tiles) with the tile rank in the SLOWEST
// dimension, decoded heaviest-first: the long blocks start first and the kernel ends on the short ones (no tail of lone 8-unit
// blocks), and since workgroups go to the XCDs round-robin in linear order, every tile of one (head, sample) lands on the SAME XCD
// whenever heads * samples is a multiple of 8.  (Measured round 6, profiles/r06_attn_pmc_l2_by_block_order.txt: the L2 hit rate of this order is 5 % - 512 other
// pairs run between two tiles of a pair; an XCD-local order reaches 68 % and is SLOWER in the step: the operands come from the Infinity Cache either way and the
// heaviest-first balance matters more.  The tile-fastest grid of round 1 was removed.)
// A wave's 32 x 64 bf16 output block (this lane: row l & 31, columns db * 32 + 8 rg + 4 hi .. + 3) leaves through a wave-private 4 KiB LDS image:
// 16-byte stores, 8 lanes per 128-byte row = 8 cache lines per instruction.  The direct form - 8 bytes per lane in accumulator shape - puts
// 32 rows behind every store instruction and visits each line 8 times (the address coalescer walks the lines one by one: the same effect that
// cost the NT epilogues 7 k clocks per tile, gemm.hip staged_epilogue_bf16).  `st` must be free: the callers pass a tile buffer behind a barrier.
// Falls back to the direct stores when a row is not 16-byte aligned.
TFX_DEV void wave_block_store(bf16* st, const bf16x4 (&v)[2][4], bf16* g0, int ld, int rows_valid, int lane = -1) {
  const int l = (lane >= 0 ? lane : (int)threadIdx.x) & 63, hi = l >> 5, r = l & 31, ch = l & 7;
  if (((ld & 7) | (int)(((uintptr_t)g0 >> 1) & 7)) != 0) {                      // wave-uniform
    if (r < rows_valid) {
#pragma unroll
      for (int db = 0; db < 2; db++)
#pragma unroll
        for (int rg = 0; rg < 4; rg++) *(bf16x4*)(g0 + (size_t)r * ld + db * 32 + 8 * rg + 4 * hi) = v[db][rg];
    }
    return;
  }
#pragma unroll
  for (int db = 0; db < 2; db++)
#pragma unroll
    for (int rg = 0; rg < 4; rg++) {
      const int col = db * 32 + 8 * rg + 4 * hi;
      *(bf16x4*)(st + r * 64 + (((col >> 3) ^ ((r >> 1) & 7)) << 3) + (col & 7)) = v[db][rg];
    }
  bf16x8 t[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int row = q * 8 + (l >> 3);
    t[q] = *(const bf16x8*)(st + row * 64 + ((ch ^ ((row >> 1) & 7)) << 3));
  }
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int row = q * 8 + (l >> 3);
    if (row < rows_valid) *(bf16x8*)(g0 + (size_t)row * ld + ch * 8) = t[q];
  }
}

// wave_block_store with the backward of QK-RMSNorm + RoPE applied on the way out (tfx.h tfx_attn_args.nr_*; arithmetic of tokenwise.hip qk_norm_rope_bwd_k on
// the same bf16-rounded d q~ / d k~ chunk).  The read-back shape of the staging image IS that kernel's: lane = (row q * 8 + (l >> 3), chunk l & 7), the 8 lanes
// of a row hold one 64-wide head vector, so both of its reductions are group8_sum.  `tok` = token index of the block's first row, `colq` = column of this head
// in the raw / output matrices (WHICH = 1: the k half).  Rows past the end are computed on the last valid row and not stored (the cross-lane sums need
// every lane).  pg: this lane's gain-gradient partials of columns (l & 7) * 8 .. + 7.
template <int WHICH>
TFX_DEV void wave_block_store_nr(bf16* st, const bf16x4 (&v)[2][4], const tfx_attn_args& p, size_t tok, int colq, int rows_valid, float (&pg)[8], int lane = -1) {
  const int l = (lane >= 0 ? lane : (int)threadIdx.x) & 63, hi = l >> 5, r = l & 31, ch = l & 7;
  if (rows_valid <= 0) return;                                    // (wave-uniform) the whole block lies past the sample's end: nothing to read or write
#pragma unroll
  for (int db = 0; db < 2; db++)
#pragma unroll
    for (int rg = 0; rg < 4; rg++) {
      const int col = db * 32 + 8 * rg + 4 * hi;
      *(bf16x4*)(st + r * 64 + (((col >> 3) ^ ((r >> 1) & 7)) << 3) + (col & 7)) = v[db][rg];
    }
  const float sc = (WHICH == 0 ? p.nr_q_scale : 1.f) * (p.nr_norm_scale > 0.f ? p.nr_norm_scale : 8.f);
  const float* gmp = (WHICH == 0 ? p.nr_gamma_q : p.nr_gamma_k) + ch * 8;
  const f32x4 g0 = *(const f32x4*)gmp, g1 = *(const f32x4*)(gmp + 4);
  // every global request of the block ahead of the first use: raw chunk, rotary position -> cos / sin rows
  bf16x8 x8[4], t[4];
  f32x4 cs[4], sn[4];
  const int last = max(rows_valid - 1, 0);
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int row = min(q * 8 + (l >> 3), last);
    const size_t tt = tok + row;
    x8[q] = *(const bf16x8*)(p.nr_qkv + tt * p.nr_ld_qkv + colq + ch * 8);
    const int pos = p.nr_rot_pos[tt];
    cs[q] = *(const f32x4*)(p.nr_cos + (size_t)pos * 32 + ch * 4);
    sn[q] = *(const f32x4*)(p.nr_sin + (size_t)pos * 32 + ch * 4);
  }
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int row = min(q * 8 + (l >> 3), last);
    t[q] = *(const bf16x8*)(st + row * 64 + ((ch ^ ((row >> 1) & 7)) << 3));
  }
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int row = q * 8 + (l >> 3);
    const bool live = row < rows_valid;
    float xv[8], qq = 0.f;
#pragma unroll
    for (int e = 0; e < 8; e++) { xv[e] = bf2f(x8[q][e]); qq = fmaf(xv[e], xv[e], qq); }
    qq = group8_sum(qq);
    // (v_rsq / v_rcp forms: the epilogue runs inside two VALU-bound kernels - an IEEE divide is a ten-instruction sequence; 1 ulp against the token-wise kernel)
    const float inv = qq > 1e-24f ? __builtin_amdgcn_rsqf(qq) : 1e12f;
    const float isc = live ? inv * sc : 0.f;                        // rows past the end add nothing to the gain gradients
    float dyn[8], S = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const float da = bf2f(t[q][2 * i]), db = bf2f(t[q][2 * i + 1]);
      const float ga = fmaf(db, sn[q][i], da * cs[q][i]);            // inverse rotation
      const float gb = fmaf(-da, sn[q][i], db * cs[q][i]);
      const float gma = i < 2 ? g0[2 * i] : g1[2 * i - 4], gmb = i < 2 ? g0[2 * i + 1] : g1[2 * i - 3];
      const float ca = sc * (1.f + gma), cb = sc * (1.f + gmb);
      const float ta = ga * xv[2 * i], tb = gb * xv[2 * i + 1];     // shared by the gain gradient and by S = sum dyn x
      pg[2 * i] = fmaf(ta, isc, pg[2 * i]); pg[2 * i + 1] = fmaf(tb, isc, pg[2 * i + 1]);
      S = fmaf(ta, ca, S); S = fmaf(tb, cb, S);
      dyn[2 * i] = ga * ca; dyn[2 * i + 1] = gb * cb;
    }
    S = group8_sum(S);
    const float k = S * inv * inv * inv;
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; e++) o[e] = f2bf(fmaf(dyn[e], inv, -(xv[e] * k)));
    if (live) *(bf16x8*)(p.nr_dqkv + (tok + row) * p.nr_ld_dqkv + colq + ch * 8) = o;
  }
}
// gain gradients of a block: the 8 lanes of a wave that own the same 8 columns are summed by three exchanges, the four waves through LDS, 64 atomics
TFX_DEV void nr_flush_dgamma(float (&pg)[8], float* sg /* [4][64] */, float* dgamma, int t = -1) {
  const int tx = t >= 0 ? t : (int)threadIdx.x;
#pragma unroll
  for (int e = 0; e < 8; e++) {
#pragma unroll
    for (int m = 8; m < 64; m <<= 1) pg[e] += __shfl_xor(pg[e], m, 64);
  }
  const int lane = tx & 63, wv = tx >> 6;
  if (lane < 8) {
#pragma unroll
    for (int e = 0; e < 8; e++) sg[wv * 64 + lane * 8 + e] = pg[e];
  }
  __syncthreads();
  if (tx < 64) {
    const float s4 = sg[tx] + sg[64 + tx] + sg[128 + tx] + sg[192 + tx];
    if (s4 != 0.f) atomicAdd(dgamma + tx, s4);
  }
}
struct BlockId { int tile, h, b; };
TFX_DEV BlockId decode_block(int order, int ntile, bool heavy_last_tile) {     // (scalars only: no reference to the kernel-argument struct)
  BlockId o;
  (void)order;
  o.tile = heavy_last_tile ? ntile - 1 - (int)blockIdx.z : (int)blockIdx.z;
  o.h = blockIdx.x; o.b = blockIdx.y;
  return o;
}

// ------------------------------------------------------------------------------------------------
// forward, software-pipelined form (the plain loop it replaced in round 2 was removed in round 6)
// ------------------------------------------------------------------------------------------------
// Measured on gfx950 (tools/valu_probe.hip, tools/overlap_probe.hip): a wave64 v_fma_f32 issues in 2.7 clocks, the PACKED f32 forms in 6.5
// (slower than two scalar ops), v_exp_f32 in 8.5, v_cvt_pk_bf16_f32 in 4.6 - and one v_mfma_f32_32x32x16_bf16 occupies the SIMD's matrix
// pipe for 32 clocks during which the SAME wave can issue ~10 independent vector instructions for free (MFMA + 8 v_fma: 44 clocks vs
// 32 + 42 serial).  The soft-capped softmax costs ~24 VALU clocks per score against 16 MFMA clocks per score (S and P.V), so the kernel is
// VALU-bound and the matrix work only hides if it is issued INSIDE the vector stream.  The plain loop cannot do that: S(j) -> softmax(j) ->
// P.V(j) is one dependency chain per tile.  Here the chain is cut in 32-key UNITS u = 0, 1, ...  and phase u runs
//       vector: soft-cap + exp2 + row sums + bf16 packing of unit u            (16 scores per lane, 8 chunks of 2)
//       matrix: P.V of unit u - 1 (4 MFMAs)  and  S of unit u + 1 (4 MFMAs)    (one MFMA issued after each chunk)
// - three independent streams, so every MFMA has ~40 clocks of vector work to hide behind.  K / V tiles arrive by LDS-DMA into rings of
// three 64-key tiles (K two tiles ahead, V one), ONE barrier per tile.  A wave stops at its own last unit (causal diagonal), so the
// fully masked units of the upper waves of a block are never computed.
// accumulate IN PLACE: the asm form ties destination and C operand (the builtin may pick a second register set and copy).  Its only readers are further MFMAs on the same accumulator (no wait
// states needed for an accumulate chain) and the epilogue, which pads the MFMA -> VALU hazard itself (cdna_hip_programming.md 5.7).
#define MFMA_ACC(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc), "+v"(a), "+v"(b))
// s_cur: this unit's scores, already soft-capped into the log2 domain (and -inf where masked) by the caller: the matrix work of the phase
// hides under exp2 / row sums / packing, ONE straight-line variant (branchy code around accumulators that live across it makes hipcc
// shuffle them between register sets)
TFX_DEV void fwd_phase(const f32x16& s_cur, f32x16& s_nxt, u32x4 (&p_prev)[2], u32x4 (&p_cur)[2], f32x16 (&o)[2], const bf16x8 (&qf)[4],
                       const bf16* Kt_nxt, int kb_nxt, const bf16* Vt_prev, int kb_prev, float (&lsum)[2]) {
  const int hi = (threadIdx.x >> 5) & 1;
  // operand of MFMA m: 0..3 = K fragments of unit u + 1 (ks = m), 4..7 = V^T fragments of unit u - 1 (tt = (m - 4) >> 1, db = m & 1).
  // S first: its result is read by vector code at the start of the next phase - the four P.V MFMAs behind it are the hazard distance
  // (all MFMAs of the phase are asm volatile: program order is issue order, and hipcc pads nothing for asm results)
  bf16x8 frag[2];
  auto fetch = [&](int m) -> bf16x8 {
    if (m < 4) return dma_rowfrag(Kt_nxt, kb_nxt * 32, m);
    const int ra = kb_prev * 32 + 16 * ((m - 4) >> 1) + 4 * hi;
    return dma_tr8(Vt_prev, ra, ra + 8, (m & 1) * 32);
  };
  frag[0] = fetch(0);
#pragma unroll
  for (int ch = 0; ch < 8; ch++) {
    if (ch < 7) frag[(ch + 1) & 1] = fetch(ch + 1);               // LDS reads of the NEXT MFMA's operand fly under this chunk's vector work
    float e0 = __builtin_amdgcn_exp2f(s_cur[2 * ch]), e1 = __builtin_amdgcn_exp2f(s_cur[2 * ch + 1]);
    // the chunk's exponentials are operands of the MFMA statement: they are computed BEFORE it, their consumers (row sums, packing) after
    // The A / B operands are in-out ("+v") although the MFMA only reads them: hipcc would otherwise hand a dead operand register to the very
    // next vector instruction, and a VALU write in the first cycles after the issue corrupts the operand the MFMA is still reading (seen:
    // wrong O columns, timing dependent).  Kept "live", they are next written by the LDS reads / packing two chunks later.
    if (ch == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %3, %4, 0" : "=&v"(s_nxt), "+v"(e0), "+v"(e1), "+v"(frag[0]) : "v"(qf[0]));
    else if (ch < 4) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %3, %4, %0" : "+v"(s_nxt), "+v"(e0), "+v"(e1), "+v"(frag[ch & 1]) : "v"(qf[ch]));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %3, %4, %0" : "+v"(o[ch & 1]), "+v"(e0), "+v"(e1), "+v"(frag[ch & 1]), "+v"(p_prev[(ch - 4) >> 1]));
    lsum[0] += e0; lsum[1] += e1;
    bf16x2 pk2; pk2[0] = f2bf(e0); pk2[1] = f2bf(e1);
    uint32_t pk = __builtin_bit_cast(uint32_t, pk2);
    asm volatile("" : "+v"(lsum[0]), "+v"(lsum[1]), "+v"(pk));      // ... and before the next chunk's MFMA (volatile statements keep their order)
    p_cur[ch >> 2][ch & 3] = pk;
    __builtin_amdgcn_sched_barrier(0);                              // keep one MFMA per chunk: the matrix pipe stays busy under the vector stream
  }
}

// soft-cap of the unit (polynomial degree by the wave's largest |score|, see softcap16), mask where the unit crosses kv_end, then the phase
TFX_DEV void fwd_phase_any(f32x16& s_cur, f32x16& s_nxt, u32x4 (&p_prev)[2], u32x4 (&p_cur)[2], f32x16 (&o)[2], const bf16x8 (&qf)[4],
                           const bf16* Kt_nxt, int kb_nxt, const bf16* Vt_prev, int kb_prev, float (&lsum)[2], const SoftCap& c, int key0, int kve,
                           bool mask, unsigned long long* ts = nullptr) {
#ifdef TFX_ATTN_TIMING
  unsigned long long t0 = __builtin_readcyclecounter();
#endif
  softcap16(s_cur, c);
  if (mask) {                                                       // boundary units only (one or two per wave): exp2(-inf) = 0
#pragma unroll
    for (int r = 0; r < 16; r++) s_cur[r] = key0 + (r & 3) + 8 * (r >> 2) < kve ? s_cur[r] : -__builtin_inff();
  }
#ifdef TFX_ATTN_TIMING
  asm volatile("" : "+v"(s_cur));
  unsigned long long t1 = __builtin_readcyclecounter();
#endif
  fwd_phase(s_cur, s_nxt, p_prev, p_cur, o, qf, Kt_nxt, kb_nxt, Vt_prev, kb_prev, lsum);
#ifdef TFX_ATTN_TIMING
  asm volatile("" : "+v"(s_nxt), "+v"(o[0]), "+v"(o[1]));
  ts[2] += t1 - t0; ts[3] += __builtin_readcyclecounter() - t1;
#endif
}

TFX_DEV void fwd_drain(u32x4 (&p_prev)[2], f32x16 (&o)[2], const bf16* Vt_prev, int kb_prev) {
  const int hi = (threadIdx.x >> 5) & 1;
  bf16x8 vt[4];
#pragma unroll
  for (int m = 0; m < 4; m++) { const int ra = kb_prev * 32 + 16 * (m >> 1) + 4 * hi; vt[m] = dma_tr8(Vt_prev, ra, ra + 8, (m & 1) * 32); }
#pragma unroll
  for (int m = 0; m < 4; m++) MFMA_ACC(o[m & 1], vt[m], p_prev[m >> 1]);
  asm volatile("s_nop 7" : "+v"(vt[0]), "+v"(vt[1]), "+v"(vt[2]), "+v"(vt[3]), "+v"(p_prev[0]), "+v"(p_prev[1]));     // operands stay untouched while the last MFMA reads them
}

// ASM (round 6): the tiles every row of the block sees in full run in ONE generated asm statement (tools/gen_attn_loops.py: fixed registers, every LDS address a
// lane register + an immediate, the vector stream at its 88-instruction floor per unit with the MFMAs and LDS reads placed inside it); the boundary tiles - the
// last two of a causal block - stay on the C++ phases below.  The statement follows the C++ loop's tile protocol (wait + barrier at the top of a tile, then the
// requests for K(j + 2) / V(j + 1)) and its arithmetic order: the hand-over is at a tile boundary and the outputs are bit-identical to the <false> form.
// Training layouts only (no KV cache, no compacted rows: kv_end non-decreasing over a sample's rows), soft-cap plan modes 0 / 1.
template <bool ASM>
__global__ __launch_bounds__(256, 2) void attn_fwd_pipe_kernel(tfx_attn_args p) {
  __shared__ __attribute__((aligned(1024))) bf16 Ks[3][64 * 64];      // rings of LDS-DMA tiles (see swz_f): K(j), K(j+1), K(j+2) / V(j-1), V(j), V(j+1)
  __shared__ __attribute__((aligned(1024))) bf16 Vs[3][64 * 64];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, hi = l >> 5;
  const BlockId bi = decode_block(p.order, (p.n + 127) / 128, true);
  const int h = bi.h, b = bi.b, q0 = bi.tile * 128;
  const int n = p.q_cnt ? p.q_cnt[b] : p.n;                      // compacted decode steps (tfx.h q_row0, q_cnt)
  if (q0 >= n) return;                                           // (block-uniform: before any barrier)
  const int nkv = p.n_kv > 0 ? p.n_kv : n;
  const size_t tok0 = p.q_row0 ? (size_t)p.q_row0[b] : (size_t)b * n, tokk = (size_t)b * nkv;
  const bf16* qb = p.q + tok0 * p.ld_q + h * DH;
  const bf16* kb_ = p.k + tokk * p.ld_k + h * DH;
  const bf16* vb = p.v + tokk * p.ld_v + h * DH;
  const int qrow = q0 + w * 32 + (l & 31);
  const int qc = min(qrow, n - 1);
  const int kve = p.kv_end[tok0 + qc];
  const int kv_limit = tile_kv_limit(p, tok0, q0, n, kve, qrow < n);
  const int nt = (kv_limit + 63) / 64;
  bf16x8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ks++) qf[ks] = g_rowfrag(qb, p.ld_q, qrow, n, ks);
  const SoftCap sc_ = make_softcap(p.softcap, p.sc_plan);
  const int kve_min = wave_min_i(kve);
  asm volatile("" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3]));     // consume the compiler-visible loads before the counted DMA waits
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  tile_dma(kb_, p.ld_k, 0, nkv, Ks[0]);
  tile_dma(vb, p.ld_v, 0, nkv, Vs[0]);
  if (nt > 1) tile_dma(kb_, p.ld_k, 64, nkv, Ks[1]);

  f32x16 o[2], sA, sB;
  u32x4 pA[2], pB[2];                                             // P^T of a unit as MFMA operands: 4 packed bf16 pairs per 16 keys
#pragma unroll
  for (int i = 0; i < 2; i++) {
#pragma unroll
    for (int r = 0; r < 16; r++) o[i][r] = 0.f;
#pragma unroll
    for (int e = 0; e < 4; e++) { pA[i][e] = 0u; pB[i][e] = 0u; }
  }
#pragma unroll
  for (int r = 0; r < 16; r++) { sA[r] = 0.f; sB[r] = 0.f; }
  float lsum[2] = {0.f, 0.f};
  int kslot = 0, vslot = 0;                                       // ring slots of K(j) / V(j)
#ifdef TFX_ATTN_TIMING
  unsigned long long tsec[4] = {0, 0, 0, 0}, tprev = __builtin_readcyclecounter();
#define AT_MARK(i) { const unsigned long long tn = __builtin_readcyclecounter(); tsec[i] += tn - tprev; tprev = tn; }
#define TS_ARG , tsec
#else
#define AT_MARK(i)
#define TS_ARG
#endif
  int j0 = 0;
  if constexpr (ASM) {
    // tiles 0 .. nfull - 1: every key visible to every row of the block (kv_end is non-decreasing: the block's first row sees the fewest), and the
    // statement's unconditional requests for K(j + 2) / V(j + 1) stay inside the sample
    const int kve_blk = __builtin_amdgcn_readfirstlane(p.kv_end[tok0 + q0]);
    const int nfull = __builtin_amdgcn_readfirstlane(min(kve_blk >> 6, (nkv - 128) >> 6));
    if (nfull > 0 && sc_.mode <= 1) {                              // (block-uniform)
      const int wu = __builtin_amdgcn_readfirstlane(w);
      const uint32_t ldsK = (uint32_t)(size_t)(lds_void_t*)&Ks[0][0], ldsV = (uint32_t)(size_t)(lds_void_t*)&Vs[0][0];
      uint32_t ka[4], vaA[2], vaB[2], dk[2], dv[2];
#pragma unroll
      for (int ks = 0; ks < 4; ks++) ka[ks] = ldsK + 2u * (uint32_t)((l & 31) * 64 + (((2 * ks + hi) ^ swz_f(l & 31)) << 3));      // dma_rowfrag, slot 0, key block 0
      {
        const int q = l & 15, ra = 4 * hi + (q >> 2), rb = ra + 8;                                                               // dma_tr8, slot 0, rowA = 4 hi
#pragma unroll
        for (int db = 0; db < 2; db++) {
          const int col = db * 32 + 16 * ((l >> 4) & 1) + 4 * (q & 3);
          vaA[db] = ldsV + 2u * (uint32_t)(ra * 64 + (((col >> 3) ^ swz_f(ra)) << 3) + (col & 7));
          vaB[db] = ldsV + 2u * (uint32_t)(rb * 64 + (((col >> 3) ^ swz_f(rb)) << 3) + (col & 7));
        }
      }
#pragma unroll
      for (int jp = 0; jp < 2; jp++) {                                                                                           // tile_dma's pieces as buffer offsets
        const int r = w * 16 + jp * 8 + (l >> 3), c = (l & 7) ^ swz_f(r);
        dk[jp] = 2u * (uint32_t)(r * p.ld_k + c * 8);
        dv[jp] = 2u * (uint32_t)(r * p.ld_v + c * 8);
      }
      const uint64_t baseK = (uint64_t)(uintptr_t)kb_, baseV = (uint64_t)(uintptr_t)vb;
      u32x4 rsK, rsV;
      rsK[0] = __builtin_amdgcn_readfirstlane((uint32_t)baseK); rsK[1] = __builtin_amdgcn_readfirstlane((uint32_t)(baseK >> 32) & 0xffffu); rsK[2] = 0xffffff00u; rsK[3] = 0x00020000u;
      rsV[0] = __builtin_amdgcn_readfirstlane((uint32_t)baseV); rsV[1] = __builtin_amdgcn_readfirstlane((uint32_t)(baseV >> 32) & 0xffffu); rsV[2] = 0xffffff00u; rsV[3] = 0x00020000u;
      const uint32_t stk = __builtin_amdgcn_readfirstlane(128u * (uint32_t)p.ld_k), stv = __builtin_amdgcn_readfirstlane(128u * (uint32_t)p.ld_v);
      uint32_t sko = 2u * stk, svo = stv, cnt = (uint32_t)nfull;
      const uint32_t mk = __builtin_amdgcn_readfirstlane(ldsK + (uint32_t)wu * 2048u), mv = __builtin_amdgcn_readfirstlane(ldsV + (uint32_t)wu * 2048u);
      const float p1 = sc_.p1, p3 = sc_.p3, p5 = sc_.p5;
      u32x4 q0f = __builtin_bit_cast(u32x4, qf[0]), q1f = __builtin_bit_cast(u32x4, qf[1]), q2f = __builtin_bit_cast(u32x4, qf[2]), q3f = __builtin_bit_cast(u32x4, qf[3]);
      // EARLY-CLOBBER in-outs ("+&"): `svo` starts out equal to the input `stv`; without the & hipcc gave both ONE register (it assumes a statement reads its inputs
      // before it writes its outputs) and the V tile offset doubled per tile instead of advancing (first GPU trips: V(4) in V(3)'s slot - and only in the -fPIC build)
#define AF_OPERANDS                                                                                                                                  \
          : [o0] "+&v"(o[0]), [o1] "+&v"(o[1]), "+{v[64:79]}"(sA), "+{v[104:107]}"(pB[0]), "+{v[108:111]}"(pB[1]), "+{v112}"(lsum[0]), "+{v113}"(lsum[1]),    \
            [sko] "+&s"(sko), [svo] "+&s"(svo), [cnt] "+&s"(cnt)                                                                                     \
          : [qf0] "v"(q0f), [qf1] "v"(q1f), [qf2] "v"(q2f), [qf3] "v"(q3f), [ka0] "v"(ka[0]), [ka1] "v"(ka[1]), [ka2] "v"(ka[2]), [ka3] "v"(ka[3]),  \
            [va0] "v"(vaA[0]), [va1] "v"(vaA[1]), [vb0] "v"(vaB[0]), [vb1] "v"(vaB[1]), [dk0] "v"(dk[0]), [dk1] "v"(dk[1]), [dv0] "v"(dv[0]), [dv1] "v"(dv[1]), \
            [rsk] "s"(rsK), [rsv] "s"(rsV), [stk] "s"(stk), [stv] "s"(stv), [mk] "s"(mk), [mv] "s"(mv), [p1] "v"(p1), [p3] AF_P3(p3), [p5] "s"(p5)   \
          : "memory", "scc", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95",         \
            "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124",      \
            "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141",  \
            "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151"
      if (sc_.mode == 0) {
#define AF_P3 "s"
        asm volatile(
#include "attn_fwd_loop_m0.inc"
            AF_OPERANDS);
#undef AF_P3
      } else {
#define AF_P3 "v"
        asm volatile(
#include "attn_fwd_loop_m1.inc"
            AF_OPERANDS);
#undef AF_P3
      }
#undef AF_OPERANDS
      j0 = nfull;
      kslot = vslot = nfull % 3;
    }
  }
  for (int j = j0; j < nt; j++) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's pieces of K(j + 1), V(j) have landed
    __builtin_amdgcn_s_barrier();                                 // ... everyone's have, and everyone is done with K(j - 1), V(j - 2)
    AT_MARK(0)
    const int kslot1 = kslot == 2 ? 0 : kslot + 1, kslot2 = kslot1 == 2 ? 0 : kslot1 + 1;
    const int vslot1 = vslot == 2 ? 0 : vslot + 1, vprev = vslot == 0 ? 2 : vslot - 1;
    if (j + 2 < nt) tile_dma(kb_, p.ld_k, (j + 2) * 64, nkv, Ks[kslot2]);
    if (j + 1 < nt) tile_dma(vb, p.ld_v, (j + 1) * 64, nkv, Vs[vslot1]);
    AT_MARK(1)
    if (j == 0) {                                                 // pipeline fill: S of unit 0
      sA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dma_rowfrag(Ks[0], 0, 0), qf[0], f32x16{}, 0, 0, 0);
#pragma unroll
      for (int ks = 1; ks < 4; ks++) sA = MFMA(dma_rowfrag(Ks[0], 0, ks), qf[ks], sA);
    }
    // even unit u = 2j: softmax of sA -> pA; P.V of unit 2j - 1 (pB, second half of V(j - 1); p = 0 for j = 0); S of unit 2j + 1 -> sB.
    // Units past a wave's last visible key run fully masked (p = 0): no per-wave control flow around the accumulators.
    int u = 2 * j;
    fwd_phase_any(sA, sB, pB, pA, o, qf, Ks[kslot], 1, j == 0 ? Vs[vslot] : Vs[vprev], 1, lsum, sc_, u * 32 + 4 * hi, kve, (u + 1) * 32 > kve_min TS_ARG);
    // odd unit u = 2j + 1: softmax of sB -> pB; P.V of unit 2j (pA, first half of V(j)); S of unit 2j + 2 -> sA (first half of K(j + 1))
    u = 2 * j + 1;
    fwd_phase_any(sB, sA, pA, pB, o, qf, Ks[kslot1], 0, Vs[vslot], 0, lsum, sc_, u * 32 + 4 * hi, kve, (u + 1) * 32 > kve_min TS_ARG);
    kslot = kslot1; vslot = vslot1;
#ifdef TFX_ATTN_TIMING
    tprev = __builtin_readcyclecounter();
#endif
  }
#ifdef TFX_ATTN_TIMING
  if (l == 0) {
    unsigned long long* ob = (unsigned long long*)p.dq + ((((size_t)b * p.h + h) * ((n + 127) / 128) + bi.tile) * 4 + w) * 5;
    for (int i = 0; i < 4; i++) ob[i] = tsec[i];
    ob[4] = nt;
  }
#endif
#undef AT_MARK
#undef TS_ARG
  fwd_drain(pB, o, Vs[vslot == 0 ? 2 : vslot - 1], 1);                        // P.V of the very last unit
  asm volatile("s_nop 15\n\ts_nop 15" : "+v"(o[0]), "+v"(o[1]));              // the asm MFMAs' results are read by vector code from here on
  float ls = lsum[0] + lsum[1];
  ls += __shfl_xor(ls, 32, 64);
  {
    const float g = sigmoidf_(bf2f(p.gate[(tok0 + qc) * p.ld_gate + h]));
    const float sc = g * __builtin_amdgcn_rcpf(ls);
    bf16x4 ov[2][4];
#pragma unroll
    for (int db = 0; db < 2; db++)
#pragma unroll
      for (int rg = 0; rg < 4; rg++)
#pragma unroll
        for (int e = 0; e < 4; e++) ov[db][rg][e] = f2bf(o[db][rg * 4 + e] * sc);
    __syncthreads();                                            // every wave is through with the K / V tiles: their LDS becomes the staging area
    wave_block_store(&Ks[0][0] + w * 2048, ov, p.out + (tok0 + q0 + w * 32) * p.ld_out + h * DH, p.ld_out, n - (q0 + w * 32));
    if (qrow < n && hi == 0) p.lse[((size_t)b * p.h + h) * p.n + qrow] = __log2f(ls) * LN2;
  }
}

// ------------------------------------------------------------------------------------------------
// backward prep: delta = sum_d dout*og ; dgate = delta*(1-sigmoid(g)) ; do_eff = dout*sigmoid(g)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_bwd_prep_kernel(tfx_attn_args p) {
  // 32-bit index math (the launcher checks b * n * h * 8 < 2^31): 64-bit divides by run-time values cost ~100 instructions each
  const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned vid = gid >> 3;
  const int sub = gid & 7;
  const unsigned T = (unsigned)p.b * p.n;
  if (vid >= T * p.h) return;
  const unsigned t = vid / (unsigned)p.h; const int h = (int)(vid - t * p.h);
  const bf16x8 d8 = *(const bf16x8*)(p.dout + (size_t)t * p.ld_dout + h * DH + sub * 8);
  const bf16x8 o8 = *(const bf16x8*)(p.out + (size_t)t * p.ld_out + h * DH + sub * 8);
  const float g = sigmoidf_(bf2f(p.gate[(size_t)t * p.ld_gate + h]));
  float dl = 0.f; bf16x8 e8;
#pragma unroll
  for (int e = 0; e < 8; e++) { float d = bf2f(d8[e]); dl += d * bf2f(o8[e]); e8[e] = f2bf(d * g); }
  *(bf16x8*)(p.do_eff + (size_t)t * p.ld_do + h * DH + sub * 8) = e8;
  dl = group8_sum(dl);
  if (sub == 0) {
    const unsigned bb = t / (unsigned)p.n, i = t - bb * p.n;
    p.delta[((size_t)bb * p.h + h) * p.n + i] = dl;
    p.dgate[(size_t)t * p.ld_dgate + h] = f2bf(dl * (1.f - g));
  }
}

// ------------------------------------------------------------------------------------------------
// backward dQ: block = 128 query rows, loop over 64-key tiles (forward structure + one more MFMA)
// (the plain loop: kept as the A/B reference of the pipelined kernel below, TFX_ATTN_BWD_PIPE=0)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(tfx_attn_args p) {
  __shared__ __attribute__((aligned(16))) bf16 Ks[64 * LDT];
  __shared__ __attribute__((aligned(16))) bf16 Vs[64 * LDT];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, hi = l >> 5;
  const int n = p.n;
  const BlockId bi = decode_block(p.order, (n + 127) / 128, true);
  const int h = bi.h, b = bi.b, q0 = bi.tile * 128;
  const size_t tok0 = (size_t)b * n;
  const bf16* qb = p.q + tok0 * p.ld_q + h * DH;
  const bf16* kb_ = p.k + tok0 * p.ld_k + h * DH;
  const bf16* vb = p.v + tok0 * p.ld_v + h * DH;
  const bf16* dob = p.do_eff + tok0 * p.ld_do + h * DH;
  const int qrow = q0 + w * 32 + (l & 31);
  const int qc = min(qrow, n - 1);
  const int kve = p.kv_end[tok0 + qc];
  const int kv_limit = p.kv_end[tok0 + min(q0 + 127, n - 1)];
  const int nt = (kv_limit + 63) / 64;
  const float lse2 = p.lse[((size_t)b * p.h + h) * n + qc] * LOG2E;
  float dlt;
  const int kve_min = wave_min_i(kve);

  bf16x8 qf[4], dof[4];
  dlt = p.delta[((size_t)b * p.h + h) * n + qc];
#pragma unroll
  for (int ks = 0; ks < 4; ks++) { qf[ks] = g_rowfrag(qb, p.ld_q, qrow, n, ks); dof[ks] = g_rowfrag(dob, p.ld_do, qrow, n, ks); }
  f32x16 dq[2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) dq[i][r] = 0.f;
  const SoftCap sc_ = make_softcap(p.softcap, p.sc_plan);
  f32x16 zero16;
#pragma unroll
  for (int r = 0; r < 16; r++) zero16[r] = 0.f;
  asm volatile("" : "+v"(zero16));

  TileRegs kr, vr;
  tile_gload(kr, kb_, p.ld_k, 0, n);
  tile_gload(vr, vb, p.ld_v, 0, n);
  for (int j = 0; j < nt; j++) {
    __syncthreads();
    tile_sstore(kr, Ks); tile_sstore(vr, Vs);
    __syncthreads();
    if (j + 1 < nt) { tile_gload(kr, kb_, p.ld_k, (j + 1) * 64, n); tile_gload(vr, vb, p.ld_v, (j + 1) * 64, n); }
    const bool need_mask = (j + 1) * 64 > kve_min;
#pragma unroll
    for (int kb = 0; kb < 2; kb++) {
      f32x16 s = MFMA(lds_rowfrag(Ks, kb * 32, 0), qf[0], zero16);   // S^T[key][q]; shared zero C operand
      f32x16 dp = MFMA(lds_rowfrag(Vs, kb * 32, 0), dof[0], zero16);   // dP^T[key][q]
#pragma unroll
      for (int ks = 1; ks < 4; ks++) {
        s = MFMA(lds_rowfrag(Ks, kb * 32, ks), qf[ks], s);
        dp = MFMA(lds_rowfrag(Vs, kb * 32, ks), dof[ks], dp);
      }
      if (sc_.mode == 0) {                                           // the layer's cubic and ITS derivative, both from u = s^2 (4 ops instead of 5.6)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const float u = s[r] * s[r];
          dp[r] = (dp[r] - dlt) * fmaf(u, sc_.d3, sc_.d1);
          s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], fmaf(u, sc_.p3, sc_.p1), -lse2));       // P^T
        }
      } else if (sc_.mode == 1) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const float u = s[r] * s[r];
          dp[r] = (dp[r] - dlt) * fmaf(u, fmaf(u, sc_.d5, sc_.d3), sc_.d1);
          s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], fmaf(u, fmaf(u, sc_.p5, sc_.p3), sc_.p1), -lse2));
        }
      } else {
        softcap16(s, sc_);                                           // s = s2 = cap*log2e*tanh(s/cap)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const float dth = fmaf(s[r] * s[r], -sc_.g2, 1.f);         // 1 - tanh^2
          dp[r] = (dp[r] - dlt) * dth;
          s[r] = __builtin_amdgcn_exp2f(s[r] - lse2);                 // P^T
        }
      }
      if (need_mask) {                                               // one scalar branch per 32-key block (boundary tiles only)
        const int key0 = j * 64 + kb * 32 + 4 * hi;
#pragma unroll
        for (int r = 0; r < 16; r++) s[r] = key0 + (r & 3) + 8 * (r >> 2) < kve ? s[r] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; r++) s[r] *= dp[r];                     // dS_raw^T
#pragma unroll
      for (int tt = 0; tt < 2; tt++) {
        const bf16x8 dsf = pack8(s, tt);
        const int ra = kb * 32 + 16 * tt + 4 * hi;
#pragma unroll
        for (int db = 0; db < 2; db++) dq[db] = MFMA(lds_tr8(Ks, LDT, ra, ra + 8, db * 32), dsf, dq[db]);   // dQ^T[d][q]
      }
    }
  }
  {
    bf16x4 ov[2][4];
#pragma unroll
    for (int db = 0; db < 2; db++)
#pragma unroll
      for (int rg = 0; rg < 4; rg++)
#pragma unroll
        for (int e = 0; e < 4; e++) ov[db][rg][e] = f2bf(dq[db][rg * 4 + e]);
    __syncthreads();                                            // the K / V tiles are free: staging area of the coalesced stores
    if (p.nr_qkv) {                                             // (kernel argument: block-uniform) QK-norm + RoPE backward on the way out
      __shared__ float sg[4 * 64];
      float pg[8];
#pragma unroll
      for (int e = 0; e < 8; e++) pg[e] = 0.f;
      wave_block_store_nr<0>((w < 2 ? Ks : Vs) + (w & 1) * 2048, ov, p, tok0 + q0 + w * 32, h * DH, n - (q0 + w * 32), pg);
      nr_flush_dgamma(pg, sg, p.nr_dgamma_q);
    } else
    wave_block_store((w < 2 ? Ks : Vs) + (w & 1) * 2048, ov, p.dq + (tok0 + q0 + w * 32) * p.ld_dq + h * DH, p.ld_dq, n - (q0 + w * 32));
  }
}

// ------------------------------------------------------------------------------------------------
// backward dQ, software-pipelined form (round 6; the product kernel for soft-cap plan modes 0 / 1)
// ------------------------------------------------------------------------------------------------
// What bounded the plain loop above (profiles/r06_attn_fwd_asm_ablation.txt, item 5): per launch its vector work is ~100 us, its MFMAs 55 us, its LDS traffic
// 55-65 us - and the kernel took their SUM (225 us).  S / dP -> soft-max arithmetic -> dQ is one dependency chain per 32-key block, the two waves of a SIMD
// drift into the same part of it, and K / V tiles went global -> registers -> LDS behind two block barriers per tile.  Here, as in the forward's pipe kernel,
// the chain is cut in 32-key UNITS and phase u runs three independent streams:
//       vector: u^2, soft-cap derivative, exp2, dS = P (dP - delta) tanh', bf16 packing of unit u         (16 scores per lane, 8 chunks of 2)
//       matrix: S and dP of unit u + 1 (8 MFMAs, two per chunk 0..3)  and  dQ += dS K of unit u - 1 (4 MFMAs, chunks 4..7)
// K tiles live in a ring of four (a tile is read from the S of its first unit to the dQ of its last one, two phases later), V tiles in a ring of three, both by
// LDS-DMA two tiles ahead, ONE barrier per tile.  Same arithmetic per element, same accumulation order per accumulator: bit-identical to the plain kernel.
template <int MODE>
TFX_DEV void dq_phase(const f32x16& s_cur, const f32x16& dp_cur, f32x16& s_nxt, f32x16& dp_nxt, u32x4 (&ds_prev)[2], u32x4 (&ds_cur)[2], f32x16 (&dq)[2],
                      const bf16x8 (&qf)[4], const bf16x8 (&dof)[4], const bf16* Kt_nxt, const bf16* Vt_nxt, int kb_nxt, const bf16* Kt_prev, int kb_prev,
                      float lse2, float dlt, const SoftCap& c, int lane) {
  const int hi = (lane >> 5) & 1;
  // operand of MFMA m: 0..7 = row fragments of unit u + 1 (k-step m >> 1; even: K for S, odd: V for dP), 8..11 = K^T fragments of unit u - 1 (tt = (m - 8) >> 1, db = m & 1)
  auto fetch = [&](int m) -> bf16x8 {
    if (m < 8) return dma_rowfrag((m & 1) ? Vt_nxt : Kt_nxt, kb_nxt * 32, m >> 1, lane);
    const int ra = kb_prev * 32 + 16 * ((m - 8) >> 1) + 4 * hi;
    return dma_tr8(Kt_prev, ra, ra + 8, (m & 1) * 32, lane);
  };
  // MFMA order over the 8 chunks: 2, 1, 2, 1, 2, 1, 2, 1 (S0 dP0 | S1 | dP1 S2 | dP2 | S3 dP3 | dQ0 | dQ1 dQ2 | dQ3) - never more than three operand fragments alive
  // (this chunk's and the next one's): MFMA m takes frag[m % 3], the next chunk's are requested before this chunk's vector work
  constexpr int first[9] = {0, 2, 3, 5, 6, 8, 9, 11, 12};          // first MFMA of chunk ch
  // MFMA index -> operand: 0 S0, 1 dP0, 2 S1, 3 dP1, 4 S2, 5 dP2, 6 S3, 7 dP3 (fetch(m): k-step m >> 1, odd = V), 8..11 dQ
  bf16x8 frag[3];
  frag[0] = fetch(0); frag[1] = fetch(1);
#pragma unroll
  for (int ch = 0; ch < 8; ch++) {
    if (ch < 7) {
#pragma unroll
      for (int m = first[ch + 1]; m < first[ch + 2 > 8 ? 8 : ch + 2]; m++) frag[m % 3] = fetch(m);
    }
    float d0, d1;
    {
#pragma unroll
      for (int e = 0; e < 2; e++) {
        const int r = 2 * ch + e;
        const float a = s_cur[r], u = a * a;
        float dth, arg;
        if constexpr (MODE == 0) { dth = fmaf(u, c.d3, c.d1); arg = fmaf(a, fmaf(u, c.p3, c.p1), -lse2); }
        else if constexpr (MODE == 1) { dth = fmaf(u, fmaf(u, c.d5, c.d3), c.d1); arg = fmaf(a, fmaf(u, fmaf(u, c.p5, c.p3), c.p1), -lse2); }
        else { dth = fmaf(u, -c.g2, 1.f); arg = a - lse2; }          // s_cur was soft-capped by the caller (softcap16): 1 - tanh^2 from the capped score
        const float dpv = (dp_cur[r] - dlt) * dth;
        const float pv = __builtin_amdgcn_exp2f(arg);
        (e == 0 ? d0 : d1) = pv * dpv;                              // dS_raw^T
      }
    }
    // the chunk's results are operands of the MFMA statements: computed BEFORE them, packed after (see fwd_phase)
#pragma unroll
    for (int m = first[ch]; m < first[ch + 1]; m++) {
      if (m == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %3, %4, 0" : "=&v"(s_nxt), "+v"(d0), "+v"(d1), "+v"(frag[m % 3]) : "v"(qf[0]));
      else if (m == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %3, %4, 0" : "=&v"(dp_nxt), "+v"(d0), "+v"(d1), "+v"(frag[m % 3]) : "v"(dof[0]));
      else if (m < 8 && !(m & 1)) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %3, %4, %0" : "+v"(s_nxt), "+v"(d0), "+v"(d1), "+v"(frag[m % 3]) : "v"(qf[m >> 1]));
      else if (m < 8) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %3, %4, %0" : "+v"(dp_nxt), "+v"(d0), "+v"(d1), "+v"(frag[m % 3]) : "v"(dof[m >> 1]));
      else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %3, %4, %0" : "+v"(dq[m & 1]), "+v"(d0), "+v"(d1), "+v"(frag[m % 3]), "+v"(ds_prev[(m - 8) >> 1]));
    }
    bf16x2 pk2; pk2[0] = f2bf(d0); pk2[1] = f2bf(d1);
    uint32_t pk = __builtin_bit_cast(uint32_t, pk2);
    asm volatile("" : "+v"(pk));
    ds_cur[ch >> 2][ch & 3] = pk;
    __builtin_amdgcn_sched_barrier(0);
  }
}
template <int MODE>
TFX_DEV void dq_phase_any(f32x16& s_cur, f32x16& dp_cur, f32x16& s_nxt, f32x16& dp_nxt, u32x4 (&ds_prev)[2], u32x4 (&ds_cur)[2], f32x16 (&dq)[2],
                          const bf16x8 (&qf)[4], const bf16x8 (&dof)[4], const bf16* Kt_nxt, const bf16* Vt_nxt, int kb_nxt, const bf16* Kt_prev, int kb_prev,
                          float lse2, float dlt, const SoftCap& c, int key0, int kve, bool mask, int lane) {
  if constexpr (MODE == 2) softcap16(s_cur, c);                     // no plan: the degree follows the wave's scores (not overlapped with the matrix work)
  // boundary units (one or two per wave): a masked score gets dP = delta, i.e. dS = P (delta - delta) tanh' = 0 - the plain kernel's P = 0 up to the sign of the
  // zero - and the score 0, so that its P = exp2(-lse2) stays finite (a large masked score against a small lse: inf x 0); ONE straight-line phase (a second,
  // masked instantiation of it pushed the kernel from 230 registers to 256 + 131 spilled)
  if (mask) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const bool vis = key0 + (r & 3) + 8 * (r >> 2) < kve;
      dp_cur[r] = vis ? dp_cur[r] : dlt;
      s_cur[r] = vis ? s_cur[r] : 0.f;
    }
  }
  dq_phase<MODE>(s_cur, dp_cur, s_nxt, dp_nxt, ds_prev, ds_cur, dq, qf, dof, Kt_nxt, Vt_nxt, kb_nxt, Kt_prev, kb_prev, lse2, dlt, c, lane);
}

template <int MODE>
TFX_DEV void dq_pipe_loop(const tfx_attn_args& p, bf16 (&Ks)[4][64 * 64], bf16 (&Vs)[3][64 * 64], const bf16* kb_, const bf16* vb, int n, int nt, const bf16x8 (&qf)[4],
                          const bf16x8 (&dof)[4], f32x16 (&dq)[2], float lse2_, float dlt_, const SoftCap& sc_, int kve, int kve_min, int tx) {
  const int hi = (tx >> 5) & 1, lane = tx & 63;
  f32x16 sA, sB, dpA, dpB;
  u32x4 dsA[2], dsB[2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int e = 0; e < 4; e++) { dsA[i][e] = 0u; dsB[i][e] = 0u; }
#pragma unroll
  for (int r = 0; r < 16; r++) { sA[r] = 0.f; sB[r] = 0.f; dpA[r] = 0.f; dpB[r] = 0.f; }
  int kslot = 0, vslot = 0;                                       // ring slots of K(j) / V(j)
  for (int j = 0; j < nt; j++) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's pieces of K(j + 1), V(j + 1) have landed
    __builtin_amdgcn_s_barrier();                                 // ... everyone's have, and everyone is done with K(j - 2), V(j - 1)
    const int kslot1 = (kslot + 1) & 3, kslot2 = (kslot + 2) & 3, kprev = (kslot + 3) & 3;
    const int vslot1 = vslot == 2 ? 0 : vslot + 1, vslot2 = vslot1 == 2 ? 0 : vslot1 + 1;
    if (j + 2 < nt) { tile_dma(kb_, p.ld_k, (j + 2) * 64, n, Ks[kslot2], tx); tile_dma(vb, p.ld_v, (j + 2) * 64, n, Vs[vslot2], tx); }
    if (j == 0) {                                                 // pipeline fill: S and dP of unit 0
      sA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dma_rowfrag(Ks[0], 0, 0, lane), qf[0], f32x16{}, 0, 0, 0);
      dpA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dma_rowfrag(Vs[0], 0, 0, lane), dof[0], f32x16{}, 0, 0, 0);
#pragma unroll
      for (int ks = 1; ks < 4; ks++) { sA = MFMA(dma_rowfrag(Ks[0], 0, ks, lane), qf[ks], sA); dpA = MFMA(dma_rowfrag(Vs[0], 0, ks, lane), dof[ks], dpA); }
    }
    // even unit u = 2j: vector work on (sA, dpA) -> dsA ; S / dP of unit 2j + 1 -> (sB, dpB) (second halves of K(j), V(j)) ; dQ of unit 2j - 1 (dsB, second half of
    // K(j - 1); dsB = 0 for j = 0, against K(0)'s own tile so that no uninitialised LDS meets the zero operand)
    int u = 2 * j;
    dq_phase_any<MODE>(sA, dpA, sB, dpB, dsB, dsA, dq, qf, dof, Ks[kslot], Vs[vslot], 1, j == 0 ? Ks[kslot] : Ks[kprev], 1, lse2_, dlt_, sc_, u * 32 + 4 * hi, kve, (u + 1) * 32 > kve_min, lane);
    // odd unit u = 2j + 1: (sB, dpB) -> dsB ; S / dP of unit 2j + 2 -> (sA, dpA) (first halves of K(j + 1), V(j + 1); past the last tile: a stale tile, unused results) ; dQ of unit 2j
    u = 2 * j + 1;
    dq_phase_any<MODE>(sB, dpB, sA, dpA, dsA, dsB, dq, qf, dof, Ks[kslot1], Vs[vslot1], 0, Ks[kslot], 0, lse2_, dlt_, sc_, u * 32 + 4 * hi, kve, (u + 1) * 32 > kve_min, lane);
    kslot = kslot1; vslot = vslot1;
  }
  {                                                               // drain: dQ of the very last unit (dsB, second half of K(nt - 1))
    const bf16* Kt = Ks[(kslot + 3) & 3];
    bf16x8 kt[4];
#pragma unroll
    for (int m = 0; m < 4; m++) { const int ra = 32 + 16 * (m >> 1) + 4 * hi; kt[m] = dma_tr8(Kt, ra, ra + 8, (m & 1) * 32, lane); }
#pragma unroll
    for (int m = 0; m < 4; m++) MFMA_ACC(dq[m & 1], kt[m], dsB[m >> 1]);
    asm volatile("s_nop 7" : "+v"(kt[0]), "+v"(kt[1]), "+v"(kt[2]), "+v"(kt[3]), "+v"(dsB[0]), "+v"(dsB[1]));
  }
}

// The whole tile loop is ONE generated asm statement (tools/gen_attn_loops.py program_dq) for plan modes 0 / 1; mode 2 (no plan) takes the C++ pipeline above.
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_pipe_kernel(tfx_attn_args p) {
  __shared__ __attribute__((aligned(1024))) bf16 ring[8][64 * 64];    // rings of LDS-DMA tiles (see swz_f): K(j - 1) .. K(j + 2) in slots 0 .. 3, V in slots 4 .. 7
  bf16 (&Ks)[4][64 * 64] = *reinterpret_cast<bf16 (*)[4][64 * 64]>(&ring[0][0]);
  bf16 (&Vs)[3][64 * 64] = *reinterpret_cast<bf16 (*)[3][64 * 64]>(&ring[4][0]);
  const int n = p.n, ntile = (n + 127) / 128;
  // (A pair-persistent grid - one block per (head, sample) walking its 8 query tiles - measured no faster, 227 vs 221 us: two blocks per CU already hide one
  //  block's start-up behind the other's loop.  profiles/r06_attn_bwd_what_bounds_it.txt)
  const int h = blockIdx.x, b = blockIdx.y, z = blockIdx.z;
  const int tx = threadIdx.x, l = tx & 63, w = tx >> 6, hi = l >> 5;
  const int q0 = (ntile - 1 - z) * 128;
  const size_t tok0 = (size_t)b * n;
  const bf16* qb = p.q + tok0 * p.ld_q + h * DH;
  const bf16* kb_ = p.k + tok0 * p.ld_k + h * DH;
  const bf16* vb = p.v + tok0 * p.ld_v + h * DH;
  const bf16* dob = p.do_eff + tok0 * p.ld_do + h * DH;
  // the first two K / V tiles are requested BEFORE anything is loaded: their addresses depend on the block index alone
  tile_dma(kb_, p.ld_k, 0, n, Ks[0], tx);
  tile_dma(vb, p.ld_v, 0, n, Vs[0], tx);
  if (n > 64) { tile_dma(kb_, p.ld_k, 64, n, Ks[1], tx); tile_dma(vb, p.ld_v, 64, n, Vs[1], tx); }
  const int qrow = q0 + w * 32 + (l & 31);
  const int qc = min(qrow, n - 1);
  const int kve = p.kv_end[tok0 + qc];
  const int kv_limit = p.kv_end[tok0 + min(q0 + 127, n - 1)];
  const int nt = (kv_limit + 63) / 64;
  const float lse2 = p.lse[((size_t)b * p.h + h) * n + qc] * LOG2E;
  const float dlt = p.delta[((size_t)b * p.h + h) * n + qc];
  const int kve_min = wave_min_i(kve);
  bf16x8 qf[4], dof[4];
#pragma unroll
  for (int ks = 0; ks < 4; ks++) { qf[ks] = g_rowfrag(qb, p.ld_q, qrow, n, ks, l); dof[ks] = g_rowfrag(dob, p.ld_do, qrow, n, ks, l); }
  const SoftCap sc_ = make_softcap(p.softcap, p.sc_plan);
  // compiler-visible loads are consumed before the counted DMA waits (see attn_fwd_pipe_kernel)
  float lse2_ = lse2, dlt_ = dlt;
  asm volatile("" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3]), "+v"(dof[0]), "+v"(dof[1]), "+v"(dof[2]), "+v"(dof[3]), "+v"(lse2_), "+v"(dlt_));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  f32x16 dq[2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) dq[i][r] = 0.f;
  bool done = false;
  {
    if (sc_.mode <= 1) {                                           // (scalar branch: the layer's plan)
      const int wu = __builtin_amdgcn_readfirstlane(w);
      const uint32_t lds0 = (uint32_t)(size_t)(lds_void_t*)&ring[0][0];
      uint32_t ka[4], ta[2], tb[2], dk[2], dv[2];
#pragma unroll
      for (int ks = 0; ks < 4; ks++) ka[ks] = lds0 + 2u * (uint32_t)((l & 31) * 64 + (((2 * ks + hi) ^ swz_f(l & 31)) << 3));      // dma_rowfrag, slot 0, key block 0
      {
        const int q = l & 15, ra = 4 * hi + (q >> 2), rb = ra + 8;                                                               // dma_tr8, slot 0, rowA = 4 hi
#pragma unroll
        for (int db = 0; db < 2; db++) {
          const int col = db * 32 + 16 * ((l >> 4) & 1) + 4 * (q & 3);
          ta[db] = lds0 + 2u * (uint32_t)(ra * 64 + (((col >> 3) ^ swz_f(ra)) << 3) + (col & 7));
          tb[db] = lds0 + 2u * (uint32_t)(rb * 64 + (((col >> 3) ^ swz_f(rb)) << 3) + (col & 7));
        }
      }
#pragma unroll
      for (int jp = 0; jp < 2; jp++) {                                                                                           // tile_dma's pieces as buffer offsets
        const int r = w * 16 + jp * 8 + (l >> 3), c = (l & 7) ^ swz_f(r);
        dk[jp] = 2u * (uint32_t)(r * p.ld_k + c * 8);
        dv[jp] = 2u * (uint32_t)(r * p.ld_v + c * 8);
      }
      // raw buffer resources over the sample's rows: a row past the end reads as zeros (its keys are masked)
      const uint64_t baseK = (uint64_t)(uintptr_t)kb_, baseV = (uint64_t)(uintptr_t)vb;
      u32x4 rsK, rsV;
      rsK[0] = __builtin_amdgcn_readfirstlane((uint32_t)baseK); rsK[1] = __builtin_amdgcn_readfirstlane((uint32_t)(baseK >> 32) & 0xffffu);
      rsK[2] = __builtin_amdgcn_readfirstlane((uint32_t)(n - 1) * (uint32_t)p.ld_k * 2u + 128u); rsK[3] = 0x00020000u;
      rsV[0] = __builtin_amdgcn_readfirstlane((uint32_t)baseV); rsV[1] = __builtin_amdgcn_readfirstlane((uint32_t)(baseV >> 32) & 0xffffu);
      rsV[2] = __builtin_amdgcn_readfirstlane((uint32_t)(n - 1) * (uint32_t)p.ld_v * 2u + 128u); rsV[3] = 0x00020000u;
      const uint32_t stk = __builtin_amdgcn_readfirstlane(128u * (uint32_t)p.ld_k), stv = __builtin_amdgcn_readfirstlane(128u * (uint32_t)p.ld_v);
      uint32_t sko = 2u * stk, svo = 2u * stv, cnt = (uint32_t)__builtin_amdgcn_readfirstlane(nt), su1 = 32u, live = 1u, stmp;
      const int kve_max = wave_max_i(kve);                         // the wave's last visible key + 1: units from there on are skipped (generator: `live`)
      int rem2 = __builtin_amdgcn_readfirstlane(nt - 2);
      const uint32_t mk = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)wu * 2048u);
      const int kmaskp = kve - 4 * hi + 32;
      const float p1 = sc_.p1, p3 = sc_.p3, p5 = sc_.p5, d1 = sc_.d1, d3 = sc_.d3, d5 = sc_.d5;
      u32x4 q0f = __builtin_bit_cast(u32x4, qf[0]), q1f = __builtin_bit_cast(u32x4, qf[1]), q2f = __builtin_bit_cast(u32x4, qf[2]), q3f = __builtin_bit_cast(u32x4, qf[3]);
      u32x4 d0f = __builtin_bit_cast(u32x4, dof[0]), d1f = __builtin_bit_cast(u32x4, dof[1]), d2f = __builtin_bit_cast(u32x4, dof[2]), d3f = __builtin_bit_cast(u32x4, dof[3]);
      // in-outs are EARLY-CLOBBER (see the forward's statement): sko / svo start out equal to multiples of the inputs stk / stv
#define DQ_OPERANDS                                                                                                                                  \
          : [dq0] "+&v"(dq[0]), [dq1] "+&v"(dq[1]), [sko] "+&s"(sko), [svo] "+&s"(svo), [cnt] "+&s"(cnt), [rem2] "+&s"(rem2), [su1] "+&s"(su1),           \
            [live] "+&s"(live), [stmp] "=&s"(stmp)                                                                                                   \
          : [qf0] "v"(q0f), [qf1] "v"(q1f), [qf2] "v"(q2f), [qf3] "v"(q3f), [df0] "v"(d0f), [df1] "v"(d1f), [df2] "v"(d2f), [df3] "v"(d3f),          \
            [ka0] "v"(ka[0]), [ka1] "v"(ka[1]), [ka2] "v"(ka[2]), [ka3] "v"(ka[3]), [ta0] "v"(ta[0]), [ta1] "v"(ta[1]), [tb0] "v"(tb[0]), [tb1] "v"(tb[1]), \
            [dk0] "v"(dk[0]), [dk1] "v"(dk[1]), [dv0] "v"(dv[0]), [dv1] "v"(dv[1]), [rsk] "s"(rsK), [rsv] "s"(rsV), [stk] "s"(stk), [stv] "s"(stv),  \
            [mk] "s"(mk), [kvemin] "s"(kve_min), [kvemax] "s"(kve_max), [p1] "v"(p1), [d1] "v"(d1), [p3] DQ_C3(p3), [d3] DQ_C3(d3), [p5] "s"(p5), [d5] "s"(d5),            \
            [lse2] "v"(lse2_), [dlt] "v"(dlt_), [kmaskp] "v"(kmaskp)                                                                                 \
          : "memory", "scc", "vcc", TFX_DQ_CLOBBERS
      if (sc_.mode == 0) {
#define DQ_C3 "s"
        asm volatile(
#include "attn_dq_loop_m0.inc"
            DQ_OPERANDS);
#undef DQ_C3
      } else {
#define DQ_C3 "v"
        asm volatile(
#include "attn_dq_loop_m1.inc"
            DQ_OPERANDS);
#undef DQ_C3
      }
#undef DQ_OPERANDS
      done = true;
    }
  }
  if (!done) dq_pipe_loop<2>(p, Ks, Vs, kb_, vb, n, nt, qf, dof, dq, lse2_, dlt_, sc_, kve, kve_min, tx);
  asm volatile("s_nop 15\n\ts_nop 15" : "+v"(dq[0]), "+v"(dq[1]));                // the asm MFMAs' results are read by vector code from here on
  {
    bf16x4 ov[2][4];
#pragma unroll
    for (int db = 0; db < 2; db++)
#pragma unroll
      for (int rg = 0; rg < 4; rg++)
#pragma unroll
        for (int e = 0; e < 4; e++) ov[db][rg][e] = f2bf(dq[db][rg * 4 + e]);
    __syncthreads();                                            // the K / V tiles are free: staging area of the coalesced stores
    bf16* st = &ring[0][0] + w * 2048;
    if (p.nr_qkv) {                                             // (kernel argument: block-uniform) QK-norm + RoPE backward on the way out
      __shared__ float sg[4 * 64];
      float pg[8];
#pragma unroll
      for (int e = 0; e < 8; e++) pg[e] = 0.f;
      wave_block_store_nr<0>(st, ov, p, tok0 + q0 + w * 32, h * DH, n - (q0 + w * 32), pg, l);
      nr_flush_dgamma(pg, sg, p.nr_dgamma_q, tx);
    } else
    wave_block_store(st, ov, p.dq + (tok0 + q0 + w * 32) * p.ld_dq + h * DH, p.ld_dq, n - (q0 + w * 32), l);
  }
}

// dK/dV kernel: from s = soft-capped scores (log2 domain) and dp = dP of one 32-query x 32-key block: pr = P (masked when MASK), s <- dS_raw.
// Streams register pairs (the kernel sits at the 256-VGPR limit); the mask test is hoisted into the template parameter so that interior
// tiles carry no compare / select at all.
// MODE 0 / 1: s holds the RAW scores - the layer's cubic / quintic (tfx.h soft-cap plan) and its derivative are formed here from u = s^2;
// MODE 2: s was soft-capped by softcap16 (degree by the wave's scores), the derivative is 1 - (s2 / cap2)^2.
template <bool MASK, int MODE>
TFX_DEV void dkv_scores(f32x16& s, const f32x16& dp, f32x16& pr, const float* s_lse, const float* s_dlt, const int* s_kve, int ql0, int krow, float g2,
                        const SoftCap& c) {
  // single-instruction fp32 forms throughout (round 4): the packed v_pk_* forms this function used since round 1 issue in 6.5 clocks per wave
  // against 2 x 2.7 for the pair of scalar instructions they replace (tools/valu_probe.hip)
#pragma unroll
  for (int rg = 0; rg < 4; rg++) {
    const int ql = ql0 + 8 * rg;
    const f32x4 ls4 = *(const f32x4*)(s_lse + ql), dl4 = *(const f32x4*)(s_dlt + ql);
    const int* kv4 = s_kve + ql;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int r = rg * 4 + e;
      const float a = s[r];
      float arg, dth;
      if constexpr (MODE == 0) {
        const float u = a * a;
        dth = fmaf(u, c.d3, c.d1);
        arg = fmaf(a, fmaf(u, c.p3, c.p1), -ls4[e]);
      } else if constexpr (MODE == 1) {
        const float u = a * a;
        dth = fmaf(u, fmaf(u, c.d5, c.d3), c.d1);
        arg = fmaf(a, fmaf(u, fmaf(u, c.p5, c.p3), c.p1), -ls4[e]);
      } else {
        dth = fmaf(a * a, -g2, 1.f);                                   // 1 - tanh^2 from the soft-capped score
        arg = a - ls4[e];
      }
      float pv = __builtin_amdgcn_exp2f(arg);
      if (MASK) pv = krow < kv4[e] ? pv : 0.f;
      pr[r] = pv;
      s[r] = pv * ((dp[r] - dl4[e]) * dth);                             // dS_raw[q][key]
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward dK/dV: block = 128 keys (4 waves x 32), loop over 64-query tiles that can see them
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(tfx_attn_args p) {
  __shared__ __attribute__((aligned(16))) bf16 Qs[64 * LDT];
  __shared__ __attribute__((aligned(16))) bf16 Ds[64 * LDT];
  __shared__ __attribute__((aligned(16))) float s_lse[64], s_dlt[64];
  __shared__ __attribute__((aligned(16))) int s_kve[64];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, hi = l >> 5;
  const int n = p.n;
  const BlockId bi = decode_block(p.order, (n + 127) / 128, false);
  const int h = bi.h, b = bi.b, k0 = bi.tile * 128;
  const size_t tok0 = (size_t)b * n;
  const bf16* qb = p.q + tok0 * p.ld_q + h * DH;
  const bf16* kb_ = p.k + tok0 * p.ld_k + h * DH;
  const bf16* vb = p.v + tok0 * p.ld_v + h * DH;
  const bf16* dob = p.do_eff + tok0 * p.ld_do + h * DH;
  const float* lseb = p.lse + ((size_t)b * p.h + h) * n;
  const float* dltb = p.delta + ((size_t)b * p.h + h) * n;
  const int krow = k0 + w * 32 + (l & 31);                 // this lane's key
  const int qt0 = p.q_start[tok0 + min(k0, n - 1)] / 64;   // q_start is non-decreasing in the key index
  const int qt1 = (n + 63) / 64;

  bf16x8 kf[4], vf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ks++) { kf[ks] = g_rowfrag(kb_, p.ld_k, krow, n, ks); vf[ks] = g_rowfrag(vb, p.ld_v, krow, n, ks); }
  f32x16 dk[2], dv[2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) { dk[i][r] = 0.f; dv[i][r] = 0.f; }
  const SoftCap sc_ = make_softcap(p.softcap, p.sc_plan);
  const int kw_last = k0 + w * 32 + 31;                    // last key of this wave's 32-key block

  TileRegs qr, dr;
  // per-query statistics of the tile (lse, delta, kv_end) travel with it: fetched one tile AHEAD into registers of the first wave, like the
  // Q / dO tiles - a load issued between the two barriers would put a global round trip on every tile's critical path for all four waves
  float st_l = 0.f, st_d = 0.f; int st_k = 0;
  auto stats_gload = [&](int jt) {
    if (threadIdx.x < 64) {
      const int qi = jt * 64 + threadIdx.x;
      const int qcl = min(qi, n - 1);
      st_l = lseb[qcl]; st_d = dltb[qcl];
      st_k = qi < n ? p.kv_end[tok0 + qcl] : 0;                     // rows past the end see nothing
    }
  };
  tile_gload(qr, qb, p.ld_q, qt0 * 64, n);
  tile_gload(dr, dob, p.ld_do, qt0 * 64, n);
  stats_gload(qt0);
  for (int jt = qt0; jt < qt1; jt++) {
    __syncthreads();
    tile_sstore(qr, Qs); tile_sstore(dr, Ds);
    if (threadIdx.x < 64) { s_lse[threadIdx.x] = st_l * LOG2E; s_dlt[threadIdx.x] = st_d; s_kve[threadIdx.x] = st_k; }
    __syncthreads();
    if (jt + 1 < qt1) { tile_gload(qr, qb, p.ld_q, (jt + 1) * 64, n); tile_gload(dr, dob, p.ld_do, (jt + 1) * 64, n); stats_gload(jt + 1); }
#pragma unroll
    for (int qb2 = 0; qb2 < 2; qb2++) {
      // (round 6) a 32-query block none of whose rows sees any of this wave's 32 keys contributes exact zeros: skipped per wave (wave-uniform branch; kv_end is
      // non-decreasing over a sample's real rows, rows past the end carry 0: the block's largest kv_end sits on its last real row).  The diagonal tiles of a
      // 128-key block hold 6 such (wave, block) pairs of 16 - 8 % of the kernel's units at n = 1024, and the kernel is power-bound: work not done is time
      {
        const int lastr = min(31, n - 1 - (jt * 64 + qb2 * 32));
        if (__builtin_amdgcn_readfirstlane((int)(lastr < 0 || s_kve[qb2 * 32 + max(lastr, 0)] <= k0 + w * 32)) != 0) continue;
      }
      f32x16 s, dp;                                                // (no spare registers here for a shared zero accumulator)
#pragma unroll
      for (int r = 0; r < 16; r++) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < 4; ks++) {
        s = MFMA(lds_rowfrag(Qs, qb2 * 32, ks), kf[ks], s);       // S[q][key]
        dp = MFMA(lds_rowfrag(Ds, qb2 * 32, ks), vf[ks], dp);     // dP[q][key]
      }
      if (sc_.mode == 2) softcap16(s, sc_);                          // s = s2 = cap*log2e*tanh(s/cap); modes 0 / 1: inside dkv_scores
      // kv_end is non-decreasing over the real queries: the 32-query block is fully visible to this wave's keys
      // iff its first query sees the wave's last key and the block holds no rows past the end (wave-uniform)
      const bool need_mask = __builtin_amdgcn_readfirstlane((int)(kw_last >= s_kve[qb2 * 32] || jt * 64 + qb2 * 32 + 31 >= n)) != 0;
      f32x16 pr;
      const int ql0 = qb2 * 32 + 4 * hi;
      if (sc_.mode == 0) {                                           // scalar branches: one variant runs per 32-query block
        if (need_mask) dkv_scores<true, 0>(s, dp, pr, s_lse, s_dlt, s_kve, ql0, krow, sc_.g2, sc_);
        else dkv_scores<false, 0>(s, dp, pr, s_lse, s_dlt, s_kve, ql0, krow, sc_.g2, sc_);
      } else if (sc_.mode == 1) {
        if (need_mask) dkv_scores<true, 1>(s, dp, pr, s_lse, s_dlt, s_kve, ql0, krow, sc_.g2, sc_);
        else dkv_scores<false, 1>(s, dp, pr, s_lse, s_dlt, s_kve, ql0, krow, sc_.g2, sc_);
      } else {
        if (need_mask) dkv_scores<true, 2>(s, dp, pr, s_lse, s_dlt, s_kve, ql0, krow, sc_.g2, sc_);
        else dkv_scores<false, 2>(s, dp, pr, s_lse, s_dlt, s_kve, ql0, krow, sc_.g2, sc_);
      }
#pragma unroll
      for (int tt = 0; tt < 2; tt++) {
        const bf16x8 pf = pack8(pr, tt), dsf = pack8(s, tt);
        const int ra = qb2 * 32 + 16 * tt + 4 * hi;
#pragma unroll
        for (int db = 0; db < 2; db++) {
          dv[db] = MFMA(lds_tr8(Ds, LDT, ra, ra + 8, db * 32), pf, dv[db]);    // dV^T[d][key]
          dk[db] = MFMA(lds_tr8(Qs, LDT, ra, ra + 8, db * 32), dsf, dk[db]);   // dK^T[d][key]
        }
      }
    }
  }
  {
    bf16x4 kv[2][4], vv[2][4];
#pragma unroll
    for (int db = 0; db < 2; db++)
#pragma unroll
      for (int rg = 0; rg < 4; rg++)
#pragma unroll
        for (int e = 0; e < 4; e++) { kv[db][rg][e] = f2bf(dk[db][rg * 4 + e]); vv[db][rg][e] = f2bf(dv[db][rg * 4 + e]); }
    __syncthreads();                                            // the Q / dO tiles are free: staging area of the coalesced stores
    bf16* st = (w < 2 ? Qs : Ds) + (w & 1) * 2048;
    const int rows = n - (k0 + w * 32);
    if (p.nr_qkv) {                                             // (kernel argument: block-uniform) QK-norm + RoPE backward of d k~ on the way out
      __shared__ float sg[4 * 64];
      float pg[8];
#pragma unroll
      for (int e = 0; e < 8; e++) pg[e] = 0.f;
      wave_block_store_nr<1>(st, kv, p, tok0 + k0 + w * 32, (p.h + h) * DH, rows, pg);
      wave_block_store(st, vv, p.dv + (tok0 + k0 + w * 32) * p.ld_dv + h * DH, p.ld_dv, rows);
      nr_flush_dgamma(pg, sg, p.nr_dgamma_k);
    } else {
      wave_block_store(st, kv, p.dk + (tok0 + k0 + w * 32) * p.ld_dk + h * DH, p.ld_dk, rows);
      wave_block_store(st, vv, p.dv + (tok0 + k0 + w * 32) * p.ld_dv + h * DH, p.ld_dv, rows);
    }
  }
}

static dim3 attn_grid(const tfx_attn_args& q) { return dim3(q.h, q.b, (q.n + 127) / 128); }
int attn_fwd(const tfx_attn_args& p, hipStream_t s) {
  if (p.n <= 0 || p.b <= 0 || p.h <= 0) return -1;
  if ((p.ld_q | p.ld_k | p.ld_v | p.ld_out) & 7) return -2;
  if (!(p.softcap > 0.f) || p.softcap * LOG2E > 96.f) return -3;     // fixed-reference softmax needs exp2(cap*log2e) finite in fp32 sums
  tfx_attn_args q = p; q.order = 1;
  // TFX_ATTN_ASM=1: the generated main loop for the unmasked tiles (bit-identical; OFF - measured round 6, profiles/r06_attn_fwd_asm_*: the loop's 88-instruction
  // vector stream is 27 us of the launch, but 76 us of the 134 are prologue, boundary tiles and epilogue, and at 179 registers two blocks share a CU where the
  // hipcc form (165) fits three: 145 against 134 us per launch in the step)
  static int use_asm = -1;
  if (use_asm < 0) { const char* e = getenv("TFX_ATTN_ASM"); use_asm = (e && e[0] == '1') ? 1 : 0; }
  if (use_asm && p.n_kv == 0 && !p.q_cnt && !p.q_row0 && p.sc_plan) hipLaunchKernelGGL(attn_fwd_pipe_kernel<true>, attn_grid(q), dim3(256), 0, s, q);
  else hipLaunchKernelGGL(attn_fwd_pipe_kernel<false>, attn_grid(q), dim3(256), 0, s, q);
  return (int)hipGetLastError();
}
int attn_bwd(const tfx_attn_args& p, hipStream_t s) {
  if (p.n <= 0 || p.b <= 0 || p.h <= 0) return -1;
  if ((p.ld_q | p.ld_k | p.ld_v | p.ld_out | p.ld_dout | p.ld_do | p.ld_dq | p.ld_dk | p.ld_dv) & 7) return -2;
  if (!(p.softcap > 0.f) || p.softcap * LOG2E > 96.f) return -3;
  long long nthreads = (long long)p.b * p.n * p.h * 8;
  if (nthreads >= (1ll << 31)) return -4;
  if (p.nr_qkv) {                                               // fused QK-norm / RoPE backward: complete argument set, 16-byte rows
    if (!p.nr_dqkv || !p.nr_gamma_q || !p.nr_gamma_k || !p.nr_rot_pos || !p.nr_cos || !p.nr_sin || !p.nr_dgamma_q || !p.nr_dgamma_k) return -5;
    if (((p.nr_ld_qkv | p.nr_ld_dqkv) & 7) || (((uintptr_t)p.nr_qkv | (uintptr_t)p.nr_dqkv) & 15)) return -5;
  }
  tfx_attn_args q = p; q.order = 1;
  // TFX_ATTN_BWD_PIPE=0: the plain-loop dQ kernel (A/B; bit-identical results)
  static int pipe = -1;
  if (pipe < 0) { const char* e = getenv("TFX_ATTN_BWD_PIPE"); pipe = (e && e[0] == '0') ? 0 : 1; }
  hipLaunchKernelGGL(attn_bwd_prep_kernel, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, s, q);
  hipLaunchKernelGGL(attn_bwd_dkv_kernel, attn_grid(q), dim3(256), 0, s, q);
  if (pipe) hipLaunchKernelGGL(attn_bwd_dq_pipe_kernel, attn_grid(q), dim3(256), 0, s, q);
  else hipLaunchKernelGGL(attn_bwd_dq_kernel, attn_grid(q), dim3(256), 0, s, q);
  return (int)hipGetLastError();
}

}  // namespace tfx
