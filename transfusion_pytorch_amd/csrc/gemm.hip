// bf16 MFMA GEMMs for gfx950 (CDNA4), fp32 accumulate.
//
//   gemm_nt : C[M,N] = A[M,K] . B[N,K]^T   (both operands K-contiguous; forward + dX GEMMs)
//   gemm_tn : C[N,K] += A[M,N]^T . B[M,K]  (contraction over rows; weight-gradient GEMMs, split-M + fp32 atomics)
//
// Tile 128x128, 256 threads = 4 waves (2x2), wave tile 64x64 = 2x2 v_mfma_f32_32x32x16_bf16.
// Operands are computed SWAPPED (D' = B.A^T) so that a lane owns one output row and 4-element runs of
// consecutive columns: the epilogue stores 8-byte bf16x4 / 16-byte f32x4 vectors and per-row side data
// (row scatter map) is one lookup per lane.
#include "tfx_common.h"
#include "tfx_kernels.h"
#include <type_traits>
#include <cmath>
#include <cstring>
#include <vector>
#include <algorithm>
#include <mutex>

namespace tfx {

constexpr int BM = 128, BN = 128, BK = 64;

// ------------------------------------------------------------------------------------------------
// epilogues.  `v` = 4 consecutive columns n..n+3 of output row m (already bounds-checked for m).
// ------------------------------------------------------------------------------------------------
template <int EPI> struct Epilogue;

TFX_DEV void store_bf16x4(bf16* dst, f32x4 v) {
  bf16x4 o; o[0] = f2bf(v[0]); o[1] = f2bf(v[1]); o[2] = f2bf(v[2]); o[3] = f2bf(v[3]);
  *(bf16x4*)dst = o;
}

TFX_DEV f32x4 add_bias(const GemmNT& p, int n, f32x4 v) {
  if (p.bias) { f32x4 b = *(const f32x4*)(p.bias + n); v += b; }
  return v;
}

template <> struct Epilogue<EPI_BF16> {      // C(bf16) = acc + bias
  static TFX_DEV void apply(const GemmNT& p, int m, int mo, int n, f32x4 v) {
    v = add_bias(p, n, v);
    bf16* c = (bf16*)p.C + (size_t)mo * p.ldc + n;
    if (n + 3 < p.N) store_bf16x4(c, v);
    else for (int e = 0; e < 4; e++) if (n + e < p.N) c[e] = f2bf(v[e]);
  }
};
template <> struct Epilogue<EPI_F32> {       // C(fp32) = acc + bias
  static TFX_DEV void apply(const GemmNT& p, int m, int mo, int n, f32x4 v) {
    v = add_bias(p, n, v);
    float* c = (float*)p.C + (size_t)mo * p.ldc + n;
    if (n + 3 < p.N) *(f32x4*)c = v;
    else for (int e = 0; e < 4; e++) if (n + e < p.N) c[e] = v[e];
  }
};
template <> struct Epilogue<EPI_SILU> {      // C2(bf16) = pre = acc + bias ; C(bf16) = silu(pre)
  static TFX_DEV void apply(const GemmNT& p, int m, int mo, int n, f32x4 v) {
    v = add_bias(p, n, v);
    f32x4 s; for (int e = 0; e < 4; e++) s[e] = v[e] * sigmoidf_(v[e]);
    bf16* c = (bf16*)p.C + (size_t)mo * p.ldc + n;
    bf16* c2 = (bf16*)p.C2 + (size_t)mo * p.ldc2 + n;
    if (n + 3 < p.N) { store_bf16x4(c, s); store_bf16x4(c2, v); }
    else for (int e = 0; e < 4; e++) if (n + e < p.N) { c[e] = f2bf(s[e]); c2[e] = f2bf(v[e]); }
  }
};
template <> struct Epilogue<EPI_RESID> {     // C(bf16) = acc + bias + R
  static TFX_DEV void apply(const GemmNT& p, int m, int mo, int n, f32x4 v) {
    v = add_bias(p, n, v);
    const bf16* r = p.R + (size_t)(p.resid_mapped ? mo : m) * p.ldr + n;
    bf16* c = (bf16*)p.C + (size_t)mo * p.ldc + n;
    if (n + 3 < p.N) {
      bf16x4 rv = *(const bf16x4*)r;
      for (int e = 0; e < 4; e++) v[e] += bf2f(rv[e]);
      store_bf16x4(c, v);
    } else for (int e = 0; e < 4; e++) if (n + e < p.N) c[e] = f2bf(v[e] + bf2f(r[e]));
  }
};

// GEGLU pair epilogues work on the wave's two 32-column halves (value half / gate half of one 64-column
// interleave block, see tfx_kernels.h "GEGLU layout").
struct GegluFwd {   // C(bf16, ld 2*dip) = [u|v] = [gelu(g) | a gelu'(g)] (a, g = pre-activation + bias; geglu_uvh) ; C2(bf16, ld dip) = a * gelu(g)
  static TFX_DEV void apply(const GemmNT& p, int m, int n_a, f32x4 a, f32x4 g) {
    // n_a = physical column of the value half inside the interleaved layout (multiple of 4, (n_a % 64) < 32)
    if (p.bias) { a += *(const f32x4*)(p.bias + n_a); g += *(const f32x4*)(p.bias + n_a + 32); }
    bf16* c = (bf16*)p.C + (size_t)m * p.ldc + n_a;
    f32x4 u, v, h; for (int e = 0; e < 4; e++) { const GegluUVH t_ = geglu_uvh(a[e], g[e]); u[e] = t_.u; v[e] = t_.v; h[e] = t_.h; }
    store_bf16x4(c, u); store_bf16x4(c + 32, v);
    int feat = (n_a >> 6) * 32 + (n_a & 31);
    store_bf16x4((bf16*)p.C2 + (size_t)m * p.ldc2 + feat, h);
  }
};

// dh -> d[a|g]:  acc = dh[m][feat..feat+3];  aux = the forward's saved [u|v] = [gelu(g) | a gelu'(g)] (ld 2*dip):  da = dh u, dg = dh v
template <> struct Epilogue<EPI_GEGLU_BWD> {
  static TFX_DEV void apply(const GemmNT& p, int m, int mo, int n, f32x4 dh) {
    if (n >= p.N) return;                       // N (= dip) is a multiple of 64
    int col = (n >> 5) * 64 + (n & 31);
    const bf16* ag = p.aux + (size_t)m * p.ldaux + col;
    bf16x4 u4 = *(const bf16x4*)ag, v4 = *(const bf16x4*)(ag + 32);
    f32x4 da, dg;
    for (int e = 0; e < 4; e++) { da[e] = dh[e] * bf2f(u4[e]); dg[e] = dh[e] * bf2f(v4[e]); }
    bf16* c = (bf16*)p.C + (size_t)mo * p.ldc + col;
    store_bf16x4(c, da); store_bf16x4(c + 32, dg);
  }
};

// ------------------------------------------------------------------------------------------------
// Pipelined epilogue for the LDS-DMA kernels (N % 4 == 0).  gfx950 has ONE in-order vmcnt for loads and stores, so an
// epilogue that loads side data (row map, bias, residual, saved activations) between its stores drains every store
// issued so far at each load's wait - a chain of memory round trips that cost ~12 us per 256x256 tile (the "fixed" part
// of the K=512 GEMMs).  Here every load of 32-row block i+1 is issued BEFORE the stores of block i, the row map and the
// bias are fetched once up front, and column guards are plain exec masks: no wait ever follows a store.
// ------------------------------------------------------------------------------------------------
template <int EPI> struct EpiIn { };
template <> struct EpiIn<EPI_RESID> { bf16x4 r[2][4]; };
template <> struct EpiIn<EPI_GEGLU_BWD> { bf16x4 a[2][4], g[2][4]; };

// (lane: the persistent kernel hands in an opaque copy of the lane id, so that hipcc does not hoist the lane-derived values out of its tile loop - and spill them)
template <int EPI, int NI>
TFX_DEV void fast_epilogue(const GemmNT& p, f32x16 (&acc)[NI][2], int m_w, int n_w, int lane = -1) {
  const int l = lane >= 0 ? lane : (int)(threadIdx.x & 63), hi = l >> 5, r = l & 31;
  int mo[NI];
#pragma unroll
  for (int i = 0; i < NI; i++) {
    const int m = m_w + i * 32 + r;
    mo[i] = m < p.M ? (p.rowmap ? p.rowmap[m] : m) : -1;
  }
  // ONE branch around all eight loads: with a branch per load hipcc waits (`s_waitcnt vmcnt(0)`) at every join - eight serial round trips to L2
  f32x4 bias[2][4];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 2; j++)
#pragma unroll
    for (int g = 0; g < 4; g++) bias[j][g] = zero4;
  if constexpr (EPI != EPI_GEGLU_BWD) {
    if (p.bias) {
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int g = 0; g < 4; g++) bias[j][g] = *(const f32x4*)(p.bias + min(n_w + j * 32 + 8 * g + 4 * hi, p.N - 4));
    }
  }
  // side-data loads are unconditional (addresses clamped into range): a load under a divergent branch makes the
  // compiler fall back to vmcnt(0) at the join, which would drain the stores again
  auto load_in = [&](EpiIn<EPI>& in, int i) {
    const int m = min(m_w + i * 32 + r, p.M - 1);
    if constexpr (EPI == EPI_RESID) {
      const bf16* rp = p.R + (size_t)(p.resid_mapped ? max(mo[i], 0) : m) * p.ldr;
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int g = 0; g < 4; g++) in.r[j][g] = *(const bf16x4*)(rp + min(n_w + j * 32 + 8 * g + 4 * hi, p.N - 4));
    } else if constexpr (EPI == EPI_GEGLU_BWD) {
      const bf16* ap = p.aux + (size_t)m * p.ldaux + 4 * hi;
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int blk = min((n_w >> 5) + j, (p.N >> 5) - 1);
#pragma unroll
        for (int g = 0; g < 4; g++) { in.a[j][g] = *(const bf16x4*)(ap + blk * 64 + 8 * g); in.g[j][g] = *(const bf16x4*)(ap + blk * 64 + 32 + 8 * g); }
      }
    }
  };
  auto emit = [&](const EpiIn<EPI>& in, int i) {
    if (mo[i] < 0) return;
    if constexpr (EPI == EPI_GEGLU) {
      bf16* c = (bf16*)p.C + (size_t)mo[i] * p.ldc + n_w + 4 * hi;
      bf16* c2 = (bf16*)p.C2 + (size_t)mo[i] * p.ldc2 + (n_w >> 6) * 32 + 4 * hi;
#pragma unroll
      for (int g = 0; g < 4; g++) {
        if (n_w + 8 * g + 4 * hi >= p.N) continue;
        f32x4 u, v, h;
#pragma unroll
        for (int e = 0; e < 4; e++) { const GegluUVH t_ = geglu_uvh(acc[i][0][4 * g + e] + bias[0][g][e], acc[i][1][4 * g + e] + bias[1][g][e]); u[e] = t_.u; v[e] = t_.v; h[e] = t_.h; }
        store_bf16x4(c + 8 * g, u); store_bf16x4(c + 32 + 8 * g, v); store_bf16x4(c2 + 8 * g, h);
      }
    } else if constexpr (EPI == EPI_GEGLU_BWD) {
      bf16* c = (bf16*)p.C + (size_t)mo[i] * p.ldc + (n_w >> 5) * 64 + 4 * hi;
#pragma unroll
      for (int j = 0; j < 2; j++) {
        if (n_w + j * 32 >= p.N) continue;                   // N (= dip) is a multiple of 64
#pragma unroll
        for (int g = 0; g < 4; g++) {
          f32x4 da, dg;
#pragma unroll
          for (int e = 0; e < 4; e++) {                        // in.a / in.g hold the saved u = gelu(g) / v = a gelu'(g)
            const float dh = acc[i][j][4 * g + e];
            da[e] = dh * bf2f(in.a[j][g][e]);
            dg[e] = dh * bf2f(in.g[j][g][e]);
          }
          store_bf16x4(c + j * 64 + 8 * g, da); store_bf16x4(c + j * 64 + 32 + 8 * g, dg);
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const int n = n_w + j * 32 + 8 * g + 4 * hi;
          if (n >= p.N) continue;
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] = acc[i][j][4 * g + e] + bias[j][g][e];
          if constexpr (EPI == EPI_BF16) {
            store_bf16x4((bf16*)p.C + (size_t)mo[i] * p.ldc + n, v);
          } else if constexpr (EPI == EPI_F32) {
            *(f32x4*)((float*)p.C + (size_t)mo[i] * p.ldc + n) = v;
          } else if constexpr (EPI == EPI_SILU) {
            f32x4 s;
#pragma unroll
            for (int e = 0; e < 4; e++) s[e] = v[e] * sigmoidf_(v[e]);
            store_bf16x4((bf16*)p.C + (size_t)mo[i] * p.ldc + n, s);
            store_bf16x4((bf16*)p.C2 + (size_t)mo[i] * p.ldc2 + n, v);
          } else if constexpr (EPI == EPI_RESID) {
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] += bf2f(in.r[j][g][e]);
            store_bf16x4((bf16*)p.C + (size_t)mo[i] * p.ldc + n, v);
          }
        }
    }
  };
  EpiIn<EPI> in[2];
  load_in(in[0], 0);
#pragma unroll
  for (int i = 0; i < NI; i++) {
    if (i + 1 < NI) load_in(in[(i + 1) & 1], i + 1);
    emit(in[i & 1], i);
  }
}

// ------------------------------------------------------------------------------------------------
// LDS-staged store of a wave's 32x64 bf16 block.  fast_epilogue's bf16x4 stores put 64 different rows (64 cache
// lines) behind every store instruction; s_memtime stamps of the 256x256 kernel show that issuing them takes ~10k
// cycles per tile - as long as 3.3 K-tiles of MFMA work - because the address coalescer walks the lines one by one.
// Staging the block through a wave-private 4 KiB LDS area (128-byte rows, chunk index XOR ((row >> 1) & 7): conflict-free
// for both the 8-byte writes and the 16-byte reads) turns them into 16-byte stores with 8 lanes per 128-byte row segment:
// 8 lines per instruction.  No block barrier: the area is wave-private and one wave's LDS operations execute in order.
// ------------------------------------------------------------------------------------------------
TFX_DEV void stage_put4(bf16* st, int row, int col, f32x4 v) {
  bf16x4 o; o[0] = f2bf(v[0]); o[1] = f2bf(v[1]); o[2] = f2bf(v[2]); o[3] = f2bf(v[3]);
  *(bf16x4*)(st + row * 64 + (((col >> 3) ^ ((row >> 1) & 7)) << 3) + (col & 7)) = o;
}
// C(bf16)[mo][n_w .. n_w+63] = acc + bias for the NI 32-row blocks of a wave; needs ldc % 8 == 0, N % 8 == 0, C 16-byte aligned
// PLAIN (no row map, no bias): no load at all - and therefore no wait: hipcc puts the `s_waitcnt vmcnt(0)` of the (conditional) side loads on the common path, where it
// drains whatever is in flight: the stores of a wave's previous 64-column half (one-wave-per-SIMD kernels: two calls per wave), the next tile's DMAs (persistent kernel)
template <int NI, bool PLAIN = false>
TFX_DEV void staged_epilogue_bf16(const GemmNT& p, f32x16 (&acc)[NI][2], int m_w, int n_w, bf16* st, int lane = -1) {
  const int l = lane >= 0 ? lane : (int)(threadIdx.x & 63), hi = l >> 5, r = l & 31, ch = l & 7;
  int mo[NI][4];                                               // output row of (block i, flush pass q): row q * 8 + (l >> 3)
#pragma unroll
  for (int i = 0; i < NI; i++)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int m = m_w + i * 32 + q * 8 + (l >> 3);
      if constexpr (PLAIN) mo[i][q] = m < p.M ? m : -1;
      else mo[i][q] = m < p.M ? (p.rowmap ? p.rowmap[m] : m) : -1;
    }
  f32x4 bias[2][4];                                            // one branch around the eight loads (see fast_epilogue)
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 2; j++)
#pragma unroll
    for (int g = 0; g < 4; g++) bias[j][g] = zero4;
  if constexpr (!PLAIN) {
    if (p.bias) {
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int g = 0; g < 4; g++) bias[j][g] = *(const f32x4*)(p.bias + min(n_w + j * 32 + 8 * g + 4 * hi, p.N - 4));
    }
  }
  const bool col_ok = n_w + ch * 8 < p.N;
#ifdef TFX_PP_TIMING
  unsigned long long* stamps = (unsigned long long*)p.aux + (size_t)blockIdx.x * 8;
  if (threadIdx.x == 0) stamps[5] = __builtin_readcyclecounter();
#endif
#pragma unroll
  for (int i = 0; i < NI; i++) {
    bf16* s = st + (i & 1) * 2048;                             // two areas: block i+1 is written while block i's stores drain
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = PLAIN ? acc[i][j][4 * g + e] : acc[i][j][4 * g + e] + bias[j][g][e];
        stage_put4(s, r, j * 32 + 8 * g + 4 * hi, v);
      }
    bf16x8 v[4];                                               // all four LDS reads ahead of the (branch-guarded) stores: one LDS latency per block, not four
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int row = q * 8 + (l >> 3);
      v[q] = *(const bf16x8*)(s + row * 64 + ((ch ^ ((row >> 1) & 7)) << 3));
    }
#pragma unroll
    for (int q = 0; q < 4; q++)
      if (col_ok && mo[i][q] >= 0) *(bf16x8*)((bf16*)p.C + (size_t)mo[i][q] * p.ldc + n_w + ch * 8) = v[q];
#ifdef TFX_PP_TIMING
    if (threadIdx.x == 0 && i < 2) stamps[6 + i] = __builtin_readcyclecounter();
#endif
  }
}
// C(bf16)[mo][n_w .. n_w+63] = acc + bias + R.  The residual block (32 rows x 64 columns) comes in the way the result goes out: 16-byte
// accesses, 8 lanes per 128-byte row segment, through the staging layout - written there by chunk, read back in fragment shape (the addresses
// stage_put4 writes).  Needs what staged_epilogue_bf16 needs plus ldr % 8 == 0 and a 16-byte aligned R.
template <int NI>
TFX_DEV void staged_epilogue_resid(const GemmNT& p, f32x16 (&acc)[NI][2], int m_w, int n_w, bf16* st) {
  const int l = threadIdx.x & 63, hi = l >> 5, r = l & 31, ch = l & 7;
  int mo[NI][4], mr[NI][4];                                    // output row / residual row of (block i, pass q): tile row q * 8 + (l >> 3)
#pragma unroll
  for (int i = 0; i < NI; i++)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int m = m_w + i * 32 + q * 8 + (l >> 3);
      mo[i][q] = m < p.M ? (p.rowmap ? p.rowmap[m] : m) : -1;
      mr[i][q] = p.resid_mapped ? max(mo[i][q], 0) : min(m, p.M - 1);
    }
  f32x4 bias[2][4];                                            // one branch around the eight loads (see fast_epilogue)
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 2; j++)
#pragma unroll
    for (int g = 0; g < 4; g++) bias[j][g] = zero4;
  if (p.bias) {
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int g = 0; g < 4; g++) bias[j][g] = *(const f32x4*)(p.bias + min(n_w + j * 32 + 8 * g + 4 * hi, p.N - 4));
  }
  const bool col_ok = n_w + ch * 8 < p.N;
  const int ncol = min(n_w + ch * 8, p.N - 8);
  bf16* sr = st + 4096;                                        // residual block; the two output areas keep st[0, 4096)
  bf16x8 pre[4];
  auto load_r = [&](int i) {
#pragma unroll
    for (int q = 0; q < 4; q++) pre[q] = *(const bf16x8*)(p.R + (size_t)mr[i][q] * p.ldr + ncol);
  };
  load_r(0);
#pragma unroll
  for (int i = 0; i < NI; i++) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int row = q * 8 + (l >> 3);
      *(bf16x8*)(sr + row * 64 + ((ch ^ ((row >> 1) & 7)) << 3)) = pre[q];
    }
    if (i + 1 < NI) load_r(i + 1);
    bf16* s = st + (i & 1) * 2048;
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int col = j * 32 + 8 * g + 4 * hi;
        const bf16x4 rv = *(const bf16x4*)(sr + r * 64 + (((col >> 3) ^ ((r >> 1) & 7)) << 3) + (col & 7));
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = acc[i][j][4 * g + e] + bias[j][g][e] + bf2f(rv[e]);
        stage_put4(s, r, col, v);
      }
    bf16x8 v[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int row = q * 8 + (l >> 3);
      v[q] = *(const bf16x8*)(s + row * 64 + ((ch ^ ((row >> 1) & 7)) << 3));
    }
#pragma unroll
    for (int q = 0; q < 4; q++)
      if (col_ok && mo[i][q] >= 0) *(bf16x8*)((bf16*)p.C + (size_t)mo[i][q] * p.ldc + n_w + ch * 8) = v[q];
  }
}
// 32x32 bf16 block (64-byte rows; chunk XOR ((row >> 2) & 3)) for the GEGLU product h = a * gelu(g)
TFX_DEV void stage_put4_32(bf16* st, int row, int col, f32x4 v) {
  bf16x4 o; o[0] = f2bf(v[0]); o[1] = f2bf(v[1]); o[2] = f2bf(v[2]); o[3] = f2bf(v[3]);
  *(bf16x4*)(st + row * 32 + (((col >> 3) ^ ((row >> 2) & 3)) << 3) + (col & 7)) = o;
}
// GEGLU forward / backward with staged stores.  Forward: per 32-row block one [a|g] block (32x64) and one h block (32x32).
// Backward: per 32-row block two [da|dg] blocks (one per 32 dh columns).  The saved-activation loads of the backward stay
// per-lane 8-byte loads, issued one block ahead of their use (see fast_epilogue).
// (Round 4 kept a second 32 KiB table here, indexed by the bits of the saved bf16 gate, for the backward epilogue; round 5 saves u = gelu(g) and
// v = a gelu'(g) in the forward instead - geglu_uvh, tfx_common.h - so the backward epilogue has no function left to evaluate.)
constexpr int GTAB_N = 4096;                                      // floats x 2: the forward grid's 32 KiB
// Forward: g is an fp32 accumulator, so the table is a uniform GRID over g (step 2^-7 on [-8, 8): 2048 entries x {gelu, h gelu', h^2 gelu''/2, pad} = 32 KiB)
// and gelu(g) is its second-order Taylor polynomial around the nearest node: |error| <= (2^-8)^3 |d3 gelu| / 6 < 1e-8, also relative to gelu near 0
// (node 0 carries 0.5 g + 0.399 g^2 exactly); beyond +-8 the clamped end nodes extrapolate linearly to g / to 0.  6 index + 3 arithmetic VALU slots
// against ~16 incl. v_exp / v_rcp for the polynomial erf.
TFX_DEV float gelu_grid(const float* gtab, float g) {
  const float x = g * 128.f;
  const float xr = __builtin_amdgcn_fmed3f(__builtin_rintf(x), -1024.f, 1023.f);
  const float d = x - xr;
  const f32x4 c = *(const f32x4*)(gtab + 4 * ((int)xr + 1024));
  return fmaf(fmaf(c[2], d, c[1]), d, c[0]);
}
// geglu_uvh on the grid: u = the polynomial above; gelu'(g) = its derivative (c1 + 2 c2 d) / h - first order around the node, |error| <= (2^-8)^2 max|d3 gelu| / 2
// < 7e-6, three orders below the bf16 rounding of v; beyond +-8 the end nodes give gelu' = 1 / 0.  3 VALU slots more than gelu_grid.
TFX_DEV GegluUVH geglu_uvh_grid(const float* gtab, float a, float g) {
  const float x = g * 128.f;
  const float xr = __builtin_amdgcn_fmed3f(__builtin_rintf(x), -1024.f, 1023.f);
  const float d = x - xr;
  const f32x4 c = *(const f32x4*)(gtab + 4 * ((int)xr + 1024));
  const float t = fmaf(c[2], d, c[1]);
  GegluUVH r; r.u = fmaf(t, d, c[0]);
  r.h = a * r.u;
  r.v = (a * 128.f) * fmaf(c[2], d, t);
  return r;
}
template <int EPI, int NI>
TFX_DEV void staged_epilogue_geglu(const GemmNT& p, f32x16 (&acc)[NI][2], int m_w, int n_w, bf16* st, const float* gtab = nullptr, bool use_tab = false) {
  const int l = threadIdx.x & 63, hi = l >> 5, r = l & 31, ch = l & 7;
  int mo[NI][4], mo2[NI][2];                                   // rows of the 8-lane-per-row / 4-lane-per-row flush passes
#pragma unroll
  for (int i = 0; i < NI; i++) {
#pragma unroll
    for (int q = 0; q < 4; q++) { const int m = m_w + i * 32 + q * 8 + (l >> 3); mo[i][q] = m < p.M ? (p.rowmap ? p.rowmap[m] : m) : -1; }
#pragma unroll
    for (int q = 0; q < 2; q++) { const int m = m_w + i * 32 + q * 16 + (l >> 2); mo2[i][q] = (EPI == EPI_GEGLU && m < p.M) ? (p.rowmap ? p.rowmap[m] : m) : -1; }
  }
  auto flush64 = [&](const bf16* s, int i, int n_base, int Nout) {
    bf16x8 v[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int row = q * 8 + (l >> 3);
      v[q] = *(const bf16x8*)(s + row * 64 + ((ch ^ ((row >> 1) & 7)) << 3));
    }
#pragma unroll
    for (int q = 0; q < 4; q++)
      if (n_base + ch * 8 < Nout && mo[i][q] >= 0) *(bf16x8*)((bf16*)p.C + (size_t)mo[i][q] * p.ldc + n_base + ch * 8) = v[q];
  };
  if constexpr (EPI == EPI_GEGLU) {
    f32x4 ba[4], bg[4];                                        // one branch around the eight loads (see fast_epilogue)
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 4; g++) { ba[g] = zero4; bg[g] = zero4; }
    if (p.bias) {
#pragma unroll
      for (int g = 0; g < 4; g++) { const int n_a = min(n_w + 8 * g + 4 * hi, p.N - 36); ba[g] = *(const f32x4*)(p.bias + n_a); bg[g] = *(const f32x4*)(p.bias + n_a + 32); }
    }
    const int c4 = l & 3;
#pragma unroll
    for (int i = 0; i < NI; i++) {
      bf16* s = st + (i & 1) * 3072;                           // [a|g] 2048 elements + h 1024 elements, double-buffered
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int nl = 8 * g + 4 * hi;
        f32x4 u, v, h;                                           // saved for the backward: u = gelu(g), v = a gelu'(g) (geglu_uvh)
        if (use_tab) {                                           // (block-uniform) table form, see geglu_uvh_grid
#pragma unroll
          for (int e = 0; e < 4; e++) { const GegluUVH t_ = geglu_uvh_grid(gtab, acc[i][0][4 * g + e] + ba[g][e], acc[i][1][4 * g + e] + bg[g][e]); u[e] = t_.u; v[e] = t_.v; h[e] = t_.h; }
        } else {
#pragma unroll
          for (int e = 0; e < 4; e++) { const GegluUVH t_ = geglu_uvh(acc[i][0][4 * g + e] + ba[g][e], acc[i][1][4 * g + e] + bg[g][e]); u[e] = t_.u; v[e] = t_.v; h[e] = t_.h; }
        }
        stage_put4(s, r, nl, u); stage_put4(s, r, 32 + nl, v); stage_put4_32(s + 2048, r, nl, h);
      }
      flush64(s, i, n_w, p.N);
      bf16x8 hv[2];
#pragma unroll
      for (int q = 0; q < 2; q++) {
        const int row = q * 16 + (l >> 2);
        hv[q] = *(const bf16x8*)(s + 2048 + row * 32 + ((c4 ^ ((row >> 2) & 3)) << 3));
      }
      const int feat = (n_w >> 6) * 32 + c4 * 8;
#pragma unroll
      for (int q = 0; q < 2; q++)
        if (n_w < p.N && mo2[i][q] >= 0) *(bf16x8*)((bf16*)p.C2 + (size_t)mo2[i][q] * p.ldc2 + feat) = hv[q];
    }
  } else {   // EPI_GEGLU_BWD
    // The saved [a|g] of a 32-row block - for the wave's 64 dh columns 128 contiguous bf16 per row - comes in with 16-byte loads, 16 lanes per row
    // (8 cache lines per instruction), passes through a wave-private LDS image (chunk index XOR (row & 15)) and is read back in fragment shape.
    // Round 2 loaded it as 8-byte per-lane fragments: 32 rows = 32 cache lines behind every load instruction, 512 line visits per block against 64
    // (the same address-coalescer walk that made the direct stores slow, see staged_epilogue_bf16).
    bf16* sag = st + 2048;                                       // [32][128]; the output staging block keeps st[0, 2048)
    const int cbase = min(n_w >> 5, (p.N >> 5) - 2) * 64;
    bf16x8 pre[8];
    auto load_aux = [&](int i) {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int q = l + 64 * k, row = q >> 4, cc = q & 15;
        const int m = min(m_w + i * 32 + row, p.M - 1);
        pre[k] = *(const bf16x8*)(p.aux + (size_t)m * p.ldaux + cbase + cc * 8);
      }
    };
    load_aux(0);
#pragma unroll
    for (int i = 0; i < NI; i++) {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int q = l + 64 * k, row = q >> 4, cc = q & 15;
        *(bf16x8*)(sag + row * 128 + ((cc ^ (row & 15)) << 3)) = pre[k];
      }
      if (i + 1 < NI) load_aux(i + 1);
#pragma unroll
      for (int j = 0; j < 2; j++) {
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const bf16x4 u4 = *(const bf16x4*)(sag + r * 128 + (((j * 8 + g) ^ (r & 15)) << 3) + 4 * hi);          // saved u = gelu(g)
          const bf16x4 v4 = *(const bf16x4*)(sag + r * 128 + (((j * 8 + 4 + g) ^ (r & 15)) << 3) + 4 * hi);      // saved v = a gelu'(g)
          f32x4 da, dg;
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const float dh = acc[i][j][4 * g + e];
            da[e] = dh * bf2f(u4[e]);
            dg[e] = dh * bf2f(v4[e]);
          }
          stage_put4(st, r, 8 * g + 4 * hi, da); stage_put4(st, r, 32 + 8 * g + 4 * hi, dg);
        }
        if (n_w + j * 32 < p.N) flush64(st, i, ((n_w >> 5) + j) * 64, 2 * p.N);
      }
    }
  }
}
// Fused QK-RMSNorm + RoPE behind the [q | k | v | gates] projection (TFX_EPI_QKV_NORM_ROPE; SURVEY K4, reference T:946-965).  A wave's 64 columns are ONE
// head, and the staged store already brings the tile back from LDS in the token-wise kernel's own shape - 8 lanes per row, 8 contiguous columns
// (4 rotary pairs) each - so the wave norms and rotates what it has just read back, with qk_norm_rope_fwd_k's arithmetic on the same bf16-rounded
// values (bit-identical output), and stores q~ / k~ next to the raw projection.  Side data: the rotary position of a row (requested two half-blocks
// ahead), its cos / sin row (one half-block ahead, always BEFORE the stores of the current half: loads and stores retire through one in-order
// counter, a load behind a store waits for the store's round trip).  v and gate tiles take the plain path.
// GUARD = false: the wave's 32 NI x 64 block lies wholly inside the matrix - no per-store predicate, hence no branches around the stores: hipcc's
// wait-count pass merges the two sides of such a branch pessimistically and ends up draining the queue (vmcnt(0)) in the middle of the pipeline.
template <int NI, bool GUARD>
TFX_DEV void staged_epilogue_qknr_(const GemmNT& p, f32x16 (&acc)[NI][2], int m_w, int n_w, bf16* st) {
  const int l = threadIdx.x & 63, hi = l >> 5, r = l & 31, ch = l & 7;
  const int hd = p.qk_heads * 64;
  const int which = n_w >= hd;
  const bool col_ok = n_w + ch * 8 < p.N;
  constexpr int NS = 2 * NI;                                       // half-blocks of 16 rows: step s = block s >> 1, row groups q = 2 (s & 1), 2 (s & 1) + 1
  auto row_of = [&](int s, int k) { return m_w + (s >> 1) * 32 + (2 * (s & 1) + k) * 8 + (l >> 3); };
  f32x4 g0 = {0.f, 0.f, 0.f, 0.f}, g1 = g0;
  float rs = 0.f;
  int pos_a[2] = {0, 0}, pos_b[2] = {0, 0};                        // positions of step s + 1 / s + 2
  f32x4 cs_n[2], sn_n[2];                                          // cos / sin rows of step s + 1
  auto load_pos = [&](int s, int (&pos)[2]) {
#pragma unroll
    for (int k = 0; k < 2; k++) pos[k] = p.qk_rot_pos[min(row_of(s, k), p.M - 1)];
  };
  // decode steps: the k~ rows also go to the KV cache (row qk_cache_pos[m]); the positions of all NS half-blocks are fetched up front with the
  // other side data (a load behind a store waits for the store's round trip)
  // (NI == 1 = the decode-step kernel's instantiation: the 256 x 256 kernel never appends - qknr_fusable - and has no registers to spare for the positions)
  const bool to_cache = NI == 1 && p.qk_cache != nullptr && which == 1;      // wave-uniform
  int cpos[NI == 1 ? NS : 1][2];
  if constexpr (NI == 1) {
#pragma unroll
    for (int s = 0; s < NS; s++)
#pragma unroll
      for (int k = 0; k < 2; k++) { const int m = row_of(s, k); cpos[s][k] = (to_cache && m < p.M) ? p.qk_cache_pos[m] : -1; }
  }
  auto load_cs = [&](const int (&pos)[2], f32x4 (&cs)[2], f32x4 (&sn)[2]) {
#pragma unroll
    for (int k = 0; k < 2; k++) {
      cs[k] = *(const f32x4*)(p.qk_cos + (size_t)pos[k] * 32 + ch * 4);
      sn[k] = *(const f32x4*)(p.qk_sin + (size_t)pos[k] * 32 + ch * 4);
    }
  };
  {
    const float* gm = (which ? p.qk_gamma_k : p.qk_gamma_q) + ch * 8;
    g0 = *(const f32x4*)gm; g1 = *(const f32x4*)(gm + 4);
    rs = (p.qk_norm_scale > 0.f ? p.qk_norm_scale : 8.f);
    load_pos(0, pos_a);
    load_cs(pos_a, cs_n, sn_n);                                    // step 0's rows
    load_pos(1, pos_a);                                            // step 1's positions
    if (NS > 2) load_pos(2, pos_b);
    asm volatile("" ::: "memory");
  }
  bf16x8 v[4];
#pragma unroll
  for (int s = 0; s < NS; s++) {
    const int i = s >> 1, qh = s & 1;
    f32x4 cs[2] = {cs_n[0], cs_n[1]}, sn[2] = {sn_n[0], sn_n[1]};
    {
      if (s + 1 < NS) load_cs(pos_a, cs_n, sn_n);                  // next half-block's rows, ahead of this half's stores
      pos_a[0] = pos_b[0]; pos_a[1] = pos_b[1];
      if (s + 3 < NS) load_pos(s + 3, pos_b);
      // compiler fence: hipcc otherwise sinks these loads to their first use, BEHIND this half's stores - and the wait for a load then
      // drains every older store (one in-order counter): a store round trip per half-block (seen in the ISA: vmcnt waits right behind the loads)
      asm volatile("" ::: "memory");
    }
    if (qh == 0) {                                                 // stage the 32-row block, read it back row-major, store the raw projection
      bf16* sbuf = st + (i & 1) * 2048;
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
          f32x4 a4;
#pragma unroll
          for (int e = 0; e < 4; e++) a4[e] = acc[i][j][4 * g + e];
          stage_put4(sbuf, r, j * 32 + 8 * g + 4 * hi, a4);
        }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int row = q * 8 + (l >> 3);
        v[q] = *(const bf16x8*)(sbuf + row * 64 + ((ch ^ ((row >> 1) & 7)) << 3));
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int m = m_w + i * 32 + q * 8 + (l >> 3);
        if (!GUARD || (col_ok && m < p.M)) *(bf16x8*)((bf16*)p.C + (size_t)m * p.ldc + n_w + ch * 8) = v[q];
      }
    }
    {
#pragma unroll
      for (int k = 0; k < 2; k++) {
        float gm[8];
#pragma unroll
        for (int e = 0; e < 4; e++) { gm[e] = g0[e]; gm[4 + e] = g1[e]; }
        const bf16x8 o = qk_norm_rope_chunk(v[2 * qh + k], rs, which == 0 ? p.qk_q_scale : 1.f, gm, cs[k], sn[k]);
        const int m = row_of(s, k);
        if (!GUARD || m < p.M) *(bf16x8*)((bf16*)p.C2 + (size_t)m * p.ldc2 + n_w + ch * 8) = o;
        if constexpr (NI == 1) {
          if (to_cache && cpos[s][k] >= 0) *(bf16x8*)((bf16*)p.qk_cache + (size_t)cpos[s][k] * p.qk_ld_cache + (n_w - hd) + ch * 8) = o;
        }
      }
    }
  }
}
// v columns of a decode step: the raw projection goes to C and, as bf16 rows, to the KV cache behind the k~ columns (tfx_qk_norm_rope_fwd's `cache` path)
template <int NI>
TFX_DEV void staged_epilogue_vcache(const GemmNT& p, f32x16 (&acc)[NI][2], int m_w, int n_w, bf16* st) {
  const int l = threadIdx.x & 63, hi = l >> 5, r = l & 31, ch = l & 7;
  const int hd = p.qk_heads * 64;
  const bool col_ok = n_w + ch * 8 < p.N;
  int cp[NI][4];
#pragma unroll
  for (int i = 0; i < NI; i++)
#pragma unroll
    for (int q = 0; q < 4; q++) { const int m = m_w + i * 32 + q * 8 + (l >> 3); cp[i][q] = m < p.M ? p.qk_cache_pos[m] : -1; }
  asm volatile("" ::: "memory");
#pragma unroll
  for (int i = 0; i < NI; i++) {
    bf16* sbuf = st + (i & 1) * 2048;
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        f32x4 a4;
#pragma unroll
        for (int e = 0; e < 4; e++) a4[e] = acc[i][j][4 * g + e];
        stage_put4(sbuf, r, j * 32 + 8 * g + 4 * hi, a4);
      }
    bf16x8 v[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int row = q * 8 + (l >> 3);
      v[q] = *(const bf16x8*)(sbuf + row * 64 + ((ch ^ ((row >> 1) & 7)) << 3));
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int m = m_w + i * 32 + q * 8 + (l >> 3);
      if (col_ok && m < p.M) *(bf16x8*)((bf16*)p.C + (size_t)m * p.ldc + n_w + ch * 8) = v[q];
      if (col_ok && cp[i][q] >= 0) *(bf16x8*)((bf16*)p.qk_cache + (size_t)cp[i][q] * p.qk_ld_cache + (n_w - hd) + ch * 8) = v[q];
    }
  }
}
template <int NI>
TFX_DEV void staged_epilogue_qknr(const GemmNT& p, f32x16 (&acc)[NI][2], int m_w, int n_w, bf16* st) {
  const int hd_ = p.qk_heads * 64;
  if constexpr (NI == 1) {
    if (p.qk_cache && n_w >= 2 * hd_ && n_w + 64 <= 3 * hd_) { staged_epilogue_vcache<NI>(p, acc, m_w, n_w, st); return; }      // v columns of a decode step (wave-uniform)
  }
  if (n_w + 64 > 2 * p.qk_heads * 64) { staged_epilogue_bf16<NI>(p, acc, m_w, n_w, st); return; }      // v / gate columns: the plain staged store (wave-uniform)
  if (m_w + 32 * NI <= p.M) staged_epilogue_qknr_<NI, false>(p, acc, m_w, n_w, st);
  else staged_epilogue_qknr_<NI, true>(p, acc, m_w, n_w, st);
}
// C(fp32)[mo][n_w .. n_w+63] = acc + bias (the logits / flow-prediction projections): the same staging idea for 4-byte outputs.  The direct form
// stores 16 bytes per lane in accumulator shape - 32 rows x 32 bytes behind every instruction, 32 cache lines walked one by one (the logits
// GEMM of the step, 65536 x 448 x 512 with a 117 MB fp32 result, took 195 us: 0.9 TB/s).  Here a 32 x 64 block goes through a wave-private
// [32][68] fp32 image (272-byte rows: the 16-byte accumulator-shaped writes of rows r and r + 16 share banks, nothing else does) and leaves as
// 16-byte stores with 16 lanes per 256-byte row segment: 8 lines per instruction.  Needs ldc % 4 == 0, N % 4 == 0, C 16-byte aligned.
template <int NI>
TFX_DEV void staged_epilogue_f32(const GemmNT& p, f32x16 (&acc)[NI][2], int m_w, int n_w, bf16* st_) {
  float* st = (float*)st_;                                       // 8704 of the wave's 16384 bytes
  const int l = threadIdx.x & 63, hi = l >> 5, r = l & 31, c16 = l & 15;
  int mo[NI][8];                                               // output row of (block i, flush pass q): row q * 4 + (l >> 4)
#pragma unroll
  for (int i = 0; i < NI; i++)
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const int m = m_w + i * 32 + q * 4 + (l >> 4);
      mo[i][q] = m < p.M ? (p.rowmap ? p.rowmap[m] : m) : -1;
    }
  f32x4 bias[2][4];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 2; j++)
#pragma unroll
    for (int g = 0; g < 4; g++) bias[j][g] = zero4;
  if (p.bias) {
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int g = 0; g < 4; g++) bias[j][g] = *(const f32x4*)(p.bias + min(n_w + j * 32 + 8 * g + 4 * hi, p.N - 4));
  }
  asm volatile("" ::: "memory");                               // side loads stay ahead of the first store (one in-order counter)
  const bool col_ok = n_w + c16 * 4 < p.N;
#pragma unroll
  for (int i = 0; i < NI; i++) {
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = acc[i][j][4 * g + e] + bias[j][g][e];
        *(f32x4*)(st + r * 68 + j * 32 + 8 * g + 4 * hi) = v;
      }
    f32x4 v[8];
#pragma unroll
    for (int q = 0; q < 8; q++) v[q] = *(const f32x4*)(st + (q * 4 + (l >> 4)) * 68 + c16 * 4);
#pragma unroll
    for (int q = 0; q < 8; q++)
      if (col_ok && mo[i][q] >= 0) *(f32x4*)((float*)p.C + (size_t)mo[i][q] * p.ldc + n_w + c16 * 4) = v[q];
  }
}
template <int EPI> TFX_DEV bool can_stage(const GemmNT& p) {
  bool ok = ((p.ldc | p.N) & 7) == 0 && (((uintptr_t)p.C) & 15) == 0;
  if constexpr (EPI == EPI_GEGLU) ok = ok && (p.ldc2 & 7) == 0 && (((uintptr_t)p.C2) & 15) == 0;
  if constexpr (EPI == EPI_GEGLU_BWD) ok = ok && (p.ldaux & 7) == 0 && (((uintptr_t)p.aux) & 15) == 0 && p.N >= 64;
  if constexpr (EPI == EPI_RESID) ok = ok && (p.ldr & 7) == 0 && (((uintptr_t)p.R) & 15) == 0 && p.N >= 8;
  return ok;
}
// epilogue of the LDS-DMA kernels: staged stores where the layout allows, else the direct pipelined form.
// `st` = this wave's private LDS staging area (>= 12 KiB), valid once every wave has left the K loop.
template <int EPI, int NI>
TFX_DEV void nt_epilogue(const GemmNT& p, f32x16 (&acc)[NI][2], int m_w, int n_w, bf16* st, const float* gtab = nullptr, bool use_tab = false, int lane = -1) {
  if constexpr (EPI == EPI_BF16) {
    if (can_stage<EPI>(p)) {
      if (!p.rowmap && !p.bias) staged_epilogue_bf16<NI, true>(p, acc, m_w, n_w, st, lane);
      else staged_epilogue_bf16<NI, false>(p, acc, m_w, n_w, st, lane);
      return;
    }
  } else if constexpr (EPI == EPI_F32) {
    if (((p.ldc | p.N) & 3) == 0 && (((uintptr_t)p.C) & 15) == 0) { staged_epilogue_f32<NI>(p, acc, m_w, n_w, st); return; }
  } else if constexpr (EPI == EPI_GEGLU || EPI == EPI_GEGLU_BWD) {
    if (can_stage<EPI>(p)) { staged_epilogue_geglu<EPI, NI>(p, acc, m_w, n_w, st, gtab, use_tab); return; }
  } else if constexpr (EPI == EPI_RESID) {
    if (can_stage<EPI>(p)) { staged_epilogue_resid<NI>(p, acc, m_w, n_w, st); return; }
  }
  if constexpr (EPI == EPI_QKNR) staged_epilogue_qknr<NI>(p, acc, m_w, n_w, st);      // (the launcher only takes stageable layouts here: qknr_fusable)
  else fast_epilogue<EPI, NI>(p, acc, m_w, n_w, lane);
}


// ------------------------------------------------------------------------------------------------
// gemm_nt
// ------------------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmNT p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16* As = (bf16*)smem_raw;                 // [2][128*64]
  bf16* Bs = As + 2 * BM * BK;                // [2][128*64]

  const int t = threadIdx.x, l = t & 63, w = t >> 6;
  const int wm = w >> 1, wn = w & 1, hi = l >> 5;
  const int ntn = (p.N + BN - 1) / BN;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (bid / ntn) * BM, n0 = (bid % ntn) * BN;
  const int nk = p.K / BK;

  // per-thread staging coordinates: 4 x 16-byte chunks of A and of B per k-tile
  int srow[4], skc[4];
  const bf16 *ga[4], *ga2[4], *gb[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    int c = t + 256 * i;
    srow[i] = c >> 3; skc[i] = c & 7;
    int rm = min(m0 + srow[i], p.M - 1);
    if (p.a_rowmap) rm = p.a_rowmap[rm];
    int rn = min(n0 + srow[i], p.N - 1);
    ga[i] = p.A + (size_t)rm * p.lda + skc[i] * 8;
    ga2[i] = p.A2 ? p.A2 + (size_t)rm * p.lda2 + skc[i] * 8 : nullptr;
    gb[i] = p.B + (size_t)rn * p.ldb + skc[i] * 8;
  }
  u32x4 ra[4], rb[4];
  auto gload = [&](int kt) {
    int k0 = kt * BK;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const bf16* src = (p.A2 && k0 >= p.K1) ? ga2[i] + (k0 - p.K1) : ga[i] + k0;
      ra[i] = *(const u32x4*)src;
      rb[i] = *(const u32x4*)(gb[i] + k0);
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      int off = srow[i] * BK + ((skc[i] ^ ((srow[i] >> 1) & 7)) << 3);
      *(u32x4*)(As + buf * BM * BK + off) = ra[i];
      *(u32x4*)(Bs + buf * BN * BK + off) = rb[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  gload(0);
  sstore(0);
  __syncthreads();

  const int arow0 = wm * 64 + (l & 31), brow0 = wn * 64 + (l & 31);
  for (int kt = 0; kt < nk; kt++) {
    const int cur = kt & 1;
    if (kt + 1 < nk) gload(kt + 1);
    const bf16* as = As + cur * BM * BK;
    const bf16* bs = Bs + cur * BN * BK;
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      bf16x8 af[2], bfr[2];
#pragma unroll
      for (int i = 0; i < 2; i++) {
        int ar = arow0 + i * 32, br = brow0 + i * 32;
        af[i] = *(const bf16x8*)(as + ar * BK + (((ks * 2 + hi) ^ ((ar >> 1) & 7)) << 3));
        bfr[i] = *(const bf16x8*)(bs + br * BK + (((ks * 2 + hi) ^ ((br >> 1) & 7)) << 3));
      }
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);   // D'[n][m]
    }
    if (kt + 1 < nk) sstore(cur ^ 1);
    __syncthreads();
  }

  // epilogue: lane owns row m (= column of D'), register group g owns columns n..n+3
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int m = m0 + wm * 64 + i * 32 + (l & 31);
    if (m >= p.M) continue;
    const int mo = p.rowmap ? p.rowmap[m] : m;
    if (mo < 0) continue;
    if constexpr (EPI == EPI_GEGLU) {
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int n_a = n0 + wn * 64 + 8 * g + 4 * hi;
        if (n_a >= p.N) continue;
        f32x4 a, gt;
#pragma unroll
        for (int e = 0; e < 4; e++) { a[e] = acc[i][0][4 * g + e]; gt[e] = acc[i][1][4 * g + e]; }
        GegluFwd::apply(p, mo, n_a, a, gt);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const int n = n0 + wn * 64 + j * 32 + 8 * g + 4 * hi;
          if (n >= p.N) continue;
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] = acc[i][j][4 * g + e];
          Epilogue<EPI>::apply(p, m, mo, n, v);
        }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// gemm_tn: C[rowmap[n]][k] += sum_m A[m][n] * B[m][k]
// ------------------------------------------------------------------------------------------------
constexpr int TN_BMK = 64;      // contraction rows per LDS tile
constexpr int TN_LD = 136;      // padded LDS row stride (elements); 136 keeps 2 blocks/CU resident (69.6 KB LDS each)

// output column of product column k: identity, or the compaction of per-head padded columns (tfx.h `k_group`)
TFX_DEV int tn_out_col(const GemmTN& p, int k) {
  if (p.k_group <= 0) return k;
  const int c = k & 63;
  return c < p.k_group ? (k >> 6) * p.k_group + c : -1;
}

// Block order of the TN kernels (block b runs on XCD b % 8, each with a private L2).  The ntile x splits (tile, row chunk) pairs are numbered
// chunk-major and cut into 8 contiguous runs, one per XCD: an XCD walks all tiles of a chunk side by side, so every slab of A and B is fetched
// from HBM once per chunk and re-read by the other tiles from that XCD's L2.  Any split count works (the launcher rounds the grid up to a
// multiple of 8; surplus blocks leave at once), so it can be chosen to fill the chip: 22 tiles x 11 chunks = 242 blocks on 256 CUs where the
// multiple-of-8 rule of rounds 1-2 left 176 or two half-length rounds of 352.
struct TnBlock { int tile, mbeg, mend; };
TFX_DEV TnBlock tn_block(const GemmTN& p, int ntile) {
  const int per = gridDim.x >> 3;
  const int g = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  TnBlock b;
  const int chunk = ((p.M + p.splits - 1) / p.splits + TN_BMK - 1) / TN_BMK * TN_BMK;
  const int split = g / ntile;
  b.tile = g - split * ntile;
  b.mbeg = split < p.splits ? min(split * chunk, p.M) : p.M;      // chunks are rounded up to 64 rows: trailing splits can start past M (M = 2112, 8 splits: chunk 320)
  b.mend = min(p.M, b.mbeg + chunk);                              // -> empty range, the kernels return before forming an address
  return b;
}

// Ramped row chunks (one-wave kernel; round 5): with equal chunks every block of a launch finishes its row loop at the same moment and the fp32 atomics of ALL of them
// - tiles x chunks x 256 KiB, which the chip retires at ~1.25 TB/s whatever XCD they come from (tools/atomic_xcd_probe.hip): 48 us for the 60 MiB of a 2816 x 512
// launch, 27-42 % of the config-2 weight-gradient kernels (profiles/r05b_tn_atomics_fixed_cost.txt) - queue up behind an idle matrix pipe.  Chunk s is therefore
// c0 + d s steps of 64 rows long: the blocks of chunk 0 finish first and their atomics drain while the others still multiply; the last chunk's (tiles x 256 KiB)
// is what stays exposed.  d = 0: equal chunks (the other kernels' tn_block, multiples of 64 rows).  M % 64 == 0.
TFX_DEV TnBlock tn_block_ramp(const GemmTN& p, int ntile, int d) {
  if (d <= 0) return tn_block(p, ntile);
  const int per = gridDim.x >> 3;
  const int g = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  const int S = p.splits, U = p.M / TN_BMK;
  const int tri = d * S * (S - 1) / 2;
  const int c0 = (U - tri) / S, rem = U - tri - c0 * S;                 // the remainder goes to the last (longest) chunk
  TnBlock b;
  const int split = g / ntile;
  b.tile = g - split * ntile;
  if (split >= S) { b.mbeg = b.mend = p.M; return b; }
  const int start = c0 * split + d * split * (split - 1) / 2, len = c0 + d * split + (split == S - 1 ? rem : 0);
  b.mbeg = start * TN_BMK; b.mend = (start + len) * TN_BMK;
  return b;
}

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(GemmTN p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16* As = (bf16*)smem_raw;                         // [2][64*160]
  bf16* Bs = As + 2 * TN_BMK * TN_LD;                 // [2][64*160]

  const int t = threadIdx.x, l = t & 63, w = t >> 6;
  const int wn = w >> 1, wk = w & 1, hi = l >> 5;
  const int ntn = (p.N + 127) / 128, ntk = (p.K + 127) / 128;
  const int ntile = ntn * ntk;
  const TnBlock blk = tn_block(p, ntile);
  const int bid = blk.tile, mbeg = blk.mbeg, mend = blk.mend;
  const int n0 = (bid / ntk) * 128, k0 = (bid % ntk) * 128;
  if (mbeg >= mend) return;
  const int nsteps = (mend - mbeg + TN_BMK - 1) / TN_BMK;

  int srow[4], sch[4];
#pragma unroll
  for (int i = 0; i < 4; i++) { int c = t + 256 * i; srow[i] = c >> 4; sch[i] = c & 15; }
  u32x4 ra[4], rb[4];
  auto gload = [&](int st) {
    int mrow0 = mbeg + st * TN_BMK;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      int m = mrow0 + srow[i];
      int ca = n0 + sch[i] * 8, cb = k0 + sch[i] * 8;
      u32x4 z = {0, 0, 0, 0};
      const bool ok = m < mend;
      const int ma = (ok && p.a_rowmap) ? p.a_rowmap[m] : m;
      const int mb = (ok && p.b_rowmap) ? p.b_rowmap[m] : m;
      ra[i] = (ok && ca < p.a_cols) ? *(const u32x4*)(p.A + (size_t)ma * p.lda + ca) : z;
      rb[i] = (ok && cb < p.b_cols) ? *(const u32x4*)(p.B + (size_t)mb * p.ldb + cb) : z;
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      int off = srow[i] * TN_LD + sch[i] * 8;
      *(u32x4*)(As + buf * TN_BMK * TN_LD + off) = ra[i];
      *(u32x4*)(Bs + buf * TN_BMK * TN_LD + off) = rb[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  gload(0);
  sstore(0);
  __syncthreads();
  for (int st = 0; st < nsteps; st++) {
    const int cur = st & 1;
    if (st + 1 < nsteps) gload(st + 1);
    const bf16* as = As + cur * TN_BMK * TN_LD;
    const bf16* bs = Bs + cur * TN_BMK * TN_LD;
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      const int r0 = ks * 16 + 8 * hi;
      bf16x8 af[2], bfr[2];
#pragma unroll
      for (int i = 0; i < 2; i++) {
        af[i] = lds_tr8(as, TN_LD, r0, r0 + 4, wn * 64 + i * 32);
        bfr[i] = lds_tr8(bs, TN_LD, r0, r0 + 4, wk * 64 + i * 32);
      }
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);   // D[n][k]
    }
    if (st + 1 < nsteps) sstore(cur ^ 1);
    __syncthreads();
  }

  // D layout: col = lane&31 -> k, row(reg) -> n.  fp32 atomics: 32 consecutive k per half-wave.
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int n = n0 + wn * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (n >= p.N) continue;
      const int no = p.rowmap ? p.rowmap[n] : n;
      if (no < 0) continue;
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int k = k0 + wk * 64 + j * 32 + (l & 31);
        if (k < p.k_valid) {
          float v = acc[i][j][r] * p.alpha;
          const int ko = tn_out_col(p, k);
          if (ko >= 0) atomicAdd(p.C + (size_t)no * p.ldc + ko, v);
        }
      }
    }
}

// ------------------------------------------------------------------------------------------------
// LDS-DMA variants (global_load_lds, 16 B per lane): the register -> ds_write_b128 staging pass is what
// bounds the register-staged kernels above (ds_write_b128 ~13 LDS cycles per wave-instruction: ~415 cycles
// per 128x128x64 tile-step against 512 MFMA cycles).  The DMA writes lane-linear 1 KiB pieces, so the bank
// swizzle is applied to the SOURCE address (each lane fetches the chunk its LDS slot must hold) and to the
// read address - same involution on both sides (cdna_hip_programming.md rule 21).
// ------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(1))) const void gbl_void_t;
TFX_DEV void glds16(const bf16* g, bf16* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gbl_void_t*)g, (lds_void_t*)lds_wave_base, 16, 0, 0);
}

// Cold-operand prefetch (tfx.h tfx_gemm_nt_args.prefetch): blocks [grid0, gridDim) of a launch touch the next GEMM's weights - one dword per 64 bytes,
// 64 KiB per block, all of a thread's loads in flight at once - and leave: the lines then sit in the Infinity Cache (memory side: whichever XCD asks).
// Measured and not kept (LAB_NOTEBOOK round 6): touching each B tile from the XCD that will read it (into that L2) - the prefetch blocks carry the launch's
// LDS allocation, so a few of them with several tiles each stretch the launch; and a block touching its OWN panel ahead of its ring requests - the
// requests queue behind the touches.
constexpr int PF_BLOCK_BYTES = 64 * 1024;
TFX_DEV bool prefetch_block(const GemmNT& p, int grid0) {
  if ((int)blockIdx.x < grid0) return false;
  const long long o0 = (long long)((int)blockIdx.x - grid0) * PF_BLOCK_BYTES + (long long)threadIdx.x * 64;
  const char* base = (const char*)p.prefetch;
  uint32_t acc = 0;                                                 // (compiler-visible loads: an asm load's late result would land in a register hipcc has reused)
#pragma unroll
  for (int k = 0; k < PF_BLOCK_BYTES / (256 * 64); k++) {
    const long long o = o0 + (long long)k * 256 * 64;
    if (o + 4 <= (long long)p.prefetch_bytes) acc ^= *(const uint32_t*)(base + o);
  }
  asm volatile("" ::"v"(acc));
  return true;
}
static int prefetch_blocks(const GemmNT& p) {
  if (!p.prefetch || p.prefetch_bytes <= 0) return 0;
  const long long n = ((long long)p.prefetch_bytes + PF_BLOCK_BYTES - 1) / PF_BLOCK_BYTES;
  return (int)(n < 512 ? n : 512);
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt_glds_kernel(GemmNT p, int grid0) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  if (prefetch_block(p, grid0)) return;                           // (block-uniform, before any barrier)
  bf16* As = (bf16*)smem_raw;                 // [2][128*64], row r chunk c' holds global chunk c' ^ ((r >> 1) & 7): conflict-free for the ds_read_b128 lane groups
  bf16* Bs = As + 2 * BM * BK;

  const int t = threadIdx.x, l = t & 63, hi = l >> 5;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int ntn = (p.N + BN - 1) / BN;
  const int bid = xcd_remap(blockIdx.x, grid0);
  const int m0 = (bid / ntn) * BM, n0 = (bid % ntn) * BN;
  const int nk = p.K / BK;

  // wave w stages rows [32w, 32w+32) of both tiles: 4 DMA pieces of 8 rows x 128 B each
  const bf16 *ga[4], *ga2[4], *gb[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int row = w * 32 + j * 8 + (l >> 3);
    const int c = (l & 7) ^ ((row >> 1) & 7);
    int rm = min(m0 + row, p.M - 1);
    if (p.a_rowmap) rm = p.a_rowmap[rm];
    const int rn = min(n0 + row, p.N - 1);
    ga[j] = p.A + (size_t)rm * p.lda + c * 8;
    ga2[j] = p.A2 ? p.A2 + (size_t)rm * p.lda2 + c * 8 : nullptr;
    gb[j] = p.B + (size_t)rn * p.ldb + c * 8;
  }
  auto issue = [&](int kt, int buf) {
    const int k0 = kt * BK;
    const bool second = p.A2 && k0 >= p.K1;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      glds16(second ? ga2[j] + (k0 - p.K1) : ga[j] + k0, As + buf * BM * BK + (w * 32 + j * 8) * BK);
      glds16(gb[j] + k0, Bs + buf * BN * BK + (w * 32 + j * 8) * BK);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  issue(0, 0);
  const int arow0 = wm * 64 + (l & 31), brow0 = wn * 64 + (l & 31);
  // fragment addresses (bytes folded by the compiler): row r, k-step ks -> chunk (2*ks + hi) ^ ((r >> 1) & 7)
  auto ldfrag = [&](const bf16* base, int r, int ks) { return *(const bf16x8*)(base + r * BK + (((ks * 2 + hi) ^ ((r >> 1) & 7)) << 3)); };
  for (int kt = 0; kt < nk; kt++) {
    const int cur = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's pieces of tile kt have landed
    __builtin_amdgcn_s_barrier();                            // ... everyone's have, and everyone finished reading the other buffer
    if (kt + 1 < nk) issue(kt + 1, cur ^ 1);                 // DMA of tile kt+1 overlaps the MFMAs of tile kt
    const bf16* as = As + cur * BM * BK;
    const bf16* bs = Bs + cur * BN * BK;
    bf16x8 af[2][2], bfr[2][2];                              // register double-buffered fragments: reads of k-step ks+1 fly under the MFMAs of ks
#pragma unroll
    for (int i = 0; i < 2; i++) { af[0][i] = ldfrag(as, arow0 + i * 32, 0); bfr[0][i] = ldfrag(bs, brow0 + i * 32, 0); }
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      const int c = ks & 1;
      if (ks + 1 < 4) {
#pragma unroll
        for (int i = 0; i < 2; i++) { af[c ^ 1][i] = ldfrag(as, arow0 + i * 32, ks + 1); bfr[c ^ 1][i] = ldfrag(bs, brow0 + i * 32, ks + 1); }
      }
      __builtin_amdgcn_sched_barrier(0);                     // keep the prefetch ahead of this k-step's MFMAs (distinct registers)
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[c][j], af[c][i], acc[i][j], 0, 0, 0);
    }
  }

  __builtin_amdgcn_s_barrier();                                  // every wave is done with the operand tiles: LDS becomes staging space
  nt_epilogue<EPI, 2>(p, acc, m0 + wm * 64, n0 + wn * 64, As + w * 8192);   // 16 KiB per wave of the 64 KiB ring
}

// ------------------------------------------------------------------------------------------------
// Mid-size NT: at most ONE 128 x 128 tile per CU (mixed decode steps at samples x (modality length + 1) rows, prefills, latent projections,
// conditioning MLPs).  The 2-stage kernel above is then alone on its CU and every K-tile waits out a full memory round trip (1.4 us per tile
// measured at M = 640, N = 5504, K = 1024 - 15 % of the MFMA rate).  Same tile, fragments, accumulation order and epilogue, but a 4-slot ring of
// 32 KiB stages (three K-tiles of LDS-DMA in flight, counted vmcnt: 8 instructions per wave and tile), one block per CU.
// ------------------------------------------------------------------------------------------------
constexpr int MD_ST = 4;
constexpr int MD_STAGE = (BM + BN) * BK;                // elements per ring slot: A [128][64] then B [128][64]

template <int EPI>
__global__ __launch_bounds__(256, 1) void gemm_nt_mid_kernel(GemmNT p, int grid0) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  if (prefetch_block(p, grid0)) return;                           // (block-uniform, before any barrier)
  bf16* ring = (bf16*)smem_raw;

  const int t = threadIdx.x, l = t & 63, hi = l >> 5;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int ntn = (p.N + BN - 1) / BN;
  const int bid = xcd_remap(blockIdx.x, grid0);
  const int m0 = (bid / ntn) * BM, n0 = (bid % ntn) * BN;
  const int nk = p.K / BK;

  const bf16 *ga[4], *ga2[4], *gb[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int row = w * 32 + j * 8 + (l >> 3);
    const int c = (l & 7) ^ ((row >> 1) & 7);
    int rm = min(m0 + row, p.M - 1);
    if (p.a_rowmap) rm = p.a_rowmap[rm];
    const int rn = min(n0 + row, p.N - 1);
    ga[j] = p.A + (size_t)rm * p.lda + c * 8;
    ga2[j] = p.A2 ? p.A2 + (size_t)rm * p.lda2 + c * 8 : nullptr;
    gb[j] = p.B + (size_t)rn * p.ldb + c * 8;
  }
  auto issue = [&](int kt) {                               // 8 DMA instructions per wave and K-tile
    const int k0 = kt * BK;
    bf16* slot = ring + (kt % MD_ST) * MD_STAGE;
    const bool second = p.A2 && k0 >= p.K1;
#pragma unroll
    for (int j = 0; j < 4; j++) glds16_asm(second ? ga2[j] + (k0 - p.K1) : ga[j] + k0, slot + (w * 32 + j * 8) * BK);
#pragma unroll
    for (int j = 0; j < 4; j++) glds16_asm(gb[j] + k0, slot + BM * BK + (w * 32 + j * 8) * BK);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  for (int kt = 0; kt < min(MD_ST - 1, nk); kt++) issue(kt);
  const int arow0 = wm * 64 + (l & 31), brow0 = wn * 64 + (l & 31);
  auto ldfrag = [&](const bf16* base, int r, int ks) { return *(const bf16x8*)(base + r * BK + (((ks * 2 + hi) ^ ((r >> 1) & 7)) << 3)); };
  for (int kt = 0; kt < nk; kt++) {
    const int ahead = min(MD_ST - 2, nk - 1 - kt);           // K-tiles after kt that may still be in flight (in-order counter)
    switch (ahead) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    }
    __builtin_amdgcn_s_barrier();                            // tile kt has landed for everyone, and everyone finished reading slot (kt - 1) % MD_ST
    if (kt + MD_ST - 1 < nk) issue(kt + MD_ST - 1);          // refill the slot read in the previous iteration
    const bf16* as = ring + (kt % MD_ST) * MD_STAGE;
    const bf16* bs = as + BM * BK;
    bf16x8 af[2][2], bfr[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++) { af[0][i] = ldfrag(as, arow0 + i * 32, 0); bfr[0][i] = ldfrag(bs, brow0 + i * 32, 0); }
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      const int c = ks & 1;
      if (ks + 1 < 4) {
#pragma unroll
        for (int i = 0; i < 2; i++) { af[c ^ 1][i] = ldfrag(as, arow0 + i * 32, ks + 1); bfr[c ^ 1][i] = ldfrag(bs, brow0 + i * 32, ks + 1); }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[c][j], af[c][i], acc[i][j], 0, 0, 0);
    }
  }

  __builtin_amdgcn_s_barrier();                                  // the ring becomes staging space: 16 KiB per wave
  nt_epilogue<EPI, 2>(p, acc, m0 + wm * 64, n0 + wn * 64, ring + w * 8192);
}

// ------------------------------------------------------------------------------------------------
// Skinny NT (decode steps, time-conditioning MLPs: M <= 512 rows).  With a handful of 64-row blocks the 2-stage kernel
// above is pure latency: every K-tile waits one full memory round trip (12.9 us for M = 64, K = 512).  Here a block owns a
// 64 x 128 tile and keeps SK_ST - 1 K-tiles of LDS-DMA in flight (a 6-slot ring of 24 KiB stages = the whole K = 512 panel
// is requested up front), one wave per 32 x 64 sub-tile, same LDS layout / fragments / epilogues as the kernel above.
// ------------------------------------------------------------------------------------------------
constexpr int SK_BM = 64, SK_BN = 128, SK_ST = 6;
constexpr int SK_STAGE = (SK_BM + SK_BN) * BK;          // elements per ring slot: A [64][64] then B [128][64]

template <int EPI>
__global__ __launch_bounds__(256, 1) void gemm_nt_skinny_kernel(GemmNT p, int grid0) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  if (prefetch_block(p, grid0)) return;                           // (block-uniform, before any barrier)
  bf16* ring = (bf16*)smem_raw;

  const int t = threadIdx.x, l = t & 63, hi = l >> 5;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int ntn = (p.N + SK_BN - 1) / SK_BN;
  const int m0 = (blockIdx.x / ntn) * SK_BM, n0 = (blockIdx.x % ntn) * SK_BN;
  const int nk = p.K / BK;

  // wave w stages rows [16w, 16w+16) of the A slot (2 DMA pieces of 8 rows x 128 B) and rows [32w, 32w+32) of the B slot (4 pieces)
  const bf16 *ga[2], *ga2[2], *gb[4];
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int row = w * 16 + j * 8 + (l >> 3);
    const int c = (l & 7) ^ ((row >> 1) & 7);
    int rm = min(m0 + row, p.M - 1);
    if (p.a_rowmap) rm = p.a_rowmap[rm];
    ga[j] = p.A + (size_t)rm * p.lda + c * 8;
    ga2[j] = p.A2 ? p.A2 + (size_t)rm * p.lda2 + c * 8 : nullptr;
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int row = w * 32 + j * 8 + (l >> 3);
    const int c = (l & 7) ^ ((row >> 1) & 7);
    gb[j] = p.B + (size_t)min(n0 + row, p.N - 1) * p.ldb + c * 8;
  }
  auto issue = [&](int kt) {                               // 6 DMA instructions per wave and K-tile
    const int k0 = kt * BK;
    bf16* slot = ring + (kt % SK_ST) * SK_STAGE;
    const bool second = p.A2 && k0 >= p.K1;
#pragma unroll
    for (int j = 0; j < 2; j++) glds16_asm(second ? ga2[j] + (k0 - p.K1) : ga[j] + k0, slot + (w * 16 + j * 8) * BK);
#pragma unroll
    for (int j = 0; j < 4; j++) glds16_asm(gb[j] + k0, slot + SK_BM * BK + (w * 32 + j * 8) * BK);
  };

  f32x16 acc[1][2];
#pragma unroll
  for (int j = 0; j < 2; j++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[0][j][r] = 0.f;

  for (int kt = 0; kt < min(SK_ST - 1, nk); kt++) issue(kt);
  const int arow = wm * 32 + (l & 31), brow0 = wn * 64 + (l & 31);
  auto ldfrag = [&](const bf16* base, int r, int ks) { return *(const bf16x8*)(base + r * BK + (((ks * 2 + hi) ^ ((r >> 1) & 7)) << 3)); };
  for (int kt = 0; kt < nk; kt++) {
    // K-tiles kt+1 .. kt+ahead of this wave are still allowed in flight (6 instructions each, in-order counter)
    const int ahead = min(SK_ST - 2, nk - 1 - kt);
    switch (ahead) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
    }
    __builtin_amdgcn_s_barrier();                            // tile kt has landed for everyone, and everyone finished reading slot (kt - 1) % SK_ST
    if (kt + SK_ST - 1 < nk) issue(kt + SK_ST - 1);          // refill the slot read in the previous iteration
    const bf16* as = ring + (kt % SK_ST) * SK_STAGE;
    const bf16* bs = as + SK_BM * BK;
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      const bf16x8 af = ldfrag(as, arow, ks);
#pragma unroll
      for (int j = 0; j < 2; j++)
        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ldfrag(bs, brow0 + j * 32, ks), af, acc[0][j], 0, 0, 0);
    }
  }

  __builtin_amdgcn_s_barrier();                                  // the ring becomes staging space: 16 KiB per wave
  nt_epilogue<EPI, 1>(p, acc, m0 + wm * 32, n0 + wn * 64, ring + w * 8192);
}

// Decode-step form of the skinny kernel.  At M = 64 ... 256 rows the weights are what is streamed, a launch has a handful of tiles, and its time is
// ONE block's walk over K: ~0.36 us per 64-wide K-tile in the kernel above whatever the ring depth or tile width (measured: 5 or 8 tiles in flight,
// 64 x 128 or 64 x 64 tiles - the same 23 us at K = 2752) - a wave's barrier -> fragment reads -> four dependent MFMAs per accumulator is a serial
// chain with nothing to overlap it.  So K is split ACROSS THE FOUR WAVES of a block: a 64 x 64 output tile, wave w walks K quarter w in 32-wide
// slabs through its OWN 4-slot LDS-DMA ring (8 KiB slots, three slabs in flight, no barrier in the loop - the ring is private), and the four
// partial sums meet in LDS (fragment-major = conflict-free), where waves 0 / 1 add them and run the usual epilogue on 32 rows each.
// 64-byte LDS rows: 16-byte chunk index XOR ((row >> 2) & 3), applied to the DMA source and to the fragment reads.
constexpr int SD_BK = 32, SD_ST = 4;
constexpr int SD_SLOT = (64 + 64) * SD_BK;              // elements per ring slot: A [64][32] then B [64][32]

template <int EPI>
__global__ __launch_bounds__(256, 1) void gemm_nt_decode_kernel(GemmNT p, int grid0) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  if (prefetch_block(p, grid0)) return;                           // (block-uniform, before any barrier)
  const int t = threadIdx.x, l = t & 63, hi = l >> 5;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  bf16* ring = (bf16*)smem_raw + w * SD_ST * SD_SLOT;              // this wave's private ring (32 KiB)
  const int ntn = (p.N + 63) / 64;
  const int m0 = (blockIdx.x / ntn) * 64, n0 = (blockIdx.x % ntn) * 64;
  const int ns = p.K / SD_BK;
  const int per = (ns + 3) / 4;
  const int s0 = min(w * per, ns), s1 = min(s0 + per, ns);

  // a slab = 4 + 4 DMA pieces of 16 rows x 64 B; lane l of a piece fetches chunk (l & 3) ^ swz of row 16 j + (l >> 2)
  const bf16 *ga[4], *ga2[4], *gb[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int row = j * 16 + (l >> 2);
    const int c = (l & 3) ^ ((row >> 2) & 3);
    int rm = min(m0 + row, p.M - 1);
    if (p.a_rowmap) rm = p.a_rowmap[rm];
    ga[j] = p.A + (size_t)rm * p.lda + c * 8;
    ga2[j] = p.A2 ? p.A2 + (size_t)rm * p.lda2 + c * 8 : nullptr;
    gb[j] = p.B + (size_t)min(n0 + row, p.N - 1) * p.ldb + c * 8;
  }
  auto issue = [&](int sl) {                               // 8 DMA instructions per slab
    const int k0 = sl * SD_BK;
    bf16* slot = ring + ((sl - s0) & (SD_ST - 1)) * SD_SLOT;
    const bool second = p.A2 && k0 >= p.K1;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      glds16_asm(second ? ga2[j] + (k0 - p.K1) : ga[j] + k0, slot + j * 16 * SD_BK);
      glds16_asm(gb[j] + k0, slot + 64 * SD_BK + j * 16 * SD_BK);
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  if constexpr (EPI == EPI_QKNR) {                                 // the layer's soft-cap plan (tfx.h), by the first wave of the first block
    if (p.qk_plan && blockIdx.x == 0 && w == 0) softcap_plan_write(p.qk_gamma_q, p.qk_gamma_k, p.qk_norm_scale, p.qk_q_scale, p.qk_softcap, p.qk_plan);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // retire the row-map loads before the counted waits
  for (int sl = s0; sl < min(s0 + SD_ST - 1, s1); sl++) issue(sl);
  const int swz = ((l & 31) >> 2) & 3;
  const int frow = (l & 31) * SD_BK;
  for (int sl = s0; sl < s1; sl++) {
    const int ahead = min(SD_ST - 2, s1 - 1 - sl);                 // slabs sl+1 .. sl+ahead may still be in flight
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (sl + SD_ST - 1 < s1) issue(sl + SD_ST - 1);                // its slot was last read one iteration ago, by this wave only
    const bf16* as = ring + ((sl - s0) & (SD_ST - 1)) * SD_SLOT;
    const bf16* bs = as + 64 * SD_BK;
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      const int ck = ((ks * 2 + hi) ^ swz) << 3;
      bf16x8 af[2], bfr[2];
#pragma unroll
      for (int i = 0; i < 2; i++) { af[i] = *(const bf16x8*)(as + i * 32 * SD_BK + frow + ck); bfr[i] = *(const bf16x8*)(bs + i * 32 * SD_BK + frow + ck); }
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();                                                // every ring is idle: LDS becomes the reduction space
  float* red = (float*)smem_raw;                                  // [4 waves][4 fragments][16 registers][64 lanes] = 64 KiB
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) red[((w * 4 + i * 2 + j) * 16 + r) * 64 + l] = acc[i][j][r];
  __syncthreads();
  f32x16 sum[1][2];
  if (w < 2) {
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        float v = 0.f;
#pragma unroll
        for (int ww = 0; ww < 4; ww++) v += red[((ww * 4 + w * 2 + j) * 16 + r) * 64 + l];
        sum[0][j][r] = v;
      }
  }
  __syncthreads();                                                // the reduction space becomes the epilogue's staging area (16 KiB per wave)
  if (w < 2) nt_epilogue<EPI, 1>(p, sum, m0 + w * 32, n0, (bf16*)smem_raw + w * 8192);
}

constexpr int BM2 = 256, BN2 = 256;     // tile of the ping-pong kernel below

// TN with LDS-DMA: unpadded [64][128] tiles, 16-byte chunk index XOR ((row & 3) << 2) keeps the four rows of a
// ds_read_b64_tr_b16 group on distinct banks.  Needs M % 64 == 0 and no row maps (rows are never zero-filled).
TFX_DEV bf16x8 lds_tr8_swz(const bf16* tile, int rowA, int rowB, int c0) {
  const int l = threadIdx.x & 63;
  const int q = l & 15;
  const int col = c0 + 16 * ((l >> 4) & 1) + 4 * (q & 3);
  const int ra = rowA + (q >> 2), rb = rowB + (q >> 2);
  s16x4 lo = lds_tr4(tile + ra * 128 + ((((col >> 3) ^ ((ra & 3) << 2))) << 3) + (col & 7));
  s16x4 hi = lds_tr4(tile + rb * 128 + ((((col >> 3) ^ ((rb & 3) << 2))) << 3) + (col & 7));
  const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
  const u32x4 v = {l2[0], l2[1], h2[0], h2[1]};                                   // dword-granular register sequence
  return __builtin_bit_cast(bf16x8, v);
}


// ------------------------------------------------------------------------------------------------
// Ping-pong 256x256x64 NT kernel ("pp").  In the kernels above both waves of a SIMD are always in the same part of the
// K-step (both fetch fragments, then both issue MFMAs), so the matrix pipe idles while LDS / DMA work is issued and vice
// versa (PMC: MFMA busy 35-42 %).  Here a K-tile is cut into 4 half-tiles (A rows {a0 | a1}, B columns {b0 | b1} of every
// wave) and 4 phases, one 64x32 output quadrant x K=64 each (8 MFMAs per wave):
//      phase 1: read a0,b0 ; MFMA (a0,b0)      phase 2: read b1 ; MFMA (a0,b1)
//      phase 3: read a1    ; MFMA (a1,b1)      phase 4: -       ; MFMA (a1,b0)
// Every phase is  [ds_reads + LDS-DMA issue] s_barrier [MFMAs at raised priority] s_barrier , and the two wave groups
// (wr = 0 / 1; one wave of each per SIMD) run ONE barrier apart, so one group's MFMAs always overlap the other group's
// loads.  LDS holds two K-tiles (2 x 4 half-tiles x 16 KiB).  A half-tile slot is re-filled two phases after its last
// read (safe under the one-barrier skew) and three half-tiles stay in flight across the only wait of a K-tile,
// `s_waitcnt vmcnt(6)` before phase 4's first barrier, which certifies the next K-tile one phase before its first read.
// (Schedule after the 8-phase template of cdna_hip_programming.md; DMAs are inline asm so hipcc does not drain them.)
// ------------------------------------------------------------------------------------------------
// De-phasing.  Every tile of a GEMM takes the same time, so all 256 CUs run their K loops together and reach their epilogues
// together: the C tiles of a round (33 MB for 256 x 256 bf16 tiles) hit HBM as one burst and the chip's power draw swings
// between an all-MFMA and an all-store phase.  Holding the odd blocks of the FIRST round back by `cycles` shader clocks shifts
// half of the CUs by that much for the rest of the kernel.  Measured on three MI355X boxes (full training step, dim512/d8):
// 18000 cycles (half a K = 512 tile) = -2.0 ... -3.4 % step time; the delayed blocks' tiles run faster (80 % of the inserted
// wait is recovered inside the GEMM) and the kernels that FOLLOW (attention -6 %, TN -3 %) run at higher clocks - the GEMM
// phase is power-limited (tools/clock_probe.hip) and the smoother draw leaves the controller more headroom.  The same delay
// in the TN kernel measured no effect.  Round 3, after the epilogues got shorter: 12000 cycles (-0.13 ... -0.25 ms against 18000 on two boxes,
// flat between 6000 and 15000 on a third; 0 = +0.2 ms).
TFX_DEV void dephase_first_round(int cycles, int first_round_blocks) {
  if (cycles > 0 && (int)blockIdx.x < first_round_blocks && (blockIdx.x & 1)) {
    const unsigned long long t0 = __builtin_readcyclecounter();
    // s_sleep 8 = 512 clocks; the trip count is bounded as well, should the counter ever tick slower than the shader clock
    for (int it = 0; it < cycles / 512 + 8 && __builtin_readcyclecounter() - t0 < (unsigned long long)cycles; it++) __builtin_amdgcn_s_sleep(8);
  }
}

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt_pp_kernel(GemmNT p, int stagger, const float* gtab) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16* lds = (bf16*)smem_raw;                                    // [2 K-tiles][A0, A1, B0, B1][128 rows x 64]
  constexpr int HALF = 128 * BK;
  dephase_first_round(stagger, 256);

  const int t = threadIdx.x, l = t & 63, hi = l >> 5;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = w >> 2, wc = w & 3;
  const int ntn = (p.N + BN2 - 1) / BN2;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (bid / ntn) * BM2, n0 = (bid % ntn) * BN2;
  const int nk = p.K / BK;

  // half-tile row r' (0..127):  A_h -> tile row (r' >> 6) * 128 + h * 64 + (r' & 63) ;  B_h -> tile col (r' >> 5) * 64 + h * 32 + (r' & 31)
  // wave w stages rows [16w, 16w+16) of every half-tile: 2 DMA pieces of 8 rows x 128 B
  const bf16 *gA[2][2], *gA2[2][2], *gB[2][2];
#pragma unroll
  for (int h = 0; h < 2; h++)
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int rp = w * 16 + j * 8 + (l >> 3);
      const int c = (l & 7) ^ ((rp >> 1) & 7);
      int rm = min(m0 + (rp >> 6) * 128 + h * 64 + (rp & 63), p.M - 1);
      if (p.a_rowmap) rm = p.a_rowmap[rm];
      const int rn = min(n0 + (rp >> 5) * 64 + h * 32 + (rp & 31), p.N - 1);
      gA[h][j] = p.A + (size_t)rm * p.lda + c * 8;
      gA2[h][j] = p.A2 ? p.A2 + (size_t)rm * p.lda2 + c * 8 : nullptr;
      gB[h][j] = p.B + (size_t)rn * p.ldb + c * 8;
    }
  auto issueA = [&](int kt, int h) {
    const int k0 = kt * BK;
    const bool second = p.A2 && k0 >= p.K1;
    bf16* dst = lds + ((kt & 1) * 4 + h) * HALF + w * 16 * BK;
#pragma unroll
    for (int j = 0; j < 2; j++) glds16_asm(second ? gA2[h][j] + (k0 - p.K1) : gA[h][j] + k0, dst + j * 8 * BK);
  };
  auto issueB = [&](int kt, int h) {
    bf16* dst = lds + ((kt & 1) * 4 + 2 + h) * HALF + w * 16 * BK;
#pragma unroll
    for (int j = 0; j < 2; j++) glds16_asm(gB[h][j] + kt * BK, dst + j * 8 * BK);
  };
  auto issueA1 = [&](int kt, int h, int j) {                      // one piece (8 rows) of a half-tile
    const int k0 = kt * BK;
    const bool second = p.A2 && k0 >= p.K1;
    glds16_asm(second ? gA2[h][j] + (k0 - p.K1) : gA[h][j] + k0, lds + ((kt & 1) * 4 + h) * HALF + (w * 16 + j * 8) * BK);
  };
  auto issueB1 = [&](int kt, int h, int j) { glds16_asm(gB[h][j] + kt * BK, lds + ((kt & 1) * 4 + 2 + h) * HALF + (w * 16 + j * 8) * BK); };
  const int ra = wr * 64 + (l & 31), rb = wc * 32 + (l & 31);
  auto ldA = [&](int kt, int h, int il, int ks) {
    const int r = ra + il * 32;
    return *(const bf16x8*)(lds + ((kt & 1) * 4 + h) * HALF + r * BK + (((ks * 2 + hi) ^ ((r >> 1) & 7)) << 3));
  };
  auto ldB = [&](int kt, int h, int ks) {
    return *(const bf16x8*)(lds + ((kt & 1) * 4 + 2 + h) * HALF + rb * BK + (((ks * 2 + hi) ^ ((rb >> 1) & 7)) << 3));
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

#ifdef TFX_PP_TIMING
  // debug build: wave 0 / lane 0 of every block stamps s_memtime into (uint64*)aux[blockIdx * 8 + i] (EPI_BF16 does not use aux)
  // (EPI_BF16 / GEGLU do not use `aux`; GEGLU backward does - its stamps travel through the unused `R`)
  unsigned long long* stamps = (unsigned long long*)(EPI == EPI_GEGLU_BWD ? (const void*)p.R : (const void*)p.aux) + (size_t)blockIdx.x * 8;
#define PP_STAMP(i) { if (t == 0) stamps[i] = __builtin_readcyclecounter(); }
#else
#define PP_STAMP(i)
#endif
  PP_STAMP(0)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // retire the row-map loads before the counted waits
  // GEGLU forward: the 32 KiB GELU table (geglu_uvh_grid) goes into the LDS the operand ring leaves free, 4 KiB per wave, issued AHEAD of the first
  // K-tile - the DMAs retire in order, so the prologue's wait for K-tile 0 covers the table as well
  if constexpr (EPI == EPI_QKNR) {                                 // the layer's soft-cap plan (tfx.h), by the first wave of the first block
    if (p.qk_plan && blockIdx.x == 0 && w == 0) softcap_plan_write(p.qk_gamma_q, p.qk_gamma_k, p.qk_norm_scale, p.qk_q_scale, p.qk_softcap, p.qk_plan);
  }
  const float* gtab_lds = (const float*)(smem_raw + 8 * HALF * 2);  // (only dereferenced when the launch reserved it: gtab != nullptr)
  if constexpr (EPI == EPI_GEGLU) {
    if (gtab) {                                                   // kernel argument: block-uniform
#pragma unroll
      for (int j = 0; j < 4; j++) glds16_asm((const bf16*)gtab + (size_t)(w * 4 + j) * 512 + l * 8, (const bf16*)gtab_lds + (w * 4 + j) * 512);
    }
  }
  // DMA schedule, 2 pieces per wave and phase, issued inside the MFMA blocks:  p1: A1(kt+1)  p2: B1(kt+1)  p3: A0(kt+2)  p4: B0(kt+2)
  // (every slot is re-filled >= 2 phases after its last read: A1 read p3, B1 read p2, A0/B0 read p1 of the K-tile before).
  // One wait per K-tile: vmcnt(2) before p4's first barrier leaves only A0(kt+2) outstanding, i.e. certifies all of K-tile kt+1
  // one phase before its first read.
  issueA(0, 0); issueB(0, 0); issueA(0, 1); issueB(0, 1);
  if (nk > 1) {
    issueA(1, 0); issueB(1, 0);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");             // all of K-tile 0 landed, A0,B0 of K-tile 1 in flight
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  PP_STAMP(1)
  if (wr == 1) __builtin_amdgcn_s_barrier();                      // group 1 runs one barrier behind group 0 from here on

#define PP_BAR() { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
// MFMA block of one phase; the phase's two LDS-DMA pieces are issued BETWEEN its MFMAs (after k-steps 0 and 2), where their
// issue cost hides under the matrix pipe (~60 cycles there against 100-185 in a segment that also carries the ds_reads).
#define PP_MFMA(I0, J, AF, BF, DMA0, DMA1)                                                                    \
  __builtin_amdgcn_s_setprio(1);                                                                              \
  _Pragma("unroll") for (int ks = 0; ks < 4; ks++) {                                                          \
    acc[I0][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF[ks], AF[0][ks], acc[I0][J], 0, 0, 0);            \
    acc[I0 + 1][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF[ks], AF[1][ks], acc[I0 + 1][J], 0, 0, 0);    \
    if (ks == 0) { DMA0; }                                                                                    \
    if (ks == 2) { DMA1; }                                                                                    \
  }                                                                                                           \
  __builtin_amdgcn_s_setprio(0);                                                                              \
  asm volatile("" : "+v"(acc[I0][J]), "+v"(acc[I0 + 1][J]));   /* pin: LLVM may sink pure MFMAs past the barrier */

  for (int kt = 0; kt < nk; kt++) {
    bf16x8 a[2][4], b0[4], b1[4];
    const bool n1 = kt + 1 < nk, n2 = kt + 2 < nk;
    // ---- phase 1: read a0 (, b0) ; MFMA (a0, b0) + DMA A1(kt+1)
#pragma unroll
    for (int ks = 0; ks < 4; ks++) b0[ks] = ldB(kt, 0, ks);
#pragma unroll
    for (int il = 0; il < 2; il++)
#pragma unroll
      for (int ks = 0; ks < 4; ks++) a[il][ks] = ldA(kt, 0, il, ks);
    PP_BAR()
    PP_MFMA(0, 0, a, b0, if (n1) issueA1(kt + 1, 1, 0), if (n1) issueA1(kt + 1, 1, 1))
    PP_BAR()
    // ---- phase 2: read b1 ; MFMA (a0, b1) + DMA B1(kt+1)
#pragma unroll
    for (int ks = 0; ks < 4; ks++) b1[ks] = ldB(kt, 1, ks);
    PP_BAR()
    PP_MFMA(0, 1, a, b1, if (n1) issueB1(kt + 1, 1, 0), if (n1) issueB1(kt + 1, 1, 1))
    PP_BAR()
    // ---- phase 3: read a1 ; MFMA (a1, b1) + DMA A0(kt+2)
#pragma unroll
    for (int il = 0; il < 2; il++)
#pragma unroll
      for (int ks = 0; ks < 4; ks++) a[il][ks] = ldA(kt, 1, il, ks);
    PP_BAR()
    PP_MFMA(2, 1, a, b1, if (n2) issueA1(kt + 2, 0, 0), if (n2) issueA1(kt + 2, 0, 1))
    PP_BAR()
    // ---- phase 4: (read B0 of K-tile kt+1 ;) certify K-tile kt+1 (only A0(kt+2) may still be outstanding) ; MFMA (a1, b0) + DMA B0(kt+2)
    if (n2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PP_BAR()
    PP_MFMA(2, 0, a, b0, if (n2) issueB1(kt + 2, 0, 0), if (n2) issueB1(kt + 2, 0, 1))
    PP_BAR()
  }
#undef PP_BAR
#undef PP_MFMA
  if (wr == 0) __builtin_amdgcn_s_barrier();
  PP_STAMP(2)
  nt_epilogue<EPI, 4>(p, acc, m0 + wr * 128, n0 + wc * 64, lds + w * 8192, gtab_lds, EPI == EPI_GEGLU && gtab != nullptr);   // all of the ring is free after the last barrier: 16 KiB per wave
  PP_STAMP(3)
#ifdef TFX_PP_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  PP_STAMP(4)
#endif
#undef PP_STAMP
}

// ------------------------------------------------------------------------------------------------
// One-wave-per-SIMD 256 x 256 x 64 NT kernel ("ow", round 5): 4 waves, each owns 128 x 128 of the tile = 16 accumulators of 32 x 32 in 256 AGPRs,
// the shape of the vendor library's kernel for these GEMMs (8 fragment reads per 16 MFMAs instead of the ping-pong kernel's 12, no phase
// barriers).  Round 4 built this shape under hipcc twice and landed on the ping-pong kernel's speed; here the K loop is ONE hand-scheduled
// inline-asm block (gemm_nt_ow_loop.inc, written by tools/gen_nt_ow_loop.py - the schedule is documented there):
//   * operands travel by LDS-DMA (`buffer_load_dwordx4 ... offen lds`): one lane offset per operand, the piece (16 rows further down) in a scalar
//     offset, the LDS target in M0 - no vector address arithmetic in the loop (the ping-pong kernel forms a 64-bit address per piece);
//   * the fragments of a WHOLE K-tile sit in registers (128 VGPRs), so an LDS buffer is free a quarter into its K-tile and K-tile kt+2 is
//     fetched into it: two K-tiles in flight with two 64 KiB buffers;
//   * two barriers per K-tile, each behind a wait whose operations were issued 100+ clocks earlier.
// Same LDS layout (128-byte rows, 16-byte slot index XOR ((row >> 1) & 7)), same per-accumulator k order and same epilogues as the ping-pong
// kernel: results are bit-identical to it (tests/test_kernels_gpu.py).  Not for row-gathered A, split A or the fused QK-norm epilogue (launch_ow).
// ------------------------------------------------------------------------------------------------
#define OW_ACC(n) "+a"(acc[(n) >> 3][((n) >> 1) & 3][(n) & 1])
#define OW_ACC8(b) OW_ACC(b), OW_ACC(b + 1), OW_ACC(b + 2), OW_ACC(b + 3), OW_ACC(b + 4), OW_ACC(b + 5), OW_ACC(b + 6), OW_ACC(b + 7)
#define OW_FR(n) "=&v"(fr[n])
#define OW_FR8(b) OW_FR(b), OW_FR(b + 1), OW_FR(b + 2), OW_FR(b + 3), OW_FR(b + 4), OW_FR(b + 5), OW_FR(b + 6), OW_FR(b + 7)
template <int EPI>
__global__ __launch_bounds__(256, 1) void gemm_nt_ow_kernel(GemmNT p, int stagger, const float* gtab) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16* lds = (bf16*)smem_raw;                                    // [2 K-tiles][A 256 rows x 64 | B 256 rows x 64], then (GEGLU forward) the 32 KiB GELU grid
  dephase_first_round(stagger, 256);
  const int t = threadIdx.x, l = t & 63, hi = l >> 5;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = w >> 1, wc = w & 1;
  const int ntn = (p.N + BN2 - 1) / BN2;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (bid / ntn) * BM2, n0 = (bid % ntn) * BN2;

  const float* gtab_lds = (const float*)(smem_raw + 2 * (BM2 + BN2) * BK * 2);      // (only dereferenced when the launch reserved it: gtab != nullptr)
  if constexpr (EPI == EPI_GEGLU) {
    if (gtab) {                                                   // issued ahead of K-tile 0: the DMAs retire in order, the loop's first wait covers the table
#pragma unroll
      for (int j = 0; j < 8; j++) glds16_asm((const bf16*)gtab + (size_t)(w * 8 + j) * 512 + l * 8, (const bf16*)gtab_lds + (w * 8 + j) * 512);
    }
  }
  // DMA pieces: operand tile rows 16 j + 8 (w & 1) + (l >> 3) of the wave pair's half (w >> 1), j = 0 .. 7 - the slot swizzle ((row >> 1) & 7) is then the
  // same for all eight pieces of a lane: ONE lane offset per operand.  Raw buffer resources: rows past M / N lie past num_records and read as zeros.
  const uint32_t rowp = (uint32_t)((w >> 1) * 128 + 8 * (w & 1) + (l >> 3));
  const uint32_t chunk = (uint32_t)((l & 7) ^ ((4 * (w & 1) + (l >> 4)) & 7));
  uint32_t voA = (((uint32_t)m0 + rowp) * (uint32_t)p.lda + chunk * 8u) * 2u;
  uint32_t voB = (((uint32_t)n0 + rowp) * (uint32_t)p.ldb + chunk * 8u) * 2u;
  const uint64_t baseA = (uint64_t)(uintptr_t)p.A, baseB = (uint64_t)(uintptr_t)p.B;
  u32x4 rsA, rsB;
  rsA[0] = __builtin_amdgcn_readfirstlane((uint32_t)baseA); rsA[1] = __builtin_amdgcn_readfirstlane((uint32_t)(baseA >> 32) & 0xffffu);
  rsA[2] = __builtin_amdgcn_readfirstlane(((uint32_t)(p.M - 1) * (uint32_t)p.lda + (uint32_t)p.K) * 2u); rsA[3] = 0x00020000u;
  rsB[0] = __builtin_amdgcn_readfirstlane((uint32_t)baseB); rsB[1] = __builtin_amdgcn_readfirstlane((uint32_t)(baseB >> 32) & 0xffffu);
  rsB[2] = __builtin_amdgcn_readfirstlane(((uint32_t)(p.N - 1) * (uint32_t)p.ldb + (uint32_t)p.K) * 2u); rsB[3] = 0x00020000u;
  const uint32_t stA = __builtin_amdgcn_readfirstlane(32u * (uint32_t)p.lda), stB = __builtin_amdgcn_readfirstlane(32u * (uint32_t)p.ldb);
  const uint32_t lds0 = (uint32_t)(size_t)(lds_void_t*)lds;
  uint32_t sM = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)((w >> 1) * 16384 + (w & 1) * 1024));
  uint32_t cnt = __builtin_amdgcn_readfirstlane((uint32_t)(p.K / BK)), so, delta = 65536u;
  // fragment reads: row (l & 31) of 32-row block i, k-step ks -> slot (2 ks + hi) ^ ((l >> 1) & 7); block i is an immediate offset of 4 KiB
  uint32_t raA[4], raB[4];
#pragma unroll
  for (int ks = 0; ks < 4; ks++) {
    const uint32_t slot = (uint32_t)(((2 * ks + hi) ^ ((l >> 1) & 7)) << 4);
    raA[ks] = lds0 + (uint32_t)((wr * 128 + (l & 31)) * 128) + slot;
    raB[ks] = lds0 + (uint32_t)(BM2 * BK * 2 + (wc * 128 + (l & 31)) * 128) + slot;
  }
  if constexpr (EPI == EPI_QKNR) {                                 // (never launched: launch_ow)
    if (p.qk_plan && blockIdx.x == 0 && w == 0) softcap_plan_write(p.qk_gamma_q, p.qk_gamma_k, p.qk_norm_scale, p.qk_q_scale, p.qk_softcap, p.qk_plan);
  }

  f32x16 acc[2][4][2];                                             // [column half][row block][column block of the half]: acc[h] is what nt_epilogue<EPI, 4> takes
#pragma unroll
  for (int h = 0; h < 2; h++)
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[h][i][j][r] = 0.f;
  u32x4 fr[32];
  asm volatile(
#include "gemm_nt_ow_loop.inc"
      : OW_ACC8(0), OW_ACC8(8), OW_FR8(0), OW_FR8(8), OW_FR8(16), OW_FR8(24),
        "+v"(raA[0]), "+v"(raA[1]), "+v"(raA[2]), "+v"(raA[3]), "+v"(raB[0]), "+v"(raB[1]), "+v"(raB[2]), "+v"(raB[3]),
        "+v"(voA), "+v"(voB), "+s"(sM), "+s"(cnt), "=&s"(so), "+s"(delta)
      : "s"(rsA), "s"(rsB), "s"(stA), "s"(stB)
      : "memory", "scc");

  __builtin_amdgcn_s_barrier();                                     // every wave is done with the operand tiles: 16 KiB of staging space per call below
#pragma unroll
  for (int h = 0; h < 2; h++)
    nt_epilogue<EPI, 4>(p, acc[h], m0 + wr * 128, n0 + wc * 128 + h * 64, lds + (w * 2 + h) * 8192, gtab_lds, EPI == EPI_GEGLU && gtab != nullptr);
}
#undef OW_ACC
#undef OW_ACC8
#undef OW_FR
#undef OW_FR8

// ------------------------------------------------------------------------------------------------
// Persistent form of the kernel above ("owp"): at most one block per CU walks its XCD's tiles, and the K-tile stream does not stop at a tile boundary -
// the last two K-tiles of a tile fetch K-tiles 0 / 1 of the block's NEXT tile and the last one reads its first fragments, so the next tile's operands
// travel (and its prologue's memory round trip passes) while this tile's epilogue runs; the accumulators of a tile start from the MFMA's inline-constant
// zero (no zeroing pass).  What it removes per tile: the prologue (~3 k clocks of exposed latency), the block relaunch, the accumulator initialisation.
// The operand buffers stay busy through the epilogue, so its staging area is the 32 KiB the two buffers leave of the CU's 160 KiB (8 KiB per wave):
// epilogues that stage more (fp32, residual, GEGLU) run on the one-tile-per-block kernel.  Needs K >= 192 (three K-tiles).
// ------------------------------------------------------------------------------------------------
#define OWP_OPERANDS(ACC_C, F0_C)                                                                                                                  \
      : OWP_A8(ACC_C, 0), OWP_A8(ACC_C, 8), OWP_F8(F0_C, 0), OWP_F8(F0_C, 8), OWP_F8("=&v", 16), OWP_F8("=&v", 24),                                 \
        "+v"(raA[0]), "+v"(raA[1]), "+v"(raA[2]), "+v"(raA[3]), "+v"(raB[0]), "+v"(raB[1]), "+v"(raB[2]), "+v"(raB[3]),                             \
        "+v"(voA), "+v"(voB), "+s"(sM), "+s"(cnt), "=&s"(so), "+s"(delta)                                                                           \
      : "s"(rsA), "s"(rsB), "s"(stA), "s"(stB), "v"(voAn), "v"(voBn)                                                                                \
      : "memory", "scc"
#define OWP_A(C, n) C(acc[(n) >> 3][((n) >> 1) & 3][(n) & 1])
#define OWP_A8(C, b) OWP_A(C, b), OWP_A(C, b + 1), OWP_A(C, b + 2), OWP_A(C, b + 3), OWP_A(C, b + 4), OWP_A(C, b + 5), OWP_A(C, b + 6), OWP_A(C, b + 7)
#define OWP_F(C, n) C(fr[n])
#define OWP_F8(C, b) OWP_F(C, b), OWP_F(C, b + 1), OWP_F(C, b + 2), OWP_F(C, b + 3), OWP_F(C, b + 4), OWP_F(C, b + 5), OWP_F(C, b + 6), OWP_F(C, b + 7)
template <int EPI>
__global__ __launch_bounds__(256, 1) void gemm_nt_owp_kernel(GemmNT p, int stagger, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16* lds = (bf16*)smem_raw;                                    // [2 K-tiles][A 256 rows x 64 | B 256 rows x 64] + 4 x 8 KiB of epilogue staging
  // block b runs on XCD b % 8 (private L2): every XCD owns a contiguous run of tile ids and deals them round-robin to its blocks, so the blocks
  // of an XCD work on neighbouring tiles (shared operand panels) at any time
  const int per = (int)gridDim.x >> 3, chunk = (ntiles + 7) >> 3;
  int tile = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
  const int t_end = min(((int)(blockIdx.x & 7) + 1) * chunk, ntiles);
  if (tile >= t_end) return;
  dephase_first_round(stagger, 256);
  const int t = threadIdx.x, l = t & 63, hi = l >> 5;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = w >> 1, wc = w & 1;
  const int ntn = (p.N + BN2 - 1) / BN2;
  const uint64_t baseA = (uint64_t)(uintptr_t)p.A, baseB = (uint64_t)(uintptr_t)p.B;
  u32x4 rsA, rsB;
  rsA[0] = __builtin_amdgcn_readfirstlane((uint32_t)baseA); rsA[1] = __builtin_amdgcn_readfirstlane((uint32_t)(baseA >> 32) & 0xffffu);
  rsA[2] = __builtin_amdgcn_readfirstlane(((uint32_t)(p.M - 1) * (uint32_t)p.lda + (uint32_t)p.K) * 2u); rsA[3] = 0x00020000u;
  rsB[0] = __builtin_amdgcn_readfirstlane((uint32_t)baseB); rsB[1] = __builtin_amdgcn_readfirstlane((uint32_t)(baseB >> 32) & 0xffffu);
  rsB[2] = __builtin_amdgcn_readfirstlane(((uint32_t)(p.N - 1) * (uint32_t)p.ldb + (uint32_t)p.K) * 2u); rsB[3] = 0x00020000u;
  const uint32_t stA = __builtin_amdgcn_readfirstlane(32u * (uint32_t)p.lda), stB = __builtin_amdgcn_readfirstlane(32u * (uint32_t)p.ldb);
  const uint32_t lds0 = (uint32_t)(size_t)(lds_void_t*)lds;
  uint32_t sM = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)((w >> 1) * 16384 + (w & 1) * 1024));
  const uint32_t nk = __builtin_amdgcn_readfirstlane((uint32_t)(p.K / BK));
  uint32_t cnt = nk, so, delta = 65536u;
  uint32_t raA[4], raB[4];
  // fragment-read addresses in buffer `par` and the lane offsets of K-tile `kt` of tile (m0, n0): formed afresh for every asm statement (a dozen vector
  // instructions) - kept alive across the epilogue they, and whatever else hipcc hoists above the K loop, spill and the reloads wait on the in-flight DMAs
  auto lane_state = [&](int par, int m0_, int n0_, int kt, uint32_t& voA_, uint32_t& voB_) {
    int lx = l;
    asm volatile("" : "+v"(lx));                                    // opaque lane id: recomputed here, not carried (in scratch) around the tile loop
    const int hx = lx >> 5;
    const uint32_t rowx = (uint32_t)((w >> 1) * 128 + 8 * (w & 1) + (lx >> 3));      // (DMA pieces / lane offsets: see gemm_nt_ow_kernel)
    const uint32_t chunkx = (uint32_t)((lx & 7) ^ ((4 * (w & 1) + (lx >> 4)) & 7)) * 8u;
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      const uint32_t slot = (uint32_t)(((2 * ks + hx) ^ ((lx >> 1) & 7)) << 4);
      raA[ks] = lds0 + (uint32_t)(par * 65536 + (wr * 128 + (lx & 31)) * 128) + slot;
      raB[ks] = lds0 + (uint32_t)(par * 65536 + BM2 * BK * 2 + (wc * 128 + (lx & 31)) * 128) + slot;
    }
    voA_ = (((uint32_t)m0_ + rowx) * (uint32_t)p.lda + chunkx + (uint32_t)(kt * BK)) * 2u;
    voB_ = (((uint32_t)n0_ + rowx) * (uint32_t)p.ldb + chunkx + (uint32_t)(kt * BK)) * 2u;
  };
  bf16* stage = lds + 2 * (BM2 + BN2) * BK + w * 4096;             // this wave's 8 KiB behind the operand buffers
  int m0 = (tile / ntn) * BM2, n0 = (tile % ntn) * BN2;
  uint32_t voA, voB, voAn = 0, voBn = 0;
  f32x16 acc[2][4][2];
  u32x4 fr[32];
  lane_state(0, m0, n0, 0, voA, voB);
  asm volatile(
#include "gemm_nt_owp_pro.inc"
      OWP_OPERANDS("=&a", "=&v"));
  // (scalar asm outputs that feed the next asm statement go through readfirstlane: the compiler's divergence analysis treats asm results as per-lane values)
  sM = __builtin_amdgcn_readfirstlane(sM); delta = __builtin_amdgcn_readfirstlane(delta);
  while (true) {
    const int nxt = tile + per;
    const bool has_next = nxt < t_end;                              // block-uniform
    cnt = nk;
    lane_state(delta == 65536u ? 0 : 1, m0, n0, 2, voA, voB);       // delta > 0: the tile's K-tile 0 sits in buffer 0
    if (has_next) {
      const int m1 = (nxt / ntn) * BM2, n1 = (nxt % ntn) * BN2;
      voAn = voA + (uint32_t)((m1 - m0) * p.lda - 2 * BK) * 2u; voBn = voB + (uint32_t)((n1 - n0) * p.ldb - 2 * BK) * 2u;     // (voA / voB stand at K-tile 2 of this tile)
      asm volatile(
#include "gemm_nt_owp_next.inc"
          OWP_OPERANDS("=&a", "=&v"));
    } else {
      voAn = voA; voBn = voB;                                       // (unused by this variant)
      asm volatile(
#include "gemm_nt_owp_last.inc"
          OWP_OPERANDS("=&a", "=&v"));
    }
    sM = __builtin_amdgcn_readfirstlane(sM); delta = __builtin_amdgcn_readfirstlane(delta);
    int me = m0, ne = n0, le = l;
    asm volatile("" : "+s"(me), "+s"(ne), "+v"(le));                // opaque: nothing of the epilogue's address arithmetic moves above the K loop or out of the tile loop
#pragma unroll
    for (int h = 0; h < 2; h++)
      nt_epilogue<EPI, 4>(p, acc[h], me + wr * 128, ne + wc * 128 + h * 64, stage, nullptr, false, le & 63);
    if (!has_next) break;
    tile = nxt; m0 = (tile / ntn) * BM2; n0 = (tile % ntn) * BN2;
  }
}
#undef OWP_OPERANDS
#undef OWP_A
#undef OWP_A8
#undef OWP_F
#undef OWP_F8

// TN with LDS-DMA (round 3: one template for every tiling).  A wave owns (32 FA) x (32 FB) of the output, WN x WK waves share a block tile
// of (32 FA WN) x (32 FB WK); operands arrive as 32-row slabs of 128-column sub-slabs (swizzled [32][128] layout above) in an NST-slot ring,
// NST - 1 slabs in flight, one barrier per slab.  Instantiated as
//   <2,2,2,2,4>  128 x 128, 4 waves, 16 KiB slabs, 64 KiB  -> 2 blocks per CU   (round 2's kernel; few-tile products: 512 x 512)
//   <2,4,4,2,4>  256 x 256, 8 waves, 32 KiB slabs, 128 KiB -> 1 block per CU    (7.8 KiB of L2 -> LDS traffic per MFLOP instead of 15.6)
// What the counters and stamps said (profiles/r03_c_*): no bank conflicts and 3.5 % LDS issue stalls with either fragment ratio (1.0 or 0.75
// transposed reads per MFMA) - the loop was neither LDS- nor MFMA-bound; three tilings (also 128 x 256 with 4 waves) ran within 2 % of each
// other until the DMA issue was spread over the MFMAs (see the slab loop), after which the wide tiles lead: 2816 x 512 at M = 65536
// 713 -> 810 TFLOP/s, 4096 x 1024 830 -> 1140.
// Fragments that lie wholly outside N / k_valid skip their MFMAs (wave-uniform), so the ragged last tile of N = 1544 costs its DMA only.
constexpr int MW_ROWS = 32;
template <bool SUM, int FA, int FB, int WN, int WK, int NST>
__global__ __launch_bounds__(64 * WN * WK, 2) void gemm_tn_wide_kernel(GemmTN p) {
  constexpr int NW = WN * WK;
  constexpr int TILE_N = 32 * FA * WN, TILE_K = 32 * FB * WK;
  constexpr int SA = TILE_N / 128, SB = TILE_K / 128, NSUB = SA + SB;   // 128-column sub-slabs per stage
  constexpr int SUBE = MW_ROWS * 128, STAGE = NSUB * SUBE;               // elements
  constexpr int NPW = NSUB * 8 / NW;                                     // DMA pieces (4 rows x 256 B) per wave and slab
  static_assert(NSUB * 8 % NW == 0 && TILE_N % 128 == 0 && TILE_K % 128 == 0, "tile shape");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16* S = (bf16*)smem_raw;                                             // [NST][NSUB][32 * 128]

  const int t = threadIdx.x, l = t & 63, hi = l >> 5;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wn = w / WK, wk = w % WK;
  const int ntn = (p.N + TILE_N - 1) / TILE_N, ntk = (p.K + TILE_K - 1) / TILE_K;
  const int ntile = ntn * ntk;
  const TnBlock blk = tn_block(p, ntile);
  const int bid = blk.tile, mbeg = blk.mbeg, mend = blk.mend;
  const int n0 = (bid / ntk) * TILE_N, k0 = (bid % ntk) * TILE_K;
  if (mbeg >= mend) return;
  const int nst = (mend - mbeg) / MW_ROWS;

  // piece q = w * NPW + j of a slab: sub-slab q / 8 (A's first), rows 4 (q % 8) ... + 3
  const bf16* gp[NPW];
  size_t gstep[NPW];
  int loff[NPW];
#pragma unroll
  for (int j = 0; j < NPW; j++) {
    const int q = w * NPW + j, sub = q >> 3, g = q & 7;
    const int row = g * 4 + (l >> 4);
    const int c = (l & 15) ^ ((row & 3) << 2);
    if (sub < SA) {
      gp[j] = p.A + (size_t)(mbeg + row) * p.lda + min(n0 + sub * 128 + c * 8, p.a_cols - 8);
      gstep[j] = (size_t)MW_ROWS * p.lda;
    } else {
      gp[j] = p.B + (size_t)(mbeg + row) * p.ldb + min(k0 + (sub - SA) * 128 + c * 8, p.b_cols - 8);
      gstep[j] = (size_t)MW_ROWS * p.ldb;
    }
    loff[j] = sub * SUBE + g * 4 * 128;
  }
  auto issue = [&](int buf) {
#pragma unroll
    for (int j = 0; j < NPW; j++) {
      glds16_asm(gp[j], S + buf * STAGE + loff[j]);
      gp[j] += gstep[j];
    }
  };
  static_assert(NPW <= FA * FB, "one DMA piece per MFMA pair");

  bool live_a[FA], live_b[FB];                                          // wave-uniform: the fragment holds at least one written row / column
#pragma unroll
  for (int i = 0; i < FA; i++) live_a[i] = n0 + (wn * FA + i) * 32 < p.N;
#pragma unroll
  for (int j = 0; j < FB; j++) live_b[j] = k0 + (wk * FB + j) * 32 < p.k_valid;

  f32x16 acc[FA][FB];
#pragma unroll
  for (int i = 0; i < FA; i++)
#pragma unroll
    for (int j = 0; j < FB; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  const bool do_sum = SUM && k0 == 0 && wk == 0;                       // bias gradient: A^T x ones on the waves of the first K columns
  f32x16 accs[SUM ? FA : 1];
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; e++) ones[e] = f2bf(1.f);
#pragma unroll
  for (int i = 0; i < (SUM ? FA : 1); i++)
#pragma unroll
    for (int r = 0; r < 16; r++) accs[i][r] = 0.f;

#ifdef TFX_TN_TIMING      // tools/tn_timing.py: per-block stamps through `a_rowmap` (0 start, 1 first slab landed, 2 loop done, 3 atomics issued, 4 retired,
                          // 5 / 6 / 7 = cycles summed over the loop in [waitcnt + barrier] / [DMA issue] / [fragment reads + MFMA issue])
  unsigned long long* stamps = (unsigned long long*)p.a_rowmap + (size_t)blockIdx.x * 8;
  unsigned long long tw = 0, ti = 0, tc = 0, c0 = 0, c1 = 0, c2 = 0;
#define TN_STAMP(i) { if (t == 0) stamps[i] = __builtin_readcyclecounter(); }
#define TN_CLK(x) x = __builtin_readcyclecounter();
#else
#define TN_STAMP(i)
#define TN_CLK(x)
#endif
  TN_STAMP(0)
  const int pre = min(nst, NST - 1);
  for (int s0 = 0; s0 < pre; s0++) issue(s0);
  int buf = 0, nbuf = NST - 1;                                          // slab st lives in `buf`; slab st + NST - 1 goes to `nbuf`
  // One slab.  MORE: slab st + NST - 1 exists and is fetched here (every iteration but the last NST - 1); CHECK: some fragment of this wave lies
  // outside N / k_valid and its MFMAs are skipped.  Both are template-like flags so that the common loop body is ONE basic block: with a branch
  // per MFMA hipcc gathers the conditional DMA issues back into a burst (sched_barrier only orders within a block).
  auto slab = [&](int st, auto more_c, auto check_c) {
    constexpr bool MORE = decltype(more_c)::value, CHECK = decltype(check_c)::value;
    TN_CLK(c0)
#ifdef TFX_TN_TIMING
    if (st > 0) tc += c0 - c2;
#endif
    if (MORE) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * NPW) : "memory");
    else {
      const int ahead = min(NST - 2, nst - 1 - st);                     // slabs issued beyond `st` at this point (NPW DMAs each)
      if (NST >= 4 && ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPW) : "memory");
      else if (ahead >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                                       // slab st landed for every wave; slab st - 1's slot is free
    TN_CLK(c1)
#ifdef TFX_TN_TIMING
    tw += c1 - c0;
    if (st == 0) TN_STAMP(1)
#endif
    TN_CLK(c2)
#ifdef TFX_TN_TIMING
    ti += c2 - c1;
#endif
    const bf16* sa = S + buf * STAGE;
    const bf16* sb = sa + SA * SUBE;
    bf16x8 af[2][FA], bfr[2][FB];
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
#pragma unroll
      for (int i = 0; i < FA; i++) {
        const int c = (wn * FA + i) * 32;
        af[ks][i] = lds_tr8_swz(sa + (c >> 7) * SUBE, 16 * ks + 8 * hi, 16 * ks + 8 * hi + 4, c & 127);
      }
#pragma unroll
      for (int j = 0; j < FB; j++) {
        const int c = (wk * FB + j) * 32;
        bfr[ks][j] = lds_tr8_swz(sb + (c >> 7) * SUBE, 16 * ks + 8 * hi, 16 * ks + 8 * hi + 4, c & 127);
      }
    }
    // The next slab's DMA pieces go out one per MFMA pair, not as a burst behind the barrier.  Phase stamps of the burst form (tools/tn_timing.py,
    // profiles/r03_c_tn_timing.txt, 2816 x 512): of 1570 clocks per slab and wave, 570 passed in the six `global_load_lds` issues - every wave
    // of the CU queues its pieces at the address unit at the same moment (64 B/clk, 16 clocks per 1 KiB piece) and feeds no MFMA meanwhile.
#pragma unroll
    for (int ks = 0; ks < 2; ks++)
#pragma unroll
      for (int i = 0; i < FA; i++)
#pragma unroll
        for (int j = 0; j < FB; j++) {
          if (!CHECK || (live_a[i] && live_b[j])) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][i], bfr[ks][j], acc[i][j], 0, 0, 0);
          const int idx = (ks * FA + i) * FB + j;
          if (MORE && (idx & 1) && (idx >> 1) < NPW) {
            glds16_asm(gp[idx >> 1], S + nbuf * STAGE + loff[idx >> 1]);
            gp[idx >> 1] += gstep[idx >> 1];
            __builtin_amdgcn_sched_barrier(0);
          }
        }
    if (do_sum) {
#pragma unroll
      for (int ks = 0; ks < 2; ks++)
#pragma unroll
        for (int i = 0; i < (SUM ? FA : 1); i++) accs[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][i], ones, accs[i], 0, 0, 0);
    }
    buf = buf + 1 == NST ? 0 : buf + 1;
    nbuf = nbuf + 1 == NST ? 0 : nbuf + 1;
  };
  bool all_live = true;
#pragma unroll
  for (int i = 0; i < FA; i++) all_live = all_live && live_a[i];
#pragma unroll
  for (int j = 0; j < FB; j++) all_live = all_live && live_b[j];
  const int n_main = max(0, nst - (NST - 1));
  using yes = std::true_type; using no = std::false_type;
  if (all_live) {
    for (int st = 0; st < n_main; st++) slab(st, yes{}, no{});
    for (int st = n_main; st < nst; st++) slab(st, no{}, no{});
  } else {
    for (int st = 0; st < n_main; st++) slab(st, yes{}, yes{});
    for (int st = n_main; st < nst; st++) slab(st, no{}, yes{});
  }
  TN_STAMP(2)

  // Output rows first, atomics after: loads, stores and atomics retire through the one in-order vmcnt, so a row-map load between two groups of
  // atomics (`s_waitcnt vmcnt(0)` before its use) made every group wait for the previous group's round trip to L2 - 32 serial round trips,
  // 47-64 k clocks per block = 13-29 % of the kernel in the phase stamps (profiles/r03_c_tn_timing.txt).
  int out_row[FA][16];
#pragma unroll
  for (int i = 0; i < FA; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int n = n0 + (wn * FA + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      out_row[i][r] = n >= p.N ? -1 : p.rowmap ? p.rowmap[n] : n;
    }
  int out_col[FB];
#pragma unroll
  for (int j = 0; j < FB; j++) {
    const int k = k0 + (wk * FB + j) * 32 + (l & 31);
    out_col[j] = k < p.k_valid ? tn_out_col(p, k) : -1;
  }
#pragma unroll
  for (int i = 0; i < FA; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int no = out_row[i][r];
      if (no < 0) continue;
      if (SUM && do_sum && (l & 31) == 0) atomicAdd(p.colsum + no, accs[SUM ? i : 0][r]);
      float* crow = p.C + (size_t)no * p.ldc;
#pragma unroll
      for (int j = 0; j < FB; j++)
        if (out_col[j] >= 0) atomicAdd(crow + out_col[j], acc[i][j][r] * p.alpha);
    }
  TN_STAMP(3)
#ifdef TFX_TN_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TN_STAMP(4)
  if (t == 0) { unsigned long long e = __builtin_readcyclecounter(); tc += stamps[2] - c2; stamps[5] = tw; stamps[6] = ti; stamps[7] = tc; (void)e; }
#endif
#undef TN_STAMP
#undef TN_CLK
}

// ------------------------------------------------------------------------------------------------
// One-wave-per-SIMD form of the 256 x 256 weight-gradient kernel ("tn ow", round 5): the plan of gemm_nt_ow_kernel - 4 waves x 128 x 128, 256 accumulator
// AGPRs, the fragments of a whole 64-row step in registers (here FIXED registers v[64:191]: a fragment is two transposed 8-byte reads into the halves of
// one MFMA operand), two 64 KiB LDS buffers fed by `buffer_load_dwordx4 ... lds` with two steps in flight, two barriers per step - in ONE generated
// inline-asm block (gemm_tn_ow_loop.inc, tools/gen_tn_ow_loop.py).  Per MFMA it reads 0.5 KiB of fragments from LDS where the 8-wave kernel above reads
// 0.75 KiB.  Same sub-slab layout, same accumulation order (rows in order, 16 per MFMA), same split-M atomics.  No dead-fragment skipping (ragged N / K
// tiles compute on whatever the next rows hold and drop it at the store); the folded bias gradient is a second pair of loop forms (gemm_tn_ow_sum01 / sum23.inc)
// taken by the waves of the first K columns.
// Needs chunks of >= 192 rows in multiples of 64.
// ------------------------------------------------------------------------------------------------
#define OWT_OUT                                                                                                                                     \
        "=&a"(acc[0][0]), "=&a"(acc[0][1]), "=&a"(acc[0][2]), "=&a"(acc[0][3]), "=&a"(acc[1][0]), "=&a"(acc[1][1]), "=&a"(acc[1][2]), "=&a"(acc[1][3]),    \
        "=&a"(acc[2][0]), "=&a"(acc[2][1]), "=&a"(acc[2][2]), "=&a"(acc[2][3]), "=&a"(acc[3][0]), "=&a"(acc[3][1]), "=&a"(acc[3][2]), "=&a"(acc[3][3]),    \
        "+v"(raA[0]), "+v"(raA[1]), "+v"(raA[2]), "+v"(raA[3]), "+v"(raB[0]), "+v"(raB[1]), "+v"(raB[2]), "+v"(raB[3]),                                     \
        "+v"(voA), "+v"(voB), "+s"(sM), "+s"(cnt), "=&s"(so), "+s"(delta)
#define OWT_IN "s"(rsA), "s"(rsB), "s"(stA), "s"(stB), "s"(ksA), "s"(ksB)
#define OWT_CLOBBER                                                                                                                                 \
        "memory", "scc",                                                                                                                            \
        "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79",                             \
        "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95",                             \
        "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111",                 \
        "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127",             \
        "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143",             \
        "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159",             \
        "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175",             \
        "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191"
template <bool SUM>
TFX_DEV void tn_ow_body(const GemmTN& p, const TnBlock& blk) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16* S = (bf16*)smem_raw;                                             // [2 steps][A sub-slabs 0, 1 | B sub-slabs 0, 1][64 rows x 128 columns]
  const int t = threadIdx.x, l = t & 63, hi = l >> 5;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wn = w >> 1, wk = w & 1;
  const int ntk = (p.K + 255) / 256;
  const int mbeg = blk.mbeg, mend = blk.mend;
  const int n0 = (blk.tile / ntk) * 256, k0 = (blk.tile % ntk) * 256;
  if (mend - mbeg < 192) return;                                         // (the launcher only sends chunks of >= 192 rows in multiples of 64)
  // DMA pieces of a wave: sub-slab (w & 1) of each operand, rows 32 (w >> 1) + 4 j + (l >> 4), j = 0 .. 7; the lane fetches the 16-byte chunk its LDS slot holds
  const uint32_t prow = (uint32_t)((w >> 1) * 32 + (l >> 4));
  const uint32_t pchunk = (uint32_t)((l & 15) ^ (((l >> 4) & 3) << 2));
  uint32_t voA = (((uint32_t)mbeg + prow) * (uint32_t)p.lda + (uint32_t)(n0 + (w & 1) * 128) + pchunk * 8u) * 2u;
  const uint32_t ldb = (uint32_t)p.ldb, bcols = (uint32_t)p.b_cols, kb = (uint32_t)k0;
  uint32_t voB = (((uint32_t)mbeg + prow) * ldb + kb + (uint32_t)((w & 1) * 128) + pchunk * 8u) * 2u;
  const uint64_t baseA = (uint64_t)(uintptr_t)p.A, baseB = (uint64_t)(uintptr_t)p.B;
  u32x4 rsA, rsB;
  rsA[0] = __builtin_amdgcn_readfirstlane((uint32_t)baseA); rsA[1] = __builtin_amdgcn_readfirstlane((uint32_t)(baseA >> 32) & 0xffffu);
  rsA[2] = __builtin_amdgcn_readfirstlane(((uint32_t)(p.M - 1) * (uint32_t)p.lda + (uint32_t)p.a_cols) * 2u); rsA[3] = 0x00020000u;
  rsB[0] = __builtin_amdgcn_readfirstlane((uint32_t)baseB); rsB[1] = __builtin_amdgcn_readfirstlane((uint32_t)(baseB >> 32) & 0xffffu);
  rsB[2] = __builtin_amdgcn_readfirstlane(((uint32_t)(p.M - 1) * ldb + bcols) * 2u); rsB[3] = 0x00020000u;
  const uint32_t stA = __builtin_amdgcn_readfirstlane(8u * (uint32_t)p.lda), stB = __builtin_amdgcn_readfirstlane(8u * ldb);          // 4 rows
  const uint32_t ksA = __builtin_amdgcn_readfirstlane(128u * (uint32_t)p.lda), ksB = __builtin_amdgcn_readfirstlane(128u * ldb);     // 64 rows
  const uint32_t lds0 = (uint32_t)(size_t)(lds_void_t*)S;
  uint32_t sM = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)((w & 1) * 16384 + (w >> 1) * 8192));
  uint32_t cnt = __builtin_amdgcn_readfirstlane((uint32_t)((mend - mbeg) / 64)), so, delta = 65536u;
  // fragment reads (lds_tr8_swz): 16-lane group g = l >> 4 reads the 4 rows x 16 columns block at rows 8 (g >> 1) + ..., columns 16 (g & 1) + ...; its lane
  // q = l & 15 addresses row q >> 2, 4-column piece q & 3.  Column block i of the wave's sub-slab: chunk 4 i + 2 (g & 1) + ((q & 3) >> 1), XOR ((row & 3) << 2).
  uint32_t raA[4], raB[4];
  {
    const int g = l >> 4, q = l & 15;
    const int row = 8 * (g >> 1) + (q >> 2);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int ch = (4 * i + 2 * (g & 1) + ((q & 3) >> 1)) ^ ((row & 3) << 2);
      const uint32_t off = (uint32_t)(row * 256 + ch * 16 + (q & 1) * 8);
      raA[i] = lds0 + (uint32_t)(wn * 16384) + off;
      raB[i] = lds0 + 32768u + (uint32_t)(wk * 16384) + off;
    }
  }
  f32x16 acc[4][4];
  // bias gradient: A^T x ones in the blocks of the first K columns, shared by the two waves that hold the same A fragments - wk = 0 sums its A blocks 0, 1, wk = 1
  // blocks 2, 3 (8 more MFMAs per step each instead of 16 on one wave: the block's pace is its slowest wave's).  Every loop form passes the same barriers.
  f32x16 accs[2];
  const bool do_sum = SUM && p.colsum != nullptr && k0 == 0;             // block-uniform; the wave's half by wk
  if (do_sum) {
    u32x4 ones = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    if (wk == 0) {
      asm volatile(
#include "gemm_tn_ow_sum01.inc"
          : OWT_OUT, "=&v"(accs[0]), "=&v"(accs[1])
          : OWT_IN, "v"(ones)
          : OWT_CLOBBER);
    } else {
      asm volatile(
#include "gemm_tn_ow_sum23.inc"
          : OWT_OUT, "=&v"(accs[0]), "=&v"(accs[1])
          : OWT_IN, "v"(ones)
          : OWT_CLOBBER);
    }
  } else {
    asm volatile(
#include "gemm_tn_ow_loop.inc"
        : OWT_OUT
        : OWT_IN
        : OWT_CLOBBER);
  }

  // output rows first, atomics after (see gemm_tn_wide_kernel): D[n][k], lane = column k, register r = row (r & 3) + 8 (r >> 2) + 4 hi
  int out_col[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int k = k0 + (wk * 4 + j) * 32 + (l & 31);
    out_col[j] = k < p.k_valid ? tn_out_col(p, k) : -1;
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    int out_row[16];                                                     // (per 32-row block: 16 row-map loads in flight, then its 64 atomics)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int n = n0 + (wn * 4 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      out_row[r] = n >= p.N ? -1 : p.rowmap ? p.rowmap[n] : n;
    }
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int no = out_row[r];
      if (no < 0) continue;
      if (SUM && do_sum && (i >> 1) == wk && (l & 31) == 0) atomicAdd(p.colsum + no, accs[i & 1][r]);
      float* crow = p.C + (size_t)no * p.ldc;
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (out_col[j] >= 0) atomicAdd(crow + out_col[j], acc[i][j][r] * p.alpha);
    }
  }
}
template <bool SUM>
__global__ __launch_bounds__(256, 1) void gemm_tn_ow_kernel(GemmTN p, int ramp) {
  const int ntn = (p.N + 255) / 256, ntk = (p.K + 255) / 256;
  tn_ow_body<SUM>(p, tn_block_ramp(p, ntn * ntk, ramp));
}
// grouped launch (tfx.h group_next): up to TN_GROUP_MAX products over the same M rows share one grid; the (row chunk, tile) pairs are numbered over the CONCATENATED tile
// lists, so an XCD still walks all tiles of a chunk side by side; a block finds its product by its tile number (block-uniform) and runs the body on it
constexpr int TN_GROUP_MAX = 6;
struct TnGroup { int count; int tile_end[TN_GROUP_MAX]; GemmTN p[TN_GROUP_MAX]; };
template <bool SUM>
__global__ __launch_bounds__(256, 1) void gemm_tn_ow_group_kernel(TnGroup g, int ramp) {
  TnBlock blk = tn_block_ramp(g.p[0], g.tile_end[g.count - 1], ramp);    // (M and splits of the head)
  int pid = 0, first = 0;
#pragma unroll
  for (int i = 0; i < TN_GROUP_MAX - 1; i++)
    if (i + 1 < g.count && blk.tile >= g.tile_end[i]) { pid = i + 1; first = g.tile_end[i]; }
  blk.tile -= first;
  tn_ow_body<SUM>(g.p[__builtin_amdgcn_readfirstlane(pid)], blk);        // (uniform index into the kernel-argument segment: scalar loads)
}
#undef OWT_OUT
#undef OWT_IN
#undef OWT_CLOBBER

// split count of a TN launch with `tiles` output tiles over M rows (tfx.h: splits == 0): the smallest count that puts `fill` x (resident block
// slots) blocks on the chip, never more than one round of blocks.  Measured in the training step (profiles/r03_c_tn_splits.txt, one stream):
// 256 x 256 tiles - 2816 x 512 (22 tiles) 256 / 225 / 220 / 215 us at 154 / 198 / 220 / 242 blocks and 341 us at 264 (a second round);
// 1544 x 512 and 512 x 1408 flat (144 us) from 216 blocks on; 128 x 128 tiles (512 slots) - 512 x 512 (16 tiles) 63 / 60 / 74 / 66 / 68 us at
// 208 / 256 / 320 / 384 / 512 blocks.  Past the knee more splits only add fp32 atomics: the kernels run against the power limit, not against
// idle CUs.  Chunks stay >= 256 rows.
static int tn_auto_splits(int M, int tiles, int kind) {
  constexpr double fill2 = 0.9, fill0 = 0.5;   // 256 x 256 tiles (256 slots) / the 4-wave tilings (512 slots); swept in round 3 (profiles/r03_c_tn_splits.txt)
  const int slots = kind == 2 ? 256 : 512;
  const double want = (kind == 2 ? fill2 : fill0) * slots;
  int s = 1;
  while (tiles * s < want && tiles * (s + 1) <= slots && s < 64 && M / (s + 1) >= 256) s++;
  return s;
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static bool use_glds() {             // TFX_GEMM_GLDS=0 forces the register-staged fallback kernels (used by the tests to cover them)
  static int v = -1;
  if (v < 0) { const char* e = getenv("TFX_GEMM_GLDS"); v = (e && e[0] == '0') ? 0 : 1; }
  return v == 1;
}

// which NT kernel a shape runs on, and its grid (pure host logic: tfx.h tfx_gemm_nt_plan, pinned by the CPU tests):
//   5 decode    M <= 1024 rows (decode steps: samples x 1 ... x (modality length + 1) rows - 640 at SURVEY 8(d) config 5), K split across the waves,
//               as long as the 64 x 64 tiles fit the chip in one round (128 KiB of LDS = one block per CU; 344 tiles at M = 256, N = 5504 measured
//               19.9 us against 13.7 for the 64 x 128 kernel)
//   4 skinny    few row blocks: latency-bound, deep DMA ring (gemm_nt_skinny_kernel)
//   3 ping-pong 256 x 256 tiles whenever they still fill the chip (>= 2 tiles per CU); a ragged last N tile costs less than the 128 x 128 kernel
//               loses (measured on N = 1544 / 1408).  One tile per block: a persistent walk with cross-tile prefetch measured no faster
//   2 mid       at most one 128 x 128 tile per CU and K >= 256
//   1 glds      128 x 128 tiles, LDS-DMA, two stages
//   0 register-staged fallback: N % 4 != 0 (the LDS-DMA kernels' pipelined epilogue stores whole 4-column groups) or TFX_GEMM_GLDS=0
//   6 one-wave  the 256 x 256 family on the one-wave-per-SIMD kernel, one tile per block (fp32 outputs from K = 1024; bf16 outputs with K < 192)
//   7 one-wave, persistent (bf16 outputs, K >= 192)
enum { NT_FALLBACK = 0, NT_GLDS = 1, NT_MID = 2, NT_PP = 3, NT_SKINNY = 4, NT_DECODE = 5, NT_OW = 6, NT_OWP = 7 };

// device copy of the GELU grid of the GEGLU-forward epilogue (geglu_uvh_grid): built once per process and device in double precision.
// TFX_GELU_TABLE=0 keeps the polynomial form (A/B, tests).  Returns nullptr when disabled or when the allocation fails (the epilogue then evaluates the
// polynomial: same results to ~1e-7).  Guarded by a mutex: two host threads (or two devices driven from one process) may reach the first launch together.
static const float* gelu_table() {
  static const float* tab[16] = {nullptr};
  static bool tried[16] = {false};
  static std::mutex mu;
  static int enabled = -1;
  std::lock_guard<std::mutex> lock(mu);
  if (enabled < 0) { const char* e = getenv("TFX_GELU_TABLE"); enabled = e ? atoi(e) : 1; }
  if (!(enabled & 1)) return nullptr;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  if (!tried[dev]) {
    tried[dev] = true;
    std::vector<float> h(2 * GTAB_N);
    auto Phi = [](double g) { return 0.5 * erfc(-g * 0.70710678118654752440); };
    auto phi = [](double g) { return 0.39894228040143267794 * exp(-0.5 * g * g); };
    const double st = 1.0 / 128.0;
    for (int i = 0; i < 2048; i++) {
      const double g = (i - 1024) * st;
      h[4 * i] = (float)(g * Phi(g));                                   // gelu
      h[4 * i + 1] = (float)((Phi(g) + g * phi(g)) * st);               // h x first derivative
      h[4 * i + 2] = (float)(0.5 * phi(g) * (2.0 - g * g) * st * st);   // h^2 x half the second derivative (= phi (2 - g^2))
      h[4 * i + 3] = 0.f;
    }
    float* d = nullptr;
    if (hipMalloc(&d, h.size() * 4) == hipSuccess && hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice) == hipSuccess) tab[dev] = d;
  }
  return tab[dev];
}
struct NtPlan { int kind, grid; };
static bool nt_ow_takes(const GemmNT& p);
static bool nt_owp_takes(const GemmNT& p);
static NtPlan nt_plan(const GemmNT& p) {
  const int grid = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  const int t256 = ((p.M + BM2 - 1) / BM2) * ((p.N + BN2 - 1) / BN2);
  const bool dma = use_glds() && (p.N & 3) == 0;
  const int grid_sd = ((p.M + 63) / 64) * ((p.N + 63) / 64);
  const int grid_sk = ((p.M + SK_BM - 1) / SK_BM) * ((p.N + SK_BN - 1) / SK_BN);
  if (dma && p.M <= 1024 && p.K % SD_BK == 0 && p.K >= 8 * SD_BK && grid_sd <= 256) return {NT_DECODE, grid_sd};
  if (dma && (p.M <= 512 || (p.M <= 1024 && grid_sk <= 256))) return {NT_SKINNY, grid_sk};
  static int pp_min = -1;             // TFX_NT_PP_MIN: fewest 256 x 256 tiles that still take the 256 x 256 family (tests, probe; default 512 = two per CU)
  if (pp_min < 0) { const char* e = getenv("TFX_NT_PP_MIN"); pp_min = e ? atoi(e) : 512; }
  if (dma && t256 >= pp_min) return {!nt_ow_takes(p) ? NT_PP : nt_owp_takes(p) ? NT_OWP : NT_OW, t256};
  if (dma && grid <= 256 && p.K >= 4 * BK) return {NT_MID, grid};
  return {dma ? NT_GLDS : NT_FALLBACK, grid};
}

constexpr int PP_STAGGER = 12000;       // de-phasing delay of the first round's odd blocks in shader clocks (dephase_first_round; swept in rounds 2, 3, 5)
template <int EPI> static void launch_pp(const GemmNT& p, int grid, hipStream_t s) {
  static uint32_t attr_pp = 0;
  const float* gtab = EPI == EPI_GEGLU ? gelu_table() : nullptr;
  const int smem2 = 2 * (BM2 * BK + BN2 * BK) * 2 + (gtab ? GTAB_N * 8 : 0);       // GEGLU forward / backward: + the 32 KiB GELU table = all 160 KiB of the CU
  ensure_smem_attr((const void*)gemm_nt_pp_kernel<EPI>, 2 * (BM2 * BK + BN2 * BK) * 2 + GTAB_N * 8, attr_pp);
  hipLaunchKernelGGL(gemm_nt_pp_kernel<EPI>, dim3(grid), dim3(512), smem2, s, p, PP_STAGGER, gtab);
}
// TFX_NT_OW=0: the ping-pong kernel for every launch of the 256 x 256 family (A/B, tests).  Default: the one-wave-per-SIMD kernels where they measured faster
// inside the training step (bf16 outputs; fp32 outputs from K = 1024).  Their loops read both operands through raw buffer resources with 32-bit offsets and
// take neither a row-gathered nor a split A.  (The other epilogues on four waves measured slower and two instantiations spilled: removed in round 6.)
static int nt_ow_mode() {
  static int ow = -1;
  if (ow < 0) { const char* e = getenv("TFX_NT_OW"); ow = e ? atoi(e) : 1; }
  return ow;
}
static bool nt_ow_takes(const GemmNT& p) {
  const int ow = nt_ow_mode();
  if (ow == 0 || p.a_rowmap || p.A2) return false;
  if (!(p.epi == EPI_BF16 || (p.epi == EPI_F32 && p.K >= 1024))) return false;
  if ((((uintptr_t)p.A | (uintptr_t)p.B) & 15) != 0) return false;
  const long long lim = 1ll << 32;
  return ((long long)p.M + 2 * BM2) * p.lda * 2 < lim && ((long long)p.N + 2 * BN2) * p.ldb * 2 < lim;
}
// TFX_NT_OWP=0: every shape of the one-wave-per-SIMD family on the one-tile-per-block kernel (A/B)
static bool nt_owp_takes(const GemmNT& p) {
  static int owp = -1;
  if (owp < 0) { const char* e = getenv("TFX_NT_OWP"); owp = e ? atoi(e) : 1; }
  return owp != 0 && p.epi == EPI_BF16 && p.K >= 3 * BK && ((p.ldc | p.N) & 7) == 0 && (((uintptr_t)p.C) & 15) == 0;      // (the staged bf16 epilogue: 8 KiB per wave)
}
template <int EPI> static void launch_owp(const GemmNT& p, int tiles, hipStream_t s) {
  static uint32_t attr_owp = 0;
  const int smem = 2 * (BM2 * BK + BN2 * BK) * 2 + 4 * 8192;      // all 160 KiB of the CU
  ensure_smem_attr((const void*)gemm_nt_owp_kernel<EPI>, smem, attr_owp);
  static int cus[16] = {0};
  int dev = 0; (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 16) dev = 0;
  if (!cus[dev]) { hipDeviceProp_t pr; cus[dev] = hipGetDeviceProperties(&pr, dev) == hipSuccess ? pr.multiProcessorCount : 256; }
  const int grid = std::min(cus[dev] / 8 * 8, (tiles + 7) / 8 * 8);
  hipLaunchKernelGGL(gemm_nt_owp_kernel<EPI>, dim3(grid), dim3(256), smem, s, p, PP_STAGGER, tiles);
}
template <int EPI> static void launch_ow(const GemmNT& p, int grid, hipStream_t s) {
  static uint32_t attr_ow = 0;
  const float* gtab = EPI == EPI_GEGLU ? gelu_table() : nullptr;
  const int smem = 2 * (BM2 * BK + BN2 * BK) * 2 + (gtab ? GTAB_N * 8 : 0);
  ensure_smem_attr((const void*)gemm_nt_ow_kernel<EPI>, 2 * (BM2 * BK + BN2 * BK) * 2 + GTAB_N * 8, attr_ow);
  hipLaunchKernelGGL(gemm_nt_ow_kernel<EPI>, dim3(grid), dim3(256), smem, s, p, PP_STAGGER, gtab);
}
template <int EPI> static int launch_nt(const GemmNT& p, hipStream_t s) {
  const NtPlan pl = nt_plan(p);
  const int smem = 2 * (BM * BK + BN * BK) * 2;
  switch (pl.kind) {
    case NT_DECODE: {
      static uint32_t attr_sd = 0;
      const int smem_sd = 4 * SD_ST * SD_SLOT * 2;
      ensure_smem_attr((const void*)gemm_nt_decode_kernel<EPI>, smem_sd, attr_sd);
      hipLaunchKernelGGL(gemm_nt_decode_kernel<EPI>, dim3(pl.grid + prefetch_blocks(p)), dim3(256), smem_sd, s, p, pl.grid);
      break;
    }
    case NT_SKINNY: {
      static uint32_t attr_sk = 0;
      const int smem_sk = SK_ST * SK_STAGE * 2;
      ensure_smem_attr((const void*)gemm_nt_skinny_kernel<EPI>, smem_sk, attr_sk);
      hipLaunchKernelGGL(gemm_nt_skinny_kernel<EPI>, dim3(pl.grid + prefetch_blocks(p)), dim3(256), smem_sk, s, p, pl.grid);
      break;
    }
    case NT_PP: launch_pp<EPI>(p, pl.grid, s); break;
    case NT_OW: if constexpr (EPI == EPI_BF16 || EPI == EPI_F32) launch_ow<EPI>(p, pl.grid, s); break;        // (nt_ow_takes: these two epilogues only)
    case NT_OWP: if constexpr (EPI == EPI_BF16) launch_owp<EPI>(p, pl.grid, s); break;
    case NT_MID: {
      static uint32_t attr_md = 0;
      const int smem_md = MD_ST * MD_STAGE * 2;
      ensure_smem_attr((const void*)gemm_nt_mid_kernel<EPI>, smem_md, attr_md);
      hipLaunchKernelGGL(gemm_nt_mid_kernel<EPI>, dim3(pl.grid + prefetch_blocks(p)), dim3(256), smem_md, s, p, pl.grid);
      break;
    }
    case NT_GLDS: hipLaunchKernelGGL(gemm_nt_glds_kernel<EPI>, dim3(pl.grid + prefetch_blocks(p)), dim3(256), smem, s, p, pl.grid); break;
    default: hipLaunchKernelGGL(gemm_nt_kernel<EPI>, dim3(pl.grid), dim3(256), smem, s, p); break;
  }
  return (int)hipGetLastError();
}

int gemm_nt_plan(const GemmNT& p, int* kind, int* grid) {
  if (p.K % BK != 0 || (p.A2 && p.K1 % BK != 0) || p.M <= 0 || p.N <= 0) return -1;
  const NtPlan pl = nt_plan(p);
  if (kind) *kind = pl.kind; if (grid) *grid = pl.grid;
  return 0;
}

// TFX_EPI_QKV_NORM_ROPE: fused in the ping-pong kernel when the shape runs there and the staged (16-byte, row-contiguous) stores apply; otherwise the
// plain projection followed by the token-wise kernel - same results either way.
static bool qknr_fusable(const GemmNT& p) {
  const int kind = nt_plan(p).kind;
  if (kind != NT_PP && kind != NT_DECODE) return false;
  if (kind == NT_PP && p.qk_cache) return false;                  // (the cache append exists in the decode-step kernel's instantiation only)
  const int hd = p.qk_heads * 64;
  if (p.qk_cache && (!p.qk_cache_pos || (p.qk_ld_cache & 7) || (((uintptr_t)p.qk_cache) & 15) || p.N < 3 * hd)) return false;
  return !p.rowmap && !p.a_rowmap && !p.bias && ((p.ldc | p.ldc2 | p.N) & 7) == 0 && (((uintptr_t)p.C | (uintptr_t)p.C2) & 15) == 0 && 2 * hd <= p.N && p.ldc2 >= 2 * hd;
}
static int gemm_nt_qknr(const GemmNT& p, hipStream_t s) {
  if (p.qk_heads <= 0 || !p.C2 || !p.qk_gamma_q || !p.qk_gamma_k || !p.qk_rot_pos || !p.qk_cos || !p.qk_sin) return -5;
  if (qknr_fusable(p)) {
    const NtPlan pl = nt_plan(p);
    if (pl.kind == NT_PP) { launch_pp<EPI_QKNR>(p, pl.grid, s); return (int)hipGetLastError(); }
    static uint32_t attr_sd = 0;
    const int smem_sd = 4 * SD_ST * SD_SLOT * 2;
    ensure_smem_attr((const void*)gemm_nt_decode_kernel<EPI_QKNR>, smem_sd, attr_sd);
    hipLaunchKernelGGL(gemm_nt_decode_kernel<EPI_QKNR>, dim3(pl.grid + prefetch_blocks(p)), dim3(256), smem_sd, s, p, pl.grid);
    return (int)hipGetLastError();
  }
  GemmNT q = p; q.epi = EPI_BF16; q.C2 = nullptr;
  const int rc = launch_nt<EPI_BF16>(q, s);
  if (rc) return rc;
  tfx_qk_norm_rope_args a;
  memset(&a, 0, sizeof(a));
  a.T = p.M; a.H = p.qk_heads; a.qkv = (const tfx_bf16*)p.C; a.ld_qkv = p.ldc; a.qk = (tfx_bf16*)p.C2; a.ld_qk = p.ldc2;
  a.gamma_q = p.qk_gamma_q; a.gamma_k = p.qk_gamma_k; a.rot_pos = p.qk_rot_pos; a.cos_tab = p.qk_cos; a.sin_tab = p.qk_sin;
  a.q_scale = p.qk_q_scale; a.norm_scale = p.qk_norm_scale; a.sc_plan = p.qk_plan; a.softcap = p.qk_softcap;
  a.cache = p.qk_cache; a.ld_cache = p.qk_ld_cache; a.cache_pos = p.qk_cache_pos;
  return tfx_qk_norm_rope_fwd(&a, (void*)s);
}

int gemm_nt(const GemmNT& p, hipStream_t s) {
  if (p.K % BK != 0 || (p.A2 && p.K1 % BK != 0) || p.M <= 0 || p.N <= 0) return -1;
  if ((p.lda | p.ldb) & 7) return -2;
  switch (p.epi) {
    case EPI_BF16: return launch_nt<EPI_BF16>(p, s);
    case EPI_F32: return launch_nt<EPI_F32>(p, s);
    case EPI_SILU: return launch_nt<EPI_SILU>(p, s);
    case EPI_RESID: return launch_nt<EPI_RESID>(p, s);
    case EPI_GEGLU: return (p.N % 64) ? -3 : launch_nt<EPI_GEGLU>(p, s);
    case EPI_GEGLU_BWD: return (p.N % 64) ? -3 : launch_nt<EPI_GEGLU_BWD>(p, s);
    case EPI_QKNR: return gemm_nt_qknr(p, s);
  }
  return -4;
}

template <bool SUM, int FA, int FB, int WN, int WK, int NST> static void launch_tn_wide(const GemmTN& q, int grid, hipStream_t s) {
  constexpr int smem = NST * (FA * WN + FB * WK) / 4 * MW_ROWS * 128 * 2;
  static uint32_t attr = 0;
  ensure_smem_attr((const void*)gemm_tn_wide_kernel<SUM, FA, FB, WN, WK, NST>, smem, attr);
  hipLaunchKernelGGL((gemm_tn_wide_kernel<SUM, FA, FB, WN, WK, NST>), dim3(grid), dim3(64 * WN * WK), smem, s, q);
}

// what gemm_tn launches for a shape: kernel form (-1 register-staged fallback, 0 = 128 x 128 / 4 waves, 2 = 256 x 256 / 8 waves), output tiles,
// row chunks, grid.  Pure host logic (tfx.h tfx_gemm_tn_plan: the CPU tests pin the split rule through it).
struct TnPlan { int kind, tiles, splits, grid; };
static int tn_ow_mode();
static bool tn_ow_operands_ok(const GemmTN& q);
static TnPlan tn_plan(const GemmTN& q) {
#ifdef TFX_TN_TIMING
  const bool dma_ok = true;
#else
  const bool dma_ok = use_glds() && q.M % TN_BMK == 0 && !q.a_rowmap && !q.b_rowmap && q.a_cols >= 8 && q.b_cols >= 8;
#endif
  const int t44 = ((q.N + 255) / 256) * ((q.K + 255) / 256), t22 = ((q.N + 127) / 128) * ((q.K + 127) / 128);
  TnPlan pl;
  // 256 x 256 tiles once there are enough of them to fill the chip at <= 32 splits; below that (512 x 512: 4 tiles) the 128 x 128 blocks
  pl.kind = !dma_ok ? -1 : (t44 >= 8 ? 2 : 0);
  pl.tiles = pl.kind == 2 ? t44 : t22;
  pl.splits = q.splits == 0 ? tn_auto_splits(q.M, pl.tiles, pl.kind) : q.splits;
  pl.grid = (pl.tiles * pl.splits + 7) / 8 * 8;                   // tn_block: 8 equal runs of (chunk, tile) pairs, one per XCD
  // kind 3: the 256 x 256 launches that the one-wave-per-SIMD kernel takes (TFX_TN_OW=0: none): 32-bit operand offsets, every chunk
  // >= 192 rows in multiples of 64.  Ragged N / K tiles included: the kernel computes their dead fragments too and still measured 10-13 % faster on
  // 1544 x 512 and 512 x 1408 (gpurun_out/ow9_tn.txt)
  if (tn_ow_mode() && pl.kind == 2 && tn_ow_operands_ok(q)) {
    const int chunk = ((q.M + pl.splits - 1) / pl.splits + TN_BMK - 1) / TN_BMK * TN_BMK;
    const int last = q.M % chunk == 0 ? chunk : q.M % chunk;      // rows of the last non-empty chunk (trailing empty chunks: the kernel returns on them)
    if (chunk >= 192 && last >= 192) pl.kind = 3;
  }
  return pl;
}
// grouped launch (tfx.h group_next): the chain's products on ONE grid of the one-wave kernel when every one of them qualifies; `tiles` / `splits` / `grid` of the group
struct TnGroupPlan { bool ok; int count, tiles, splits, grid; TnGroup g; };
static int tn_ow_mode() {
  static int tnow = -1;
  if (tnow < 0) { const char* e = getenv("TFX_TN_OW"); tnow = e ? atoi(e) : 1; }
  return tnow;
}
static bool tn_ow_operands_ok(const GemmTN& q) {        // what the one-wave kernel needs of a product, whatever its tile count
  const long long lim = 1ll << 32;
  const bool b_ok = q.b_cols >= q.K;
  return use_glds() && q.M % 64 == 0 && !q.a_rowmap && !q.b_rowmap && (((uintptr_t)q.A | (uintptr_t)q.B) & 15) == 0 && ((q.lda | q.ldb | q.a_cols | q.b_cols) & 7) == 0 &&
         (long long)(q.M + 64) * q.lda * 2 < lim && (long long)(q.M + 64) * q.ldb * 2 < lim && q.a_cols >= q.N && b_ok && q.N > 0 && q.K > 0;
}
static TnGroupPlan tn_group_plan(const GemmTN& head) {
  TnGroupPlan gp; memset(&gp, 0, sizeof(gp));
  static int grp = -1;                // TFX_TN_GROUP=0: chains run product by product (A/B)
  if (grp < 0) { const char* e = getenv("TFX_TN_GROUP"); grp = e ? atoi(e) : 1; }
  const GemmTN* q = &head;
  int n = 0, tiles = 0;
  bool ok = grp != 0 && tn_ow_mode() != 0;
  while (q && n < TN_GROUP_MAX) {
    ok = ok && q->M == head.M && tn_ow_operands_ok(*q);
    tiles += ((q->N + 255) / 256) * ((q->K + 255) / 256);
    gp.g.p[n] = *q; gp.g.p[n].group_next = nullptr; gp.g.tile_end[n] = tiles;
    n++;
    q = (const GemmTN*)q->group_next;
  }
  ok = ok && q == nullptr && n >= 2;                                 // (longer chains: one by one)
  gp.count = n; gp.tiles = tiles; gp.g.count = n;
  if (!ok) return gp;
  gp.splits = head.splits == 0 ? tn_auto_splits(head.M, tiles, 2) : head.splits;
  const int chunk = ((head.M + gp.splits - 1) / gp.splits + TN_BMK - 1) / TN_BMK * TN_BMK;
  const int last = head.M % chunk == 0 ? chunk : head.M % chunk;
  if (chunk < 192 || last < 192) return gp;
  gp.grid = (tiles * gp.splits + 7) / 8 * 8;
  // a group must still fill the chip: 132 tiles (dim 1024: 1024 x 2752 + 5504 x 1024) cannot be cut into two chunk rounds and would run on 132 of 256 CUs where
  // the two products alone put 220 and 176 blocks out (config 3 measured 183.1 ms grouped against 179.4; gpurun_out/ow40.txt) - such chains run one by one
  if (head.splits == 0 && tiles * gp.splits < 0.8 * 256) return gp;
  for (int i = 0; i < n; i++) gp.g.p[i].splits = gp.splits;
  gp.ok = true;
  return gp;
}
static int tn_ramp(int M, int tiles, int splits) {   // tn_block_ramp's d for a launch (see the kind-3 branch of gemm_tn)
  constexpr int ramp_min = 8;          // fewest row chunks that get a ramp (2-5-chunk launches measured nothing to +1 %; 6 - the 7-chunk FeedForward group - 4.19 -> 4.17 ms, noise)
  int ramp = 0;
  if (splits >= ramp_min && M % TN_BMK == 0) {
    const int S = splits, U = M / TN_BMK;
    ramp = (int)(0.148 * tiles + 0.5);                                 // (factors 0.5 / 1.5 / 2 all measured worse: profiles/r05b_tn_ramped_chunks.txt)
    while (ramp > 0 && (U - ramp * S * (S - 1) / 2) / S < 8) ramp--;   // the shortest chunk keeps >= 8 steps
  }
  return ramp;
}

int gemm_tn_plan(const GemmTN& p, int* kind, int* tiles, int* splits, int* grid) {
  if (p.M <= 0 || p.N <= 0 || p.K <= 0 || p.splits < 0) return -1;
  if (p.group_next) {                                               // a chain that runs as one launch: kind 3 with the group's tiles / chunks / grid
    const TnGroupPlan gp = tn_group_plan(p);
    if (gp.ok) { if (kind) *kind = 3; if (tiles) *tiles = gp.tiles; if (splits) *splits = gp.splits; if (grid) *grid = gp.grid; return 0; }
  }
  const TnPlan pl = tn_plan(p);
  if (kind) *kind = pl.kind; if (tiles) *tiles = pl.tiles; if (splits) *splits = pl.splits; if (grid) *grid = pl.grid;
  return 0;
}

int gemm_tn(const GemmTN& p, hipStream_t s) {
  if (p.M <= 0 || p.N <= 0 || p.K <= 0 || p.splits < 0) return -1;
  if ((p.lda | p.ldb | p.a_cols | p.b_cols) & 7) return -2;
  if (p.group_next) {
    const TnGroupPlan gp = tn_group_plan(p);
    if (gp.ok) {
      bool sum = false;
      for (int i = 0; i < gp.count; i++) sum = sum || gp.g.p[i].colsum != nullptr;
      const int ramp = tn_ramp(p.M, gp.tiles, gp.splits);
      static uint32_t attr_g0 = 0, attr_g1 = 0;
      if (sum) { ensure_smem_attr((const void*)gemm_tn_ow_group_kernel<true>, 131072, attr_g1); hipLaunchKernelGGL(gemm_tn_ow_group_kernel<true>, dim3(gp.grid), dim3(256), 131072, s, gp.g, ramp); }
      else { ensure_smem_attr((const void*)gemm_tn_ow_group_kernel<false>, 131072, attr_g0); hipLaunchKernelGGL(gemm_tn_ow_group_kernel<false>, dim3(gp.grid), dim3(256), 131072, s, gp.g, ramp); }
      return (int)hipGetLastError();
    }
    int guard = 0;
    for (const GemmTN* q = &p; q; q = (const GemmTN*)q->group_next) {     // not groupable: the chain's products one by one
      if (++guard > 64) return -7;                                       // (a chain that loops back on itself)
      GemmTN one = *q; one.group_next = nullptr;
      const int rc = gemm_tn(one, s);
      if (rc) return rc;
    }
    return 0;
  }
  static uint32_t attr_set = 0;
  const int smem = 2 * 2 * TN_BMK * TN_LD * 2;
  ensure_smem_attr((const void*)gemm_tn_kernel, smem, attr_set);
  GemmTN q = p;
  const TnPlan pl = tn_plan(q);
  q.splits = pl.splits;
  const int kind = pl.kind, grid = pl.grid;
  if (kind == 3) {
    static uint32_t attr_tnow = 0;
    // ramp (tn_block_ramp): the finish times of the chunks should spread over about the time the launch's atomics take, tiles x chunks x 0.2 us (256 KiB at
    // 1.25 TB/s), i.e. d = 0.2 us x tiles / (a step's 1.35 us) steps of 64 rows per chunk index.  Measured
    // (gpurun_out/ow31.txt, steady state): -3.5 ... -7.8 % on the 11-20-chunk launches of config 2 at factor 1, worse at 0.5 / 1.5 / 2, nothing to +1 % on the
    // 2-5-chunk launches - so from 8 chunks on
    const int ramp = tn_ramp(q.M, pl.tiles, q.splits);
    if (q.colsum) {
      static uint32_t attr_tnows = 0;
      ensure_smem_attr((const void*)gemm_tn_ow_kernel<true>, 131072, attr_tnows);
      hipLaunchKernelGGL(gemm_tn_ow_kernel<true>, dim3(grid), dim3(256), 131072, s, q, ramp);
    } else {
      ensure_smem_attr((const void*)gemm_tn_ow_kernel<false>, 131072, attr_tnow);
      hipLaunchKernelGGL(gemm_tn_ow_kernel<false>, dim3(grid), dim3(256), 131072, s, q, ramp);
    }
  } else if (kind == 2) {
    if (q.colsum) launch_tn_wide<true, 2, 4, 4, 2, 4>(q, grid, s);
    else launch_tn_wide<false, 2, 4, 4, 2, 4>(q, grid, s);
  } else if (kind == 0) {
    if (q.colsum) launch_tn_wide<true, 2, 2, 2, 2, 4>(q, grid, s);
    else launch_tn_wide<false, 2, 2, 2, 2, 4>(q, grid, s);
  }
  else {
    hipLaunchKernelGGL(gemm_tn_kernel, dim3(grid), dim3(256), smem, s, q);
    if (q.colsum)      // the register-staged fallback does not fold the bias gradient in
      return tfx_colsum_bf16((const tfx_bf16*)q.A, q.lda, q.M, q.N, q.rowmap, q.a_rowmap, q.colsum, (void*)s);
  }
  return (int)hipGetLastError();
}

}  // namespace tfx
