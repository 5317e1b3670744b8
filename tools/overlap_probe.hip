// Do the matrix pipe and the vector ALU of ONE SIMD overlap - across two co-resident waves, and inside one wave's instruction stream?
// One workgroup on one CU; s_memtime around 256 MFMAs (v_mfma_f32_32x32x16_bf16, 4 independent accumulators) and / or N plain VALU.
//   hipcc --offload-arch=gfx950 -O3 tools/overlap_probe.hip -o tools/overlap_probe && tools/overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
#define MFMA(acc) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
#define VALU4 asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(y0), "v"(y1));
#define EXP4 asm volatile("v_exp_f32 %0, %4\n v_exp_f32 %1, %4\n v_exp_f32 %2, %4\n v_exp_f32 %3, %4" : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3) : "v"(y0));

// MODE 0: every wave MFMA only; 1: every wave VALU only (1024 v_fma); 2: even waves MFMA, odd waves VALU (pairs share a SIMD when the block has 8 waves);
// 3: one stream, each MFMA followed by 4 v_fma; 4: each MFMA followed by 8 v_fma; 5: each MFMA followed by 12 v_fma; 6: MFMA + 4 v_exp
template <int MODE>
__global__ void probe(unsigned long long* out, float seed) {
  const int wave = threadIdx.x >> 6;
  bf16x8 a, b;
  for (int i = 0; i < 8; i++) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed * 0.5f + i); }
  f32x16 c0, c1, c2, c3;
  for (int i = 0; i < 16; i++) { c0[i] = c1[i] = c2[i] = c3[i] = 0.f; }
  float x0 = seed, x1 = seed * 2, x2 = seed * 3, x3 = seed * 4, y0 = 1.0001f, y1 = 0.5f;
  unsigned long long dt = 0;
  for (int w = 0; w < 2; w++) {
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_readcyclecounter();
    const bool do_mfma = MODE == 0 || (MODE == 2 && (wave & 1) == 0) || MODE >= 3;
    const bool do_valu = MODE == 1 || (MODE == 2 && (wave & 1) == 1);
    if (MODE <= 2) {
      if (do_mfma) { REP64(MFMA(c0) MFMA(c1) MFMA(c2) MFMA(c3)) }
      if (do_valu) { REP64(VALU4 VALU4 VALU4 VALU4) }
    }
    if (MODE == 3) { REP64(MFMA(c0) VALU4 MFMA(c1) VALU4 MFMA(c2) VALU4 MFMA(c3) VALU4) }
    if (MODE == 4) { REP64(MFMA(c0) VALU4 VALU4 MFMA(c1) VALU4 VALU4 MFMA(c2) VALU4 VALU4 MFMA(c3) VALU4 VALU4) }
    if (MODE == 5) { REP64(MFMA(c0) VALU4 VALU4 VALU4 MFMA(c1) VALU4 VALU4 VALU4 MFMA(c2) VALU4 VALU4 VALU4 MFMA(c3) VALU4 VALU4 VALU4) }
    if (MODE == 6) { REP64(MFMA(c0) EXP4 MFMA(c1) EXP4 MFMA(c2) EXP4 MFMA(c3) EXP4) }
    dt = __builtin_readcyclecounter() - t0;
  }
  if ((threadIdx.x & 63) == 0) out[wave] = dt;
  float s = x0 + x1 + x2 + x3;
  for (int i = 0; i < 16; i++) s += c0[i] + c1[i] + c2[i] + c3[i];
  if (s == 12345.f) out[63] = 0;
}

template <int MODE>
void run(const char* name, unsigned long long* out, int waves) {
  hipMemset(out, 0, 64 * 8);
  hipLaunchKernelGGL(probe<MODE>, dim3(1), dim3(64 * waves), 0, 0, out, 1.0f);
  hipDeviceSynchronize();
  unsigned long long h[16]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  unsigned long long mx = 0, mn = ~0ull;
  for (int i = 0; i < waves; i++) { mx = h[i] > mx ? h[i] : mx; mn = h[i] < mn ? h[i] : mn; }
  printf("%-62s %d waves: %7llu .. %7llu clocks (per MFMA %.1f)\n", name, waves, mn, mx, mx / 256.0);
}

int main() {
  unsigned long long* out; hipMalloc(&out, 64 * 8);
  run<0>("256 MFMA per wave", out, 4);
  run<0>("256 MFMA per wave", out, 8);
  run<1>("1024 v_fma per wave", out, 4);
  run<1>("1024 v_fma per wave", out, 8);
  run<2>("even waves 256 MFMA, odd waves 1024 v_fma (pairs per SIMD)", out, 8);
  run<3>("one stream: 256 x (MFMA + 4 v_fma)", out, 4);
  run<3>("one stream: 256 x (MFMA + 4 v_fma)", out, 8);
  run<4>("one stream: 256 x (MFMA + 8 v_fma)", out, 4);
  run<4>("one stream: 256 x (MFMA + 8 v_fma)", out, 8);
  run<5>("one stream: 256 x (MFMA + 12 v_fma)", out, 4);
  run<5>("one stream: 256 x (MFMA + 12 v_fma)", out, 8);
  run<6>("one stream: 256 x (MFMA + 4 v_exp)", out, 4);
  run<6>("one stream: 256 x (MFMA + 4 v_exp)", out, 8);
  return 0;
}
