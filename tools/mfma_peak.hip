// Measured dense bf16 MFMA peak of THIS box: a register-resident loop (no memory traffic) of v_mfma_f32_32x32x16_bf16 on every SIMD of every CU,
// timed with HIP events; s_memtime ticks / event time = the shader clock the chip actually sustains under that load.  Zero-filled vs random
// operands (the guide: a zero-fill clocks ~20 % higher - power, not work).  Also a mixed MFMA + VALU loop (the attention-like instruction mix).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/mfma_peak && tools/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int VALU_PER_MFMA>
__global__ __launch_bounds__(256) void burn(const bf16x8* ops, float* out, unsigned long long* ticks, int iters) {
  bf16x8 a = ops[threadIdx.x & 63], b = ops[64 + (threadIdx.x & 63)];
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  float x0 = out[0], x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
#define M(acc) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
#define VV asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(x0), "v"(x1));
    M(c0) if (VALU_PER_MFMA >= 4) { VV } if (VALU_PER_MFMA >= 8) { VV }
    M(c1) if (VALU_PER_MFMA >= 4) { VV } if (VALU_PER_MFMA >= 8) { VV }
    M(c2) if (VALU_PER_MFMA >= 4) { VV } if (VALU_PER_MFMA >= 8) { VV }
    M(c3) if (VALU_PER_MFMA >= 4) { VV } if (VALU_PER_MFMA >= 8) { VV }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = x0 + x1 + x2 + x3;
  for (int i = 0; i < 16; i++) s += c0[i] + c1[i] + c2[i] + c3[i];
  if (s == 1.2345f) out[1] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

// the 16x16x32 shape (what the vendor library's kernels issue, `MI16x16x1`): 8 independent accumulators of 4 registers, half the flops per instruction
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ __launch_bounds__(256) void burn16(const bf16x8* ops, float* out, unsigned long long* ticks, int iters) {
  bf16x8 a = ops[threadIdx.x & 63], b = ops[64 + (threadIdx.x & 63)];
  f32x4 c[8] = {};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c[k]) : "v"(a), "v"(b));
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int k = 0; k < 8; k++) for (int i = 0; i < 4; i++) s += c[k][i];
  if (s == 1.2345f) out[1] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}
void run16(const bf16x8* ops, float* out, unsigned long long* ticks, int blocks_per_cu) {
  const int iters = 20000, nblk = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(burn16, dim3(nblk), dim3(256), 0, 0, ops, out, ticks, iters / 10);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(burn16, dim3(nblk), dim3(256), 0, 0, ops, out, ticks, iters);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long t; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
  const double flop = 2.0 * 16 * 16 * 32 * 8.0 * iters * (nblk * 4.0);
  printf("%-44s %d block(s)/CU: %7.1f TFLOP/s  (%.3f ms, shader clock %.2f GHz, %.1f clocks per MFMA per SIMD)\n", "MFMA only, 16x16x32", blocks_per_cu, flop / ms * 1e-9, ms,
         t / (ms * 1e6), (double)t / (8.0 * iters) / blocks_per_cu);
}

template <int NV> void run(const char* name, const bf16x8* ops, float* out, unsigned long long* ticks, int blocks_per_cu) {
  const int iters = 20000, nblk = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(burn<NV>, dim3(nblk), dim3(256), 0, 0, ops, out, ticks, iters / 10);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(burn<NV>, dim3(nblk), dim3(256), 0, 0, ops, out, ticks, iters);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long t; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
  const double flop = 2.0 * 32 * 32 * 16 * 4.0 * iters * (nblk * 4.0);
  printf("%-44s %d block(s)/CU: %7.1f TFLOP/s  (%.3f ms, shader clock %.2f GHz, %.1f clocks per MFMA per SIMD)\n", name, blocks_per_cu, flop / ms * 1e-9, ms,
         t / (ms * 1e6), (double)t / (4.0 * iters) / blocks_per_cu);
}

int main() {
  bf16x8* ops; float* out; unsigned long long* ticks;
  hipMalloc(&ops, 128 * sizeof(bf16x8)); hipMalloc(&out, 64); hipMalloc(&ticks, 8);
  hipMemset(out, 0, 64);
  unsigned short h[128 * 8];
  for (int fill = 0; fill < 2; fill++) {
    for (int i = 0; i < 128 * 8; i++) { float v = fill ? (float)rand() / RAND_MAX * 2.f - 1.f : 0.f; unsigned u; memcpy(&u, &v, 4); h[i] = (unsigned short)(u >> 16); }
    hipMemcpy(ops, h, sizeof(h), hipMemcpyHostToDevice);
    printf("== operands: %s\n", fill ? "uniform random [-1, 1)" : "zero");
    run<0>("MFMA only", ops, out, ticks, 1);
    run<0>("MFMA only", ops, out, ticks, 2);
    run16(ops, out, ticks, 1);
    run16(ops, out, ticks, 2);
    run<4>("MFMA + 4 v_fma per MFMA", ops, out, ticks, 1);
    run<8>("MFMA + 8 v_fma per MFMA", ops, out, ticks, 1);
    run<8>("MFMA + 8 v_fma per MFMA", ops, out, ticks, 2);
  }
  return 0;
}
