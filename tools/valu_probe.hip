// Issue cost (shader clocks per wave64 instruction) of the VALU / transcendental / conversion instructions the attention soft-max path is
// made of, measured with s_memtime around a block of 256 independent instructions, for 1, 2 and 3 resident waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_probe.hip -o /tmp/valu_probe && /tmp/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

template <int OP>
__global__ void probe(unsigned long long* out, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 * 0.5f, a2 = a0 * 0.25f, a3 = a0 * 0.125f, b0 = 1.0001f, b1 = 0.9999f;
  float c0 = a0, c1 = a1, c2 = a2, c3 = a3;
  unsigned u0 = 0, u1 = 0;
  // warm the instruction cache / clocks
  for (int w = 0; w < 2; w++) {
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (OP == 0) { REP64(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));) }
    if (OP == 1) { REP64(asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3" : "+v"(*(double*)&a0), "+v"(*(double*)&a2) : "v"(*(double*)&b0), "v"(*(double*)&c0));) }
    if (OP == 2) { REP64(asm volatile("v_exp_f32 %0, %4\n v_exp_f32 %1, %4\n v_exp_f32 %2, %4\n v_exp_f32 %3, %4" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(b0));) }
    if (OP == 3) { REP64(asm volatile("v_cvt_pk_bf16_f32 %0, %2, %3\n v_cvt_pk_bf16_f32 %1, %2, %3\n v_cvt_pk_bf16_f32 %0, %3, %2\n v_cvt_pk_bf16_f32 %1, %3, %2" : "=v"(u0), "=v"(u1) : "v"(b0), "v"(b1));) }
    if (OP == 4) { REP64(asm volatile("v_max3_f32 %0, %0, |%4|, |%5|\n v_max3_f32 %1, %1, |%4|, |%5|\n v_max3_f32 %2, %2, |%4|, |%5|\n v_max3_f32 %3, %3, |%4|, |%5|" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));) }
    if (OP == 5) { REP64(asm volatile("v_pk_mul_f32 %0, %0, %2\n v_pk_mul_f32 %1, %1, %2\n v_pk_mul_f32 %0, %0, %2\n v_pk_mul_f32 %1, %1, %2" : "+v"(*(double*)&a0), "+v"(*(double*)&a2) : "v"(*(double*)&b0));) }
    if (OP == 6) { REP64(asm volatile("v_mul_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));) }
    if (OP == 7) { REP64(asm volatile("v_exp_f32 %0, %4\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "=v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));) }
    if (OP == 8) { REP64(asm volatile("v_cndmask_b32 %0, 0, %0, vcc\n v_cndmask_b32 %1, 0, %1, vcc\n v_cndmask_b32 %2, 0, %2, vcc\n v_cndmask_b32 %3, 0, %3, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) :: "vcc");) }
    if (OP == 9) { REP64(asm volatile("v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2\n v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2" : "+v"(*(double*)&a0), "+v"(*(double*)&a2) : "v"(*(double*)&b0));) }
    if (OP == 10) { REP64(asm volatile("v_ldexp_f32 %0, %0, %4\n v_ldexp_f32 %1, %1, %4\n v_rndne_f32 %2, %2\n v_rndne_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(u0));) }
    if (OP == 11) { REP64(asm volatile("v_rcp_f32 %0, %4\n v_rcp_f32 %1, %4\n v_rcp_f32 %2, %4\n v_rcp_f32 %3, %4" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(b0));) }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (w == 1 && (threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
  }
  if (a0 + a1 + a2 + a3 + c0 + c1 + c2 + c3 + (float)(u0 + u1) == 12345.f) out[0] = 0;
}

template <int OP>
void run(const char* name, unsigned long long* out) {
  for (int wps = 1; wps <= 3; wps++) {            // waves per SIMD: block of 256 * wps threads on one CU... use 4*wps waves
    hipMemset(out, 0, 64 * 8);
    hipLaunchKernelGGL(probe<OP>, dim3(1), dim3(256 * wps), 0, 0, out, 1.0f);
    hipDeviceSynchronize();
    unsigned long long h[16]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    unsigned long long mx = 0; for (int i = 0; i < 4 * wps; i++) mx = h[i] > mx ? h[i] : mx;
    printf("%-34s %d wave(s)/SIMD: %6.2f clocks per instruction per wave (%5.2f per SIMD issue)\n", name, wps, mx / 256.0, mx / 256.0 / wps);
  }
}

int main() {
  unsigned long long* out; hipMalloc(&out, 64 * 8);
  run<0>("v_fma_f32", out);
  run<6>("v_mul_f32 / v_add_f32", out);
  run<1>("v_pk_fma_f32", out);
  run<5>("v_pk_mul_f32", out);
  run<9>("v_pk_add_f32", out);
  run<2>("v_exp_f32", out);
  run<11>("v_rcp_f32", out);
  run<7>("1 v_exp_f32 + 3 v_fma_f32", out);
  run<3>("v_cvt_pk_bf16_f32", out);
  run<4>("v_max3_f32 |a| |b|", out);
  run<8>("v_cndmask_b32", out);
  run<10>("v_ldexp_f32 / v_rndne_f32", out);
  return 0;
}
