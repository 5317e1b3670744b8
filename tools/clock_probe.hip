// What does __builtin_readcyclecounter() (s_memtime) count on this GPU?  One wave spins for N ticks; the host times it with events.
// Also reports s_memrealtime (constant 100 MHz) over the same interval.   hipcc --offload-arch=gfx950 tools/clock_probe.hip -o /tmp/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(unsigned long long n, unsigned long long* out) {
  const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long t;
  do { t = __builtin_readcyclecounter(); } while (t - t0 < n);
  out[0] = t - t0; out[1] = __builtin_amdgcn_s_memrealtime() - r0;
}
__global__ void burn(float* x, int iters) {      // keeps every CU busy with FMAs so the probe also runs under load
  float a = x[threadIdx.x], b = 1.0001f;
  for (int i = 0; i < iters; i++) { a = a * b + 0.5f; b = b * a + 0.25f; }
  x[blockIdx.x * blockDim.x + threadIdx.x] = a + b;
}
int main() {
  unsigned long long* out; hipMalloc(&out, 16);
  float* x; hipMalloc(&x, 2048 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int load = 0; load < 2; load++) {
    hipStream_t s2; hipStreamCreate(&s2);
    if (load) hipLaunchKernelGGL(burn, dim3(2048), dim3(256), 0, s2, x, 4000000);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, 0, 20000000ull, out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    printf("%s: %llu s_memtime ticks, %llu s_memrealtime ticks in %.3f ms -> s_memtime %.1f MHz, s_memrealtime %.1f MHz\n", load ? "other CUs busy" : "idle GPU",
           h[0], h[1], ms, h[0] / (ms * 1e3), h[1] / (ms * 1e3));
    hipDeviceSynchronize();
  }
  return 0;
}
