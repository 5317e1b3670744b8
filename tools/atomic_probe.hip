// How fast are fp32 atomic adds in the dQ-accumulation pattern of a fused attention backward?  Grid = (n/128 key blocks, h, b) like
// attn_bwd_dkv_kernel; every block walks the 64-query tiles that see its keys and adds a [64 q][64 d] fp32 tile per iteration into
// dq[b][q][h*64 + d] (wave w: q-block w>>1, d-block w&1; lane&31 = d, 16 rows per lane) - 128-byte contiguous segments per row.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/atomic_probe.hip -o /tmp/atomic_probe && /tmp/atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>   // 0: atomics, 1: plain stores (traffic only), 2: nothing but the loop
__global__ __launch_bounds__(256, 2) void probe(float* dq, int n, int H, int ld, int spin) {
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, hi = l >> 5;
  const int h = blockIdx.y, b = blockIdx.z, k0 = blockIdx.x * 128;
  float v = (float)(l + 1) * 1e-6f;
  for (int jt = k0 / 64; jt < n / 64; jt++) {
    for (int i = 0; i < spin; i++) v = v * 1.0001f + 1e-7f;       // stand-in for the MFMA / soft-cap work of one tile
    float* base = dq + ((size_t)b * n + jt * 64 + (w >> 1) * 32) * ld + h * 64 + (w & 1) * 32 + (l & 31);
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int q = (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (MODE == 0) atomicAdd(base + (size_t)q * ld, v);
      else if (MODE == 1) base[(size_t)q * ld] = v;
    }
  }
  if (v == 123.456f) dq[0] = v;
}
int main() {
  const int b = 64, n = 1024, H = 8, ld = 512;
  float* dq; hipMalloc(&dq, (size_t)b * n * ld * 4); hipMemset(dq, 0, (size_t)b * n * ld * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int spin : {0, 2000, 6000}) {
    float ms[3];
    for (int mode = 0; mode < 3; mode++) {
      for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0, 0);
        if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(n / 128, H, b), dim3(256), 0, 0, dq, n, H, ld, spin);
        if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(n / 128, H, b), dim3(256), 0, 0, dq, n, H, ld, spin);
        if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(n / 128, H, b), dim3(256), 0, 0, dq, n, H, ld, spin);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms[mode], e0, e1);
      }
    }
    const double tiles = (double)b * H * (8 * 9 / 2 * 2 + 0);     // sum over key blocks of remaining 64-q tiles: 16+14+...+2 = 72
    printf("spin %5d: atomics %.1f us, plain stores %.1f us, loop only %.1f us  (%.0f M atomic dwords, %.0f MB)\n", spin, ms[0] * 1e3, ms[1] * 1e3,
           ms[2] * 1e3, tiles * 4096 / 1e6, tiles * 16384 / 1e6);
  }
  return 0;
}
