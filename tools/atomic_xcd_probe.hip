// Are fp32 atomics faster when every block that adds into one output tile sits on the same XCD (private L2)?  242 blocks x 256 KiB of `atomicAdd` into 22 tiles
// (the weight-gradient kernel's 2816 x 512 launch), block b on XCD b % 8:  mode 0: tile = (b / 8) % 22 style spread (chunks of a tile on DIFFERENT XCDs, as
// tn_block maps them), mode 1: tile chosen so that all adders of a tile share b % 8.  Also 4-byte vs (emulated) wider patterns are not available for fp32.
//   hipcc --offload-arch=gfx950 -O3 tools/atomic_xcd_probe.hip -o tools/atomic_xcd_probe && tools/atomic_xcd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ __launch_bounds__(256) void adders(float* C, int ntile, int mode, int nblk) {
  const int b = blockIdx.x;
  int tile;
  if (mode == 0) tile = b % ntile;                       // consecutive blocks (different XCDs) walk the tiles: a tile's adders are spread over the XCDs
  else { const int x = b & 7, q = b >> 3; const int per = (ntile + 7) / 8; tile = (x * per + q % per) % ntile; }   // a tile's adders all have the same b % 8
  float* base = C + (size_t)tile * 65536;                // 256 x 256 fp32
  const int t = threadIdx.x, l = t & 63, w = t >> 6;
  // the kernel's pattern: a wave owns 128 x 128; per instruction 2 rows x 32 consecutive floats
  for (int i = 0; i < 4; i++)
    for (int r = 0; r < 16; r++) {
      const int row = (w >> 1) * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
      for (int j = 0; j < 4; j++) atomicAdd(base + row * 256 + (w & 1) * 128 + j * 32 + (l & 31), 1.0f);
    }
}
int main() {
  const int ntile = 22, nblk = 242;
  float* C; CK(hipMalloc(&C, (size_t)ntile * 65536 * 4)); CK(hipMemset(C, 0, (size_t)ntile * 65536 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 2; mode++) {
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(adders, dim3(nblk), dim3(256), 0, 0, C, ntile, mode, nblk);
    CK(hipEventRecord(e0, 0));
    const int reps = 200;
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(adders, dim3(nblk), dim3(256), 0, 0, C, ntile, mode, nblk);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1000.0 / reps, mib = nblk * 0.25;
    printf("mode %d (%s): %.1f us per launch of %d blocks x 256 KiB = %.1f MiB of fp32 atomics -> %.2f TB/s of added data\n", mode, mode ? "a tile's adders on ONE XCD" : "a tile's adders spread over the XCDs", us, nblk, mib, mib * 1.048576e6 / us * 1e-6);
  }
  // plain stores of the same volume for scale
  return 0;
}
