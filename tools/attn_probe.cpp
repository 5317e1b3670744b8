// Stand-alone probe of tfx_attn_fwd / tfx_attn_bwd (no PyTorch: starts in seconds on a fresh GPU box).  Built by tools/build_attn_probe.sh into tools/attn_probe.
//
//   TFX_ATTN_ASM=0 tools/attn_probe run ref [case ...]   # every case on the hipcc-scheduled kernels: time / hash, outputs dumped to /tmp/attnp_ref_<case>.bin
//   TFX_ATTN_ASM=1 tools/attn_probe run asm [case ...]   # the same cases with the generated main loops (tools/gen_attn_loops.py)
//   tools/attn_probe cmp ref asm                         # byte comparison of the dumps (same arithmetic order: bit-identical) + max abs difference
//
// The kernel choice is an environment switch the library reads once per process, hence two runs.  `bwd` in argv[1] position 2 runs the backward too.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <cmath>
#include <string>
#include <vector>
#include "../include/tfx.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

struct Case { const char* name; int b, h, n; float cap; int blocks; bool timed; };   // blocks: modality blocks per sample (rows of a block see the whole block)
static const Case CASES[] = {
  {"n256",     3, 2,  256, 50.f, 0, false},
  {"n320",     2, 3,  320, 50.f, 2, false},
  {"n384",     2, 2,  384, 50.f, 0, false},
  {"n512",     2, 2,  512, 50.f, 0, false},
  {"n1000",    2, 2, 1000, 50.f, 3, false},      // ragged last tile
  {"n1024m1",  2, 2, 1024, 25.f, 4, false},      // soft-cap plan mode 1 (quintic)
  {"n1088",    2, 2, 1088, 50.f, 5, false},
  {"bench",   64, 8, 1024, 50.f, 16, true},      // BASELINE config 2: 16 blocks of 4 latents per sample
  {"bench_m1", 64, 8, 1024, 25.f, 16, true},
  {"cfg3",    64, 16, 1024, 50.f, 16, true},
};

static uint64_t fnv(const void* p, size_t n) {
  const uint8_t* b = (const uint8_t*)p; uint64_t h = 1469598103934665603ull;
  for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
  return h;
}
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
struct Rng { uint64_t s; float next() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (float)((s >> 40) & 0xffff) / 32768.f - 1.f; } };

// the layer's soft-cap plan as tfx_common.h softcap_plan_write forms it, for gains 0 (|q~| = norm_scale q_scale, |k~| = norm_scale)
static void host_plan(float cap, float* o) {
  const float ns = 8.f, q_scale = 0.125f, L2E = 1.4426950408889634f;
  const float B = 1.02f * ns * ns * q_scale, bx = B / cap, b2 = bx * bx, ic2 = 1.f / (cap * cap);
  float mode = 2.f, a1 = 1.f, a3 = -1.f / 3.f, a5 = 0.f;
  if (bx <= 0.2f) { mode = 0.f; a1 = 1.f - b2 * b2 / 24.f; a3 = -1.f / 3.f + b2 / 6.f; }
  else if (bx <= 0.35f) { const float c7 = 17.f / 315.f; mode = 1.f; a1 = 1.f - c7 * (7.f / 64.f) * b2 * b2 * b2; a3 = -1.f / 3.f + c7 * (7.f / 8.f) * b2 * b2; a5 = 2.f / 15.f - c7 * (7.f / 4.f) * b2; }
  o[0] = mode; o[1] = L2E * a1; o[2] = L2E * a3 * ic2; o[3] = L2E * a5 * ic2 * ic2; o[4] = a1; o[5] = 3.f * a3 * ic2; o[6] = 5.f * a5 * ic2 * ic2; o[7] = B;
}

static bool want(int argc, char** argv, int first, const char* name) {
  if (argc <= first) return true;
  for (int i = first; i < argc; i++) if (!strcmp(argv[i], name)) return true;
  return false;
}

static int run_main(int argc, char** argv) {
  const std::string tag = argv[2];
  const bool bwd = getenv("ATTNP_BWD") && atoi(getenv("ATTNP_BWD"));
  const int reps = getenv("ATTNP_REPS") ? atoi(getenv("ATTNP_REPS")) : 20;
  const char* libp = getenv("TFX_LIB") ? getenv("TFX_LIB") : "transfusion_pytorch_amd/lib/libtfx_hip.so";
  void* hnd = dlopen(libp, RTLD_NOW);
  if (!hnd) { fprintf(stderr, "dlopen %s: %s\n", libp, dlerror()); return 2; }
  auto afwd = (int (*)(const tfx_attn_args*, void*))dlsym(hnd, "tfx_attn_fwd");
  auto abwd = (int (*)(const tfx_attn_args*, void*))dlsym(hnd, "tfx_attn_bwd");
  if (!afwd || !abwd) { fprintf(stderr, "symbols missing\n"); return 2; }
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (const Case& c : CASES) {
    if (!want(argc, argv, 3, c.name)) continue;
    const int T = c.b * c.n, H = c.h, ld = 3 * H * 64 + 8;                // q | k | v in one token-major buffer (as the engine lays them out), padded
    std::vector<uint16_t> hq((size_t)T * ld), hg((size_t)T * H), hdo((size_t)T * H * 64);
    Rng r{(uint64_t)c.n * 977 + c.h};
    for (int t = 0; t < T; t++)
      for (int hh = 0; hh < H; hh++) {
        for (int which = 0; which < 3; which++) {
          float v[64], ss = 0.f;
          for (int i = 0; i < 64; i++) { v[i] = r.next(); ss += v[i] * v[i]; }
          const float sc = which == 0 ? 1.f / sqrtf(ss) : which == 1 ? 8.f / sqrtf(ss) : 1.f;     // |q~| = 1 (8 x q_scale 1/8), |k~| = 8, v ~ U(-1, 1)
          for (int i = 0; i < 64; i++) hq[(size_t)t * ld + which * H * 64 + hh * 64 + i] = f2bf(v[i] * sc);
        }
        if (getenv("ATTNP_VCODE"))                                         // V[key][c] = (key / 32 == c): out[i][c] = the softmax mass row i puts on 32-key unit c (x the gate)
          for (int i = 0; i < 64; i++) hq[(size_t)t * ld + 2 * H * 64 + hh * 64 + i] = f2bf(((t % c.n) / 32 == i) ? 1.f : 0.f);
        hg[(size_t)t * H + hh] = f2bf(2.f * r.next());
        for (int i = 0; i < 64; i++) hdo[((size_t)t * H + hh) * 64 + i] = f2bf(0.5f * r.next());
      }
    std::vector<int32_t> kve(T), qst(T);
    for (int s = 0; s < c.b; s++) {
      for (int i = 0; i < c.n; i++) { kve[s * c.n + i] = i + 1; qst[s * c.n + i] = i; }
      for (int k = 0; k < c.blocks; k++) {                                // blocks of 4 rows (BASELINE's latents) at pseudo-random aligned-or-not starts
        const int start = (int)(((uint64_t)(k + 1) * 2654435761u + s * 97) % (uint64_t)(c.n - 8));
        for (int i = start; i < start + 4; i++) {
          kve[s * c.n + i] = std::max(kve[s * c.n + i], start + 4);
          qst[s * c.n + i] = std::min(qst[s * c.n + i], start);
        }
      }
      for (int i = 1; i < c.n; i++) kve[s * c.n + i] = std::max(kve[s * c.n + i], kve[s * c.n + i - 1]);          // non-decreasing (prefix-extension form)
      for (int i = c.n - 2; i >= 0; i--) qst[s * c.n + i] = std::min(qst[s * c.n + i], qst[s * c.n + i + 1]);
    }
    float plan[8]; host_plan(c.cap, plan);
    uint16_t *dq, *dg, *dout, *ddo, *ddoeff, *ddgate, *ddqkv; int32_t *dkve, *dqst; float *dlse, *dplan, *ddelta;
    const size_t nout = (size_t)T * H * 64;
    CK(hipMalloc(&dq, hq.size() * 2)); CK(hipMalloc(&dg, hg.size() * 2)); CK(hipMalloc(&dout, nout * 2)); CK(hipMalloc(&ddo, nout * 2)); CK(hipMalloc(&ddoeff, nout * 2));
    CK(hipMalloc(&ddgate, hg.size() * 2)); CK(hipMalloc(&ddqkv, hq.size() * 2));
    CK(hipMalloc(&dkve, T * 4)); CK(hipMalloc(&dqst, T * 4)); CK(hipMalloc(&dlse, (size_t)T * H * 4)); CK(hipMalloc(&ddelta, (size_t)T * H * 4)); CK(hipMalloc(&dplan, 32));
    CK(hipMemcpy(dq, hq.data(), hq.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dg, hg.data(), hg.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(ddo, hdo.data(), nout * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dkve, kve.data(), T * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dqst, qst.data(), T * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dplan, plan, 32, hipMemcpyHostToDevice));
    CK(hipMemset(dout, 0, nout * 2)); CK(hipMemset(dlse, 0, (size_t)T * H * 4)); CK(hipMemset(ddqkv, 0, hq.size() * 2));
    tfx_attn_args a; memset(&a, 0, sizeof(a));
    a.q = dq; a.k = dq + H * 64; a.v = dq + 2 * H * 64; a.ld_q = a.ld_k = a.ld_v = ld;
    a.gate = dg; a.ld_gate = H; a.kv_end = dkve; a.q_start = dqst; a.out = dout; a.ld_out = H * 64; a.lse = dlse; a.b = c.b; a.h = H; a.n = c.n; a.softcap = c.cap; a.sc_plan = dplan;
    a.dout = ddo; a.ld_dout = H * 64; a.do_eff = ddoeff; a.ld_do = H * 64; a.delta = ddelta; a.dgate = ddgate; a.ld_dgate = H;
    a.dq = ddqkv; a.dk = ddqkv + H * 64; a.dv = ddqkv + 2 * H * 64; a.ld_dq = a.ld_dk = a.ld_dv = ld;
    int rc = afwd(&a, (void*)st);
    if (rc == 0 && bwd) rc = abwd(&a, (void*)st);
    hipError_t se = hipStreamSynchronize(st);
    if (rc != 0 || se != hipSuccess) { printf("%-10s rc %d sync %s\n", c.name, rc, hipGetErrorString(se)); return 3; }
    std::vector<uint16_t> ho(nout), hd(hq.size()); std::vector<float> hl((size_t)T * H);
    CK(hipMemcpy(ho.data(), dout, nout * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(hl.data(), dlse, hl.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hd.data(), ddqkv, hd.size() * 2, hipMemcpyDeviceToHost));
    double tf = 0, tb = 0;
    if (c.timed) {
      for (int phase = 0; phase < (bwd ? 2 : 1); phase++) {
        auto fn = phase == 0 ? afwd : abwd;
        for (int i = 0; i < 3; i++) fn(&a, (void*)st);
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; i++) fn(&a, (void*)st);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        (phase == 0 ? tf : tb) = ms * 1e3 / reps;
      }
    }
    int nan = 0; for (auto x : ho) { const float f = bf2f(x); if (f != f) nan++; }
    printf("%-10s b %3d h %2d n %5d cap %4.0f  fwd %8.1f us  bwd %8.1f us  out %016llx lse %016llx dqkv %016llx nan %d\n", c.name, c.b, c.h, c.n, c.cap, tf, tb,
           (unsigned long long)fnv(ho.data(), nout * 2), (unsigned long long)fnv(hl.data(), hl.size() * 4), (unsigned long long)fnv(hd.data(), hd.size() * 2), nan);
    fflush(stdout);
    FILE* f = fopen(("/tmp/attnp_" + tag + "_" + c.name + ".bin").c_str(), "wb");
    if (f) { fwrite(ho.data(), 2, nout, f); fwrite(hl.data(), 4, hl.size(), f); if (bwd) fwrite(hd.data(), 2, hd.size(), f); fclose(f); }
    hipFree(dq); hipFree(dg); hipFree(dout); hipFree(ddo); hipFree(ddoeff); hipFree(ddgate); hipFree(ddqkv); hipFree(dkve); hipFree(dqst); hipFree(dlse); hipFree(ddelta); hipFree(dplan);
  }
  return 0;
}

static int cmp_main(int argc, char** argv) {
  const std::string ta = argv[2], tb = argv[3];
  int bad = 0;
  for (const Case& c : CASES) {
    FILE* fa = fopen(("/tmp/attnp_" + ta + "_" + c.name + ".bin").c_str(), "rb");
    FILE* fb = fopen(("/tmp/attnp_" + tb + "_" + c.name + ".bin").c_str(), "rb");
    if (!fa || !fb) { if (fa) fclose(fa); if (fb) fclose(fb); continue; }
    const size_t nout = (size_t)c.b * c.n * c.h * 64;
    std::vector<uint16_t> a(nout), b(nout);
    if (fread(a.data(), 2, nout, fa) != nout || fread(b.data(), 2, nout, fb) != nout) { printf("%-10s short dump\n", c.name); bad++; fclose(fa); fclose(fb); continue; }
    size_t diff = 0, first = (size_t)-1; double mx = 0;
    for (size_t i = 0; i < nout; i++) if (a[i] != b[i]) { if (!diff) first = i; diff++; mx = std::max(mx, (double)fabsf(bf2f(a[i]) - bf2f(b[i]))); }
    // the rest of the dumps (lse, gradients) byte by byte
    size_t rest = 0, rdiff = 0; int ca, cb;
    while ((ca = fgetc(fa)) != EOF && (cb = fgetc(fb)) != EOF) { rest++; if (ca != cb) rdiff++; }
    if (diff) {                                                       // where: differing elements per 128-row query tile (all samples / heads)
      std::vector<size_t> per((c.n + 127) / 128, 0);
      for (size_t i = 0; i < nout; i++) if (a[i] != b[i]) per[((i / ((size_t)c.h * 64)) % c.n) / 128]++;
      printf("%-10s differing elements per query tile:", c.name); for (size_t x : per) printf(" %zu", x); printf("\n");
      if (getenv("ATTNP_SHOW")) {                                     // the first differing row of every 128-row query tile (sample 0 first)
        std::vector<char> seen((c.n + 127) / 128, 0);
        for (size_t i = 0; i < nout; i++) if (a[i] != b[i]) {
          const size_t tile = ((i / ((size_t)c.h * 64)) % c.n) / 128;
          if (seen[tile]) continue;
          seen[tile] = 1;
          const size_t r0 = i / 64 * 64;
          const int ncol = getenv("ATTNP_VCODE") ? (c.n + 31) / 32 : 64;
          printf("  token %zu head %zu\n   A:", r0 / ((size_t)c.h * 64), (r0 / 64) % c.h);
          for (int k = 0; k < ncol; k++) printf(" %.3f", bf2f(a[r0 + k])); printf("\n   B:");
          for (int k = 0; k < ncol; k++) printf(" %.3f", bf2f(b[r0 + k])); printf("\n");
        }
      }
    }
    printf("%-10s out: %zu of %zu elements differ (max abs %.3g", c.name, diff, nout, mx);
    if (diff) printf(", first at token %zu head %zu col %zu", first / ((size_t)c.h * 64), (first / 64) % c.h, first % 64);
    printf(") ; lse / gradients: %zu of %zu bytes differ\n", rdiff, rest);
    if (diff || rdiff) bad++;
    fclose(fa); fclose(fb);
  }
  printf(bad ? "DIFFERENCES in %d case(s)\n" : "ALL IDENTICAL\n", bad);
  return bad ? 1 : 0;
}

int main(int argc, char** argv) {
  if (argc >= 3 && !strcmp(argv[1], "run")) return run_main(argc, argv);
  if (argc >= 4 && !strcmp(argv[1], "cmp")) return cmp_main(argc, argv);
  fprintf(stderr, "usage: attn_probe run <tag> [case ...] | attn_probe cmp <tagA> <tagB>\n");
  return 2;
}
