// Stand-alone probe of tfx_gemm_nt (no PyTorch: starts in seconds on a fresh GPU box).  Built by tools/build_ow_probe.sh into tools/ow_probe.
//
//   TFX_NT_OW=0 tools/ow_probe run pp      # every case on the ping-pong kernel: prints time / TFLOP/s / hash, dumps the outputs to /tmp/owp_pp_<case>.bin
//   TFX_NT_OW=1 tools/ow_probe run ow      # the same cases on the one-wave-per-SIMD kernel
//   tools/ow_probe cmp pp ow               # byte comparison of the two dumps (the two kernels accumulate in the same order: bit-identical)
//
// The kernel choice is an environment switch the library reads once per process, hence two runs.
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

struct Case { const char* name; int M, N, K, epi; int lda_pad; bool bias; bool timed; };
static const Case CASES[] = {
  // correctness corners: one / two / three K-tiles (the loop's three entry paths), ragged M and N, padded lda
  // (run with TFX_NT_PP_MIN=1 so that these few-tile shapes take the 256 x 256 family; M > 1024 keeps them off the decode kernels)
  {"k64",      2304,  512,   64, 0, 0, false, false},
  {"k128",     2304,  512,  128, 0, 0, false, false},
  {"k192",     2100,  520,  192, 0, 0, true,  false},
  {"k256",     2304,  768,  256, 0, 64, false, false},
  {"ragged",  66000, 1544,  512, 0, 0, true,  true},
  {"f32",     65536,  448,  512, 1, 0, true,  true},
  {"resid",   65536,  512,  512, 3, 0, true,  true},
  {"geglu",   65536, 2816,  512, 4, 0, true,  true},
  {"geglub",  65536, 1408,  512, 5, 0, false, true},
  // in-step shapes (config 2 / 3) and the library yardstick's square
  {"n512k512",   65536,  512,  512, 0, 0, false, true},
  {"n512k1408",  65536,  512, 1408, 0, 0, false, true},
  {"n512k2816",  65536,  512, 2816, 0, 0, false, true},
  {"n1544k512",  65536, 1544,  512, 0, 0, false, true},
  {"n1024k1024", 65536, 1024, 1024, 0, 0, false, true},
  {"n1024k2752", 65536, 1024, 2752, 0, 0, false, true},
  {"n5504k1024", 65536, 5504, 1024, 0, 0, false, true},
  {"sq4096",      8192, 4096, 4096, 0, 0, false, true},
};

static uint64_t fnv(const void* p, size_t n) {
  const uint8_t* b = (const uint8_t*)p; uint64_t h = 1469598103934665603ull;
  for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
  return h;
}
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static void fill_bf16(std::vector<uint16_t>& v, uint64_t seed, float scale) {
  uint64_t s = seed * 0x9E3779B97F4A7C15ull + 12345;
  for (auto& x : v) { s = s * 6364136223846793005ull + 1442695040888963407ull; x = f2bf(scale * ((float)((s >> 40) & 0xffff) / 32768.f - 1.f)); }
}

// ---- weight-gradient products: tools/ow_probe tn <tag> ; tools/ow_probe tncmp <tagA> <tagB>  (split-M sums through fp32 atomics: compared by relative error)
struct TnCase { const char* name; int M, N, K, splits; };
static const TnCase TN_CASES[] = {
  {"t_one",     4096,  512,  512, 1},          // one block per tile: deterministic
  {"t_2816",   65536, 2816,  512, 0}, {"t_1544", 65536, 1544, 512, 0}, {"t_1408", 65536, 512, 1408, 0}, {"t_1024", 65536, 1024, 1024, 0},
  {"t_5504",   65536, 5504, 1024, 0}, {"t_2752", 65536, 1024, 2752, 0}, {"t_4096", 65536, 1024, 4096, 0},
  {"t_cond",    2048, 24576, 2048, 1},         // the AdaLN conditioning weights: more tiles than CUs, one block per tile
};
// tools/ow_probe tnsum: the folded bias gradient on an all-ones A (every column sum = M): which output rows miss how much
static int tnsum_main() {
  const char* libp = getenv("TFX_LIB") ? getenv("TFX_LIB") : "transfusion_pytorch_amd/lib/libtfx_hip.so";
  void* h = dlopen(libp, RTLD_NOW);
  if (!h) { fprintf(stderr, "dlopen %s: %s\n", libp, dlerror()); return 2; }
  auto gemm = (int (*)(const tfx_gemm_tn_args*, void*))dlsym(h, "tfx_gemm_tn");
  hipStream_t st; CK(hipStreamCreate(&st));
  const int M = 4096, N = 1544, K = 512;
  std::vector<uint16_t> hA((size_t)M * N, 0x3f80), hB((size_t)M * K); fill_bf16(hB, 5, 0.05f);
  uint16_t *dA, *dB; float *dC, *dS; CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dB, hB.size() * 2)); CK(hipMalloc(&dC, (size_t)N * K * 4)); CK(hipMalloc(&dS, N * 4));
  CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
  for (int splits : {1, 8}) {
    CK(hipMemset(dC, 0, (size_t)N * K * 4)); CK(hipMemset(dS, 0, N * 4));
    tfx_gemm_tn_args a; memset(&a, 0, sizeof(a));
    a.A = dA; a.lda = N; a.a_cols = N; a.B = dB; a.ldb = K; a.b_cols = K; a.M = M; a.N = N; a.K = K; a.C = dC; a.ldc = K; a.k_valid = K; a.splits = splits; a.accumulate = 1; a.alpha = 1.f; a.colsum = dS;
    int rc = gemm(&a, (void*)st); CK(hipStreamSynchronize(st));
    std::vector<float> s(N); CK(hipMemcpy(s.data(), dS, N * 4, hipMemcpyDeviceToHost));
    printf("[tnsum] splits %d rc %d: ", splits, rc);
    int bad = 0;
    for (int n = 0; n < N; n++) if (s[n] != (float)M) { if (bad < 24) printf(" n=%d:%g", n, s[n]); bad++; }
    printf("  -> %d of %d wrong\n", bad, N);
  }
  return 0;
}

static int tn_main(int argc, char** argv) {
  if (!strcmp(argv[1], "tncmp")) {
    int bad = 0;
    for (const TnCase& c : TN_CASES) {
      std::string fa = std::string("/tmp/owp_") + argv[2] + "_" + c.name + ".bin", fb = std::string("/tmp/owp_") + argv[3] + "_" + c.name + ".bin";
      FILE* a = fopen(fa.c_str(), "rb"); FILE* b = fopen(fb.c_str(), "rb");
      if (!a || !b) { if (a) fclose(a); if (b) fclose(b); printf("[tncmp] %-8s MISSING\n", c.name); continue; }
      const size_t n = (size_t)c.N * c.K; std::vector<float> da(n), db(n);
      if (fread(da.data(), 4, n, a) != n || fread(db.data(), 4, n, b) != n) { printf("[tncmp] read error\n"); return 3; }
      fclose(a); fclose(b);
      double num = 0, den = 0, maxd = 0; size_t nbad = 0;
      for (size_t i = 0; i < n; i++) { const double d = (double)da[i] - db[i]; num += d * d; den += (double)da[i] * da[i]; if (fabs(d) > maxd) maxd = fabs(d); if (!(fabs(d) <= 1e-2 * (fabs((double)da[i]) + 1.0))) nbad++; }
      const double rel = sqrt(num / (den + 1e-30));
      printf("[tncmp] %-8s rel-Frobenius %.3g  max |d| %.4g  outliers %zu  %s\n", c.name, rel, maxd, nbad, (rel < 2e-5 && nbad == 0) ? "ok" : "DIFFERENT");
      if (!(rel < 2e-5 && nbad == 0)) bad++;
    }
    printf("[tncmp] %s\n", bad ? "MISMATCH" : "ALL WITHIN fp32 SUMMATION ORDER");
    return bad ? 1 : 0;
  }
  const char* tag = argv[2];
  const char* libp = getenv("TFX_LIB") ? getenv("TFX_LIB") : "transfusion_pytorch_amd/lib/libtfx_hip.so";
  void* h = dlopen(libp, RTLD_NOW);
  if (!h) { fprintf(stderr, "dlopen %s: %s\n", libp, dlerror()); return 2; }
  auto gemm = (int (*)(const tfx_gemm_tn_args*, void*))dlsym(h, "tfx_gemm_tn");
  auto plan = (int (*)(const tfx_gemm_tn_args*, int32_t*, int32_t*, int32_t*, int32_t*))dlsym(h, "tfx_gemm_tn_plan");
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("[tn %s] TFX_TN_OW=%s\n", tag, getenv("TFX_TN_OW") ? getenv("TFX_TN_OW") : "(unset)");
  for (TnCase c : TN_CASES) {
    if (getenv("OWP_TN_M") && c.splits == 0) c.M = atoi(getenv("OWP_TN_M"));        // (row count override: time(M) = a + b M separates the atomics / prologue from the row loop)
    std::vector<uint16_t> hA((size_t)c.M * c.N), hB((size_t)c.M * c.K);
    fill_bf16(hA, 3 + c.N, 1.f); fill_bf16(hB, 5 + c.K, 0.05f);
    uint16_t *dA, *dB; float* dC; CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dB, hB.size() * 2)); CK(hipMalloc(&dC, (size_t)c.N * c.K * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(dC, 0, (size_t)c.N * c.K * 4));
    tfx_gemm_tn_args a; memset(&a, 0, sizeof(a));
    a.A = dA; a.lda = c.N; a.a_cols = c.N; a.B = dB; a.ldb = c.K; a.b_cols = c.K; a.M = c.M; a.N = c.N; a.K = c.K; a.C = dC; a.ldc = c.K; a.k_valid = c.K; a.splits = c.splits; a.accumulate = 1; a.alpha = 1.f;
    int32_t kind = -9, tiles = 0, splits = 0, grid = 0; if (plan) plan(&a, &kind, &tiles, &splits, &grid);
    int rc = gemm(&a, (void*)st); CK(hipStreamSynchronize(st));
    std::vector<float> out((size_t)c.N * c.K); CK(hipMemcpy(out.data(), dC, out.size() * 4, hipMemcpyDeviceToHost));
    { std::string f = std::string("/tmp/owp_") + tag + "_" + c.name + ".bin"; FILE* fp = fopen(f.c_str(), "wb"); if (fp) { fwrite(out.data(), 4, out.size(), fp); fclose(fp); } }
    // OWP_REPS: timed launches per case (default 20).  Short bursts run at boost clocks; a few thousand launches reach the power-limited steady state of a training step
    const int reps = getenv("OWP_REPS") ? atoi(getenv("OWP_REPS")) : 20;
    for (int i = 0; i < 3 + reps / 2; i++) gemm(&a, (void*)st);
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; i++) gemm(&a, (void*)st);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); const double us = ms * 1000.0 / reps;
    const double fl = 2.0 * c.M * (double)c.N * c.K;
    printf("[%s] %-8s M %6d N %5d K %5d rc %d kind %d tiles %4d splits %3d grid %4d  %9.1f us %8.1f TF/s  hash %016llx\n", tag, c.name, c.M, c.N, c.K, rc, kind, tiles, splits, grid, us, fl / us * 1e-6, (unsigned long long)fnv(out.data(), out.size() * 4));
    fflush(stdout);
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
  }
  return 0;
}

int main(int argc, char** argv) {
  if (argc >= 2 && !strcmp(argv[1], "tnsum")) return tnsum_main();
  if (argc >= 3 && (!strcmp(argv[1], "tn") || !strcmp(argv[1], "tncmp"))) return tn_main(argc, argv);
  if (argc >= 4 && !strcmp(argv[1], "cmp")) {
    int bad = 0;
    for (const Case& c : CASES) {
      for (int which = 0; which < 2; which++) {
        std::string fa = std::string("/tmp/owp_") + argv[2] + "_" + c.name + (which ? "_c2" : "") + ".bin", fb = std::string("/tmp/owp_") + argv[3] + "_" + c.name + (which ? "_c2" : "") + ".bin";
        FILE* a = fopen(fa.c_str(), "rb"); FILE* b = fopen(fb.c_str(), "rb");
        if (!a || !b) { if (a) fclose(a); if (b) fclose(b); if (!which) { printf("[cmp] %-12s MISSING\n", c.name); bad++; } continue; }
        fseek(a, 0, SEEK_END); size_t na = ftell(a); fseek(a, 0, SEEK_SET); fseek(b, 0, SEEK_END); size_t nb = ftell(b); fseek(b, 0, SEEK_SET);
        std::vector<uint8_t> da(na), db(nb);
        if (fread(da.data(), 1, na, a) != na || fread(db.data(), 1, nb, b) != nb) { printf("[cmp] read error\n"); return 3; }
        fclose(a); fclose(b);
        const bool f32 = c.epi == 1;
        const size_t es = f32 ? 4 : 2, n = na / es;
        const int ldc = (c.epi == 4 && !which) ? c.N : (c.epi == 4 && which) ? c.N / 2 : (c.epi == 5) ? 2 * c.N : c.N;
        size_t mism = 0, first = (size_t)-1; double maxd = 0;
        if (na != nb) { printf("[cmp] %-12s size differs\n", c.name); bad++; continue; }
        for (size_t i = 0; i < n; i++) {
          if (memcmp(&da[i * es], &db[i * es], es)) {
            float x, y;
            if (f32) { memcpy(&x, &da[i * 4], 4); memcpy(&y, &db[i * 4], 4); } else { uint16_t hx, hy; memcpy(&hx, &da[i * 2], 2); memcpy(&hy, &db[i * 2], 2); x = bf2f(hx); y = bf2f(hy); }
            const double d = fabs((double)x - (double)y); if (d > maxd || d != d) maxd = d;
            if (first == (size_t)-1) first = i; mism++;
          }
        }
        printf("[cmp] %-12s %s %zu elements, %zu differ", c.name, which ? "C2" : "C ", n, mism);
        if (mism) {
          printf(", max |d| %.4g, first at row %zu col %zu; rows%%256 histogram of the first 2000:", maxd, first / ldc, first % ldc);
          // where are they?  count mismatches by (row mod 256) / 32 and (col mod 256) / 32
          int hr[8] = {0}, hc[8] = {0}; size_t seen = 0;
          for (size_t i = 0; i < n && seen < 200000; i++) if (memcmp(&da[i * es], &db[i * es], es)) { hr[((i / ldc) % 256) / 32]++; hc[((i % ldc) % 256) / 32]++; seen++; }
          printf(" rowblk"); for (int k = 0; k < 8; k++) printf(" %d", hr[k]); printf(" colblk"); for (int k = 0; k < 8; k++) printf(" %d", hc[k]);
          bad++;
        }
        printf("\n");
      }
    }
    printf("[cmp] %s\n", bad ? "MISMATCH" : "ALL IDENTICAL");
    return bad ? 1 : 0;
  }
  if (argc < 3 || strcmp(argv[1], "run")) { fprintf(stderr, "usage: ow_probe run <tag> [case-substring] | ow_probe cmp <tagA> <tagB>\n"); return 2; }
  const char* tag = argv[2];
  const char* only = argc > 3 ? argv[3] : nullptr;
  const char* libp = getenv("TFX_LIB") ? getenv("TFX_LIB") : "transfusion_pytorch_amd/lib/libtfx_hip.so";
  void* h = dlopen(libp, RTLD_NOW);
  if (!h) { fprintf(stderr, "dlopen %s: %s\n", libp, dlerror()); return 2; }
  auto gemm = (int (*)(const tfx_gemm_nt_args*, void*))dlsym(h, "tfx_gemm_nt");
  auto plan = (int (*)(const tfx_gemm_nt_args*, int32_t*, int32_t*))dlsym(h, "tfx_gemm_nt_plan");
  if (!gemm) { fprintf(stderr, "no tfx_gemm_nt\n"); return 2; }
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("[run %s] TFX_NT_OW=%s\n", tag, getenv("TFX_NT_OW") ? getenv("TFX_NT_OW") : "(unset)");
  for (const Case& c : CASES) {
    if (only) {                                              // comma-separated substrings
      bool hit = false; std::string o(only); size_t b = 0;
      while (b <= o.size()) { size_t e = o.find(',', b); if (e == std::string::npos) e = o.size(); if (e > b && strstr(c.name, o.substr(b, e - b).c_str())) hit = true; b = e + 1; }
      if (!hit) continue;
    }
    const int lda = c.K + c.lda_pad, ldb = c.K;
    std::vector<uint16_t> hA((size_t)c.M * lda), hB((size_t)c.N * ldb);
    fill_bf16(hA, 1 + c.M + c.K, 1.f); fill_bf16(hB, 7 + c.N + c.K, 0.05f);
    uint16_t *dA, *dB; CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dB, hB.size() * 2));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
    tfx_gemm_nt_args a; memset(&a, 0, sizeof(a));
    a.A = dA; a.lda = lda; a.B = dB; a.ldb = ldb; a.M = c.M; a.N = c.N; a.K = c.K; a.epi = c.epi;
    const size_t es = c.epi == 1 ? 4 : 2;
    const int ldc = c.epi == 5 ? 2 * c.N : c.N;
    const size_t nC = (size_t)c.M * ldc, nC2 = c.epi == 4 ? (size_t)c.M * (c.N / 2) : 0;
    void *dC, *dC2 = nullptr; CK(hipMalloc(&dC, nC * es)); CK(hipMemset(dC, 0xff, nC * es));
    a.C = dC; a.ldc = ldc;
    if (nC2) { CK(hipMalloc(&dC2, nC2 * 2)); CK(hipMemset(dC2, 0xff, nC2 * 2)); a.C2 = dC2; a.ldc2 = c.N / 2; }
    float* dBias = nullptr;
    if (c.bias) { std::vector<float> hb(c.N); for (int i = 0; i < c.N; i++) hb[i] = 0.01f * (float)((i * 37) % 41 - 20); CK(hipMalloc(&dBias, c.N * 4)); CK(hipMemcpy(dBias, hb.data(), c.N * 4, hipMemcpyHostToDevice)); a.bias = dBias; }
    uint16_t* dR = nullptr;
    if (c.epi == 3) { std::vector<uint16_t> hr((size_t)c.M * c.N); fill_bf16(hr, 99, 1.f); CK(hipMalloc(&dR, hr.size() * 2)); CK(hipMemcpy(dR, hr.data(), hr.size() * 2, hipMemcpyHostToDevice)); a.R = dR; a.ldr = c.N; }
    if (c.epi == 5) { std::vector<uint16_t> hx((size_t)c.M * 2 * c.N); fill_bf16(hx, 55, 1.f); CK(hipMalloc(&dR, hx.size() * 2)); CK(hipMemcpy(dR, hx.data(), hx.size() * 2, hipMemcpyHostToDevice)); a.aux = dR; a.ldaux = 2 * c.N; }
    int32_t kind = -1, grid = -1; if (plan) plan(&a, &kind, &grid);
    int rc = gemm(&a, (void*)st); CK(hipStreamSynchronize(st));
    if (rc) { printf("[%s] %-12s rc %d\n", tag, c.name, rc); continue; }
    std::vector<uint8_t> out(nC * es); CK(hipMemcpy(out.data(), dC, nC * es, hipMemcpyDeviceToHost));
    uint64_t hh = fnv(out.data(), out.size());
    { std::string f = std::string("/tmp/owp_") + tag + "_" + c.name + ".bin"; FILE* fp = fopen(f.c_str(), "wb"); if (fp) { fwrite(out.data(), 1, out.size(), fp); fclose(fp); } }
    if (nC2) { std::vector<uint8_t> o2(nC2 * 2); CK(hipMemcpy(o2.data(), dC2, nC2 * 2, hipMemcpyDeviceToHost)); hh ^= fnv(o2.data(), o2.size()) * 31; std::string f = std::string("/tmp/owp_") + tag + "_" + c.name + "_c2.bin"; FILE* fp = fopen(f.c_str(), "wb"); if (fp) { fwrite(o2.data(), 1, o2.size(), fp); fclose(fp); } }
    double us = 0;
    if (c.timed) {
      const int reps = getenv("OWP_REPS") ? atoi(getenv("OWP_REPS")) : 20;
      for (int i = 0; i < 3 + reps / 2; i++) gemm(&a, (void*)st);
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < reps; i++) gemm(&a, (void*)st);
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); us = ms * 1000.0 / reps;
    }
    const double fl = 2.0 * c.M * (double)c.N * c.K;
    printf("[%s] %-12s M %6d N %5d K %5d epi %d kind %d grid %5d  %9.1f us %8.1f TF/s  hash %016llx\n", tag, c.name, c.M, c.N, c.K, c.epi, kind, grid, us, us > 0 ? fl / us * 1e-6 : 0.0, (unsigned long long)hh);
    fflush(stdout);
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC)); if (dC2) CK(hipFree(dC2)); if (dBias) CK(hipFree(dBias)); if (dR) CK(hipFree(dR));
  }
  return 0;
}
