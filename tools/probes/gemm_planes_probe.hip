// Standalone probe for the PRE-SPLIT plane GEMMs (round 3): fp32 results on the bf16 matrix cores with the operands already
// split into three bf16 planes in memory (panel layout [3][cols/16][rows][16], hi / mid / lo), staged by direct-to-LDS loads (global_load_lds,
// 16 bytes per lane) into an NS-stage ring — no VALU split, no ds_write, counted vmcnt across raw barriers.
//   tn    : C[M,N]  = A[M,K] . B[N,K]^T          (Linear forward / input gradient; contraction along the rows' memory order)
//   wgrad : dW[N,K] = dY[M,N]^T . X[M,K]         (contraction over the rows: ds_read_b64_tr_b16 transposes on the way out of LDS)
// Build + run (GPU box):  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/gemm_planes_probe.hip -o gpurun_out/gpp -ldl && gpurun_out/gpp
// It checks every variant against an fp64 host reference on sampled rows and times it next to the library's in-kernel-split
// kernels (dlopen of partdistillation_amd/libpd_hip.so when present).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "gemm_planes.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

using namespace pdplanes;

static float frand(uint64_t &s) { s = s * 6364136223846793005ull + 1442695040888963407ull; return (float)((s >> 40) & 0xffffff) / 8388608.f - 1.f; }

struct Dev {
  float *f = nullptr; bf16_t *p = nullptr; std::vector<float> h; int R, C;
  void init(int R_, int C_, uint64_t seed, float scale) {
    R = R_; C = C_; h.resize((size_t)R * C);
    uint64_t s = seed;
    for (auto &v : h) v = frand(s) * scale;
    CK(hipMalloc(&f, h.size() * 4)); CK(hipMemcpy(f, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&p, h.size() * 6));
    hipLaunchKernelGGL(split3_panels, dim3((unsigned)((h.size() / 4 + 255) / 256)), dim3(256), 0, 0, f, C, p, R, C);
    CK(hipDeviceSynchronize());
  }
};

template <typename F> static float time_us(F &&f, int iters = 20)
{
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) f();
  CK(hipEventRecord(a));
  for (int i = 0; i < iters; ++i) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms * 1000.f / iters;
}

typedef int (*tn_fn)(const float *, const float *, const float *, float *, int, int, int, int, int, int, int, void *);
typedef int (*wg_fn)(const float *, const float *, float *, float *, float *, int64_t, int, int, int, int, int, int, void *);
typedef int64_t (*wsz_fn)(int, int);

static double check_tn(const Dev &A, const Dev &B, const std::vector<float> &bias, const float *dC, int M, int N, int K, bool relu)
{
  std::vector<float> row(N);
  double worst = 0, scale = 0;
  for (int s = 0; s < 48; ++s) {
    const int m = (int)(((int64_t)s * 7919 + 13) % M);
    CK(hipMemcpy(row.data(), dC + (size_t)m * N, (size_t)N * 4, hipMemcpyDeviceToHost));
    for (int n = 0; n < N; ++n) {
      double acc = bias.empty() ? 0.0 : bias[n];
      for (int k = 0; k < K; ++k) acc += (double)A.h[(size_t)m * K + k] * (double)B.h[(size_t)n * K + k];
      if (relu && acc < 0) acc = 0;
      worst = fmax(worst, fabs(acc - row[n])); scale = fmax(scale, fabs(acc));
    }
  }
  return worst / scale;
}

int main(int argc, char **argv)
{
  const int M = argc > 1 ? atoi(argv[1]) : 43008;
  void *lib = dlopen("partdistillation_amd/libpd_hip.so", RTLD_NOW);
  tn_fn lib_tn = lib ? (tn_fn)dlsym(lib, "pd_gemm_tn_f32x3") : nullptr;
  wg_fn lib_wg = lib ? (wg_fn)dlsym(lib, "pd_gemm_wgrad_acc_f32x3_ws") : nullptr;
  wsz_fn lib_wsz = lib ? (wsz_fn)dlsym(lib, "pd_gemm_wgrad_f32x3_ws_floats") : nullptr;
  printf("M = %d, library %s\n", M, lib ? "loaded" : "absent");
  struct Shape { int N, K; bool relu; } shapes[] = {{1024, 256, true}, {256, 1024, false}, {256, 256, false}};
  for (auto sh : shapes) {
    const int N = sh.N, K = sh.K;
    Dev A, B; A.init(M, K, 1 + N, 1.f); B.init(N, K, 77 + K, 0.1f);
    std::vector<float> bias(N); { uint64_t s = 5; for (auto &v : bias) v = frand(s); }
    float *dbias, *dC; CK(hipMalloc(&dbias, N * 4)); CK(hipMemcpy(dbias, bias.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dC, (size_t)M * N * 4));
    const double gf = 2.0 * M * N * K * 1e-9;
    printf("---- tn  C[%d,%d] = A[%d,%d] B[%d,%d]^T  (%.1f GFLOP fp32, x6 bf16 products) relu=%d\n", M, N, M, K, N, K, gf, (int)sh.relu);
    if (lib_tn) {
      float us = time_us([&] { lib_tn(A.f, B.f, dbias, dC, M, N, K, K, K, N, sh.relu, nullptr); });
      printf("  library in-kernel split          : %8.1f us  %6.1f TF fp32-eq  err %.2e\n", us, gf / us * 1e3, check_tn(A, B, bias, dC, M, N, K, sh.relu));
    }
    struct V { const char *name; int wvm, ns, orient, abl, sched, delay; } vars[] = {
        {"planes 256x256 NS3 dma-spread", 4, 3, 0, 0, 1, 0}, {"planes 128x256 NS2 dma-spread", 2, 2, 0, 0, 1, 0}};
    for (auto v : vars) {
      CK(hipMemset(dC, 0xff, (size_t)M * N * 4));
      auto run = [&] { return launch_tn_planes(v.wvm, v.ns, v.orient, v.abl ? 10 + v.abl : sh.relu ? 1 : 0, v.sched, A.p, (int64_t)M * K, B.p, (int64_t)N * K, dbias, dC, N, M, N, K, nullptr, v.delay); };
      if (run() != 0) { printf("  %-32s : unsupported\n", v.name); continue; }
      CK(hipDeviceSynchronize());
      const double err = v.abl ? -1. : check_tn(A, B, bias, dC, M, N, K, sh.relu);
      float us = time_us([&] { run(); });
      {
        unsigned long long *tr; const int nb = 64; CK(hipMalloc(&tr, nb * 8 * 4 * 8)); CK(hipMemset(tr, 0, nb * 8 * 4 * 8));
        launch_tn_planes(v.wvm, v.ns, v.orient, sh.relu ? 1 : 0, v.sched, A.p, (int64_t)M * K, B.p, (int64_t)N * K, dbias, dC, N, M, N, K, nullptr, 0, tr);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(nb * 8 * 4); CK(hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull; for (auto x : h) if (x && x < t0) t0 = x;
        printf("    trace (clocks since the first stamp; wave 0 of every 64th workgroup): start / loop entered / loop done / stores issued\n");
        for (int b = 0; b < nb; ++b) { const unsigned long long *e = &h[(size_t)b * 32]; if (!e[0]) continue;
          printf("      wg %4d: %8llu %8llu %8llu %8llu   (loop %llu, epilogue %llu)\n", b * 64, e[0] - t0, e[1] - t0, e[2] - t0, e[3] - t0, e[2] - e[1], e[3] - e[2]); }
        CK(hipFree(tr));
      }
      printf("  %-36s : %8.1f us  %6.1f TF fp32-eq (%.2f of 2.5 PF bf16)  err %.2e\n", v.name, us, gf / us * 1e3, 6 * gf / us * 1e3 / 2500., err);
    }
    {   // the same kernels on ZERO operands: what the clock does when the matrix pipe toggles nothing (DVFS, guide rule 25)
      CK(hipMemset(A.p, 0, (size_t)M * K * 6)); CK(hipMemset(B.p, 0, (size_t)N * K * 6)); CK(hipMemset(A.f, 0, (size_t)M * K * 4)); CK(hipMemset(B.f, 0, (size_t)N * K * 4));
      float us = time_us([&] { launch_tn_planes(4, 3, 0, sh.relu ? 1 : 0, 1, A.p, (int64_t)M * K, B.p, (int64_t)N * K, dbias, dC, N, M, N, K, nullptr); });
      printf("  planes 256x256 NS3 dma-spread on ZEROS : %8.1f us\n", us);
      if (lib_tn) { us = time_us([&] { lib_tn(A.f, B.f, dbias, dC, M, N, K, K, K, N, sh.relu, nullptr); }); printf("  library on ZEROS                       : %8.1f us\n", us); }
    }
    CK(hipFree(A.f)); CK(hipFree(A.p)); CK(hipFree(B.f)); CK(hipFree(B.p)); CK(hipFree(dC)); CK(hipFree(dbias));
  }
  // ---- weight gradients
  struct WS { int N, K; } wshapes[] = {{1024, 256}};
  for (auto sh : wshapes) {
    const int N = sh.N, K = sh.K;
    Dev Y, X; Y.init(M, N, 3 + N, 0.05f); X.init(M, K, 9 + K, 1.f);
    float *dW, *ws; CK(hipMalloc(&dW, (size_t)N * K * 4));
    const int64_t wsf = 64ll << 20; CK(hipMalloc(&ws, wsf * 4));
    const double gf = 2.0 * M * N * K * 1e-9;
    printf("---- wgrad  dW[%d,%d] = dY[%d,%d]^T X[%d,%d]  (%.1f GFLOP fp32)\n", N, K, M, N, M, K, gf);
    std::vector<double> ref((size_t)8 * K);                       // rows 0..7 of dW (n = 0..7 scattered) on the host
    std::vector<int> nidx = {0, 1, N / 2 + 3, N - 1, 37 % N, 130 % N, 255 % N, (N - 66 + N) % N};
    for (int r = 0; r < 8; ++r)
      for (int k = 0; k < K; ++k) {
        double a = 0;
        for (int m = 0; m < M; ++m) a += (double)Y.h[(size_t)m * N + nidx[r]] * (double)X.h[(size_t)m * K + k];
        ref[(size_t)r * K + k] = a;
      }
    auto check = [&] {
      std::vector<float> row(K);
      double worst = 0, scale = 0;
      for (int r = 0; r < 8; ++r) {
        CK(hipMemcpy(row.data(), dW + (size_t)nidx[r] * K, (size_t)K * 4, hipMemcpyDeviceToHost));
        for (int k = 0; k < K; ++k) { worst = fmax(worst, fabs(ref[(size_t)r * K + k] - row[k])); scale = fmax(scale, fabs(ref[(size_t)r * K + k])); }
      }
      return worst / scale;
    };
    if (lib_wg) {
      CK(hipMemset(dW, 0, (size_t)N * K * 4));
      lib_wg(Y.f, X.f, dW, nullptr, ws, lib_wsz(N, K), M, N, K, N, K, K, nullptr); CK(hipDeviceSynchronize());
      const double err = check();
      float us = time_us([&] { lib_wg(Y.f, X.f, dW, nullptr, ws, lib_wsz(N, K), M, N, K, N, K, K, nullptr); });
      printf("  library in-kernel split (tr)     : %8.1f us  %6.1f TF fp32-eq  err %.2e\n", us, gf / us * 1e3, err);
    }
    for (int ns = 2; ns <= 3; ++ns) {
      CK(hipMemset(dW, 0, (size_t)N * K * 4));
      auto run = [&] { return launch_wgrad_planes(ns, Y.p, (int64_t)M * N, X.p, (int64_t)M * K, dW, K, ws, wsf, M, N, K, nullptr); };
      if (run() != 0) { printf("  planes wgrad NS%d : unsupported\n", ns); continue; }
      CK(hipDeviceSynchronize());
      const double err = check();
      float us = time_us([&] { run(); });
      printf("  planes wgrad 128x128 NS%d         : %8.1f us  %6.1f TF fp32-eq (%.2f of 2.5 PF bf16)  err %.2e\n", ns, us, gf / us * 1e3, 6 * gf / us * 1e3 / 2500., err);
    }
    CK(hipFree(Y.f)); CK(hipFree(Y.p)); CK(hipFree(X.f)); CK(hipFree(X.p)); CK(hipFree(dW)); CK(hipFree(ws));
  }
  return 0;
}
