// Micro-benchmark of the fused spectral kernels (standalone, hipEvent timed).
#include "../../online_gp_amd/csrc/spectral.hip"
#include <cstdio>
#include <vector>
int main() {
  wiski_grid g; g.d = 3; for (int q = 0; q < 3; ++q) { g.g[q] = 50; g.g0[q] = -1.19; g.h[q] = 0.048; }
  GridDev<float> G; make_grid_dev<float>(&g, &G);
  const int m = G.m, k = 1;
  float *r, *w0, *w1, *ty, *evec, *evals, *big; double* rho;
  hipMalloc(&r, m * 4); hipMalloc(&w0, 2 * m * 4); hipMalloc(&w1, 2 * m * 4); hipMalloc(&ty, 2 * m * 4);
  hipMalloc(&evec, 3 * 2500 * 4); hipMalloc(&evals, 150 * 4); hipMalloc(&rho, 8);
  size_t bigN = 400u << 20; hipMalloc(&big, bigN);
  std::vector<float> h(m, 1.0f); hipMemcpy(r, h.data(), m * 4, hipMemcpyHostToDevice);
  std::vector<float> hv(7500, 0.01f); hipMemcpy(evec, hv.data(), 7500 * 4, hipMemcpyHostToDevice);
  std::vector<float> he(150, 0.5f); hipMemcpy(evals, he.data(), 150 * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int flush = 0; flush < 2; ++flush) {
    float tot = 0; int reps = 20;
    for (int it = 0; it < reps + 2; ++it) {
      if (flush) hipMemsetAsync(big, 0, bigN, 0);
      hipEventRecord(e0, 0);
      launch_spectral_fused<float>(G, evec, evec, evals, 1.4f, 3.0f, r, k, w0, w1, ty, rho, 0);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (it >= 2) tot += ms;
    }
    printf("fused precond (3 kernels) flush=%d: %.1f us\n", flush, tot / reps * 1e3);
  }
#ifdef SPEC_TIMING
  long long h_dbg[16];
  hipMemcpyFromSymbol(h_dbg, HIP_SYMBOL(g_spec_dbg), sizeof(h_dbg));
  for (int i = 1; i < 6; ++i) printf("slab phase %d: %lld cycles\n", i, h_dbg[i] - h_dbg[i - 1]);
  printf("load phase: issue %lld | commit %lld | barrier %lld\n", h_dbg[6] - h_dbg[0], h_dbg[7] - h_dbg[6], h_dbg[1] - h_dbg[7]);
#endif
  return 0;
}
