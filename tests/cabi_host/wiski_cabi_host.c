/* TEST INFRASTRUCTURE -- a plain C99 host of the C ABI (include/wiski.h), no Python and no torch anywhere:
 * what a C / C++ caller of the reference's hot path would write (last sentence).
 *
 *   absorb n points            wiski_scatter_stats_sym_f64     (BFN:31-60,155-171 + URLT:58)
 *   posterior mean solve       wiski_pcg_f64, P = Kt           (BFN:368-383)
 *   predictive mean            wiski_gather_f64                (BFN:206-210)
 *   stored-rows predictive MVM wiski_interp_f64 + wiski_gather_ell_f64 / wiski_gather_ell_grid_f64  (BFN:206-210,235)
 *
 * and the same numbers from the CPU oracle (oracle/wiski_oracle.c, compiled into this program: the checker, never the thing
 * measured).  Prints the deviations and returns 0 iff all are within tolerance.  Built by __graft_entry__.build() with
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ tests/cabi_host/wiski_cabi_host.c oracle/wiski_oracle.c -lwiski_hip -lamdhip64 -lm
 * and run by tests/test_cabi_host_gpu.py on the GPU box. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "wiski.h"

/* the oracle's entry points (oracle/wiski_oracle_impl.h, REAL = double) */
int wo_gather_f64(const double* x, long n, int d, const double* g0, const double* h, const int* g, const double* V, long m, int k, double* out);
int wo_scatter_stats_f64(const double* x, const double* y, const double* wa, const double* wb, const double* noise, long n, int d, const double* g0,
                         const double* h, const int* g, long m, double* b, double* A_st, double* c_ld);
int wo_pcg_f64(const double* A_st, const double* tcol, int d, const int* g, long m, double kscale, const double* RHS, int k, double tol, int max_iter,
               double* U, double* rel_res_out);

#define CHECK_HIP(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %d at line %d\n", (int)e_, __LINE__); return 3; } } while (0)
#define CHECK_W(e) do { int rc_ = (e); if (rc_ != 0) { fprintf(stderr, "wiski error %d at line %d\n", rc_, __LINE__); return 4; } } while (0)

static double lcg(uint64_t* s) {                 /* uniform in [0, 1) */
  *s = *s * 6364136223846793005ull + 1442695040888963407ull;
  return (double)(*s >> 11) / 9007199254740992.0;
}
static double maxrel(const double* a, const double* b, long n) {
  double num = 0, den = 0;
  for (long i = 0; i < n; ++i) { double e = fabs(a[i] - b[i]); if (e > num) num = e; if (fabs(b[i]) > den) den = fabs(b[i]); }
  return den > 0 ? num / den : num;
}

int main(void) {
  enum { D = 3, G = 12, N = 3000, Q = 20000 };   /* Q * 4^D taps >= 2^20: the stored-rows product takes the LDS-DMA staged kernel */
  const double lo = -1.1, hi = 1.1, ell = 0.6931471805599453, os = 0.6931471805599453, sigma2 = 0.6931471805599453;
  wiski_grid grid;
  memset(&grid, 0, sizeof grid);
  grid.d = D;
  int g[D];
  double g0[D], h[D];
  long m = 1;
  for (int q = 0; q < D; ++q) {                  /* gpytorch's grid: delta = (hi - lo) / (g - 2), linspace(lo - delta, hi + delta, g) */
    const double delta = (hi - lo) / (G - 2);
    g[q] = G; g0[q] = lo - delta; h[q] = ((hi + delta) - g0[q]) / (G - 1);
    grid.g[q] = G; grid.g0[q] = g0[q]; grid.h[q] = h[q];
    m *= G;
  }
  const long R = 343, H = (R + 1) / 2, T = 64;
  uint64_t seed = 12345;
  double *x = malloc(sizeof(double) * N * D), *y = malloc(sizeof(double) * N), *one = malloc(sizeof(double) * N), *xq = malloc(sizeof(double) * Q * D);
  for (long i = 0; i < N; ++i) {
    double s = 0;
    for (int q = 0; q < D; ++q) { x[i * D + q] = 2 * lcg(&seed) - 1; s += x[i * D + q]; }
    y[i] = sin(2 * s) + 0.1 * (lcg(&seed) - 0.5);
    one[i] = 1.0;
  }
  for (long i = 0; i < Q * D; ++i) xq[i] = 2.2 * lcg(&seed) - 1.1;      /* the whole extent: boundary cells included */
  double* tcol = malloc(sizeof(double) * D * G);
  for (int q = 0; q < D; ++q)
    for (int j = 0; j < G; ++j) { const double r = j * h[q] / ell; tcol[q * G + j] = os * exp(-0.5 * r * r); }

  /* ---- oracle (CPU) */
  double *b_o = calloc(m, sizeof(double)), *A_o = calloc(R * m, sizeof(double)), cld_o[2] = {0, 0}, *U_o = calloc(m, sizeof(double));
  double* mean_o = malloc(sizeof(double) * Q);
  if (wo_scatter_stats_f64(x, y, one, one, one, N, D, g0, h, g, m, b_o, A_o, cld_o)) return 5;
  double rr = 0;
  const int it_o = wo_pcg_f64(A_o, tcol, D, g, m, 1.0 / sigma2, b_o, 1, 1e-13, 2000, U_o, &rr);
  if (wo_gather_f64(xq, Q, D, g0, h, g, U_o, m, 1, mean_o)) return 5;

  /* ---- the library (GPU), through the C ABI only */
  hipStream_t s;
  CHECK_HIP(hipStreamCreate(&s));
  double *dx, *dy, *dw, *dxq, *dtcol, *db, *dA, *dU, *dZ, *dmean, *dstats, *dval, *dmean_ell, *dmean_grid, *dpack = NULL;
  int32_t *derr, *didx;
  CHECK_HIP(hipMalloc((void**)&dx, sizeof(double) * N * D)); CHECK_HIP(hipMalloc((void**)&dy, sizeof(double) * N)); CHECK_HIP(hipMalloc((void**)&dw, sizeof(double) * N));
  CHECK_HIP(hipMalloc((void**)&dxq, sizeof(double) * Q * D)); CHECK_HIP(hipMalloc((void**)&dtcol, sizeof(double) * D * G));
  CHECK_HIP(hipMalloc((void**)&db, sizeof(double) * m)); CHECK_HIP(hipMalloc((void**)&dA, sizeof(double) * H * m));
  CHECK_HIP(hipMalloc((void**)&dU, sizeof(double) * m)); CHECK_HIP(hipMalloc((void**)&dZ, sizeof(double) * m));
  CHECK_HIP(hipMalloc((void**)&dmean, sizeof(double) * Q)); CHECK_HIP(hipMalloc((void**)&dmean_ell, sizeof(double) * Q)); CHECK_HIP(hipMalloc((void**)&dmean_grid, sizeof(double) * Q));
  CHECK_HIP(hipMalloc((void**)&dstats, sizeof(double) * 2)); CHECK_HIP(hipMalloc((void**)&derr, sizeof(int32_t)));
  CHECK_HIP(hipMalloc((void**)&didx, sizeof(int32_t) * Q * T)); CHECK_HIP(hipMalloc((void**)&dval, sizeof(double) * Q * T));
  CHECK_HIP(hipMemcpy(dx, x, sizeof(double) * N * D, hipMemcpyHostToDevice)); CHECK_HIP(hipMemcpy(dy, y, sizeof(double) * N, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(dw, one, sizeof(double) * N, hipMemcpyHostToDevice)); CHECK_HIP(hipMemcpy(dxq, xq, sizeof(double) * Q * D, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(dtcol, tcol, sizeof(double) * D * G, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemset(db, 0, sizeof(double) * m)); CHECK_HIP(hipMemset(dA, 0, sizeof(double) * H * m)); CHECK_HIP(hipMemset(dstats, 0, sizeof(double) * 2));
  CHECK_HIP(hipMemset(derr, 0, sizeof(int32_t))); CHECK_HIP(hipMemset(dU, 0, sizeof(double) * m)); CHECK_HIP(hipMemset(dZ, 0, sizeof(double) * m));

  CHECK_W(wiski_scatter_stats_sym_f64(&grid, dx, dy, dw, dw, dw, N, db, dA, dstats, derr, s));
  const int64_t wb = wiski_pcg_workspace_bytes(&grid, 1, 2000, 8);
  void* dwork;
  CHECK_HIP(hipMalloc(&dwork, (size_t)wb));
  int32_t iters = 0, herr = 0;
  double relres = 0;
  CHECK_W(wiski_pcg_f64(&grid, dA, dtcol, 1.0 / sigma2, NULL, NULL, NULL, 0.0, db, 1, dU, dZ, 0, 1e-13, 2000, 10, 0, dwork, wb, &iters, &relres, derr, &herr, 1, NULL, s));
  CHECK_W(wiski_gather_f64(&grid, dxq, Q, dU, 1, 0, dmean, derr, s));
  CHECK_W(wiski_interp_f64(&grid, dxq, Q, didx, dval, derr, s));
  CHECK_W(wiski_gather_ell_f64(didx, dval, Q, (int32_t)T, dU, dmean_ell, s));
  const int64_t npack = wiski_gather_ell_pack_elems(&grid);
  if (npack > 0) CHECK_HIP(hipMalloc((void**)&dpack, sizeof(double) * (size_t)npack));
  CHECK_W(wiski_gather_ell_grid_f64(&grid, didx, dval, Q, dU, dpack, dmean_grid, s));
  CHECK_HIP(hipStreamSynchronize(s));

  double *b_g = malloc(sizeof(double) * m), stats_g[2], *mean_g = malloc(sizeof(double) * Q), *mean_e = malloc(sizeof(double) * Q), *mean_gr = malloc(sizeof(double) * Q);
  int32_t err_g = 0;
  CHECK_HIP(hipMemcpy(b_g, db, sizeof(double) * m, hipMemcpyDeviceToHost)); CHECK_HIP(hipMemcpy(stats_g, dstats, sizeof(double) * 2, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(mean_g, dmean, sizeof(double) * Q, hipMemcpyDeviceToHost)); CHECK_HIP(hipMemcpy(mean_e, dmean_ell, sizeof(double) * Q, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(mean_gr, dmean_grid, sizeof(double) * Q, hipMemcpyDeviceToHost)); CHECK_HIP(hipMemcpy(&err_g, derr, sizeof(int32_t), hipMemcpyDeviceToHost));

  const double e_b = maxrel(b_g, b_o, m), e_c = fabs(stats_g[0] - cld_o[0]) / fabs(cld_o[0]), e_m = maxrel(mean_g, mean_o, Q), e_e = maxrel(mean_e, mean_o, Q),
               e_g = maxrel(mean_gr, mean_o, Q);
  printf("wiski_version %d; grid %d^%d (m = %ld), %d points absorbed, %d queries\n", wiski_version(), G, D, m, N, Q);
  printf("W^T y: max rel dev vs oracle %.3e;  y^T y: %.3e;  out-of-grid flag %d (poll: %d)\n", e_b, e_c, (int)err_g, (int)herr);
  printf("solve: %d iterations (oracle %d), relative residual %.2e (oracle %.2e)\n", (int)iters, it_o, relres, rr);
  printf("predictive mean vs oracle: fused gather %.3e, stored rows %.3e, stored rows (grid-aware entry) %.3e\n", e_m, e_e, e_g);
  const int ok = e_b < 1e-12 && e_c < 1e-12 && e_m < 1e-8 && e_e < 1e-8 && e_g < 1e-8 && err_g == 0 && herr == 0;
  printf(ok ? "PARITY OK\n" : "PARITY FAILED\n");
  return ok ? 0 : 1;
}
