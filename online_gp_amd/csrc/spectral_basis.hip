// Reduced Kronecker-eigenbasis ("spectral") Woodbury factor: device pieces.
//
// For a smooth stationary kernel the prior Kt = kron_q K_q / sigma2 has a tiny numerical rank: with K_q = V_q diag(ev_q) V_q^T
// the r tensor-product eigenvectors b_j = kron_q V_q[:, S_q[j]] with the largest eigenvalues carry all but a 1e-6..1e-9
// fraction of trace(Kt) (r ~ 400 at the default RBF hyper-parameters on a 50^3 grid).  In that basis the reference's
// Woodbury algebra (BFN:343-404: Q = I + L^T Kt L, pred_mean, pred_cov) is a dense r x r problem,
//     G = B^T A B,   C = I + Lam^(1/2) G Lam^(1/2) = chol chol^T,   M ~= B Lam^(1/2) C^-1 Lam^(1/2) B^T,
// which lives on the MFMA GEMM / Cholesky / TRSM kernels of dense.hip.  The two kernels here are what is not a GEMM:
//
//   k_basis_project   F[p, j] = scale_p * colscale_j * prod_q (w_q(x_p) . V_q[j0_q(x_p) .. +3, S_q[j]])      (= (W B)[p, j])
//                     (+ the prior variance prod_q w_q^T K_q w_q of each point, for the truncation bound)
//   k_pair_reduce     D_q[a, a'] = sum over basis pairs (j, j') that agree in every dim but q, S_q[j] = a, S_q[j'] = a', of
//                     Wt[j, j'] * prod_{p != q} ev_p[S_p[j]]  -- the contraction that turns an r x r weight matrix into the
//                     gradient of sum_jj' Wt[j,j'] b_j^T Kt b_j' w.r.t. the Toeplitz columns (MLL backward, BWM:19-51)
//
// Always fp64 (the factor is small; fp64 MFMA is 78 TF), x in the model's dtype.
#include "wiski_common.h"

constexpr int SPB_KMAX = 32;      // eigenvectors kept per dim (table columns)

template <typename real>
__global__ __launch_bounds__(256) void k_basis_project(GridDev<double> G, const real* __restrict__ x, int64_t n, const double* __restrict__ V, int kmax,
                                                       const int32_t* __restrict__ S, int r, const real* __restrict__ scale,
                                                       const double* __restrict__ colscale, const double* __restrict__ tcol, double* __restrict__ F,
                                                       int64_t ldf, double* __restrict__ prior, int32_t* __restrict__ err) {
  __shared__ double s_P[4][WISKI_MAX_DIM][SPB_KMAX];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, d = G.d;
  for (int64_t p = (int64_t)blockIdx.x * 4 + wave; p < n; p += (int64_t)gridDim.x * 4) {
    // per-dim taps (every lane computes them: 4 x d cubic evaluations), then lane a < kmax projects dim q onto eigenvector a
    double w[WISKI_MAX_DIM][4];
    int j0[WISKI_MAX_DIM];
    bool ok = true;
    for (int q = 0; q < d; ++q) {
      const double xq = (double)x[p * d + q];
      int j = dim_stencil<double>(xq, G.g0[q], G.h[q], G.hi[q], G.g[q], w[q]);
      if (j < 0) {
        ok = false;
        j = 0;
        for (int c = 0; c < 4; ++c) w[q][c] = 0.0;
      }
      j0[q] = j;
    }
    if (!ok && lane == 0 && err) atomicOr(err, 1);
    int voff = 0;
    for (int q = 0; q < d; ++q) {
      if (lane < kmax) {
        const double* __restrict__ vq = V + (int64_t)voff + (int64_t)j0[q] * kmax + lane;
        s_P[wave][q][lane] = w[q][0] * vq[0] + w[q][1] * vq[kmax] + w[q][2] * vq[2 * kmax] + w[q][3] * vq[3 * kmax];
      }
      voff += G.g[q] * kmax;
    }
    if (prior && lane == 0) {
      double pr = ok ? 1.0 : 0.0;
      int toff = 0;
      for (int q = 0; q < d; ++q) {
        double qf = 0;
        for (int a = 0; a < 4; ++a)
          for (int b = 0; b < 4; ++b) qf += w[q][a] * w[q][b] * tcol[toff + (a > b ? a - b : b - a)];
        pr *= qf;
        toff += G.g[q];
      }
      prior[p] = pr;
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const double sc = scale ? (double)scale[p] : 1.0;
    for (int j = lane; j < r; j += 64) {
      double v = sc;
      for (int q = 0; q < d; ++q) v *= s_P[wave][q][S[(int64_t)q * r + j]];
      if (colscale) v *= colscale[j];
      F[p * ldf + j] = v;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// D[q][a][a'] (kmax x kmax per dim, zeroed by the caller) from the r x r weight matrix Wt (row-major, ld = r).
__global__ __launch_bounds__(256) void k_pair_reduce(int d, int r, int kmax, const double* __restrict__ Wt, const int32_t* __restrict__ S,
                                                     const double* __restrict__ ev, double* __restrict__ D) {
  extern __shared__ double s_D[];                 // [d][kmax][kmax]
  const int nD = d * kmax * kmax;
  for (int i = threadIdx.x; i < nD; i += blockDim.x) s_D[i] = 0.0;
  __syncthreads();
  const int64_t npair = (int64_t)r * r;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < npair; e += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(e / r), jp = (int)(e % r);
    int sj[WISKI_MAX_DIM], sp[WISKI_MAX_DIM];
    int ndiff = 0, qdiff = -1;
    for (int q = 0; q < d; ++q) {
      sj[q] = S[(int64_t)q * r + j];
      sp[q] = S[(int64_t)q * r + jp];
      if (sj[q] != sp[q]) { ++ndiff; qdiff = q; }
    }
    if (ndiff > 1) continue;
    const double wv = Wt[e];
    if (wv == 0.0) continue;
    for (int q = 0; q < d; ++q) {
      if (ndiff == 1 && q != qdiff) continue;     // the pair differs in dim qdiff: it only feeds D[qdiff]
      double pr = wv;
      for (int p = 0; p < d; ++p)
        if (p != q) pr *= ev[p * kmax + sj[p]];
      unsafeAtomicAdd(&s_D[(q * kmax + sj[q]) * kmax + sp[q]], pr);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nD; i += blockDim.x)
    if (s_D[i] != 0.0) unsafeAtomicAdd(&D[i], s_D[i]);
}

template <typename real>
static int basis_project_impl(const wiski_grid* grid, const real* d_x, int64_t n, const double* d_V, int32_t kmax, const int32_t* d_S, int32_t r,
                              const real* d_scale, const double* d_colscale, const double* d_tcol, double* d_F, int64_t ldf, double* d_prior,
                              int32_t* d_err, void* stream) {
  GridDev<double> G;
  int rc = make_grid_dev<double>(grid, &G);
  if (rc) return rc;
  if (!d_x || !d_V || !d_S || !d_F || n < 0 || r < 1 || kmax < 1 || kmax > SPB_KMAX || ldf < r) return WISKI_E_BADARG;
  if (d_prior && !d_tcol) return WISKI_E_BADARG;
  if (n == 0) return WISKI_OK;
  int64_t nb = (n + 3) / 4;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL((k_basis_project<real>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, G, d_x, n, d_V, (int)kmax, d_S, (int)r, d_scale,
                     d_colscale, d_tcol, d_F, ldf, d_prior, d_err);
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}

extern "C" {
int wiski_basis_project_f32(const wiski_grid* grid, const float* d_x, int64_t n, const double* d_V, int32_t kmax, const int32_t* d_S, int32_t r,
                            const float* d_scale, const double* d_colscale, const double* d_tcol, double* d_F, int64_t ldf, double* d_prior,
                            int32_t* d_err, void* stream) {
  return basis_project_impl<float>(grid, d_x, n, d_V, kmax, d_S, r, d_scale, d_colscale, d_tcol, d_F, ldf, d_prior, d_err, stream);
}
int wiski_basis_project_f64(const wiski_grid* grid, const double* d_x, int64_t n, const double* d_V, int32_t kmax, const int32_t* d_S, int32_t r,
                            const double* d_scale, const double* d_colscale, const double* d_tcol, double* d_F, int64_t ldf, double* d_prior,
                            int32_t* d_err, void* stream) {
  return basis_project_impl<double>(grid, d_x, n, d_V, kmax, d_S, r, d_scale, d_colscale, d_tcol, d_F, ldf, d_prior, d_err, stream);
}
int wiski_basis_pair_reduce(int32_t d, int32_t r, int32_t kmax, const double* d_Wt, const int32_t* d_S, const double* d_ev, double* d_D, void* stream) {
  if (d < 1 || d > WISKI_MAX_DIM || r < 1 || kmax < 1 || kmax > SPB_KMAX || !d_Wt || !d_S || !d_ev || !d_D) return WISKI_E_BADARG;
  const int64_t npair = (int64_t)r * r;
  int64_t nb = (npair + 255) / 256;
  if (nb > 512) nb = 512;
  const size_t sh = (size_t)d * kmax * kmax * sizeof(double);
  hipLaunchKernelGGL(k_pair_reduce, dim3((unsigned)nb), dim3(256), sh, (hipStream_t)stream, (int)d, (int)r, (int)kmax, d_Wt, d_S, d_ev, d_D);
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}
}
