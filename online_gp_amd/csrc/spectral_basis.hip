// Reduced Kronecker-eigenbasis ("spectral") Woodbury factor: device pieces.
//
// For a smooth stationary kernel the prior Kt = kron_q K_q / sigma2 has a tiny numerical rank: with K_q = V_q diag(ev_q) V_q^T
// the r tensor-product eigenvectors b_j = kron_q V_q[:, S_q[j]] with the largest eigenvalues carry all but a 1e-6..1e-9
// fraction of trace(Kt) (r ~ 400 at the default RBF hyper-parameters on a 50^3 grid).  In that basis the reference's
// Woodbury algebra (BFN:343-404: Q = I + L^T Kt L, pred_mean, pred_cov) is a dense r x r problem,
//     G = B^T A B,   C = I + Lam^(1/2) G Lam^(1/2) = chol chol^T,   M ~= B Lam^(1/2) C^-1 Lam^(1/2) B^T,
// which lives on the MFMA GEMM / Cholesky / TRSM kernels of dense.hip.  The kernels here are what is not a GEMM -- the first two
// below, and (further down, each with its own header) the device-side refresh after a hyper-parameter step: k_eig_update
// (eigenvectors of the Toeplitz factors refined from the previous ones), k_basis_change (change of basis + verdict),
// k_woodbury_c, k_tail_a / b / c (everything derived from the factor), k_spectral_var (query variances), k_lag_grad (MLL backward).
//
//   k_basis_project   F[p, j] = scale_p * colscale_j * prod_q (w_q(x_p) . V_q[j0_q(x_p) .. +3, S_q[j]])      (= (W B)[p, j])
//                     (+ the prior variance prod_q w_q^T K_q w_q of each point, for the truncation bound)
//   k_pair_reduce     D_q[a, a'] = sum over basis pairs (j, j') that agree in every dim but q, S_q[j] = a, S_q[j'] = a', of
//                     Wt[j, j'] * prod_{p != q} ev_p[S_p[j]]  -- the contraction that turns an r x r weight matrix into the
//                     gradient of sum_jj' Wt[j,j'] b_j^T Kt b_j' w.r.t. the Toeplitz columns (MLL backward, BWM:19-51)
//
// Always fp64 (the factor is small; fp64 MFMA is 78 TF), x in the model's dtype.
#include "wiski_common.h"
#include <cstdlib>

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
    if (prior && lane == 0 && blockIdx.y == 0) {
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
    // (gridDim.y > 1: few points -- evaluate() of a batch of 1..16, the absorb of the same batch -- with the r columns dealt to
    //  gridDim.y workgroups instead of one wave walking all of them: r = 1000, one point: 20 -> 5 us)
    const int jper = ((r + (int)gridDim.y - 1) / (int)gridDim.y + 63) & ~63;
    const int jlo = (int)blockIdx.y * jper, jhi = jlo + jper < r ? jlo + jper : r;
    for (int j = jlo + lane; j < jhi; j += 64) {
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
  unsigned ny = 1;                                   // few points, many columns: split the columns over workgroups
  if (n <= 64 && r > 128) ny = (unsigned)((r + 127) / 128 < 16 ? (r + 127) / 128 : 16);
  hipLaunchKernelGGL((k_basis_project<real>), dim3((unsigned)nb, ny), dim3(256), 0, (hipStream_t)stream, G, d_x, n, d_V, (int)kmax, d_S, (int)r, d_scale,
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

// ---------------------------------------------------------------- eigen-update ---
// Hyper-parameter steps move the Toeplitz factors a little at a time, and the spectral factor only needs their DOMINANT
// eigenvectors.  Instead of a host eigh of every factor per step (0.4 ms for three 50 x 50 matrices, plus the device-to-host
// copy of the columns that forces a synchronisation), the previous step's eigenvectors are refined on the device:
// two steps of subspace iteration Z = K V, V = orth(Z) (modified Gram-Schmidt, fp64) on kw >= kmax working vectors, then
// Rayleigh-Ritz -- H = V^T K V (kw x kw), parallel cyclic Jacobi -- and V <- V U with the Ritz values sorted descending.
// The spectrum of a smooth kernel's factor decays geometrically, so the wanted vectors converge at (lam_{kw+1} / lam_j)^2 per
// call from an O(drift) start; the largest residual |K v - theta v| / theta_1 of the vectors actually used is written out and
// checked by the host one step later (a failed check falls back to the host eigh).  One workgroup per dim; g <= 64, kw <= 32.
constexpr int EIG_G = 64, EIG_K = 32;

#ifdef WISKI_EIG_TIMING                               // phase stamps of block 0 (tools: a -DWISKI_EIG_TIMING build; resid_out must hold 8 + 16 doubles)
#define EIG_STAMP(k) do { __syncthreads(); if (blockIdx.x == 0 && threadIdx.x == 0) resid_out[8 + (k)] = (double)wall_clock64(); } while (0)
#else
#define EIG_STAMP(k) do { } while (0)
#endif
#define WAVE_SYNC()                                   \
  do {                                                \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    __builtin_amdgcn_wave_barrier();                  \
  } while (0)

__device__ __forceinline__ double eig_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = y * (2.0 - x * y);
  return y * (2.0 - x * y);
}
__device__ __forceinline__ double eig_rsq(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * (1.5 - 0.5 * x * y * y);
  return y * (1.5 - 0.5 * x * y * y);
}

__global__ __launch_bounds__(256) void k_eig_update(int d, const int* __restrict__ gs, const double* __restrict__ tcol, const double* __restrict__ Vin,
                                                    int kw, int kuse, double* __restrict__ Vout, double* __restrict__ ev_out,
                                                    double* __restrict__ resid_out, const double* __restrict__ Vref, int kref,
                                                    double* __restrict__ Tq_out, int niter, double resid_ok) {
  __shared__ double sT[EIG_G], sV[EIG_G][EIG_K + 1], sZ[EIG_G][EIG_K + 1], sH[EIG_K][EIG_K + 1], sU[EIG_K][EIG_K + 1], sCS[EIG_K / 2][2], sTh[EIG_K];
  __shared__ int sPr[EIG_K / 2][2], sRank[EIG_K];
  __shared__ double sRed[4], sRed2[4];
  const int q = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
  int g = gs[q], toff = 0, voff = 0;
  for (int p = 0; p < q; ++p) { toff += gs[p]; voff += gs[p] * kw; }
  if (t < g) sT[t] = tcol[toff + t];
  for (int e = t; e < g * kw; e += 256) sV[e / kw][e % kw] = Vin[voff + e];
  __syncthreads();
  auto applyK = [&]() {                               // sZ = K sV,  K[i][j] = sT[|i - j|]
    for (int e = t; e < g * kw; e += 256) {
      const int i = e / kw, c = e % kw;
      double acc = 0;
      for (int j = 0; j < g; ++j) acc += sT[i > j ? i - j : j - i] * sV[j][c];
      sZ[i][c] = acc;
    }
    __syncthreads();
  };
  EIG_STAMP(0);
  // Pass 0: Rayleigh-Ritz in the span of the previous vectors (+ guard vectors), preceded by `niter` steps of subspace iteration
  // (default 0: after one Adam step the new eigenvectors lie in that span to ~1e-14 -- measured over 60 steps at lr 1e-3 / 1e-2 --,
  // because the guard vectors carry what rotates in).  The residual decides: above `resid_ok` (the caller passes an eighth of the limit
  // its verdict applies) the kernel makes a second pass with two steps of subspace iteration, which re-centres the span.
  __shared__ double sRes;
  const double jac_tol = resid_ok < 1.0 ? fmax(1e-30, resid_ok * resid_ok * 1e-4) : 1e-30;
  for (int pass = 0; pass < 2; ++pass) {
  for (int iter = 0; iter < (pass == 0 ? niter : 2); ++iter) {
    applyK();
    // modified Gram-Schmidt on the columns of sZ (rows = lanes of a wave, g <= 64); column c is normalised by wave 0, the
    // later columns are dealt to the 4 waves
    for (int c = 0; c < kw; ++c) {
      if (wv == 0) {
        const double v = lane < g ? sZ[lane][c] : 0.0;
        const double nrm = sqrt(wave_reduce_sum<double>(v * v));
        if (lane < g) sZ[lane][c] = nrm > 0 ? v / nrm : 0.0;
      }
      __syncthreads();
      for (int c2 = c + 1 + wv; c2 < kw; c2 += 4) {
        const double qv = lane < g ? sZ[lane][c] : 0.0, zv = lane < g ? sZ[lane][c2] : 0.0;
        const double dot = wave_reduce_sum<double>(qv * zv);
        if (lane < g) sZ[lane][c2] = zv - dot * qv;
      }
      __syncthreads();
    }
    for (int e = t; e < g * kw; e += 256) sV[e / kw][e % kw] = sZ[e / kw][e % kw];
    __syncthreads();
  }
  EIG_STAMP(1);
  applyK();                                           // sZ = K V
  for (int e = t; e < kw * kw; e += 256) {            // H = V^T K V (symmetrised)
    const int a = e / kw, b = e % kw;
    double acc = 0;
    for (int i = 0; i < g; ++i) acc += sV[i][a] * sZ[i][b];
    sH[a][b] = acc;
    sU[a][b] = a == b ? 1.0 : 0.0;
  }
  __syncthreads();
  for (int e = t; e < kw * kw; e += 256) {
    const int a = e / kw, b = e % kw;
    if (a < b) { const double s = 0.5 * (sH[a][b] + sH[b][a]); sH[a][b] = s; sH[b][a] = s; }
  }
  __syncthreads();
  EIG_STAMP(2);
  // Rayleigh-Ritz by parallel cyclic Jacobi on ONE wave (the ~100 dependent phases of a few sweeps then cost their instructions, not a
  // workgroup barrier each).  A step rotates kw / 2 disjoint pairs at once; disjoint rotations commute, so the whole step is ONE pass
  // over the matrix:  H' = J^T H J  element by element from the old H (four reads), U' = U J (two reads), written to a second buffer
  // (rows 0 .. and 32 .. of sZ, free here) -- two wave-level synchronisations per step instead of three dependent read-modify-write
  // phases.  Index a is rotated with partner(a) (round robin: (2 step - a) mod (n - 1), the last index against `step`):
  //     x'_a = alpha_a x_a + beta_a x_partner(a),   alpha = c,  beta = -s for the smaller index of the pair, +s for the larger.
  const int n = kw;                                   // (even: checked by the host entry)
  {
    // One wave issues a VALU / LDS instruction every ~6 cycles (tools/ubench/clock_ubench.hip), and a step on ONE wave was ~650 of them
    // (4000 cycles, 35 us per sweep): the element pass is dealt to all four waves (<= 4 elements per thread) between two workgroup
    // barriers; the rotation parameters stay on wave 0.
    double(*Hc)[EIG_K + 1] = sH;
    double(*Uc)[EIG_K + 1] = sU;
    double(*Hn)[EIG_K + 1] = sZ;
    double(*Un)[EIG_K + 1] = sZ + EIG_K;
    double* sAl = &sCS[0][0];                         // alpha[a], a < n <= 32  (sCS holds 32 doubles)
    double* sBe = sTh;                                // beta[a]                (sTh is filled after the sweeps)
    int* sPa = &sPr[0][0];                            // partner[a]
    __shared__ int sStop;
    int ea[EIG_K * EIG_K / 256];                      // this thread's elements (a << 8 | b); slots past n^2 are not stored
#pragma unroll
    for (int i = 0; i < EIG_K * EIG_K / 256; ++i) {
      const int e = t + 256 * i;
      ea[i] = e < n * n ? ((e / n) << 8) | (e % n) : 0;
    }
    for (int sweep = 0; sweep < 12; ++sweep) {
      if (wv == 0) {
        double off2 = 0, dg2 = 0;
        for (int e = lane; e < kw * kw; e += 64) {
          const double v = Hc[e / kw][e % kw];
          if (e / kw == e % kw) dg2 += v * v; else off2 += v * v;
        }
        off2 = wave_reduce_sum<double>(off2);
        dg2 = wave_reduce_sum<double>(dg2);
#ifdef WISKI_EIG_TIMING
        if (blockIdx.x == 0 && lane == 0) { resid_out[16 + pass] = (double)sweep; resid_out[18 + pass] = off2 / dg2; }
#endif
        // (the start is nearly diagonal -- the previous eigenvectors: 1-3 sweeps.  The sweeps stop where the off-diagonal mass is two orders
        //  below the residual the caller accepts: for the fp32 model's 1e-10 that is a sweep less than the 1e-15 the fixed form iterates to)
        if (lane == 0) sStop = off2 <= jac_tol * dg2;
      }
      __syncthreads();
      if (sStop) break;                               // (block-uniform)
      for (int step = 0; step < n - 1; ++step) {
        if (t < n) {
          const int a = t;
          const int pa = a == n - 1 ? step : (a == step ? n - 1 : (2 * step - a + 2 * (n - 1)) % (n - 1));
          const int p_ = a < pa ? a : pa, q_ = a < pa ? pa : a;
          double c = 1.0, sn = 0.0;
          const double apq = Hc[p_][q_];
          if (fabs(apq) > 1e-290) {
            // (reciprocal / reciprocal square root + two Newton steps each instead of IEEE division and square root -- ~50 instead of ~150
            //  dependent fp64 instructions on the step's serial path; c^2 + s^2 = 1 holds to rounding either way)
            const double theta = (Hc[q_][q_] - Hc[p_][p_]) * eig_rcp(2.0 * apq);
            const double at = fabs(theta);
            double tt;
            if (at > 1e100) tt = 0.5 * eig_rcp(theta);
            else {
              const double w2 = at * at + 1.0;
              tt = eig_rcp(at + w2 * eig_rsq(w2));
              tt = theta >= 0 ? tt : -tt;
            }
            c = eig_rsq(tt * tt + 1.0);
            sn = tt * c;
          }
          sAl[a] = c;
          sBe[a] = a == p_ ? -sn : sn;
          sPa[a] = pa;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < EIG_K * EIG_K / 256; ++i) {
          const int a = ea[i] >> 8, b2 = ea[i] & 255;
          const int pa = sPa[a], pb = sPa[b2];
          const double ala = sAl[a], bea = sBe[a], alb = sAl[b2], beb = sBe[b2];
          const double h = ala * (alb * Hc[a][b2] + beb * Hc[a][pb]) + bea * (alb * Hc[pa][b2] + beb * Hc[pa][pb]);
          const double u = alb * Uc[a][b2] + beb * Uc[a][pb];
          if (t + 256 * i < n * n) {
            Hn[a][b2] = h;
            Un[a][b2] = u;
          }
        }
        __syncthreads();
        { double(*tH)[EIG_K + 1] = Hc; Hc = Hn; Hn = tH; }
        { double(*tU)[EIG_K + 1] = Uc; Uc = Un; Un = tU; }
      }
    }
    if (Hc != sH) {                                   // (block-uniform) an odd number of steps: the result sits in the second buffer
      for (int e = t; e < n * n; e += 256) {
        sH[e / n][e % n] = Hc[e / n][e % n];
        sU[e / n][e % n] = Uc[e / n][e % n];
      }
    }
  }
  __syncthreads();
  EIG_STAMP(3);
  if (t < kw) sTh[t] = sH[t][t];
  __syncthreads();
  if (t < kw) {                                       // rank of each Ritz value (descending; ties by index)
    int rk = 0;
    for (int b = 0; b < kw; ++b) rk += (sTh[b] > sTh[t]) || (sTh[b] == sTh[t] && b < t);
    sRank[t] = rk;
  }
  __syncthreads();
  for (int e = t; e < g * kw; e += 256) {             // V <- V U, columns in descending order of the Ritz values
    const int i = e / kw, a = e % kw;
    double acc = 0;
    for (int b = 0; b < kw; ++b) acc += sV[i][b] * sU[b][a];
    sZ[i][sRank[a]] = acc;
  }
  __syncthreads();
  for (int e = t; e < g * kw; e += 256) sV[e / kw][e % kw] = sZ[e / kw][e % kw];
  __syncthreads();
  EIG_STAMP(4);
  // residual of the vectors that are used (the first kuse): max_i |K v - theta v|_i / theta_max
  applyK();
  double worst = 0;
  for (int e = t; e < g * kuse; e += 256) {
    const int i = e / kuse, a = e % kuse;
    int src = 0;                                      // Ritz value of sorted column a
    for (int b = 0; b < kw; ++b) if (sRank[b] == a) src = b;
    const double r = fabs(sZ[i][a] - sTh[src] * sV[i][a]);
    worst = r > worst ? r : worst;
  }
  for (int off = 32; off > 0; off >>= 1) {
    const double o = __shfl_xor(worst, off, 64);
    worst = o > worst ? o : worst;
  }
  EIG_STAMP(6);
  __syncthreads();
  if (lane == 0) sRed[wv] = worst;
  __syncthreads();
  if (t == 0) {
    double w = sRed[0];
    for (int i = 1; i < 4; ++i) w = sRed[i] > w ? sRed[i] : w;
    double tmax = 0;
    for (int b = 0; b < kw; ++b) tmax = sTh[b] > tmax ? sTh[b] : tmax;
    sRes = tmax > 0 ? w / tmax : 0.0;
  }
  __syncthreads();
  if (sRes <= resid_ok) break;                        // (block-uniform)
  }
  // ---- outputs: the vectors (sorted), the Ritz values, T_q = Vref_q^T Vnew_q for the change of basis (while the vectors sit in LDS)
  for (int e = t; e < g * kw; e += 256) Vout[voff + e] = sV[e / kw][e % kw];
  if (t < kw) ev_out[q * kw + sRank[t]] = sTh[t] > 0 ? sTh[t] : 0.0;
  if (t == 0) resid_out[q] = sRes;
  if (Vref && Tq_out) {
    int roff = 0;
    for (int p = 0; p < q; ++p) roff += gs[p] * kref;
    for (int e = t; e < kref * kw; e += 256) {
      const int a = e / kw, b = e % kw;
      double acc = 0;
      for (int i = 0; i < g; ++i) acc += Vref[roff + i * kref + a] * sV[i][b];
      Tq_out[((int64_t)q * SPB_KMAX + a) * SPB_KMAX + b] = acc;
    }
  }
  EIG_STAMP(5);
}

static int eig_update_launch(int32_t d, const int32_t* d_g, const double* d_tcol, const double* d_Vin, int32_t kw, int32_t kuse, double* d_Vout, double* d_ev,
                             double* d_resid, const double* d_Vref, int32_t kref, double* d_Tq, int niter, double resid_ok, void* stream) {
  if (d < 1 || d > WISKI_MAX_DIM || !d_g || !d_tcol || !d_Vin || !d_Vout || !d_ev || !d_resid || kw < 2 || kw > EIG_K || (kw & 1) || kuse < 1 || kuse > kw)
    return WISKI_E_BADARG;
  if ((d_Vref || d_Tq) && (!d_Vref || !d_Tq || kref < 1 || kref > SPB_KMAX)) return WISKI_E_BADARG;
  if (niter < 0 || niter > 8 || !(resid_ok >= 0)) return WISKI_E_BADARG;
  hipLaunchKernelGGL(k_eig_update, dim3((unsigned)d), dim3(256), 0, (hipStream_t)stream, (int)d, d_g, d_tcol, d_Vin, (int)kw, (int)kuse, d_Vout, d_ev, d_resid,
                     d_Vref, (int)kref, d_Tq, niter, resid_ok);
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}
// two steps of subspace iteration, then Rayleigh-Ritz (one pass, whatever the residual)
extern "C" int wiski_basis_eig_update(int32_t d, const int32_t* d_g, const double* d_tcol, const double* d_Vin, int32_t kw, int32_t kuse, double* d_Vout,
                                      double* d_ev, double* d_resid, const double* d_Vref, int32_t kref, double* d_Tq, void* stream) {
  return eig_update_launch(d, d_g, d_tcol, d_Vin, kw, kuse, d_Vout, d_ev, d_resid, d_Vref, kref, d_Tq, 2, 1e300, stream);
}
// Rayleigh-Ritz in the span of the previous vectors first (after `niter` steps of subspace iteration, normally 0); a second pass with two
// steps of subspace iteration only if the relative residual of the first exceeds resid_ok
extern "C" int wiski_basis_eig_update_adaptive(int32_t d, const int32_t* d_g, const double* d_tcol, const double* d_Vin, int32_t kw, int32_t kuse,
                                               double* d_Vout, double* d_ev, double* d_resid, const double* d_Vref, int32_t kref, double* d_Tq,
                                               int32_t niter, double resid_ok, void* stream) {
  return eig_update_launch(d, d_g, d_tcol, d_Vin, kw, kuse, d_Vout, d_ev, d_resid, d_Vref, kref, d_Tq, niter, resid_ok, stream);
}

// ------------------------------------------------------------ change of basis ---
// After wiski_basis_eig_update: everything that connects the refreshed eigenvectors to the factor's reference statistics, in
// ONE launch (it was ~35 small torch launches):
//   T_q = Vref_q^T Vnew_q  (d x 32 x 32, written by wiski_basis_eig_update while the new vectors sat in its LDS)
//   TS[i, j] = prod_q T_q[Sref[q, i], S[q, j]]          the Kronecker-structured change of basis, r_ref x r
//   lam[j]   = prod_q ev[q, S[q, j]]                    eigenvalues of Kuu on the kept index set
//   verdict  = [ max_q resid_q,  1 - sum_j lam_j / trace(Kuu),  r * max_j lam_j (1 - |TS[:, j]|^2) / trace(Kuu) ]
// (eigen-residual of the refresh, trace left out by the kept index set, eigenvalue-weighted defect of the reference span --
// the three numbers the host checks before it uses the refreshed factor).  Column norms are accumulated with atomics into
// work[0 .. r) (zero on entry, re-zeroed by the last block, which also writes lam and the verdict); work[r] is the block counter.
constexpr int BC_ROWS = 8;

__global__ __launch_bounds__(256) void k_basis_change(int d, const int* __restrict__ gs, int kref, int kw, int r_ref, int r, const double* __restrict__ Tq,
                                                      const int32_t* __restrict__ Sref, const int32_t* __restrict__ S,
                                                      const double* __restrict__ ev, const double* __restrict__ tcol, const double* __restrict__ resid,
                                                      double* __restrict__ TS, double* __restrict__ lam, double* __restrict__ work,
                                                      double* __restrict__ verdict) {
  __shared__ double sT[WISKI_MAX_DIM][SPB_KMAX][SPB_KMAX + 1];
  __shared__ double s_red[3][4];
  __shared__ int s_last;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int goff[WISKI_MAX_DIM + 1];
  goff[0] = 0;
  for (int q = 0; q < d; ++q) goff[q + 1] = goff[q] + gs[q];
  for (int e = tid; e < d * SPB_KMAX * SPB_KMAX; e += 256) {
    const int q = e / (SPB_KMAX * SPB_KMAX), a = (e / SPB_KMAX) % SPB_KMAX, b = e % SPB_KMAX;
    sT[q][a][b] = (a < kref && b < kw) ? Tq[e] : 0.0;      // T_q = Vref_q^T Vnew_q from wiski_basis_eig_update
  }
  __syncthreads();
  const int i0 = blockIdx.x * BC_ROWS;
  int sref[BC_ROWS][WISKI_MAX_DIM];
#pragma unroll
  for (int u = 0; u < BC_ROWS; ++u)
    for (int q = 0; q < d; ++q) sref[u][q] = i0 + u < r_ref ? Sref[(int64_t)q * r_ref + i0 + u] : 0;
  for (int j = tid; j < r; j += 256) {
    int sj[WISKI_MAX_DIM];
    for (int q = 0; q < d; ++q) sj[q] = S[(int64_t)q * r + j];
    double nrm = 0;
#pragma unroll
    for (int u = 0; u < BC_ROWS; ++u) {
      if (i0 + u >= r_ref) break;
      double v = 1.0;
      for (int q = 0; q < d; ++q) v *= sT[q][sref[u][q]][sj[q]];
      TS[(int64_t)(i0 + u) * r + j] = v;
      nrm += v * v;
    }
    unsafeAtomicAdd(&work[j], nrm);
  }
  __threadfence();
  __syncthreads();
  if (tid == 0) {
    const unsigned long long done = atomicAdd(reinterpret_cast<unsigned long long*>(work + r), 1ull);
    s_last = done + 1 == gridDim.x;
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  double total = 1.0;
  for (int q = 0; q < d; ++q) total *= (double)(goff[q + 1] - goff[q]) * tcol[goff[q]];
  double sum = 0, wmax = 0;
  for (int j = tid; j < r; j += 256) {
    double l = 1.0;
    for (int q = 0; q < d; ++q) l *= ev[q * kw + S[(int64_t)q * r + j]];
    lam[j] = l;
    const double nrm = __hip_atomic_load(&work[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    work[j] = 0.0;
    const double def = 1.0 - nrm > 0 ? 1.0 - nrm : 0.0;
    sum += l;
    wmax = l * def > wmax ? l * def : wmax;
  }
  sum = wave_reduce_sum<double>(sum);
  for (int off = 32; off > 0; off >>= 1) {
    const double o = __shfl_xor(wmax, off, 64);
    wmax = o > wmax ? o : wmax;
  }
  if (lane == 0) { s_red[0][wv] = sum; s_red[1][wv] = wmax; }
  __syncthreads();
  if (tid == 0) {
    double s = 0, w = 0, rs = 0;
    for (int i = 0; i < 4; ++i) { s += s_red[0][i]; w = s_red[1][i] > w ? s_red[1][i] : w; }
    for (int q = 0; q < d; ++q) rs = resid[q] > rs ? resid[q] : rs;
    verdict[0] = rs;
    verdict[1] = total > 0 ? 1.0 - s / total : 1.0;
    verdict[2] = total > 0 ? w / total * (double)r : 1.0;
    *reinterpret_cast<unsigned long long*>(work + r) = 0ull;
  }
}

// C = I + Lam^1/2 G Lam^1/2 with lam = lam_kuu * kscale (also written out, with its square root): the matrix the spectral
// Woodbury factor factorises, in one launch.
__global__ __launch_bounds__(256) void k_woodbury_c(int r, const double* __restrict__ G, const double* __restrict__ lam_kuu, double kscale,
                                                    double* __restrict__ C, double* __restrict__ lam, double* __restrict__ sq, double* __restrict__ sqG) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)r * r) return;
  const int i = (int)(e / r), j = (int)(e % r);
  const double li = lam_kuu[i] * kscale, lj = lam_kuu[j] * kscale;
  C[e] = sqrt(li) * G[e] * sqrt(lj) + (i == j ? 1.0 : 0.0);
  if (sqG) sqG[e] = sqrt(li) * G[e];                 // Lam^1/2 G: the MLL backward's right-hand side (chol^-1 Lam^1/2 G)
  if (j == 0) { lam[i] = li; sq[i] = sqrt(li); }
}

extern "C" int wiski_basis_change(int32_t d, const int32_t* d_g, int32_t kref, int32_t kw, int32_t r_ref, int32_t r, const double* d_Tq,
                                  const int32_t* d_Sref, const int32_t* d_S, const double* d_ev, const double* d_tcol,
                                  const double* d_resid, double* d_TS, double* d_lam, double* d_work, double* d_verdict, void* stream) {
  if (d < 1 || d > WISKI_MAX_DIM || !d_g || kref < 1 || kref > SPB_KMAX || kw < 1 || kw > SPB_KMAX || r_ref < 1 || r < 1 || !d_Tq || !d_Sref ||
      !d_S || !d_ev || !d_tcol || !d_resid || !d_TS || !d_lam || !d_work || !d_verdict)
    return WISKI_E_BADARG;
  const unsigned nb = (unsigned)((r_ref + BC_ROWS - 1) / BC_ROWS);
  hipLaunchKernelGGL(k_basis_change, dim3(nb), dim3(256), 0, (hipStream_t)stream, (int)d, d_g, (int)kref, (int)kw, (int)r_ref, (int)r, d_Tq, d_Sref,
                     d_S, d_ev, d_tcol, d_resid, d_TS, d_lam, d_work, d_verdict);
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}

// MLL backward, weights of the reduced-basis gradient in one launch:  Wt = g_ld (G - P) + g_b zeta zeta^T  (P = Y2^T Y2 from the GEMM),
// g_kap = sum_i Wt[i, i] lam_kuu[i]  (block 0).  g_b, g_ld: device scalars (the incoming gradients are never read by the host).
__global__ __launch_bounds__(256) void k_mll_weights(int r, const double* __restrict__ G, const double* __restrict__ P, const double* __restrict__ zeta,
                                                     const double* __restrict__ lam_kuu, const double* __restrict__ g_b, const double* __restrict__ g_ld,
                                                     double* __restrict__ Wt, double* __restrict__ g_kap) {
  __shared__ double s_red[16];
  const double gb = g_b[0], gl = g_ld[0];
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < (int64_t)r * r; e += (int64_t)gridDim.x * 256) {
    const int i = (int)(e / r), j = (int)(e % r);
    Wt[e] = gl * (G[e] - P[e]) + gb * zeta[i] * zeta[j];
  }
  if (blockIdx.x == 0) {
    double acc = 0;
    for (int i = threadIdx.x; i < r; i += 256) {
      const int64_t e = (int64_t)i * r + i;
      acc += (gl * (G[e] - P[e]) + gb * zeta[i] * zeta[i]) * lam_kuu[i];
    }
    acc = block_reduce_sum(acc, s_red);
    if (threadIdx.x == 0) g_kap[0] = acc;
  }
}

extern "C" int wiski_mll_weights(int32_t r, const double* d_G, const double* d_P, const double* d_zeta, const double* d_lam_kuu, const double* d_gb,
                                 const double* d_gld, double* d_Wt, double* d_gkap, void* stream) {
  if (r < 1 || !d_G || !d_P || !d_zeta || !d_lam_kuu || !d_gb || !d_gld || !d_Wt || !d_gkap) return WISKI_E_BADARG;
  int64_t nb = ((int64_t)r * r + 255) / 256;
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(k_mll_weights, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (int)r, d_G, d_P, d_zeta, d_lam_kuu, d_gb, d_gld, d_Wt, d_gkap);
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}

extern "C" int wiski_woodbury_c(int32_t r, const double* d_G, const double* d_lam_kuu, double kscale, double* d_C, double* d_lam, double* d_sq, double* d_sqG,
                                void* stream) {
  if (r < 1 || !d_G || !d_lam_kuu || !d_C || !d_lam || !d_sq) return WISKI_E_BADARG;
  const int64_t tot = (int64_t)r * r;
  hipLaunchKernelGGL(k_woodbury_c, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (int)r, d_G, d_lam_kuu, kscale, d_C, d_lam, d_sq, d_sqG);
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}

// ------------------------------------------------------------- lag gradient ---
// MLL backward, last step: the pair-reduced weights D_q (kw x kw per dim, wiski_basis_pair_reduce) back to the Toeplitz columns,
//   g_tcol_q[l] = scale * sum_{|i - j| = l} (V_q D_q V_q^T)[i, j],
// one workgroup per dim (it was two batched GEMMs and a [d, g^2] x [g^2, g] GEMM for which the BLAS picked a 139 us kernel).
__global__ __launch_bounds__(256) void k_lag_grad(int d, const int* __restrict__ gs, int kw, const double* __restrict__ V, const double* __restrict__ D,
                                                  double scale_val, const double* __restrict__ scale_dev, double* __restrict__ out) {
  __shared__ double sV[EIG_G][SPB_KMAX + 1], sU[EIG_G][SPB_KMAX + 1], sD[SPB_KMAX][SPB_KMAX + 1], sAcc[4][EIG_G];
  const int q = blockIdx.x, t = threadIdx.x;
  const int g = gs[q];
  const double scale = scale_dev ? scale_dev[0] : scale_val;      // (device scalar inside a captured graph)
  int toff = 0;
  for (int p = 0; p < q; ++p) toff += gs[p];
  for (int e = t; e < g * kw; e += 256) sV[e / kw][e % kw] = V[(int64_t)toff * kw + e];
  for (int e = t; e < kw * kw; e += 256) sD[e / kw][e % kw] = D[(int64_t)q * kw * kw + e];
  __syncthreads();
  for (int e = t; e < g * kw; e += 256) {
    const int i = e / kw, b = e % kw;
    double acc = 0;
    for (int a = 0; a < kw; ++a) acc += sV[i][a] * sD[a][b];
    sU[i][b] = acc;
  }
  __syncthreads();
  const int l = t & 63, part = t >> 6;                 // lag l, rows i = part, part + 4, ...
  double acc = 0;
  if (l < g) {
    for (int i = part; i + l < g; i += 4) {
      double h1 = 0, h2 = 0;
      for (int b = 0; b < kw; ++b) {
        h1 += sU[i][b] * sV[i + l][b];                  // H[i, i + l]
        h2 += sU[i + l][b] * sV[i][b];                  // H[i + l, i]
      }
      acc += l == 0 ? h1 : h1 + h2;
    }
  }
  sAcc[part][l] = acc;
  __syncthreads();
  if (t < g) out[toff + t] = scale * (sAcc[0][t] + sAcc[1][t] + sAcc[2][t] + sAcc[3][t]);
}

extern "C" int wiski_basis_lag_grad(int32_t d, const int32_t* d_g, int32_t kw, const double* d_V, const double* d_D, double scale, const double* d_scale,
                                    double* d_out, void* stream) {
  if (d < 1 || d > WISKI_MAX_DIM || !d_g || kw < 1 || kw > SPB_KMAX || !d_V || !d_D || !d_out) return WISKI_E_BADARG;
  hipLaunchKernelGGL(k_lag_grad, dim3((unsigned)d), dim3(256), 0, (hipStream_t)stream, (int)d, d_g, (int)kw, d_V, d_D, scale, d_scale, d_out);
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}

// --------------------------------------------------------- query variances ---
// diag_j = |Y[:, j]|^2 (Y = chol^-1 F^T, [r, n]),  tail_j = max(prior_j * kscale - |F[j, :]|^2, 0) (the prior variance the
// reduced basis leaves out of query j): the two vectors a predictive variance is the sum of, in one launch.
__global__ __launch_bounds__(256) void k_spectral_var(int n, int r, const double* __restrict__ Y, const double* __restrict__ F, const double* __restrict__ prior,
                                                      double kscale, double* __restrict__ diag, double* __restrict__ tail) {
  // 64 queries per block (lanes: coalesced across the columns of Y), the r rows dealt to the 4 waves
  __shared__ double s_d[4][64], s_c[4][64];
  const int c = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + c;
  double dsum = 0, cap = 0;
  if (j < n) {
    for (int i = part; i < r; i += 4) {
      const double y = Y[(int64_t)i * n + j];
      dsum += y * y;
    }
    const double* __restrict__ f = F + (int64_t)j * r;
    for (int k = part; k < r; k += 4) cap += f[k] * f[k];
  }
  s_d[part][c] = dsum;
  s_c[part][c] = cap;
  __syncthreads();
  if (part == 0 && j < n) {
    diag[j] = s_d[0][c] + s_d[1][c] + s_d[2][c] + s_d[3][c];
    const double tl = prior[j] * kscale - (s_c[0][c] + s_c[1][c] + s_c[2][c] + s_c[3][c]);
    tail[j] = tl > 0 ? tl : 0.0;
  }
}

// the same for a handful of queries (n < 16: the streaming loop asks for one): a block per query, 256 threads over the rows
__global__ __launch_bounds__(256) void k_spectral_var_few(int n, int r, const double* __restrict__ Y, const double* __restrict__ F, const double* __restrict__ prior,
                                                          double kscale, double* __restrict__ diag, double* __restrict__ tail) {
  __shared__ double s_red[16];
  const int j = blockIdx.x;
  double dsum = 0, cap = 0;
  for (int i = threadIdx.x; i < r; i += 256) {
    const double y = Y[(int64_t)i * n + j], f = F[(int64_t)j * r + i];
    dsum += y * y;
    cap += f * f;
  }
  dsum = block_reduce_sum(dsum, s_red);
  __syncthreads();
  cap = block_reduce_sum(cap, s_red);
  if (threadIdx.x == 0) {
    diag[j] = dsum;
    const double tl = prior[j] * kscale - cap;
    tail[j] = tl > 0 ? tl : 0.0;
  }
}

extern "C" int wiski_spectral_var(int32_t n, int32_t r, const double* d_Y, const double* d_F, const double* d_prior, double kscale, double* d_diag,
                                  double* d_tail, void* stream) {
  if (n < 1 || r < 1 || !d_Y || !d_F || !d_prior || !d_diag || !d_tail) return WISKI_E_BADARG;
  if (n < 16)
    hipLaunchKernelGGL(k_spectral_var_few, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, (int)n, (int)r, d_Y, d_F, d_prior, kscale, d_diag, d_tail);
  else
    hipLaunchKernelGGL(k_spectral_var, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, (hipStream_t)stream, (int)n, (int)r, d_Y, d_F, d_prior, kscale, d_diag, d_tail);
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}

// ------------------------------------------------------------- factor tail ---
// Everything the spectral factor derives from (T, h_ref, sqrt(lam), chol, chol^-1) once the factorisation is done -- it was ten
// small framework launches (two GEMVs from the BLAS at 5-23 us each, products, reductions) on the critical path of every
// hyper-parameter step -- as three launches of 1024-thread workgroups (a matrix-vector product wants many CUs' worth of L2
// bandwidth and many loads in flight: one workgroup doing all three took 119 us):
//   a  hr = T^T h_ref,  v = sqrt(lam) o hr                       64 columns per workgroup, the rows dealt to its 16 waves
//   b  c = chol^-1 v                                              a wave per row
//   c  t = chol^-T c,  coef = sqrt(lam) o t,  zeta = t / sqrt(lam),  bMb = |c|^2,  logdet = 2 sum log diag chol
// out (packed, fp64): hr [r] | c [r] | t [r] | coef [r] | zeta [r] | bMb | logdet | v [r] (scratch).
__global__ __launch_bounds__(1024) void k_tail_a(int r_ref, int r, const double* __restrict__ TS, const double* __restrict__ h_ref, const double* __restrict__ sq,
                                                 double* __restrict__ out) {
  // 16 columns per workgroup (r / 16 workgroups: 63 at r = 1000 -- it was r / 64 = 16, i.e. 16 CUs for an 8 MB matrix-vector product),
  // the rows dealt to 64 groups of 16 lanes (a row segment of 16 doubles = one 128-byte line)
  __shared__ double s_p[64][17];
  const int c = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int j = blockIdx.x * 16 + c;
  double acc = 0;
  if (j < r) {
#pragma unroll 4
    for (int i = rg; i < r_ref; i += 64) acc += TS[(int64_t)i * r + j] * h_ref[i];
  }
  s_p[rg][c] = acc;
  __syncthreads();
  if (rg == 0 && j < r) {
    double hr = 0;
#pragma unroll
    for (int w = 0; w < 64; ++w) hr += s_p[w][c];
    out[j] = hr;
    out[5 * r + 2 + j] = sq[j] * hr;
  }
}

__global__ __launch_bounds__(1024) void k_tail_b(int r, const double* __restrict__ Linv, double* __restrict__ out) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int i = blockIdx.x * 16 + wv;
  if (i >= r) return;
  const double* __restrict__ row = Linv + (int64_t)i * r;
  const double* __restrict__ v = out + 5 * r + 2;
  double acc = 0;
  for (int j = lane; j <= i; j += 64) acc += row[j] * v[j];
  acc = wave_reduce_sum<double>(acc);
  if (lane == 0) out[r + i] = acc;
}

__global__ __launch_bounds__(1024) void k_tail_c(int r, const double* __restrict__ Linv, const double* __restrict__ chol, const double* __restrict__ sq,
                                                 double* __restrict__ out) {
  __shared__ double s_p[64][17], s_red[16];
  const int cc = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int j = blockIdx.x * 16 + cc;                    // (16 columns per workgroup, as k_tail_a)
  const double* __restrict__ c = out + r;
  double acc = 0;
  if (j < r) {
    const int i0 = blockIdx.x * 16;                      // rows below the block's first column (the strict upper part is zero)
#pragma unroll 4
    for (int i = i0 + rg; i < r; i += 64) acc += Linv[(int64_t)i * r + j] * c[i];
  }
  s_p[rg][cc] = acc;
  __syncthreads();
  if (rg == 0 && j < r) {
    double tv = 0;
#pragma unroll
    for (int w = 0; w < 64; ++w) tv += s_p[w][cc];
    const double s = sq[j];
    out[2 * r + j] = tv;
    out[3 * r + j] = s * tv;
    out[4 * r + j] = tv / s;
  }
  if (blockIdx.x == 0) {
    double c2 = 0, ld = 0;
    for (int i = threadIdx.x; i < r; i += 1024) {
      c2 += c[i] * c[i];
      ld += log(chol[(int64_t)i * r + i]);
    }
    __syncthreads();
    c2 = block_reduce_sum(c2, s_red);
    __syncthreads();
    ld = block_reduce_sum(ld, s_red);
    if (threadIdx.x == 0) {
      out[5 * r] = c2;
      out[5 * r + 1] = 2.0 * ld;
    }
  }
}

// h[j] += sum_p F[p][j] t_p, t_p = wby[p] (/ scale[p] when the rows of F carry a scale): the right-hand side of the statistics in the factor's
// reference basis, after the batch's Gram update (was a framework element-wise op + a library gemv).  Columns over threads, points in chunks.
template <typename real>
__global__ __launch_bounds__(256) void k_absorb_h(int n, int r, const double* __restrict__ F, int64_t ldf, const real* __restrict__ wby,
                                                  const real* __restrict__ scale, double* __restrict__ h) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int p0 = blockIdx.y * 64, p1 = p0 + 64 < n ? p0 + 64 : n;
  if (j >= r) return;
  double acc = 0;
  for (int p = p0; p < p1; ++p) {
    const double t = scale ? (double)wby[p] / (double)scale[p] : (double)wby[p];
    acc += F[(int64_t)p * ldf + j] * t;
  }
  if (gridDim.y == 1) h[j] += acc;
  else unsafeAtomicAdd(h + j, acc);
}
extern "C" int wiski_basis_absorb_h_f32(int64_t n, int32_t r, const double* d_F, int64_t ldf, const float* d_wby, const float* d_scale, double* d_h, void* stream) {
  if (n < 0 || r < 1 || !d_F || !d_wby || !d_h || ldf < r || n > (int64_t)1 << 30) return WISKI_E_BADARG;
  if (n == 0) return WISKI_OK;
  hipLaunchKernelGGL((k_absorb_h<float>), dim3((unsigned)((r + 255) / 256), (unsigned)((n + 63) / 64)), dim3(256), 0, (hipStream_t)stream, (int)n, (int)r, d_F, ldf,
                     d_wby, d_scale, d_h);
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}
extern "C" int wiski_basis_absorb_h_f64(int64_t n, int32_t r, const double* d_F, int64_t ldf, const double* d_wby, const double* d_scale, double* d_h, void* stream) {
  if (n < 0 || r < 1 || !d_F || !d_wby || !d_h || ldf < r || n > (int64_t)1 << 30) return WISKI_E_BADARG;
  if (n == 0) return WISKI_OK;
  hipLaunchKernelGGL((k_absorb_h<double>), dim3((unsigned)((r + 255) / 256), (unsigned)((n + 63) / 64)), dim3(256), 0, (hipStream_t)stream, (int)n, (int)r, d_F, ldf,
                     d_wby, d_scale, d_h);
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}

extern "C" int wiski_factor_tail(int32_t r_ref, int32_t r, const double* d_TS, const double* d_href, const double* d_sq, const double* d_Linv,
                                 const double* d_chol, double* d_out, void* stream) {
  if (r_ref < 1 || r < 1 || !d_TS || !d_href || !d_sq || !d_Linv || !d_chol || !d_out) return WISKI_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_tail_a, dim3((unsigned)((r + 15) / 16)), dim3(1024), 0, s, (int)r_ref, (int)r, d_TS, d_href, d_sq, d_out);
  hipLaunchKernelGGL(k_tail_b, dim3((unsigned)((r + 15) / 16)), dim3(1024), 0, s, (int)r, d_Linv, d_out);
  hipLaunchKernelGGL(k_tail_c, dim3((unsigned)((r + 15) / 16)), dim3(1024), 0, s, (int)r, d_Linv, d_chol, d_sq, d_out);
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}

// ------------------------------------------------- evaluate() of a small batch ---
// The reference loop scores every incoming batch before it absorbs it (OSR:56-78: predictive mean and variance of the batch -> rmse,
// nll).  From the factor that is  mean_j = F_j . t,  var_j = (|chol^-1 F_j|^2 + max(prior_j kscale - |F_j|^2, 0)) sigma2  and two
// reductions -- it used to be 18 launches (two BLAS GEMVs, the variance kernel, a dozen framework element-wise ops and reductions,
// the metrics kernel, the packing of the three numbers the host reads).  Here it is ONE launch for n <= 64 queries: workgroup
// (bx, by) takes 8 rows of chol^-1 and 8 queries (rows and query vectors in registers, lane-strided), adds its share of the squared
// norms to the accumulators in d_ws; the workgroups with bx = 0 also form the means and |F_j|^2; the last workgroup to finish (a
// ticket) turns the accumulators into variances and the batch metrics, writes
//     d_out = { rmse, mean nll, out-of-grid flag (d_err), max_j |mean_j| }           (fp64: one host read)
// (and, if asked, the means / latent variances in the data dtype), and zeroes d_ws again.  d_ws: 200 doubles, ZERO on first use.
// (M: 64-wide chunks of a row of chol^-1 a lane holds -- 8 for r <= 512, 16 for r <= 1024: the full-rank factors of the small grids)
template <typename real, int M>
__global__ __launch_bounds__(256) void k_spectral_eval(int n, int r, const double* __restrict__ F, const double* __restrict__ prior, const double* __restrict__ Linv,
                                                       int ldl, const double* __restrict__ t, double kscale, const real* __restrict__ s2p,
                                                       const real* __restrict__ y, const int32_t* __restrict__ err, double* ws, double* __restrict__ out,
                                                       real* __restrict__ mean_out, real* __restrict__ var_out) {
  __shared__ double s_acc[4][8];
  __shared__ double s_red[16];
  __shared__ int s_last;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int q0 = blockIdx.y * 8, nq = n - q0 < 8 ? n - q0 : 8;
  double f[8][M];
#pragma unroll
  for (int qq = 0; qq < 8; ++qq)
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const int k = lane + 64 * m;
      f[qq][m] = (qq < nq && k < r) ? F[(int64_t)(q0 + qq) * r + k] : 0.0;
    }
  double acc[8];
#pragma unroll
  for (int qq = 0; qq < 8; ++qq) acc[qq] = 0.0;
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
    const int i = blockIdx.x * 8 + w * 2 + rr;
    double li[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const int k = lane + 64 * m;
      li[m] = (i < r && k <= i) ? Linv[(int64_t)i * ldl + k] : 0.0;      // (chol^-1 is lower triangular)
    }
#pragma unroll
    for (int qq = 0; qq < 8; ++qq) {
      double d = 0.0;
#pragma unroll
      for (int m = 0; m < M; ++m) d += li[m] * f[qq][m];
      d = wave_reduce_sum<double>(d);
      acc[qq] += d * d;
    }
  }
  if (lane == 0)
#pragma unroll
    for (int qq = 0; qq < 8; ++qq) s_acc[w][qq] = acc[qq];
  __syncthreads();
  if (threadIdx.x < nq) unsafeAtomicAdd(ws + q0 + threadIdx.x, s_acc[0][threadIdx.x] + s_acc[1][threadIdx.x] + s_acc[2][threadIdx.x] + s_acc[3][threadIdx.x]);
  if (blockIdx.x == 0) {
    // means and |F_j|^2 of this workgroup's queries: wave w takes queries w and w + 4
    double tv[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const int k = lane + 64 * m;
      tv[m] = k < r ? t[k] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int qq = w + 4 * u;
      double mu = 0.0, cap = 0.0;
#pragma unroll
      for (int m = 0; m < M; ++m) {
        // (f is indexed by the compile-time unrolled u, w is wave-uniform: select instead of a dynamic register index)
        const double fv = w == 0 ? f[4 * u][m] : w == 1 ? f[4 * u + 1][m] : w == 2 ? f[4 * u + 2][m] : f[4 * u + 3][m];
        mu += fv * tv[m];
        cap += fv * fv;
      }
      mu = wave_reduce_sum<double>(mu);
      cap = wave_reduce_sum<double>(cap);
      if (lane == 0 && qq < nq) {
        __hip_atomic_store(ws + 64 + q0 + qq, mu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(ws + 128 + q0 + qq, cap, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  // ticket: this workgroup's contributions are out (vmcnt: the atomics and stores above have been acknowledged)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const int total = (int)(gridDim.x * gridDim.y);
    const int tk = __hip_atomic_fetch_add(reinterpret_cast<int*>(ws + 192), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = tk == total - 1;
  }
  __syncthreads();
  if (!s_last) return;
  // ---- the last workgroup: variances, metrics, reset
  const double s2 = (double)s2p[0];
  double sq = 0.0, nl = 0.0, amax = 0.0;
  if (threadIdx.x < n) {
    const int j = threadIdx.x;
    const double dg = __hip_atomic_load(ws + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double mu = __hip_atomic_load(ws + 64 + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double cap = __hip_atomic_load(ws + 128 + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    double tl = prior[j] * kscale - cap;
    tl = tl > 0 ? tl : 0.0;
    // (rounded to the data dtype where the op-by-op path rounds: the latent variance, the mean, their use in the metrics)
    real var = (real)((dg + tl) * s2);
    var = var > (real)0 ? var : (real)0;
    const real mur = (real)mu;
    if (mean_out) mean_out[j] = mur;
    if (var_out) var_out[j] = var;
    const real df = mur - y[j];
    const real v = (real)((double)var + s2);
    const real s = df * df;
    sq = (double)s;
    nl = (double)((real)0.5 * (s / v + (real)log((double)v) + (real)1.8378770664093453));
    amax = fabs(mu);
    __hip_atomic_store(ws + j, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (threadIdx.x == 0) __hip_atomic_store(reinterpret_cast<int*>(ws + 192), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  sq = block_reduce_sum(sq, s_red);
  nl = block_reduce_sum(nl, s_red);
  // max |mean|: n <= 64 values sit in wave 0
  double mx = amax;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const double other = __shfl_xor(mx, o);
    mx = other > mx ? other : mx;
  }
  if (threadIdx.x == 0) {
    out[0] = sqrt(sq / (double)n);
    out[1] = nl / (double)n;
    out[2] = err ? (double)err[0] : 0.0;
    out[3] = mx;
  }
}

// The same for larger batches (64 < n <= 1024, one evaluate() chunk): Y = chol^-1 F^T comes from the MFMA GEMM ([r, n]); a workgroup takes
// 64 queries (lanes: coalesced across the columns of Y; the r rows dealt to its 4 waves) for |Y[:, j]|^2, |F_j|^2 and F_j . t, forms the
// moments and its share of the two metric sums (fp64 atomics into d_ws[0..1]); the last workgroup out writes d_out and zeroes d_ws.
template <typename real>
__global__ __launch_bounds__(256) void k_spectral_eval_y(int n, int r, const double* __restrict__ Y, const double* __restrict__ F, const double* __restrict__ prior,
                                                         const double* __restrict__ t, double kscale, const real* __restrict__ s2p, const real* __restrict__ y,
                                                         const int32_t* __restrict__ err, double* ws, double* __restrict__ out, real* __restrict__ mean_out,
                                                         real* __restrict__ var_out) {
  __shared__ double s_d[4][64], s_c[4][64], s_m[4][64];
  __shared__ double s_red[16];
  __shared__ int s_last;
  const int c = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + c;
  double dsum = 0, cap = 0, mu = 0;
  if (j < n) {
    for (int i = part; i < r; i += 4) {
      const double yv = Y[(int64_t)i * n + j];
      dsum += yv * yv;
    }
    const double* __restrict__ f = F + (int64_t)j * r;
    for (int k = part; k < r; k += 4) {
      const double fv = f[k];
      cap += fv * fv;
      mu += fv * t[k];
    }
  }
  s_d[part][c] = dsum;
  s_c[part][c] = cap;
  s_m[part][c] = mu;
  __syncthreads();
  double sq = 0, nl = 0, amax = 0;
  if (part == 0 && j < n) {
    const double s2 = (double)s2p[0];
    const double dg = s_d[0][c] + s_d[1][c] + s_d[2][c] + s_d[3][c];
    const double m = s_m[0][c] + s_m[1][c] + s_m[2][c] + s_m[3][c];
    double tl = prior[j] * kscale - (s_c[0][c] + s_c[1][c] + s_c[2][c] + s_c[3][c]);
    tl = tl > 0 ? tl : 0.0;
    real var = (real)((dg + tl) * s2);
    var = var > (real)0 ? var : (real)0;
    const real mur = (real)m;
    if (mean_out) mean_out[j] = mur;
    if (var_out) var_out[j] = var;
    const real df = mur - y[j];
    const real v = (real)((double)var + s2);
    const real s = df * df;
    sq = (double)s;
    nl = (double)((real)0.5 * (s / v + (real)log((double)v) + (real)1.8378770664093453));
    amax = fabs(m);
  }
  sq = block_reduce_sum(sq, s_red);
  nl = block_reduce_sum(nl, s_red);
  double mx = amax;                                  // (the 64 values sit in wave 0)
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const double other = __shfl_xor(mx, o);
    mx = other > mx ? other : mx;
  }
  if (threadIdx.x == 0) {
    unsafeAtomicAdd(ws + 0, sq);
    unsafeAtomicAdd(ws + 1, nl);
    // max |mean| over the workgroups: non-negative doubles order like their bit patterns
    atomicMax(reinterpret_cast<unsigned long long*>(ws + 2), (unsigned long long)__double_as_longlong(mx));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int tk = __hip_atomic_fetch_add(reinterpret_cast<int*>(ws + 192), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = tk == (int)gridDim.x - 1;
  }
  __syncthreads();
  if (!s_last || threadIdx.x != 0) return;
  const double tsq = __hip_atomic_load(ws + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const double tnl = __hip_atomic_load(ws + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const double tmx = __hip_atomic_load(ws + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  out[0] = sqrt(tsq / (double)n);
  out[1] = tnl / (double)n;
  out[2] = err ? (double)err[0] : 0.0;
  out[3] = tmx;
  __hip_atomic_store(ws + 0, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(ws + 1, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(ws + 2, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(reinterpret_cast<int*>(ws + 192), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <typename real>
static int spectral_evaluate_impl(int32_t n, int32_t r, const double* d_F, const double* d_prior, const double* d_Linv, int32_t ldl, const double* d_t,
                                  double kscale, const real* d_s2, const real* d_y, const int32_t* d_err, double* d_ws, double* d_out, real* d_mean,
                                  real* d_var, void* stream) {
  if (n < 1 || n > 64 || r < 1 || r > 1024 || ldl < r || !d_F || !d_prior || !d_Linv || !d_t || !d_s2 || !d_y || !d_ws || !d_out) return WISKI_E_BADARG;
  if (r <= 512)
    hipLaunchKernelGGL((k_spectral_eval<real, 8>), dim3((unsigned)((r + 7) / 8), (unsigned)((n + 7) / 8)), dim3(256), 0, (hipStream_t)stream, (int)n, (int)r, d_F,
                       d_prior, d_Linv, (int)ldl, d_t, kscale, d_s2, d_y, d_err, d_ws, d_out, d_mean, d_var);
  else
    hipLaunchKernelGGL((k_spectral_eval<real, 16>), dim3((unsigned)((r + 7) / 8), (unsigned)((n + 7) / 8)), dim3(256), 0, (hipStream_t)stream, (int)n, (int)r, d_F,
                       d_prior, d_Linv, (int)ldl, d_t, kscale, d_s2, d_y, d_err, d_ws, d_out, d_mean, d_var);
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}
extern "C" int wiski_spectral_evaluate_f32(int32_t n, int32_t r, const double* d_F, const double* d_prior, const double* d_Linv, int32_t ldl, const double* d_t,
                                           double kscale, const float* d_s2, const float* d_y, const int32_t* d_err, double* d_ws, double* d_out,
                                           float* d_mean, float* d_var, void* stream) {
  return spectral_evaluate_impl<float>(n, r, d_F, d_prior, d_Linv, ldl, d_t, kscale, d_s2, d_y, d_err, d_ws, d_out, d_mean, d_var, stream);
}
extern "C" int wiski_spectral_evaluate_f64(int32_t n, int32_t r, const double* d_F, const double* d_prior, const double* d_Linv, int32_t ldl, const double* d_t,
                                           double kscale, const double* d_s2, const double* d_y, const int32_t* d_err, double* d_ws, double* d_out,
                                           double* d_mean, double* d_var, void* stream) {
  return spectral_evaluate_impl<double>(n, r, d_F, d_prior, d_Linv, ldl, d_t, kscale, d_s2, d_y, d_err, d_ws, d_out, d_mean, d_var, stream);
}

template <typename real>
static int spectral_evaluate_y_impl(int32_t n, int32_t r, const double* d_Y, const double* d_F, const double* d_prior, const double* d_t, double kscale,
                                    const real* d_s2, const real* d_y, const int32_t* d_err, double* d_ws, double* d_out, real* d_mean, real* d_var,
                                    void* stream) {
  if (n < 1 || r < 1 || !d_Y || !d_F || !d_prior || !d_t || !d_s2 || !d_y || !d_ws || !d_out) return WISKI_E_BADARG;
  hipLaunchKernelGGL((k_spectral_eval_y<real>), dim3((unsigned)((n + 63) / 64)), dim3(256), 0, (hipStream_t)stream, (int)n, (int)r, d_Y, d_F, d_prior, d_t,
                     kscale, d_s2, d_y, d_err, d_ws, d_out, d_mean, d_var);
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}
extern "C" int wiski_spectral_evaluate_y_f32(int32_t n, int32_t r, const double* d_Y, const double* d_F, const double* d_prior, const double* d_t, double kscale,
                                             const float* d_s2, const float* d_y, const int32_t* d_err, double* d_ws, double* d_out, float* d_mean,
                                             float* d_var, void* stream) {
  return spectral_evaluate_y_impl<float>(n, r, d_Y, d_F, d_prior, d_t, kscale, d_s2, d_y, d_err, d_ws, d_out, d_mean, d_var, stream);
}
extern "C" int wiski_spectral_evaluate_y_f64(int32_t n, int32_t r, const double* d_Y, const double* d_F, const double* d_prior, const double* d_t, double kscale,
                                             const double* d_s2, const double* d_y, const int32_t* d_err, double* d_ws, double* d_out, double* d_mean,
                                             double* d_var, void* stream) {
  return spectral_evaluate_y_impl<double>(n, r, d_Y, d_F, d_prior, d_t, kscale, d_s2, d_y, d_err, d_ws, d_out, d_mean, d_var, stream);
}

// ------------------------------------------------ the factor's refresh after a hyper-parameter step, ONE host call ---
// wiski_basis_eig_update_adaptive -> wiski_basis_change -> G = T^T G_ref T (two GEMMs) -> wiski_woodbury_c -> wiski_potrf_inverse ->
// wiski_factor_tail, queued back to back (the same launches as the seven calls; the host-language wrapper made each of them cost
// ~6-8 us of interpreter time in a step that is bound by exactly that).  Every product lives in ONE packed fp64 buffer whose layout both
// sides compute with wiski_factor_refresh_layout: offsets (in doubles, multiples of 32) of
//   0 Vout [nV]  1 ev [d kw]  2 resid [d]  3 Tq [d 32 32]  4 TS [r_ref r]  5 lam_kuu [r]  6 GT [r_ref r]  7 G [r r]  8 C -> chol [r r]
//   9 sqG [r r]  10 lam [r]  11 sq [r]  12 Linv [r r]  13 tail [6 r + 2]  and the total in off[14].
extern "C" int wiski_factor_refresh_layout(int32_t d, int64_t nV, int32_t kw, int32_t r_ref, int32_t r, int64_t* off) {
  if (d < 1 || d > WISKI_MAX_DIM || nV < 1 || kw < 1 || r_ref < 1 || r < 1 || !off) return WISKI_E_BADARG;
  const int64_t sz[14] = {nV, (int64_t)d * kw, d, (int64_t)d * 32 * 32, (int64_t)r_ref * r, r, (int64_t)r_ref * r, (int64_t)r * r, (int64_t)r * r,
                          (int64_t)r * r, r, r, (int64_t)r * r, 6 * (int64_t)r + 2};
  int64_t o = 0;
  for (int i = 0; i < 14; ++i) {
    off[i] = o;
    o += (sz[i] + 31) / 32 * 32;
  }
  off[14] = o;
  return WISKI_OK;
}
extern "C" int wiski_factor_refresh(int32_t d, const int32_t* d_g, int64_t nV, const double* d_tcol, const double* d_Vin, int32_t kw, int32_t kuse,
                                    const double* d_Vref, int32_t kref, int32_t niter, double resid_ok, const int32_t* d_Sref, const int32_t* d_S,
                                    int32_t r_ref, int32_t r, double* d_work, double* d_verdict, const double* d_Gref, const double* d_href,
                                    double kscale, int32_t* d_info, double* d_out, void* verdict_event, void* stream) {
  int64_t o[15];
  if (int rc = wiski_factor_refresh_layout(d, nV, kw, r_ref, r, o)) return rc;
  if (!d_g || !d_tcol || !d_Vin || !d_Vref || !d_Sref || !d_S || !d_work || !d_verdict || !d_Gref || !d_href || !d_info || !d_out) return WISKI_E_BADARG;
  double *Vout = d_out + o[0], *ev = d_out + o[1], *resid = d_out + o[2], *Tq = d_out + o[3], *TS = d_out + o[4], *lamk = d_out + o[5], *GT = d_out + o[6],
         *G = d_out + o[7], *C = d_out + o[8], *sqG = d_out + o[9], *lam = d_out + o[10], *sq = d_out + o[11], *Linv = d_out + o[12], *tail = d_out + o[13];
  if (int rc = eig_update_launch(d, d_g, d_tcol, d_Vin, kw, kuse, Vout, ev, resid, d_Vref, kref, Tq, niter, resid_ok, stream)) return rc;
  if (int rc = wiski_basis_change(d, d_g, kref, kw, r_ref, r, Tq, d_Sref, d_S, ev, d_tcol, resid, TS, lamk, d_work, d_verdict, stream)) return rc;
  // (the verdict sits in the caller's pinned memory from here on: an event of the caller's marks the spot, so that reading it does not
  //  wait for the factorisation behind it)
  if (verdict_event && hipEventRecord((hipEvent_t)verdict_event, (hipStream_t)stream) != hipSuccess) return WISKI_E_LAUNCH;
  if (int rc = wiski_gemm_f64(1, 0, r_ref, r, r_ref, 1.0, d_Gref, r_ref, TS, r, 0.0, GT, r, stream)) return rc;      // G_ref TS (G_ref symmetric)
  if (int rc = wiski_gemm_f64(1, 0, r, r, r_ref, 1.0, TS, r, GT, r, 0.0, G, r, stream)) return rc;                    // T^T G_ref T
  if (int rc = wiski_woodbury_c(r, G, lamk, kscale, C, lam, sq, sqG, stream)) return rc;
  if (int rc = wiski_potrf_inverse_f64(r, C, r, Linv, r, d_info, stream)) return rc;
  return wiski_factor_tail(r_ref, r, TS, d_href, sq, Linv, C, tail, stream);
}
