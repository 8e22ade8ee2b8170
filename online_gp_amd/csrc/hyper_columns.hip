// Host-bound glue of the hyper-parameter step, as single launches.
//
// One step of the reference's online loop (experiments/regression.py:48-54: evaluate -> Adam step on the MLL -> condition) spends
// most of its wall time in ~100 tiny framework launches of a few elements each: the kernel's Toeplitz columns as a function of the
// lengthscales (gpytorch evaluates RBF / Matern on the grid through ~10 broadcasting ops, and autograd replays ~12 more), and the
// Gaussian test metrics of the batch.  A launch costs ~5 us of GPU time and ~10 us of host time whatever it computes, so these are
// done in one kernel each:
//   k_stationary_columns        tcol[q][l] = S * k(l h_q / ell_q)          (fp64, all dims; k = RBF | Matern-1/2, -3/2, -5/2)
//   k_stationary_columns_grad   d/d ell_q, d/d S of <g, tcol>             (closed forms of the same profiles)
//   k_gaussian_metrics          [ sqrt(mean (mu - y)^2),  mean 1/2 ((mu - y)^2 / v + log v + log 2 pi) ]   (OSR:88-111)
#include "wiski_common.h"

// k(r) and dk/dr for r = lag / lengthscale >= 0
__device__ __forceinline__ void stationary_profile(int kind, double r, double* k, double* dk) {
  if (kind == 0) {                       // RBF: exp(-r^2 / 2)
    const double e = exp(-0.5 * r * r);
    *k = e;
    *dk = -r * e;
  } else if (kind == 1) {                // Matern 1/2: exp(-r)
    const double e = exp(-r);
    *k = e;
    *dk = -e;
  } else if (kind == 2) {                // Matern 3/2: (1 + s) exp(-s), s = sqrt(3) r
    const double c = 1.7320508075688772, s = c * r, e = exp(-s);
    *k = (1.0 + s) * e;
    *dk = -c * s * e;
  } else {                               // Matern 5/2: (1 + s + s^2 / 3) exp(-s), s = sqrt(5) r
    const double c = 2.23606797749979, s = c * r, e = exp(-s);
    *k = (1.0 + s + s * s / 3.0) * e;
    *dk = -c * (s / 3.0) * (1.0 + s) * e;
  }
}

struct ColumnsGrid {
  int d;
  int g[WISKI_MAX_DIM];
  double h[WISKI_MAX_DIM];
};

template <typename real>
__global__ __launch_bounds__(256) void k_stationary_columns(ColumnsGrid G, int kind, const real* __restrict__ ell, int nell, const real* __restrict__ scale,
                                                            double* __restrict__ out) {
  const double S = scale ? (double)scale[0] : 1.0;
  int off = 0;
  for (int q = 0; q < G.d; ++q) {
    const double el = (double)ell[nell > 1 ? q : 0];
    for (int l = threadIdx.x; l < G.g[q]; l += 256) {
      double k, dk;
      stationary_profile(kind, (G.h[q] * (double)l) / el, &k, &dk);
      out[off + l] = S * k;
    }
    off += G.g[q];
  }
}

template <typename real>
__global__ __launch_bounds__(256) void k_stationary_columns_grad(ColumnsGrid G, int kind, const real* __restrict__ ell, int nell, const real* __restrict__ scale,
                                                                 const double* __restrict__ gout, real* __restrict__ g_ell, real* __restrict__ g_scale) {
  __shared__ double s_red[16];
  const double S = scale ? (double)scale[0] : 1.0;
  double gS = 0, gE[WISKI_MAX_DIM];
  int off = 0;
  for (int q = 0; q < G.d; ++q) {
    const double el = (double)ell[nell > 1 ? q : 0];
    double ge = 0;
    for (int l = threadIdx.x; l < G.g[q]; l += 256) {
      const double r = (G.h[q] * (double)l) / el;
      double k, dk;
      stationary_profile(kind, r, &k, &dk);
      const double go = gout[off + l];
      gS += go * k;
      ge += go * S * dk * (-r / el);
    }
    gE[q] = ge;
    off += G.g[q];
  }
  double iso = 0;
  for (int q = 0; q < G.d; ++q) {
    const double v = block_reduce_sum(gE[q], s_red);
    __syncthreads();
    if (nell > 1) { if (threadIdx.x == 0) g_ell[q] = (real)v; }
    else iso += v;
  }
  if (nell <= 1 && threadIdx.x == 0) g_ell[0] = (real)iso;
  gS = block_reduce_sum(gS, s_red);
  if (threadIdx.x == 0 && g_scale) g_scale[0] = (real)gS;
}

// out[0] = sqrt(mean (mu - y)^2), out[1] = mean 1/2 ((mu - y)^2 / v + log v + log 2 pi), v = var + add_var[0] (add_var may be null).
// One block (batches of the reference loop are <= 1024 points).
template <typename real>
__global__ __launch_bounds__(256) void k_gaussian_metrics(int64_t n, const real* __restrict__ mu, const real* __restrict__ var, const real* __restrict__ y,
                                                          const real* __restrict__ add_var, real* __restrict__ out) {
  __shared__ double s_red[16];
  const double av = add_var ? (double)add_var[0] : 0.0;
  double sq = 0, nl = 0;
  for (int64_t i = threadIdx.x; i < n; i += 256) {
    // (in the data dtype, as the reference computes it: the difference and the variance sum are rounded to `real` first)
    const real df = mu[i] - y[i];
    const real v = (real)((double)var[i] + av);
    const real s = df * df;
    sq += (double)s;
    nl += (double)((real)0.5 * (s / v + (real)log((double)v) + (real)1.8378770664093453));
  }
  sq = block_reduce_sum(sq, s_red);
  __syncthreads();
  nl = block_reduce_sum(nl, s_red);
  if (threadIdx.x == 0) {
    out[0] = (real)sqrt(sq / (double)n);
    out[1] = (real)(nl / (double)n);
  }
}

// Scalar tail of the Woodbury MLL (BWM:34-47) for one output, value and gradient, so that the ~15 zero-dimensional framework ops
// (and their ~20 autograd nodes) around the factor's two numbers become one launch each way:
//   val = -1/2 ( (c - bMb) / s2 + logdet + ld + n log(2 pi) + n log s2 ),   coef = { 1/2 / s2, -1/2, c - bMb }  (d val / d (bMb, logdet), and c - bMb)
//   g_s2 = g * ( 1/2 (c - bMb) / s2^2 - 1/2 n / s2 ) - g_kap / s2^2     (g_kap: gradient w.r.t. kappa = 1 / s2 from the factor)
template <typename real>
__global__ void k_mll_value(const double* __restrict__ bMb, const double* __restrict__ logdet, const real* __restrict__ s2p, const double* __restrict__ c,
                            const double* __restrict__ ld, double n_val, const double* __restrict__ n_dev, double* __restrict__ val,
                            double* __restrict__ coef) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double s2 = (double)s2p[0];
  const double n = n_dev ? n_dev[0] : n_val;       // (a device scalar when the call is part of a captured graph: the count moves on)
  val[0] = -0.5 * ((c[0] - bMb[0]) / s2 + (logdet ? logdet[0] : 0.0) + ld[0] + n * (1.8378770664093453 + log(s2)));
  coef[0] = 0.5 / s2;
  coef[1] = -0.5;
  coef[2] = c[0] - bMb[0];             // kept for the backward pass (c lives in the model's statistics, which move on)
}

template <typename real>
__global__ void k_mll_s2_grad(const double* __restrict__ g, const double* __restrict__ coef, const real* __restrict__ s2p, double n_val,
                              const double* __restrict__ n_dev, const double* __restrict__ g_kap, real* __restrict__ g_s2) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double s2 = (double)s2p[0];
  const double n = n_dev ? n_dev[0] : n_val;
  g_s2[0] = (real)(g[0] * (0.5 * coef[2] / (s2 * s2) - 0.5 * n / s2) - (g_kap ? g_kap[0] / (s2 * s2) : 0.0));
}

static int columns_grid(const wiski_grid* grid, ColumnsGrid* G) {
  if (!grid || grid->d < 1 || grid->d > WISKI_MAX_DIM) return WISKI_E_BADARG;
  G->d = grid->d;
  for (int q = 0; q < grid->d; ++q) {
    if (grid->g[q] < 1 || !(grid->h[q] > 0)) return WISKI_E_BADARG;
    G->g[q] = grid->g[q];
    G->h[q] = grid->h[q];
  }
  return WISKI_OK;
}

template <typename real>
static int columns_impl(const wiski_grid* grid, int kind, const real* d_ell, int nell, const real* d_scale, double* d_out, void* stream) {
  ColumnsGrid G;
  const int rc = columns_grid(grid, &G);
  if (rc) return rc;
  if (kind < 0 || kind > 3 || !d_ell || !d_out || (nell != 1 && nell != grid->d)) return WISKI_E_BADARG;
  hipLaunchKernelGGL((k_stationary_columns<real>), dim3(1), dim3(256), 0, (hipStream_t)stream, G, kind, d_ell, nell, d_scale, d_out);
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}

template <typename real>
static int columns_grad_impl(const wiski_grid* grid, int kind, const real* d_ell, int nell, const real* d_scale, const double* d_gout, real* d_gell,
                             real* d_gscale, void* stream) {
  ColumnsGrid G;
  const int rc = columns_grid(grid, &G);
  if (rc) return rc;
  if (kind < 0 || kind > 3 || !d_ell || !d_gout || !d_gell || (nell != 1 && nell != grid->d)) return WISKI_E_BADARG;
  hipLaunchKernelGGL((k_stationary_columns_grad<real>), dim3(1), dim3(256), 0, (hipStream_t)stream, G, kind, d_ell, nell, d_scale, d_gout, d_gell, d_gscale);
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}

template <typename real>
static int metrics_impl(int64_t n, const real* d_mu, const real* d_var, const real* d_y, const real* d_add, real* d_out, void* stream) {
  if (n < 1 || !d_mu || !d_var || !d_y || !d_out) return WISKI_E_BADARG;
  hipLaunchKernelGGL((k_gaussian_metrics<real>), dim3(1), dim3(256), 0, (hipStream_t)stream, n, d_mu, d_var, d_y, d_add, d_out);
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}

extern "C" {
int wiski_stationary_columns_f32(const wiski_grid* grid, int32_t kind, const float* d_ell, int32_t nell, const float* d_scale, double* d_out, void* stream) {
  return columns_impl<float>(grid, kind, d_ell, nell, d_scale, d_out, stream);
}
int wiski_stationary_columns_f64(const wiski_grid* grid, int32_t kind, const double* d_ell, int32_t nell, const double* d_scale, double* d_out, void* stream) {
  return columns_impl<double>(grid, kind, d_ell, nell, d_scale, d_out, stream);
}
int wiski_stationary_columns_grad_f32(const wiski_grid* grid, int32_t kind, const float* d_ell, int32_t nell, const float* d_scale, const double* d_gout,
                                      float* d_gell, float* d_gscale, void* stream) {
  return columns_grad_impl<float>(grid, kind, d_ell, nell, d_scale, d_gout, d_gell, d_gscale, stream);
}
int wiski_stationary_columns_grad_f64(const wiski_grid* grid, int32_t kind, const double* d_ell, int32_t nell, const double* d_scale, const double* d_gout,
                                      double* d_gell, double* d_gscale, void* stream) {
  return columns_grad_impl<double>(grid, kind, d_ell, nell, d_scale, d_gout, d_gell, d_gscale, stream);
}
int wiski_mll_value_f32(const double* d_bMb, const double* d_logdet, const float* d_s2, const double* d_c, const double* d_ld, double n, const double* d_n,
                        double* d_val, double* d_coef, void* stream) {
  if (!d_bMb || !d_s2 || !d_c || !d_ld || !d_val || !d_coef) return WISKI_E_BADARG;
  hipLaunchKernelGGL((k_mll_value<float>), dim3(1), dim3(64), 0, (hipStream_t)stream, d_bMb, d_logdet, d_s2, d_c, d_ld, n, d_n, d_val, d_coef);
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}
int wiski_mll_value_f64(const double* d_bMb, const double* d_logdet, const double* d_s2, const double* d_c, const double* d_ld, double n, const double* d_n,
                        double* d_val, double* d_coef, void* stream) {
  if (!d_bMb || !d_s2 || !d_c || !d_ld || !d_val || !d_coef) return WISKI_E_BADARG;
  hipLaunchKernelGGL((k_mll_value<double>), dim3(1), dim3(64), 0, (hipStream_t)stream, d_bMb, d_logdet, d_s2, d_c, d_ld, n, d_n, d_val, d_coef);
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}
int wiski_mll_s2_grad_f32(const double* d_g, const double* d_coef, const float* d_s2, double n, const double* d_n, const double* d_gkap, float* d_gs2,
                          void* stream) {
  if (!d_g || !d_coef || !d_s2 || !d_gs2) return WISKI_E_BADARG;
  hipLaunchKernelGGL((k_mll_s2_grad<float>), dim3(1), dim3(64), 0, (hipStream_t)stream, d_g, d_coef, d_s2, n, d_n, d_gkap, d_gs2);
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}
int wiski_mll_s2_grad_f64(const double* d_g, const double* d_coef, const double* d_s2, double n, const double* d_n, const double* d_gkap, double* d_gs2,
                          void* stream) {
  if (!d_g || !d_coef || !d_s2 || !d_gs2) return WISKI_E_BADARG;
  hipLaunchKernelGGL((k_mll_s2_grad<double>), dim3(1), dim3(64), 0, (hipStream_t)stream, d_g, d_coef, d_s2, n, d_n, d_gkap, d_gs2);
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}
int wiski_gaussian_metrics_f32(int64_t n, const float* d_mu, const float* d_var, const float* d_y, const float* d_add_var, float* d_out, void* stream) {
  return metrics_impl<float>(n, d_mu, d_var, d_y, d_add_var, d_out, stream);
}
int wiski_gaussian_metrics_f64(int64_t n, const double* d_mu, const double* d_var, const double* d_y, const double* d_add_var, double* d_out, void* stream) {
  return metrics_impl<double>(n, d_mu, d_var, d_y, d_add_var, d_out, stream);
}
}
