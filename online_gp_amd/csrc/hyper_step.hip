// The Adam step of the streaming loop on the Woodbury MLL, for the standard parameterisation, without the framework's autograd.
//
// experiments/regression.py:48-54 takes ONE Adam step on the marginal log-likelihood between two batches (OSR:135-147).  With the
// spectral factor serving the MLL the arithmetic of that step is a handful of library launches (the factor's reduced-basis
// gradient) -- but recorded through autograd it is a graph of 40 nodes, 28 of them framework element-wise ops on 1-5 numbers
// (softplus of the raw parameters, reshapes, the scalar tail, autograd's accumulation nodes, the multi-tensor Adam), each a
// ~4.6 us dispatch: 183 us for ~60 us of work.  When the model is (Scale of)* RBF | Matern with a homoskedastic second noise, no
// registered priors and a plain Adam -- the reference's own configuration -- those 28 nodes are three kernels here:
//   k_hyper_columns   raw parameters -> lengthscales, product of output scales, sigma2 (the constraints' transforms:
//                     lower + softplus, or lower + (upper - lower) sigmoid) and, on request, the Toeplitz columns (fp64 + data dtype)
//   k_hyper_mid       MLL value and its scalar coefficients (k_mll_value's arithmetic), the loss -val / n and the incoming
//                     gradients of b^T M b and logdet that the factor's backward pass takes from the device
//   k_hyper_adam      chain rule back to the raw parameters (d sigma2, d lengthscale, d outputscale factors) and the Adam update
//                     of every parameter and its moments (torch.optim.Adam's arithmetic in the parameter dtype, step counters fp32)
// The caller (models/_graphed_step.py) records them with the factor's backward launches into one HIP graph.
#include "wiski_common.h"

struct ColumnsGridS {
  int d;
  int g[WISKI_MAX_DIM];
  double h[WISKI_MAX_DIM];
};

__device__ __forceinline__ double hs_profile(int kind, double r) {
  if (kind == 0) return exp(-0.5 * r * r);
  if (kind == 1) return exp(-r);
  if (kind == 2) { const double s = 1.7320508075688772 * r; return (1.0 + s) * exp(-s); }
  const double s = 2.23606797749979 * r;
  return (1.0 + s + s * s / 3.0) * exp(-s);
}

// constraint transform and its backward pass, in the parameter dtype and in the framework's own operation order (softplus:
// log1p(exp(x)) above / below the threshold of 20, backward g z / (z + 1) with z = exp(x); sigmoid: backward ((g w) (1 - y)) y) -- a
// trajectory recorded here and one recorded through autograd then agree to the last bit of an fp32 parameter, not just to 1e-7
__device__ __forceinline__ float hs_exp(float x) { return expf(x); }
__device__ __forceinline__ double hs_exp(double x) { return exp(x); }
__device__ __forceinline__ float hs_log1p(float x) { return log1pf(x); }
__device__ __forceinline__ double hs_log1p(double x) { return log1p(x); }
template <typename real>
__device__ __forceinline__ real hs_transform(const wiski_hyper_param& p, real raw) {
  if (p.kind == 0) return (raw > (real)20 ? raw : hs_log1p(hs_exp(raw))) + (real)p.lower;
  const real sg = (real)1 / ((real)1 + hs_exp(-raw));
  return (real)p.lower + (real)(p.upper - p.lower) * sg;
}
template <typename real>
__device__ __forceinline__ real hs_backward(const wiski_hyper_param& p, real raw, real g) {
  if (p.kind == 0) {
    if (raw > (real)20) return g;
    const real z = hs_exp(raw);
    return g * z / (z + (real)1);
  }
  const real y = (real)1 / ((real)1 + hs_exp(-raw));
  return ((g * (real)(p.upper - p.lower)) * ((real)1 - y)) * y;
}

template <typename real>
__global__ __launch_bounds__(256) void k_hyper_columns(wiski_hyper_plan plan, ColumnsGridS G, int kind, real* __restrict__ ell, real* __restrict__ scale,
                                                       real* __restrict__ s2, double* __restrict__ s2_f64, double* __restrict__ tcol64,
                                                       real* __restrict__ tcol) {
  __shared__ double s_ell[WISKI_MAX_DIM];
  __shared__ double s_scale;
  if (threadIdx.x == 0) {
    real sc = (real)1;
    bool has_scale = false;
    int nell = 0;
    for (int i = 0; i < plan.count; ++i) {
      const wiski_hyper_param& p = plan.p[i];
      const real* raw = (const real*)p.raw;
      if (p.role == 0) {
        nell = p.numel;
        for (int e = 0; e < p.numel; ++e) {
          const real v = hs_transform<real>(p, raw[e]);
          ell[e] = v;
          s_ell[e] = (double)v;
        }
      } else if (p.role == 1) {
        const real v = hs_transform<real>(p, raw[0]);
        sc = has_scale ? sc * v : v;
        has_scale = true;
      } else {
        const real v = hs_transform<real>(p, raw[0]);
        s2[0] = v;
        if (s2_f64) s2_f64[0] = (double)v;
      }
    }
    if (scale) scale[0] = sc;
    s_scale = has_scale ? (double)sc : 1.0;
    if (nell == 1)
      for (int q = 1; q < G.d; ++q) s_ell[q] = s_ell[0];
  }
  if (!tcol64) return;
  __syncthreads();
  int off = 0;
  for (int q = 0; q < G.d; ++q) {
    for (int l = threadIdx.x; l < G.g[q]; l += 256) {
      const double v = s_scale * hs_profile(kind, (G.h[q] * (double)l) / s_ell[q]);
      tcol64[off + l] = v;
      if (tcol) tcol[off + l] = (real)v;
    }
    off += G.g[q];
  }
}

// out = { val, coef0, coef1, coef2, g = -1 / n, g coef0, g coef1, loss = -val / n, 1 / s2 }      (k_mll_value's arithmetic, BWM:34-47)
template <typename real>
__global__ void k_hyper_mid(const double* __restrict__ bMb, const double* __restrict__ logdet, const real* __restrict__ s2p, const double* __restrict__ c,
                            const double* __restrict__ ld, const double* __restrict__ n_dev, double* __restrict__ out, double* __restrict__ loss_out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double s2 = (double)s2p[0], n = n_dev[0];
  const double val = -0.5 * ((c[0] - bMb[0]) / s2 + (logdet ? logdet[0] : 0.0) + ld[0] + n * (1.8378770664093453 + log(s2)));
  const double g = -1.0 / n;
  out[0] = val;
  out[1] = 0.5 / s2;
  out[2] = -0.5;
  out[3] = c[0] - bMb[0];
  out[4] = g;
  out[5] = g * (0.5 / s2);
  out[6] = g * -0.5;
  out[7] = -val / n;
  out[8] = 1.0 / s2;
  if (loss_out) loss_out[0] = -val / n;
}

template <typename real>
__global__ void k_hyper_adam(wiski_hyper_plan plan, const real* __restrict__ scale, const real* __restrict__ s2p, const real* __restrict__ g_ell,
                             const real* __restrict__ g_scale, const double* __restrict__ mid, const double* __restrict__ g_kap,
                             const double* __restrict__ n_dev, double lr, double beta1, double beta2, double eps) {
  // one thread per parameter element (a handful)
  int idx = threadIdx.x, pi = -1, e = 0;
  for (int i = 0; i < plan.count; ++i) {
    if (idx < plan.p[i].numel) { pi = i; e = idx; break; }
    idx -= plan.p[i].numel;
  }
  if (pi >= 0) {
    const wiski_hyper_param& p = plan.p[pi];
    real* raw = (real*)p.raw;
    const real val = hs_transform<real>(p, raw[e]);
    real gv;                                                    // d loss / d (constrained value)
    if (p.role == 0) gv = g_ell[e];
    else if (p.role == 1) gv = g_scale[0] * (scale[0] / val);   // the product of the factors, divided by this one
    else {
      const double s2 = (double)s2p[0], n = n_dev[0];
      gv = (real)(mid[4] * (0.5 * mid[3] / (s2 * s2) - 0.5 * n / s2) - g_kap[0] / (s2 * s2));
    }
    const real grad = hs_backward<real>(p, raw[e], gv);
    // torch.optim.Adam (no weight decay, no amsgrad), arithmetic in the parameter dtype, step counter fp32
    real* m = (real*)p.exp_avg;
    real* v = (real*)p.exp_avg_sq;
    const float* stp = (const float*)p.step;
    const float step = stp[e < p.step_numel ? e : 0] + 1.f;
    // (the constants as the framework's fused kernel forms them: 1 - beta and beta^step in double, then rounded to the parameter dtype --
    //  1 - (float)0.999 is off by 1e-4 relative, which is what the second moment would then be off by)
    const real b2 = (real)beta2, omb1 = (real)(1.0 - beta1), omb2 = (real)(1.0 - beta2);
    const real mn = m[e] + omb1 * (grad - m[e]);                // lerp
    const real vn = b2 * v[e] + omb2 * grad * grad;
    m[e] = mn;
    v[e] = vn;
    const real bc1 = (real)(1.0 - pow(beta1, (double)step));
    const real bc2 = (real)(1.0 - pow(beta2, (double)step));
    const real step_size = (real)lr / bc1;
    const real denom = (real)sqrt((double)vn) / (real)sqrt((double)bc2) + (real)eps;
    raw[e] = raw[e] - step_size * (mn / denom);
  }
  // (the step counters are bumped once every element has read them)
  __syncthreads();
  if (threadIdx.x == 0)
    for (int i = 0; i < plan.count; ++i) {
      float* stp = (float*)plan.p[i].step;
      for (int e = 0; e < plan.p[i].step_numel; ++e) stp[e] += 1.f;
    }
}

static int plan_ok(const wiski_hyper_plan* plan, int d) {
  if (!plan || plan->count < 1 || plan->count > WISKI_HYPER_MAX_PARAMS) return 0;
  int total = 0, nell = 0, nnoise = 0;
  for (int i = 0; i < plan->count; ++i) {
    const wiski_hyper_param& p = plan->p[i];
    if (!p.raw || p.numel < 1 || p.role < 0 || p.role > 2 || p.kind < 0 || p.kind > 1) return 0;
    if (p.role == 0) { if (nell || (p.numel != 1 && p.numel != d)) return 0; nell = p.numel; }
    else if (p.numel != 1) return 0;
    if (p.role == 2) ++nnoise;
    total += p.numel;
  }
  return nell && nnoise == 1 && total <= 64;
}

template <typename real>
static int hyper_columns_impl(const wiski_hyper_plan* plan, const wiski_grid* grid, int kind, real* d_ell, real* d_scale, real* d_s2, double* d_s2_f64,
                              double* d_tcol64, real* d_tcol, void* stream) {
  if (!grid || grid->d < 1 || grid->d > WISKI_MAX_DIM || !plan_ok(plan, grid->d) || kind < 0 || kind > 3 || !d_ell || !d_s2) return WISKI_E_BADARG;
  ColumnsGridS G;
  G.d = grid->d;
  for (int q = 0; q < grid->d; ++q) {
    if (grid->g[q] < 1 || !(grid->h[q] > 0)) return WISKI_E_BADARG;
    G.g[q] = grid->g[q];
    G.h[q] = grid->h[q];
  }
  hipLaunchKernelGGL((k_hyper_columns<real>), dim3(1), dim3(256), 0, (hipStream_t)stream, *plan, G, kind, d_ell, d_scale, d_s2, d_s2_f64, d_tcol64, d_tcol);
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}

template <typename real>
static int hyper_adam_impl(const wiski_hyper_plan* plan, const real* d_scale, const real* d_s2, const real* d_gell, const real* d_gscale, const double* d_mid,
                           const double* d_gkap, const double* d_n, double lr, double beta1, double beta2, double eps, void* stream) {
  if (!plan || plan->count < 1 || plan->count > WISKI_HYPER_MAX_PARAMS || !d_s2 || !d_gell || !d_mid || !d_gkap || !d_n) return WISKI_E_BADARG;
  int total = 0;
  for (int i = 0; i < plan->count; ++i) {
    const wiski_hyper_param& p = plan->p[i];
    if (!p.raw || p.numel < 1 || p.role < 0 || p.role > 2 || p.kind < 0 || p.kind > 1 || (p.role != 0 && p.numel != 1)) return WISKI_E_BADARG;
    total += p.numel;
  }
  if (total > 64) return WISKI_E_BADARG;                  // (one thread per element, one wave)
  for (int i = 0; i < plan->count; ++i) {
    const wiski_hyper_param& p = plan->p[i];
    if (!p.exp_avg || !p.exp_avg_sq || !p.step || (p.step_numel != 1 && p.step_numel != p.numel)) return WISKI_E_BADARG;
    if (p.role == 1 && (!d_scale || !d_gscale)) return WISKI_E_BADARG;
  }
  hipLaunchKernelGGL((k_hyper_adam<real>), dim3(1), dim3(64), 0, (hipStream_t)stream, *plan, d_scale, d_s2, d_gell, d_gscale, d_mid, d_gkap, d_n, lr, beta1, beta2,
                     eps);
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}

extern "C" {
int wiski_hyper_columns_f32(const wiski_hyper_plan* plan, const wiski_grid* grid, int32_t kind, float* d_ell, float* d_scale, float* d_s2, double* d_s2_f64,
                            double* d_tcol64, float* d_tcol, void* stream) {
  return hyper_columns_impl<float>(plan, grid, kind, d_ell, d_scale, d_s2, d_s2_f64, d_tcol64, d_tcol, stream);
}
int wiski_hyper_columns_f64(const wiski_hyper_plan* plan, const wiski_grid* grid, int32_t kind, double* d_ell, double* d_scale, double* d_s2, double* d_s2_f64,
                            double* d_tcol64, double* d_tcol, void* stream) {
  return hyper_columns_impl<double>(plan, grid, kind, d_ell, d_scale, d_s2, d_s2_f64, d_tcol64, d_tcol, stream);
}
int wiski_hyper_mid_f32(const double* d_bMb, const double* d_logdet, const float* d_s2, const double* d_c, const double* d_ld, const double* d_n, double* d_out,
                        double* d_loss, void* stream) {
  if (!d_bMb || !d_s2 || !d_c || !d_ld || !d_n || !d_out) return WISKI_E_BADARG;
  hipLaunchKernelGGL((k_hyper_mid<float>), dim3(1), dim3(64), 0, (hipStream_t)stream, d_bMb, d_logdet, d_s2, d_c, d_ld, d_n, d_out, d_loss);
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}
int wiski_hyper_mid_f64(const double* d_bMb, const double* d_logdet, const double* d_s2, const double* d_c, const double* d_ld, const double* d_n, double* d_out,
                        double* d_loss, void* stream) {
  if (!d_bMb || !d_s2 || !d_c || !d_ld || !d_n || !d_out) return WISKI_E_BADARG;
  hipLaunchKernelGGL((k_hyper_mid<double>), dim3(1), dim3(64), 0, (hipStream_t)stream, d_bMb, d_logdet, d_s2, d_c, d_ld, d_n, d_out, d_loss);
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}
int wiski_hyper_adam_f32(const wiski_hyper_plan* plan, const float* d_scale, const float* d_s2, const float* d_gell, const float* d_gscale, const double* d_mid,
                         const double* d_gkap, const double* d_n, double lr, double beta1, double beta2, double eps, void* stream) {
  return hyper_adam_impl<float>(plan, d_scale, d_s2, d_gell, d_gscale, d_mid, d_gkap, d_n, lr, beta1, beta2, eps, stream);
}
int wiski_hyper_adam_f64(const wiski_hyper_plan* plan, const double* d_scale, const double* d_s2, const double* d_gell, const double* d_gscale, const double* d_mid,
                         const double* d_gkap, const double* d_n, double lr, double beta1, double beta2, double eps, void* stream) {
  return hyper_adam_impl<double>(plan, d_scale, d_s2, d_gell, d_gscale, d_mid, d_gkap, d_n, lr, beta1, beta2, eps, stream);
}
}

// ---------------------------------------------------------------- staging ---
// Before a replay the factor state of this step (three r x r matrices, seven small vectors / scalars, the data count) goes into the
// static buffers the recorded kernels read: ONE launch for all segments (it was three copies, a concatenation and a pinned upload).
__global__ __launch_bounds__(256) void k_multi_copy(wiski_copy_plan plan) {
  for (int s = 0; s < plan.count; ++s) {
    const double* __restrict__ src = (const double*)plan.src[s];
    double* __restrict__ dst = (double*)plan.dst[s];
    const int64_t n = plan.n[s];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
  }
  if (plan.scalar_dst && blockIdx.x == 0 && threadIdx.x == 0) ((double*)plan.scalar_dst)[0] = plan.scalar;
}

extern "C" int wiski_multi_copy_f64(const wiski_copy_plan* plan, void* stream) {
  if (!plan || plan->count < 0 || plan->count > WISKI_COPY_MAX_SEGMENTS) return WISKI_E_BADARG;
  int64_t most = 1;
  for (int s = 0; s < plan->count; ++s) {
    if (!plan->src[s] || !plan->dst[s] || plan->n[s] < 0) return WISKI_E_BADARG;
    if (plan->n[s] > most) most = plan->n[s];
  }
  const unsigned blocks = (unsigned)((most + 255) / 256 < 1024 ? (most + 255) / 256 : 1024);
  hipLaunchKernelGGL(k_multi_copy, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *plan);
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}
