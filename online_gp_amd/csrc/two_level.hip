// Refresh of the two-level preconditioner's exact block (include/wiski.h: wiski_twolevel, wiski_twolevel_refresh) in ONE host
// call: for the points absorbed since the last refresh
//     F = diag(sqrt wa) W(x) X_S            (wiski_basis_project, n x r fp64)
//     G += F^T F                            (k_gram_acc: tall-skinny Gram product on v_mfma_f64_16x16x4_f64, split over the
//                                            rows, lower-triangle tiles only, fp64 atomics into both halves of G)
// then  C = I + Lam^1/2 G Lam^1/2 (wiski_woodbury_c),  chol, chol^-1 (wiski_potrf_inverse: one workgroup),
//       N = Lam^1/2 chol^-T chol^-1 Lam^1/2  (wiski_gemm + k_tl_scale_cast, fp32 out).
// Everything is queued on the caller's stream (the package uses a side stream: a stale block costs CG iterations, never accuracy).
#include "wiski_common.h"

extern "C" void wiski_potrf_quiet(int on);   // dense_small.h (dense.hip): the next factorisations take the one-workgroup path

using tl_f64x4 = __attribute__((ext_vector_type(4))) double;

// G[i][j] += sum_{p in chunk} F[p][i] F[p][j] for one 32 x 32 tile (ti >= tj) and one chunk of rows (blockIdx.y).
// 4 waves, one 16 x 16 MFMA tile each; the two 32-row x 32-column operand tiles are staged in LDS with the next pair's loads
// in flight (as k_gemm32).  fp64 C fragment: row = (lane >> 4) + 4 reg, column = lane & 15.
__global__ __launch_bounds__(256) void k_gram_acc(int n, int r, int chunk, const double* __restrict__ F, int64_t ldf, double* __restrict__ G) {
  __shared__ double sA[32][33];        // sA[k][i] = F[p0 + k][i0 + i]
  __shared__ double sB[32][33];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int qa = w >> 1, qb = w & 1;
  // lower-triangle tile (ti >= tj) number blockIdx.x = ti (ti + 1) / 2 + tj
  int ti = (int)((sqrtf(8.f * (float)blockIdx.x + 1.f) - 1.f) * 0.5f);
  while ((ti + 1) * (ti + 2) / 2 <= (int)blockIdx.x) ++ti;
  while (ti * (ti + 1) / 2 > (int)blockIdx.x) --ti;
  const int tj = (int)blockIdx.x - ti * (ti + 1) / 2;
  const int i0 = ti * 32, j0 = tj * 32;
  const int p_lo = blockIdx.y * chunk;
  const int p_hi = min(n, p_lo + chunk);
  const int r8 = tid >> 3, c4 = (tid & 7) * 4;
  tl_f64x4 acc = {0.0, 0.0, 0.0, 0.0};
  double pa[4], pb[4];
  auto fetch = [&](int p0) {
    const int p = p0 + r8;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int gi = i0 + c4 + u, gj = j0 + c4 + u;
      pa[u] = (p < p_hi && gi < r) ? F[(int64_t)p * ldf + gi] : 0.0;
      pb[u] = (p < p_hi && gj < r) ? F[(int64_t)p * ldf + gj] : 0.0;
    }
  };
  fetch(p_lo);
  for (int p0 = p_lo; p0 < p_hi; p0 += 32) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      sA[r8][c4 + u] = pa[u];
      sB[r8][c4 + u] = pb[u];
    }
    __syncthreads();
    if (p0 + 32 < p_hi) fetch(p0 + 32);
#pragma unroll
    for (int ks = 0; ks < 32; ks += 4) {
      const int kk = ks + (lane >> 4);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sA[kk][qa * 16 + (lane & 15)], sB[kk][qb * 16 + (lane & 15)], acc, 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int gi = i0 + qa * 16 + (lane >> 4) + 4 * q, gj = j0 + qb * 16 + (lane & 15);
    if (gi < r && gj < r) {
      unsafeAtomicAdd(G + (int64_t)gi * r + gj, acc[q]);
      if (ti != tj) unsafeAtomicAdd(G + (int64_t)gj * r + gi, acc[q]);
    }
  }
}

// N32[i][j] = sq[i] sq[j] T[i][j] / gscale; a failed factorisation (*info != 0: G was corrupted) poisons N with NaNs so that the block can
// never be used by accident, and the refresh's verdict goes where the caller reads it without a launch of its own: `bad` (device or pinned
// host memory, zeroed by the caller, may be NULL) |= 1 for a failed factorisation or a non-finite entry of N, |= 2 when the slab kernel's
// exchange has counted a word that never arrived since the block was built (*sticky != 0; wiski_twolevel.d_cs[r]).
__global__ void k_tl_scale_cast(int r, const double* __restrict__ T, const double* __restrict__ sq, double inv_gscale, const int32_t* __restrict__ info,
                                float* __restrict__ N, const unsigned long long* __restrict__ sticky, int32_t* bad) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < r * r) {
    const int i = e / r, j = e - i * r;
    const float v = *info ? __builtin_nanf("") : (float)(sq[i] * sq[j] * T[e] * inv_gscale);
    N[e] = v;
    if (bad) {
      int b = (e == 0 && *info) || !(fabsf(v) <= 3.4028234e38f) ? 1 : 0;
      if (e == 0 && sticky && *sticky != 0ull) b |= 2;
      if (b) __hip_atomic_fetch_or(bad, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

constexpr int TL_CHUNK_POINTS = 16384;      // points projected per pass (bounds the F buffer: 16384 x r x 8 B)

static inline int64_t tl_align(int64_t v) { return (v + 255) & ~(int64_t)255; }

extern "C" {
int64_t wiski_twolevel_refresh_workspace_bytes(int32_t r) {
  if (r < 1 || r > 512) return WISKI_E_BADARG;
  return tl_align((int64_t)TL_CHUNK_POINTS * r * 8) + 3 * tl_align((int64_t)r * r * 8) + 2 * tl_align((int64_t)r * 8) + tl_align(16);
}

int wiski_twolevel_refresh_f32(const wiski_grid* grid, const float* d_x, int64_t n, const float* d_scale, const double* d_V, int32_t kw, const int32_t* d_S,
                               int32_t r, const double* d_lam_unit, double kscale, double gscale, double* d_G, void* d_work, int64_t work_bytes, float* d_N,
                               const uint64_t* d_sticky, int32_t* d_bad,
                               void* stream) {
  if (!(gscale > 0)) return WISKI_E_BADARG;
  if (!grid || n < 0 || (n > 0 && !d_x) || !d_V || !d_S || r < 1 || r > 512 || !d_lam_unit || !d_G || !d_work || !d_N) return WISKI_E_BADARG;
  if (work_bytes < wiski_twolevel_refresh_workspace_bytes(r)) return WISKI_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  char* w = (char*)d_work;
  double* F = (double*)w;                 w += tl_align((int64_t)TL_CHUNK_POINTS * r * 8);
  double* C = (double*)w;                 w += tl_align((int64_t)r * r * 8);
  double* Li = (double*)w;                w += tl_align((int64_t)r * r * 8);
  double* T = (double*)w;                 w += tl_align((int64_t)r * r * 8);
  double* lam = (double*)w;               w += tl_align((int64_t)r * 8);
  double* sq = (double*)w;                w += tl_align((int64_t)r * 8);
  const int nt32 = (r + 31) / 32, nt = nt32 * (nt32 + 1) / 2;
  int32_t* info = (int32_t*)w;
  for (int64_t p0 = 0; p0 < n; p0 += TL_CHUNK_POINTS) {
    const int64_t nc = n - p0 < TL_CHUNK_POINTS ? n - p0 : TL_CHUNK_POINTS;
    if (int rc = wiski_basis_project_f32(grid, d_x + p0 * grid->d, nc, d_V, kw, d_S, r, d_scale ? d_scale + p0 : nullptr, nullptr, nullptr, F, r, nullptr,
                                         nullptr, stream))
      return rc;
    // rows per workgroup: enough workgroups to fill the chip a little more than once, at least 256 rows each
    int chunk = (int)((nc * nt + 383) / 384);
    chunk = (chunk + 31) & ~31;
    if (chunk < 256) chunk = 256;
    const int ny = (int)((nc + chunk - 1) / chunk);
    hipLaunchKernelGGL(k_gram_acc, dim3((unsigned)nt, (unsigned)ny), dim3(256), 0, s, (int)nc, (int)r, chunk, (const double*)F, (int64_t)r, d_G);
    if (hipGetLastError() != hipSuccess) return WISKI_E_LAUNCH;
  }
  // N = (D^-1 + gscale G)^-1 with D = kscale lam:  C = I + (gscale D)^1/2 G (gscale D)^1/2,  N = (gscale D)^1/2 C^-1 (gscale D)^1/2 / gscale
  if (int rc = wiski_woodbury_c(r, d_G, d_lam_unit, kscale * gscale, C, lam, sq, nullptr, stream)) return rc;
  if (hipMemsetAsync(info, 0, sizeof(int32_t), s) != hipSuccess) return WISKI_E_LAUNCH;
  {
    // the refresh has two streaming steps of slack and runs beside the solver: the one-workgroup factorisation (one CU) instead of the
    // cooperating one (WISKI_TL_QUIET_POTRF=0: the latter)
    static const bool quiet = [] { const char* e = getenv("WISKI_TL_QUIET_POTRF"); return !(e && e[0] == '0'); }();
    if (quiet) wiski_potrf_quiet(1);
    const int rcp = wiski_potrf_inverse_f64(r, C, r, Li, r, info, stream);
    if (quiet) wiski_potrf_quiet(0);
    if (rcp) return rcp;
  }
  if (int rc = wiski_gemm_f64(1, 0, r, r, r, 1.0, Li, r, Li, r, 0.0, T, r, stream)) return rc;
  hipLaunchKernelGGL(k_tl_scale_cast, dim3((unsigned)((r * r + 255) / 256)), dim3(256), 0, s, (int)r, (const double*)T, (const double*)sq, 1.0 / gscale,
                     (const int32_t*)info, d_N, (const unsigned long long*)d_sticky, d_bad);
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}
}
