// a1 / a14: cubic interpolation stencils and the predictive gather SpMM.
//   reference call sites: BFN:143,205,261,421 (evaluate_kernel -> interp
//   indices/values) and BFN:206-210,235 (left_interp).
#include "wiski_common.h"

// ---------------------------------------------------------------- interp ---
// One thread per (point, tap): the T-wide rows of idx/val are written fully
// coalesced (tap fastest).  The per-dim stencil is recomputed per tap -- a few
// dozen flops against an 8..12-byte store, this kernel is write-bound.
template <typename real, int D>
__global__ __launch_bounds__(256) void k_interp(GridDev<real> G, const real* __restrict__ x, int64_t n,
                                                int32_t* __restrict__ idx, real* __restrict__ val, int32_t* __restrict__ err) {
  constexpr int T = 1 << (2 * D);
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * T) return;
  int64_t p = e / T;
  int a = (int)(e - p * T);
  int flat = 0;
  real v = (real)1;
  bool ok = true;
#pragma unroll
  for (int q = 0; q < D; ++q) {
    real w[4];
    int j0 = dim_stencil<real>(x[p * D + q], G.g0[q], G.h[q], G.hi[q], G.g[q], w);
    int c = (a >> (2 * (D - 1 - q))) & 3;
    if (j0 < 0) { ok = false; j0 = 0; w[c] = (real)0; }
    flat += (j0 + c) * G.stride[q];
    v *= w[c];
  }
  idx[e] = flat;
  val[e] = v;
  if (!ok && a == 0) atomicOr(err, 1);
}

// ---------------------------------------------------------- fused gather ---
// One thread per query row; the 4^D taps are a fully unrolled loop nest, the
// 4^D loads per column are independent (deep memory-level parallelism) and hit
// L2 / MALL: V (k*m reals) is small and read-mostly, x is the only HBM stream.
template <typename real, int D>
__device__ __forceinline__ real gather_one(const GridDev<real>& G, const int j0[D], const real w[D][4], const real* __restrict__ v) {
  constexpr int off = 0;
  real acc = (real)0;
  if constexpr (D == 1) {
#pragma unroll
    for (int c0 = 0; c0 < 4; ++c0) acc += w[0][c0] * v[j0[0] + c0];
  } else if constexpr (D == 2) {
#pragma unroll
    for (int c0 = 0; c0 < 4; ++c0) {
      const real* r0 = v + (j0[0] + c0) * G.stride[off] + j0[1];
      real s = (real)0;
#pragma unroll
      for (int c1 = 0; c1 < 4; ++c1) s += w[1][c1] * r0[c1];
      acc += w[0][c0] * s;
    }
  } else if constexpr (D == 3) {
#pragma unroll
    for (int c0 = 0; c0 < 4; ++c0) {
      real s0 = (real)0;
#pragma unroll
      for (int c1 = 0; c1 < 4; ++c1) {
        const real* r = v + (j0[0] + c0) * G.stride[off] + (j0[1] + c1) * G.stride[off + 1] + j0[2];
        real s1 = (real)0;
#pragma unroll
        for (int c2 = 0; c2 < 4; ++c2) s1 += w[2][c2] * r[c2];
        s0 += w[1][c1] * s1;
      }
      acc += w[0][c0] * s0;
    }
  } else {
#pragma unroll
    for (int c0 = 0; c0 < 4; ++c0) {
      real s0 = (real)0;
#pragma unroll
      for (int c1 = 0; c1 < 4; ++c1) {
        real s1 = (real)0;
#pragma unroll
        for (int c2 = 0; c2 < 4; ++c2) {
          const real* r = v + (j0[0] + c0) * G.stride[0] + (j0[1] + c1) * G.stride[1] + (j0[2] + c2) * G.stride[2] + j0[3];
          real s2 = (real)0;
#pragma unroll
          for (int c3 = 0; c3 < 4; ++c3) s2 += w[3][c3] * r[c3];
          s1 += w[2][c2] * s2;
        }
        s0 += w[1][c1] * s1;
      }
      acc += w[0][c0] * s0;
    }
  }
  return acc;
}

template <typename real, int D>
__global__ __launch_bounds__(256) void k_gather(GridDev<real> G, const real* __restrict__ x, int64_t n, const real* __restrict__ V,
                                                int k, int diag, real* __restrict__ out, int32_t* __restrict__ err) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  int j0[D];
  real w[D][4];
  real xp[D];
#pragma unroll
  for (int q = 0; q < D; ++q) xp[q] = x[p * D + q];
  if (!point_stencil<real, D>(G, xp, j0, w)) atomicOr(err, 1);
  if (diag) {
    out[p] = gather_one<real, D>(G, j0, w, V + (int64_t)p * G.m);
  } else {
    for (int c = 0; c < k; ++c) out[p * k + c] = gather_one<real, D>(G, j0, w, V + (int64_t)c * G.m);
  }
}

// Small batches on 3-D grids (the predictive mean of a streamed batch: q = 4 096 points are only 64 waves for the kernel
// above, each walking its 64 taps alone -- 10 us, latency-bound): 16 lanes per point, lane = (c0, c1) tap prefix with its four
// consecutive innermost taps in one 16-byte load; the 16-lane groups are DPP rows, so the tap sums meet in lane 15 of the row.
template <typename real>
__global__ __launch_bounds__(256) void k_gather_coop3(GridDev<real> G, const real* __restrict__ x, int64_t n, const real* __restrict__ V,
                                                      int k, real* __restrict__ out, int32_t* __restrict__ err,
                                                      uint32_t* __restrict__ z1 = nullptr, int64_t n1 = 0, uint32_t* __restrict__ z2 = nullptr,
                                                      int64_t n2 = 0) {
  // optional: zero two word arrays on the way (wiski_gather_zero: the scalar block and the accumulated partial vector of the
  // solve that follows in the same streaming step -- saves that solve's own zero launch)
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n1; e += (int64_t)gridDim.x * blockDim.x) z1[e] = 0u;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n2; e += (int64_t)gridDim.x * blockDim.x) z2[e] = 0u;
  const int sub = threadIdx.x & 15;
  const int64_t p = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const bool live = p < n;
  const int64_t pp = live ? p : 0;
  int j0[3];
  real w[3][4], xp[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) xp[q] = x[pp * 3 + q];
  if (!point_stencil<real, 3>(G, xp, j0, w) && live && sub == 0) atomicOr(err, 1);
  const int c0 = sub >> 2, c1 = sub & 3;
  const real w01 = (c0 == 0 ? w[0][0] : c0 == 1 ? w[0][1] : c0 == 2 ? w[0][2] : w[0][3]) * (c1 == 0 ? w[1][0] : c1 == 1 ? w[1][1] : c1 == 2 ? w[1][2] : w[1][3]);
  const int64_t flat = (int64_t)(j0[0] + c0) * G.stride[0] + (int64_t)(j0[1] + c1) * G.stride[1] + j0[2];
  typedef real vec4u __attribute__((ext_vector_type(4), aligned(4)));
  for (int c = 0; c < k; ++c) {
    const vec4u r = *reinterpret_cast<const vec4u*>(V + (int64_t)c * G.m + flat);
    real s = w01 * (w[2][0] * r[0] + w[2][1] * r[1] + w[2][2] * r[2] + w[2][3] * r[3]);
    s = wave_dpp_add<0x111, 0xf>(s);
    s = wave_dpp_add<0x112, 0xf>(s);
    s = wave_dpp_add<0x114, 0xf>(s);
    s = wave_dpp_add<0x118, 0xf>(s);
    if (sub == 15 && live) out[p * k + c] = s;
  }
}

// ------------------------------------------------------------ ELL gather ---
// Rows of (idx, val) are streamed from HBM with 16-byte-per-lane loads:
// LPR = T/4 lanes cooperate on one row (4 taps each), then an xor-shuffle
// reduction inside the LPR-lane group.  v is L2-resident.
template <typename real, int LPR>
__global__ __launch_bounds__(256) void k_gather_ell(const int32_t* __restrict__ idx, const real* __restrict__ val, int64_t n,
                                                    const real* __restrict__ v, real* __restrict__ out) {
  constexpr int RPB = 256 / LPR;  // rows per block pass
  constexpr int U = 4;            // rows in flight per lane group: all idx / val loads of a pass are issued before the first gather
  const int sub = threadIdx.x % LPR;
  const int rloc = threadIdx.x / LPR;
  const int64_t stride = (int64_t)gridDim.x * RPB;
  typedef real vec4u __attribute__((ext_vector_type(4), aligned(4)));
  for (int64_t row0 = (int64_t)blockIdx.x * RPB + rloc; row0 < n; row0 += U * stride) {
    int4 id[U];
    real a[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = row0 + u * stride;
      const int64_t base = ((row < n ? row : row0) * LPR + sub) * 4;
      typedef int nt_i4 __attribute__((ext_vector_type(4)));
      typedef real nt_r4 __attribute__((ext_vector_type(4)));
      // idx / val are read exactly once: non-temporal, so the stream does not push v out of L2
      const nt_i4 qi = __builtin_nontemporal_load(reinterpret_cast<const nt_i4*>(idx + base));
      const nt_r4 qv = __builtin_nontemporal_load(reinterpret_cast<const nt_r4*>(val + base));
      id[u] = make_int4(qi[0], qi[1], qi[2], qi[3]);
      a[u][0] = qv[0]; a[u][1] = qv[1]; a[u][2] = qv[2]; a[u][3] = qv[3];
    }
    real acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      // the 4 taps of a lane are the innermost digits of one tap prefix: 4 CONSECUTIVE grid indices except in the one-hot
      // boundary cells, so v is fetched with one (dword-aligned) 16-byte load instead of four gathers
      const int4 q = id[u];
      if (q.y == q.x + 1 && q.z == q.x + 2 && q.w == q.x + 3) {
        const vec4u g4 = *reinterpret_cast<const vec4u*>(v + q.x);
        acc[u] = a[u][0] * g4[0] + a[u][1] * g4[1] + a[u][2] * g4[2] + a[u][3] * g4[3];
      } else {
        acc[u] = a[u][0] * v[q.x] + a[u][1] * v[q.y] + a[u][2] * v[q.z] + a[u][3] * v[q.w];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int o = LPR / 2; o > 0; o >>= 1) acc[u] += __shfl_xor(acc[u], o, 64);
      const int64_t row = row0 + u * stride;
      if (sub == 0 && row < n) out[row] = acc[u];
    }
  }
}

#include "gather_ell_dma.h"

// ------------------------------------------------------------- W^T columns --
// out[p][idx] += val for the T taps of query p (one dense m-column per query).
template <typename real, int D>
__global__ __launch_bounds__(256) void k_wt_columns(GridDev<real> G, const real* __restrict__ x, int64_t n, real* __restrict__ out,
                                                    int32_t* __restrict__ err) {
  constexpr int T = 1 << (2 * D);
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * T) return;
  int64_t p = e / T;
  int a = (int)(e - p * T);
  int flat = 0;
  real v = (real)1;
  bool ok = true;
#pragma unroll
  for (int q = 0; q < D; ++q) {
    real w[4];
    int j0 = dim_stencil<real>(x[p * D + q], G.g0[q], G.h[q], G.hi[q], G.g[q], w);
    int c = (a >> (2 * (D - 1 - q))) & 3;
    if (j0 < 0) { ok = false; j0 = 0; w[c] = (real)0; }
    flat += (j0 + c) * G.stride[q];
    v *= w[c];
  }
  if (v != (real)0) atomic_add_real(out + p * (int64_t)G.m + flat, v);
  if (!ok && a == 0) atomicOr(err, 1);
}

// ------------------------------------------------ gradient w.r.t. the inputs ---
// out[p][q] = d/dx_q ( W(x_p) . V_c ),  c = 0 (k == 1) or c = p (diag): the derivative of the
// cubic interpolation row, needed when a learned stem feeds the GP (sm_partial_mll,
// reference online_gp/mlls/streaming_partial_mll.py:20-36 differentiates through W).
// One thread per (query, dim): product of the other dims' weights and this dim's
// weight derivative k'(s)/h.  Boundary (one-hot) cells have zero derivative.
template <typename real, int D>
__global__ __launch_bounds__(256) void k_gather_grad(GridDev<real> G, const real* __restrict__ x, int64_t n, const real* __restrict__ V, int diag,
                                                     real* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * D) return;
  const int64_t p = e / D;
  const int qd = (int)(e - p * D);
  int j0[D];
  real w[D][4];
#pragma unroll
  for (int q = 0; q < D; ++q) {
    const real xv = x[p * D + q];
    int j = dim_stencil<real>(xv, G.g0[q], G.h[q], G.hi[q], G.g[q], w[q]);
    if (j < 0) {
      j = 0;
#pragma unroll
      for (int c = 0; c < 4; ++c) w[q][c] = (real)0;
    }
    if (q == qd) {
      const real u = (xv - G.g0[q]) / G.h[q];
      const real fl = floor(u);
      const real t = u - fl;
      const int jj = (int)fl - 1;
      const bool interior = !(jj < 0 || jj > G.g[q] - 4);
#pragma unroll
      for (int c = 0; c < 4; ++c) w[q][c] = interior ? keys_cubic_deriv<real>(t + (real)1 - (real)c) / G.h[q] : (real)0;
    }
    j0[q] = j;
  }
  const real* __restrict__ v = V + (diag ? p : 0) * (int64_t)G.m;
  out[e] = gather_one<real, D>(G, j0, w, v);
}

template <typename real>
static int gather_grad_impl(const wiski_grid* grid, const real* d_x, int64_t n, const real* d_V, int32_t diag, real* d_out, void* stream) {
  GridDev<real> G;
  int rc = make_grid_dev<real>(grid, &G);
  if (rc) return rc;
  if (n == 0) return WISKI_OK;
  if (!d_x || !d_V || !d_out) return WISKI_E_BADARG;
  dim3 grd((unsigned)((n * G.d + 255) / 256));
#define CALL(DD) hipLaunchKernelGGL((k_gather_grad<real, DD>), grd, dim3(256), 0, (hipStream_t)stream, G, d_x, n, d_V, diag, d_out)
  WISKI_DISPATCH_D(G.d, CALL)
#undef CALL
  WISKI_LAUNCH_CHECK();
  return WISKI_OK;
}

// ------------------------------------------------------- row-major gather ---
// out[p][c] = sum_a val_a(x_p) * Vr[idx_a(x_p)][c]  for Vr stored row-major [m][ncols]
// (the layout of left_interp's dense operand, BFN:206-210).  One block per query:
// the 4^D taps are computed once into LDS, then lanes run over columns so that every
// tap reads a contiguous row segment of Vr (coalesced; Vr is L2 / MALL resident).
// Used with Vr = M (dense posterior of small grids): W* M in n* x 4^D row reads.
template <typename real, int D>
__global__ __launch_bounds__(256) void k_gather_rows(GridDev<real> G, const real* __restrict__ x, int64_t n, const real* __restrict__ Vr,
                                                     int ncols, real* __restrict__ out, int32_t* __restrict__ err) {
  constexpr int T = 1 << (2 * D);
  __shared__ int s_idx[T];
  __shared__ real s_val[T];
  const int64_t p = blockIdx.x;
  for (int a = threadIdx.x; a < T; a += blockDim.x) {
    int flat = 0;
    real v = (real)1;
    bool ok = true;
#pragma unroll
    for (int q = 0; q < D; ++q) {
      real w[4];
      int j0 = dim_stencil<real>(x[p * D + q], G.g0[q], G.h[q], G.hi[q], G.g[q], w);
      const int c = (a >> (2 * (D - 1 - q))) & 3;
      if (j0 < 0) { ok = false; j0 = 0; w[c] = (real)0; }
      flat += (j0 + c) * G.stride[q];
      v *= w[c];
    }
    s_idx[a] = flat;
    s_val[a] = v;
    if (!ok && a == 0) atomicOr(err, 1);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < ncols; c += blockDim.x) {
    real acc = (real)0;
#pragma unroll 8
    for (int a = 0; a < T; ++a) acc += s_val[a] * Vr[(int64_t)s_idx[a] * ncols + c];
    out[p * ncols + c] = acc;
  }
}

template <typename real>
static int gather_rows_impl(const wiski_grid* grid, const real* d_x, int64_t n, const real* d_Vr, int32_t ncols, real* d_out, int32_t* d_err,
                            void* stream) {
  GridDev<real> G;
  int rc = make_grid_dev<real>(grid, &G);
  if (rc) return rc;
  if (n == 0) return WISKI_OK;
  if (!d_x || !d_Vr || !d_out || !d_err || ncols < 1) return WISKI_E_BADARG;
  dim3 grd((unsigned)n);
#define CALL(DD) hipLaunchKernelGGL((k_gather_rows<real, DD>), grd, dim3(256), 0, (hipStream_t)stream, G, d_x, n, d_Vr, ncols, d_out, d_err)
  WISKI_DISPATCH_D(G.d, CALL)
#undef CALL
  WISKI_LAUNCH_CHECK();
  return WISKI_OK;
}

// ------------------------------------------------------------- host side ---
template <typename real>
static int interp_impl(const wiski_grid* grid, const real* d_x, int64_t n, int32_t* d_idx, real* d_val, int32_t* d_err, void* stream) {
  GridDev<real> G;
  int rc = make_grid_dev<real>(grid, &G);
  if (rc) return rc;
  if (n == 0) return WISKI_OK;
  if (!d_x || !d_idx || !d_val || !d_err) return WISKI_E_BADARG;
  int64_t total = n * G.T;
  dim3 grd((unsigned)((total + 255) / 256));
#define CALL(DD) hipLaunchKernelGGL((k_interp<real, DD>), grd, dim3(256), 0, (hipStream_t)stream, G, d_x, n, d_idx, d_val, d_err)
  WISKI_DISPATCH_D(G.d, CALL)
#undef CALL
  WISKI_LAUNCH_CHECK();
  return WISKI_OK;
}

template <typename real>
static int gather_impl(const wiski_grid* grid, const real* d_x, int64_t n, const real* d_V, int32_t k, int32_t diag, real* d_out,
                       int32_t* d_err, void* stream) {
  GridDev<real> G;
  int rc = make_grid_dev<real>(grid, &G);
  if (rc) return rc;
  if (n == 0) return WISKI_OK;
  if (!d_x || !d_V || !d_out || !d_err || k < 1) return WISKI_E_BADARG;
  if (diag && k != n) return WISKI_E_BADARG;
  if (G.d == 3 && !diag && k <= 4 && n <= 65536 && G.g[2] >= 4) {
    hipLaunchKernelGGL((k_gather_coop3<real>), dim3((unsigned)((n + 15) / 16)), dim3(256), 0, (hipStream_t)stream, G, d_x, n, d_V, k, d_out, d_err);
    WISKI_LAUNCH_CHECK();
    return WISKI_OK;
  }
  dim3 grd((unsigned)((n + 255) / 256));
#define CALL(DD) hipLaunchKernelGGL((k_gather<real, DD>), grd, dim3(256), 0, (hipStream_t)stream, G, d_x, n, d_V, k, diag, d_out, d_err)
  WISKI_DISPATCH_D(G.d, CALL)
#undef CALL
  WISKI_LAUNCH_CHECK();
  return WISKI_OK;
}

template <typename real>
static int gather_zero_impl(const wiski_grid* grid, const real* d_x, int64_t n, const real* d_V, int32_t k, real* d_out, int32_t* d_err, void* z1,
                            int64_t n1_bytes, void* z2, int64_t n2_bytes, int32_t* zeroed, void* stream) {
  if (zeroed) *zeroed = 0;
  GridDev<real> G;
  int rc = make_grid_dev<real>(grid, &G);
  if (rc) return rc;
  const bool coop = G.d == 3 && k >= 1 && k <= 4 && n >= 1 && n <= 65536 && G.g[2] >= 4 && zeroed && (n1_bytes % 4) == 0 && (n2_bytes % 4) == 0 &&
                    (n1_bytes == 0 || z1) && (n2_bytes == 0 || z2);
  if (!coop) return gather_impl<real>(grid, d_x, n, d_V, k, 0, d_out, d_err, stream);
  if (!d_x || !d_V || !d_out || !d_err) return WISKI_E_BADARG;
  hipLaunchKernelGGL((k_gather_coop3<real>), dim3((unsigned)((n + 15) / 16)), dim3(256), 0, (hipStream_t)stream, G, d_x, n, d_V, k, d_out, d_err,
                     (uint32_t*)z1, n1_bytes / 4, (uint32_t*)z2, n2_bytes / 4);
  WISKI_LAUNCH_CHECK();
  *zeroed = 1;
  return WISKI_OK;
}

static int ell_cu_count() {
  static int n = 0;
  if (n == 0) {
    int dev = 0, v = 0;
    n = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  }
  return n;
}
// tuning hooks of tools/gather_ell_probe.py (passes per tile, total waves, register-staged kernel instead, contiguous tile ranges; 0 = default)
static int g_ell_p = 0, g_ell_waves = 0, g_ell_off = 0, g_ell_contig = 0;
extern "C" void wiski_gather_ell_tune(int32_t p, int32_t waves, int32_t off, int32_t contig) { g_ell_p = p; g_ell_waves = waves; g_ell_off = off; g_ell_contig = contig; }

template <typename real, int LPR, int P>
static int gather_ell_dma_launch(const int32_t* d_idx, const real* d_val, int64_t n, const real* d_v, real* d_out, hipStream_t s) {
  using Gm = EllDmaGeom<real, LPR, P>;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)k_gather_ell_dma<real, LPR, P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Gm::LDS_B) != hipSuccess) return WISKI_E_LAUNCH;
    attr_done = true;
  }
  const int64_t ntiles = (n + Gm::RPT - 1) / Gm::RPT;
  int64_t waves = g_ell_waves > 0 ? g_ell_waves : (int64_t)ell_cu_count() * 2;
  if (waves > ntiles) waves = ntiles;
  hipLaunchKernelGGL((k_gather_ell_dma<real, LPR, P>), dim3((unsigned)waves), dim3(64), Gm::LDS_B, s, d_idx, d_val, n, d_v, d_out, ntiles, g_ell_contig, EllV4Geo{});
  WISKI_LAUNCH_CHECK();
  return WISKI_OK;
}

template <typename real>
static int gather_ell_impl(const int32_t* d_idx, const real* d_val, int64_t n, int32_t T, const real* d_v, real* d_out, void* stream);

static inline bool ell_v4_geo(const wiski_grid* grid, EllV4Geo* geo, int64_t* groups) {
  if (!grid || grid->d < 2 || grid->d > WISKI_MAX_DIM) return false;
  int64_t m = 1;
  for (int q = 0; q < grid->d; ++q) m *= grid->g[q];
  if (m >= ((int64_t)1 << 29)) return false;                  // (flat and packed indices stay below 2^31)
  geo->gL = (unsigned)grid->g[grid->d - 1];
  geo->gK = (unsigned)grid->g[grid->d - 2];
  geo->nbk = (geo->gK + 3) / 4;
  ell_magic(geo->gL, &geo->mulL, &geo->shL);
  ell_magic(geo->gK, &geo->mulK, &geo->shK);
  *groups = m / ((int64_t)geo->gK * geo->gL) * geo->nbk * geo->gL;
  if (16 * *groups >= ((int64_t)1 << 31)) return false;
  geo->groups = (unsigned)*groups;
  return true;
}

template <typename real, int LPR, int P>
static int gather_ell_v4_launch(const EllV4Geo& geo, const int32_t* d_idx, const real* d_val, int64_t n, const real* d_v4, real* d_out, hipStream_t s) {
  using Gm = EllDmaGeom<real, LPR, P>;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)k_gather_ell_dma<real, LPR, P, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Gm::LDS_B) != hipSuccess) return WISKI_E_LAUNCH;
    attr_done = true;
  }
  const int64_t ntiles = (n + Gm::RPT - 1) / Gm::RPT;
  int64_t waves = g_ell_waves > 0 ? g_ell_waves : (int64_t)ell_cu_count() * (sizeof(real) == 4 ? 3 : 2);     // (measured: fp32 2-3 per CU alike, fp64 2: 149 / 156 / 202 us at 2 / 3 / 4)
  if (waves > ntiles) waves = ntiles;
  hipLaunchKernelGGL((k_gather_ell_dma<real, LPR, P, true>), dim3((unsigned)waves), dim3(64), Gm::LDS_B, s, d_idx, d_val, n, d_v4, d_out, ntiles, g_ell_contig, geo);
  WISKI_LAUNCH_CHECK();
  return WISKI_OK;
}

template <typename real>
static int gather_ell_grid_impl(const wiski_grid* grid, const int32_t* d_idx, const real* d_val, int64_t n, const real* d_v, real* d_vpack, real* d_out, void* stream) {
  if (!grid || grid->d < 1 || grid->d > WISKI_MAX_DIM) return WISKI_E_BADARG;
  int T = 1;
  for (int q = 0; q < grid->d; ++q) T *= 4;
  if (n == 0) return WISKI_OK;
  if (!d_idx || !d_val || !d_v || !d_out) return WISKI_E_BADARG;
  {
  EllV4Geo geo;
  int64_t groups = 0;
  // the blocked copy pays from ~2^14 rows on (one extra pass over v); d = 1 rows touch one or two lines as they are
  if (!d_vpack || g_ell_off || !ell_v4_geo(grid, &geo, &groups) || n * (int64_t)T < ((int64_t)1 << 20) || (((uintptr_t)d_idx | (uintptr_t)d_val | (uintptr_t)d_vpack) & 15) != 0)
    return gather_ell_impl<real>(d_idx, d_val, n, T, d_v, d_out, stream);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL((k_ell_pack_v4s<real>), dim3((unsigned)((4 * groups + 255) / 256)), dim3(256), 0, s, d_v, d_vpack, geo, groups);
  const int pp = g_ell_p ? g_ell_p : 8;
#define ELL_V4(LPR_)                                                                                     \
  do {                                                                                                   \
    if (pp == 4) return gather_ell_v4_launch<real, LPR_, 4>(geo, d_idx, d_val, n, d_vpack, d_out, s);    \
    return gather_ell_v4_launch<real, LPR_, 8>(geo, d_idx, d_val, n, d_vpack, d_out, s);                 \
  } while (0)
  switch (T) {
    case 16: ELL_V4(4);
    case 64: ELL_V4(16);
    case 256: ELL_V4(64);
    default: return WISKI_E_BADARG;
  }
#undef ELL_V4
  }
}

template <typename real>
static int gather_ell_impl(const int32_t* d_idx, const real* d_val, int64_t n, int32_t T, const real* d_v, real* d_out, void* stream) {
  if (n == 0) return WISKI_OK;
  if (!d_idx || !d_val || !d_v || !d_out) return WISKI_E_BADARG;
  // large row counts with 16-byte aligned arrays: the LDS-DMA staged kernel (gather_ell_dma.h); small ones are launch-bound either way
  if (!g_ell_off && n * (int64_t)T >= ((int64_t)1 << 20) && (((uintptr_t)d_idx | (uintptr_t)d_val) & 15) == 0) {
    hipStream_t s = (hipStream_t)stream;
    const int pp = g_ell_p ? g_ell_p : 8;
#define ELL_DMA(LPR_)                                                                                  \
  do {                                                                                                 \
    if (pp == 4) return gather_ell_dma_launch<real, LPR_, 4>(d_idx, d_val, n, d_v, d_out, s);          \
    if constexpr (sizeof(real) == 4) if (pp == 16) return gather_ell_dma_launch<real, LPR_, 16>(d_idx, d_val, n, d_v, d_out, s); \
    return gather_ell_dma_launch<real, LPR_, 8>(d_idx, d_val, n, d_v, d_out, s);                       \
  } while (0)
    switch (T) {
      case 4: ELL_DMA(1);
      case 16: ELL_DMA(4);
      case 64: ELL_DMA(16);
      case 256: ELL_DMA(64);
      default: return WISKI_E_BADARG;
    }
#undef ELL_DMA
  }
  int lpr = T / 4;
  int64_t rpb = 256 / (lpr > 0 ? lpr : 1);
  int64_t blocks = (n + rpb - 1) / rpb;
  if (blocks > 256 * 16) blocks = 256 * 16;  // 16 blocks per CU, grid-stride the rest
  dim3 grd((unsigned)blocks);
  hipStream_t s = (hipStream_t)stream;
  switch (T) {
    case 4: hipLaunchKernelGGL((k_gather_ell<real, 1>), grd, dim3(256), 0, s, d_idx, d_val, n, d_v, d_out); break;
    case 16: hipLaunchKernelGGL((k_gather_ell<real, 4>), grd, dim3(256), 0, s, d_idx, d_val, n, d_v, d_out); break;
    case 64: hipLaunchKernelGGL((k_gather_ell<real, 16>), grd, dim3(256), 0, s, d_idx, d_val, n, d_v, d_out); break;
    case 256: hipLaunchKernelGGL((k_gather_ell<real, 64>), grd, dim3(256), 0, s, d_idx, d_val, n, d_v, d_out); break;
    default: return WISKI_E_BADARG;
  }
  WISKI_LAUNCH_CHECK();
  return WISKI_OK;
}

template <typename real>
static int wt_columns_impl(const wiski_grid* grid, const real* d_x, int64_t n, real* d_out, int32_t* d_err, void* stream) {
  GridDev<real> G;
  int rc = make_grid_dev<real>(grid, &G);
  if (rc) return rc;
  if (n == 0) return WISKI_OK;
  if (!d_x || !d_out || !d_err) return WISKI_E_BADARG;
  int64_t total = n * G.T;
  dim3 grd((unsigned)((total + 255) / 256));
#define CALL(DD) hipLaunchKernelGGL((k_wt_columns<real, DD>), grd, dim3(256), 0, (hipStream_t)stream, G, d_x, n, d_out, d_err)
  WISKI_DISPATCH_D(G.d, CALL)
#undef CALL
  WISKI_LAUNCH_CHECK();
  return WISKI_OK;
}

extern "C" {
int wiski_version(void) { return WISKI_VERSION; }
// A stream for work off the critical path (the refresh of the two-level block): created at the device's LOWEST priority, so that where it
// shares the device with a kernel of the caller's stream that wants every wave slot at once (the LDS-DMA SpMV) the dispatcher serves the
// caller's queue first.  *out: a hipStream_t the caller owns (never destroyed by the library).
int wiski_side_stream_create(int32_t lowest_priority, void** out) {
  if (!out) return WISKI_E_BADARG;
  int least = 0, greatest = 0;
  if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void)hipGetLastError(); least = 0; }
  hipStream_t s = nullptr;
  if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, lowest_priority ? least : 0) != hipSuccess) { (void)hipGetLastError(); return WISKI_E_LAUNCH; }
  *out = (void*)s;
  return WISKI_OK;
}
int wiski_interp_f32(const wiski_grid* g, const float* x, int64_t n, int32_t* idx, float* val, int32_t* err, void* s) { return interp_impl<float>(g, x, n, idx, val, err, s); }
int wiski_interp_f64(const wiski_grid* g, const double* x, int64_t n, int32_t* idx, double* val, int32_t* err, void* s) { return interp_impl<double>(g, x, n, idx, val, err, s); }
int wiski_gather_zero_f32(const wiski_grid* g, const float* x, int64_t n, const float* V, int32_t k, float* out, int32_t* err, void* z1, int64_t n1, void* z2, int64_t n2, int32_t* zeroed, void* s) { return gather_zero_impl<float>(g, x, n, V, k, out, err, z1, n1, z2, n2, zeroed, s); }
int wiski_gather_zero_f64(const wiski_grid* g, const double* x, int64_t n, const double* V, int32_t k, double* out, int32_t* err, void* z1, int64_t n1, void* z2, int64_t n2, int32_t* zeroed, void* s) { return gather_zero_impl<double>(g, x, n, V, k, out, err, z1, n1, z2, n2, zeroed, s); }
int wiski_gather_f32(const wiski_grid* g, const float* x, int64_t n, const float* V, int32_t k, int32_t diag, float* out, int32_t* err, void* s) { return gather_impl<float>(g, x, n, V, k, diag, out, err, s); }
int wiski_gather_f64(const wiski_grid* g, const double* x, int64_t n, const double* V, int32_t k, int32_t diag, double* out, int32_t* err, void* s) { return gather_impl<double>(g, x, n, V, k, diag, out, err, s); }
int wiski_gather_grad_f32(const wiski_grid* g, const float* x, int64_t n, const float* V, int32_t diag, float* out, void* s) { return gather_grad_impl<float>(g, x, n, V, diag, out, s); }
int wiski_gather_grad_f64(const wiski_grid* g, const double* x, int64_t n, const double* V, int32_t diag, double* out, void* s) { return gather_grad_impl<double>(g, x, n, V, diag, out, s); }
int wiski_gather_rows_f32(const wiski_grid* g, const float* x, int64_t n, const float* Vr, int32_t ncols, float* out, int32_t* err, void* s) { return gather_rows_impl<float>(g, x, n, Vr, ncols, out, err, s); }
int wiski_gather_rows_f64(const wiski_grid* g, const double* x, int64_t n, const double* Vr, int32_t ncols, double* out, int32_t* err, void* s) { return gather_rows_impl<double>(g, x, n, Vr, ncols, out, err, s); }
int wiski_gather_ell_f32(const int32_t* idx, const float* val, int64_t n, int32_t T, const float* v, float* out, void* s) { return gather_ell_impl<float>(idx, val, n, T, v, out, s); }
int wiski_gather_ell_f64(const int32_t* idx, const double* val, int64_t n, int32_t T, const double* v, double* out, void* s) { return gather_ell_impl<double>(idx, val, n, T, v, out, s); }
int wiski_gather_ell_grid_f32(const wiski_grid* g, const int32_t* idx, const float* val, int64_t n, const float* v, float* vpack, float* out, void* s) { return gather_ell_grid_impl<float>(g, idx, val, n, v, vpack, out, s); }
int wiski_gather_ell_grid_f64(const wiski_grid* g, const int32_t* idx, const double* val, int64_t n, const double* v, double* vpack, double* out, void* s) { return gather_ell_grid_impl<double>(g, idx, val, n, v, vpack, out, s); }
int64_t wiski_gather_ell_pack_elems(const wiski_grid* g) {
  EllV4Geo geo;
  int64_t groups = 0;
  return ell_v4_geo(g, &geo, &groups) ? 16 * groups : 0;
}
int wiski_wt_columns_f32(const wiski_grid* g, const float* x, int64_t n, float* out, int32_t* err, void* s) { return wt_columns_impl<float>(g, x, n, out, err, s); }
int wiski_wt_columns_f64(const wiski_grid* g, const double* x, int64_t n, double* out, int32_t* err, void* s) { return wt_columns_impl<double>(g, x, n, out, err, s); }
}
