// Dense m x m Woodbury-factor path for small inducing grids (m <= ~4096):
// the regime the reference itself runs in (BFN:343-404 with a full Cholesky root,
// m <= max_cholesky_size).  Three hand-written building blocks:
//   wiski_gemm  : C = alpha op(A) op(B) + beta C on the matrix cores
//                 (v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64, exact fp32 / fp64)
//   wiski_potrf : blocked right-looking lower Cholesky (64-wide panels; the
//                 trailing update is the MFMA GEMM)               -- chol(Q), BFN:375
//   wiski_trsm  : blocked forward / backward substitution, many right-hand sides
// All matrices are row-major with explicit leading dimensions.
#include "wiski_common.h"

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f64x4 = __attribute__((ext_vector_type(4))) double;

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f64x4 mfma16(double a, double b, f64x4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

template <typename real> struct Acc4;
template <> struct Acc4<float> { using type = f32x4; };
template <> struct Acc4<double> { using type = f64x4; };

// C/D fragment row of register `reg` for lane `lane` (16x16 tiles; MI355X guide section 3):
// f32: row = (lane>>4)*4 + reg ; f64: row = (lane>>4) + 4*reg ; col = lane & 15 for both.
template <typename real>
__device__ __forceinline__ int frag_row(int lane, int reg) {
  if constexpr (sizeof(real) == 4) return (lane >> 4) * 4 + reg;
  else return (lane >> 4) + 4 * reg;
}

// ------------------------------------------------------------------ GEMM ---
// 64 x 64 block tile, K step 16, 4 waves each computing a 32 x 32 sub-tile as
// 2 x 2 MFMA tiles.  op(A) tile is staged in LDS as sA[i][k], op(B) as sB[k][j]
// whatever the storage order, so one compute core serves all four transposes.
constexpr int GBM = 64, GBN = 64, GBK = 16;

template <typename real, bool TA, bool TB>
__global__ __launch_bounds__(256) void k_gemm(int M, int N, int K, real alpha, const real* __restrict__ A, int lda, const real* __restrict__ B,
                                              int ldb, real beta, real* __restrict__ C, int ldc) {
  __shared__ real sA[GBM][GBK + 1];
  __shared__ real sB[GBK][GBN + 4];
  using acc_t = typename Acc4<real>::type;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wr = w >> 1, wc = w & 1;
  const int m0 = blockIdx.y * GBM, n0 = blockIdx.x * GBN;
  acc_t acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[a][b][r] = (real)0;

  for (int k0 = 0; k0 < K; k0 += GBK) {
    // stage op(A)[m0.., k0..] -> sA[i][k]
    if constexpr (!TA) {
      const int i = tid >> 2, kk = (tid & 3) * 4;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int gi = m0 + i, gk = k0 + kk + u;
        sA[i][kk + u] = (gi < M && gk < K) ? A[(int64_t)gi * lda + gk] : (real)0;
      }
    } else {
      const int kk = tid >> 4, i = (tid & 15) * 4;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int gi = m0 + i + u, gk = k0 + kk;
        sA[i + u][kk] = (gi < M && gk < K) ? A[(int64_t)gk * lda + gi] : (real)0;
      }
    }
    // stage op(B)[k0.., n0..] -> sB[k][j]
    if constexpr (!TB) {
      const int kk = tid >> 4, j = (tid & 15) * 4;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int gk = k0 + kk, gj = n0 + j + u;
        sB[kk][j + u] = (gk < K && gj < N) ? B[(int64_t)gk * ldb + gj] : (real)0;
      }
    } else {
      const int j = tid >> 2, kk = (tid & 3) * 4;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int gk = k0 + kk + u, gj = n0 + j;
        sB[kk + u][j] = (gk < K && gj < N) ? B[(int64_t)gj * ldb + gk] : (real)0;
      }
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < GBK; ks += 4) {
      real af[2], bf[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) af[a] = sA[wr * 32 + a * 16 + (lane & 15)][ks + (lane >> 4)];
#pragma unroll
      for (int b = 0; b < 2; ++b) bf[b] = sB[ks + (lane >> 4)][wc * 32 + b * 16 + (lane & 15)];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = mfma16(af[a], bf[b], acc[a][b]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gi = m0 + wr * 32 + a * 16 + frag_row<real>(lane, r);
        const int gj = n0 + wc * 32 + b * 16 + (lane & 15);
        if (gi < M && gj < N) {
          const int64_t e = (int64_t)gi * ldc + gj;
          const real v = alpha * acc[a][b][r];
          C[e] = beta == (real)0 ? v : v + beta * C[e];
        }
      }
}

template <typename real>
static int launch_gemm(int ta, int tb, int M, int N, int K, real alpha, const real* A, int lda, const real* B, int ldb, real beta, real* C,
                       int ldc, hipStream_t s) {
  if (M <= 0 || N <= 0) return WISKI_OK;
  dim3 grd((unsigned)((N + GBN - 1) / GBN), (unsigned)((M + GBM - 1) / GBM));
  if (!ta && !tb) hipLaunchKernelGGL((k_gemm<real, false, false>), grd, dim3(256), 0, s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc);
  else if (ta && !tb) hipLaunchKernelGGL((k_gemm<real, true, false>), grd, dim3(256), 0, s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc);
  else if (!ta && tb) hipLaunchKernelGGL((k_gemm<real, false, true>), grd, dim3(256), 0, s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc);
  else hipLaunchKernelGGL((k_gemm<real, true, true>), grd, dim3(256), 0, s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc);
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}

// -------------------------------------------------------------- Cholesky ---
constexpr int NB = 64;

// In-place Cholesky of one nb x nb diagonal block (nb <= 64) held in LDS.
// info != 0 when a non-positive pivot was met (the caller adds jitter and retries,
// as psd_safe_cholesky does for the reference -- URLT:5).
template <typename real>
__global__ __launch_bounds__(256) void k_potrf_diag(int nb, real* __restrict__ A, int lda, int32_t* __restrict__ info) {
  __shared__ real sL[NB][NB + 1];
  const int tid = threadIdx.x;
  for (int e = tid; e < nb * nb; e += 256) sL[e / nb][e % nb] = A[(int64_t)(e / nb) * lda + (e % nb)];
  __syncthreads();
  for (int k = 0; k < nb; ++k) {
    const real piv = sL[k][k];
    if (!(piv > (real)0)) {
      if (tid == 0) atomicOr(info, 1);
    }
    const real d = sqrt(piv > (real)0 ? piv : (real)1);
    __syncthreads();
    if (tid == 0) sL[k][k] = d;
    for (int i = k + 1 + tid; i < nb; i += 256) sL[i][k] /= d;
    __syncthreads();
    const int rem = nb - k - 1;
    for (int e = tid; e < rem * rem; e += 256) {
      const int i = k + 1 + e / rem, j = k + 1 + e % rem;
      if (j <= i) sL[i][j] -= sL[i][k] * sL[j][k];
    }
    __syncthreads();
  }
  for (int e = tid; e < nb * nb; e += 256) {
    const int i = e / nb, j = e % nb;
    A[(int64_t)i * lda + j] = j <= i ? sL[i][j] : (real)0;
  }
}

// Panel: X L11^T = A21  ->  rows of A21 (below the nb x nb block L11) overwritten by X.
// One thread per row, L11 broadcast from LDS, the row kept in LDS column-major.
template <typename real>
__global__ __launch_bounds__(64) void k_trsm_panel(int rows, int nb, const real* __restrict__ L11, int ldl, real* __restrict__ A21, int lda) {
  __shared__ real sL[NB][NB + 1];
  __shared__ real sX[NB][64 + 1];
  const int tid = threadIdx.x;
  for (int e = tid; e < nb * nb; e += 64) sL[e / nb][e % nb] = L11[(int64_t)(e / nb) * ldl + (e % nb)];
  const int r = blockIdx.x * 64 + tid;
  __syncthreads();
  if (r < rows) {
    for (int j = 0; j < nb; ++j) sX[j][tid] = A21[(int64_t)r * lda + j];
    for (int j = 0; j < nb; ++j) {
      real sacc = sX[j][tid];
      for (int k = 0; k < j; ++k) sacc -= sX[k][tid] * sL[j][k];
      sX[j][tid] = sacc / sL[j][j];
    }
    for (int j = 0; j < nb; ++j) A21[(int64_t)r * lda + j] = sX[j][tid];
  }
}

template <typename real>
static int potrf_impl(int n, real* d_A, int lda, int32_t* d_info, hipStream_t s) {
  if (n < 1 || !d_A || !d_info || lda < n) return WISKI_E_BADARG;
  for (int j = 0; j < n; j += NB) {
    const int nb = n - j < NB ? n - j : NB;
    real* Ajj = d_A + (int64_t)j * lda + j;
    hipLaunchKernelGGL((k_potrf_diag<real>), dim3(1), dim3(256), 0, s, nb, Ajj, lda, d_info);
    const int rows = n - j - nb;
    if (rows > 0) {
      real* A21 = d_A + (int64_t)(j + nb) * lda + j;
      hipLaunchKernelGGL((k_trsm_panel<real>), dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, s, rows, nb, (const real*)Ajj, lda, A21, lda);
      real* A22 = d_A + (int64_t)(j + nb) * lda + (j + nb);
      int rc = launch_gemm<real>(0, 1, rows, rows, nb, (real)-1, A21, lda, A21, lda, (real)1, A22, lda, s);   // A22 -= L21 L21^T
      if (rc) return rc;
    }
  }
  // zero the strict upper triangle (the trailing updates wrote it)
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}

template <typename real>
__global__ __launch_bounds__(256) void k_zero_upper(int n, real* __restrict__ A, int lda) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)n * n) return;
  const int i = (int)(e / n), j = (int)(e % n);
  if (j > i) A[(int64_t)i * lda + j] = (real)0;
}

// --------------------------------------------------------------- TRSM ---
// Diagonal-block solves for nrhs columns: one thread per right-hand side.
//   trans == 0:  L11 X = B   (forward) ; trans == 1:  L11^T X = B  (backward)
template <typename real>
__global__ __launch_bounds__(64) void k_trsm_diag(int nb, int nrhs, int trans, const real* __restrict__ L11, int ldl, real* __restrict__ Bm, int ldb) {
  __shared__ real sL[NB][NB + 1];
  __shared__ real sX[NB][64 + 1];
  const int tid = threadIdx.x;
  for (int e = tid; e < nb * nb; e += 64) sL[e / nb][e % nb] = L11[(int64_t)(e / nb) * ldl + (e % nb)];
  const int c = blockIdx.x * 64 + tid;
  __syncthreads();
  if (c < nrhs) {
    for (int j = 0; j < nb; ++j) sX[j][tid] = Bm[(int64_t)j * ldb + c];
    if (!trans) {
      for (int j = 0; j < nb; ++j) {
        real sacc = sX[j][tid];
        for (int k = 0; k < j; ++k) sacc -= sL[j][k] * sX[k][tid];
        sX[j][tid] = sacc / sL[j][j];
      }
    } else {
      for (int j = nb - 1; j >= 0; --j) {
        real sacc = sX[j][tid];
        for (int k = j + 1; k < nb; ++k) sacc -= sL[k][j] * sX[k][tid];
        sX[j][tid] = sacc / sL[j][j];
      }
    }
    for (int j = 0; j < nb; ++j) Bm[(int64_t)j * ldb + c] = sX[j][tid];
  }
}

// Solve L X = B (trans = 0) or L^T X = B (trans = 1) in place; L lower n x n, B n x nrhs.
template <typename real>
static int trsm_impl(int trans, int n, int nrhs, const real* d_L, int ldl, real* d_B, int ldb, hipStream_t s) {
  if (n < 1 || nrhs < 1 || !d_L || !d_B || ldl < n || ldb < nrhs) return WISKI_E_BADARG;
  const int nblk = (n + NB - 1) / NB;
  for (int bi = 0; bi < nblk; ++bi) {
    const int I = trans ? nblk - 1 - bi : bi;
    const int i0 = I * NB;
    const int nb = n - i0 < NB ? n - i0 : NB;
    real* BI = d_B + (int64_t)i0 * ldb;
    if (!trans && i0 > 0) {
      // B_I -= L[I, 0:i0] X[0:i0]
      int rc = launch_gemm<real>(0, 0, nb, nrhs, i0, (real)-1, d_L + (int64_t)i0 * ldl, ldl, d_B, ldb, (real)1, BI, ldb, s);
      if (rc) return rc;
    } else if (trans && i0 + nb < n) {
      // B_I -= L[i0+nb:, I]^T X[i0+nb:]
      const int rest = n - i0 - nb;
      int rc = launch_gemm<real>(1, 0, nb, nrhs, rest, (real)-1, d_L + (int64_t)(i0 + nb) * ldl + i0, ldl, d_B + (int64_t)(i0 + nb) * ldb, ldb,
                                 (real)1, BI, ldb, s);
      if (rc) return rc;
    }
    hipLaunchKernelGGL((k_trsm_diag<real>), dim3((unsigned)((nrhs + 63) / 64)), dim3(64), 0, s, nb, nrhs, trans,
                       d_L + (int64_t)i0 * ldl + i0, ldl, BI, ldb);
  }
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}

// sum_i log(A[i,i]) into a double (logdet of a Cholesky factor = 2 * this)
template <typename real>
__global__ __launch_bounds__(256) void k_logdiag(int n, const real* __restrict__ A, int lda, double* __restrict__ out) {
  __shared__ double s_red[16];
  double acc = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) acc += log((double)A[(int64_t)i * lda + i]);
  acc = block_reduce_sum(acc, s_red);
  if (threadIdx.x == 0) unsafeAtomicAdd(out, acc);
}

template <typename real>
static int potrf_full(int n, real* d_A, int lda, int32_t* d_info, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  int rc = potrf_impl<real>(n, d_A, lda, d_info, s);
  if (rc) return rc;
  const int64_t tot = (int64_t)n * n;
  hipLaunchKernelGGL((k_zero_upper<real>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, n, d_A, lda);
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}

template <typename real>
static int logdiag_impl(int n, const real* d_A, int lda, double* d_out, void* stream) {
  if (n < 1 || !d_A || !d_out) return WISKI_E_BADARG;
  int blocks = (n + 255) / 256;
  hipLaunchKernelGGL((k_logdiag<real>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n, d_A, lda, d_out);
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}

extern "C" {
int wiski_gemm_f32(int32_t ta, int32_t tb, int32_t M, int32_t N, int32_t K, float alpha, const float* A, int32_t lda, const float* B, int32_t ldb, float beta, float* C, int32_t ldc, void* s) {
  if (!A || !B || !C || K < 0) return WISKI_E_BADARG;
  return launch_gemm<float>(ta, tb, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, (hipStream_t)s);
}
int wiski_gemm_f64(int32_t ta, int32_t tb, int32_t M, int32_t N, int32_t K, double alpha, const double* A, int32_t lda, const double* B, int32_t ldb, double beta, double* C, int32_t ldc, void* s) {
  if (!A || !B || !C || K < 0) return WISKI_E_BADARG;
  return launch_gemm<double>(ta, tb, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, (hipStream_t)s);
}
int wiski_potrf_f32(int32_t n, float* A, int32_t lda, int32_t* info, void* s) { return potrf_full<float>(n, A, lda, info, s); }
int wiski_potrf_f64(int32_t n, double* A, int32_t lda, int32_t* info, void* s) { return potrf_full<double>(n, A, lda, info, s); }
int wiski_trsm_f32(int32_t trans, int32_t n, int32_t nrhs, const float* L, int32_t ldl, float* B, int32_t ldb, void* s) { return trsm_impl<float>(trans, n, nrhs, L, ldl, B, ldb, (hipStream_t)s); }
int wiski_trsm_f64(int32_t trans, int32_t n, int32_t nrhs, const double* L, int32_t ldl, double* B, int32_t ldb, void* s) { return trsm_impl<double>(trans, n, nrhs, L, ldl, B, ldb, (hipStream_t)s); }
int wiski_logdiag_f32(int32_t n, const float* A, int32_t lda, double* out, void* s) { return logdiag_impl<float>(n, A, lda, out, s); }
int wiski_logdiag_f64(int32_t n, const double* A, int32_t lda, double* out, void* s) { return logdiag_impl<double>(n, A, lda, out, s); }
}
