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

#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <vector>

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

// Large products (round 3): 128 x 128 block tile, 4 waves each computing a 64 x 64 sub-tile as 4 x 4 MFMA tiles (16 MFMAs per
// 8 LDS operand reads), K tile BK = 32 (fp32) / 16 (fp64) -- 64 MFMAs = 2048 / 4096 matrix-core cycles per wave and tile --
// with the NEXT K tile's global loads in flight in registers while the current one is in the matrix cores (one LDS buffer,
// register prefetch): the 64 x 64 kernel above stages, waits, computes 16 MFMAs and waits again.  LDS rows are k-major with
// a stride of 144 reals (= 16 mod 32 banks): the 16-lane groups of an operand read fall on disjoint banks.
// Global loads are 16-byte vectors where a tile is interior and the leading dimension allows it, scalar and predicated at the
// edges.  Used from 128 x 128 outputs with >= 2 * (number of CUs) ... see launch_gemm.
constexpr int G2M = 128, G2N = 128;
// (round 5) the same kernel on a 64 x 64 tile (TM = 64, 8 waves of 32 x 16): for outputs of ~150..500 such tiles -- the r x r products of the
// spectral factor at rank 700..1000, the trailing updates of the two-level Cholesky -- where the 128-tile grid leaves most CUs idle and the
// un-pipelined 64 x 64 / 32 x 32 kernels wait for every K tile; LDS row stride TM + 16 (= 16 mod 32 banks for both tile sizes)
template <typename real> struct G2K { static constexpr int value = sizeof(real) == 4 ? 32 : 16; };

template <typename real, bool TA, bool TB, int NW, int TM = 128>
__global__ __launch_bounds__(64 * NW) void k_gemm128(int M, int N, int K, real alpha, const real* __restrict__ A, int lda, const real* __restrict__ B,
                                                 int ldb, real beta, real* __restrict__ C, int ldc) {
  constexpr int NT = 64 * NW;                             // 4 waves (64 x 64 each) or 8 waves (64 x 32 each: twice the waves per CU when the grid is one block per CU)
  constexpr int BK = G2K<real>::value, EPT = BK * TM / NT;   // elements per thread and operand tile
  constexpr int TPL = NT / TM;                            // threads per row of an operand tile stored along k
  // LDS row stride.  fp32: TM + 16 (= 16 mod 32 banks: the two 16-lane rows of an operand read fall on disjoint banks).  fp64: a read phase is ONE
  // row of 16 doubles (all 32 banks, any stride); what the stride decides there is the k-major re-tiling of an operand that is contiguous along k
  // (A in A B, B in A B^T): 8 lanes write the same column of rows kb, kb + 2, ...: TM + 16 doubles = 0 mod 32 banks put all of them on one bank
  // pair (8-way conflict), TM + 2 spreads them over all (round 5: A B at n = 1000 62 -> see profiles/r05_gemm_mid.txt)
  constexpr int G2LD = sizeof(real) == 8 ? TM + 2 : TM + 16;
  constexpr int AT = TM / 2 / 16;                         // 16-row MFMA tiles per wave along M (2 wave rows)
  constexpr int BT = TM / (NW / 2) / 16;                  // 16-column MFMA tiles per wave along N (NW / 2 wave columns)
  constexpr int VW = 16 / (int)sizeof(real);              // reals per 16-byte vector
  __shared__ real sA2[2][BK][G2LD];                       // two stages: the next tile is written while this one is read (one barrier per K tile)
  __shared__ real sB2[2][BK][G2LD];
  using acc_t = typename Acc4<real>::type;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wr = NW == 4 ? w >> 1 : w >> 2, wc = NW == 4 ? w & 1 : w & 3;
  const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TM;
  acc_t acc[AT][BT];
#pragma unroll
  for (int a = 0; a < AT; ++a)
#pragma unroll
    for (int b = 0; b < BT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[a][b][r] = (real)0;

  // operand tile X (128 along `long`, BK along k) from a row-major matrix P:
  //   ALONG_K = true : P[(l0 + l) * ld + (k0 + k)]   (contiguous along k): thread -> l = tid >> 1, k in [(tid & 1) * EPT, + EPT)
  //   ALONG_K = false: P[(k0 + k) * ld + (l0 + l)]   (contiguous along l): thread -> k = tid / (256 / BK), l in [(tid % (256 / BK)) * EPT, + EPT)
  auto fetch = [&](auto along_k_tag, const real* __restrict__ P, int ld, int l0, int L, int k0, real (&reg)[EPT]) {
    constexpr bool ALONG_K = decltype(along_k_tag)::value;
    if constexpr (ALONG_K) {
      const int l = tid / TPL, kb = (tid % TPL) * EPT;
      const int gl = l0 + l, gk = k0 + kb;
      const real* __restrict__ src = P + (int64_t)gl * ld + gk;
      if (gl < L && gk + EPT <= K && (ld % VW) == 0 && ((uintptr_t)P % 16) == 0) {
#pragma unroll
        for (int u = 0; u < EPT; u += VW) {
          if constexpr (VW == 4) { const float4 v = *reinterpret_cast<const float4*>(src + u); reg[u] = v.x; reg[u + 1] = v.y; reg[u + 2] = v.z; reg[u + 3] = v.w; }
          else { const double2 v = *reinterpret_cast<const double2*>(src + u); reg[u] = v.x; reg[u + 1] = v.y; }
        }
      } else {
#pragma unroll
        for (int u = 0; u < EPT; ++u) reg[u] = (gl < L && gk + u < K) ? src[u] : (real)0;
      }
    } else {
      constexpr int TPR = NT / BK;
      const int k = tid / TPR, lb = (tid % TPR) * EPT;
      const int gk = k0 + k, gl = l0 + lb;
      const real* __restrict__ src = P + (int64_t)gk * ld + gl;
      if (gk < K && gl + EPT <= L && (ld % VW) == 0 && ((uintptr_t)P % 16) == 0) {
#pragma unroll
        for (int u = 0; u < EPT; u += VW) {
          if constexpr (VW == 4) { const float4 v = *reinterpret_cast<const float4*>(src + u); reg[u] = v.x; reg[u + 1] = v.y; reg[u + 2] = v.z; reg[u + 3] = v.w; }
          else { const double2 v = *reinterpret_cast<const double2*>(src + u); reg[u] = v.x; reg[u + 1] = v.y; }
        }
      } else {
#pragma unroll
        for (int u = 0; u < EPT; ++u) reg[u] = (gk < K && gl + u < L) ? src[u] : (real)0;
      }
    }
  };
  auto stash = [&](auto along_k_tag, real (*S)[G2LD], const real (&reg)[EPT]) {
    constexpr bool ALONG_K = decltype(along_k_tag)::value;
    if constexpr (ALONG_K) {
      const int l = tid / TPL, kb = (tid % TPL) * EPT;
#pragma unroll
      for (int u = 0; u < EPT; ++u) S[kb + u][l] = reg[u];
    } else {
      constexpr int TPR = NT / BK;
      const int k = tid / TPR, lb = (tid % TPR) * EPT;
#pragma unroll
      for (int u = 0; u < EPT; ++u) S[k][lb + u] = reg[u];
    }
  };
  using AK = std::integral_constant<bool, !TA>;           // op(A)[i][k]: A row-major [M][K] is contiguous along k unless transposed
  using BKt = std::integral_constant<bool, TB>;           // op(B)[k][j]: B row-major [K][N] is contiguous along j unless transposed
  real ra[EPT], rb[EPT];
  fetch(AK{}, A, lda, m0, M, 0, ra);
  fetch(BKt{}, B, ldb, n0, N, 0, rb);
  stash(AK{}, sA2[0], ra);
  stash(BKt{}, sB2[0], rb);
  __syncthreads();
  int cur = 0;
  for (int k0 = 0; k0 < K; k0 += BK) {
    const bool more = k0 + BK < K;
    if (more) {                                           // the next tile's loads fly while this one is in the matrix cores
      fetch(AK{}, A, lda, m0, M, k0 + BK, ra);
      fetch(BKt{}, B, ldb, n0, N, k0 + BK, rb);
    }
    real(*sA)[G2LD] = sA2[cur];
    real(*sB)[G2LD] = sB2[cur];
#pragma unroll
    for (int ks = 0; ks < BK; ks += 4) {
      real af[AT], bf[BT];
#pragma unroll
      for (int a = 0; a < AT; ++a) af[a] = sA[ks + (lane >> 4)][wr * (TM / 2) + a * 16 + (lane & 15)];
#pragma unroll
      for (int b = 0; b < BT; ++b) bf[b] = sB[ks + (lane >> 4)][wc * (16 * BT) + b * 16 + (lane & 15)];
#pragma unroll
      for (int a = 0; a < AT; ++a)
#pragma unroll
        for (int b = 0; b < BT; ++b) acc[a][b] = mfma16(af[a], bf[b], acc[a][b]);
    }
    if (more) {                                           // into the other stage: its readers passed the barrier of the previous tile
      stash(AK{}, sA2[cur ^ 1], ra);
      stash(BKt{}, sB2[cur ^ 1], rb);
    }
    __syncthreads();
    cur ^= 1;
  }
#pragma unroll
  for (int a = 0; a < AT; ++a)
#pragma unroll
    for (int b = 0; b < BT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gi = m0 + wr * (TM / 2) + a * 16 + frag_row<real>(lane, r);
        const int gj = n0 + wc * (16 * BT) + b * 16 + (lane & 15);
        if (gi < M && gj < N) {
          const int64_t e = (int64_t)gi * ldc + gj;
          const real v = alpha * acc[a][b][r];
          C[e] = beta == (real)0 ? v : v + beta * C[e];
        }
      }
}

static int gemm128_min_blocks() { return 32; }

// 32 x 32 tiles, BK = 32, register prefetch: for products whose 64 x 64 tiling leaves most of the chip idle (the spectral factor's
// T^T G_ref T: 540 x 327 outputs are 54 big tiles on 256 CUs -- and 187 small ones).  4 waves, one 16 x 16 MFMA tile each.
template <typename real, bool TA, bool TB>
__global__ __launch_bounds__(256) void k_gemm32(int M, int N, int K, real alpha, const real* __restrict__ A, int lda, const real* __restrict__ B, int ldb,
                                                real beta, real* __restrict__ C, int ldc) {
  __shared__ real sA[32][33];        // op(A) tile as sA[i][k]
  __shared__ real sB[32][36];        // op(B) tile as sB[k][j]
  using acc_t = typename Acc4<real>::type;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int qa = w >> 1, qb = w & 1;
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  const int r8 = tid >> 3, c4 = (tid & 7) * 4;
  acc_t acc;
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = (real)0;
  real pa[4], pb[4];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if constexpr (!TA) { const int gi = m0 + r8, gk = k0 + c4 + u; pa[u] = (gi < M && gk < K) ? A[(int64_t)gi * lda + gk] : (real)0; }
      else { const int gk = k0 + r8, gi = m0 + c4 + u; pa[u] = (gi < M && gk < K) ? A[(int64_t)gk * lda + gi] : (real)0; }
      if constexpr (!TB) { const int gk = k0 + r8, gj = n0 + c4 + u; pb[u] = (gk < K && gj < N) ? B[(int64_t)gk * ldb + gj] : (real)0; }
      else { const int gj = n0 + r8, gk = k0 + c4 + u; pb[u] = (gk < K && gj < N) ? B[(int64_t)gj * ldb + gk] : (real)0; }
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < K; k0 += 32) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if constexpr (!TA) sA[r8][c4 + u] = pa[u]; else sA[c4 + u][r8] = pa[u];
      if constexpr (!TB) sB[r8][c4 + u] = pb[u]; else sB[c4 + u][r8] = pb[u];
    }
    __syncthreads();
    if (k0 + 32 < K) fetch(k0 + 32);                // next tile's loads fly under this tile's MFMAs
#pragma unroll
    for (int ks = 0; ks < 32; ks += 4) {
      const int kk = ks + (lane >> 4);
      acc = mfma16(sA[qa * 16 + (lane & 15)][kk], sB[kk][qb * 16 + (lane & 15)], acc);
    }
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int gi = m0 + qa * 16 + frag_row<real>(lane, r), gj = n0 + qb * 16 + (lane & 15);
    if (gi < M && gj < N) {
      const int64_t e = (int64_t)gi * ldc + gj;
      const real v = alpha * acc[r];
      C[e] = beta == (real)0 ? v : v + beta * C[e];
    }
  }
}

template <typename real>
static int launch_gemm(int ta, int tb, int M, int N, int K, real alpha, const real* A, int lda, const real* B, int ldb, real beta, real* C,
                       int ldc, hipStream_t s) {
  if (M <= 0 || N <= 0) return WISKI_OK;
  {
    // Small tiles while the 64 x 64 grid is about one tile per CU or less (and there is enough K to matter).  Round 4 (tools/gemm_mid_probe.py,
    // profiles/r04_gemm_mid.txt): a 64 x 64 tile is ONE 4-wave workgroup, so up to ~256 tiles a CU holds one wave per SIMD and nothing hides
    // the LDS / global latency of its K loop; the 32 x 32 kernel puts 4 workgroups on a CU for the same output.  fp64 n = 800 / 1000 / 1200 /
    // 1400: 92 / 116 / 157 / 198 us -> 38 / 62 / 105 / 178 us (27..33 TF instead of 11..28); fp32: 65 / 84 / 101 / 117 -> 34 / 50 / 103 / 131 us.
    // Hence below 500 (fp64) / 300 (fp32) tiles of 64 x 64; it was 128 for both.  WISKI_GEMM32_MAX_TILES overrides.
    const int64_t nb64 = (int64_t)((N + GBN - 1) / GBN) * ((M + GBM - 1) / GBM);
    {
      // (round 5) the pipelined kernel on a 64 x 64 tile with 8 waves, between WISKI_GEMM64P_MIN and _MAX output tiles
      // where it wins (tools/gemm_mid_probe.py, profiles/r05_gemm_mid.txt; n = side of a square product):
      //   fp32  n = 800 .. 1600: 35 / 50 / 98 / 116 / 132 us -> 29 / 37 / 65 / 75 / 109 us (n = 1000: 40 -> 54 TF, TN 62 TF)   => 110 .. 800 tiles
      //   fp64  n = 1000 / 1400 / 1600: 61 / 178 / 249 us -> 57 / 124 / 189 us (A^T B 53 / 116 / 175, A B^T 71 -> 60 at 1000)   => 240 .. 900 tiles
      const int p_min = sizeof(real) == 4 ? 110 : 240;
      const int p_max = sizeof(real) == 4 ? 800 : 900;
      if (nb64 >= p_min && nb64 <= p_max && K >= 128) {
        dim3 g6((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64));
#define G64P(NWV)                                                                                                                                      \
  do {                                                                                                                                                 \
    if (!ta && !tb) hipLaunchKernelGGL((k_gemm128<real, false, false, NWV, 64>), g6, dim3(64 * NWV), 0, s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc);      \
    else if (ta && !tb) hipLaunchKernelGGL((k_gemm128<real, true, false, NWV, 64>), g6, dim3(64 * NWV), 0, s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc);   \
    else if (!ta && tb) hipLaunchKernelGGL((k_gemm128<real, false, true, NWV, 64>), g6, dim3(64 * NWV), 0, s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc);   \
    else hipLaunchKernelGGL((k_gemm128<real, true, true, NWV, 64>), g6, dim3(64 * NWV), 0, s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc);                   \
  } while (0)
        G64P(8);
#undef G64P
        return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
      }
    }
    constexpr bool small_on = true;
    static const int small_max_env = [] { const char* e = getenv("WISKI_GEMM32_MAX_TILES"); return e ? atoi(e) : 0; }();
    const int small_max = small_max_env > 0 ? small_max_env : (sizeof(real) == 8 ? 500 : 300);
    if (small_on && nb64 < small_max && K >= 64 && (int64_t)M * N >= 32 * 32 * 8) {
      dim3 g3((unsigned)((N + 31) / 32), (unsigned)((M + 31) / 32));
      if (!ta && !tb) hipLaunchKernelGGL((k_gemm32<real, false, false>), g3, dim3(256), 0, s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc);
      else if (ta && !tb) hipLaunchKernelGGL((k_gemm32<real, true, false>), g3, dim3(256), 0, s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc);
      else if (!ta && tb) hipLaunchKernelGGL((k_gemm32<real, false, true>), g3, dim3(256), 0, s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc);
      else hipLaunchKernelGGL((k_gemm32<real, true, true>), g3, dim3(256), 0, s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc);
      return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
    }
  }
  {
    // the 128 x 128 kernel once its grid covers at least half the chip (n >= ~1500 square): below that the 64 x 64 tiles fill more CUs
    const int64_t nb = (int64_t)((N + G2N - 1) / G2N) * ((M + G2M - 1) / G2M);
    const int minb = gemm128_min_blocks();
    // (K >= 256: the rank-64 trailing updates of wiski_potrf / wiski_trsm are 4 K tiles of fp64 -- prologue-bound here, faster on the
    // small kernel; fp64 needs twice the grid before the large tile wins: 24 vs 27 TF at n = 1536, 42 vs 39 at 2048)
    // (round 3, 8-wave form: fp32 wins from 32 blocks -- 24 vs 21 TF at n = 1024 --, fp64 from 128 -- 31 vs 27 TF at n = 1536, but 14 vs 18 at 1024)
    if (minb > 0 && nb >= (sizeof(real) == 8 ? 4 * minb : minb) && K >= 256) {
      dim3 g2((unsigned)((N + G2N - 1) / G2N), (unsigned)((M + G2M - 1) / G2M));
      // 8 waves per 128 x 128 tile (64 x 32 per wave) beat 4 at every size measured: twice the waves per CU hide the LDS and
      // barrier latency of the K loop (fp32 n = 2048: 89 -> 100 TF, fp64 n = 4096: 45 -> 61 TF)
      {
        if (!ta && !tb) hipLaunchKernelGGL((k_gemm128<real, false, false, 8>), g2, dim3(512), 0, s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc);
        else if (ta && !tb) hipLaunchKernelGGL((k_gemm128<real, true, false, 8>), g2, dim3(512), 0, s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc);
        else if (!ta && tb) hipLaunchKernelGGL((k_gemm128<real, false, true, 8>), g2, dim3(512), 0, s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc);
        else hipLaunchKernelGGL((k_gemm128<real, true, true, 8>), g2, dim3(512), 0, s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc);
        return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
      }
    }
  }
  dim3 grd((unsigned)((N + GBN - 1) / GBN), (unsigned)((M + GBM - 1) / GBM));
  if (!ta && !tb) hipLaunchKernelGGL((k_gemm<real, false, false>), grd, dim3(256), 0, s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc);
  else if (ta && !tb) hipLaunchKernelGGL((k_gemm<real, true, false>), grd, dim3(256), 0, s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc);
  else if (!ta && tb) hipLaunchKernelGGL((k_gemm<real, false, true>), grd, dim3(256), 0, s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc);
  else hipLaunchKernelGGL((k_gemm<real, true, true>), grd, dim3(256), 0, s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc);
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}

// ------------------------------------------------------------ Cholesky ---
// Blocked right-looking lower Cholesky, NB = 64.  The O(n^3) work is the trailing update on the MFMA GEMM; at the
// sizes of the dense regime (n <= 2048) the run time is the latency of the 2 n / NB small steps in between, so those
// are kept short: the diagonal block is factorised AND inverted by one workgroup in LDS, and every triangular solve
// against a diagonal block -- the panel of the factorisation, both sweeps of wiski_trsm -- is a tile product with
// that explicit 64 x 64 inverse instead of a per-row / per-column substitution.
//   k_potrf_diag   : A11 -> L11 (in place) and L11^-1 (scratch), one wave, rows in registers
//   k_apply_inv    : 64-row or 64-column tiles  T <- T Linv^T (right) | Linv T | Linv^T T (left), in place
constexpr int NB = 64;

// Column c of the inverse of the lower-triangular 64 x 64 matrix in sM (LDS), by forward substitution on e_c with the
// whole column in registers (compile-time indices, fully unrolled: 2016 FMAs fed by broadcast LDS reads).
// RECIP: the diagonal of sM holds the RECIPROCALS of the factor's diagonal (an fp64 division is ~20 dependent instructions and
// this chain has 64 of them; a separate reciprocal array reached through a pointer parameter made the compiler fall back to
// flat loads -- 3x slower than the divisions it was meant to remove).
template <typename real, bool RECIP = false>
__device__ __forceinline__ void tri_inv_column(const real (*sM)[NB + 1], int c, real x[NB]) {
#pragma unroll
  for (int i = 0; i < NB; ++i) x[i] = i == c ? (real)1 : (real)0;
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    const real xk = RECIP ? x[k] * sM[k][k] : x[k] / sM[k][k];          // rows k < c: x[k] == 0 stays 0
    x[k] = xk;
#pragma unroll
    for (int i = k + 1; i < NB; ++i) x[i] -= sM[i][k] * xk;
  }
}

// One wave factorises the diagonal block: lane t owns row t in registers; per step the pivot comes by a lane read and
// column k is published through LDS for broadcast reads (no block barrier chain, no integer division).  The inverse of
// the factor follows from the same wave (tri_inv_column).  nb < 64 is padded with an identity block.
template <typename real>
__global__ __launch_bounds__(64) void k_potrf_diag(int nb, real* __restrict__ A, int lda, real* __restrict__ Linv, int32_t* __restrict__ info) {
  __shared__ real sM[NB][NB + 1];
  __shared__ real sCol[2][NB];
  const int t = threadIdx.x;
  real myri = (real)1;
  for (int r = 0; r < NB; ++r) sM[r][t] = (r < nb && t < nb) ? A[(int64_t)r * lda + t] : (r == t ? (real)1 : (real)0);   // coalesced rows
  __syncthreads();
  real row[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) row[j] = sM[t][j];
  bool bad = false;
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    // pivot by a lane read; 1 / sqrt(pivot) once (rsqrt + one Newton step: full precision without the division and square-root
    // expansions, which dominated this 64-step dependent chain in fp64), the column published through a 2-deep LDS slot (one
    // barrier per step: the slot written two steps ago has no readers left)
    const real piv0 = __shfl(row[k], k, 64);
    if (!(piv0 > (real)0)) bad = true;
    const real piv = piv0 > (real)0 ? piv0 : (real)1;
    real ri;
    if constexpr (sizeof(real) == 4) {
      ri = __builtin_amdgcn_rsqf(piv);                     // v_rsq_f32 (1 ulp) + one Newton step
      ri = ri * (1.5f - 0.5f * piv * ri * ri);
    } else {
      ri = __builtin_amdgcn_rsq(piv);                      // v_rsq_f64 (~26 bits) + two Newton steps
      ri = ri * (1.5 - 0.5 * piv * ri * ri);
      ri = ri * (1.5 - 0.5 * piv * ri * ri);
    }
    const real lk = t > k ? row[k] * ri : (t == k ? piv * ri : (real)0);
    row[k] = lk;
    if (t == k) myri = ri;
    sCol[k & 1][t] = lk;
    __syncthreads();
    if (t > k) {
#pragma unroll
      for (int j = k + 1; j < NB; ++j) row[j] -= lk * sCol[k & 1][j];      // entries j > t are never used (upper triangle)
    }
  }
  if (bad && t == 0) atomicOr(info, 1);
  __syncthreads();
#pragma unroll
  for (int j = 0; j < NB; ++j) sM[t][j] = j <= t ? row[j] : (real)0;
  __syncthreads();
  for (int r = 0; r < nb; ++r)
    if (t < nb) A[(int64_t)r * lda + t] = sM[r][t];
  __syncthreads();
  sM[t][t] = myri;                                                   // the inversion below multiplies by the reciprocal diagonal
  __syncthreads();
  real x[NB];
  tri_inv_column<real, true>(sM, t, x);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NB; ++i) sM[i][t] = x[i];                      // column t of the inverse
  __syncthreads();
  for (int r = 0; r < NB; ++r) Linv[r * NB + t] = sM[r][t];          // [64][64] dense, identity-padded
}

// Inverses of all diagonal blocks of an existing factor (wiski_trsm): block b of the grid handles L[b*64.., b*64..].
template <typename real>
__global__ __launch_bounds__(64) void k_tri_inv_blocks(int n, const real* __restrict__ L, int ldl, real* __restrict__ Linv) {
  __shared__ real sM[NB][NB + 1];
  const int b = blockIdx.x, i0 = b * NB, t = threadIdx.x;
  const int nb = n - i0 < NB ? n - i0 : NB;
  for (int r = 0; r < NB; ++r) sM[r][t] = (r < nb && t < nb && t <= r) ? L[(int64_t)(i0 + r) * ldl + i0 + t] : (r == t ? (real)1 : (real)0);
  __syncthreads();
  sM[t][t] = (real)1 / sM[t][t];                           // one division per row instead of one per substitution step
  __syncthreads();
  real x[NB];
  tri_inv_column<real, true>(sM, t, x);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NB; ++i) sM[i][t] = x[i];
  __syncthreads();
  real* out = Linv + (int64_t)b * NB * NB;
  for (int r = 0; r < NB; ++r) out[r * NB + t] = sM[r][t];
}

// In-place tile product with a 64 x 64 inverse block Linv (row-major, lower triangular, identity-padded):
//   MODE 0 (right):  T [rows x nb]  <- T Linv^T          (panel of the factorisation: X L11^T = A21)
//   MODE 1 (left) :  T [nb x cols]  <- Linv T            (forward sweep:  L11 X = B)
//   MODE 2 (left) :  T [nb x cols]  <- Linv^T T          (backward sweep: L11^T X = B)
// One block per 64 rows (MODE 0) or 64 columns (MODE 1, 2); 256 threads, 16 outputs each.
template <typename real, int MODE>
__global__ __launch_bounds__(256) void k_apply_inv(int ext, int nb, const real* __restrict__ Linv, real* __restrict__ T, int ldt) {
  // 64 x 64 x 64 tile product on the matrix cores (round 3; the scalar-FMA version it replaces spent 29 us per fp64 tile on one
  // LDS read per FMA): 4 waves, each a 32 x 32 quadrant as 2 x 2 MFMA tiles over 16 k-steps.
  __shared__ real sI[NB][NB + 1];
  __shared__ real sT[NB][NB + 1];
  using acc_t = typename Acc4<real>::type;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wr = w >> 1, wc = w & 1;
  const int o0 = blockIdx.x * NB;                        // first row (MODE 0) / column (MODE 1, 2) of this tile
  for (int e = tid; e < NB * NB; e += 256) {
    const int i = e >> 6, j = e & 63;
    sI[i][j] = Linv[e];
    real v = (real)0;
    if (MODE == 0) { if (o0 + i < ext && j < nb) v = T[(int64_t)(o0 + i) * ldt + j]; }
    else { if (i < nb && o0 + j < ext) v = T[(int64_t)i * ldt + o0 + j]; }
    sT[i][j] = v;
  }
  __syncthreads();
  acc_t acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[a][b][r] = (real)0;
#pragma unroll
  for (int ks = 0; ks < NB; ks += 4) {
    const int kk = ks + (lane >> 4);
    real af[2], bf[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int i = wr * 32 + a * 16 + (lane & 15);
      af[a] = MODE == 0 ? sT[i][kk] : (MODE == 1 ? sI[i][kk] : sI[kk][i]);      // A operand (i, k)
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int j = wc * 32 + b * 16 + (lane & 15);
      bf[b] = MODE == 0 ? sI[j][kk] : sT[kk][j];                                   // B operand (k, j)
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = mfma16(af[a], bf[b], acc[a][b]);
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = wr * 32 + a * 16 + frag_row<real>(lane, r);
        const int j = wc * 32 + b * 16 + (lane & 15);
        if (MODE == 0) { if (o0 + i < ext && j < nb) T[(int64_t)(o0 + i) * ldt + j] = acc[a][b][r]; }
        else { if (i < nb && o0 + j < ext) T[(int64_t)i * ldt + o0 + j] = acc[a][b][r]; }
      }
}

#include "dense_small.h"

template <typename real>
static int potrf_impl(int n, real* d_A, int lda, int32_t* d_info, hipStream_t s) {
  if (n < 1 || !d_A || !d_info || lda < n) return WISKI_E_BADARG;
  real* linv = nullptr;
  if (hipMallocAsync((void**)&linv, (size_t)NB * NB * sizeof(real), s) != hipSuccess) return WISKI_E_LAUNCH;   // stream-ordered scratch
  int rc = WISKI_OK;
  for (int j = 0; j < n && rc == WISKI_OK; j += NB) {
    const int nb = n - j < NB ? n - j : NB;
    real* Ajj = d_A + (int64_t)j * lda + j;
    hipLaunchKernelGGL((k_potrf_diag<real>), dim3(1), dim3(64), 0, s, nb, Ajj, lda, linv, d_info);
    const int rows = n - j - nb;
    if (rows > 0) {
      real* A21 = d_A + (int64_t)(j + nb) * lda + j;
      hipLaunchKernelGGL((k_apply_inv<real, 0>), dim3((unsigned)((rows + NB - 1) / NB)), dim3(256), 0, s, rows, nb, (const real*)linv, A21, lda);
      real* A22 = d_A + (int64_t)(j + nb) * lda + (j + nb);
      rc = launch_gemm<real>(0, 1, rows, rows, nb, (real)-1, A21, lda, A21, lda, (real)1, A22, lda, s);   // A22 -= L21 L21^T
    }
  }
  (void)hipFreeAsync(linv, s);
  if (rc) return rc;
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
// Solve L X = B (trans = 0) or L^T X = B (trans = 1) in place; L lower n x n, B n x nrhs.  Right-looking: block row I
// is finished with the explicit inverse of its diagonal block (k_apply_inv), then ALL remaining block rows are updated
// at once -- a GEMM with a long M and K = 64 that fills the chip, instead of one 64-row GEMM with a long K per step.
template <typename real>
static int trsm_impl(int trans, int n, int nrhs, const real* d_L, int ldl, real* d_B, int ldb, hipStream_t s) {
  if (n < 1 || nrhs < 1 || !d_L || !d_B || ldl < n || ldb < nrhs) return WISKI_E_BADARG;
  const int nblk = (n + NB - 1) / NB;
  real* linv = nullptr;
  if (hipMallocAsync((void**)&linv, (size_t)nblk * NB * NB * sizeof(real), s) != hipSuccess) return WISKI_E_LAUNCH;
  hipLaunchKernelGGL((k_tri_inv_blocks<real>), dim3((unsigned)nblk), dim3(64), 0, s, n, d_L, ldl, linv);
  const unsigned ctiles = (unsigned)((nrhs + NB - 1) / NB);
  int rc = WISKI_OK;
  for (int bi = 0; bi < nblk && rc == WISKI_OK; ++bi) {
    const int I = trans ? nblk - 1 - bi : bi;
    const int i0 = I * NB;
    const int nb = n - i0 < NB ? n - i0 : NB;
    real* BI = d_B + (int64_t)i0 * ldb;
    const real* inv = linv + (int64_t)I * NB * NB;
    if (!trans) {
      hipLaunchKernelGGL((k_apply_inv<real, 1>), dim3(ctiles), dim3(256), 0, s, nrhs, nb, inv, BI, ldb);
      const int below = n - i0 - nb;
      if (below > 0)     // B[i0+nb:, :] -= L[i0+nb:, I] X_I
        rc = launch_gemm<real>(0, 0, below, nrhs, nb, (real)-1, d_L + (int64_t)(i0 + nb) * ldl + i0, ldl, BI, ldb, (real)1,
                               d_B + (int64_t)(i0 + nb) * ldb, ldb, s);
    } else {
      hipLaunchKernelGGL((k_apply_inv<real, 2>), dim3(ctiles), dim3(256), 0, s, nrhs, nb, inv, BI, ldb);
      if (i0 > 0)        // B[0:i0, :] -= L[I, 0:i0]^T X_I
        rc = launch_gemm<real>(1, 0, i0, nrhs, nb, (real)-1, d_L + (int64_t)i0 * ldl, ldl, BI, ldb, (real)1, d_B, ldb, s);
    }
  }
  (void)hipFreeAsync(linv, s);
  if (rc) return rc;
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

static bool small_path_enabled() { return true; }

// n <= 480: the one-workgroup factorisation of dense_small.h (and, with d_X, the explicit inverse of the factor)
template <typename real>
static int potrf_small_entry(int n, real* d_A, int lda, real* d_X, int ldx, int32_t* d_info, hipStream_t s) {
  if (n < 1 || !d_A || !d_info || lda < n || (d_X && ldx < n)) return WISKI_E_BADARG;
  const int nblk = (n + SNB - 1) / SNB;
  real* dinv = nullptr;
  if (hipMallocAsync((void**)&dinv, (size_t)nblk * SNB * SNB * sizeof(real), s) != hipSuccess) return WISKI_E_LAUNCH;
  const int rc = potrf_small<real>(n, d_A, lda, dinv, d_X, ldx, d_info, s);
  (void)hipFreeAsync(dinv, s);
  return rc;
}

template <typename real>
__global__ __launch_bounds__(256) void k_set_identity(int n, real* __restrict__ X, int ldx) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)n * n) return;
  const int i = (int)(e / n), j = (int)(e % n);
  X[(int64_t)i * ldx + j] = i == j ? (real)1 : (real)0;
}

// Two-level blocked Cholesky for the dense regime beyond the one-launch kernels (480 < n; the model reaches n <= 2048): big blocks of
// BB <= 448 rows.  A diagonal block is factorised AND inverted by ONE launch of the cooperating-workgroups kernel (dense_small.h /
// dense_coop.h: ~150 us at 256, ~190 us at 327), the panel below it is one MFMA GEMM with that inverse (L21 = A21 L11^-T, into
// scratch), the trailing update a second one (A22 -= L21 L21^T) and the panel is copied back: 3 launches + 1 copy per big block
// instead of the 64-wide right-looking loop's 16 x (one-wave diagonal kernel 59 us + apply-inverse 13 us + GEMM 40 us) at n = 1000
// (BASELINE config 4, the BayesOpt loop on a 10^3 grid: 5 factorisations per step were 60 % of its GPU time).
// Returns WISKI_SMALL_UNAVAILABLE before touching A when the device refuses the small kernels' LDS.
static bool two_level_potrf_enabled() {
  static const bool on = [] {
    const char* e = getenv("WISKI_POTRF_TWO_LEVEL");
    return !(e && e[0] == '0');
  }();
  return on;
}
// d_X != nullptr: also the explicit inverse X = L^-1 (n x n, ldx).  The diagonal blocks' inverses are written straight into X by the
// factorisation; block row I of the rest follows from the rows above it with two GEMMs, X[I, 0:i0] = -X_II (L[I, 0:i0] X[0:i0, 0:i0]).
template <typename real>
static int potrf_two_level(int n, real* d_A, int lda, int32_t* d_info, hipStream_t s, real* d_X = nullptr, int ldx = 0) {
  if (n < 1 || !d_A || !d_info || lda < n || (d_X && ldx < n)) return WISKI_E_BADARG;
  constexpr int BBMAX = 448;
  const int nbig = (n + BBMAX - 1) / BBMAX;
  const int BB = ((n + nbig - 1) / nbig + SNB - 1) / SNB * SNB;          // equal blocks, multiple of the small kernels' 32
  const int nblk = (BB + SNB - 1) / SNB;
  const size_t nX = d_X ? 0 : (size_t)BB * BB, nP = (size_t)(n > BB ? n - BB : 1) * BB, nD = (size_t)nblk * SNB * SNB, nS = d_X ? (size_t)BB * n : 0;
  real* scr = nullptr;
  if (hipMallocAsync((void**)&scr, (nX + nP + nD + nS) * sizeof(real), s) != hipSuccess) return WISKI_E_LAUNCH;
  real *Xs = scr, *P = scr + nX, *dinv = P + nP, *S = dinv + nD;
  int rc = WISKI_OK;
  for (int j = 0; j < n && rc == WISKI_OK; j += BB) {
    const int bb = n - j < BB ? n - j : BB;
    real* Ajj = d_A + (int64_t)j * lda + j;
    real* X = d_X ? d_X + (int64_t)j * ldx + j : Xs;
    const int ldxx = d_X ? ldx : BB;
    rc = potrf_small<real>(bb, Ajj, lda, dinv, X, ldxx, d_info, s);       // L_jj in place, X = L_jj^-1 (dense, zeros above the diagonal)
    if (rc) break;                                                       // (WISKI_SMALL_UNAVAILABLE can only come from the first block)
    const int rows = n - j - bb;
    if (rows > 0) {
      real* A21 = d_A + (int64_t)(j + bb) * lda + j;
      real* A22 = d_A + (int64_t)(j + bb) * lda + (j + bb);
      rc = launch_gemm<real>(0, 1, rows, bb, bb, (real)1, A21, lda, X, ldxx, (real)0, P, BB, s);                  // L21 = A21 L_jj^-T
      if (rc == WISKI_OK) rc = launch_gemm<real>(0, 1, rows, rows, bb, (real)-1, P, BB, P, BB, (real)1, A22, lda, s);   // A22 -= L21 L21^T
      if (rc == WISKI_OK && hipMemcpy2DAsync(A21, (size_t)lda * sizeof(real), P, (size_t)BB * sizeof(real), (size_t)bb * sizeof(real), (size_t)rows,
                                             hipMemcpyDeviceToDevice, s) != hipSuccess)
        rc = WISKI_E_LAUNCH;
    }
  }
  if (d_X && rc == WISKI_OK) {
    // X above the diagonal blocks: zero, in ONE launch (the products below read X[0:i0, 0:i0] as a full matrix; the blocks below the
    // diagonal are written by them; four 2-D fills per call before round 5)
    const int64_t tot = (int64_t)n * n;
    hipLaunchKernelGGL((k_zero_upper<real>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, n, d_X, ldx);
    if (hipGetLastError() != hipSuccess) rc = WISKI_E_LAUNCH;
  }
  if (d_X && rc == WISKI_OK)
    for (int i0 = BB; i0 < n && rc == WISKI_OK; i0 += BB) {
      const int bb = n - i0 < BB ? n - i0 : BB;
      rc = launch_gemm<real>(0, 0, bb, i0, i0, (real)1, d_A + (int64_t)i0 * lda, lda, d_X, ldx, (real)0, S, n, s);                 // S = L[I, 0:i0] X[0:i0, 0:i0]
      if (rc == WISKI_OK)
        rc = launch_gemm<real>(0, 0, bb, i0, bb, (real)-1, d_X + (int64_t)i0 * ldx + i0, ldx, S, n, (real)0, d_X + (int64_t)i0 * ldx, ldx, s);   // X[I, 0:i0] = -X_II S
    }
  (void)hipFreeAsync(scr, s);
  return rc;
}

template <typename real>
static int potrf_full(int n, real* d_A, int lda, int32_t* d_info, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (n <= SMALL_N_MAX_POTRF && small_path_enabled()) {
    const int rs = potrf_small_entry<real>(n, d_A, lda, (real*)nullptr, 0, d_info, s);
    if (rs != WISKI_SMALL_UNAVAILABLE) return rs;
  }
  int rc = WISKI_SMALL_UNAVAILABLE;
  if (n > SMALL_N_MAX_POTRF && small_path_enabled() && two_level_potrf_enabled()) {
    rc = potrf_two_level<real>(n, d_A, lda, d_info, s);
    if (rc != WISKI_OK && rc != WISKI_SMALL_UNAVAILABLE) return rc;
  }
  if (rc == WISKI_SMALL_UNAVAILABLE) rc = potrf_impl<real>(n, d_A, lda, d_info, s);
  if (rc) return rc;
  const int64_t tot = (int64_t)n * n;
  hipLaunchKernelGGL((k_zero_upper<real>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, n, d_A, lda);
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}

// Cholesky factor in place AND its explicit inverse X = L^-1 (every later solve against the factor is then one GEMM / GEMV).
template <typename real>
static int potrf_inverse(int n, real* d_A, int lda, real* d_X, int ldx, int32_t* d_info, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!d_X || ldx < n) return WISKI_E_BADARG;
  if (n <= SMALL_N_MAX && small_path_enabled()) {
    const int rs = potrf_small_entry<real>(n, d_A, lda, d_X, ldx, d_info, s);
    if (rs != WISKI_SMALL_UNAVAILABLE) return rs;
  }
  if (n > SMALL_N_MAX_POTRF && small_path_enabled() && two_level_potrf_enabled()) {     // (n = 481..512: the one-launch factor + a blocked solve, below)
    int rt = potrf_two_level<real>(n, d_A, lda, d_info, s, d_X, ldx);
    if (rt == WISKI_OK) {
      const int64_t tot2 = (int64_t)n * n;
      hipLaunchKernelGGL((k_zero_upper<real>), dim3((unsigned)((tot2 + 255) / 256)), dim3(256), 0, s, n, d_A, lda);
      return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
    }
    if (rt != WISKI_SMALL_UNAVAILABLE) return rt;
  }
  int rc = potrf_full<real>(n, d_A, lda, d_info, stream);
  if (rc) return rc;
  const int64_t tot = (int64_t)n * n;
  hipLaunchKernelGGL((k_set_identity<real>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, n, d_X, ldx);
  return trsm_impl<real>(0, n, n, (const real*)d_A, lda, d_X, ldx, s);
}

template <typename real>
static int logdiag_impl(int n, const real* d_A, int lda, double* d_out, void* stream) {
  if (n < 1 || !d_A || !d_out) return WISKI_E_BADARG;
  int blocks = (n + 255) / 256;
  hipLaunchKernelGGL((k_logdiag<real>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n, d_A, lda, d_out);
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}

// ----------------------------------------------------------------- root update (a6) ---
// Rank-q update of a root / inverse-root pair, replacing UpdatedRootLazyTensor.collect_vector (URLT:69-119):
// given L, R with L L^T = A and R^T L = I (R = L^-T), and the new columns V [m, q], produce L', R' with
// L' L'^T = A + V V^T and R'^T L' = I.  The reference takes the full SVD p = R^T V = U S W^T with an r x r U and
// forms L U S~, R U S~^-1, S~ = diag(sqrt(S^2 + 1), 1...): O(m r^2).  The same Gram matrices follow from the thin
// factor U_q = p W S^-1 of the q x q eigenproblem p^T p = W S^2 W^T:
//   L' = L + (L U_q) diag(sqrt(S^2+1) - 1) U_q^T      R' = R + (R U_q) diag(1/sqrt(S^2+1) - 1) U_q^T
// i.e. L' = L (I + p p^T)^{1/2}, O(m r q) on the MFMA GEMM; L' differs from the reference's L U S~ by a right
// orthogonal factor (roots are only defined up to one).  The q x q eigenproblem is solved on the host (cyclic
// Jacobi, fp64): one small device-to-host copy and a stream synchronisation per update.
static void jacobi_eigh(int n, std::vector<double>& a, std::vector<double>& v) {   // a (row-major, symmetric) -> eigenvalues on its diagonal
  v.assign((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) v[(size_t)i * n + i] = 1.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0, diag = 0;
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) (i == j ? diag : off) += a[(size_t)i * n + j] * a[(size_t)i * n + j];
    if (off <= 1e-30 * (diag > 0 ? diag : 1.0)) break;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = a[(size_t)p * n + q];
        if (apq == 0.0) continue;
        const double theta = (a[(size_t)q * n + q] - a[(size_t)p * n + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
        for (int k = 0; k < n; ++k) {            // columns p, q
          const double akp = a[(size_t)k * n + p], akq = a[(size_t)k * n + q];
          a[(size_t)k * n + p] = c * akp - sn * akq;
          a[(size_t)k * n + q] = sn * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {            // rows p, q
          const double apk = a[(size_t)p * n + k], aqk = a[(size_t)q * n + k];
          a[(size_t)p * n + k] = c * apk - sn * aqk;
          a[(size_t)q * n + k] = sn * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k) {
          const double vkp = v[(size_t)k * n + p], vkq = v[(size_t)k * n + q];
          v[(size_t)k * n + p] = c * vkp - sn * vkq;
          v[(size_t)k * n + q] = sn * vkp + c * vkq;
        }
      }
  }
}

// The same eigenproblem on the device for small q (streaming batches): one workgroup, parallel cyclic Jacobi in fp64 with the
// round-robin ("tournament") ordering -- n / 2 disjoint rotations per step, n - 1 steps per sweep -- on the q x q Gram matrix
// in LDS, followed by the coefficient matrix W S^-1 and the two scale vectors root_update_impl needs.  Keeps
// wiski_root_update asynchronous on the caller's stream for q <= ROOT_JACOBI_MAXQ (the host Jacobi above serves larger q and
// synchronises: an O(q^3) eigensolve is cheaper on the host than in one workgroup beyond a few dozen columns).
constexpr int ROOT_JACOBI_MAXQ = 32;
template <typename real>
__global__ __launch_bounds__(256) void k_gram_eigh(int q, const real* __restrict__ gram, real* __restrict__ coef, real* __restrict__ sc) {
  constexpr int N = ROOT_JACOBI_MAXQ;
  __shared__ double a[N][N + 1], v[N][N + 1], cs[N / 2][2], red[256];
  __shared__ int pr[N / 2][2];
  const int t = threadIdx.x;
  const int n = (q + 1) & ~1;                              // even size: an odd q gets a decoupled dummy row / column
  for (int e = t; e < N * N; e += 256) {
    const int i = e / N, j = e % N;
    a[i][j] = (i < q && j < q) ? 0.5 * ((double)gram[i * q + j] + (double)gram[j * q + i]) : 0.0;
    v[i][j] = i == j ? 1.0 : 0.0;
  }
  __syncthreads();
  const int np = n / 2;
  for (int sweep = 0; sweep < 30; ++sweep) {
    // convergence: off-diagonal mass against the diagonal's
    double off = 0, dg = 0;
    for (int e = t; e < n * n; e += 256) {
      const int i = e / n, j = e % n;
      const double x = a[i][j] * a[i][j];
      if (i == j) dg += x; else off += x;
    }
    red[t] = off;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) { if (t < w) red[t] += red[t + w]; __syncthreads(); }
    off = red[0];
    __syncthreads();
    red[t] = dg;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) { if (t < w) red[t] += red[t + w]; __syncthreads(); }
    dg = red[0];
    __syncthreads();
    if (off <= 1e-30 * (dg > 0 ? dg : 1.0)) break;
    for (int step = 0; step < n - 1; ++step) {
      if (t < np) {                                        // tournament pairing of step `step`
        int p_, q_;
        if (t == 0) { p_ = n - 1; q_ = step; }
        else { p_ = (step + t) % (n - 1); q_ = (step - t + (n - 1)) % (n - 1); }
        if (p_ > q_) { const int x = p_; p_ = q_; q_ = x; }
        pr[t][0] = p_; pr[t][1] = q_;
        const double apq = a[p_][q_];
        double c = 1.0, sn = 0.0;
        if (apq != 0.0) {
          const double theta = (a[q_][q_] - a[p_][p_]) / (2.0 * apq);
          const double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
          c = 1.0 / sqrt(tt * tt + 1.0);
          sn = tt * c;
        }
        cs[t][0] = c; cs[t][1] = sn;
      }
      __syncthreads();
      for (int e = t; e < np * n; e += 256) {              // columns p, q of A and V (the rotations of a step are disjoint)
        const int pi = e / n, k = e % n;
        const int p_ = pr[pi][0], q_ = pr[pi][1];
        const double c = cs[pi][0], sn = cs[pi][1];
        const double akp = a[k][p_], akq = a[k][q_];
        a[k][p_] = c * akp - sn * akq; a[k][q_] = sn * akp + c * akq;
        const double vkp = v[k][p_], vkq = v[k][q_];
        v[k][p_] = c * vkp - sn * vkq; v[k][q_] = sn * vkp + c * vkq;
      }
      __syncthreads();
      for (int e = t; e < np * n; e += 256) {              // rows p, q of A
        const int pi = e / n, k = e % n;
        const int p_ = pr[pi][0], q_ = pr[pi][1];
        const double c = cs[pi][0], sn = cs[pi][1];
        const double apk = a[p_][k], aqk = a[q_][k];
        a[p_][k] = c * apk - sn * aqk; a[q_][k] = sn * apk + c * aqk;
      }
      __syncthreads();
    }
  }
  // coef = W S^-1 (columns of directions with a non-negligible singular value), sc = [sqrt(S^2 + 1) - 1 | 1 / sqrt(S^2 + 1) - 1]
  red[t] = t < q ? a[t][t] : 0.0;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) { if (t < w) red[t] = red[t] > red[t + w] ? red[t] : red[t + w]; __syncthreads(); }
  const double smax = red[0] > 1e-300 ? red[0] : 1e-300;
  for (int e = t; e < q * q; e += 256) {
    const int i = e / q, j = e % q;
    const double s2 = a[j][j];
    coef[e] = s2 > 1e-14 * smax ? (real)(v[i][j] / sqrt(s2)) : (real)0;
  }
  if (t < q) {
    const double s2 = a[t][t];
    const bool keep = s2 > 1e-14 * smax;
    const double sp = sqrt(s2 + 1.0);
    sc[t] = keep ? (real)(sp - 1.0) : (real)0;
    sc[q + t] = keep ? (real)(1.0 / sp - 1.0) : (real)0;
  }
}

template <typename real>
__global__ __launch_bounds__(256) void k_scale_cols(int rows, int cols, real* __restrict__ X, int ldx, const real* __restrict__ sc) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < (int64_t)rows * cols) X[(e / cols) * ldx + e % cols] *= sc[e % cols];
}

int64_t root_update_ws_elems(int m, int r, int q) { return (int64_t)r * q * 2 + (int64_t)q * q * 2 + (int64_t)m * q + 2 * (int64_t)q; }

template <typename real>
static int root_update_impl(int m, int r, int q, real* d_L, int ldl, real* d_R, int ldr, const real* d_V, int ldv, real* d_ws, int64_t ws_elems,
                            void* stream) {
  if (m < 1 || r < 1 || q < 1 || !d_L || !d_R || !d_V || !d_ws || ws_elems < root_update_ws_elems(m, r, q)) return WISKI_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  real* p = d_ws;                       // [r, q]
  real* Uq = p + (int64_t)r * q;        // [r, q]
  real* gram = Uq + (int64_t)r * q;     // [q, q]
  real* coef = gram + (int64_t)q * q;   // [q, q]
  real* T = coef + (int64_t)q * q;      // [m, q]
  real* sc = T + (int64_t)m * q;        // [2 q]
  int rc = launch_gemm<real>(1, 0, r, q, m, (real)1, d_R, ldr, d_V, ldv, (real)0, p, q, s);              // p = R^T V        URLT:79
  if (rc) return rc;
  rc = launch_gemm<real>(1, 0, q, q, r, (real)1, p, q, p, q, (real)0, gram, q, s);                       // p^T p
  if (rc) return rc;
  const bool on_device = q <= ROOT_JACOBI_MAXQ;           // small batches: eigenproblem on the device, the call stays asynchronous
  if (on_device) {
    hipLaunchKernelGGL((k_gram_eigh<real>), dim3(1), dim3(256), 0, s, q, (const real*)gram, coef, sc);
    if (hipGetLastError() != hipSuccess) return WISKI_E_LAUNCH;
  }
  std::vector<real> hg((size_t)(on_device ? 0 : q) * q);
  if (!on_device) {
  if (hipMemcpyAsync(hg.data(), gram, sizeof(real) * q * q, hipMemcpyDeviceToHost, s) != hipSuccess) return WISKI_E_LAUNCH;
  if (hipStreamSynchronize(s) != hipSuccess) return WISKI_E_LAUNCH;
  std::vector<double> a((size_t)q * q), vec;
  for (size_t i = 0; i < a.size(); ++i) a[i] = (double)hg[i];
  for (int i = 0; i < q; ++i)
    for (int j = i + 1; j < q; ++j) a[(size_t)i * q + j] = a[(size_t)j * q + i] = 0.5 * (a[(size_t)i * q + j] + a[(size_t)j * q + i]);
  jacobi_eigh(q, a, vec);
  double smax = 0;
  for (int j = 0; j < q; ++j) smax = a[(size_t)j * q + j] > smax ? a[(size_t)j * q + j] : smax;
  std::vector<real> hcoef((size_t)q * q, (real)0), hsc((size_t)2 * q, (real)0);
  for (int j = 0; j < q; ++j) {
    const double s2 = a[(size_t)j * q + j];
    if (!(s2 > 1e-14 * (smax > 1e-300 ? smax : 1e-300))) continue;      // directions already in the null space of p: no change
    const double is = 1.0 / sqrt(s2), sp = sqrt(s2 + 1.0);
    for (int i = 0; i < q; ++i) hcoef[(size_t)i * q + j] = (real)(vec[(size_t)i * q + j] * is);          // W S^-1
    hsc[j] = (real)(sp - 1.0);
    hsc[q + j] = (real)(1.0 / sp - 1.0);
  }
  if (hipMemcpyAsync(coef, hcoef.data(), sizeof(real) * q * q, hipMemcpyHostToDevice, s) != hipSuccess) return WISKI_E_LAUNCH;
  if (hipMemcpyAsync(sc, hsc.data(), sizeof(real) * 2 * q, hipMemcpyHostToDevice, s) != hipSuccess) return WISKI_E_LAUNCH;
  if (hipStreamSynchronize(s) != hipSuccess) return WISKI_E_LAUNCH;     // the host staging buffers go out of scope at the brace
  }
  rc = launch_gemm<real>(0, 0, r, q, q, (real)1, p, q, coef, q, (real)0, Uq, q, s);                      // U_q = p W S^-1 (orthonormal columns)
  if (rc) return rc;
  const unsigned blocks = (unsigned)(((int64_t)m * q + 255) / 256);
  for (int which = 0; which < 2; ++which) {
    real* X = which == 0 ? d_L : d_R;
    const int ldx = which == 0 ? ldl : ldr;
    rc = launch_gemm<real>(0, 0, m, q, r, (real)1, X, ldx, Uq, q, (real)0, T, q, s);                     // X U_q
    if (rc) return rc;
    hipLaunchKernelGGL((k_scale_cols<real>), dim3(blocks), dim3(256), 0, s, m, q, T, q, sc + which * q);
    rc = launch_gemm<real>(0, 1, m, r, q, (real)1, T, q, Uq, q, (real)1, X, ldx, s);                     // X += (X U_q) diag(.) U_q^T
    if (rc) return rc;
  }
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}

// ------------------------------------------------------ dense Woodbury factor, one call ---
// The whole build of lazy/dense_woodbury.py (the dense regime's posterior factor, BFN:343-404 with Kt^(1/2) as the root) queued from C:
//   G = Kt^(1/2) (Kronecker eigenbasis applied to the identity),  B = I + sym(G A G),  B = C C^T and X = C^-1 (wiski_potrf_inverse),
//   T = X G,  M = T^T T = (Kt^-1 + A)^-1,  *d_logdiag += sum log diag C.
// Fifteen-odd launches issued back to back instead of one framework op at a time (the dense reference step is host-bound: DESIGN 7).
// work: 4 m^2 reals (G, AG / T, B scratch, X).  d_chol [m, m] receives C, d_M [m, m] receives M.  *d_info |= 1 on a non-positive pivot
// (the caller then takes the jitter escalation, like psd_safe_cholesky).
template <typename real>
__global__ __launch_bounds__(256) void k_sym_plus_identity(int n, const real* __restrict__ B, real* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)n * n) return;
  const int i = (int)(e / n), j = (int)(e % n);
  out[e] = (real)0.5 * (B[e] + B[(int64_t)j * n + i]) + (i == j ? (real)1 : (real)0);
}
static inline int spectral_mm_any(const wiski_grid* g, const float* ev, const float* el, float ks, const float* V, int k, float* tmp, float* out, void* s) {
  return wiski_kron_spectral_mm_f32(g, ev, el, ks, 0.f, 0.5f, 0.f, V, k, tmp, out, s);
}
static inline int spectral_mm_any(const wiski_grid* g, const double* ev, const double* el, double ks, const double* V, int k, double* tmp, double* out, void* s) {
  return wiski_kron_spectral_mm_f64(g, ev, el, ks, 0.0, 0.5, 0.0, V, k, tmp, out, s);
}
static inline int spmv_sym_any(const wiski_grid* g, const float* A, const float* V, int k, float* out, void* s) { return wiski_stencil_spmv_sym_f32(g, A, V, k, nullptr, 0.f, out, s); }
static inline int spmv_sym_any(const wiski_grid* g, const double* A, const double* V, int k, double* out, void* s) { return wiski_stencil_spmv_sym_f64(g, A, V, k, nullptr, 0.0, out, s); }

template <typename real>
static int dense_factor_impl(const wiski_grid* grid, const real* d_A_half, const real* d_evec, const real* d_eval, real kscale, real* d_work, int64_t work_elems,
                             real* d_chol, real* d_M, double* d_logdiag, int32_t* d_info, void* stream) {
  if (!grid || !d_A_half || !d_evec || !d_eval || !d_work || !d_chol || !d_M || !d_logdiag || !d_info) return WISKI_E_BADARG;
  int64_t m64 = 1;
  for (int q = 0; q < grid->d; ++q) m64 *= grid->g[q];
  if (m64 < 1 || m64 > 4096 || work_elems < 4 * m64 * m64) return WISKI_E_BADARG;
  const int m = (int)m64;
  hipStream_t s = (hipStream_t)stream;
  real *G = d_work, *T = G + m64 * m64, *Bs = T + m64 * m64, *X = Bs + m64 * m64;
  const unsigned nb = (unsigned)((m64 * m64 + 255) / 256);
  hipLaunchKernelGGL((k_set_identity<real>), dim3(nb), dim3(256), 0, s, m, X, m);                  // X as the identity, for now
  int rc = spectral_mm_any(grid, d_evec, d_eval, kscale, X, m, T, G, stream);                       // G = Kt^(1/2)      (symmetric)
  if (rc == WISKI_OK) rc = spmv_sym_any(grid, d_A_half, G, m, T, stream);                           // rows of T = columns of A G
  if (rc == WISKI_OK) rc = spectral_mm_any(grid, d_evec, d_eval, kscale, T, m, X, Bs, stream);       // G A G
  if (rc != WISKI_OK) return rc;
  hipLaunchKernelGGL((k_sym_plus_identity<real>), dim3(nb), dim3(256), 0, s, m, (const real*)Bs, d_chol);
  rc = potrf_inverse<real>(m, d_chol, m, X, m, d_info, stream);                                     // C in place, X = C^-1
  if (rc == WISKI_OK) rc = launch_gemm<real>(0, 0, m, m, m, (real)1, X, m, G, m, (real)0, T, m, s);   // T = C^-1 G
  if (rc == WISKI_OK) rc = launch_gemm<real>(1, 0, m, m, m, (real)1, T, m, T, m, (real)0, d_M, m, s); // M = T^T T
  if (rc == WISKI_OK) rc = logdiag_impl<real>(m, d_chol, m, d_logdiag, stream);
  return rc;
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
int wiski_dense_factor_f32(const wiski_grid* g, const float* A_half, const float* evec, const float* eval, float kscale, float* work, int64_t work_elems, float* chol, float* M, double* logdiag, int32_t* info, void* s) { return dense_factor_impl<float>(g, A_half, evec, eval, kscale, work, work_elems, chol, M, logdiag, info, s); }
int wiski_dense_factor_f64(const wiski_grid* g, const double* A_half, const double* evec, const double* eval, double kscale, double* work, int64_t work_elems, double* chol, double* M, double* logdiag, int32_t* info, void* s) { return dense_factor_impl<double>(g, A_half, evec, eval, kscale, work, work_elems, chol, M, logdiag, info, s); }
int wiski_potrf_f32(int32_t n, float* A, int32_t lda, int32_t* info, void* s) { return potrf_full<float>(n, A, lda, info, s); }
int wiski_potrf_f64(int32_t n, double* A, int32_t lda, int32_t* info, void* s) { return potrf_full<double>(n, A, lda, info, s); }
int wiski_potrf_inverse_f32(int32_t n, float* A, int32_t lda, float* X, int32_t ldx, int32_t* info, void* s) { return potrf_inverse<float>(n, A, lda, X, ldx, info, s); }
int wiski_potrf_inverse_f64(int32_t n, double* A, int32_t lda, double* X, int32_t ldx, int32_t* info, void* s) { return potrf_inverse<double>(n, A, lda, X, ldx, info, s); }
int wiski_trsm_f32(int32_t trans, int32_t n, int32_t nrhs, const float* L, int32_t ldl, float* B, int32_t ldb, void* s) { return trsm_impl<float>(trans, n, nrhs, L, ldl, B, ldb, (hipStream_t)s); }
int wiski_trsm_f64(int32_t trans, int32_t n, int32_t nrhs, const double* L, int32_t ldl, double* B, int32_t ldb, void* s) { return trsm_impl<double>(trans, n, nrhs, L, ldl, B, ldb, (hipStream_t)s); }
int wiski_logdiag_f32(int32_t n, const float* A, int32_t lda, double* out, void* s) { return logdiag_impl<float>(n, A, lda, out, s); }
int wiski_logdiag_f64(int32_t n, const double* A, int32_t lda, double* out, void* s) { return logdiag_impl<double>(n, A, lda, out, s); }
int64_t wiski_root_update_workspace_elems(int32_t m, int32_t r, int32_t q) { return root_update_ws_elems(m, r, q); }
int wiski_root_update_f32(int32_t m, int32_t r, int32_t q, float* L, int32_t ldl, float* R, int32_t ldr, const float* V, int32_t ldv, float* ws, int64_t ws_elems, void* s) { return root_update_impl<float>(m, r, q, L, ldl, R, ldr, V, ldv, ws, ws_elems, s); }
int wiski_root_update_f64(int32_t m, int32_t r, int32_t q, double* L, int32_t ldl, double* R, int32_t ldr, const double* V, int32_t ldv, double* ws, int64_t ws_elems, void* s) { return root_update_impl<double>(m, r, q, L, ldl, R, ldr, V, ldv, ws, ws_elems, s); }
}

#ifdef WISKI_POTRF_TIMING
extern "C" int wiski_potrf_stamps(long long* host_out) {   // tools only: the phase stamps of the last one-workgroup factorisation
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_potrf_stamp), sizeof(long long) * 17 * 8) == hipSuccess ? 0 : -2;
}
#endif
