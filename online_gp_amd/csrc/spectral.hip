// Fused Kronecker-eigenbasis preconditioner for d = 3 grids (the 50^3 workload).
//
//   [t | y] = (kron V) diag(f1 | f2) (kron V)^T r,   f1 = 1/(1 + a lam), f2 = lam f1,
//   lam = kscale * lam0[i0] lam1[i1] lam2[i2]
//
// i.e. t = (I + a Kt)^-1 r and y = (Kt^-1 + a I)^-1 r, the two vectors one PCG
// iteration needs (solve.hip).  Six mode products, done in three launches:
//   k_spec_mode0 (fwd)  : mode 0 of the forward transform, tiled over the 2500 fibres
//   k_spec_slab         : per i0-slab (g1 x g2, LDS resident): fwd modes 1,2 ->
//                         spectral scaling -> bwd modes 2,1; one block per
//                         (i0, output half), the forward part recomputed per half
//   k_spec_mode0 (bwd)  : mode 0 of the backward transform for both halves, with
//                         the r.y dot product fused into the epilogue
// All small GEMMs use one LDS primitive: Out[x][y] = sum_b F[b][x] * In[b][y] with
// a 4x4 register tile per thread and 16-byte LDS reads of both operands (2 reads
// per 16 FMAs; conflict-free: consecutive lanes read consecutive 16-B slots or
// broadcast).
#include "wiski_common.h"

#include <atomic>

#ifdef SPEC_TIMING
__device__ long long g_spec_dbg[16];
#define SPEC_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) g_spec_dbg[i] = clock64(); } while (0)
#else
#define SPEC_STAMP(i) do {} while (0)
#endif
#ifndef SPEC_ABLATE
#ifndef SPEC_ST
#define SPEC_ST 32   // fibre-tile width of the mode-0 kernels (columns of the [g0][g1 g2] view per block)
#endif
#define SPEC_ABLATE 0   // micro-benchmark hook (tools/ubench): 1 = skip LDS products, 2 = skip transposed LDS writes
#endif

template <typename real>
struct V4 { real v[4]; };

template <typename real>
__device__ __forceinline__ V4<real> lds_read4(const real* p) {
  V4<real> r;
  if constexpr (sizeof(real) == 4) {
    float4 t = *reinterpret_cast<const float4*>(p);
    r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
  } else {
    double2 a = *reinterpret_cast<const double2*>(p);
    double2 b = *reinterpret_cast<const double2*>(p + 2);
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = b.x; r.v[3] = b.y;
  }
  return r;
}

// acc[i][j] = sum_{b<gb4} F[b*ldf + 4*tx + i] * In[b*ldi + 4*ty + j]      (F stored [b][x])
// gb4 is a multiple of 4 (rows of F / In in [g, gb4) are zero).
// NOTE (measured, tools/ubench): these LDS products run ~5x below the FMA issue
// bound because one wave per SIMD exposes the LDS latency of every step and
// hipcc re-serialises source-level register rings at the loop back-edge; a
// compile-time trip count or an asm inner loop is the known next step.
template <typename real>
__device__ __forceinline__ void tile_product(const real* __restrict__ sF, int ldf, const real* __restrict__ sIn, int ldi, int gb4, int tx, int ty,
                                             real acc[4][4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (real)0;
  const real* pf = sF + 4 * tx;
  const real* pi = sIn + 4 * ty;
  if (SPEC_ABLATE & 1) { acc[0][0] = pf[0] + pi[0]; return; }
#pragma unroll 2
  for (int b0 = 0; b0 < gb4; b0 += 4) {
    V4<real> f[4], v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      f[u] = lds_read4<real>(pf + (b0 + u) * ldf);
      v[u] = lds_read4<real>(pi + (b0 + u) * ldi);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += f[u].v[i] * v[u].v[j];
  }
}

// acc[i][j] = sum_{b<gb4} F[(4*tx + i)*ldf + b] * In[b*ldi + 4*ty + j]    (F stored [x][b];
// gb4 = gb rounded up to 4; F columns and In rows in [gb, gb4) must be zero)
template <typename real>
__device__ __forceinline__ void tile_product_t(const real* __restrict__ sF, int ldf, const real* __restrict__ sIn, int ldi, int gb4, int tx, int ty,
                                               real acc[4][4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (real)0;
  const real* pf = sF + 4 * tx * ldf;
  const real* pi = sIn + 4 * ty;
  if (SPEC_ABLATE & 1) { acc[0][0] = pf[0] + pi[0]; return; }
#pragma unroll 2
  for (int b0 = 0; b0 < gb4; b0 += 4) {
    V4<real> f[4], v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      f[u] = lds_read4<real>(pf + u * ldf + b0);       // F[4tx+u][b0..b0+3]
      v[u] = lds_read4<real>(pi + (b0 + u) * ldi);     // In[b0+u][4ty..]
    }
#pragma unroll
    for (int bb = 0; bb < 4; ++bb)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += f[i].v[bb] * v[bb].v[j];
  }
}

// g x g row-major matrix -> LDS with row stride P (P = g rounded up to 4).  Split in
// two so that a kernel can issue every global load before its first barrier:
// matrix_issue() only loads into registers, matrix_commit() writes LDS (the
// destination must have been zero-filled, and a barrier passed, beforehand).
template <typename real, int NT>
struct MatrixLoad {
  static constexpr int MAXIT = (64 * 64 + NT - 1) / NT;
  real tmp[MAXIT];
  __device__ __forceinline__ void issue(const real* __restrict__ V, int g) {
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int idx = threadIdx.x + it * NT;
      tmp[it] = V[idx < g * g ? idx : 0];
    }
  }
  __device__ __forceinline__ void commit(real* sF, int g, int P) const {
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int idx = threadIdx.x + it * NT;
      if (idx < g * g) {
        const int r = idx / g;
        sF[r * P + (idx - r * g)] = tmp[it];
      }
    }
  }
};

// ------------------------------------------------------------- mode 0 ---
// dst[c][x, s] = sum_b Fm[b][x] src[c][b, s]  for an s-tile of 32 fibres, with
// Fm = V0 (forward, transposed == 0: V0^T .) or V0^T (backward: V0 .).
// grid = (ceil(S/32), ncols), block = 128 (tiles: (P0/4) x 8 <= 128 for g0 <= 64).
template <typename real, bool DOT>
__global__ __launch_bounds__(128) void k_spec_mode0(GridDev<real> G, const real* __restrict__ Va, const real* __restrict__ Vb, int split,
                                                    int transposed, const real* __restrict__ src, real* __restrict__ dst,
                                                    const real* __restrict__ rvec, int dot_c0, double* __restrict__ dots) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double s_red[16];
  constexpr int ST = SPEC_ST;
  const int g0 = G.g[0], S = G.stride[0], m = G.m;
  const int P0 = (g0 + 3) & ~3;
  real* sF = reinterpret_cast<real*>(smem);   // [P0][P0]
  real* sIn = sF + P0 * P0;                   // [P0][ST]
  const int c = blockIdx.y;
  const int s0 = blockIdx.x * ST;
  const real* __restrict__ V0 = c < split ? Va : Vb;   // generalized eigenbasis: t-half and y-half use different factors
  const real* __restrict__ sc = src + (int64_t)c * m;
  // In tile: rows b < g0 (rows up to P0 zero), 8 x 16-byte loads per row; issue all loads first
  constexpr int NV = (64 * (ST / 4) + 127) / 128;   // <= 4
  V4<real> tin[NV];
#pragma unroll
  for (int it = 0; it < NV; ++it) {
    const int t = threadIdx.x + it * 128;
    const int b = t / (ST / 4), q4 = (t - b * (ST / 4)) * 4;
    V4<real> z;
    z.v[0] = z.v[1] = z.v[2] = z.v[3] = (real)0;
    tin[it] = (b < g0 && s0 + q4 < S) ? lds_read4<real>(sc + (int64_t)b * S + s0 + q4) : z;   // S % 4 == 0 (precondition)
  }
  MatrixLoad<real, 128> ml;
  ml.issue(V0, g0);
  for (int e = threadIdx.x; e < P0 * P0; e += 128) sF[e] = (real)0;
  __syncthreads();
  ml.commit(sF, g0, P0);
#pragma unroll
  for (int it = 0; it < NV; ++it) {
    const int t = threadIdx.x + it * 128;
    const int b = t / (ST / 4), q4 = (t - b * (ST / 4)) * 4;
    if (b < P0) {
      real* d = sIn + b * ST + q4;
      d[0] = tin[it].v[0]; d[1] = tin[it].v[1]; d[2] = tin[it].v[2]; d[3] = tin[it].v[3];
    }
  }
  __syncthreads();
  const int nty = ST / 4;
  const int ntx = P0 / 4;
  const int t = threadIdx.x;
  double part = 0;
  if (t < ntx * nty) {
    const int tx = t / nty, ty = t - tx * nty;
    real acc[4][4];
    if (transposed) tile_product_t<real>(sF, P0, sIn, ST, P0, tx, ty, acc);
    else tile_product<real>(sF, P0, sIn, ST, P0, tx, ty, acc);
    const int s = s0 + 4 * ty;
    if (s < S) {
      real* __restrict__ dc = dst + (int64_t)c * m;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int x = 4 * tx + i;
        if (x < g0) {
          const int64_t e = (int64_t)x * S + s;
          if constexpr (sizeof(real) == 4) {
            *reinterpret_cast<float4*>(dc + e) = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
          } else {
            *reinterpret_cast<double2*>(dc + e) = make_double2(acc[i][0], acc[i][1]);
            *reinterpret_cast<double2*>(dc + e + 2) = make_double2(acc[i][2], acc[i][3]);
          }
          if (DOT && c >= dot_c0) {
            const V4<real> rv = lds_read4<real>(rvec + (int64_t)(c - dot_c0) * m + e);
#pragma unroll
            for (int j = 0; j < 4; ++j) part += (double)rv.v[j] * (double)acc[i][j];
          }
        }
      }
    }
  }
  if (DOT && c >= dot_c0) {
    const double tot = block_reduce_sum(part, s_red);
    if (threadIdx.x == 0) unsafeAtomicAdd(dots + (c - dot_c0), tot);
  }
}

// -------------------------------------------------------------- slab ---
// One block per (i0, half h, column c): src slab X[i1][i2] ->
//   YT[i2][i1'] = sum_i1 V1[i1][i1'] X[i1][i2]
//   ZT[i2'][i1'] = sum_i2 V2[i2][i2'] YT[i2][i1']      (full forward transform)
//   scale by f_h(lam)                                    (h = 0: f1, h = 1: f2)
//   T'T[i1'][i2] = sum_i2' V2[i2][i2'] ZT[i2'][i1']     (written transposed)
//   out[i1][i2]  = sum_i1' V1[i1][i1'] T'T[i1'][i2]  -> dst[h*k + c] slab i0
// V1, V2 stay in LDS in their natural layout; the backward products read them
// through tile_product_t.  Every global load of the block is issued up front.
template <typename real>
__global__ __launch_bounds__(256) void k_spec_slab(GridDev<real> G, const real* __restrict__ V1, const real* __restrict__ V2,
                                                   const real* __restrict__ Z1, const real* __restrict__ Z2, const real* __restrict__ evals, real kscale, real shift, const real* __restrict__ src,
                                                   real* __restrict__ dst, int k, double* __restrict__ rho) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double s_red[16];
  const int g0 = G.g[0], g1 = G.g[1], g2 = G.g[2], m = G.m;
  const int P1 = (g1 + 3) & ~3, P2 = (g2 + 3) & ~3;
  const int PM = P1 > P2 ? P1 : P2;
  real* B0 = reinterpret_cast<real*>(smem);
  real* B1 = B0 + PM * PM;
  real* sV1 = B1 + PM * PM;
  real* sV2 = sV1 + P1 * P1;
  real* sZ1 = sV2 + P2 * P2;   // backward factors of the t-half (h == 0) when they differ from V (generalized eigenbasis)
  real* sZ2 = sZ1 + P1 * P1;
  real* sE = sZ2 + P2 * P2;   // [P1 + P2] eigenvalues of dims 1, 2 (zero padded)
  const int i0 = blockIdx.x, h = blockIdx.y, c = blockIdx.z;
  const int t = threadIdx.x;
  const real* __restrict__ xs = src + (int64_t)c * m + (int64_t)i0 * g1 * g2;
  SPEC_STAMP(0);

  // X slab: g1*g2 contiguous reals, 16-byte loads (g1*g2 % 4 == 0: precondition)
  constexpr int NX = (64 * 64 / 4 + 255) / 256;   // <= 4
  V4<real> tx4[NX];
  const int nvec = g1 * g2 / 4;
#pragma unroll
  for (int it = 0; it < NX; ++it) {
    const int e4 = t + it * 256;
    V4<real> z;
    z.v[0] = z.v[1] = z.v[2] = z.v[3] = (real)0;
    tx4[it] = e4 < nvec ? lds_read4<real>(xs + 4 * e4) : z;
  }
  real ev = (real)0;
  if (t < P1) ev = t < g1 ? evals[g0 + t] : (real)0;
  else if (t < P1 + P2) ev = (t - P1) < g2 ? evals[g0 + g1 + (t - P1)] : (real)0;
  const real l0 = kscale * evals[i0];
  const bool alt = (h == 0) && (Z1 != V1 || Z2 != V2);   // block-uniform
  MatrixLoad<real, 256> m1, m2, m3, m4;
  m1.issue(V1, g1);
  m2.issue(V2, g2);
  if (alt) {
    m3.issue(Z1, g1);
    m4.issue(Z2, g2);
  }
  for (int e = t; e < 2 * PM * PM + 2 * (P1 * P1 + P2 * P2); e += 256) B0[e] = (real)0;   // B0, B1, sV1, sV2, sZ1, sZ2 (padding must be zero)
  __syncthreads();
  m1.commit(sV1, g1, P1);
  m2.commit(sV2, g2, P2);
  if (alt) {
    m3.commit(sZ1, g1, P1);
    m4.commit(sZ2, g2, P2);
  }
  const real* bV1 = alt ? sZ1 : sV1;   // factors of the backward products
  const real* bV2 = alt ? sZ2 : sV2;
  if (t < P1 + P2) sE[t] = ev;
#pragma unroll
  for (int it = 0; it < NX; ++it) {
    const int e4 = t + it * 256;
    if (e4 < nvec) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = 4 * e4 + u;
        const int b = e / g2;
        B0[b * P2 + (e - b * g2)] = tx4[it].v[u];   // X as [b = i1][y = i2], stride P2
      }
    }
  }
  __syncthreads();
  SPEC_STAMP(1);
  real acc[4][4];
  {  // step 2: Out[x = i1'][y = i2] = sum_b V1[b][x] X[b][y]  -> YT[i2][i1'] (B1, stride P1)
    const int ntx = P1 / 4, nty = P2 / 4;
    if (t < ntx * nty) {
      const int tx = t / nty, ty = t - tx * nty;
      tile_product<real>(sV1, P1, B0, P2, P1, tx, ty, acc);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) B1[(4 * ty + j) * P1 + 4 * tx + i] = acc[i][j];
    }
  }
  __syncthreads();
  SPEC_STAMP(2);
  double rho_part = 0;
  {  // step 3 + scaling: Out[x = i2'][y = i1'] = sum_b V2[b][x] YT[b][y] -> ZT[i2'][i1'] (B0, stride P1)
    const int ntx = P2 / 4, nty = P1 / 4;
    if (t < ntx * nty) {
      const int tx = t / nty, ty = t - tx * nty;
      tile_product<real>(sV2, P2, B1, P1, P2, tx, ty, acc);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const real l2 = l0 * sE[P1 + 4 * tx + i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const real lam = l2 * sE[4 * ty + j];
          real f1;
          if constexpr (sizeof(real) == 4) f1 = __frcp_rn((real)1 + shift * lam);
          else f1 = (real)1 / ((real)1 + shift * lam);
          B0[(4 * tx + i) * P1 + 4 * ty + j] = acc[i][j] * (h == 0 ? f1 : lam * f1);
          rho_part += (double)(lam * f1) * (double)acc[i][j] * (double)acc[i][j];   // r^T P r in the eigenbasis (padding: lam = 0)
        }
      }
    }
  }
  if (rho != nullptr && h == 1) {   // block-uniform
    const double tot = block_reduce_sum(rho_part, s_red);
    if (t == 0) unsafeAtomicAdd(rho + c, tot);
  }
  __syncthreads();
  SPEC_STAMP(3);
  {  // step 5: Out[x = i2][y = i1'] = sum_b V2[x][b] ZT[b][y]  -> T'T[i1'][i2] (B1, stride P2)
    const int ntx = P2 / 4, nty = P1 / 4;
    if (t < ntx * nty) {
      const int tx = t / nty, ty = t - tx * nty;
      tile_product_t<real>(bV2, P2, B0, P1, P2, tx, ty, acc);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) B1[(4 * ty + j) * P2 + 4 * tx + i] = acc[i][j];
    }
  }
  __syncthreads();
  SPEC_STAMP(4);
  {  // step 6: Out[x = i1][y = i2] = sum_b V1[x][b] T'T[b][y]  -> global
    const int ntx = P1 / 4, nty = P2 / 4;
    if (t < ntx * nty) {
      const int tx = t / nty, ty = t - tx * nty;
      tile_product_t<real>(bV1, P1, B1, P2, P1, tx, ty, acc);
      real* __restrict__ os = dst + ((int64_t)h * k + c) * m + (int64_t)i0 * g1 * g2;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int x = 4 * tx + i;
        if (x < g1) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int y = 4 * ty + j;
            if (y < g2) os[x * g2 + y] = acc[i][j];
          }
        }
      }
    }
  }
  SPEC_STAMP(5);
}

// -------------------------------------------------- slab, fp32 on the matrix cores ---
// The same four products as k_spec_slab for real = float, on v_mfma_f32_16x16x4_f32 (exact fp32).
// Every product is Out[x][y] = sum_b A[x][b] B[b][y] on a 64 x 64 padded tile; the 4 waves each own
// a 32 x 32 quadrant (2 x 2 MFMA tiles, as wiski_gemm).  An MFMA needs one LDS word per operand and
// lane (vs 8 x 16 B per 64 FMAs in the register-tile version), so the LDS latency that bounded the
// scalar kernel (~2.3 us per product at one wave per SIMD) no longer does.
//   P2  C2[i1'][i2]  = sum_i1  V1[i1][i1']  X[i1][i2]          A = V1^T (row-wise reads), B = X
//   P3  C3[i1'][i2'] = sum_i2  C2[i1'][i2]  V2[i2][i2']         A = C2,  B = V2          then * f_h(lam)
//   P5  C5[i1'][i2]  = sum_i2' C3[i1'][i2'] bV2[i2][i2']        A = C3,  B = bV2^T
//   P6  out[i1][i2]  = sum_i1' bV1[i1][i1'] C5[i1'][i2]         A = bV1, B = C5
// (bV = Z for the t-half of a generalized eigenbasis, else V).  Two LDS row strides keep every operand
// read bank-conflict free: LDT = 80 (16 mod 64) for operands whose 16-lane groups read a row
// (B[b][y], A^T[b][x]); LDN = 68 (4 mod 64) for operands whose lanes walk a column (A[x][b], B^T[y][b]).
// Out-of-range rows / columns are loaded as exact zeros, so the padded products are exact.
using spec_f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int SPEC_LDT = 80, SPEC_LDN = 68;

// RT = 16-row tiles per wave: 2 (a 32 x 32 quadrant, 4 waves per product) or 1 (a 16 x 32 strip, 8 waves; acc[1] unused)
template <bool A_NAT, bool B_NAT, int KS, int RT = 2>
__device__ __forceinline__ void spec_mfma_product(const float* __restrict__ pa, const float* __restrict__ pb, int wr, int wc, int lane,
                                                  spec_f32x4 acc[2][2]) {
#pragma unroll
  for (int a = 0; a < RT; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[a][c][r] = 0.f;
  const int l15 = lane & 15, l4 = lane >> 4;
  const float* qa = A_NAT ? pa + (wr * 16 * RT + l15) * SPEC_LDN + l4 : pa + l4 * SPEC_LDT + wr * 16 * RT + l15;
  const float* qb = B_NAT ? pb + (wc * 32 + l15) * SPEC_LDN + l4 : pb + l4 * SPEC_LDT + wc * 32 + l15;
  constexpr int SA = A_NAT ? 1 : SPEC_LDT, TA = A_NAT ? 16 * SPEC_LDN : 16;   // step per k, step per 16-row tile
  constexpr int SB = B_NAT ? 1 : SPEC_LDT, TB = B_NAT ? 16 * SPEC_LDN : 16;
  // KS = (padded inner dimension) / 4 is a compile-time constant (the zero padding makes a larger KS exact):
  // every operand word is read up front (4 KS registers), so the LDS latency is paid once and the 4 KS MFMAs
  // issue back to back.  (A run-time trip count turned this into one branch + s_waitcnt lgkmcnt(0) per step.)
  float af[KS][RT], bf[KS][2];
#pragma unroll
  for (int i = 0; i < KS; ++i) {
    af[i][0] = qa[4 * i * SA];
    if constexpr (RT == 2) af[i][1] = qa[4 * i * SA + TA];
    bf[i][0] = qb[4 * i * SB]; bf[i][1] = qb[4 * i * SB + TB];
  }
#pragma unroll
  for (int i = 0; i < KS; ++i) {
    acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][0], bf[i][0], acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][0], bf[i][1], acc[0][1], 0, 0, 0);
    if constexpr (RT == 2) {
      acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][1], bf[i][0], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][1], bf[i][1], acc[1][1], 0, 0, 0);
    }
  }
}

// rows x cols compact row-major matrix -> LDS image [64][ld] with exact-zero padding, no integer division and
// no separate zero fill: thread (r, c) mapping, 64 / VW threads per padded row, VW-wide loads and LDS stores
// (VW = 2 needs even `cols`: rows are then 8-byte aligned).  issue() only loads, commit() only stores.
template <int VW, int NT = 256>
struct SpecTile {
  static constexpr int TPR = 64 / VW;          // threads per row
  static constexpr int RPP = NT / TPR;         // rows per pass
  static constexpr int NP = 64 / RPP;          // passes
  float v[NP][VW];
  __device__ __forceinline__ void issue(const float* __restrict__ M, int rows, int cols) {
    const int r0 = threadIdx.x / TPR, c0 = (threadIdx.x % TPR) * VW;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int r = r0 + RPP * p;
      const bool ok = r < rows && c0 < cols;
      const float* q = M + (ok ? r * cols + c0 : 0);
      if constexpr (VW == 2) {
        const float2 x = *reinterpret_cast<const float2*>(q);
        v[p][0] = ok ? x.x : 0.f;
        v[p][1] = ok ? x.y : 0.f;
      } else {
        const float x = *q;
        v[p][0] = ok ? x : 0.f;
      }
    }
  }
  __device__ __forceinline__ void commit(float* __restrict__ dst, int ld) const {
    const int r0 = threadIdx.x / TPR, c0 = (threadIdx.x % TPR) * VW;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      float* d = dst + (r0 + RPP * p) * ld + c0;
      if constexpr (VW == 2) *reinterpret_cast<float2*>(d) = make_float2(v[p][0], v[p][1]);
      else *d = v[p][0];
    }
  }
};

// Two-level block (wiski_twolevel, include/wiski.h) as the kernel sees it.
struct TwoLevelDev {
  int r, nslab;
  const unsigned long long* mask;
  const int* off;
  const unsigned short* pos;
  const float* N;
  unsigned long long* cs;      // [r] {application epoch << 32 | coefficient bits}
  unsigned epoch;              // unique per application (never 0)
};
constexpr int SPEC_TL_MAXR = 512;

// NW = 4: the waves own 32 x 32 quadrants; NW = 8: 16 x 32 strips (half the MFMA chain per wave, two waves per SIMD)
// TL: the r selected modes (largest prior eigenvalues) are not scaled by the diagonal model: the blocks that hold them
// exchange their coefficients c_S through tl.cs, every block forms the rows of N c_S that belong to its slab (y-half:
// N c_S, t-half: D_S^-1 N c_S) and rho takes c_S^T N c_S for them.  The exchange needs no counter and no fence: every
// coefficient travels as ONE 64-bit word {epoch of this application, fp32 bits}, written and read with device-scope atomics
// (around the per-XCD L2s), and a reader spins on each word it wants until it carries the current epoch -- one store-to-load
// latency in all (a release / acquire arrival counter with its L2 write-back and invalidate cost 10 us per application).
// Blocks of slabs without selected modes never wait.  All g0 x 2 blocks are co-resident (one per CU), so the wait cannot
// deadlock; a bounded spin turns a scheduling accident into a failed solve (reported as non-convergence), not a hung device.
template <int KS, int VW, int NW, bool TL>
__global__ __launch_bounds__(64 * NW) void k_spec_slab_mfma(GridDev<float> G, const float* __restrict__ V1, const float* __restrict__ V2,
                                                        const float* __restrict__ Z1, const float* __restrict__ Z2,
                                                        const float* __restrict__ evals, float kscale, float shift,
                                                        const float* __restrict__ src, float* __restrict__ dst, int k, double* __restrict__ rho,
                                                        TwoLevelDev tl) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double s_red[16];
  __shared__ float sE[128];                       // eigenvalues of dims 1 | 2, zero padded to 64 each
  __shared__ float sC[TL ? SPEC_TL_MAXR : 1];     // the exchanged coefficients c_S
  __shared__ float sCp[TL ? SPEC_TL_MAXR : 1];    // this slab's rows of N c_S
  float* bufA = reinterpret_cast<float*>(smem);   // [64][80]: X, later C3 (stride LDN)
  float* bufB = bufA + 64 * SPEC_LDT;             // [64][80]: C2 (stride LDN), later C5 (stride LDT)
  float* sV1 = bufB + 64 * SPEC_LDT;              // [b][x], stride LDT
  float* sV2 = sV1 + 64 * SPEC_LDT;               // [b][y], stride LDT
  float* sB1 = sV2 + 64 * SPEC_LDT;               // bV1[x][b], stride LDN
  float* sB2 = sB1 + 64 * SPEC_LDN;               // bV2[y][b], stride LDN
  const int g0 = G.g[0], g1 = G.g[1], g2 = G.g[2], m = G.m;
  const int i0 = blockIdx.x, h = blockIdx.y, c = blockIdx.z;
  constexpr int RT = NW == 4 ? 2 : 1;         // 16-row tiles per wave
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wr = w >> 1, wc = w & 1;
  const bool alt = (h == 0) && (Z1 != V1 || Z2 != V2);   // block-uniform
  SPEC_STAMP(0);
  SpecTile<VW, 64 * NW> tX, tV1, tV2, tB1, tB2;
  tX.issue(src + (int64_t)c * m + (int64_t)i0 * g1 * g2, g1, g2);
  tV1.issue(V1, g1, g1);
  tV2.issue(V2, g2, g2);
  tB1.issue(alt ? Z1 : V1, g1, g1);
  tB2.issue(alt ? Z2 : V2, g2, g2);
  if (t < 128) {
    const int q = t & 63;
    float ev = 0.f;
    if (t < 64 ? q < g1 : q < g2) ev = evals[t < 64 ? g0 + q : g0 + g1 + q];
    sE[t] = ev;
  }
  const float l0 = kscale * evals[i0];
  // two-level: the selection masks of the rows this lane holds C3 entries of (bit y of row x), issued with the other loads
  int tl_o0 = 0, tl_ns = 0;
  unsigned long long mrow[RT][4];
  if constexpr (TL) {
    tl_o0 = tl.off[i0];
    tl_ns = tl.off[i0 + 1] - tl_o0;              // block-uniform
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) mrow[a][r] = tl_ns > 0 ? tl.mask[i0 * 64 + wr * 16 * RT + a * 16 + (lane >> 4) * 4 + r] : 0ull;
  }
  // two-level: this wave's rows of N (row e = w, w + NW, ...; three 64-column chunks each: r <= 192, <= 8 rows per wave) requested
  // NOW -- they do not depend on the exchange, and fetched after it (L2, ~0.7 us per row, one row after the other) they were most of
  // what the block cost the kernel
  constexpr int TL_PF_ROWS = 8, TL_PF_CH = 3;
  float npf[TL ? TL_PF_ROWS : 1][TL ? TL_PF_CH : 1];
  bool tl_pf = false;
  if constexpr (TL) {
    tl_pf = tl_ns > 0 && tl.r <= 64 * TL_PF_CH && tl_ns <= NW * TL_PF_ROWS;      // block-uniform
    if (tl_pf) {
#pragma unroll
      for (int i = 0; i < TL_PF_ROWS; ++i) {
        const int e = w + NW * i;
        const float* __restrict__ nrow = tl.N + (int64_t)(tl_o0 + (e < tl_ns ? e : 0)) * tl.r;
#pragma unroll
        for (int j = 0; j < TL_PF_CH; ++j) npf[i][j] = (e < tl_ns && lane + 64 * j < tl.r) ? nrow[lane + 64 * j] : 0.f;
      }
    }
  }
  SPEC_STAMP(6);
  tX.commit(bufA, SPEC_LDT);           // every image is written in full (padding = zeros); bufB is first written by P2
  tV1.commit(sV1, SPEC_LDT);
  tV2.commit(sV2, SPEC_LDT);
  tB1.commit(sB1, SPEC_LDN);
  tB2.commit(sB2, SPEC_LDN);
  SPEC_STAMP(7);
  __syncthreads();
  SPEC_STAMP(1);
  spec_f32x4 acc[2][2];
  const int l15 = lane & 15, l4 = lane >> 4;
  auto store_tiles = [&](float* __restrict__ out, int ld) {       // C fragments -> out[row][col], row-major with stride ld
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
      for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(wr * 16 * RT + a * 16 + l4 * 4 + r) * ld + wc * 32 + cc * 16 + l15] = acc[a][cc][r];
  };
  // P2: A = V1^T (sV1 [b][x]), B = X (bufA [b][y])  -> C2 natural (bufB, stride LDN)
  spec_mfma_product<false, false, KS, RT>(sV1, bufA, wr, wc, lane, acc);
  store_tiles(bufB, SPEC_LDN);
  __syncthreads();
  SPEC_STAMP(2);
  // P3: A = C2 (bufB natural), B = V2 (sV2 [b][y]) -> scaled C3 natural (bufA, stride LDN)
  spec_mfma_product<true, false, KS, RT>(bufB, sV2, wr, wc, lane, acc);
  float rho_lane = 0.f;   // 16 terms per lane in fp32, the cross-lane / cross-block sum in fp64
#pragma unroll
  for (int a = 0; a < RT; ++a)
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      const float e2 = sE[64 + wc * 32 + cc * 16 + l15];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float lam = l0 * sE[wr * 16 * RT + a * 16 + l4 * 4 + r] * e2;
        const float f1 = __frcp_rn(1.f + shift * lam);
        const float v = acc[a][cc][r];
        bool sel = false;
        if constexpr (TL) sel = (mrow[a][r] >> (wc * 32 + cc * 16 + l15)) & 1ull;
        if (!sel) {
          rho_lane += (lam * f1) * v * v;   // r^T P r in the eigenbasis (padding: lam = 0)
          acc[a][cc][r] = v * (h == 0 ? f1 : lam * f1);
        }                                   // selected modes stay unscaled: the exact block below replaces them
      }
    }
  store_tiles(bufA, SPEC_LDN);
  if (rho != nullptr && h == 1) {   // block-uniform
    const double tot = block_reduce_sum((double)rho_lane, s_red);
    if (t == 0) unsafeAtomicAdd(rho + c, tot);
  }
  __syncthreads();
  if constexpr (TL) {
    if (tl_ns > 0) {                // block-uniform
      constexpr int NT = 64 * NW;
      // (1) this slab's raw coefficients -> the exchange buffer (the y-half blocks write; the t-half computed the same numbers)
      if (h == 1)
        for (int e = t; e < tl_ns; e += NT) {
          const unsigned pp = tl.pos[tl_o0 + e];
          const unsigned long long wv = ((unsigned long long)tl.epoch << 32) | (unsigned long long)__float_as_uint(bufA[(pp >> 8) * SPEC_LDN + (pp & 255u)]);
          __hip_atomic_store(tl.cs + tl_o0 + e, wv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      // (2) all r coefficients -> LDS, each as soon as its word carries this application's epoch
      for (int e = t; e < tl.r; e += NT) {
        unsigned long long wv = __hip_atomic_load(tl.cs + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int spins = 0; (unsigned)(wv >> 32) != tl.epoch && spins < (1 << 20); ++spins) {
          __builtin_amdgcn_s_sleep(1);
          wv = __hip_atomic_load(tl.cs + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // (a word that never arrived -- its writer block was not resident: a partitioned device, CUs held by other work -- leaves a
        //  stale coefficient in this application: counted in the sticky word behind the exchange words; the owner of the block reads
        //  it with the next refresh's verdict and drops the block)
        if ((unsigned)(wv >> 32) != tl.epoch) atomicAdd(tl.cs + tl.r, 1ull);
        sC[e] = __uint_as_float((unsigned)wv);
      }
      __syncthreads();
      // (3) the rows of N c_S of this slab: one wave per row, lanes stride the columns
      float rsel = 0.f;
      if (tl_pf) {
        float cq[TL_PF_CH];
#pragma unroll
        for (int j = 0; j < TL_PF_CH; ++j) cq[j] = lane + 64 * j < tl.r ? sC[lane + 64 * j] : 0.f;
#pragma unroll
        for (int i = 0; i < TL_PF_ROWS; ++i) {
          const int e = w + NW * i;
          if (e < tl_ns) {                                     // wave-uniform
            float d = 0.f;
#pragma unroll
            for (int j = 0; j < TL_PF_CH; ++j) d += npf[i][j] * cq[j];
            d = wave_reduce_sum<float>(d);
            if (lane == 0) {
              sCp[e] = d;
              rsel += sC[tl_o0 + e] * d;
            }
          }
        }
      } else {
        for (int e = w; e < tl_ns; e += NW) {
          const float* __restrict__ nrow = tl.N + (int64_t)(tl_o0 + e) * tl.r;
          float d = 0.f;
          for (int q = lane; q < tl.r; q += 64) d += nrow[q] * sC[q];
          d = wave_reduce_sum<float>(d);
          if (lane == 0) {
            sCp[e] = d;
            rsel += sC[tl_o0 + e] * d;
          }
        }
      }
      if (rho != nullptr && h == 1) {
        const double tot = block_reduce_sum((double)rsel, s_red);     // (contains the barrier that publishes sCp)
        if (t == 0) unsafeAtomicAdd(rho + c, tot);
      } else {
        __syncthreads();
      }
      // (4) back into the C3 image: N c_S for the y-half, D_S^-1 N c_S for the t-half
      for (int e = t; e < tl_ns; e += NT) {
        const unsigned pp = tl.pos[tl_o0 + e];
        const int x = pp >> 8, y = pp & 255u;
        const float lam = l0 * sE[x] * sE[64 + y];
        bufA[x * SPEC_LDN + y] = h == 1 ? sCp[e] : sCp[e] / lam;
      }
      __syncthreads();
    }
  }
  SPEC_STAMP(3);
  // P5: A = C3 (bufA natural), B = bV2^T (sB2 [y][b]) -> C5 [b = i1'][y = i2] (bufB, stride LDT)
  spec_mfma_product<true, true, KS, RT>(bufA, sB2, wr, wc, lane, acc);
  store_tiles(bufB, SPEC_LDT);
  __syncthreads();
  SPEC_STAMP(4);
  // P6: A = bV1 (sB1 natural), B = C5 (bufB [b][y]) -> global
  spec_mfma_product<true, false, KS, RT>(sB1, bufB, wr, wc, lane, acc);
  float* __restrict__ os = dst + ((int64_t)h * k + c) * m + (int64_t)i0 * g1 * g2;
#pragma unroll
  for (int a = 0; a < RT; ++a)
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int x = wr * 16 * RT + a * 16 + l4 * 4 + r, y = wc * 32 + cc * 16 + l15;
        if (x < g1 && y < g2) os[x * g2 + y] = acc[a][cc][r];
      }
  SPEC_STAMP(5);
}

// ----------------------------------------------- two-level block, many columns ---
// The exact block for a multi-column solve.  The k = 1 slab kernel exchanges its r selected coefficients between the blocks of ONE
// launch (stamped words, a spin per application); with k columns that is k r words per application and a wait per column in every
// participating block.  Here the block is applied around the slab launch instead -- nothing spins, nothing needs co-residency:
//   k_tl_coef_mc   (one workgroup per selected slab and column, in front of the slab launch): the selected coefficients
//                  c_S = X_S^T r straight from the mode-0 image, in two stages (the distinct x of the slab's modes first)
//   k_tl_apply_mc  (one workgroup per 32 rows of N and column): d = N c_S, rho[c] += c_S . d
//   k_spec_slab_mfma_mc<..., TL = true>: the lanes that hold a selected entry of C3 replace it by d (y-half) / d / lambda (t-half)
//   and leave it out of their rho term.
// (First form: ONE kernel, a workgroup per column walking the selected slabs with a wave per mode -- 64 CUs busy, every mode a
//  latency-bound chain of g2 LDS round trips: 106 us per application at 50^3, r = 192, 64 columns.)
constexpr int TLC_NT = 256;
static inline size_t tl_coef_mc_lds(int g1, int g2) { return (size_t)(g1 * (g2 | 1) + g1 * g1 + g2 * g2 + 64 * g2) * sizeof(float); }

__global__ __launch_bounds__(TLC_NT) void k_tl_coef_mc(GridDev<float> G, const float* __restrict__ V1, const float* __restrict__ V2,
                                                       const float* __restrict__ src, TwoLevelDev tl, float* __restrict__ cS) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ float sC[SPEC_TL_MAXR];
  __shared__ int sSlab;
  const int g0 = G.g[0], g1 = G.g[1], g2 = G.g[2], m = G.m;
  const int ldy = g2 | 1;                          // odd row stride: consecutive threads walk a column of Y conflict-free
  float* sY = reinterpret_cast<float*>(smem);      // [g1][ldy]   mode-0 image of this slab
  float* sV1 = sY + g1 * ldy;                      // [b][x]
  float* sV2 = sV1 + g1 * g1;                      // [b][y]
  const int c = blockIdx.y, t = threadIdx.x;
  if (t < 64) {                                    // the blockIdx.x-th slab that holds selected modes (g0 <= 64: one ballot)
    const bool has = t < g0 && tl.off[t + 1] > tl.off[t];
    const unsigned long long bal = __ballot(has);
    if (has && __popcll(bal & ((1ull << t) - 1ull)) == (int)blockIdx.x) sSlab = t;
    if (t == 0 && __popcll(bal) <= (int)blockIdx.x) sSlab = g0;
  }
  const int ne = g1 * g2;
  // all three images in one round of independent loads (g <= 64: at most 16 elements per thread and image)
  constexpr int NB = 64 * 64 / TLC_NT;
  float v1[NB], v2[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int e = t + TLC_NT * j;
    v1[j] = e < g1 * g1 ? V1[e] : 0.f;
    v2[j] = e < g2 * g2 ? V2[e] : 0.f;
  }
  for (int e = t; e < SPEC_TL_MAXR; e += TLC_NT) sC[e] = 0.f;
  __syncthreads();                                 // sSlab
  const int i0 = sSlab;
  if (i0 >= g0) return;                            // block-uniform (nslab overstated: nothing to do)
  const float* __restrict__ col = src + (int64_t)c * m + (int64_t)i0 * ne;
  const int o0 = tl.off[i0], ns = tl.off[i0 + 1] - o0;
  float vy[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int e = t + TLC_NT * j;
    vy[j] = e < ne ? col[e] : 0.f;
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int e = t + TLC_NT * j;
    if (e < g1 * g1) sV1[e] = v1[j];
    if (e < g2 * g2) sV2[e] = v2[j];
    if (e < ne) sY[(e / g2) * ldy + (e % g2)] = vy[j];
  }
  __syncthreads();
  // stage 1: T[xi][b2] = sum_b1 V1[b1][x_xi] Y[b1][b2] for the nx distinct x of this slab's modes (rows of the mask that are not
  // empty), a thread per (xi, b2) and four xi per pass; stage 2: c_e = sum_b2 T[xi(e)][b2] V2[b2][y_e], a thread per mode.
  // (nx g1 g2 + ns g2 multiply-adds instead of ns g1 g2: the modes of a slab share few x.)
  __shared__ unsigned char sXi[64];                // x -> index among the distinct ones
  __shared__ unsigned char sXl[64];                // index -> x
  __shared__ int sNx;
  if (t < 64) {
    const bool has = tl.mask[i0 * 64 + t] != 0ull;
    const unsigned long long bal = __ballot(has);
    const int xi = __popcll(bal & ((1ull << t) - 1ull));
    sXi[t] = (unsigned char)xi;
    if (has) sXl[xi] = (unsigned char)t;
    if (t == 0) sNx = __popcll(bal);
  }
  __syncthreads();
  const int nx = sNx;
  float* sTT = sV2 + g2 * g2;                      // [nx][g2], behind the three images
  for (int o = t; o < ((nx + 3) / 4) * g2; o += TLC_NT) {
    const int xq = o / g2, b2 = o - xq * g2;
    int xs[4];
    float acc4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) xs[j] = sXl[4 * xq + j < nx ? 4 * xq + j : 0];
    for (int b = 0; b < g1; ++b) {
      const float yv = sY[b * ldy + b2];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc4[j] += sV1[b * g1 + xs[j]] * yv;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (4 * xq + j < nx) sTT[(4 * xq + j) * g2 + b2] = acc4[j];
  }
  __syncthreads();
  for (int e = t; e < ns; e += TLC_NT) {
    const unsigned pp = tl.pos[o0 + e];
    const int x = pp >> 8, y = pp & 255u;
    const float* __restrict__ tr = sTT + (int)sXi[x] * g2;
    float a0 = 0.f, a1 = 0.f;
    int b = 0;
    for (; b + 1 < g2; b += 2) {
      a0 += tr[b] * sV2[b * g2 + y];
      a1 += tr[b + 1] * sV2[(b + 1) * g2 + y];
    }
    if (b < g2) a0 += tr[b] * sV2[b * g2 + y];
    sC[e] = a0 + a1;
  }
  __syncthreads();
  for (int e = t; e < ns; e += TLC_NT) cS[(int64_t)c * tl.r + o0 + e] = sC[e];
}

constexpr int TLA_ROWS = 32;
__global__ __launch_bounds__(256) void k_tl_apply_mc(TwoLevelDev tl, const float* __restrict__ cS, float* __restrict__ dS, double* __restrict__ rho) {
  const int c = blockIdx.y, t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int r = tl.r;
  const float* __restrict__ cc = cS + (int64_t)c * r;
  const int e0 = blockIdx.x * TLA_ROWS + w * (TLA_ROWS / 4);
  float d[TLA_ROWS / 4];
#pragma unroll
  for (int i = 0; i < TLA_ROWS / 4; ++i) d[i] = 0.f;
  for (int q0 = 0; q0 < r; q0 += 64) {             // per 64 columns: the wave's 8 rows of N + c_S as 9 independent loads
    const int q = q0 + lane;
    const bool ok = q < r;
    const float cq = ok ? cc[q] : 0.f;
    float nv[TLA_ROWS / 4];
#pragma unroll
    for (int i = 0; i < TLA_ROWS / 4; ++i) {
      const int e = e0 + i;
      nv[i] = (ok && e < r) ? tl.N[(int64_t)e * r + q] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < TLA_ROWS / 4; ++i) d[i] += nv[i] * cq;
  }
  float rs = 0.f;
#pragma unroll
  for (int i = 0; i < TLA_ROWS / 4; ++i) {
    const float v = wave_reduce_sum<float>(d[i]);
    const int e = e0 + i;
    if (lane == 0 && e < r) {
      dS[(int64_t)c * r + e] = v;
      rs += cc[e] * v;
    }
  }
  if (rho != nullptr && lane == 0 && rs != 0.f) unsafeAtomicAdd(rho + c, (double)rs);
}

// ------------------------------------- slab, fp32 on the matrix cores, many columns ---
// k_spec_slab_mfma for multi-column solves (predictive variances, probe solves).  There a block per (slab, half, column)
// is 2 g0 k blocks of 115 KB LDS -- one per CU at a time, each a serial chain of [5 matrix loads, 4 products, 4 barriers]
// (8.6 us; 215 us for 64 columns at 50^3, the forward half computed twice).  Here a block owns slab i0 for a strided set of
// columns: the four eigenvector images are loaded ONCE, the forward products P2, P3 run once per column and both output halves
// are formed from the C3 fragment kept in registers (6 products per column instead of 8), and the next column's slab is
// fetched into registers while the current one is in the matrix cores.
//   LDS: bufA, bufB, sV1, sV2 (stride LDT), sB1, sB2 per half (stride LDN): 115 KB, 150 KB with a generalized eigenbasis.
// TL: the two-level block (see k_tl_coef_mc / k_tl_apply_mc, which have run on `src` and left d = N c_S in dS [k][r]).
template <int KS, int VW, int NW, bool TL>
__global__ __launch_bounds__(64 * NW) void k_spec_slab_mfma_mc(GridDev<float> G, const float* __restrict__ V1, const float* __restrict__ V2,
                                                           const float* __restrict__ Z1, const float* __restrict__ Z2,
                                                           const float* __restrict__ evals, float kscale, float shift,
                                                           const float* __restrict__ src, float* __restrict__ dst, int k, double* __restrict__ rho,
                                                           TwoLevelDev tl, const float* __restrict__ dS) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double s_red[16];
  __shared__ float sE[128];                       // eigenvalues of dims 1 | 2, zero padded to 64 each
  float* bufA = reinterpret_cast<float*>(smem);   // [64][80]: X, later scaled C3 (stride LDN)
  float* bufB = bufA + 64 * SPEC_LDT;             // [64][80]: C2 (stride LDN), later C5 (stride LDT)
  float* sV1 = bufB + 64 * SPEC_LDT;              // [b][x], stride LDT
  float* sV2 = sV1 + 64 * SPEC_LDT;               // [b][y], stride LDT
  float* sB1[2], *sB2[2];                         // backward images of half 0 (t: Z when generalized) and half 1 (y: V)
  sB1[1] = sV2 + 64 * SPEC_LDT;                   // bV1[x][b], stride LDN
  sB2[1] = sB1[1] + 64 * SPEC_LDN;                // bV2[y][b], stride LDN
  const bool alt = Z1 != V1 || Z2 != V2;          // grid-uniform
  sB1[0] = alt ? sB2[1] + 64 * SPEC_LDN : sB1[1];
  sB2[0] = alt ? sB1[0] + 64 * SPEC_LDN : sB2[1];
  const int g0 = G.g[0], g1 = G.g[1], g2 = G.g[2], m = G.m;
  const int i0 = blockIdx.x;
  constexpr int RT = NW == 4 ? 2 : 1;         // 16-row tiles per wave (NW = 8: 16 x 32 strips, as k_spec_slab_mfma)
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wr = w >> 1, wc = w & 1;
  int c = blockIdx.y;
  if (c >= k) return;
  SpecTile<VW, 64 * NW> tX;
  {
    SpecTile<VW, 64 * NW> tV1, tV2, tB1, tB2;
    tX.issue(src + (int64_t)c * m + (int64_t)i0 * g1 * g2, g1, g2);
    tV1.issue(V1, g1, g1);
    tV2.issue(V2, g2, g2);
    tB1.issue(V1, g1, g1);
    tB2.issue(V2, g2, g2);
    if (t < 128) {
      const int q = t & 63;
      float ev = 0.f;
      if (t < 64 ? q < g1 : q < g2) ev = evals[t < 64 ? g0 + q : g0 + g1 + q];
      sE[t] = ev;
    }
    tX.commit(bufA, SPEC_LDT);
    tV1.commit(sV1, SPEC_LDT);
    tV2.commit(sV2, SPEC_LDT);
    tB1.commit(sB1[1], SPEC_LDN);
    tB2.commit(sB2[1], SPEC_LDN);
    if (alt) {
      tB1.issue(Z1, g1, g1);
      tB2.issue(Z2, g2, g2);
      tB1.commit(sB1[0], SPEC_LDN);
      tB2.commit(sB2[0], SPEC_LDN);
    }
  }
  const float l0 = kscale * evals[i0];
  // (TL) selection masks of this slab's rows and, per row, the block index of its first selected mode (block order = raster order)
  __shared__ unsigned long long sMask[TL ? 64 : 1];
  __shared__ int sPre[TL ? 64 : 1];
  if constexpr (TL) {
    if (t < 64) {
      const int o0 = tl.off[i0];
      const unsigned long long mx = tl.off[i0 + 1] > o0 ? tl.mask[i0 * 64 + t] : 0ull;
      int v = __popcll(mx);
      const int own = v;
#pragma unroll
      for (int dlt = 1; dlt < 64; dlt <<= 1) {
        const int up = __shfl_up(v, dlt);
        if (t >= dlt) v += up;
      }
      sMask[t] = mx;
      sPre[t] = o0 + v - own;
    }
  }
  __syncthreads();
  spec_f32x4 acc[2][2];
  const int l15 = lane & 15, l4 = lane >> 4;
  auto store_tiles = [&](float* __restrict__ out, int ld) {       // C fragments -> out[row][col], row-major with stride ld
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
      for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(wr * 16 * RT + a * 16 + l4 * 4 + r) * ld + wc * 32 + cc * 16 + l15] = acc[a][cc][r];
  };
  // spectral factors of this thread's 16 C3 entries: the same for every column
  // (TL) eidx >= 0: this entry is selected mode eidx of the block -- its factors become 1 / lambda (t-half) and 1 (y-half) of d
  float f1v[2][2][4], f2v[2][2][4];
  int eidx[2][2][4];
  bool any_sel = false;
#pragma unroll
  for (int a = 0; a < RT; ++a)
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      const float e2 = sE[64 + wc * 32 + cc * 16 + l15];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float lam = l0 * sE[wr * 16 * RT + a * 16 + l4 * 4 + r] * e2;
        const float f1 = __frcp_rn(1.f + shift * lam);
        f1v[a][cc][r] = f1;
        f2v[a][cc][r] = lam * f1;
        eidx[a][cc][r] = -1;
        if constexpr (TL) {
          const int x = wr * 16 * RT + a * 16 + l4 * 4 + r, y = wc * 32 + cc * 16 + l15;
          const unsigned long long mx = sMask[x];
          if ((mx >> y) & 1ull) {
            eidx[a][cc][r] = sPre[x] + __popcll(mx & ((1ull << y) - 1ull));
            f1v[a][cc][r] = 1.f / lam;
            f2v[a][cc][r] = 1.f;
            any_sel = true;
          }
        }
      }
    }
  for (; c < k; c += gridDim.y) {
    const int cn = c + gridDim.y;
    const bool more = cn < k;                     // block-uniform
    if (more) tX.issue(src + (int64_t)cn * m + (int64_t)i0 * g1 * g2, g1, g2);
    float dv[2][2][4];
    if constexpr (TL) {
      if (any_sel) {
#pragma unroll
        for (int a = 0; a < RT; ++a)
#pragma unroll
          for (int cc = 0; cc < 2; ++cc)
#pragma unroll
            for (int r = 0; r < 4; ++r) dv[a][cc][r] = eidx[a][cc][r] >= 0 ? dS[(int64_t)c * tl.r + eidx[a][cc][r]] : 0.f;
      }
    }
    // P2: A = V1^T (sV1 [b][x]), B = X (bufA [b][y])  -> C2 natural (bufB, stride LDN)
    spec_mfma_product<false, false, KS, RT>(sV1, bufA, wr, wc, lane, acc);
    store_tiles(bufB, SPEC_LDN);
    __syncthreads();
    // P3: A = C2 (bufB natural), B = V2 (sV2 [b][y]) -> C3, kept in registers for both halves
    spec_mfma_product<true, false, KS, RT>(bufB, sV2, wr, wc, lane, acc);
    spec_f32x4 c3[2][2];
    float rho_lane = 0.f;   // 16 terms per lane in fp32, the cross-lane / cross-block sum in fp64
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
      for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[a][cc][r];
          bool sel = false;
          if constexpr (TL) sel = any_sel && eidx[a][cc][r] >= 0;
          if (sel) v = dv[a][cc][r];              // (its rho term c_S . N c_S was added by k_tl_apply_mc)
          else rho_lane += f2v[a][cc][r] * v * v; // r^T P r in the eigenbasis (padding: lam = 0)
          c3[a][cc][r] = v;
          acc[a][cc][r] = v * f1v[a][cc][r];
        }
    store_tiles(bufA, SPEC_LDN);                  // bufA (X) was last read by P2, a barrier ago
    if (rho != nullptr) {
      const double tot = block_reduce_sum((double)rho_lane, s_red);
      if (t == 0) unsafeAtomicAdd(rho + c, tot);
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      // P5: A = scaled C3 (bufA natural), B = bV2^T (sB2 [y][b]) -> C5 [b = i1'][y = i2] (bufB, stride LDT)
      spec_mfma_product<true, true, KS, RT>(bufA, sB2[h], wr, wc, lane, acc);
      store_tiles(bufB, SPEC_LDT);
      __syncthreads();
      // bufA is free now: half 0 -> the y-half's scaled C3; half 1 -> the next column's slab
      if (h == 0) {
#pragma unroll
        for (int a = 0; a < RT; ++a)
#pragma unroll
          for (int cc = 0; cc < 2; ++cc)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[a][cc][r] = c3[a][cc][r] * f2v[a][cc][r];
        store_tiles(bufA, SPEC_LDN);
      } else if (more) {
        tX.commit(bufA, SPEC_LDT);
      }
      // P6: A = bV1 (sB1 natural), B = C5 (bufB [b][y]) -> global
      spec_mfma_product<true, false, KS, RT>(sB1[h], bufB, wr, wc, lane, acc);
      float* __restrict__ os = dst + ((int64_t)h * k + c) * m + (int64_t)i0 * g1 * g2;
#pragma unroll
      for (int a = 0; a < RT; ++a)
#pragma unroll
        for (int cc = 0; cc < 2; ++cc)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int x = wr * 16 * RT + a * 16 + l4 * 4 + r, y = wc * 32 + cc * 16 + l15;
            if (x < g1 && y < g2) os[x * g2 + y] = acc[a][cc][r];
          }
      __syncthreads();                            // bufB (C5) read by P6; bufA written above
    }
  }
}

// ----------------------------------------------- mode 0, fp32 on the matrix cores ---
// dst[c][x, s] = sum_b A[x][b] src[c][b, s] for a tile of 32 fibres s, A = V0^T (forward) or V0 (backward):
// one 64 x 32 x (4 KS) product per block, the 4 waves own 32 x 16 output sub-tiles (2 MFMA tiles each).
//   MODE 0: plain store (+ optional dots[c - dot_c0] += rvec . dst for c >= dot_c0)       -- k_spec_mode0
//   MODE 1: CG direction update folded into the store, 2k columns                         -- k_spec_mode0_bwd_updp
//   MODE 2: forward product of the residual with the PREVIOUS iteration's vector update folded into the tile load
//           (apply != 0:  alpha = rho(it-1) / p.Hp(it-1);  u += alpha p;  z += alpha pt;  r -= alpha (pt + sum_ch part[ch]);
//           rn(it) += |r|^2 -- k_pcg_update_x; every element of r belongs to exactly one (fibre tile, column) block).
//           With 256 threads a thread updates 2 x 4 elements: all 2 x 13 vector loads of a thread are independent and
//           in flight together (the 128-thread register-tile predecessor serialised 4 such rounds and lost to a
//           separate update launch; this one saves that launch).
// (Several consecutive columns per block -- the 10 KB factor image loaded once per block instead of once per 6.4 KB tile --
// were tried for the 64-column solves: 8 columns per block make the forward / backward kernels 131 / 104 us instead of
// 94 / 79 us; the many small blocks are what keeps enough loads in flight.)
template <int KS, int VW, int MODE>
__global__ __launch_bounds__(256) void k_spec_mode0_mfma(GridDev<float> G, const float* __restrict__ Va, const float* __restrict__ Vb, int split,
                                                         int transposed, const float* __restrict__ src, float* __restrict__ dst,
                                                         const float* __restrict__ rvec, int dot_c0, double* __restrict__ dots, int k, int it,
                                                         float* __restrict__ p, float* __restrict__ pt, PcgScal S, int apply, double tol2,
                                                         float* __restrict__ part, int nch, int zl, float* __restrict__ u, float* __restrict__ z) {
  __shared__ __attribute__((aligned(16))) float sF[64 * SPEC_LDT];    // V0 image, row stride LDT (forward) or LDN (backward)
  __shared__ __attribute__((aligned(16))) float sIn[64 * SPEC_LDT];   // src tile [b][s], 32 of LDT columns used
  __shared__ double s_red[16];
  const int g0 = G.g[0], Sf = G.stride[0], m = G.m;
  const int cc = blockIdx.y;
  // Fibre tiles, XCD-contiguous (round 6): the launch pads grid.x to 8 * per, workgroup b (on XCD b % 8 for every blockIdx.y, since the row of
  // the grid is a multiple of 8 long) takes tile (b % 8) * per + b / 8.  A tile row is 32 floats = 128 bytes at a 4 Sf-byte row stride that is not
  // a multiple of 128 (50^3: 10 000), so every row segment straddles two cache lines and shares each with the neighbouring tile: with tiles dealt
  // round-robin the two halves of every line of every vector were fetched (and partially written) by two different XCDs.
  const int ntile = (Sf + 31) >> 5, per = (ntile + 7) >> 3;
  const int tile = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  if (tile >= ntile) return;
  const int s0 = tile * 32;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wr = w >> 1, wc = w & 1;
  const int l15 = lane & 15, l4 = lane >> 4;
  const float* __restrict__ V0;
  int c = cc;
  float bt = 0.f;
  if (MODE == 1) {                       // columns [0, k): pt-half with Z0 (= Va); [k, 2k): p-half with X0 (= Vb)
    const int half = cc / k;
    c = cc - half * k;
    V0 = half == 0 ? Va : Vb;
    if (it > 0) {
      const double den = S.rho(it - 1)[c];
      bt = (float)(den > 0 ? S.rho(it)[c] / den : 0.0);
    }
  } else {
    V0 = cc < split ? Va : Vb;           // generalized eigenbasis: t-half and y-half use different factors
  }
  SpecTile<VW> tV;
  tV.issue(V0, g0, g0);
  float4 tin[2];
  float al = 0.f;
  if (MODE == 2 && apply == 1) {
    const double den = S.php_sum(it - 1, cc);
    if (blockIdx.x == 0) pcg_dot_clear(S.php(it), cc, 1, S.k);   // ring entry the SpMV of this iteration accumulates into
    if (pcg_active(S, it - 1, cc, tol2) && den > 0) al = (float)(S.rho(it - 1)[cc] / den);
  }
  float rn_part = 0.f, rhs_part = 0.f;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int b = (t >> 3) + 32 * q, q4 = (t & 7) * 4;
    const bool ok = b < g0 && s0 + q4 < Sf;                  // Sf % 4 == 0 (precondition of the fused path)
    const int64_t e = (int64_t)cc * m + (ok ? (int64_t)b * Sf + s0 + q4 : 0);
    tin[q] = *reinterpret_cast<const float4*>(src + e);
    if (!ok) tin[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (MODE == 2 && apply == 2 && ok) {
      // iteration 0 of a solve whose residual was carried over: this sweep over r also forms ||r||^2 and ||rhs||^2
      // (k_pcg_init's job; `p` carries the right-hand side here) -- one launch less per streaming step
      const float4 f4 = *reinterpret_cast<const float4*>(p + e), rv = tin[q];
      rhs_part += f4.x * f4.x + f4.y * f4.y + f4.z * f4.z + f4.w * f4.w;
      rn_part += rv.x * rv.x + rv.y * rv.y + rv.z * rv.z + rv.w * rv.w;
    }
    if (MODE == 2 && apply == 1 && ok) {
      const int64_t km = (int64_t)S.k * m;
      const float4 pv = *reinterpret_cast<const float4*>(p + e), ptv = *reinterpret_cast<const float4*>(pt + e);
      float4 uv = *reinterpret_cast<const float4*>(u + e), zv = *reinterpret_cast<const float4*>(z + e);
      float4 pp[8];
#pragma unroll
      for (int ch = 0; ch < 8; ++ch)
        if (ch < nch) pp[ch] = *reinterpret_cast<const float4*>(part + (int64_t)ch * km + e);
      if (zl) *reinterpret_cast<float4*>(part + (int64_t)(nch - 1) * km + e) = make_float4(0.f, 0.f, 0.f, 0.f);   // consumed: re-zero
      float4 hv = ptv;
#pragma unroll
      for (int ch = 0; ch < 8; ++ch)
        if (ch < nch) { hv.x += pp[ch].x; hv.y += pp[ch].y; hv.z += pp[ch].z; hv.w += pp[ch].w; }
      uv.x += al * pv.x; uv.y += al * pv.y; uv.z += al * pv.z; uv.w += al * pv.w;
      zv.x += al * ptv.x; zv.y += al * ptv.y; zv.z += al * ptv.z; zv.w += al * ptv.w;
      float4 rv = tin[q];
      rv.x -= al * hv.x; rv.y -= al * hv.y; rv.z -= al * hv.z; rv.w -= al * hv.w;
      rn_part += rv.x * rv.x + rv.y * rv.y + rv.z * rv.z + rv.w * rv.w;
      *reinterpret_cast<float4*>(u + e) = uv;
      *reinterpret_cast<float4*>(z + e) = zv;
      *reinterpret_cast<float4*>(const_cast<float*>(src) + e) = rv;
      tin[q] = rv;
    }
  }
  if (MODE == 2 && apply) {                                  // block-uniform
    const double tot = block_reduce_sum((double)rn_part, s_red);
    if (t == 0) unsafeAtomicAdd(S.rn(it) + cc, tot);
    if (apply == 2) {
      const double tot0 = block_reduce_sum((double)rhs_part, s_red);
      if (t == 0) unsafeAtomicAdd(S.rn0() + cc, tot0);
    }
  }
  tV.commit(sF, transposed ? SPEC_LDN : SPEC_LDT);
#pragma unroll
  for (int q = 0; q < 2; ++q) *reinterpret_cast<float4*>(sIn + ((t >> 3) + 32 * q) * SPEC_LDT + (t & 7) * 4) = tin[q];
  __syncthreads();
  spec_f32x4 acc[2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[a][r] = 0.f;
  {
    const float* qb = sIn + l4 * SPEC_LDT + wc * 16 + l15;
    float af[KS][2], bf[KS];
    if (transposed) {                    // A[x][b] = V0[x][b], lanes walk a column (stride LDN)
      const float* qa = sF + (wr * 32 + l15) * SPEC_LDN + l4;
#pragma unroll
      for (int i = 0; i < KS; ++i) { af[i][0] = qa[4 * i]; af[i][1] = qa[4 * i + 16 * SPEC_LDN]; }
    } else {                             // A[x][b] = V0[b][x], 16-lane groups read a row (stride LDT)
      const float* qa = sF + l4 * SPEC_LDT + wr * 32 + l15;
#pragma unroll
      for (int i = 0; i < KS; ++i) { af[i][0] = qa[4 * i * SPEC_LDT]; af[i][1] = qa[4 * i * SPEC_LDT + 16]; }
    }
#pragma unroll
    for (int i = 0; i < KS; ++i) bf[i] = qb[4 * i * SPEC_LDT];
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][0], bf[i], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][1], bf[i], acc[1], 0, 0, 0);
    }
  }
  const int sidx = s0 + wc * 16 + l15;
  const bool dot = MODE == 0 && dots != nullptr && cc >= dot_c0;   // block-uniform
  float dpart = 0.f;
  if (sidx < Sf) {
    float* __restrict__ tgt = MODE == 1 ? (cc < k ? pt : p) + (int64_t)c * m : dst + (int64_t)cc * m;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int x = wr * 32 + a * 16 + l4 * 4 + r;
        if (x < g0) {
          const int64_t e = (int64_t)x * Sf + sidx;
          float o = acc[a][r];
          if (MODE == 1 && it > 0) o += bt * tgt[e];
          tgt[e] = o;
          if (dot) dpart += rvec[(int64_t)(cc - dot_c0) * m + e] * o;
        }
      }
  }
  if (dot) {
    const double tot = block_reduce_sum((double)dpart, s_red);
    if (t == 0) unsafeAtomicAdd(dots + (cc - dot_c0), tot);
  }
}

// KS / VW dispatch shared by the fp32 MFMA kernels: inner dimension padded to 16 / 32 / 48 / 52 / 64;
// 8-byte tile loads when the matrix rows are 8-byte aligned (even g)
#define SPEC_DISPATCH_KS_VW(gmax, even, CALL) \
  do {                                        \
    if (even) {                               \
      if ((gmax) <= 16) CALL(4, 2);           \
      else if ((gmax) <= 32) CALL(8, 2);      \
      else if ((gmax) <= 48) CALL(12, 2);     \
      else if ((gmax) <= 52) CALL(13, 2);     \
      else CALL(16, 2);                       \
    } else {                                  \
      if ((gmax) <= 16) CALL(4, 1);           \
      else if ((gmax) <= 32) CALL(8, 1);      \
      else if ((gmax) <= 48) CALL(12, 1);     \
      else if ((gmax) <= 52) CALL(13, 1);     \
      else CALL(16, 1);                       \
    }                                         \
  } while (0)

template <typename real>
__global__ void k_spec_mode0_bwd_updp(GridDev<real> G, const real* __restrict__ X0, const real* __restrict__ Z0, const real* __restrict__ src, int k,
                                      int it, real* __restrict__ p, real* __restrict__ pt, PcgScal S);   // defined below

// (One launch per preconditioner application was tried at the end of round 2: the three kernels as phases of one launch of
// 158 co-resident blocks -- one per CU, the slab's 115 KB of LDS -- separated by two grid-wide barriers in the cooperative-
// groups pattern (block barrier, one thread's release fence + arrival on a counter + spin + acquire fence, block barrier).
// Correct -- same iterates, 299 tests green -- and SLOWER: 0.258 instead of 0.203 ms per streaming step, i.e. +22 us per
// iteration.  A launch boundary costs ~5.5 us here (the 4 us an empty kernel takes by the same clock + ~1.5 us between
// launches); a grid barrier across 8 XCDs costs more than twice that (two L2 write-back / invalidate fences and an atomic
// round trip through the memory side), and the phases run with one 256-thread block per CU.  Removed.)
// mode-0 launch: fp32 on the matrix cores (256 threads), fp64 on the register-tile kernels (128 threads)
template <typename real, bool DOT>
static int launch_mode0(const GridDev<real>& G, const real* Va, const real* Vb, int split, int transposed, const real* src, real* dst, int ncols,
                        const real* rvec, int dot_c0, double* dots, hipStream_t s) {
  const int g0 = G.g[0], Sf = G.stride[0];
  if constexpr (sizeof(real) == 4) {
    dim3 grd((unsigned)(8 * (((Sf + 31) / 32 + 7) / 8)), (unsigned)ncols);     // (padded: XCD-contiguous fibre tiles, see the kernel)
#define M0(KS, VW)                                                                                                                          \
  hipLaunchKernelGGL((k_spec_mode0_mfma<KS, VW, 0>), grd, dim3(256), 0, s, G, Va, Vb, split, transposed, src, dst, rvec, dot_c0,            \
                     DOT ? dots : (double*)nullptr, 0, 0, (float*)nullptr, (float*)nullptr, PcgScal{nullptr, 0, nullptr}, 0, 0.0,         \
                     (float*)nullptr, 0, 0, (float*)nullptr, (float*)nullptr)
    SPEC_DISPATCH_KS_VW(g0, g0 % 2 == 0, M0);
#undef M0
  } else {
    const int P0 = (g0 + 3) & ~3;
    const size_t sh0 = (size_t)(P0 * P0 + P0 * SPEC_ST) * sizeof(real);
    hipLaunchKernelGGL((k_spec_mode0<real, DOT>), dim3((unsigned)((Sf + SPEC_ST - 1) / SPEC_ST), (unsigned)ncols), dim3(128), sh0, s, G, Va, Vb, split,
                       transposed, src, dst, rvec, dot_c0, dots);
  }
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}

// forward mode 0 of the residual with the previous iteration's vector update folded in (fp32 only)
static int launch_mode0_fwd_upd(const GridDev<float>& G, const float* V0, float* r, float* dst, int k, int it, int apply, double tol2, float* p,
                                float* pt, float* part, int nch, int zl, float* u, float* z, PcgScal S, hipStream_t s) {
  const int g0 = G.g[0], Sf = G.stride[0];
  dim3 grd((unsigned)(8 * (((Sf + 31) / 32 + 7) / 8)), (unsigned)k);
#define M2(KS, VW)                                                                                                                        \
  hipLaunchKernelGGL((k_spec_mode0_mfma<KS, VW, 2>), grd, dim3(256), 0, s, G, V0, V0, 0, 0, (const float*)r, dst, (const float*)nullptr, 0, \
                     (double*)nullptr, k, it, p, pt, S, apply, tol2, part, nch, zl, u, z)
  SPEC_DISPATCH_KS_VW(g0, g0 % 2 == 0, M2);
#undef M2
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}

template <typename real>
static int launch_mode0_bwd_updp(const GridDev<real>& G, const real* X0, const real* Z0, const real* src, int k, int it, real* p, real* pt,
                                 PcgScal S, hipStream_t s) {
  const int g0 = G.g[0], Sf = G.stride[0];
  if constexpr (sizeof(real) == 4) {
    dim3 grd((unsigned)(8 * (((Sf + 31) / 32 + 7) / 8)), (unsigned)(2 * k));
#define M1(KS, VW)                                                                                                                       \
  hipLaunchKernelGGL((k_spec_mode0_mfma<KS, VW, 1>), grd, dim3(256), 0, s, G, Z0, X0, 0, 1, src, (float*)nullptr, (const float*)nullptr, 0, \
                     (double*)nullptr, k, it, p, pt, S, 0, 0.0, (float*)nullptr, 0, 0, (float*)nullptr, (float*)nullptr)
    SPEC_DISPATCH_KS_VW(g0, g0 % 2 == 0, M1);
#undef M1
  } else {
    const int P0 = (g0 + 3) & ~3;
    const size_t sh0 = (size_t)(P0 * P0 + P0 * SPEC_ST) * sizeof(real);
    hipLaunchKernelGGL((k_spec_mode0_bwd_updp<real>), dim3((unsigned)((Sf + SPEC_ST - 1) / SPEC_ST), (unsigned)(2 * k)), dim3(128), sh0, s, G, X0, Z0,
                       src, k, it, p, pt, S);
  }
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}

static int spec_cu_count() {
  static int n = 0;
  if (n == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
    else n = 256;
  }
  return n;
}

constexpr size_t SPEC_SLAB_MFMA_LDS = (size_t)(4 * 64 * SPEC_LDT + 2 * 64 * SPEC_LDN) * sizeof(float);

// slab launch: fp32 on the matrix cores, fp64 on the register-tile kernel
template <typename real>
static int launch_slab(const GridDev<real>& G, const real* V1, const real* V2, const real* Z1, const real* Z2, const real* evals, real kscale,
                       real shift, const real* src, real* dst, int k, double* rho, hipStream_t s, const wiski_twolevel* two_level = nullptr) {
  const int g0 = G.g[0], g1 = G.g[1], g2 = G.g[2];
  if (two_level && (sizeof(real) != 4 || k < 1 || two_level->r < 1 || two_level->r > SPEC_TL_MAXR || two_level->nslab < 1 || two_level->nslab > g0 ||
                    !two_level->d_mask || !two_level->d_off || !two_level->d_pos || !two_level->d_N || !two_level->d_cs))
    return WISKI_E_BADARG;
  // one column: the exchange spins on words other blocks of the SAME launch write: all 2 g0 blocks (one per CU, ~100 KB of LDS each) must
  // be resident.  Several columns: the block is applied around the slab launch (k_tl_coef_mc, k_tl_apply_mc) and needs its 2 k r scratch instead
  if (two_level && k == 1 && 2 * g0 > spec_cu_count()) return WISKI_E_BADARG;
  if (two_level && k > 1 && (!two_level->d_mc || two_level->mc_cols < k)) return WISKI_E_BADARG;
  if constexpr (sizeof(real) == 4) {
    const int gm = g1 > g2 ? g1 : g2;
    const bool even = g1 % 2 == 0 && g2 % 2 == 0;      // 8-byte loads need 8-byte aligned rows
    TwoLevelDev tl{};
    if (two_level) {
      static std::atomic<unsigned> tl_epoch{0};            // one number per application, process-wide, never 0
      unsigned ep = ++tl_epoch;
      if (ep == 0) ep = ++tl_epoch;
      tl = TwoLevelDev{two_level->r, two_level->nslab, (const unsigned long long*)two_level->d_mask, two_level->d_off, two_level->d_pos,
                       two_level->d_N, (unsigned long long*)two_level->d_cs, ep};
    }
#define SLAB_MFMA2(KS, VW)                                                                                                                     \
  do {                                                                                                                                         \
    static bool lds_set = false;   /* > 48 KB of dynamic LDS needs an opt-in per kernel */                                                      \
    if (!lds_set) {                                                                                                                            \
      for (const void* fp : {(const void*)k_spec_slab_mfma<KS, VW, 4, false>, (const void*)k_spec_slab_mfma<KS, VW, 8, false>,                \
                             (const void*)k_spec_slab_mfma<KS, VW, 4, true>, (const void*)k_spec_slab_mfma<KS, VW, 8, true>})                  \
        if (hipFuncSetAttribute(fp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SPEC_SLAB_MFMA_LDS) != hipSuccess)                        \
          return WISKI_E_LAUNCH;                                                                                                               \
      lds_set = true;                                                                                                                          \
    }                                                                                                                                          \
    if (two_level && slab_waves == 8)                                                                                                          \
      hipLaunchKernelGGL((k_spec_slab_mfma<KS, VW, 8, true>), dim3((unsigned)g0, 2, 1), dim3(512), SPEC_SLAB_MFMA_LDS, s, G, V1, V2, Z1,       \
                         Z2, evals, kscale, shift, src, dst, k, rho, tl);                                                                      \
    else if (two_level)                                                                                                                        \
      hipLaunchKernelGGL((k_spec_slab_mfma<KS, VW, 4, true>), dim3((unsigned)g0, 2, 1), dim3(256), SPEC_SLAB_MFMA_LDS, s, G, V1, V2, Z1,       \
                         Z2, evals, kscale, shift, src, dst, k, rho, tl);                                                                      \
    else if (slab_waves == 8)                                                                                                                  \
      hipLaunchKernelGGL((k_spec_slab_mfma<KS, VW, 8, false>), dim3((unsigned)g0, 2, (unsigned)k), dim3(512), SPEC_SLAB_MFMA_LDS, s, G, V1,    \
                         V2, Z1, Z2, evals, kscale, shift, src, dst, k, rho, tl);                                                              \
    else                                                                                                                                       \
      hipLaunchKernelGGL((k_spec_slab_mfma<KS, VW, 4, false>), dim3((unsigned)g0, 2, (unsigned)k), dim3(256), SPEC_SLAB_MFMA_LDS, s, G, V1,    \
                         V2, Z1, Z2, evals, kscale, shift, src, dst, k, rho, tl);                                                              \
  } while (0)
#define SLAB_MFMA(KS)              \
  do {                             \
    if (even) SLAB_MFMA2(KS, 2);   \
    else SLAB_MFMA2(KS, 1);        \
  } while (0)
    constexpr int slab_waves = 8;                      // waves per block of the slab kernels (4 measured slower: DESIGN 3.3)
    // three or more columns: blocks that own a slab for a strided set of columns (about one block per CU)
    if (two_level ? k >= 2 : k >= 3) {
      const bool alt = Z1 != V1 || Z2 != V2;
      float* tl_c = two_level ? two_level->d_mc : nullptr;                                       // c_S [k][r]
      const float* tl_d = two_level ? two_level->d_mc + (int64_t)two_level->mc_cols * two_level->r : nullptr;   // N c_S [k][r]
      if (two_level) {
        const size_t lb = tl_coef_mc_lds(g1, g2);
        static size_t lb_set = 0;
        if (lb > lb_set) {
          if (hipFuncSetAttribute((const void*)k_tl_coef_mc, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb) != hipSuccess) return WISKI_E_LAUNCH;
          lb_set = lb;
        }
        hipLaunchKernelGGL(k_tl_coef_mc, dim3((unsigned)two_level->nslab, (unsigned)k), dim3(TLC_NT), lb, s, G, V1, V2, src, tl, tl_c);
        hipLaunchKernelGGL(k_tl_apply_mc, dim3((unsigned)((two_level->r + TLA_ROWS - 1) / TLA_ROWS), (unsigned)k), dim3(256), 0, s, tl,
                           (const float*)tl_c, const_cast<float*>(tl_d), rho);
      }
      const size_t lds = SPEC_SLAB_MFMA_LDS + (alt ? (size_t)2 * 64 * SPEC_LDN * sizeof(float) : 0);
      int nb = spec_cu_count() / g0;
      nb = nb < 1 ? 1 : (nb > k ? k : nb);
      // even out the columns per block: the fewest blocks that keep the longest chain
      const int per = (k + nb - 1) / nb;
      nb = (k + per - 1) / per;
#define SLAB_MC2(KS, VW)                                                                                                                          \
  do {                                                                                                                                            \
    static size_t lds_set = 0;                                                                                                                    \
    if (lds > lds_set) {                                                                                                                          \
      for (const void* fp : {(const void*)k_spec_slab_mfma_mc<KS, VW, 4, false>, (const void*)k_spec_slab_mfma_mc<KS, VW, 8, false>,             \
                             (const void*)k_spec_slab_mfma_mc<KS, VW, 4, true>, (const void*)k_spec_slab_mfma_mc<KS, VW, 8, true>})               \
        if (hipFuncSetAttribute(fp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return WISKI_E_LAUNCH;                   \
      lds_set = lds;                                                                                                                              \
    }                                                                                                                                             \
    if (two_level && slab_waves == 8)                                                                                                             \
      hipLaunchKernelGGL((k_spec_slab_mfma_mc<KS, VW, 8, true>), dim3((unsigned)g0, (unsigned)nb), dim3(512), lds, s, G, V1, V2, Z1, Z2, evals,   \
                         kscale, shift, src, dst, k, rho, tl, tl_d);                                                     \
    else if (two_level)                                                                                                                           \
      hipLaunchKernelGGL((k_spec_slab_mfma_mc<KS, VW, 4, true>), dim3((unsigned)g0, (unsigned)nb), dim3(256), lds, s, G, V1, V2, Z1, Z2, evals,   \
                         kscale, shift, src, dst, k, rho, tl, tl_d);                                                     \
    else if (slab_waves == 8)                                                                                                                     \
      hipLaunchKernelGGL((k_spec_slab_mfma_mc<KS, VW, 8, false>), dim3((unsigned)g0, (unsigned)nb), dim3(512), lds, s, G, V1, V2, Z1, Z2, evals,  \
                         kscale, shift, src, dst, k, rho, tl, (const float*)nullptr);                                                             \
    else                                                                                                                                          \
      hipLaunchKernelGGL((k_spec_slab_mfma_mc<KS, VW, 4, false>), dim3((unsigned)g0, (unsigned)nb), dim3(256), lds, s, G, V1, V2, Z1, Z2, evals,  \
                         kscale, shift, src, dst, k, rho, tl, (const float*)nullptr);                                                             \
  } while (0)
#define SLAB_MC(KS)              \
  do {                           \
    if (even) SLAB_MC2(KS, 2);   \
    else SLAB_MC2(KS, 1);        \
  } while (0)
      if (gm <= 16) SLAB_MC(4);
      else if (gm <= 32) SLAB_MC(8);
      else if (gm <= 48) SLAB_MC(12);
      else if (gm <= 52) SLAB_MC(13);
      else SLAB_MC(16);
#undef SLAB_MC
#undef SLAB_MC2
      return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
    }
    if (gm <= 16) SLAB_MFMA(4);          // inner dimension padded to 16 / 32 / 48 / 52 / 64
    else if (gm <= 32) SLAB_MFMA(8);
    else if (gm <= 48) SLAB_MFMA(12);
    else if (gm <= 52) SLAB_MFMA(13);
    else SLAB_MFMA(16);
#undef SLAB_MFMA
#undef SLAB_MFMA2
  } else {
    const int P1 = (g1 + 3) & ~3, P2 = (g2 + 3) & ~3;
    const int PM = P1 > P2 ? P1 : P2;
    const size_t sh1 = (size_t)(2 * PM * PM + 2 * (P1 * P1 + P2 * P2) + P1 + P2) * sizeof(real);
    static size_t slab_lds_set = 0;   // raise the dynamic-LDS limit once
    if (sh1 > 48 * 1024 && sh1 > slab_lds_set) {
      if (hipFuncSetAttribute((const void*)k_spec_slab<real>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh1) != hipSuccess)
        return WISKI_E_LAUNCH;
      slab_lds_set = sh1;
    }
    hipLaunchKernelGGL((k_spec_slab<real>), dim3((unsigned)g0, 2, (unsigned)k), dim3(256), sh1, s, G, V1, V2, Z1, Z2, evals, kscale, shift, src, dst,
                       k, rho);
  }
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}

// ------------------------------------------- fused CG back end (mode 0) ---
// Backward mode 0 on 2k columns with the direction update folded into the store:
//   columns [0,k):  t = V0 (.)  ->  pt = t + beta pt      columns [k,2k):  y = V0 (.)  ->  p = y + beta p
//   beta = rho(it)/rho(it-1) (0 at it == 0); t and y themselves are never written.   -- k_pcg_update_p
template <typename real>
__global__ __launch_bounds__(128) void k_spec_mode0_bwd_updp(GridDev<real> G, const real* __restrict__ X0, const real* __restrict__ Z0,
                                                            const real* __restrict__ src, int k, int it, real* __restrict__ p,
                                                            real* __restrict__ pt, PcgScal S) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ST = SPEC_ST;
  const int g0 = G.g[0], Sf = G.stride[0], m = G.m;
  const int P0 = (g0 + 3) & ~3;
  real* sF = reinterpret_cast<real*>(smem);
  real* sIn = sF + P0 * P0;
  const int cc = blockIdx.y;
  const int half = cc / k, c = cc - half * k;
  const int s0 = blockIdx.x * ST;
  const real* __restrict__ V0 = half == 0 ? Z0 : X0;
  const real* __restrict__ sc = src + (int64_t)cc * m;
  double beta = 0;
  if (it > 0) {
    const double den = S.rho(it - 1)[c];
    beta = den > 0 ? S.rho(it)[c] / den : 0;
  }
  const real bt = (real)beta;
  MatrixLoad<real, 128> ml;
  ml.issue(V0, g0);
  constexpr int NV = (64 * (ST / 4) + 127) / 128;
  V4<real> tin[NV];
#pragma unroll
  for (int itv = 0; itv < NV; ++itv) {
    const int t = threadIdx.x + itv * 128;
    const int b = t / (ST / 4), q4 = (t - b * (ST / 4)) * 4;
    V4<real> z4;
    z4.v[0] = z4.v[1] = z4.v[2] = z4.v[3] = (real)0;
    tin[itv] = (b < g0 && s0 + q4 < Sf) ? lds_read4<real>(sc + (int64_t)b * Sf + s0 + q4) : z4;
  }
  for (int e = threadIdx.x; e < P0 * P0; e += 128) sF[e] = (real)0;
  __syncthreads();
  ml.commit(sF, g0, P0);
#pragma unroll
  for (int itv = 0; itv < NV; ++itv) {
    const int t = threadIdx.x + itv * 128;
    const int b = t / (ST / 4), q4 = (t - b * (ST / 4)) * 4;
    if (b < P0) {
      real* d = sIn + b * ST + q4;
      d[0] = tin[itv].v[0]; d[1] = tin[itv].v[1]; d[2] = tin[itv].v[2]; d[3] = tin[itv].v[3];
    }
  }
  __syncthreads();
  const int nty = ST / 4, ntx = P0 / 4;
  const int t = threadIdx.x;
  if (t < ntx * nty) {
    const int tx = t / nty, ty = t - tx * nty;
    real acc[4][4];
    tile_product_t<real>(sF, P0, sIn, ST, P0, tx, ty, acc);
    const int sidx = s0 + 4 * ty;
    if (sidx < Sf) {
      real* __restrict__ tgt = (half == 0 ? pt : p) + (int64_t)c * m;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int x = 4 * tx + i;
        if (x < g0) {
          const int64_t e = (int64_t)x * Sf + sidx;
          real o[4] = {acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
          if (it > 0) {
            const V4<real> old = lds_read4<real>(tgt + e);
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] += bt * old.v[q];
          }
          if constexpr (sizeof(real) == 4) {
            *reinterpret_cast<float4*>(tgt + e) = make_float4(o[0], o[1], o[2], o[3]);
          } else {
            *reinterpret_cast<double2*>(tgt + e) = make_double2(o[0], o[1]);
            *reinterpret_cast<double2*>(tgt + e + 2) = make_double2(o[2], o[3]);
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------ host side ---
template <typename real>
bool spectral_fused_ok(const GridDev<real>& G) {
  if (G.d != 3) return false;
  for (int q = 0; q < 3; ++q)
    if (G.g[q] > 64) return false;
  if (G.stride[0] % 4 != 0) return false;   // 16-byte slab / fibre-tile loads
  return true;
}

// ty = [t | y] (2k columns), w0 = scratch of k*m reals; rho[c] += r[c].y[c]
template <typename real>
int launch_spectral_fused(const GridDev<real>& G, const real* evec, const real* evec2, const real* evals, real kscale, real shift, const real* r,
                          int k, real* w0, real* w1 /* 2k*m scratch */, real* ty, double* rho, hipStream_t s) {
  const int g0 = G.g[0], g1 = G.g[1], g2 = G.g[2];
  if (!evec2) evec2 = evec;
  const real* V0 = evec;
  const real* V1 = evec + g0 * g0;
  const real* V2 = V1 + g1 * g1;
  const real* Z0 = evec2;
  const real* Z1 = evec2 + g0 * g0;
  const real* Z2 = Z1 + g1 * g1;
  // forward mode 0: w0 = V0^T r
  if (int rc = launch_mode0<real, false>(G, V0, V0, 0, 0, r, w0, k, (const real*)nullptr, 0, (double*)nullptr, s)) return rc;
  // slab: forward modes 1,2 + scaling + backward modes 2,1 -> w1 = [half 0 | half 1]
  if (int rc = launch_slab<real>(G, V1, V2, Z1, Z2, evals, kscale, shift, (const real*)w0, w1, k, (double*)nullptr, s)) return rc;
  // backward mode 0 on 2k columns, rho += r . y for the second half
  return launch_mode0<real, true>(G, Z0, V0, k, 1, (const real*)w1, ty, 2 * k, r, k, rho, s);
}

// One CG iteration's preconditioner + vector updates in three launches: mode-0 forward (fp32: with the previous
// iteration's u / z / r update folded into its tile load when `apply`), slab (+ rho), mode-0 backward with p / pt updated
// in the store.  fp64 callers apply the update with a separate k_pcg_update_x launch and pass apply = 0.
template <typename real>
int launch_spectral_fused_cg(const GridDev<real>& G, const real* evec, const real* evec2, const real* evals, real kscale, real shift, real* r,
                             int k, real* w0, real* w1, int it, int apply, double tol2, real* p, real* pt, real* part, int nch, int zl, real* u,
                             real* z, PcgScal S, hipStream_t s, const real* rhs0, const wiski_twolevel* two_level) {
  const int g0 = G.g[0], g1 = G.g[1];
  if (!evec2) evec2 = evec;
  const real* V0 = evec;
  const real* V1 = evec + g0 * g0;
  const real* V2 = V1 + g1 * g1;
  const real* Z0 = evec2;
  const real* Z1 = evec2 + g0 * g0;
  const real* Z2 = Z1 + g1 * g1;
  if constexpr (sizeof(real) == 4) {
    // rhs0 (iteration 0 of a solve with a carried residual): the forward sweep also forms ||r||^2 and ||rhs||^2; the kernel's
    // `p` argument -- not read at it = 0 -- carries the right-hand side for THIS launch only (the backward kernel below writes p)
    if (rhs0 && apply) return WISKI_E_BADARG;
    if (int rc = launch_mode0_fwd_upd(G, V0, r, w0, k, it, rhs0 ? 2 : apply, tol2, rhs0 ? const_cast<float*>(rhs0) : p, pt, part, nch, zl, u, z, S, s))
      return rc;
  } else {
    if (apply || rhs0) return WISKI_E_BADARG;
    if (int rc = launch_mode0<real, false>(G, V0, V0, 0, 0, (const real*)r, w0, k, (const real*)nullptr, 0, (double*)nullptr, s)) return rc;
  }
  if (int rc = launch_slab<real>(G, V1, V2, Z1, Z2, evals, kscale, shift, (const real*)w0, w1, k, S.rho(it), s, two_level)) return rc;
  return launch_mode0_bwd_updp<real>(G, V0, Z0, (const real*)w1, k, it, p, pt, S, s);
}

template bool spectral_fused_ok<float>(const GridDev<float>&);
template int launch_spectral_fused_cg<float>(const GridDev<float>&, const float*, const float*, const float*, float, float, float*, int, float*,
                                             float*, int, int, double, float*, float*, float*, int, int, float*, float*, PcgScal, hipStream_t,
                                             const float*, const wiski_twolevel*);
template int launch_spectral_fused_cg<double>(const GridDev<double>&, const double*, const double*, const double*, double, double, double*, int,
                                              double*, double*, int, int, double, double*, double*, double*, int, int, double*, double*, PcgScal,
                                              hipStream_t, const double*, const wiski_twolevel*);
template bool spectral_fused_ok<double>(const GridDev<double>&);
template int launch_spectral_fused<float>(const GridDev<float>&, const float*, const float*, const float*, float, float, const float*, int, float*,
                                          float*, float*, double*, hipStream_t);
template int launch_spectral_fused<double>(const GridDev<double>&, const double*, const double*, const double*, double, double, const double*, int,
                                           double*, double*, double*, double*, hipStream_t);
