// Shared device helpers for libwiski_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/wiski.h"

#define WISKI_VERSION 1

// Grid geometry as a by-value kernel argument (lives in SGPRs / kernarg).
template <typename real>
struct GridDev {
  int d;
  int g[WISKI_MAX_DIM];
  int stride[WISKI_MAX_DIM];  // flat stride of dim q (dim 0 slowest)
  real g0[WISKI_MAX_DIM];
  real h[WISKI_MAX_DIM];
  real hi[WISKI_MAX_DIM];     // last grid point
  int m;                      // prod g
  int T;                      // 4^d taps
  int R;                      // 7^d stencil offsets
};

template <typename real>
static inline int make_grid_dev(const wiski_grid* grid, GridDev<real>* out) {
  if (!grid || grid->d < 1 || grid->d > WISKI_MAX_DIM) return WISKI_E_BADARG;
  GridDev<real> G;
  G.d = grid->d;
  int64_t m = 1;
  G.T = 1;
  G.R = 1;
  for (int q = 0; q < WISKI_MAX_DIM; ++q) {
    G.g[q] = q < grid->d ? grid->g[q] : 1;
    G.g0[q] = q < grid->d ? (real)grid->g0[q] : (real)0;
    G.h[q] = q < grid->d ? (real)grid->h[q] : (real)1;
    G.hi[q] = G.g0[q] + G.h[q] * (real)(G.g[q] - 1);
    if (q < grid->d) {
      if (grid->g[q] < 4) return WISKI_E_BADARG;
      m *= grid->g[q];
      G.T *= 4;
      G.R *= 7;
    }
  }
  if (m >= (int64_t)1 << 31) return WISKI_E_BADARG;
  G.m = (int)m;
  int s = 1;
  for (int q = WISKI_MAX_DIM - 1; q >= 0; --q) {
    G.stride[q] = s;
    s *= G.g[q];
  }
  *out = G;
  return WISKI_OK;
}

// Keys cubic convolution kernel (a = -0.5).
template <typename real>
__device__ __forceinline__ real keys_cubic(real s) {
  real a = s < (real)0 ? -s : s;
  real near = (((real)1.5 * a - (real)2.5) * a) * a + (real)1;
  real far = (((real)-0.5 * a + (real)2.5) * a - (real)4) * a + (real)2;
  return a <= (real)1 ? near : (a < (real)2 ? far : (real)0);
}

// d/ds of the Keys kernel.
template <typename real>
__device__ __forceinline__ real keys_cubic_deriv(real s) {
  const real a = s < (real)0 ? -s : s;
  const real sg = s < (real)0 ? (real)-1 : (real)1;
  const real near = ((real)4.5 * a - (real)5) * a;
  const real far = ((real)-1.5 * a + (real)5) * a - (real)4;
  return a <= (real)1 ? sg * near : (a < (real)2 ? sg * far : (real)0);
}

// 4-tap stencil of one coordinate: lowest tap index (or -1 if outside the
// grid) and weights.  Boundary cells collapse to a one-hot on the nearest of
// the first/last four grid points (gpytorch Interpolation.interpolate).
template <typename real>
__device__ __forceinline__ int dim_stencil(real x, real g0, real h, real hi, int g, real w[4]) {
  real u = (x - g0) / h;
  real fl = floor(u);
  real t = u - fl;
  int j0 = (int)fl - 1;
  bool oob = !(x >= g0 && x <= hi);
  w[0] = keys_cubic<real>(t + (real)1);
  w[1] = keys_cubic<real>(t);
  w[2] = keys_cubic<real>(t - (real)1);
  w[3] = keys_cubic<real>(t - (real)2);
  if (j0 < 0 || j0 > g - 4) {
    int base = j0 < 0 ? 0 : g - 4;
    int best = 0;
    real bd = (real)3.0e38;
    for (int c = 0; c < 4; ++c) {
      real dd = g0 + h * (real)(base + c) - x;
      dd = dd < (real)0 ? -dd : dd;
      if (dd < bd) { bd = dd; best = c; }
    }
    for (int c = 0; c < 4; ++c) w[c] = (c == best) ? (real)1 : (real)0;
    j0 = base;
  }
  return oob ? -1 : j0;
}

// Per-point stencil for all dims. Returns false (and leaves a safe stencil
// with zero weights) when the point is outside the grid.
template <typename real, int D>
__device__ __forceinline__ bool point_stencil(const GridDev<real>& G, const real* __restrict__ xp, int j0[D], real w[D][4]) {
  bool ok = true;
#pragma unroll
  for (int q = 0; q < D; ++q) {
    int j = dim_stencil<real>(xp[q], G.g0[q], G.h[q], G.hi[q], G.g[q], w[q]);
    if (j < 0) {
      ok = false;
      j = 0;
#pragma unroll
      for (int c = 0; c < 4; ++c) w[q][c] = (real)0;
    }
    j0[q] = j;
  }
  return ok;
}

__device__ __forceinline__ void atomic_add_real(float* p, float v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void atomic_add_real(double* p, double v) { unsafeAtomicAdd(p, v); }

// Sum over the 64 lanes of a wave; every lane receives the total.  Data-parallel primitives (DPP) instead of __shfl_down:
// a shuffle is a ds_bpermute on gfx9 (two for a double), i.e. a dependent trip through the LDS crossbar per step -- ~0.5 us
// at the very end of a kernel whose waves all finish together (the p . Ap epilogue of the SpMV: 19.2 vs 17.2 us).
// row_shr 1 / 2 / 4 / 8 build the 16-lane row totals in lane 15 of each row, row_bcast15 / row_bcast31 carry them to lane 63.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float wave_dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, true));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double wave_dpp_add(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, ROW_MASK, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xf, true);
  return v + __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}
template <typename T>
__device__ __forceinline__ T wave_reduce_sum(T v) {
  v = wave_dpp_add<0x111, 0xf>(v);   // row_shr:1
  v = wave_dpp_add<0x112, 0xf>(v);   // row_shr:2
  v = wave_dpp_add<0x114, 0xf>(v);   // row_shr:4
  v = wave_dpp_add<0x118, 0xf>(v);   // row_shr:8
  v = wave_dpp_add<0x142, 0xa>(v);   // row_bcast15 into rows 1, 3
  v = wave_dpp_add<0x143, 0xc>(v);   // row_bcast31 into rows 2, 3
  if constexpr (sizeof(T) == 8) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __builtin_bit_cast(T, ((long long)hi << 32) | (long long)(unsigned)lo);
  } else {
    return __builtin_bit_cast(T, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
  }
}

// Block-wide sum of a double; result valid in thread 0. `sm` needs >= 16 doubles.
__device__ __forceinline__ double block_reduce_sum(double v, double* sm) {
  v = wave_reduce_sum<double>(v);
  int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) sm[wid] = v;
  __syncthreads();
  double r = 0;
  if (threadIdx.x == 0) {
    int nw = (blockDim.x + 63) >> 6;
    for (int i = 0; i < nw; ++i) r += sm[i];
  }
  __syncthreads();
  return r;
}

#define WISKI_LAUNCH_CHECK()                          \
  do {                                                \
    if (hipGetLastError() != hipSuccess) return WISKI_E_LAUNCH; \
  } while (0)

#define WISKI_DISPATCH_D(d, CALL) \
  switch (d) {                    \
    case 1: { CALL(1); break; }   \
    case 2: { CALL(2); break; }   \
    case 3: { CALL(3); break; }   \
    case 4: { CALL(4); break; }   \
    default: return WISKI_E_BADARG; \
  }

// PCG scalar slots (double): [0..k) = ||rhs||^2 ; then per iteration slot it in
// [0, max_iter]: rho[k], php[k], rn[k].  rn of slot 0 = ||r0||^2.
// p.Hp is accumulated by one fp64 atomic per SpMV block.  With ~10^3 blocks finishing together, atomics
// on one address -- or on one 128-byte line -- serialise at the memory side (measured on a 20 us kernel:
// +16 us with a single address, +7 us with 16 addresses in one line, +0.8 us with 16 addresses on 16
// lines).  Each column therefore owns PCG_DOT_SLOTS addresses PCG_DOT_STRIDE doubles apart (block b adds
// to slot b % PCG_DOT_SLOTS) and the consumers sum them.  The slots live in a two-deep ring indexed by
// the iteration parity: the SpMV of iteration `it` accumulates into ring[it & 1]; the vector update that
// consumes it (k_pcg_update_x) clears ring[(it + 1) & 1], whose own reader ran one
// iteration earlier on the same stream.
constexpr int PCG_DOT_SLOTS = 16;
constexpr int PCG_DOT_STRIDE = 16;
constexpr int PCG_DOT_COL = PCG_DOT_SLOTS * PCG_DOT_STRIDE;   // doubles per column in one ring entry
// A replica is stencil-sharded when it shares the half stencil with other ranks -- or when it is the only rank of a real
// communicator (it then owns every group and its all-reduces are identities: the in-C RCCL branch of the sharded solve can be
// exercised on one device, tests/test_distributed_gpu.py).
static inline bool wiski_shard_active(const wiski_shard* s) { return s && (s->nranks > 1 || (s->nranks == 1 && s->comm)); }

struct PcgScal {
  double* base;
  int k;
  double* ring;     // [2][k][PCG_DOT_COL]
  __host__ __device__ double* rn0() const { return base; }
  __host__ __device__ double* rho(int it) const { return base + (int64_t)k * (1 + 2 * it); }
  __host__ __device__ double* rn(int it) const { return base + (int64_t)k * (2 + 2 * it); }
  __host__ __device__ double* php(int it) const { return ring + (int64_t)(it & 1) * k * PCG_DOT_COL; }   // [k][PCG_DOT_COL]
  __device__ double php_sum(int it, int c) const {
    const double* q = php(it) + (int64_t)c * PCG_DOT_COL;
    double t = 0;
#pragma unroll
    for (int j = 0; j < PCG_DOT_SLOTS; ++j) t += q[j * PCG_DOT_STRIDE];
    return t;
  }
  // + 1: the ticket counter of the publishing vector update (k_pcg_update_x), zeroed with the scalars
  static int64_t scalars(int k, int max_iter) { return (int64_t)k * (1 + 2 * (int64_t)(max_iter + 2)) + 1; }
  __host__ __device__ unsigned* ticket() const { return reinterpret_cast<unsigned*>(ring) - 2; }   // the last scalar before the ring
  static int64_t doubles(int k, int max_iter) { return scalars(k, max_iter) + 2 * (int64_t)k * PCG_DOT_COL; }
};

__device__ __forceinline__ void pcg_dot_add(double* dots, int col, double v) {
  unsafeAtomicAdd(dots + (int64_t)col * PCG_DOT_COL + (blockIdx.x & (PCG_DOT_SLOTS - 1)) * PCG_DOT_STRIDE, v);
}
__device__ __forceinline__ void pcg_dot_clear(double* dots_next, int col0, int ncol, int k) {
  for (int idx = threadIdx.x; idx < ncol * PCG_DOT_COL; idx += blockDim.x) {
    const int c = col0 + idx / PCG_DOT_COL;
    if (c < k) dots_next[(int64_t)c * PCG_DOT_COL + idx % PCG_DOT_COL] = 0.0;
  }
}

__device__ __forceinline__ bool pcg_active(const PcgScal& S, int it, int c, double tol2) {
  // column still iterating? (rn of the previous slot against the rhs norm)
  const double rn0 = S.rn0()[c];
  return rn0 > 0 && S.rn(it)[c] > tol2 * rn0;
}

// spectral.hip: fused Kronecker-eigenbasis preconditioner (d = 3)
template <typename real>
bool spectral_fused_ok(const GridDev<real>& G);
template <typename real>
int launch_spectral_fused(const GridDev<real>& G, const real* evec, const real* evec2, const real* evals, real kscale, real shift, const real* r,
                          int k, real* w0, real* w1, real* ty, double* rho, hipStream_t s);

// Fused CG-iteration front end (d = 3): [apply update_x(it-1)] + mode-0 fwd -> slab (+rho) -> mode-0 bwd (+update_p)
template <typename real>
int launch_spectral_fused_cg(const GridDev<real>& G, const real* evec, const real* evec2, const real* evals, real kscale, real shift, real* r,
                             int k, real* w0, real* w1, int it, int apply, double tol2, real* p, real* pt, real* part, int nch, int zl, real* u,
                             real* z, PcgScal S, hipStream_t s, const real* rhs0 = nullptr, const wiski_twolevel* two_level = nullptr);
