// a2+a3+a4+a5: streaming accumulation of the additive sufficient statistics
//   b  = W^T D^-1 y   (interpolation_cache, BFN:46,160)
//   A  = W^T D^-1 W   (WtW, BFN:50-53; URLT:58 `tensor + V V^T`), block stencil
//   c  = y^T D^-1 y   (response_cache, BFN:45), ld = logdet D (BFN:55)
// The reference densifies W^T (m x q) and adds a dense m x m outer product per
// update; here each streamed point touches exactly its 4^d x 4^d stencil block.
#include "wiski_common.h"

// The half-stencil atomics of k_scatter_stats_sym.  WISKI_SCATTER_ATOMIC_MOD (timing builds; the round-4 job script is in the git history): cache-policy bits on the
// atomic -- does any of them leave A_h better placed for the SpMV that follows?  0 = the default (what ships).  Measured (bench traces, SpMV
// dispatches right after the absorb / absorb kernel): default 18.0-18.2 us / 66.6-67.4 us, sc1 18.1-18.4 / 67.7, nt 18.3-18.4 / 69.3, sc1 nt 18.2-18.3 / 68.7 -- no.
#ifndef WISKI_SCATTER_ATOMIC_MOD
#define WISKI_SCATTER_ATOMIC_MOD 0
#endif
__device__ __forceinline__ void stencil_atomic(float* p, float v) {
#if WISKI_SCATTER_ATOMIC_MOD == 1
  asm volatile("global_atomic_add_f32 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
#elif WISKI_SCATTER_ATOMIC_MOD == 2
  asm volatile("global_atomic_add_f32 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
#elif WISKI_SCATTER_ATOMIC_MOD == 3
  asm volatile("global_atomic_add_f32 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
#else
  unsafeAtomicAdd(p, v);
#endif
}
__device__ __forceinline__ void stencil_atomic(double* p, double v) { unsafeAtomicAdd(p, v); }
#include <atomic>
#include <cstdlib>

// GRP = min(T, 64) lanes cooperate on one point (lane <-> tap a); each lane
// walks all taps b and issues fire-and-forget L2 atomics on A_st[o(a,b)][idx_a].
// Tap values are exchanged through LDS (broadcast reads, conflict-free).
// This is the full offset-major form A_st[o][i] (all 7^d offsets, T^2 atomics per point); the model itself
// keeps the symmetric half (k_scatter_stats_sym below: T(T+1)/2 atomics, ~5x faster).
// stats[0] += sum wb y^2, stats[1] += sum log(noise) over the points inside the grid.  Atomics of many blocks on one address
// serialise at the memory side (~12 ns each): with one pair per block the 1 024 blocks of a q = 4 096 absorb spent 25 us of
// their 87 us queueing on these two doubles.  So a few designated blocks sweep the points once more (x, y, wb, noise: 24 B per
// point) and issue one pair each.
// strides (in elements) between the outputs of a batched launch; all zero for a single output
struct ScatterBatch {
  int64_t y_stride = 0, w_stride = 0, vec_stride = 0, A_stride = 0;
};

template <typename real, int D>
__device__ __forceinline__ void scatter_stats_pass(const GridDev<real>& G, const real* __restrict__ x, const real* __restrict__ y,
                                                   const real* __restrict__ wb, const real* __restrict__ noise, int64_t n,
                                                   double* __restrict__ stats, double* s_red) {
  int64_t want = n / 512;
  want = want < 1 ? 1 : (want > 64 ? 64 : want);
  const int ns = (int64_t)gridDim.x < want ? (int)gridDim.x : (int)want;
  if ((int)blockIdx.x >= ns) return;                       // block-uniform
  double c_acc = 0, ld_acc = 0;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)ns * blockDim.x) {
    real xp[D], w[D][4];
    int j0[D];
#pragma unroll
    for (int q = 0; q < D; ++q) xp[q] = x[p * D + q];
    if (point_stencil<real, D>(G, xp, j0, w)) {
      const double yp = (double)y[p];
      c_acc += yp * yp * (double)wb[p];
      ld_acc += log((double)noise[p]);
    }
  }
  const double c_tot = block_reduce_sum(c_acc, s_red);
  const double ld_tot = block_reduce_sum(ld_acc, s_red);
  if (threadIdx.x == 0 && (c_tot != 0 || ld_tot != 0)) {
    unsafeAtomicAdd(stats + 0, c_tot);
    unsafeAtomicAdd(stats + 1, ld_tot);
  }
}

template <typename real, int D>
__global__ __launch_bounds__(256) void k_scatter_stats(GridDev<real> G, const real* __restrict__ x, const real* __restrict__ y,
                                                       const real* __restrict__ wa, const real* __restrict__ wb,
                                                       const real* __restrict__ noise, int64_t n, real* __restrict__ b,
                                                       real* __restrict__ A_st, double* __restrict__ stats, int32_t* __restrict__ err,
                                                       real* __restrict__ cnt) {
  constexpr int T = 1 << (2 * D);
  constexpr int GRP = T < 64 ? T : 64;      // lanes per point
  constexpr int PPB = 256 / GRP;            // points per block pass
  constexpr int TPL = T / GRP;              // taps per lane
  __shared__ real s_val[PPB][T];
  __shared__ double s_red[16];
  const int sub = threadIdx.x % GRP;
  const int loc = threadIdx.x / GRP;
  int R = 1;
#pragma unroll
  for (int q = 0; q < D; ++q) R *= 7;
  const int center = (R - 1) / 2;
  bool bad = false;

  for (int64_t base = (int64_t)blockIdx.x * PPB; base < n; base += (int64_t)gridDim.x * PPB) {
    const int64_t p = base + loc;
    const bool valid = p < n;
    int j0[D];
    real w[D][4];
    real yp = 0, wap = 0, wbp = 0;
    if (valid) {
      real xp[D];
#pragma unroll
      for (int q = 0; q < D; ++q) xp[q] = x[p * D + q];
      // a point outside the grid raises the flag and contributes nothing at all (zero weights; no y^2 / log-noise
      // term either, so a caller that catches the error keeps statistics that agree with A and b)
      const bool inside = point_stencil<real, D>(G, xp, j0, w);
      if (!inside) {
        bad = true;
        if (sub == 0) atomicAdd(err, 2);      // bits 1..: number of training points dropped (bit 0: any point outside)
      }
      yp = y[p];
      wap = wa[p];
      wbp = wb[p];
    } else {
#pragma unroll
      for (int q = 0; q < D; ++q) {
        j0[q] = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) w[q][c] = 0;
      }
    }
    int idx_a[TPL], code_a[TPL];
    real val_a[TPL];
#pragma unroll
    for (int t = 0; t < TPL; ++t) {
      const int a = sub + t * GRP;
      int flat = 0, code = 0;
      real v = (real)1;
#pragma unroll
      for (int q = 0; q < D; ++q) {
        const int c = (a >> (2 * (D - 1 - q))) & 3;
        flat += (j0[q] + c) * G.stride[q];
        code = code * 7 + c;
        v *= w[q][c];
      }
      idx_a[t] = flat;
      code_a[t] = code;
      val_a[t] = v;
      s_val[loc][a] = v;
    }
    __syncthreads();
    if (valid) {
#pragma unroll
      for (int t = 0; t < TPL; ++t) {
        if (val_a[t] != (real)0) {
          atomic_add_real(b + idx_a[t], val_a[t] * yp * wbp);
          if (cnt) atomic_add_real(cnt + idx_a[t], val_a[t] * wap);     // row sums of the increment (preconditioner density model)
          if (A_st) {
            const real va = val_a[t] * wap;
            real* __restrict__ Arow = A_st + idx_a[t];
            const int obase = center - code_a[t];
#pragma unroll 4
            for (int bb0 = 0; bb0 < T; ++bb0) {
              // skew the last-dim digit of b by this lane's own last-dim digit: the 4 lanes that share
              // (c0, c1) of tap a then target the same stencil offset at 4 consecutive rows, i.e. one
              // 16-byte run of A -- up to 4 lane-atomics per L2 transaction instead of 1.
              const int bb = (bb0 & ~3) | ((bb0 + (sub + t * GRP)) & 3);
              int codeb = 0;
#pragma unroll
              for (int q = 0; q < D; ++q) codeb = codeb * 7 + ((bb >> (2 * (D - 1 - q))) & 3);
              const real vb = s_val[loc][bb];
              if (vb != (real)0) atomic_add_real(Arow + (int64_t)(obase + codeb) * G.m, va * vb);
            }
          }
        }
      }
    }
    __syncthreads();
  }
  scatter_stats_pass<real, D>(G, x, y, wb, noise, n, stats, s_red);
  if (bad) atomicOr(err, 1);
}

// Symmetric half-stencil accumulation (the model's native W^T D^-1 W storage), "row-interleaved" layout:
// with P = the leading d-1 stencil digits of an offset and s its innermost digit, only offsets >= centre
// are kept, grouped by g = P - P_centre:
//   group 0      :  A_h[4 i + (s - 3)]                 s = 3..6   (4 reals per row; s = 3 is the diagonal)
//   group g >= 1 :  A_h[(7 g - 3) m + 7 i + s]         s = 0..6   (7 reals per row)
// -- (7^d + 1)/2 * m reals in total, the same as a row-major [(7^d+1)/2][m] half stencil.  The layout is
// chosen for this kernel: one wave per point, lane = (a2, b2, pair slot) with a2/b2 the innermost tap
// digits of taps a/b, looping over the (prefix_a <= prefix_b) pairs four at a time.  The 16 (a2, b2)
// combinations of a pair land in one 88-byte span (rows i..i+3 x 7 slots), so a wave instruction touches
// ~9 cache lines with ~7 lanes each instead of 16 lines with 4 lanes (offset-major layout) -- measured
// 72 us vs 208 us per 4096 uniform points at 50^3 (the memory-side atomic units are transaction-bound).
// The stencil SpMV re-tiles the 7-wide rows through LDS (solve.hip).
template <typename real, int D>
__global__ __launch_bounds__(256) void k_scatter_stats_sym(GridDev<real> G, const real* __restrict__ x, const real* __restrict__ y,
                                                           const real* __restrict__ wa, const real* __restrict__ wb,
                                                           const real* __restrict__ noise, int64_t n, real* __restrict__ b,
                                                           real* __restrict__ A, double* __restrict__ stats, int32_t* __restrict__ err,
                                                           real* __restrict__ cnt, const real* __restrict__ u, real* __restrict__ res,
                                                           real* __restrict__ mean_out = nullptr, uint32_t* __restrict__ z1 = nullptr,
                                                           int64_t n1 = 0, uint32_t* __restrict__ z2 = nullptr, int64_t n2 = 0,
                                                           const long long* __restrict__ guard = nullptr, long long guard_expect = 0,
                                                           int g_lo = 0, int g_hi = 1 << 30, ScatterBatch bt = ScatterBatch{}) {
  // several independent outputs in ONE launch (BFN:37-55 carries num_outputs as a batch dimension): blockIdx.y = output;
  // same points, per-output targets / weights / statistics at fixed strides (a weight stride of 0 shares one weight vector)
  {
    const int64_t o = blockIdx.y;
    y += o * bt.y_stride; wa += o * bt.w_stride; wb += o * bt.w_stride; noise += o * bt.w_stride;
    b += o * bt.vec_stride; stats += 2 * o;
    if (A) A += o * bt.A_stride;
    if (cnt) cnt += o * bt.vec_stride;
    if (u) u += o * bt.vec_stride;
    if (res) res += o * bt.vec_stride;
  }
  // [g_lo, g_hi): the stencil groups this replica owns (wiski_shard: a rank of a stencil-sharded step scatters the tap pairs
  // of ITS groups only -- 1 / N of the atomics per point; b, cnt, res and the statistics stay replicated)
  // speculative launch behind a solve whose convergence poll the host has not read yet (wiski_pcg_async_guard): the poll's
  // publishing block decides on the device whether this absorb happens
  if (guard && *guard != guard_expect) return;
  // optional (wiski_scatter_stats_step): zero two word arrays on the way -- the scalar block and the accumulated partial
  // vector of the solve that follows in the same streaming step
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n1; e += (int64_t)gridDim.x * blockDim.x) z1[e] = 0u;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n2; e += (int64_t)gridDim.x * blockDim.x) z2[e] = 0u;
  constexpr int T = 1 << (2 * D);
  constexpr int TP = T / 4;                      // tap prefixes (leading d-1 digits)
  constexpr int NPAIR = TP * (TP + 1) / 2;       // prefix pairs with code(pb) >= code(pa)
  constexpr int TPL = T > 64 ? T / 64 : 1;       // taps per lane when filling the per-point tables
  __shared__ real s_val[4][T];
  __shared__ int s_idx[4][T];
  __shared__ int s_pair[NPAIR];                  // pa | pb << 8 | g << 16
  __shared__ int s_scan[4];
  __shared__ double s_red[16];
  const int lane = threadIdx.x & 63, loc = threadIdx.x >> 6;
  int npair = 0;
  {  // compact list of the valid prefix pairs (block-cooperative stream compaction)
    int running = 0;
    for (int base = 0; base < TP * TP; base += 256) {
      const int idx = base + threadIdx.x;
      const int pa = idx / TP, pb = idx % TP;
      int ca = 0, cb = 0;
#pragma unroll
      for (int q = 0; q < D - 1; ++q) {
        ca = ca * 7 + ((pa >> (2 * (D - 2 - q))) & 3);
        cb = cb * 7 + ((pb >> (2 * (D - 2 - q))) & 3);
      }
      const bool ok = idx < TP * TP && cb >= ca && cb - ca >= g_lo && cb - ca < g_hi;
      const unsigned long long mask = __ballot(ok);
      if (lane == 0) s_scan[loc] = __popcll(mask);
      __syncthreads();
      int off = running;
      for (int w = 0; w < loc; ++w) off += s_scan[w];
      if (ok) s_pair[off + __popcll(mask & ((1ull << lane) - 1ull))] = pa | (pb << 8) | ((cb - ca) << 16);
      running += s_scan[0] + s_scan[1] + s_scan[2] + s_scan[3];
      __syncthreads();
    }
    npair = running;                              // block-uniform: NPAIR for the whole stencil, fewer for a shard
  }
  bool bad = false;
  const int64_t m = G.m;
  for (int64_t base = (int64_t)blockIdx.x * 4; base < n; base += (int64_t)gridDim.x * 4) {
    const int64_t p = base + loc;
    const bool valid = p < n;
    int j0[D];
    real w[D][4];
    real yp = 0, wap = 0, wbp = 0;
    if (valid) {
      real xp[D];
#pragma unroll
      for (int q = 0; q < D; ++q) xp[q] = x[p * D + q];
      // a point outside the grid raises the flag and contributes nothing at all (zero weights; no y^2 / log-noise
      // term either, so a caller that catches the error keeps statistics that agree with A and b)
      const bool inside = point_stencil<real, D>(G, xp, j0, w);
      if (!inside) {
        bad = true;
        if (lane == 0 && blockIdx.y == 0) atomicAdd(err, 2);      // bits 1..: number of training points dropped (bit 0: any point outside)
      }
      yp = y[p];
      wap = wa[p];
      wbp = wb[p];
    } else {
#pragma unroll
      for (int q = 0; q < D; ++q) {
        j0[q] = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) w[q][c] = 0;
      }
    }
    int flat_t[TPL];
    real val_t[TPL];
    real wu = (real)0;                            // this lane's share of w_p . u
#pragma unroll
    for (int t = 0; t < TPL; ++t) {
      const int a = lane + t * 64;
      flat_t[t] = 0;
      val_t[t] = (real)0;
      if (a < T) {
        int flat = 0;
        real v = (real)1;
#pragma unroll
        for (int q = 0; q < D; ++q) {
          const int c = (a >> (2 * (D - 1 - q))) & 3;
          flat += (j0[q] + c) * G.stride[q];
          v *= w[q][c];
        }
        s_val[loc][a] = v;
        s_idx[loc][a] = flat;
        flat_t[t] = flat;
        val_t[t] = v;
        if (u && v != (real)0) wu += v * u[flat];
      }
    }
    // residual carry-over (optional): res += W^T (wb y - wa (W u)) keeps res = b - z - A u exact under the
    // increment (b, A) += (W^T wb y, W^T wa W), so the next warm-started solve needs no A u product
    // w_p . u is also the predictive mean of the point under the posterior BEFORE this update (u = the current posterior mean
    // on the grid): mean_out makes the separate gather launch of a streaming step unnecessary
    real innov = yp * wbp;
    if (u) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) wu += __shfl_xor(wu, off, 64);
      innov -= wap * wu;
      if (mean_out && lane == 0 && valid) mean_out[p] = wu;
    }
#pragma unroll
    for (int t = 0; t < TPL; ++t) {
      if (valid && val_t[t] != (real)0) {
        atomic_add_real(b + flat_t[t], val_t[t] * yp * wbp);
        if (cnt) atomic_add_real(cnt + flat_t[t], val_t[t] * wap);     // row sums of the increment (preconditioner density model)
        if (res) atomic_add_real(res + flat_t[t], val_t[t] * innov);
      }
    }
    __syncthreads();
    if (valid && A) {
      const int a2 = lane & 3, b2 = (lane >> 2) & 3, ps = lane >> 4;
#pragma unroll 2
      for (int t0 = 0; t0 < npair; t0 += 4) {
        const int t = t0 + ps;
        if (t < npair) {
          const int pk = s_pair[t];
          const int g = pk >> 16;
          const int a = (pk & 0xff) * 4 + a2;
          const real v = wap * s_val[loc][a] * s_val[loc][((pk >> 8) & 0xff) * 4 + b2];
          const int64_t row = s_idx[loc][a];
          if (g == 0) {
            if (b2 >= a2 && v != (real)0) stencil_atomic(A + row * 4 + (b2 - a2), v);
          } else if (v != (real)0) {
            stencil_atomic(A + (int64_t)(7 * g - 3) * m + row * 7 + (b2 - a2 + 3), v);
          }
        }
      }
    }
    __syncthreads();
  }
  scatter_stats_pass<real, D>(G, x, y, wb, noise, n, stats, s_red);
  if (bad) atomicOr(err, 1);
}

#include "scatter_owner.h"

// Unpacks the row-interleaved half stencil into a full offset-major stencil full[o][i] = A[i, i + off(o)]
// (diagnostics, tests, and models handed a full-stencil cache):
//   full[c + oh][i] += h(oh, i);  full[c - oh][i + off(oh)] += h(oh, i) (oh > 0);  h(oh, i) = 0.
// Every full entry is touched by exactly one (oh, i), so no atomics.
template <typename real>
__global__ __launch_bounds__(256) void k_stencil_expand_add(GridDev<real> G, real* __restrict__ half, real* __restrict__ full) {
  const int m = G.m, d = G.d;
  const int c = (G.R - 1) / 2;
  const int oh = blockIdx.y;
  // flat offset of stencil index o = c + oh
  int rem = c + oh, off = 0;
  for (int q = d - 1; q >= 0; --q) {
    off += (rem % 7 - 3) * G.stride[q];
    rem /= 7;
  }
  const int g = oh < 4 ? 0 : (oh - 4) / 7 + 1;
  real* __restrict__ h = oh < 4 ? half + oh : half + (int64_t)(7 * g - 3) * m + (oh - 4) % 7;
  const int hs = oh < 4 ? 4 : 7;
  real* __restrict__ fd = full + (int64_t)(c + oh) * m;
  real* __restrict__ fm = full + (int64_t)(c - oh) * m;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
    const real v = h[(int64_t)i * hs];
    if (v != (real)0) {
      fd[i] += v;
      if (oh > 0) {
        const int j = i + off;
        if (j >= 0 && j < m) fm[j] += v;
      }
      h[(int64_t)i * hs] = (real)0;
    }
  }
}

template <typename real>
static int expand_impl(const wiski_grid* grid, real* d_half, real* d_full, void* stream) {
  GridDev<real> G;
  int rc = make_grid_dev<real>(grid, &G);
  if (rc) return rc;
  if (!d_half || !d_full) return WISKI_E_BADARG;
  int bx = (G.m + 255) / 256;
  if (bx > 64) bx = 64;
  dim3 grd((unsigned)bx, (unsigned)((G.R + 1) / 2));
  hipLaunchKernelGGL((k_stencil_expand_add<real>), grd, dim3(256), 0, (hipStream_t)stream, G, d_half, d_full);
  WISKI_LAUNCH_CHECK();
  return WISKI_OK;
}

// batches at least this large take the owner-computes absorb when the caller provides its workspace (WISKI_OWNER_MIN_POINTS;
// 0 disables): its cost is a sweep over the touched part of A_h, that of the atomic form 19 ns per point
static int64_t owner_min_points() {
  static int64_t v = -1;
  if (v < 0) {
    const char* e = getenv("WISKI_OWNER_MIN_POINTS");
    // 50^3: 69 vs 79 us at 4096 uniform points (clustered: 80 vs 80), 255 vs 585 us at 32768.  At 4096 the streaming step gains
    // only 2 % (0.209 -> 0.205 ms): the rows written through the XCDs' L2s leave A_h less Infinity-Cache resident than memory-side
    // atomics do, and the SpMVs of the following solve slow down from 20.0 to 21.0 us -- so the default starts above that size
    v = e ? atoll(e) : 8192;
    if (v <= 0) v = (int64_t)1 << 62;
  }
  return v;
}

static int owner_ablate() { return 0; }

template <typename real>
static int scatter_impl(const wiski_grid* grid, const real* d_x, const real* d_y, const real* d_wa, const real* d_wb, const real* d_noise,
                        int64_t n, real* d_b, real* d_A_st, double* d_stats, int32_t* d_err, void* stream, bool half = false,
                        real* d_cnt = nullptr, const real* d_u = nullptr, real* d_res = nullptr, real* d_mean_out = nullptr,
                        void* z1 = nullptr, int64_t n1_bytes = 0, void* z2 = nullptr, int64_t n2_bytes = 0, const void* d_guard = nullptr,
                        int64_t guard_expect = 0, void* d_bin = nullptr, int64_t bin_bytes = 0, int g_lo = 0, int g_hi = 1 << 30, int nout = 1,
                        ScatterBatch bt = ScatterBatch{}) {
  GridDev<real> G;
  int rc = make_grid_dev<real>(grid, &G);
  if (rc) return rc;
  if (n == 0) return WISKI_OK;
  // argument validation comes first: the owner-computes branch below must not start mutating statistics on arguments the
  // atomic form would have refused
  if (!d_x || !d_y || !d_wa || !d_wb || !d_noise || !d_b || !d_stats || !d_err) return WISKI_E_BADARG;
  if (d_mean_out == nullptr && (d_u != nullptr) != (d_res != nullptr)) return WISKI_E_BADARG;
  if ((d_res && !d_u) || (d_mean_out && !d_u) || (d_u && !half)) return WISKI_E_BADARG;  // residual carry-over / mean: half-stencil form only
  if (d_guard && !half) return WISKI_E_BADARG;
  if ((n1_bytes | n2_bytes) & 3 || (n1_bytes && !z1) || (n2_bytes && !z2) || ((n1_bytes || n2_bytes) && !half)) return WISKI_E_BADARG;
  // large batches on a d = 3 half stencil with a binning workspace: the owner-computes form (scatter_owner.h).  Its LDS
  // accumulators ((g2 * 175 + 512) reals per block) must fit the device limit and the opt-in must succeed BEFORE the first
  // kernel of the pair is queued -- k_bin_points already updates statistics; otherwise the atomic form runs.
  const bool ranged = g_lo > 0 || g_hi < (1 << 30);      // a stencil shard: atomic form only (the owner form walks whole lines)
  if (ranged && !half) return WISKI_E_BADARG;
  if (nout < 1 || (nout > 1 && (!half || d_mean_out || n1_bytes || n2_bytes || d_guard))) return WISKI_E_BADARG;   // batched outputs: plain absorb only
  bool owner = !ranged && nout == 1 && half && G.d == 3 && d_bin && d_cnt && d_A_st && G.g[2] <= 64 && G.g[0] > 3 && G.g[1] > 3 && G.g[2] > 3 && n < (int64_t)1 << 31 &&
               n >= owner_min_points() && bin_bytes >= owner_work_bytes<real>(grid, n);
  const size_t owner_lds = ((size_t)G.g[2] * (172 + 3) + 512) * sizeof(real);      // accumulators + one scratch word per thread
  if (owner) {
    static int lds_max = -1;
    static size_t lds_set = 0;
    if (lds_max < 0) {
      int dev = 0, v = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) v = 64 * 1024;
      lds_max = v;
    }
    if (owner_lds > (size_t)lds_max) {
      owner = false;
    } else if (owner_lds > 48 * 1024 && owner_lds > lds_set) {
      if (hipFuncSetAttribute((const void*)k_owner_lines<real, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)owner_lds) != hipSuccess ||
          hipFuncSetAttribute((const void*)k_owner_lines<real, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)owner_lds) != hipSuccess ||
          hipFuncSetAttribute((const void*)k_owner_lines<real, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)owner_lds) != hipSuccess) {
        (void)hipGetLastError();
        owner = false;                               // the atomic form needs no opt-in
      } else {
        lds_set = owner_lds;
      }
    }
  }
  if (owner) {
    static std::atomic<unsigned> epoch_src{0};
    unsigned epoch = ++epoch_src;                    // never 0 (a zero-initialised head is "empty")
    if (!epoch) epoch = ++epoch_src;
    const int64_t ncell = (int64_t)(G.g[0] - 3) * (G.g[1] - 3) * (G.g[2] - 3);
    char* w = static_cast<char*>(d_bin);
    unsigned long long* head = reinterpret_cast<unsigned long long*>(w);
    int32_t* next = reinterpret_cast<int32_t*>(w + (ncell * 8 + 255) / 256 * 256);
    real* rec = reinterpret_cast<real*>(w + (ncell * 8 + 255) / 256 * 256 + (n * 4 + 255) / 256 * 256);
    int64_t nb = (n + 3) / 4;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL((k_bin_points<real>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, G, d_x, d_y, d_wa, d_wb, d_noise, n, d_stats, d_err,
                       d_u, d_res != nullptr ? 1 : 0, d_mean_out, head, next, rec, epoch, (uint32_t*)z1, n1_bytes / 4, (uint32_t*)z2, n2_bytes / 4,
                       (const long long*)d_guard, (long long)guard_expect);
    constexpr int owner_nt = 256;                    // threads per owner block (measured at 50^3, 4096 points: 69 us with 256, 88 us with 512)
    const size_t lds = owner_lds;
    const int abl = owner_ablate();
    hipLaunchKernelGGL((k_owner_lines<real, owner_nt>), dim3((unsigned)(G.g[0] * G.g[1])), dim3(owner_nt), lds, (hipStream_t)stream, G, d_A_st, d_b, d_cnt, d_res,
                       (const unsigned long long*)head, (const int32_t*)next, (const real*)rec, epoch, (const long long*)d_guard, (long long)guard_expect, abl);
    return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
  }
  const int grp = half ? 64 : (G.T < 64 ? G.T : 64);
  const int64_t ppb = 256 / grp;
  int64_t blocks = (n + ppb - 1) / ppb;
  if (blocks > 256 * 8) blocks = 256 * 8;
  dim3 grd((unsigned)blocks, (unsigned)nout);
#define CALL(DD)                                                                                                                              \
  do {                                                                                                                                        \
    if (half) hipLaunchKernelGGL((k_scatter_stats_sym<real, DD>), grd, dim3(256), 0, (hipStream_t)stream, G, d_x, d_y, d_wa, d_wb, d_noise, n, d_b, d_A_st, d_stats, d_err, d_cnt, d_u, d_res, d_mean_out, (uint32_t*)z1, n1_bytes / 4, (uint32_t*)z2, n2_bytes / 4, (const long long*)d_guard, (long long)guard_expect, g_lo, g_hi, bt); \
    else hipLaunchKernelGGL((k_scatter_stats<real, DD>), grd, dim3(256), 0, (hipStream_t)stream, G, d_x, d_y, d_wa, d_wb, d_noise, n, d_b, d_A_st, d_stats, d_err, d_cnt);   \
  } while (0)
  WISKI_DISPATCH_D(G.d, CALL)
#undef CALL
  WISKI_LAUNCH_CHECK();
  return WISKI_OK;
}

extern "C" {
int wiski_scatter_stats_f32(const wiski_grid* g, const float* x, const float* y, const float* wa, const float* wb, const float* noise, int64_t n, float* b, float* A, double* stats, int32_t* err, void* s) {
  return scatter_impl<float>(g, x, y, wa, wb, noise, n, b, A, stats, err, s);
}
int wiski_scatter_stats_f64(const wiski_grid* g, const double* x, const double* y, const double* wa, const double* wb, const double* noise, int64_t n, double* b, double* A, double* stats, int32_t* err, void* s) {
  return scatter_impl<double>(g, x, y, wa, wb, noise, n, b, A, stats, err, s);
}
int wiski_scatter_stats_sym_f32(const wiski_grid* g, const float* x, const float* y, const float* wa, const float* wb, const float* noise, int64_t n, float* b, float* A_half, double* stats, int32_t* err, void* s) {
  return scatter_impl<float>(g, x, y, wa, wb, noise, n, b, A_half, stats, err, s, true);
}
int wiski_scatter_stats_sym_f64(const wiski_grid* g, const double* x, const double* y, const double* wa, const double* wb, const double* noise, int64_t n, double* b, double* A_half, double* stats, int32_t* err, void* s) {
  return scatter_impl<double>(g, x, y, wa, wb, noise, n, b, A_half, stats, err, s, true);
}
int wiski_scatter_stats_cnt_f32(const wiski_grid* g, const float* x, const float* y, const float* wa, const float* wb, const float* noise, int64_t n, float* b, float* A, int32_t half, float* cnt, const float* u, float* res, double* stats, int32_t* err, void* s) {
  return scatter_impl<float>(g, x, y, wa, wb, noise, n, b, A, stats, err, s, half != 0, cnt, u, res);
}
int wiski_scatter_stats_cnt_f64(const wiski_grid* g, const double* x, const double* y, const double* wa, const double* wb, const double* noise, int64_t n, double* b, double* A, int32_t half, double* cnt, const double* u, double* res, double* stats, int32_t* err, void* s) {
  return scatter_impl<double>(g, x, y, wa, wb, noise, n, b, A, stats, err, s, half != 0, cnt, u, res);
}
int wiski_scatter_stats_step_f32(const wiski_grid* g, const float* x, const float* y, const float* wa, const float* wb, const float* noise, int64_t n, float* b, float* A_half, float* cnt, const float* u, float* res, float* mean_out, double* stats, int32_t* err, void* z1, int64_t n1, void* z2, int64_t n2, const void* guard, int64_t guard_expect, void* bin, int64_t bin_bytes, void* s) {
  return scatter_impl<float>(g, x, y, wa, wb, noise, n, b, A_half, stats, err, s, true, cnt, u, res, mean_out, z1, n1, z2, n2, guard, guard_expect, bin, bin_bytes);
}
int wiski_scatter_stats_step_f64(const wiski_grid* g, const double* x, const double* y, const double* wa, const double* wb, const double* noise, int64_t n, double* b, double* A_half, double* cnt, const double* u, double* res, double* mean_out, double* stats, int32_t* err, void* z1, int64_t n1, void* z2, int64_t n2, const void* guard, int64_t guard_expect, void* bin, int64_t bin_bytes, void* s) {
  return scatter_impl<double>(g, x, y, wa, wb, noise, n, b, A_half, stats, err, s, true, cnt, u, res, mean_out, z1, n1, z2, n2, guard, guard_expect, bin, bin_bytes);
}
int wiski_scatter_stats_step_sharded_f32(const wiski_grid* g, const float* x, const float* y, const float* wa, const float* wb, const float* noise, int64_t n, float* b, float* A_half, float* cnt, const float* u, float* res, float* mean_out, double* stats, int32_t* err, void* z1, int64_t n1, void* z2, int64_t n2, const void* guard, int64_t guard_expect, int32_t g_lo, int32_t g_hi, void* s) {
  return scatter_impl<float>(g, x, y, wa, wb, noise, n, b, A_half, stats, err, s, true, cnt, u, res, mean_out, z1, n1, z2, n2, guard, guard_expect, nullptr, 0, g_lo, g_hi);
}
int wiski_scatter_stats_step_sharded_f64(const wiski_grid* g, const double* x, const double* y, const double* wa, const double* wb, const double* noise, int64_t n, double* b, double* A_half, double* cnt, const double* u, double* res, double* mean_out, double* stats, int32_t* err, void* z1, int64_t n1, void* z2, int64_t n2, const void* guard, int64_t guard_expect, int32_t g_lo, int32_t g_hi, void* s) {
  return scatter_impl<double>(g, x, y, wa, wb, noise, n, b, A_half, stats, err, s, true, cnt, u, res, mean_out, z1, n1, z2, n2, guard, guard_expect, nullptr, 0, g_lo, g_hi);
}
int wiski_scatter_stats_multi_f32(const wiski_grid* g, const float* x, const float* y, const float* wa, const float* wb, const float* noise, int64_t n, int32_t nout, int64_t y_stride, int64_t w_stride, float* b, float* A_half, int64_t A_stride, float* cnt, const float* u, float* res, double* stats, int32_t* err, void* s) {
  ScatterBatch bt; bt.y_stride = y_stride; bt.w_stride = w_stride; bt.vec_stride = g ? (int64_t)g->g[0] * (g->d > 1 ? g->g[1] : 1) * (g->d > 2 ? g->g[2] : 1) * (g->d > 3 ? g->g[3] : 1) : 0; bt.A_stride = A_stride;
  return scatter_impl<float>(g, x, y, wa, wb, noise, n, b, A_half, stats, err, s, true, cnt, u, res, nullptr, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0, 0, 1 << 30, nout, bt);
}
int wiski_scatter_stats_multi_f64(const wiski_grid* g, const double* x, const double* y, const double* wa, const double* wb, const double* noise, int64_t n, int32_t nout, int64_t y_stride, int64_t w_stride, double* b, double* A_half, int64_t A_stride, double* cnt, const double* u, double* res, double* stats, int32_t* err, void* s) {
  ScatterBatch bt; bt.y_stride = y_stride; bt.w_stride = w_stride; bt.vec_stride = g ? (int64_t)g->g[0] * (g->d > 1 ? g->g[1] : 1) * (g->d > 2 ? g->g[2] : 1) * (g->d > 3 ? g->g[3] : 1) : 0; bt.A_stride = A_stride;
  return scatter_impl<double>(g, x, y, wa, wb, noise, n, b, A_half, stats, err, s, true, cnt, u, res, nullptr, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0, 0, 1 << 30, nout, bt);
}
int64_t wiski_scatter_bin_bytes(const wiski_grid* g, int64_t n, int32_t elem_size) {
  if (!g || g->d != 3 || n < 0 || (elem_size != 4 && elem_size != 8)) return -1;
  return elem_size == 4 ? owner_work_bytes<float>(g, n) : owner_work_bytes<double>(g, n);
}
int wiski_stencil_expand_add_f32(const wiski_grid* g, float* half, float* full, void* s) { return expand_impl<float>(g, half, full, s); }
int wiski_stencil_expand_add_f64(const wiski_grid* g, double* half, double* full, void* s) { return expand_impl<double>(g, half, full, s); }
}
