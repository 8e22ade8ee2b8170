// Owner-computes absorb for large batches (d = 3, symmetric half stencil).  Included by scatter_stats.hip.
//
// k_scatter_stats_sym issues T (T + 1) / 2 = 2080 fire-and-forget atomics per point; they execute at the memory side, which
// retires ~19.4 G cache-line transactions per second whatever their order or scope: 313 + ~60 transactions per
// point = 78 us per 4096 points, and proportionally more for the N q points every rank absorbs after a point exchange.
// Turned around, every ROW of A_h has one owner and nothing is atomic at the memory side:
//   k_bin_points   one wave per point: taps and weights, w_p . u (= the batch's predictive mean and the innovation of the
//                  residual carry), a 16-real record per point, and the point is pushed onto the list of its CELL (the 4x4x4
//                  block of nodes it touches) with one atomicExch on the cell's head word.  Heads carry the call's epoch in
//                  their upper half, so they are never reset.
//   k_owner_lines  one block per grid line (i0, i1, *): the points that touch the line sit in <= 16 (g2 - 3) cells; the block
//                  walks those lists, accumulates every point's contribution to the line's rows -- 172 half-stencil slots,
//                  b, cnt and res per row -- in LDS (LDS atomics), and adds the non-zero accumulators to global memory with
//                  plain, row-contiguous read-modify-writes.
// Measured at 50^3 fp32 (tools/owner_probe.py): 69 us per 4 096 uniform points (atomic form 79; clustered 80 / 80), 255 us per
// 32 768 (585) -- binning 8 us, then per line: heads + records fetched in two batched memory latencies, ~38 point visits of ~110
// instructions each (33 us in all), and the write-back of the non-zero accumulators (17 us).  At 32 768 points: binning 27 us,
// scan 13, visits 193 (20 per point, ~720 SIMD cycles each: the record broadcast and index set-up are paid again by each of the
// 16 lines a point touches), write-back 18.  History: LDS atomics and one
// (head, record) round trip per 64 candidates 140 us; quarter-of-the-line ownership per wave (no atomics), batched fetches,
// branch-free visits with one LDS round trip 61 us; 8 or 2 waves per block are slower (88 / 87 us at 4 096 points).
#pragma once

template <typename real>
struct OwnerRec {
  static constexpr int N = 16;   // reals per point: w0[4] w1[4] w2[4] | wa | wb y | innovation | (unused)
};

__device__ __forceinline__ float wave_readlane(float v, int src) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src)); }
__device__ __forceinline__ int wave_readlane(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ double wave_readlane(double v, int src) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), src), hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}
__device__ __forceinline__ void lds_atomic_add(float* p, float v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_atomic_add(double* p, double v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// bytes of the binning workspace for n points on grid G (heads | next | records); must be zero-initialised once by the caller
template <typename real>
static inline int64_t owner_work_bytes(const wiski_grid* g, int64_t n) {
  const int64_t ncell = (int64_t)(g->g[0] - 3) * (g->g[1] - 3) * (g->g[2] - 3);
  const int64_t heads = (ncell * 8 + 255) / 256 * 256, next = (n * 4 + 255) / 256 * 256;
  return heads + next + n * OwnerRec<real>::N * (int64_t)sizeof(real);
}

template <typename real>
__global__ __launch_bounds__(256) void k_bin_points(GridDev<real> G, const real* __restrict__ x, const real* __restrict__ y, const real* __restrict__ wa,
                                                    const real* __restrict__ wb, const real* __restrict__ noise, int64_t n,
                                                    double* __restrict__ stats, int32_t* __restrict__ err, const real* __restrict__ u, int carry,
                                                    real* __restrict__ mean_out, unsigned long long* __restrict__ head, int32_t* __restrict__ next,
                                                    real* __restrict__ rec, unsigned epoch, uint32_t* __restrict__ z1, int64_t n1,
                                                    uint32_t* __restrict__ z2, int64_t n2, const long long* __restrict__ guard, long long guard_expect) {
  if (guard && *guard != guard_expect) return;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n1; e += (int64_t)gridDim.x * blockDim.x) z1[e] = 0u;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n2; e += (int64_t)gridDim.x * blockDim.x) z2[e] = 0u;
  __shared__ double s_red[16];
  const int lane = threadIdx.x & 63, loc = threadIdx.x >> 6;
  const int nc1 = G.g[1] - 3, nc2 = G.g[2] - 3;
  bool bad = false;
  for (int64_t base = (int64_t)blockIdx.x * 4; base < n; base += (int64_t)gridDim.x * 4) {
    const int64_t p = base + loc;
    if (p >= n) continue;                        // wave-uniform
    real xp[3], w[3][4];
    int j0[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) xp[q] = x[p * 3 + q];
    const bool inside = point_stencil<real, 3>(G, xp, j0, w);
    if (!inside) {                               // flagged, contributes nothing (as k_scatter_stats_sym)
      bad = true;
      if (lane == 0) {
        atomicAdd(err, 2);
        next[p] = -2;
        if (mean_out) mean_out[p] = (real)0;
      }
      continue;
    }
    // (compile-time indices only: a lane-dependent index into w[][] would put the array into scratch memory -- 30 us)
    const int a = lane >> 4, b = (lane >> 2) & 3, c = lane & 3;
    const real wa_ = a == 0 ? w[0][0] : a == 1 ? w[0][1] : a == 2 ? w[0][2] : w[0][3];
    const real wb_ = b == 0 ? w[1][0] : b == 1 ? w[1][1] : b == 2 ? w[1][2] : w[1][3];
    const real wc_ = c == 0 ? w[2][0] : c == 1 ? w[2][1] : c == 2 ? w[2][2] : w[2][3];
    const real v = wa_ * wb_ * wc_;
    const int64_t flat = (int64_t)(j0[0] + a) * G.stride[0] + (int64_t)(j0[1] + b) * G.stride[1] + (j0[2] + c);
    real wu = (u && v != (real)0) ? v * u[flat] : (real)0;
    if (u) wu = wave_reduce_sum<real>(wu);     // every lane holds the total
    {
      real wl = (real)0;
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (lane == 4 * q + k) wl = w[q][k];
      if (lane < 12) rec[p * OwnerRec<real>::N + lane] = wl;
    }
    if (lane == 12) {
      const real wap = wa[p], by = y[p] * wb[p];
      real* r = rec + p * OwnerRec<real>::N;
      r[12] = wap;
      r[13] = by;
      r[14] = carry ? by - wap * wu : (real)0;
      r[15] = (real)0;
      if (mean_out) mean_out[p] = wu;
      const int64_t cell = ((int64_t)j0[0] * nc1 + j0[1]) * nc2 + j0[2];
      const unsigned long long tag = ((unsigned long long)epoch << 32) | (unsigned long long)(unsigned)p;
      const unsigned long long old = atomicExch(head + cell, tag);
      next[p] = (unsigned)(old >> 32) == epoch ? (int32_t)(old & 0xffffffffull) : -1;
    }
  }
  scatter_stats_pass<real, 3>(G, x, y, wb, noise, n, stats, s_red);
  if (bad) atomicOr(err, 1);
}

// slot of (group g, innermost digit s) in a row's 172 accumulators: group 0 holds digits 3..6
__device__ __forceinline__ int owner_slot(int g, int s) { return g == 0 ? s - 3 : 4 + 7 * (g - 1) + s; }

// NT threads per block = NT / 64 waves, each owning 1 / (NT / 64) of the line's rows
template <typename real, int NT>
__global__ __launch_bounds__(NT) void k_owner_lines(GridDev<real> G, real* __restrict__ A, real* __restrict__ b, real* __restrict__ cnt,
                                                     real* __restrict__ res, const unsigned long long* __restrict__ head,
                                                     const int32_t* __restrict__ next, const real* __restrict__ rec, unsigned epoch,
                                                     const long long* __restrict__ guard, long long guard_expect, int abl) {
  if (guard && *guard != guard_expect) return;
  extern __shared__ __attribute__((aligned(16))) char smem_owner[];
  constexpr int NS = 172;                        // half-stencil slots per row (d = 3)
  const int g0 = G.g[0], g1 = G.g[1], g2 = G.g[2];
  const int nc0 = g0 - 3, nc1 = g1 - 3, nc2 = g2 - 3;
  const int i0 = blockIdx.x / g1, i1 = blockIdx.x % g1;
  real* acc = reinterpret_cast<real*>(smem_owner);          // [g2][NS]
  real* vb = acc + (size_t)g2 * NS;                         // [g2] b, [g2] cnt, [g2] res
  real* vc = vb + g2;
  real* vr = vc + g2;
  __shared__ int s_any;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  {
    const int nz = g2 * (NS + 3), nz4 = nz & ~3;
    for (int e = 4 * t; e < nz4; e += 4 * NT) {
      acc[e] = (real)0; acc[e + 1] = (real)0; acc[e + 2] = (real)0; acc[e + 3] = (real)0;
    }
    if (t < nz - nz4) acc[nz4 + t] = (real)0;
  }
  if (t == 0) s_any = 0;
  __syncthreads();
  // candidate cells: (d0, d1) = tap digits of this line inside the cell, every cell along the line.  A wave takes 64
  // candidates per round; every lane fetches its cell's head, then the whole record of the head point and its successor, so
  // that one memory latency covers 64 cells -- the points are then processed one by one with the record broadcast from the
  // owning lane (v_readlane: the record becomes wave-uniform).  Longer lists (a second, third ... point in a cell) are walked
  // with direct loads.
  // The four waves split the line's rows into quarters and each adds to ITS rows only (a point whose 4 rows straddle a
  // boundary is visited by both neighbours), so the accumulators need no LDS atomics: 64 lanes of one instruction hit 64
  // different slots, and a wave's LDS operations execute in order.  (With LDS atomics the kernel took 115 us per 4096
  // points: ~7 atomic instructions per point and line, serialised in the CU's LDS unit.)
  const int lo = g2 * wv / (NT / 64), hi = g2 * (wv + 1) / (NT / 64);                        // this wave's rows
  const int clo = lo - 3 < 0 ? 0 : lo - 3, chi = hi < nc2 ? hi : nc2;                        // cells that reach them
  const int ncw = chi - clo, ncand = 16 * ncw;
  const unsigned inv_ncw = (65536u + (unsigned)ncw - 1u) / (unsigned)ncw;   // ci / ncw = (ci * inv) >> 16, exact for ci < 16 * 67
  int found = 0;
  const int a = lane >> 4, bq = (lane >> 2) & 3, c = lane & 3;
  const int dummy = g2 * (NS + 3) + t;                      // this thread's scratch word behind the accumulators
  // one point of cell (.., .., c2) seen from this line, which is tap (da, db, *) of it; r = its record, p = its successor
  // lane constants: one-hot selectors of the lane's tap digits (a select chain on r[] compiles to exec-mask branches: the visit
  // was ~290 instructions, 13 of them branches) and the lane part of the accumulator index
  real m0[4], m1[4], m2[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    m0[k] = a == k ? (real)1 : (real)0;
    m1[k] = bq == k ? (real)1 : (real)0;
    m2[k] = c == k ? (real)1 : (real)0;
  }
  const int lcode = 7 * a + bq;                             // group of (row tap -> tap j) = lcode - (7 da + db)
  const int lidx = 7 * lcode + c;                           // slot of (g, c - cr + 3) in row c2 + cr: 7 g + c - cr for every g >= 0
  auto visit = [&](real (&r)[15], int p, int da, int db, int c2) {
    const int dcode = 7 * da + db;
    const bool upper = lcode > dcode, same = lcode == dcode;
    const int abase = lidx + (c2 * NS - 7 * dcode);         // accumulator of (row c2 + cr, tap j) = abase + cr (NS - 1)
    for (;;) {
      const real w0a = m0[0] * r[0] + m0[1] * r[1] + m0[2] * r[2] + m0[3] * r[3];
      const real w1b = m1[0] * r[4] + m1[1] * r[5] + m1[2] * r[6] + m1[3] * r[7];
      const real w2c = m2[0] * r[8] + m2[1] * r[9] + m2[2] * r[10] + m2[3] * r[11];
      const real w0d = da == 0 ? r[0] : da == 1 ? r[1] : da == 2 ? r[2] : r[3];      // wave-uniform selects
      const real w1d = db == 0 ? r[4] : db == 1 ? r[5] : db == 2 ? r[6] : r[7];
      const real wab = w0d * w1d;                           // w0[da] w1[db]: this line's share of the point's row weights
      const real wj = w0a * w1b * w2c;                      // this lane's tap j = (a, bq, c)
      const real wap = r[12];
      // 4 stencil rows + (lanes 0..3) b / cnt / res of one row each: seven read-modify-writes of LDS, branch-free -- every lane
      // reads, adds and writes seven words, a lane without a contribution its private dummy word -- so that they overlap in
      // ONE LDS round trip (as `if (keep) acc[i] += v` they were seven dependent round trips, ~1 us per visit)
      int ix[7];
      real add[7], cur[7];
      const real wpj = wap * wab * wj;
#pragma unroll
      for (int cr = 0; cr < 4; ++cr) {                      // the 4 rows of the line the point touches: i2 = c2 + cr
        const bool keep = (upper || (same && c >= cr)) && c2 + cr >= lo && c2 + cr < hi;
        ix[cr] = keep ? abase + cr * (NS - 1) : dummy;
        add[cr] = wpj * r[8 + cr];
      }
      {
        const bool mine = lane < 4 && c2 + lane >= lo && c2 + lane < hi;
        const real wi = wab * w2c;                          // lanes 0..3: a = bq = 0, c = lane, so w2c = w2[lane]
        const int row = g2 * NS + c2 + lane;                // vb = acc + g2 NS, vc = vb + g2, vr = vc + g2
        ix[4] = mine ? row : dummy;
        ix[5] = mine ? row + g2 : dummy;
        ix[6] = mine ? row + 2 * g2 : dummy;
        add[4] = wi * r[13];
        add[5] = wi * wap;
        add[6] = wi * r[14];
      }
#pragma unroll
      for (int k = 0; k < 7; ++k) cur[k] = acc[ix[k]];
#pragma unroll
      for (int k = 0; k < 7; ++k) acc[ix[k]] = cur[k] + add[k];
      if (p < 0) break;
      const real* __restrict__ rp = rec + (int64_t)p * OwnerRec<real>::N;   // further points of the same cell (wave-uniform loads)
#pragma unroll
      for (int k = 0; k < 15; ++k) r[k] = rp[k];
      p = next[p];
    }
  };
  // RB rounds of 64 candidates at a time: all their heads are fetched together, then all the records -- two memory latencies
  // per batch (one per ROUND and load made the kernel 83 us per 4096 points, 36 us of it waiting here)
  constexpr int RB = sizeof(real) == 4 ? 5 : 3;
  for (int cb0 = 0; cb0 < ncand; cb0 += 64 * RB) {          // wave-uniform loop
    int p0[RB], nx[RB];
    real rl[RB][15];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const int ci = cb0 + 64 * rb + lane;
      p0[rb] = -1;
      if (ci < ncand) {
        const int dd = (int)(((unsigned)ci * inv_ncw) >> 16), c2 = clo + ci - dd * ncw;
        const int c0 = i0 - (dd >> 2), c1 = i1 - (dd & 3);
        if (c0 >= 0 && c0 < nc0 && c1 >= 0 && c1 < nc1) {
          const unsigned long long h = head[((int64_t)c0 * nc1 + c1) * nc2 + c2];
          if ((unsigned)(h >> 32) == epoch) p0[rb] = (int)(unsigned)(h & 0xffffffffull);
        }
      }
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      nx[rb] = -1;
      if (p0[rb] >= 0) {
        const real* __restrict__ r = rec + (int64_t)p0[rb] * OwnerRec<real>::N;
#pragma unroll
        for (int k = 0; k < 15; ++k) rl[rb][k] = r[k];
        nx[rb] = next[p0[rb]];
      } else {
#pragma unroll
        for (int k = 0; k < 15; ++k) rl[rb][k] = (real)0;
      }
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      unsigned long long live = __ballot(p0[rb] >= 0);
      if (abl & 1) { found |= live != 0; live = 0; }        // timing ablation: scan only
      while (live) {
        const int src = __ffsll((long long)live) - 1;
        live &= live - 1;
        const int ci_s = cb0 + 64 * rb + src;
        // wave-uniform, and the compiler must know it (readfirstlane): the tap digits select record entries with scalar
        // selects instead of exec-mask branches
        const int dd = __builtin_amdgcn_readfirstlane((int)(((unsigned)ci_s * inv_ncw) >> 16));
        const int c2 = __builtin_amdgcn_readfirstlane(clo + ci_s - dd * ncw);
        found = 1;
        real r[15];
#pragma unroll
        for (int k = 0; k < 15; ++k) r[k] = wave_readlane(rl[rb][k], src);
        visit(r, wave_readlane(nx[rb], src), dd >> 2, dd & 3, c2);
      }
    }
  }
  if (found && lane == 0) s_any = 1;
  __syncthreads();
  if (!s_any || (abl & 2)) return;                          // block-uniform: nothing touches this line (abl 2: no write-back)
  const int64_t m = G.m;
  const int64_t row0 = ((int64_t)i0 * g1 + i1) * g2;
  // write-back: the line's 172 g2 accumulators group by group (group 0: 4 g2 reals, then 24 x 7 g2), each group a contiguous
  // span of A_h; only non-zero accumulators are touched, and the loads of several groups are issued before the first store
  // (written as `A[e] += v` the read-modify-writes of a thread serialise, one memory latency each).
  {  // group 0: A[4 i + k]
    const int n0 = 4 * g2;
    for (int e = t; e < n0; e += NT) {
      const real v = acc[(e >> 2) * NS + (e & 3)];
      if (v != (real)0) A[row0 * 4 + e] += v;
    }
  }
  // groups 1..24: a thread owns elements e = t and t + 256 of every group's 7 g2-real span (g2 <= 64: at most two), i.e.
  // fixed (row, digit) pairs -- no index arithmetic in the loop; GB groups = 2 GB loads in flight per thread
  const int n1 = 7 * g2;
  constexpr int NE = (448 + NT - 1) / NT;                   // elements per thread and group (7 g2 <= 448)
  bool ok[NE];
  int le[NE];
#pragma unroll
  for (int j = 0; j < NE; ++j) {
    const int e = t + NT * j;
    ok[j] = e < n1;
    le[j] = (e / 7) * NS + 4 + (e % 7);
  }
  constexpr int GB = NE <= 2 ? 6 : 3;                        // groups per batch: GB NE loads in flight per thread
  for (int gq = 0; gq < 24; gq += GB) {
    real v[GB][NE], o[GB][NE];
#pragma unroll
    for (int k = 0; k < GB; ++k)
#pragma unroll
      for (int j = 0; j < NE; ++j) v[k][j] = ok[j] ? acc[le[j] + 7 * (gq + k)] : (real)0;
#pragma unroll
    for (int k = 0; k < GB; ++k) {
      const real* __restrict__ Ag = A + (int64_t)(7 * (gq + k + 1) - 3) * m + row0 * 7;
      // untouched accumulators read a hot dummy word instead of their A_h entry (a select on the loaded VALUE would make the
      // load unconditional: the whole 86 MB read again per call)
#pragma unroll
      for (int j = 0; j < NE; ++j) o[k][j] = *(v[k][j] != (real)0 ? Ag + t + NT * j : A);
    }
#pragma unroll
    for (int k = 0; k < GB; ++k) {
      real* __restrict__ Ag = A + (int64_t)(7 * (gq + k + 1) - 3) * m + row0 * 7;
#pragma unroll
      for (int j = 0; j < NE; ++j)
        if (v[k][j] != (real)0) {
#ifdef WISKI_OWNER_NT_STORES   // tried in order to keep A_h Infinity-Cache resident for the SpMV that follows: no effect (21.4 us either way)
          __builtin_nontemporal_store(o[k][j] + v[k][j], Ag + t + NT * j);
#else
          Ag[t + NT * j] = o[k][j] + v[k][j];
#endif
        }
    }
  }
  for (int e = t; e < g2; e += NT) {
    if (vb[e] != (real)0) b[row0 + e] += vb[e];
    if (cnt && vc[e] != (real)0) cnt[row0 + e] += vc[e];
    if (res && vr[e] != (real)0) res[row0 + e] += vr[e];
  }
}
