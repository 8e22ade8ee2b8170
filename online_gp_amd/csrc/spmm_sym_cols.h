// Symmetric half-stencil SpMM for many right-hand sides (k >= 8): A_h is read ONCE per 64 columns.
// Included by solve.hip after spmv_sym_dma.h.
//
// The k = 1 kernels give a lane 4 rows and make a pass over A_h per 1 / 4 columns (16 passes for 64 predictive-variance
// right-hand sides: 783 us, 20x above the product's own roofline).  Here a lane IS a column: one 64-lane wave owns RT = 16
// consecutive rows x 64 columns, walks all (7^d + 1) / 2 stored stencil groups for them and writes finished output rows:
//
//   out[j, c] = sum_{o >= centre} a(o, j) v[j + off(o), c]  +  sum_{o > centre} a(o, j - off(o)) v[j - off(o), c]
//
//  * every stencil coefficient is wave-uniform, so A_h is fetched through the scalar unit (s_load, contiguous spans of
//    7 RT resp. 7 (RT + 6) reals per group) and enters the FMAs as an SGPR operand -- no LDS, no lane shuffles;
//  * both terms are written in pull form ("for my rows, who contributes"), so there are no atomics, no partial vectors and
//    no transposed-term window; each A_h entry is read twice (by the tile it belongs to and by the tile it points into),
//    the second time from L2 / Infinity Cache;
//  * the right-hand sides are needed ROW-major, [m][kp] (256 B per row and 64 columns: one fully coalesced load per row);
//    the 7 innermost offsets of RT rows share a window of RT + 6 rows (22 loads for 112 FMAs per direction);
//  * V arrives column-major ([k][m], the layout of every other PCG kernel): k_transpose_cm_rm converts on the way in
//    (64 x 64 tiles through LDS), also forming beta * v . add per column (CG's p . pt); the product is written column-major
//    straight from the accumulators (a lane owns RT consecutive rows of its column).
//
//  * workgroup b runs on XCD b % 8: tiles are dealt so that every XCD sweeps one contiguous eighth of the rows and the v
//    windows of neighbouring tiles meet in ITS L2 (round-robin dealing made each XCD stream all of V: 350 -> 277 us; with the constant row stride below: 240 us).
//
// Measured at 50^3, 64 columns (tools/spmv_probe.py --k 64, both transposes included): 240 us = 29 % of the vector-FMA
// peak (2.74 G lane-FMAs = 70 us).  Insensitive to the tile height (RT = 8 / 12 / 16: 291 / 276 / 277 us at 7 / 6 / 5
// waves per SIMD; 25 / 32: 309 / 413 us, fewer waves than slots), i.e. neither occupancy nor the 67 L1 row loads per output
// row bound it; the suspect is the scalar path (204 MB of A_h, every s_load a scalar-cache miss).  A line-tiled variant
// (a wave = 2..4 grid lines x 17..25 rows, window loads shared between the lines: 35 / 22 loads per row) was built and
// measured at 519 / 721 us -- 2 500 / 1 250 waves cannot hide their own load latency -- and removed.  What is left to do
// is the workgroup-tiled form: v windows shared through LDS by the waves of a block, A_h spans by LDS-DMA.
//
// Works for any d (1..4) and any m; rows past m and columns past k are masked.
#pragma once
#include <type_traits>
#include <utility>

constexpr int SPMMC_RT = 16;

template <typename real, int n, int I0, typename V, typename F, int... T>
__device__ __forceinline__ void spmmc_visit(const V& blk, F&& fn, std::integer_sequence<int, T...>) {
  (fn(std::integral_constant<int, I0 + T>{}, blk[T]), ...);
}

// Visit N contiguous wave-uniform coefficients p[0..N) in 64-byte pieces: fn(integral_constant<idx>, p[idx]).
// The vector type is only dword-aligned: hipcc turns each piece into one s_load_dwordx16 (x8 / x4 / ... for the tail).
// Software-pipelined by hand, one piece ahead: piece i + 1 is requested before the FMAs of piece i and a scheduling
// barrier closes every stage -- left alone, the scheduler hoists ALL loads of a span to its top (a 217-real span wants
// 217 live SGPRs: 500..3000 SGPR spills and one wave per SIMD in k_spmm_sym_lines).
template <typename real, int N, int I0>
struct spmmc_piece {
  static constexpr int CH = 64 / (int)sizeof(real);
  static constexpr int n = (N - I0) >= CH ? CH : ((N - I0) >= CH / 2 ? CH / 2 : ((N - I0) >= CH / 4 ? CH / 4 : ((N - I0) >= 2 ? 2 : 1)));
  typedef real vec_t __attribute__((ext_vector_type(n > 1 ? n : 2), aligned(4)));
  typedef real nvec_t __attribute__((ext_vector_type(n > 1 ? n : 2)));
  static __device__ __forceinline__ nvec_t load(const real* __restrict__ p) {
    if constexpr (n == 1) {
      nvec_t v;
      v[0] = p[I0];
      v[1] = (real)0;
      return v;
    } else {
      return *reinterpret_cast<const vec_t*>(p + I0);   // under-aligned load, naturally aligned value
    }
  }
};
template <typename real, int N, int I0, bool PIPE, typename F, typename V>
__device__ __forceinline__ void spmmc_stage(const real* __restrict__ p, F&& fn, const V cur) {
  using P = spmmc_piece<real, N, I0>;
  constexpr int I1 = I0 + P::n;
  if constexpr (I1 < N) {
    const auto nxt = spmmc_piece<real, N, I1>::load(p);
    spmmc_visit<real, P::n, I0>(cur, fn, std::make_integer_sequence<int, P::n>{});
    if constexpr (PIPE) __builtin_amdgcn_sched_barrier(0);
    spmmc_stage<real, N, I1, PIPE>(p, fn, nxt);
  } else {
    spmmc_visit<real, P::n, I0>(cur, fn, std::make_integer_sequence<int, P::n>{});
    if constexpr (PIPE) __builtin_amdgcn_sched_barrier(0);
  }
}
// PIPE = false leaves the order of loads and FMAs to the scheduler (it hoists the loads of a span as far as SGPRs allow)
template <typename real, int N, bool PIPE = false, typename F>
__device__ __forceinline__ void spmmc_span(const real* __restrict__ p, F&& fn) {
  if constexpr (PIPE) __builtin_amdgcn_sched_barrier(0);
  spmmc_stage<real, N, 0, PIPE>(p, fn, spmmc_piece<real, N, 0>::load(p));
}

// column-major [k][m] -> row-major [m][kp] (kp = k rounded up to 64; padding columns are written as zeros).
// DOT: dots[c] += beta * sum_i V[c][i] * add[c][i]  (slotted, see PcgScal).
// SLICED: one [m][64] image per 64 columns (blockIdx.y), the operand layout of k_spmm_sym_bcast; kp is a multiple of 64 then.
template <typename real, bool DOT, bool SLICED = false>
__global__ __launch_bounds__(256) void k_transpose_cm_rm(int m, int k, int kp, const real* __restrict__ V, real* __restrict__ Vt,
                                                         const real* __restrict__ add, real beta, double* __restrict__ dots) {
  __shared__ real tile[64][65];
  // XCD-contiguous row tiles (grid.x padded to 8 * per; workgroup b sits on XCD b % 8 whatever blockIdx.y): an XCD converts one contiguous eighth
  // of the rows -- the eighth the product kernel's tiles on the same XCD read afterwards -- and the 256-byte column segments of neighbouring tiles,
  // which share cache lines (a column starts at 4 c m bytes: not a multiple of 128), meet in one L2 instead of being fetched by two
  const int nti = (m + 63) >> 6, per = (nti + 7) >> 3;
  const int ti = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  if (ti >= nti) return;
  const int i0 = ti * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  double part[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int c = c0 + ty + 4 * u, i = i0 + tx;
    real v = (real)0;
    part[u] = 0;
    if (c < k && i < m) {
      v = V[(int64_t)c * m + i];
      if (DOT) part[u] = (double)v * (double)add[(int64_t)c * m + i];
    }
    tile[ty + 4 * u][tx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int i = i0 + ty + 4 * u, c = c0 + tx;
    if (SLICED) {
      if (i < m) Vt[((int64_t)blockIdx.y * m + i) * 64 + tx] = tile[tx][ty + 4 * u];
    } else if (i < m && c < kp) Vt[(int64_t)i * kp + c] = tile[tx][ty + 4 * u];
  }
  if (DOT) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const double tot = wave_reduce_sum<double>(part[u]);       // a wave holds one column (64 consecutive rows) per u
      const int c = c0 + ty + 4 * u;
      if (tx == 0 && c < k && tot != 0.0) pcg_dot_add(dots, c, (double)beta * tot);
    }
  }
}

// One wave = RT rows x 64 columns, four waves per block.  Vt row-major [m][kp], Ot column-major [k][m].
// DOT: dots[c] += sum_j Vt[j][c] * Ot[c][j].
// KP: row stride of Vt / Ot known at compile time (64: the common case) or 0 (read kp).  With a constant stride the rows of
// a window that lies inside the grid are base + e * KP: immediate offsets instead of ~9 scalar instructions per row load
// (clamp, 64-bit multiply, add) -- PMC counters showed 2.4 scalar-ALU instructions per vector one and the one scalar ALU
// of a CU as the bound of the launch (104.6 M SALU instructions = 170 us) rather than the loads or the FMAs: 277 -> 240 us.
// The rest of the scalar work is hipcc pairing the FMAs into v_pk_fma_f32 (2 s_mov + 0.5 v_mov per pair to build aligned
// operand pairs); -fno-slp-vectorize gives 212 us but changes nothing end to end (variance of 64 queries 2.53 vs 2.52 ms),
// plain v_fmac through inline asm is slower (256 us: the scheduler no longer sees through it); neither is used.
template <typename real, bool DOT, int KP>
__global__ __launch_bounds__(256) void k_spmm_sym_cols(GridDev<real> G, const real* __restrict__ A_h, const real* __restrict__ Vt, int k, int kp_rt,
                                                      int ng, real* __restrict__ Ot, double* __restrict__ dots) {
  // Ot: the product COLUMN-major [k][m] -- what the solver's vector kernels read.  A lane owns RT consecutive rows of its
  // column, i.e. 64 contiguous bytes: four 16-byte stores per lane and tile (64 lines per wave instruction, 32 MB per
  // launch) instead of a row-major image plus a 13 us transpose launch that reads and rewrites it.
  constexpr int RT = SPMMC_RT, WN = RT + 6;
  static_assert(SPMMC_RT % 4 == 0, "column-major stores are 16-byte groups of rows");
  const int kp = KP ? KP : kp_rt;
  const int m = G.m, d = G.d;
  // a block = 4 independent waves on 4 consecutive tiles; they only meet at the end, to add their p . Ap partials into ONE
  // atomic per column and block instead of one per wave (7 813 x 64 cache-line atomics per launch).  Inside a 64-column
  // solve the kernel takes 304 us against 240 us back to back in the probe either way: the solver's ten 32 MB vectors push
  // A_h out of the 256 MB Infinity Cache between products, the probe's single V does not
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: keeps the tile index, and with it every coefficient load, scalar
  const int c = blockIdx.y * 64 + lane;            // this lane's column
  const bool cok = c < kp;
  const int tile = 4 * ((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) + wv;   // XCD-contiguous row ranges (see the launch)
  const bool active = tile * RT < m;               // padding waves recompute tile 0 and discard it (they must reach the barrier)
  const int j0 = active ? tile * RT : 0;           // first row of the tile (wave-uniform)
  const real* __restrict__ vcol = Vt + (cok ? c : 0);
  real acc[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r) acc[r] = (real)0;
  const int cP = ng - 1;                           // prefix code of the centre: (7^(d-1) - 1) / 2

  // one stencil group: CENTRE = the group of the diagonal (4 reals per row, digits 3..6; digit 3 is the diagonal and
  // belongs to the direct term only), else 7 reals per row.  Compile-time strides let the coefficient loads of a row
  // (and of neighbouring rows) merge into wide scalar loads.
  // one stencil group: CENTRE = the group of the diagonal (4 reals per row, digits 3..6; digit 3 is the diagonal and
  // belongs to the direct term only), else 7 reals per row.  The coefficients of a term form ONE contiguous span of A_h
  // (RT x RS resp. (RT + 6) x RS reals); it is fetched in 64-byte pieces (s_load_dwordx16, dword-aligned) and every
  // piece is consumed straight from its SGPRs.
  auto group = [&](auto centre_tag, const real* __restrict__ Ag, int f) {
    constexpr bool CENTRE = decltype(centre_tag)::value;
    constexpr int RS = CENTRE ? 4 : 7;               // reals per row
    constexpr int S0 = CENTRE ? 3 : 0;               // first stored digit
    {  // ---- direct term: out[j0 + r] += a(s, j0 + r) * v[j0 + r + f + s - 3]
      real win[WN];
      const int wb = j0 + f - 3;
      if (wb >= 0 && wb + WN <= m) {                 // wave-uniform: the whole window exists (always, away from the grid ends)
        const real* __restrict__ wp = vcol + (int64_t)wb * kp;
#pragma unroll
        for (int e = 0; e < WN; ++e) win[e] = wp[(int64_t)e * kp];
      } else {
#pragma unroll
        for (int e = 0; e < WN; ++e) {
          int j = wb + e;
          j = j < 0 ? 0 : (j >= m ? m - 1 : j);      // clamped rows only ever meet coefficients that are exactly zero
          win[e] = vcol[(int64_t)j * kp];
        }
      }
      // ragged last tile: read the span of the last RT rows of the grid and shift the row index (rows >= m are not stored)
      const int jb = j0 + RT <= m ? j0 : (m - RT > 0 ? m - RT : 0);
      const int sh = j0 - jb;
      const real* __restrict__ a = Ag + (int64_t)RS * jb;
      if (sh == 0) {
        spmmc_span<real, RT * RS>(a, [&](auto idx_tag, real coef) {
          constexpr int idx = decltype(idx_tag)::value, r = idx / RS, s = S0 + idx % RS;
          acc[r] += coef * win[r + s];
        });
      } else {
#pragma unroll
        for (int r = 0; r < RT; ++r) {
          const int rr = r + sh < RT ? r + sh : RT - 1;
#pragma unroll
          for (int s = S0; s < 7; ++s) acc[r] += a[RS * rr + (s - S0)] * win[r + s];
        }
      }
    }
    {  // ---- transposed term: out[j] += a(s, i') * v[i'],  i' = j - f - (s - 3) = j0 - f - 3 + e,  r = e + s - 6
      constexpr int T0 = CENTRE ? 4 : 0;
      real src[WN];
      const int ib = j0 - f - 3;
      const bool whole = ib >= 0 && ib + WN <= m;    // wave-uniform: the source rows all exist (always, away from the grid ends)
      if (whole) {
        const real* __restrict__ sp = vcol + (int64_t)ib * kp;
#pragma unroll
        for (int e = 0; e < WN; ++e) src[e] = sp[(int64_t)e * kp];
      } else {
#pragma unroll
        for (int e = 0; e < WN; ++e) {
          int i = ib + e;
          i = i < 0 ? 0 : (i >= m ? m - 1 : i);
          src[e] = vcol[(int64_t)i * kp];
        }
      }
      if (whole) {
        spmmc_span<real, WN * RS>(Ag + (int64_t)RS * ib, [&](auto idx_tag, real coef) {
          constexpr int idx = decltype(idx_tag)::value, e = idx / RS, s = S0 + idx % RS, r = e + s - 6;
          if constexpr (s >= T0 && r >= 0 && r < RT) acc[r] += coef * src[e];
        });
      } else {
#pragma unroll
        for (int e = 0; e < WN; ++e) {
          const int i = ib + e;
          const bool inside = i >= 0 && i < m;
          const real* __restrict__ a = Ag + (int64_t)RS * (inside ? i : 0);
#pragma unroll
          for (int s = T0; s < 7; ++s) {
            const int r = e + s - 6;
            if (r >= 0 && r < RT) acc[r] += (inside ? a[s - S0] : (real)0) * src[e];
          }
        }
      }
    }
  };
  group(std::true_type{}, A_h, 0);
  for (int g = 1; g < ng; ++g) {
    int f = 0, rem = cP + g;                         // flat offset of the group's centre digit (leading d-1 stencil digits)
    for (int q = d - 2; q >= 0; --q) {
      f += (rem % 7 - 3) * G.stride[q];
      rem /= 7;
    }
    group(std::false_type{}, A_h + (int64_t)(7 * g - 3) * m, f);
  }
  double dot = 0;
  const bool cw = c < k && active;                  // this lane writes (padding columns and padding waves do not)
  if (j0 + RT <= m) {
    real* __restrict__ op = Ot + (int64_t)(c < k ? c : 0) * m + j0;      // j0 and m are multiples of 4: 16-byte aligned
    const real* __restrict__ vp = vcol + (int64_t)j0 * kp;
#pragma unroll
    for (int r = 0; r < RT; r += 4)
      if (cw) store4<real>(op + r, acc[r], acc[r + 1], acc[r + 2], acc[r + 3]);
#pragma unroll
    for (int r = 0; r < RT; ++r)
      if (DOT) dot += (double)vp[(int64_t)r * kp] * (double)acc[r];
  } else {
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      const int j = j0 + r;
      if (j < m && cw) Ot[(int64_t)c * m + j] = acc[r];
      if (DOT && j < m && cok && active) dot += (double)vcol[(int64_t)j * kp] * (double)acc[r];
    }
  }
  if constexpr (DOT) {
    __shared__ double s_dot[4][64];
    s_dot[wv][lane] = active ? dot : 0.0;
    __syncthreads();
    if (wv == 0 && c < k) pcg_dot_add(dots, c, s_dot[0][lane] + s_dot[1][lane] + s_dot[2][lane] + s_dot[3][lane]);
  }
}
