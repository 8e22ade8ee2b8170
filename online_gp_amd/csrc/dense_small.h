// Cholesky factorisation (n <= 512) and triangular inverse (n <= 480) of SMALL matrices in one launch each.
// Included by dense.hip (uses its MFMA helpers mfma16 / Acc4 / frag_row).
//
// The spectral Woodbury factor (lazy/spectral_woodbury.py) refactorises an r x r fp64 matrix, r ~ 300-500, after every
// hyper-parameter step.  At that size the blocked multi-launch wiski_potrf is pure latency: 3 launches per 64-wide panel, the
// one-wave diagonal kernel alone 56 us, 0.57 ms for n = 327 although the arithmetic is 12 MFLOP.  Here ONE workgroup
// (8 waves, one CU, the matrix resident in L2, the current panel in LDS) runs the whole right-looking factorisation with
// 32-wide panels:
//   C  panel  L21 = A21 L11^-T      32 x 32 MFMA tile products, one row tile per wave
//   D1 first block column of the trailing update A22 -= L21 L21^T: the next diagonal block (kept in LDS) and the next panel
//   B  wave 0 factorises the next 32 x 32 block and inverts the factor (one instruction stream for both, see wave_potrf32) ...
//   D2 ... while the other waves finish the trailing update (lower tiles, operands from the LDS panel)
//   A  the next panel into LDS
// and stores the inverses of the diagonal blocks.  k_tri_inv_small then builds the explicit inverse of the factor, one
// workgroup per 32-column block (the block columns of a triangular inverse are independent):
//   X_JJ = L_JJ^-1,   X_IJ = -L_II^-1 sum_{K=J}^{I-1} L_IK X_KJ.
#pragma once
#include <mutex>
#include <tuple>
#include <vector>

constexpr int SNB = 32;        // panel width
constexpr int SLD = 34;        // LDS row stride (reals)
constexpr int SLDX = 33;       // row stride of the inverse's block column in k_tri_inv_small4
constexpr int SMALL_N_MAX = 480;        // factor + explicit inverse in two launches
constexpr int SMALL_N_MAX_POTRF = 512;  // the factorisation alone still fits its panel in LDS (157.7 KB in fp64)
constexpr size_t SMALL_LDS_MAX = 160 * 1024;   // LDS of a CU
constexpr int SWG = 512;       // threads of the factorisation workgroup (8 waves: 256 registers each -- the one-wave diagonal step wants them)
constexpr int SNW = SWG / 64;

// 32 x 32 output tile of one wave as 2 x 2 MFMA tiles: acc[a][b] += sum_k A(i, k) B(k, j), i = a*16 + (lane & 15) etc.
// opA(i, k), opB(k, j) are accessors into LDS.
template <typename real, typename FA, typename FB>
__device__ __forceinline__ void wave_tile32(int lane, int K, FA opA, FB opB, typename Acc4<real>::type (&acc)[2][2]) {
  for (int ks = 0; ks < K; ks += 4) {
    const int kk = ks + (lane >> 4);
    real af[2], bf[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) af[a] = opA(a * 16 + (lane & 15), kk);
#pragma unroll
    for (int b = 0; b < 2; ++b) bf[b] = opB(kk, b * 16 + (lane & 15));
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = mfma16(af[a], bf[b], acc[a][b]);
  }
}

template <typename real>
__device__ __forceinline__ void zero_acc(typename Acc4<real>::type (&acc)[2][2]) {
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[a][b][r] = (real)0;
}

// Lane `src` of a wave-uniform read (compile-time lane: v_readlane, no LDS permute round trip).
__device__ __forceinline__ float read_lane(float v, int src) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src)); }
__device__ __forceinline__ double read_lane(double v, int src) {
  const long long b = __builtin_bit_cast(long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, src), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), src);
  return __builtin_bit_cast(double, (long long)(((unsigned long long)hi << 32) | lo));
}

// One wave: sD (32 x 32, LDS, identity-padded beyond nb) -> its Cholesky factor in place (strict upper zeroed) and the
// inverse of the factor in sI.  The two halves of the wave run the SAME instruction stream on different data: lane t < 32
// holds row t of the matrix, lane 32 + t holds e_t, the right-hand side whose forward substitution gives column t of the
// inverse.  Step k: a_k = v[k] / sqrt(pivot) in every lane (the factor's column entry l_tk in the lower half, the inverse's
// entry x_kt in the upper half) and v[j] -= a_k l_jk for j > k -- the rank-1 update of the factorisation in the lower half and
// the substitution step in the upper half.  The inverse costs no extra instructions.
//
// The serial chain of a step is pivot -> rsqrt (+ 2 Newton steps) -> a_k -> the next pivot, and nothing else: the update of
// the next TWO columns takes l_{k+1,k}, l_{k+2,k} by lane reads (v_readlane: no LDS round trip), the rest of the rank-1 update
// (columns >= k + 3) goes through LDS one iteration LATER -- its broadcast reads are issued at the top of the next iteration
// and land while that iteration's rsqrt chain runs (updates of a column commute; column j only has to be complete when step j
// starts, and steps j - 1 / j - 2 reach it by the lane-read path).  One wave's LDS operations execute in order, so the reads
// need no wait on the write before them.  ~9 -> ~3 us per 32 x 32 block in fp64.  Returns true if a pivot was not positive.
template <typename real>
__device__ __forceinline__ bool wave_potrf32(real (*sD)[SLD], real (*sI)[SLD], real (*sCol)[2 * SNB], int lane) {
  // (an opaque zero added to the LDS bases: the ~600 constant addresses of the unrolled loop below then stay immediate offsets of
  // one base register instead of being materialised and hoisted out of the caller's loop, which spilled ~500 registers)
  int opaque = 0;
  asm volatile("" : "+v"(opaque));
  sCol += opaque;
  sD += opaque;
  sI += opaque;
  const int t = lane & 31;
  const bool fac = lane < SNB;
  real v[SNB];
#pragma unroll
  for (int j = 0; j < SNB; ++j) {
    const real ld = sD[t][j];                    // (every lane reads: a select, not a branch -- with branches in this function the
    v[j] = fac ? ld : (j == t ? (real)1 : (real)0);   //  compiler sinks the deferred updates to their uses and spills)
  }
  bool bad = false;
  real piv0 = read_lane(v[0], 0);
  real aprev = (real)0;
#pragma unroll
  for (int k = 0; k < SNB; ++k) {
    // broadcast reads for the deferred part of step k - 1 (columns >= k + 2)
    real lc[SNB];
    if (k >= 1) {
#pragma unroll
      for (int j = k + 2; j < SNB; ++j) lc[j] = sCol[(k - 1) & 1][j];
    }
    __builtin_amdgcn_sched_barrier(0);
    // the chain
    if (!(piv0 > (real)0)) bad = true;
    const real piv = piv0 > (real)0 ? piv0 : (real)1;
    real ri;
    if constexpr (sizeof(real) == 4) {
      ri = __builtin_amdgcn_rsqf(piv);
      ri = ri * (1.5f - 0.5f * piv * ri * ri);
    } else {
      ri = __builtin_amdgcn_rsq(piv);
      ri = ri * (1.5 - 0.5 * piv * ri * ri);
      ri = ri * (1.5 - 0.5 * piv * ri * ri);
    }
    // (rows t < k of the factor half carry upper-triangle leftovers in v[k]: they scale and update them like everybody else --
    //  nobody reads those slots, the store below writes zeros there -- which keeps the loop free of branches)
    const real ak = v[k] * ri;
    v[k] = ak;
    if (k + 1 < SNB) {
      v[k + 1] -= ak * read_lane(ak, k + 1);
      piv0 = read_lane(v[k + 1], k + 1);
    }
    if (k + 2 < SNB) v[k + 2] -= ak * read_lane(ak, k + 2);
    sCol[k & 1][lane] = ak;                      // (unconditional: slots 32..63 take the upper half's values and are never read)
    __builtin_amdgcn_sched_barrier(0);
    // deferred part of step k - 1
    if (k >= 1) {
#pragma unroll
      for (int j = k + 2; j < SNB; ++j) v[j] -= aprev * lc[j];
    }
    aprev = ak;
    __builtin_amdgcn_sched_barrier(0);           // finish this step's updates here: deferring them keeps the loaded columns live (spills)
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int j = 0; j < SNB; ++j) {
    if (fac) sD[t][j] = j <= t ? v[j] : (real)0;
    else sI[j][t] = v[j];
  }
  return bad;
}

#ifdef WISKI_POTRF_TIMING                             // phase stamps of the factorisation workgroup (tools/potrf_phases.py: a -DWISKI_POTRF_TIMING build)
__device__ long long g_potrf_stamp[17 * 8];
#define POTRF_STAMP(round, k) do { if (threadIdx.x == 0 && blockIdx.x == 0 && (round) < 17) g_potrf_stamp[(round) * 8 + (k)] = wall_clock64(); } while (0)
#else
#define POTRF_STAMP(round, k) do { } while (0)
#endif

// A (n x n, lda) -> lower Cholesky factor in place (strict upper zeroed), inverses of the 32 x 32 diagonal blocks of the
// factor in dinv [nblk][32][32] (identity-padded).  One workgroup of SWG threads.  Dynamic LDS: potrf_small_lds(n).
//
// The loop is written so that every piece of code exists ONCE (the unrolled diagonal step alone is 13 KB in fp64; the kernel
// has to stay well inside the 64 KB instruction cache it shares with the neighbouring CU -- a round executes each phase once,
// so a kernel that does not fit re-fetches every instruction of every round from L2: measured 8.7 vs 5.3 us for the diagonal
// step, and the 60 KB version of this kernel lost 3 us per round to it).  A round of block k:
//   X   wave 0 factorises + inverts the diagonal block (LDS -> LDS) WHILE the other 7 waves finish the trailing update of the
//       previous round (D2: tiles with tj >= 1, operands = the previous panel in LDS) and then -- once all 7 are through,
//       counted in LDS, no workgroup barrier -- bring this round's panel (written by the previous D1) into LDS
//   --  factor block and its inverse to global
//   C   panel L21 = A21 L11^-T, one 32-row MFMA tile per wave (in place in LDS, and to global)
//   D1  first block column of A22 -= L21 L21^T: the NEXT diagonal block (stays in LDS) and the next panel (to global)
template <typename real>
__global__ __launch_bounds__(SWG) void k_potrf_small(int n, real* __restrict__ A, int lda, real* __restrict__ dinv, int32_t* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  real(*sD)[SLD] = reinterpret_cast<real(*)[SLD]>(smem_raw);
  real(*sI)[SLD] = sD + SNB;
  real(*sCol)[2 * SNB] = reinterpret_cast<real(*)[2 * SNB]>(sI + SNB);
  int* sCnt = reinterpret_cast<int*>(reinterpret_cast<real*>(sCol) + 4 * SNB);      // arrivals of the 7 update waves (monotone)
  real(*sP)[SLD] = reinterpret_cast<real(*)[SLD]>(reinterpret_cast<real*>(sCol) + 4 * SNB + 2);
  using acc_t = typename Acc4<real>::type;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int lj = tid & 31;
  // panel below the diagonal block at k0 (rows k0 + nb .., columns k0 .. k0 + nb) -> sP, zero-padded to whole tiles, by the
  // threads t0 .. t0 + 32 * rp.  Loads in batches of 8 per thread: issued back to back, one wait.
  auto load_panel = [&](int k0, int nb, int mt, int mtp, int t0, int rp) {
    const int li0 = (tid - t0) >> 5;
    real v[8];
    for (int rb = 0; rb < mtp; rb += 8 * rp) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = rb + u * rp + li0;
        v[u] = (r < mt && lj < nb) ? A[(int64_t)(k0 + nb + r) * lda + k0 + lj] : (real)0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = rb + u * rp + li0;
        if (r < mtp) sP[r][lj] = v[u];
      }
    }
  };
  // one lower tile (ti >= tj) of the trailing matrix at `base`:  A22 -= P_ti P_tj^T with the panel rows in sP; lim = its rows.
  // load_tile requests the tile of A22 (16 independent loads), finish_tile does the products and writes it back.  Tile (0, 0)
  // of D1 is the next diagonal block: it stays in LDS (identity-padded to 32 x 32).
  auto load_tile = [&](int ti, int tj, int base, int lim, real (&cv)[2][2][4]) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = ti * SNB + a * 16 + frag_row<real>(lane, r), j = tj * SNB + b * 16 + (lane & 15);
          cv[a][b][r] = (i < lim && j <= i) ? A[(int64_t)(base + i) * lda + base + j] : (real)0;
        }
  };
  auto finish_tile = [&](int ti, int tj, int base, int lim, bool diag_to_lds, int nbn, const real (&cv)[2][2][4]) {
    acc_t acc[2][2];
    zero_acc<real>(acc);
    wave_tile32<real>(lane, SNB, [&](int i, int k) { return sP[ti * SNB + i][k]; }, [&](int k, int j) { return sP[tj * SNB + j][k]; }, acc);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int il = a * 16 + frag_row<real>(lane, r), jl = b * 16 + (lane & 15);
          const int i = ti * SNB + il, j = tj * SNB + jl;
          const real v = cv[a][b][r] - acc[a][b][r];
          if (diag_to_lds && ti == 0) sD[il][jl] = (il < nbn && jl < nbn) ? v : (il == jl ? (real)1 : (real)0);
          else if (i < lim && j <= i) A[(int64_t)(base + i) * lda + base + j] = v;
        }
  };
  // ---- prologue: first diagonal block and first panel into LDS
  {
    const int nb = n < SNB ? n : SNB;
    for (int i = tid >> 5; i < SNB; i += SWG / 32) sD[i][lj] = (i < nb && lj < nb) ? A[(int64_t)i * lda + lj] : (i == lj ? (real)1 : (real)0);
    const int mt = n - nb;
    load_panel(0, nb, mt, (mt + SNB - 1) / SNB * SNB, 0, SWG / 32);
    if (tid == 0) *sCnt = 0;
  }
  __syncthreads();
  int pmt = 0;                                               // rows of the previous round's trailing matrix (it starts at k0)
  int narr = 0;                                              // arrivals the update waves have waited for so far
  for (int k0 = 0, blk = 0;; ++blk) {
    const int nb = n - k0 < SNB ? n - k0 : SNB;
    const int mt = n - k0 - nb;                              // rows below the diagonal block
    const int mtp = (mt + SNB - 1) / SNB * SNB;
    POTRF_STAMP(blk, 0);
    // ---- X: wave 0 on the diagonal block; the others on the previous round's trailing tiles, then this round's panel
    if (w == 0) {
      __builtin_amdgcn_s_setprio(3);                          // the serial chain of the round: ahead of the tile waves on this SIMD
      const bool bad = wave_potrf32<real>(sD, sI, sCol, lane);
      __builtin_amdgcn_s_setprio(0);
      if (bad && lane == 0) atomicOr(info, 1);
      POTRF_STAMP(blk, 1);
    } else {
      const int pnt = (pmt + SNB - 1) / SNB;
      const int ntile2 = (pnt - 1) * pnt / 2;                // lower triangle of the (pnt - 1) x (pnt - 1) tiles with ti >= tj >= 1
      // wave w takes the tiles tl = w - 1, w - 1 + 7, ... of the triangle (row by row).  (Requesting the next tile before this
      // tile's products was measured: no gain -- at 64 cycles per fp64 MFMA the CU's matrix pipes, not the L2 latency, bound
      // this phase -- and its 6 KB of code pushed the kernel out of the instruction cache)
      int ti = 1, tj = 1;
      for (int tl = 0; tl < ntile2; ++tl) {
        if (tl % (SNW - 1) == w - 1) {
          real cv[2][2][4];
          load_tile(ti, tj, k0, pmt, cv);
          finish_tile(ti, tj, k0, pmt, false, 0, cv);
        }
        if (++tj > ti) { ++ti; tj = 1; }
      }
      if (blk > 0 && mt > 0) {
        // all 7 waves have read their last operands from the previous panel -> it may be overwritten
        __threadfence_block();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        narr += SNW - 1;
        if (lane == 0) {
          __hip_atomic_fetch_add(sCnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          while (__hip_atomic_load(sCnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < narr) __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_wave_barrier();
        load_panel(k0, nb, mt, mtp, 64, (SWG - 64) / 32);
      }
    }
    __threadfence_block();
    __syncthreads();
    POTRF_STAMP(blk, 2);
    // ---- factor block and its inverse (LDS) -> A, dinv
    for (int i = tid >> 5; i < SNB; i += SWG / 32) {
      if (i < nb && lj < nb) A[(int64_t)(k0 + i) * lda + k0 + lj] = sD[i][lj];
      dinv[(int64_t)blk * SNB * SNB + i * SNB + lj] = sI[i][lj];
    }
    if (mt == 0) break;
    const int nt = mtp / SNB;
    const int nbn = mt < SNB ? mt : SNB;                     // size of the next diagonal block
    // ---- C: panel L21 = A21 L11^-T, one 32-row tile per wave (in place in LDS, and to global)
    for (int rt = w; rt * SNB < mt; rt += SNW) {
      const int r0 = rt * SNB;
      acc_t acc[2][2];
      zero_acc<real>(acc);
      wave_tile32<real>(lane, SNB, [&](int i, int k) { return sP[r0 + i][k]; }, [&](int k, int j) { return sI[j][k]; }, acc);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();                      // every lane has read its operands of this tile
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = r0 + a * 16 + frag_row<real>(lane, r), j = b * 16 + (lane & 15);
            sP[i][j] = acc[a][b][r];
            if (i < mt && j < nb) A[(int64_t)(k0 + nb + i) * lda + k0 + j] = acc[a][b][r];
          }
    }
    __syncthreads();
    POTRF_STAMP(blk, 3);
    // ---- D1: first block column of the update (the next diagonal block -> sD, the next panel -> global)
    for (int ti = w; ti < nt; ti += SNW) {
      real cv[2][2][4];
      load_tile(ti, 0, k0 + nb, mt, cv);
      finish_tile(ti, 0, k0 + nb, mt, true, nbn, cv);
    }
    __threadfence_block();
    __syncthreads();
    POTRF_STAMP(blk, 4);
    k0 += nb;
    pmt = mt;
  }
  // strict upper triangle: zero
  for (int i = tid >> 6; i < n; i += SNW)
    for (int j = i + 1 + lane; j < n; j += 64) A[(int64_t)i * lda + j] = (real)0;
}

// Explicit inverse X = L^-1 (n x n lower, ldx), block column J by workgroup J (256 threads); dinv from k_potrf_small.
// Dynamic LDS: (nrows_pad + 3 * 32) * SLD reals.
template <typename real>
__global__ __launch_bounds__(256) void k_tri_inv_small(int n, const real* __restrict__ L, int ldl, const real* __restrict__ dinv, real* __restrict__ X, int ldx) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  real(*sL)[SLD] = reinterpret_cast<real(*)[SLD]>(smem_raw);          // staged tile of L (or of dinv)
  real(*sS)[SLD] = sL + SNB;                                          // the accumulated sum, as an operand
  real(*sX)[SLD] = sS + SNB;                                          // block column J of X, rows from J * 32
  using acc_t = typename Acc4<real>::type;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int J = blockIdx.x, j0 = J * SNB;
  const int nblk = (n + SNB - 1) / SNB;
  const int nbj = n - j0 < SNB ? n - j0 : SNB;
  const int qa = w >> 1, qb = w & 1;                                  // this wave's 16 x 16 quadrant of a 32 x 32 tile
  // zero the part of the block column above the diagonal block, X_JJ = dinv[J]
  for (int e = tid; e < j0 * SNB; e += 256) {
    const int i = e / SNB, j = e % SNB;
    if (j < nbj) X[(int64_t)i * ldx + j0 + j] = (real)0;
  }
  for (int e = tid; e < SNB * SNB; e += 256) {
    const int i = e / SNB, j = e % SNB;
    const real v = dinv[(int64_t)J * SNB * SNB + e];
    sX[i][j] = v;
    if (i < nbj && j < nbj) X[(int64_t)(j0 + i) * ldx + j0 + j] = v;
  }
  __syncthreads();
  for (int I = J + 1; I < nblk; ++I) {
    const int i0 = I * SNB;
    const int nbi = n - i0 < SNB ? n - i0 : SNB;
    acc_t acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = (real)0;
    for (int K = J; K < I; ++K) {
      for (int e = tid; e < SNB * SNB; e += 256) {
        const int i = e / SNB, k = e % SNB;
        sL[i][k] = i < nbi ? L[(int64_t)(i0 + i) * ldl + K * SNB + k] : (real)0;      // (K < I: columns always inside the matrix)
      }
      __syncthreads();
      const int xr = (K - J) * SNB;
#pragma unroll
      for (int ks = 0; ks < SNB; ks += 4) {
        const int kk = ks + (lane >> 4);
        acc = mfma16(sL[qa * 16 + (lane & 15)][kk], sX[xr + kk][qb * 16 + (lane & 15)], acc);
      }
      __syncthreads();
    }
    // S -> LDS, dinv[I] -> LDS, X_IJ = -dinv[I] S
#pragma unroll
    for (int r = 0; r < 4; ++r) sS[qa * 16 + frag_row<real>(lane, r)][qb * 16 + (lane & 15)] = acc[r];
    for (int e = tid; e < SNB * SNB; e += 256) sL[e / SNB][e % SNB] = dinv[(int64_t)I * SNB * SNB + e];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = (real)0;
#pragma unroll
    for (int ks = 0; ks < SNB; ks += 4) {
      const int kk = ks + (lane >> 4);
      acc = mfma16(sL[qa * 16 + (lane & 15)][kk], sS[kk][qb * 16 + (lane & 15)], acc);
    }
    const int xr = (I - J) * SNB;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = qa * 16 + frag_row<real>(lane, r), j = qb * 16 + (lane & 15);
      const real v = -acc[r];
      sX[xr + i][j] = v;
      if (i < nbi && j < nbj) X[(int64_t)(i0 + i) * ldx + j0 + j] = v;
    }
    __syncthreads();
  }
}

// The same inverse with the sum over K dealt to the 4 waves: wave w stages the tiles L_IK, K = J + w, J + w + 4, ... in its OWN LDS
// region and accumulates a full 32 x 32 partial product (no workgroup barrier inside the K loop: X_KJ is read-only by then), the
// partials are summed through LDS, and X_IJ = -L_II^-1 S is one more tile product.  3 barriers per block row instead of 2 per
// (row, K) pair, and a quarter of the tile products on the critical path: 88 -> ~40 us at n = 327.  Needs 4 tile regions of LDS
// besides the block column (161.5 KB at n = 480 in fp64).
// Cross-workgroup flags of the cooperative factorisation (dense_coop.h): lane 0 polls until *p >= need.  What the flag guards is
// read with agent-scope atomic loads afterwards (coop_load), so no cache invalidate is needed -- only that those loads are
// issued after the poll has returned, which in-order issue gives once the compiler keeps them below the loop.
__device__ __forceinline__ void coop_wait_ge(const int* p, int need) {
  if ((threadIdx.x & 63) == 0)
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) __builtin_amdgcn_s_sleep(1);
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ bool coop_poll_ge(const int* p, int need) {        // one look, wave-uniform answer
  int v = 0;
  if ((threadIdx.x & 63) == 0) v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  v = __builtin_amdgcn_readfirstlane(v);
  asm volatile("" ::: "memory");
  return v >= need;
}
template <typename real>
__device__ __forceinline__ real coop_load(const real* p, bool coherent) {
  return coherent ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
}

// Block column J of X = L^-1 by the first 4 waves of the calling workgroup.  ready != nullptr: block row I of L (and dinv[I])
// may only be read once ready[I] >= 1 -- the column then TRAILS a factorisation that is still running in another workgroup.
template <typename real>
__device__ __forceinline__ void tri_inv_column4(int n, const real* L, int ldl, const real* dinv, real* __restrict__ X, int ldx, int J,
                                                const int* ready) {      // (L, dinv not __restrict__: another workgroup may still be writing them)
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  real(*sW)[SNB][SLD] = reinterpret_cast<real(*)[SNB][SLD]>(smem_raw);      // [4] wave-private: staged L tile, then the wave's partial
  real(*sS)[SLD] = sW[0];                                                    // summed partials  (take over regions 0 and 1 once the partials
  real(*sDI)[SLD] = sW[1];                                                   // dinv[I]           have been read: same element, same thread)
  real(*sX)[SLDX] = reinterpret_cast<real(*)[SLDX]>(sW + 4);                // block column J of X, rows from J * 32 (stride 33: n = 480 fits in fp64)
  using acc_t = typename Acc4<real>::type;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int j0 = J * SNB;
  const int nblk = (n + SNB - 1) / SNB;
  const int nbj = n - j0 < SNB ? n - j0 : SNB;
  const int qa = w >> 1, qb = w & 1;
  for (int e = tid; e < j0 * SNB; e += 256) {
    const int i = e / SNB, j = e % SNB;
    if (j < nbj) X[(int64_t)i * ldx + j0 + j] = (real)0;
  }
  if (ready) coop_wait_ge(ready + J, 1);
  for (int e = tid; e < SNB * SNB; e += 256) {
    const int i = e / SNB, j = e % SNB;
    const real v = coop_load(&dinv[(int64_t)J * SNB * SNB + e], ready != nullptr);
    sX[i][j] = v;
    if (i < nbj && j < nbj) X[(int64_t)(j0 + i) * ldx + j0 + j] = v;
  }
  __syncthreads();
  for (int I = J + 1; I < nblk; ++I) {
    const int i0 = I * SNB;
    const int nbi = n - i0 < SNB ? n - i0 : SNB;
    acc_t acc[2][2];
    zero_acc<real>(acc);
    if (ready) coop_wait_ge(ready + I, 1);
    for (int K = J + w; K < I; K += 4) {
      // stage L[I][K] (32 x 32) into this wave's region: 16 elements per lane, rows of 256 B
      real v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int e = u * 64 + lane, i = e >> 5, k = e & 31;
        v[u] = i < nbi ? coop_load(&L[(int64_t)(i0 + i) * ldl + K * SNB + k], ready != nullptr) : (real)0;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();                       // the previous tile's operand reads are done
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int e = u * 64 + lane;
        sW[w][e >> 5][e & 31] = v[u];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      const int xr = (K - J) * SNB;
      wave_tile32<real>(lane, SNB, [&](int i, int k) { return sW[w][i][k]; }, [&](int k, int j) { return sX[xr + k][j]; }, acc);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) sW[w][a * 16 + frag_row<real>(lane, r)][b * 16 + (lane & 15)] = acc[a][b][r];
    __syncthreads();
    for (int e = tid; e < SNB * SNB; e += 256) {
      const int i = e / SNB, j = e % SNB;
      const real sum = sW[0][i][j] + sW[1][i][j] + sW[2][i][j] + sW[3][i][j];
      const real di = coop_load(&dinv[(int64_t)I * SNB * SNB + e], ready != nullptr);
      sS[i][j] = sum;                                        // (= sW[0][i][j], sW[1][i][j]: this thread's own elements)
      sDI[i][j] = di;
    }
    __syncthreads();
    acc_t q;
#pragma unroll
    for (int r = 0; r < 4; ++r) q[r] = (real)0;
#pragma unroll
    for (int ks = 0; ks < SNB; ks += 4) {
      const int kk = ks + (lane >> 4);
      q = mfma16(sDI[qa * 16 + (lane & 15)][kk], sS[kk][qb * 16 + (lane & 15)], q);
    }
    const int xr = (I - J) * SNB;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = qa * 16 + frag_row<real>(lane, r), j = qb * 16 + (lane & 15);
      const real v = -q[r];
      sX[xr + i][j] = v;
      if (i < nbi && j < nbj) X[(int64_t)(i0 + i) * ldx + j0 + j] = v;
    }
    __syncthreads();
  }
}

template <typename real>
__global__ __launch_bounds__(256) void k_tri_inv_small4(int n, const real* __restrict__ L, int ldl, const real* __restrict__ dinv, real* __restrict__ X, int ldx) {
  tri_inv_column4<real>(n, L, ldl, dinv, X, ldx, (int)blockIdx.x, nullptr);
}

template <typename real>
static inline size_t tri_inv_small4_lds(int n) {
  const int mtp = (n + SNB - 1) / SNB * SNB;
  return (size_t)(mtp * SLDX + 4 * SNB * SLD) * sizeof(real);
}

template <typename real>
static inline size_t potrf_small_lds(int n) {
  const int mtp = (n + SNB - 1) / SNB * SNB;
  return (size_t)((mtp + 2 * SNB) * SLD + 4 * SNB + 2) * sizeof(real);
}
template <typename real>
static inline size_t tri_inv_small_lds(int n) {
  const int mtp = (n + SNB - 1) / SNB * SNB;
  return (size_t)((mtp + 3 * SNB) * SLD) * sizeof(real);
}

constexpr int WISKI_SMALL_UNAVAILABLE = 1;   // (internal) the small-matrix path cannot run on this device: use the blocked one

#include "dense_coop.h"

// Persistent flag block of the cooperative kernel, one per (device, stream): calls on one stream are ordered, calls on different
// streams must not share flags.  Zeroed once at allocation; every launch leaves it zeroed (dense_coop.h).
static CoopSync* coop_sync_for(int dev, hipStream_t s) {
  static std::mutex mu;
  static std::vector<std::tuple<int, hipStream_t, CoopSync*>> pool;
  std::lock_guard<std::mutex> lk(mu);
  for (auto& e : pool)
    if (std::get<0>(e) == dev && std::get<1>(e) == s) return std::get<2>(e);
  CoopSync* p = nullptr;
  if (hipMalloc((void**)&p, sizeof(CoopSync)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  if (hipMemset(p, 0, sizeof(CoopSync)) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(p); return nullptr; }
  pool.emplace_back(dev, s, p);
  return p;
}

// A caller off the critical path (the two-level block's refresh on its side stream) asks for the ONE-workgroup factorisation: the cooperating
// form spreads ~18 workgroups with the panel's LDS over as many CUs for ~100 us, and a kernel of the main stream that needs every wave slot
// at once (the LDS-DMA SpMV: 248 of an XCD's 256) then pays a second dispatch round.  Set / cleared around the call by wiski_potrf_quiet().
static int g_potrf_quiet = 0;
extern "C" void wiski_potrf_quiet(int on) { g_potrf_quiet = on; }
static bool coop_path_enabled() {
  static const bool on = [] {
    const char* e = getenv("WISKI_POTRF_COOP");
    return !(e && e[0] == '0');
  }();
  return on && !g_potrf_quiet;
}

// Factor (and optionally invert: d_X != nullptr) a small matrix.  d_dinv: scratch [nblk][32][32].
template <typename real>
static int potrf_small(int n, real* d_A, int lda, real* d_dinv, real* d_X, int ldx, int32_t* d_info, hipStream_t s) {
  // the ~160 KB dynamic-LDS opt-in is a per-device function attribute: tracked per device, set once under a lock; a device that
  // refuses it (less LDS) answers WISKI_SMALL_UNAVAILABLE and the callers take the blocked path
  static std::mutex mu;
  static int state[64] = {0};                     // 0 unknown, 1 granted, -1 refused
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return WISKI_SMALL_UNAVAILABLE;
  {
    std::lock_guard<std::mutex> lk(mu);
    if (state[dev] == 0) {
      const bool ok = hipFuncSetAttribute((const void*)k_potrf_small<real>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)potrf_small_lds<real>(SMALL_N_MAX_POTRF)) == hipSuccess &&
                      hipFuncSetAttribute((const void*)k_potrf_coop<real>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMALL_LDS_MAX) == hipSuccess &&
                      hipFuncSetAttribute((const void*)k_tri_inv_small<real>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tri_inv_small_lds<real>(SMALL_N_MAX)) == hipSuccess &&
                      hipFuncSetAttribute((const void*)k_tri_inv_small4<real>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMALL_LDS_MAX) == hipSuccess;
      if (!ok) (void)hipGetLastError();
      state[dev] = ok ? 1 : -1;
    }
    if (state[dev] < 0) return WISKI_SMALL_UNAVAILABLE;
  }
  const int nblk = (n + SNB - 1) / SNB;
  // The cooperating-workgroups kernel (one launch for factor + inverse) unless the stream is being captured into a graph (its
  // flag block is allocated on first use) or its LDS does not fit; then the one-workgroup factorisation and the separate inverse.
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusActive; }
  const bool inv4 = d_X && tri_inv_small4_lds<real>(n) <= SMALL_LDS_MAX;
  if (coop_path_enabled() && nblk <= 16 && (!d_X || inv4) && potrf_coop_lds<real>(n, d_X != nullptr) <= SMALL_LDS_MAX) {
    CoopSync* sy = cap == hipStreamCaptureStatusNone ? coop_sync_for(dev, s) : nullptr;
    if (sy) {
      const int ntiles = nblk >= 3 ? (nblk - 2) * (nblk - 1) / 2 : 0;
      const int n_owner_wg = (ntiles + SNW - 1) / SNW;
      const unsigned grid = 1u + (unsigned)n_owner_wg + (d_X ? (unsigned)nblk : 0u);
      hipLaunchKernelGGL((k_potrf_coop<real>), dim3(grid), dim3(SWG), potrf_coop_lds<real>(n, d_X != nullptr), s, n, d_A, lda, d_dinv, d_X, ldx, d_info, sy,
                         n_owner_wg);
      return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
    }
  }
  hipLaunchKernelGGL((k_potrf_small<real>), dim3(1), dim3(SWG), potrf_small_lds<real>(n), s, n, d_A, lda, d_dinv, d_info);
  if (d_X) {
    if (inv4)
      hipLaunchKernelGGL((k_tri_inv_small4<real>), dim3((unsigned)nblk), dim3(256), tri_inv_small4_lds<real>(n), s, n, (const real*)d_A, lda, (const real*)d_dinv,
                         d_X, ldx);
    else
      hipLaunchKernelGGL((k_tri_inv_small<real>), dim3((unsigned)nblk), dim3(256), tri_inv_small_lds<real>(n), s, n, (const real*)d_A, lda, (const real*)d_dinv,
                         d_X, ldx);
  }
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}
