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
// inverse.  Step k: the pivot by a lane read, a_k = v[k] / sqrt(pivot) in every lane (the factor's column entry l_tk in the
// lower half, the inverse's entry x_kt in the upper half), column k of the factor published through LDS, and
// v[j] -= a_k l_jk for j > k -- which is the rank-1 update of the factorisation in the lower half and the substitution step in
// the upper half.  The inverse costs no extra instructions.  Returns true if a pivot was not positive.
template <typename real>
__device__ __forceinline__ bool wave_potrf32(real (*sD)[SLD], real (*sI)[SLD], real (*sCol)[2 * SNB], int lane) {
  const int t = lane & 31;
  const bool fac = lane < SNB;
  // (an opaque zero added to the LDS bases: the ~600 constant addresses of the unrolled loop below then stay immediate offsets of
  // one base register instead of being materialised and hoisted out of the caller's loop, which spilled ~500 registers)
  int opaque = 0;
  asm volatile("" : "+v"(opaque));
  sCol += opaque;
  sD += opaque;
  sI += opaque;
  real v[SNB];
#pragma unroll
  for (int j = 0; j < SNB; ++j) v[j] = fac ? sD[t][j] : (j == t ? (real)1 : (real)0);
  bool bad = false;
#pragma unroll
  for (int k = 0; k < SNB; ++k) {
    const real piv0 = read_lane(v[k], k);
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
    const real ak = (fac && t < k) ? (real)0 : v[k] * ri;      // rows above the pivot take no part (their v[k] is upper-triangle data)
    v[k] = ak;
    sCol[k & 1][lane] = ak;                      // (unconditional: slots 32..63 take the upper half's values and are never read; a branch here
                                                 //  splits the block and the compiler then sinks the updates below to their distant uses -- spills)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = k + 1; j < SNB; ++j) v[j] -= ak * sCol[k & 1][j];
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

// A (n x n, lda) -> lower Cholesky factor in place (strict upper zeroed), inverses of the 32 x 32 diagonal blocks of the
// factor in dinv [nblk][32][32] (identity-padded).  One workgroup of SWG threads.  Dynamic LDS: (nrow_pad + 64) * SLD + 128 reals.
template <typename real>
__global__ __launch_bounds__(SWG) void k_potrf_small(int n, real* __restrict__ A, int lda, real* __restrict__ dinv, int32_t* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  real(*sD)[SLD] = reinterpret_cast<real(*)[SLD]>(smem_raw);
  real(*sI)[SLD] = sD + SNB;
  real(*sCol)[2 * SNB] = reinterpret_cast<real(*)[2 * SNB]>(sI + SNB);
  real(*sP)[SLD] = reinterpret_cast<real(*)[SLD]>(reinterpret_cast<real*>(sCol) + 4 * SNB);
  using acc_t = typename Acc4<real>::type;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  constexpr int RP = SWG / 32;                               // rows per pass of the cooperative loads (32 columns x RP rows)
  const int lj = tid & 31, li0 = tid >> 5;
  // panel below the diagonal block at k0 (rows k0 + nb .., columns k0 .. k0 + nb) -> sP, zero-padded to whole tiles.  Loads in
  // batches of 8 per thread: issued back to back, one wait (a load-store loop pays the L2 latency per row).
  auto load_panel = [&](int k0, int nb, int mt, int mtp) {
    real v[8];
    for (int rb = 0; rb < mtp; rb += 8 * RP) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = rb + u * RP + li0;
        v[u] = (r < mt && lj < nb) ? A[(int64_t)(k0 + nb + r) * lda + k0 + lj] : (real)0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = rb + u * RP + li0;
        if (r < mtp) sP[r][lj] = v[u];
      }
    }
  };
  auto store_diag = [&](int k0, int nb, int blk) {           // factor block and its inverse (LDS) -> A, dinv
    for (int i = li0; i < SNB; i += RP) {
      if (i < nb && lj < nb) A[(int64_t)(k0 + i) * lda + k0 + lj] = sD[i][lj];
      dinv[(int64_t)blk * SNB * SNB + i * SNB + lj] = sI[i][lj];
    }
  };
  // ---- prologue: first diagonal block factorised, first panel loaded
  {
    const int nb = n < SNB ? n : SNB;
    for (int i = li0; i < SNB; i += RP) sD[i][lj] = (i < nb && lj < nb) ? A[(int64_t)i * lda + lj] : (i == lj ? (real)1 : (real)0);
    const int mt = n - nb;
    load_panel(0, nb, mt, (mt + SNB - 1) / SNB * SNB);
    __syncthreads();
    if (w == 0) {
      const bool bad = wave_potrf32<real>(sD, sI, sCol, lane);
      if (bad && lane == 0) atomicOr(info, 1);
    }
    __syncthreads();
    store_diag(0, nb, 0);
  }
  // Each round: panel of block k (C), then the first block column of the trailing update (D1: it yields the NEXT diagonal block,
  // kept in LDS, and the next panel, to global), then -- concurrently -- wave 0 factorises that next diagonal block while the
  // other waves do the rest of the trailing update (D2): the one-wave factorisation (~9 us) is the longest serial piece of a round
  // and now hides behind the tile products.
  for (int k0 = 0, blk = 0; k0 < n; k0 += SNB, ++blk) {
    const int nb = n - k0 < SNB ? n - k0 : SNB;
    const int mt = n - k0 - nb;                              // rows below the diagonal block
    if (mt == 0) break;
    const int mtp = (mt + SNB - 1) / SNB * SNB;
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
    const int nt = mtp / SNB;
    const int nbn = mt < SNB ? mt : SNB;                     // size of the next diagonal block
    // one lower tile (ti >= tj) of A22 -= L21 L21^T; the tile of A22 first (16 independent loads in flight under the MFMA loop)
    auto load_tile = [&](int ti, int tj, real (&cv)[2][2][4]) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = ti * SNB + a * 16 + frag_row<real>(lane, r), j = tj * SNB + b * 16 + (lane & 15);
            cv[a][b][r] = (i < mt && j <= i) ? A[(int64_t)(k0 + nb + i) * lda + k0 + nb + j] : (real)0;
          }
    };
    auto finish_tile = [&](int ti, int tj, const real (&cv)[2][2][4]) {
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
            if (ti == 0 && tj == 0) sD[il][jl] = (il < nbn && jl < nbn) ? v : (il == jl ? (real)1 : (real)0);   // next diagonal block: stays in LDS
            else if (i < mt && j <= i) A[(int64_t)(k0 + nb + i) * lda + k0 + nb + j] = v;
          }
    };
    auto update_tile = [&](int ti, int tj) {
      real cv[2][2][4];
      load_tile(ti, tj, cv);
      finish_tile(ti, tj, cv);
    };
    // ---- D1: first block column of the update
    for (int ti = w; ti < nt; ti += SNW) update_tile(ti, 0);
    __threadfence_block();
    __syncthreads();
    // ---- B (wave 0: next diagonal block) alongside D2 (the other waves: tiles with tj >= 1)
    if (w == 0) {
      const bool bad = wave_potrf32<real>(sD, sI, sCol, lane);
      if (bad && lane == 0) atomicOr(info, 1);
    } else {
      const int ntile2 = (nt - 1) * nt / 2;                  // lower triangle of the (nt - 1) x (nt - 1) tiles with ti >= tj >= 1
      // (the next tile's 16 loads are issued before this tile's products: with two waves per SIMD little else hides the L2 latency)
      auto decode = [&](int tl, int& ti, int& tj) {
        int t = 0, rem = tl;
        while (rem > t) { rem -= t + 1; ++t; }
        ti = t + 1; tj = rem + 1;
      };
      int tl = w - 1, ti = 0, tj = 0;
      if constexpr (sizeof(real) == 8) {                     // (fp64: two tiles of C in registers beside the diagonal step's spill: plain loop)
        for (; tl < ntile2; tl += SNW - 1) {
          decode(tl, ti, tj);
          update_tile(ti, tj);
        }
      } else {
      real cva[2][2][4], cvb[2][2][4];
      if (tl < ntile2) { decode(tl, ti, tj); load_tile(ti, tj, cva); }
      while (tl < ntile2) {
        const int tn = tl + SNW - 1;
        int ti2 = 0, tj2 = 0;
        if (tn < ntile2) { decode(tn, ti2, tj2); load_tile(ti2, tj2, cvb); }
        finish_tile(ti, tj, cva);
        tl = tn;
        if (tl < ntile2) {
          const int tn2 = tl + SNW - 1;
          int ti3 = 0, tj3 = 0;
          if (tn2 < ntile2) { decode(tn2, ti3, tj3); load_tile(ti3, tj3, cva); }
          finish_tile(ti2, tj2, cvb);
          tl = tn2; ti = ti3; tj = tj3;
        }
      }
      }
    }
    __threadfence_block();
    __syncthreads();
    // ---- next round's inputs: factor block to global, panel below it into LDS
    {
      const int k1 = k0 + nb, mt1 = n - k1 - nbn;
      store_diag(k1, nbn, blk + 1);
      load_panel(k1, nbn, mt1, (mt1 + SNB - 1) / SNB * SNB);
    }
    __syncthreads();
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
template <typename real>
__global__ __launch_bounds__(256) void k_tri_inv_small4(int n, const real* __restrict__ L, int ldl, const real* __restrict__ dinv, real* __restrict__ X, int ldx) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  real(*sW)[SNB][SLD] = reinterpret_cast<real(*)[SNB][SLD]>(smem_raw);      // [4] wave-private: staged L tile, then the wave's partial
  real(*sS)[SLD] = sW[0];                                                    // summed partials  (take over regions 0 and 1 once the partials
  real(*sDI)[SLD] = sW[1];                                                   // dinv[I]           have been read: same element, same thread)
  real(*sX)[SLDX] = reinterpret_cast<real(*)[SLDX]>(sW + 4);                // block column J of X, rows from J * 32 (stride 33: n = 480 fits in fp64)
  using acc_t = typename Acc4<real>::type;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int J = blockIdx.x, j0 = J * SNB;
  const int nblk = (n + SNB - 1) / SNB;
  const int nbj = n - j0 < SNB ? n - j0 : SNB;
  const int qa = w >> 1, qb = w & 1;
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
    acc_t acc[2][2];
    zero_acc<real>(acc);
    for (int K = J + w; K < I; K += 4) {
      // stage L[I][K] (32 x 32) into this wave's region: 16 elements per lane, rows of 256 B
      real v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int e = u * 64 + lane, i = e >> 5, k = e & 31;
        v[u] = i < nbi ? L[(int64_t)(i0 + i) * ldl + K * SNB + k] : (real)0;
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
      const real di = dinv[(int64_t)I * SNB * SNB + e];
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
static inline size_t tri_inv_small4_lds(int n) {
  const int mtp = (n + SNB - 1) / SNB * SNB;
  return (size_t)(mtp * SLDX + 4 * SNB * SLD) * sizeof(real);
}

template <typename real>
static inline size_t potrf_small_lds(int n) {
  const int mtp = (n + SNB - 1) / SNB * SNB;
  return (size_t)((mtp + 2 * SNB) * SLD + 4 * SNB) * sizeof(real);
}
template <typename real>
static inline size_t tri_inv_small_lds(int n) {
  const int mtp = (n + SNB - 1) / SNB * SNB;
  return (size_t)((mtp + 3 * SNB) * SLD) * sizeof(real);
}

constexpr int WISKI_SMALL_UNAVAILABLE = 1;   // (internal) the one-workgroup path cannot run on this device: use the blocked one
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
                      hipFuncSetAttribute((const void*)k_tri_inv_small<real>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tri_inv_small_lds<real>(SMALL_N_MAX)) == hipSuccess &&
                      hipFuncSetAttribute((const void*)k_tri_inv_small4<real>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMALL_LDS_MAX) == hipSuccess;
      if (!ok) (void)hipGetLastError();
      state[dev] = ok ? 1 : -1;
    }
    if (state[dev] < 0) return WISKI_SMALL_UNAVAILABLE;
  }
  hipLaunchKernelGGL((k_potrf_small<real>), dim3(1), dim3(SWG), potrf_small_lds<real>(n), s, n, d_A, lda, d_dinv, d_info);
  if (d_X) {
    const int nblk = (n + SNB - 1) / SNB;
    if (tri_inv_small4_lds<real>(n) <= SMALL_LDS_MAX)
      hipLaunchKernelGGL((k_tri_inv_small4<real>), dim3((unsigned)nblk), dim3(256), tri_inv_small4_lds<real>(n), s, n, (const real*)d_A, lda, (const real*)d_dinv,
                         d_X, ldx);
    else
      hipLaunchKernelGGL((k_tri_inv_small<real>), dim3((unsigned)nblk), dim3(256), tri_inv_small_lds<real>(n), s, n, (const real*)d_A, lda, (const real*)d_dinv,
                         d_X, ldx);
  }
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}
