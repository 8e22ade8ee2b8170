// Cholesky factorisation + explicit inverse of a SMALL matrix (n <= 512 / 480) in ONE launch of COOPERATING workgroups.
// Included by dense.hip after dense_small.h (uses its one-wave diagonal step, tile products and the inverse's block column).
//
// dense_small.h's one-workgroup factorisation is bound by one CU's matrix pipes: the trailing update of n = 327 alone is ~60 us
// of fp64 MFMA time on a single CU (64 cycles per v_mfma_f64_16x16x4), the serial chain of the 11 diagonal steps another
// 60 us, and the explicit inverse is a second launch that cannot start before the first has ended.  Here the three kinds of
// work run in different workgroups of one grid and hand data to each other through global memory: write-through stores, flags,
// loads that bypass the reader's L2 (workgroups sit on different XCDs: different L2s; see coh_store / coh_load below):
//   workgroup 0       the serial chain: per round the diagonal step (one wave), the panel L21 = A21 L11^-T, and the update of
//                     the NEXT block column with this panel (in place in LDS: that IS the next panel -- no global round trip);
//                     publishes ready[k] once block column k of L (and dinv[k]) is in global memory
//   owner waves       one wave per lower tile (I, J), J >= 2, of the trailing matrix: A_IJ -= sum_{k <= J-2} L_Ik L_Jk^T, each
//                     term as soon as ready[k] is up, accumulator in registers, operands straight from global (L2); the
//                     finished tile goes back to A and counts into done[J].  Workgroup 0 waits for done[k+1] (a full round
//                     after the last panel the owners needed) before it applies panel k to block column k + 1
//   inverse columns   block column J of X = L^-1 (k_tri_inv_small4's algorithm), block row I as soon as ready[I] is up: the
//                     inverse ends a few microseconds after the factorisation instead of 46 us
// The flags live in a small persistent scratch per (device, stream), zeroed at allocation; the last workgroup to finish (a ticket)
// zeroes them again.  Forward progress: workgroup 0 waits only for tiles whose inputs it has already published; owners and
// inverse columns wait only for workgroup 0; a workgroup that is not resident yet delays the others, it cannot deadlock them.
#pragma once

// Data handed from one workgroup to another travels by agent-scope relaxed atomic stores / loads (write-through stores, loads that
// do not trust this XCD's L2) and is ordered against its flag by s_waitcnt vmcnt(0) on the producer side: no buffer_wbl2 /
// buffer_inv on the chain (a release + acquire pair measured ~8 us per hand-over: four of them per round).
template <typename real>
__device__ __forceinline__ void coh_store(real* p, real v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename real>
__device__ __forceinline__ real coh_load(const real* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct CoopSync {
  int ready[16];   // block column k of L and dinv[k] are in global memory
  int done[16];    // owner tiles of block column J finished
  int ticket;      // workgroups finished
  int pad[31];
};

template <typename real>
static inline size_t potrf_coop_lds(int n, bool inverse) {
  const size_t a = potrf_small_lds<real>(n), b = inverse ? tri_inv_small4_lds<real>(n) : 0;
  return a > b ? a : b;
}

template <typename real>
__global__ __launch_bounds__(SWG) void k_potrf_coop(int n, real* A, int lda, real* dinv, real* X, int ldx, int32_t* __restrict__ info, CoopSync* sy,
                                                    int n_owner_wg) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  using acc_t = typename Acc4<real>::type;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int nblk = (n + SNB - 1) / SNB;
  const int l15 = lane & 15, l4 = lane >> 4;
  auto leave = [&]() {                                     // (one thread per workgroup) the last workgroup out resets the flags for the next call
    const int t = __hip_atomic_fetch_add(&sy->ticket, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (t == (int)gridDim.x - 1) {
      for (int i = 0; i < 16; ++i) {
        __hip_atomic_store(&sy->ready[i], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&sy->done[i], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __hip_atomic_store(&sy->ticket, 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  auto finish = [&]() {
    __syncthreads();
    if (tid == 0) leave();
  };

  if (blockIdx.x > (unsigned)n_owner_wg) {
    // ------------------------------------------------------------------------------------------ inverse column
    // (waves 4..7 have nothing to do and leave; the column's barriers count the waves still alive.  Its last flag read lies before the
    //  barrier that ends its last block row, so thread 0 may take the ticket as soon as it is through)
    if (w >= 4) return;
    tri_inv_column4<real>(n, A, lda, dinv, X, ldx, (int)blockIdx.x - 1 - n_owner_wg, sy->ready);
    if (tid == 0) leave();
    return;
  }

  if (blockIdx.x >= 1) {
    // ------------------------------------------------------------------------------------------ owner waves
    const int g = ((int)blockIdx.x - 1) * SNW + w;         // tiles in the order workgroup 0 needs them: column by column
    int I = -1, J = -1;
    {
      int rem = g;
      for (int jj = 2; jj < nblk; ++jj) {
        const int cnt = nblk - jj;
        if (rem < cnt) { J = jj; I = jj + rem; break; }
        rem -= cnt;
      }
    }
    if (J >= 0) {
      const int i0 = I * SNB, j0 = J * SNB;
      real cv[2][2][4];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = i0 + a * 16 + frag_row<real>(lane, r), j = j0 + b * 16 + l15;
            cv[a][b][r] = (i < n && j <= i) ? A[(int64_t)i * lda + j] : (real)0;
          }
      acc_t acc[2][2];
      zero_acc<real>(acc);
      // operand rows of this lane: A-operand i = i0 + a * 16 + l15, B-operand j = j0 + b * 16 + l15 (both rows of L); k = 4 ks + l4
      const real* pa[2];
      const real* pb[2];
      bool va[2], vb[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int i = i0 + a * 16 + l15, j = j0 + a * 16 + l15;
        va[a] = i < n;
        vb[a] = j < n;
        pa[a] = A + (int64_t)(va[a] ? i : 0) * lda + l4;
        pb[a] = A + (int64_t)(vb[a] ? j : 0) * lda + l4;
      }
      for (int k = 0; k + 2 <= J; ++k) {
        coop_wait_ge(&sy->ready[k], 1);
        real af[8][2], bf[8][2];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            const real x = coh_load(pa[a] + k * SNB + ks * 4), y = coh_load(pb[a] + k * SNB + ks * 4);
            af[ks][a] = va[a] ? x : (real)0;
            bf[ks][a] = vb[a] ? y : (real)0;
          }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = mfma16(af[ks][a], bf[ks][b], acc[a][b]);
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = i0 + a * 16 + frag_row<real>(lane, r), j = j0 + b * 16 + l15;
            if (i < n && j <= i) coh_store(&A[(int64_t)i * lda + j], cv[a][b][r] - acc[a][b][r]);
          }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the tile is out
      if (lane == 0) __hip_atomic_fetch_add(&sy->done[J], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    finish();
    return;
  }

  // ---------------------------------------------------------------------------------------------- workgroup 0: the chain
  real(*sD)[SLD] = reinterpret_cast<real(*)[SLD]>(smem_raw);
  real(*sI)[SLD] = sD + SNB;
  real(*sCol)[2 * SNB] = reinterpret_cast<real(*)[2 * SNB]>(sI + SNB);
  real(*sP)[SLD] = reinterpret_cast<real(*)[SLD]>(reinterpret_cast<real*>(sCol) + 4 * SNB + 2);   // the panel; its base moves down 32 rows per round
  const int lj = tid & 31;
  {
    const int nb = n < SNB ? n : SNB;
    for (int i = tid >> 5; i < SNB; i += SWG / 32) sD[i][lj] = (i < nb && lj < nb) ? A[(int64_t)i * lda + lj] : (i == lj ? (real)1 : (real)0);
    const int mt = n - nb, mtp = (mt + SNB - 1) / SNB * SNB;
    real v[8];
    for (int rb = 0; rb < mtp; rb += 8 * (SWG / 32)) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = rb + u * (SWG / 32) + (tid >> 5);
        v[u] = (r < mt && lj < nb) ? A[(int64_t)(nb + r) * lda + lj] : (real)0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = rb + u * (SWG / 32) + (tid >> 5);
        if (r < mtp) sP[r][lj] = v[u];
      }
    }
  }
  __syncthreads();
  for (int k0 = 0, blk = 0;; ++blk) {
    const int nb = n - k0 < SNB ? n - k0 : SNB;
    const int mt = n - k0 - nb;                              // rows below the diagonal block
    const int nt = (mt + SNB - 1) / SNB;
    const int nbn = mt < SNB ? mt : SNB;                     // size of the next diagonal block
    const int base = k0 + nb;                                // first row / column of the trailing matrix
    POTRF_STAMP(blk, 0);
    // ---- X: wave 0 factorises + inverts the diagonal block; the other waves fetch the tiles of block column blk + 1 that D1 will
    // update (rows of tiles ti = w - 1 and w + 6), once their owners are through with them
    real cva[2][2][4], cvb[2][2][4];
    bool have = false;
    auto load_tile = [&](int ti, real (&cv)[2][2][4]) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = ti * SNB + a * 16 + frag_row<real>(lane, r), j = b * 16 + l15;
            cv[a][b][r] = (i < mt && j <= i) ? coh_load(&A[(int64_t)(base + i) * lda + base + j]) : (real)0;
          }
    };
    if (w == 0) {
      const bool bad = wave_potrf32<real>(sD, sI, sCol, lane);
      if (bad && lane == 0) atomicOr(info, 1);
      POTRF_STAMP(blk, 1);
    } else if (mt > 0) {
      // (not a blocking wait: the owners' hand-over takes about as long as D1 + this diagonal step; if the tiles are not there yet
      //  they are fetched at D1 itself, after the panel products, instead of stalling the barrier below)
      const int need = blk + 1 >= 2 ? nblk - (blk + 1) : 0;
      have = need == 0 || coop_poll_ge(&sy->done[blk + 1], need);
      if (have) {
        if (w - 1 < nt) load_tile(w - 1, cva);
        if (w + 6 < nt) load_tile(w + 6, cvb);
      }
    }
    __syncthreads();
    POTRF_STAMP(blk, 2);
    // ---- factor block and its inverse (LDS) -> A, dinv
    for (int i = tid >> 5; i < SNB; i += SWG / 32) {
      if (i < nb && lj < nb) coh_store(&A[(int64_t)(k0 + i) * lda + k0 + lj], sD[i][lj]);
      coh_store(&dinv[(int64_t)blk * SNB * SNB + i * SNB + lj], sI[i][lj]);
    }
    // ---- C: panel L21 = A21 L11^-T, one 32-row tile per wave (in place in LDS, and to global)
    for (int rt = w; rt < nt; rt += SNW) {
      real(*sT)[SLD] = sP + blk * SNB + rt * SNB;
      acc_t acc[2][2];
      zero_acc<real>(acc);
      wave_tile32<real>(lane, SNB, [&](int i, int k) { return sT[i][k]; }, [&](int k, int j) { return sI[j][k]; }, acc);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();                      // every lane has read its operands of this tile
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int il = a * 16 + frag_row<real>(lane, r), j = b * 16 + l15;
            const int i = rt * SNB + il;
            sT[il][j] = acc[a][b][r];
            if (i < mt && j < nb) coh_store(&A[(int64_t)(base + i) * lda + k0 + j], acc[a][b][r]);
          }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's part of block column blk has reached memory
    __syncthreads();
    POTRF_STAMP(blk, 3);
    // block column blk of L, its diagonal block and dinv[blk] are complete: publish (owners, inverse columns)
    if (tid == 0) __hip_atomic_store(&sy->ready[blk], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (w > 0 && mt > 0) {
      if (!have) {
        coop_wait_ge(&sy->done[blk + 1], nblk - (blk + 1));
        if (w - 1 < nt) load_tile(w - 1, cva);
        if (w + 6 < nt) load_tile(w + 6, cvb);
      }
      // ---- D1: block column blk + 1 of the trailing matrix -= L21 L21(first 32 rows)^T: tile 0 is the next diagonal block (-> sD),
      // tile ti >= 1 is row block ti - 1 of the next panel: written over row block ti of this one (only this wave reads it)
      real(*sB)[SLD] = sP + blk * SNB;
      auto finish_tile = [&](int ti, const real (&cv)[2][2][4]) {
        real(*sT)[SLD] = sB + ti * SNB;
        acc_t acc[2][2];
        zero_acc<real>(acc);
        wave_tile32<real>(lane, SNB, [&](int i, int k) { return sT[i][k]; }, [&](int k, int j) { return sB[j][k]; }, acc);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int il = a * 16 + frag_row<real>(lane, r), jl = b * 16 + l15;
              const real v = cv[a][b][r] - acc[a][b][r];
              if (ti == 0) sD[il][jl] = (il < nbn && jl < nbn) ? v : (il == jl ? (real)1 : (real)0);
              else sT[il][jl] = (ti * SNB + il < mt && jl < nbn) ? v : (real)0;
            }
      };
      if (w - 1 < nt) finish_tile(w - 1, cva);
      if (w + 6 < nt) finish_tile(w + 6, cvb);
      for (int ti = w + 13; ti < nt; ti += SNW - 1) {        // (n > 480 only)
        real cv[2][2][4];
        load_tile(ti, cv);
        finish_tile(ti, cv);
      }
    }
    if (mt == 0) break;
    __syncthreads();
    POTRF_STAMP(blk, 4);
    k0 += nb;
  }
  // strict upper triangle: zero
  for (int i = tid >> 6; i < n; i += SNW)
    for (int j = i + 1 + lane; j < n; j += 64) A[(int64_t)i * lda + j] = (real)0;
  finish();
}
