// Predictive interpolated MVM from stored (idx, val) rows, LDS-DMA staged form:  out[r] = sum_t val[r][t] * v[idx[r][t]].
// Included by interp_gather.hip.  reference call sites: BFN:206-210,235 (left_interp of the cached interpolation rows against
// pred_mean); SURVEY 8(d) row 1 -- T (4 + s) + s bytes per row (516 B at d = 3 fp32), HBM-bound: idx / val are read exactly
// once, v (m reals) lives in L2.
//
// k_gather_ell (interp_gather.hip) streams the rows through VGPRs: 4 rows per lane group are requested, then the dependent
// gathers of v wait behind them, then the wave starts over -- 0.41 of HBM on 2^20 rows.  Here the stream never touches a VGPR:
//
//  * one 64-lane wave per workgroup, tiles of P passes; a pass is 64 / LPR consecutive rows (LPR = T / 4 lanes per row, 4 taps
//    per lane), i.e. exactly 1 KiB of idx and 256 * sizeof(real) bytes of val -- one `global_load_lds_dwordx4 ... sc0 nt` wave
//    instruction per KiB, LDS image = memory image.  NST (= 2) stages: while tile k is consumed, tile k + 1 has landed or is
//    landing and tile k + 2 is requested into the stage tile k was copied out of.
//  * the vector-memory queue returns in issue order, and the gathers of v are vector-memory loads too: a gather issued behind
//    the refill waits for it.  So a tile's gathers go out BEFORE its stage is refilled, and the wave waits for them with a
//    counted `s_waitcnt vmcnt(<instructions of one refill>)`: the refill stays in flight across the wait and the arithmetic.
//    The gathers are inline asm as well -- the compiler sees no vector-memory load in the loop and inserts no wait of its own.
//    (Stores also count on vmcnt but complete out of order with respect to loads: a count that ignores them can only wait
//    longer than needed, never too short -- the tile's one store is not counted.)
//  * a lane's 4 taps are the innermost digits of one tap prefix: 4 consecutive grid indices for every row wiski_interp writes,
//    fetched with one dword-aligned 16-byte (fp64: 2 x 16) load; a tile in which any lane sees anything else takes four single
//    loads per lane and drains the queue (correct for any idx; never taken by the product).
//  * the LPR lanes of a row meet through DPP row shifts (no LDS round trip); the tile's sums are collected in LDS and leave as
//    one coalesced store.
//
// Tiles are dealt to the waves round-robin (tile = wave + k * waves): the launch reads ONE interleaved address stream per
// array, which HBM prefers to per-wave contiguous ranges.  Requires 16-byte aligned idx / val (the host falls back otherwise).
#pragma once

#include "lds_dma.h"

template <int N>
__device__ __forceinline__ void wait_vmcnt_imm() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field on gfx9");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Four gathered reals of one lane and pass, held in the register tuple(s) the load instruction names.  The compiler believes the
// values exist as soon as the asm statement has been issued; they do once wait_gathers() has returned, which ties every tuple
// to the wait so that no read of them (not even a sub-register copy) can be scheduled above it: elements are taken out with
// get() only after that.
template <typename real>
struct EllQuad;
template <>
struct EllQuad<float> {
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 t;
  __device__ __forceinline__ void load4(const float* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(t) : "v"(p) : "memory"); }
  __device__ __forceinline__ void load1(int j, const float* p) {      // slow path: one element (the caller drains the queue)
    float x;
    asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(x) : "v"(p) : "memory");
    t[j] = x;
  }
  __device__ __forceinline__ void tie() { asm volatile("" : "+v"(t)); }
  __device__ __forceinline__ float get(int j) const { return t[j]; }
};
template <>
struct EllQuad<double> {
  typedef double d2 __attribute__((ext_vector_type(2)));
  d2 t0, t1;
  __device__ __forceinline__ void load4(const double* p) {
    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16" : "=&v"(t0), "=&v"(t1) : "v"(p) : "memory");
  }
  __device__ __forceinline__ void load1(int j, const double* p) {
    double x;
    asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(x) : "v"(p) : "memory");
    if (j < 2) t0[j] = x; else t1[j - 2] = x;
  }
  __device__ __forceinline__ void tie() { asm volatile("" : "+v"(t0), "+v"(t1)); }
  __device__ __forceinline__ double get(int j) const { return j < 2 ? t0[j] : t1[j - 2]; }
};
// the counted wait that makes gathered registers usable
template <int N, typename real, int P>
__device__ __forceinline__ void wait_gathers(EllQuad<real> (&g)[P]) {
  wait_vmcnt_imm<N>();
#pragma unroll
  for (int p = 0; p < P; ++p) g[p].tie();
}

// sum over the LPR lanes of a row group; the total is valid in the group's LAST lane
template <typename real, int LPR>
__device__ __forceinline__ real ell_group_sum(real s) {
  if constexpr (LPR >= 2) s = wave_dpp_add<0x111, 0xf>(s);
  if constexpr (LPR >= 4) s = wave_dpp_add<0x112, 0xf>(s);
  if constexpr (LPR >= 8) s = wave_dpp_add<0x114, 0xf>(s);
  if constexpr (LPR >= 16) s = wave_dpp_add<0x118, 0xf>(s);
  if constexpr (LPR == 64) {
    s = wave_dpp_add<0x142, 0xa>(s);   // row_bcast15 into rows 1, 3
    s = wave_dpp_add<0x143, 0xc>(s);   // row_bcast31 into rows 2, 3: lane 63 holds the wave total
  }
  return s;
}

template <typename real, int LPR, int P>
struct EllDmaGeom {
  static constexpr int RPP = 64 / LPR;                       // rows per pass
  static constexpr int RPT = P * RPP;                        // rows per tile
  static constexpr int VI = (int)sizeof(real) / 4;           // val DMA instructions (KiB) per pass
  static constexpr int IPT = P * (1 + VI);                   // DMA instructions per tile
  static constexpr int GPT = P * VI;                         // gather instructions per tile (fast path)
  static constexpr unsigned IDX_B = P * 1024u;               // bytes of a stage's idx image
  static constexpr unsigned STAGE_B = P * 1024u * (1 + VI);
  static constexpr int NST = 2;
  static constexpr unsigned LDS_B = NST * STAGE_B + RPT * (unsigned)sizeof(real);
};

// ---- grid-aware form (wiski_gather_ell_grid): v re-laid so that every lane's four taps are ONE aligned 16-byte group ------------
// What bounds the kernel above is not the idx / val stream but the gathers of v: in the row-major v a row's T taps sit in T / 4
// different cache lines (one per tap prefix: 16 at d = 3, ~17.5 with the 16-byte groups that straddle a line), every one an L2
// request that queues in the CU's in-order vector-memory path behind the stream's HBM requests (measured, tools/gather_ell_probe.py:
// all gathers served by L1 -> 6.4 TB/s; a quarter of the accesses -> 6.1 TB/s; as they are -> 4.3-4.7 TB/s).  wiski_interp's rows are
// structured -- idx[tap] = base + sum_q c_q stride_q -- so for them v can be read from a BLOCKED copy in which the second-to-last
// dim K is cut into groups of 4, stored once for every alignment s = 0..3 of the stencil's first K index (4 x the memory of v: 2 MB
// at 50^3, L2-resident):
//     v4s[s][outer][b][jL][0..3] = v[outer][4 b + s + 0..3][jL]            (outer = the leading dims, jL the last one; zeros past gK)
// Lane (prefix, cL) of a row loads the ONE aligned 16-byte group (s = jK & 3, b = jK >> 2) at jL + cL: 16 accesses per row as in the
// plain form, but the four lanes of a prefix read 64 contiguous bytes, so a row touches 4^(d-2) x 1.4 = ~5.5 lines at d = 3 instead of
// ~17.5.  (Built first: K-blocks of 4 with block b + 1 fetched separately -- as many lines as it saves in L it adds in K, 112-118 us
// against 116 for the plain form; then blocks stored 8 wide, two loads per lane, 108-110 us: the cost is per 16-byte access as much as per
// line.)  fp64 alike (a group is 32 bytes, two loads per lane: 183 -> 149 us at 50^3).
struct EllV4Geo {
  unsigned gL, gK, nbk;      // sizes of the last and second-to-last dim, K blocks of 4
  unsigned mulL, shL;        // n / gL = mulL ? __umulhi(n, mulL) >> shL : n   (exact for 0 <= n < 2^31, Granlund-Montgomery)
  unsigned mulK, shK;
  unsigned groups;           // 16-byte groups per alignment copy: outer * nbk * gL
};
static inline void ell_magic(unsigned d, unsigned* mul, unsigned* sh) {
  if (d <= 1) { *mul = 0; *sh = 0; return; }
  unsigned L = 0;
  while ((1ull << L) < d) ++L;
  *mul = (unsigned)(((1ull << (31 + L)) + d - 1) / d);
  *sh = L - 1;
}
__device__ __forceinline__ unsigned ell_div(unsigned n, unsigned mul, unsigned sh) { return mul ? __umulhi(n, mul) >> sh : n; }

// v [m] (row-major grid vector) -> v4s [4][outer][nbk][gL][4]; one thread per 16-byte group
template <typename real>
__global__ __launch_bounds__(256) void k_ell_pack_v4s(const real* __restrict__ v, real* __restrict__ v4s, EllV4Geo geo, int64_t groups) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // (shift, group)
  if (e >= 4 * groups) return;
  const unsigned sft = (unsigned)(e / groups);
  const int64_t g = e - (int64_t)sft * groups;
  const unsigned jL = (unsigned)(g % geo.gL);
  const int64_t r = g / geo.gL;
  const unsigned b = (unsigned)(r % geo.nbk);
  const int64_t outer = r / geo.nbk;
  real q[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const unsigned jK = 4 * b + sft + t;
    q[t] = jK < geo.gK ? v[(outer * geo.gK + jK) * geo.gL + jL] : (real)0;
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) v4s[4 * e + t] = q[t];
}

template <typename real, int LPR, int P, bool V4 = false>
__global__ __launch_bounds__(64) void k_gather_ell_dma(const int32_t* __restrict__ idx, const real* __restrict__ val, int64_t n,
                                                       const real* __restrict__ v, real* __restrict__ out, int64_t ntiles, int contig,
                                                       EllV4Geo geo) {
  using Gm = EllDmaGeom<real, LPR, P>;
  constexpr int RPP = Gm::RPP, RPT = Gm::RPT, VI = Gm::VI, IPT = Gm::IPT, NST = Gm::NST;
  constexpr int T = 4 * LPR;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x;
  const unsigned stage_a = __builtin_amdgcn_readfirstlane(lds_addr(smem));
  real* __restrict__ obuf = reinterpret_cast<real*>(smem + NST * Gm::STAGE_B);
  const int64_t nwaves = gridDim.x;
  const int64_t w = blockIdx.x;
  // tiles of this wave: w, w + nwaves, ... (one interleaved address stream per array), or with contig the contiguous range
  // [w per, (w + 1) per)
#ifdef WISKI_ELL_ABLATE   // timing ablations (wrong results; tools/gather_ell_probe.py --ablate with a -DWISKI_ELL_ABLATE build): gathers served by L1,
  // no refill of the stages, the 4 lanes of a quad gather the same 16 bytes (a quarter of the accesses per row)
  const bool abl_gather = (contig & 2) != 0, abl_stream = (contig & 4) != 0, abl_quad = (contig & 8) != 0;
#else
  constexpr bool abl_gather = false, abl_stream = false, abl_quad = false;
#endif
  contig &= 1;
  const int64_t per = (ntiles + nwaves - 1) / nwaves;
  const int64_t nt = contig ? (w * per < ntiles ? (ntiles - w * per < per ? ntiles - w * per : per) : 0)
                            : (w < ntiles ? (ntiles - w + nwaves - 1) / nwaves : 0);
  const int64_t t0 = contig ? w * per : w, tstep = contig ? 1 : nwaves;

  auto issue_tile = [&](int64_t k) {       // always IPT wave instructions (the counted waits rely on it)
    const int64_t row0 = (t0 + k * tstep) * RPT;
    const unsigned dst = stage_a + (unsigned)(k % NST) * Gm::STAGE_B;
    constexpr int EPG = 16 / (int)sizeof(real);                // reals per 16-byte group
    if (row0 + RPT <= n) {                                     // (wave-uniform) a whole tile: lane l copies 16-byte group l of every KiB
      const int32_t* __restrict__ si = idx + row0 * T + 4 * lane;
      const real* __restrict__ sv = val + row0 * T + EPG * lane;
#pragma unroll
      for (int p = 0; p < P; ++p) glds_b128_stream(si + 256 * p, dst + 1024u * p);
#pragma unroll
      for (int j = 0; j < P * VI; ++j) glds_b128_stream(sv + 64 * EPG * j, dst + Gm::IDX_B + 1024u * j);
    } else {                                                   // the stream's last tile: groups of rows past the end read row 0 instead
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int64_t r = row0 + p * RPP + lane / LPR;
        glds_b128_stream(idx + (r < n ? (row0 + p * RPP) * T + 4 * lane : (int64_t)(4 * (lane % LPR))), dst + 1024u * p);
      }
#pragma unroll
      for (int j = 0; j < P * VI; ++j) {
        const int64_t e = (int64_t)(64 * j + lane) * EPG;        // element of the tile's val image
        glds_b128_stream(val + (row0 + e / T < n ? row0 * T + e : e % T), dst + Gm::IDX_B + 1024u * j);
      }
    }
  };

  if (nt == 0) return;
  issue_tile(0);
  if (nt > 1) issue_tile(1);
  for (int64_t k = 0; k < nt; ++k) {
    // tile k has landed once at most the younger tile is outstanding (after the first iteration the previous gather wait has
    // already seen to that: this is then a no-op)
    if (k + 1 < nt) wait_vmcnt_imm<IPT>(); else wait_vmcnt_imm<0>();
    const char* st = smem + (k % NST) * Gm::STAGE_B;
    if constexpr (V4) {
      // lane (row in pass, prefix, cL): idx of its tap (prefix, cK = 0, cL) and the four weights val[(prefix, cK, cL)], cK = 0..3
      static_assert(LPR >= 4, "the blocked form needs d >= 2");
      const int sub = lane % LPR;
      const int e0 = (lane / LPR) * T + (sub >> 2) * 16 + (sub & 3);
      int i0[P];
      real a[P][4];
#pragma unroll
      for (int p = 0; p < P; ++p) {
        i0[p] = reinterpret_cast<const int*>(st + 1024 * p)[e0];
        const real* av = reinterpret_cast<const real*>(st + Gm::IDX_B + 1024 * VI * p) + e0;
#pragma unroll
        for (int c = 0; c < 4; ++c) a[p][c] = av[4 * c];
      }
      wave_lgkm_fence();                       // the stage has been copied out
      EllQuad<real> g0[P];
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const unsigned q = ell_div((unsigned)i0[p], geo.mulL, geo.shL), jL = (unsigned)i0[p] - q * geo.gL;
        const unsigned o = ell_div(q, geo.mulK, geo.shK), jK = q - o * geo.gK;
        g0[p].load4(v + 4 * ((size_t)(jK & 3) * geo.groups + ((size_t)o * geo.nbk + (jK >> 2)) * geo.gL + jL));
      }
      if (k + NST < nt) {
        issue_tile(k + NST);
        wait_vmcnt_imm<IPT>();
      } else {
        wait_vmcnt_imm<0>();
      }
#pragma unroll
      for (int p = 0; p < P; ++p) g0[p].tie();
      const int64_t row0 = (t0 + k * tstep) * RPT;
#pragma unroll
      for (int p = 0; p < P; ++p) {
        real s = a[p][0] * g0[p].get(0) + a[p][1] * g0[p].get(1) + a[p][2] * g0[p].get(2) + a[p][3] * g0[p].get(3);
        s = ell_group_sum<real, LPR>(s);
        if (lane % LPR == LPR - 1) obuf[p * RPP + lane / LPR] = s;
      }
      wave_lgkm_fence();
      for (int r = lane; r < RPT; r += 64)
        if (row0 + r < n) out[row0 + r] = obuf[r];
      wave_lgkm_fence();
      continue;
    }
    int4 id[P];
    real a[P][4];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      id[p] = *reinterpret_cast<const int4*>(st + 1024 * p + 16 * lane);
      const real* av = reinterpret_cast<const real*>(st + Gm::IDX_B + 1024 * VI * p) + 4 * lane;
      if constexpr (sizeof(real) == 4) {
        const float4 q = *reinterpret_cast<const float4*>(av);
        a[p][0] = q.x; a[p][1] = q.y; a[p][2] = q.z; a[p][3] = q.w;
      } else {
        const double2 q0 = *reinterpret_cast<const double2*>(av), q1 = *reinterpret_cast<const double2*>(av + 2);
        a[p][0] = q0.x; a[p][1] = q0.y; a[p][2] = q1.x; a[p][3] = q1.y;
      }
    }
    wave_lgkm_fence();                       // the stage has been copied out
    int broken = 0;                           // (bitwise: no branches)
#pragma unroll
    for (int p = 0; p < P; ++p) broken |= (id[p].y - id[p].x - 1) | (id[p].z - id[p].x - 2) | (id[p].w - id[p].x - 3);
    EllQuad<real> g[P];
    const bool refill = k + NST < nt;
    if (__builtin_amdgcn_ballot_w64(broken != 0) == 0) {
#pragma unroll
      for (int p = 0; p < P; ++p) {
        int ix = id[p].x;
        if (abl_quad) ix = __builtin_amdgcn_mov_dpp(ix, 0x00, 0xf, 0xf, true);   // quad_perm [0,0,0,0]
        g[p].load4(v + (abl_gather ? 4 * (lane & 15) : ix));
      }
      if (refill && !abl_stream) { issue_tile(k + NST); wait_gathers<IPT>(g); } else wait_gathers<0>(g);
    } else {
      // arbitrary indices (never produced by wiski_interp): single loads, each waited for on the spot (this drains the queue;
      // the counted waits that follow can then only wait longer than needed)
#pragma unroll
      for (int p = 0; p < P; ++p) {
        g[p].load1(0, v + id[p].x); g[p].load1(1, v + id[p].y);
        g[p].load1(2, v + id[p].z); g[p].load1(3, v + id[p].w);
      }
      if (refill) issue_tile(k + NST);
    }
    const int64_t row0 = (t0 + k * tstep) * RPT;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      real s = a[p][0] * g[p].get(0) + a[p][1] * g[p].get(1) + a[p][2] * g[p].get(2) + a[p][3] * g[p].get(3);
      s = ell_group_sum<real, LPR>(s);
      if (lane % LPR == LPR - 1) obuf[p * RPP + lane / LPR] = s;
    }
    wave_lgkm_fence();                       // (one wave: its LDS operations execute in order; the fence is for the reads below)
    for (int r = lane; r < RPT; r += 64)
      if (row0 + r < n) out[row0 + r] = obuf[r];
    wave_lgkm_fence();                       // obuf is rewritten by the next tile
  }
}
