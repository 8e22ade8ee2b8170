// Symmetric half-stencil SpMV, LDS-DMA pipelined form (d = 3, one right-hand side, fp32).
// Included by solve.hip after Vec4 / load4 / store4.
//
// Same product, storage and partial-vector contract as k_stencil_spmv4_sym (solve.hip), restructured
// around the one thing that kernel could not do: keep HBM requests in flight while a wave works.
//
//  * One 64-lane wave per workgroup owns 256 consecutive rows and one *part* of the half stencil: a run of
//    groups (= values of the leading two stencil digits) that share the dim-0 offset d0 = 0..3.  nparts = 4
//    (default) takes whole d0 chunks (7, 7, 7 and 4 groups, the centre group with 4 reals per row first): 1 956 waves,
//    all resident at once (8 per CU), the two chunks dispatched last at raised wave priority so that every chunk
//    finishes together.  nparts = 5..7 split 7-group chunks into 4 + 3 (a second dispatch round refills the slots the
//    early finishers free).  Every part writes its own direct partial vector (+ one atomically accumulated transposed
//    vector), so nparts + 1 partials in all.
//  * Every operand enters through `global_load_lds` (LDS-DMA, no VGPR destination): the 7 KB A_h tile of a
//    group (256 rows x 7 reals, one contiguous span: 7 x 1 KiB wave instructions) goes into an NST-deep ring
//    of LDS stages and the next tile is issued as soon as the current one has been copied to registers, so
//    NST tiles per wave stay in flight during the FMAs, the LDS window updates and the next wait.  The waits
//    are counted `s_waitcnt vmcnt(7 * tiles_ahead)`; the main loop contains no compiler-visible global
//    access (hipcc would drain the queue with vmcnt(0) at the first one).
//  * Within a part the 10-wide windows of v of all its groups lie inside one span of 256 + (ntile-1)*g2 + 10
//    elements, so v is fetched ONCE per part (three LDS-DMA copies from a 16-byte aligned origin) instead of
//    10 L2 loads per group; the transposed-term window has the same extent ([j & 3][j >> 2] planes with an
//    odd plane stride, so the lanes' 4-row groups update it conflict-free at any shift), is never moved, and
//    is flushed once at the end with coalesced fire-and-forget atomics.
//
// Measured at 50^3 (MI355X, tools/spmv_probe.py; A_h = 86 MB re-read every launch, i.e. served by the 256 MB Infinity
// Cache): 18.4..18.7 us back to back, 20.2 us per dispatch inside bench.py (k_stencil_spmv4_sym: 21.9).  History: plain
// DMA loads, nparts 6: 20.0 (20.5 / 21.4 with 7 / 4 parts; a 3-deep ring leaves 6 waves per CU: 22.8); "sc0 nt" on the
// A_h tiles: 19.8 (nparts 6), 19.2 (nparts 4); + wave priority for the late chunks: 18.4..18.7.  Where the rest goes
// (tools/ubench/stream_ubench.hip, WISKI_DMA_ABLATE builds, tools/dma_timing.py): an empty kernel already measures
// 4.0 us by the same clock; the bare DMA ring moves the 84 MB in 12.0 us (10.8 with sc0 nt); the wave timeline of this
// kernel spans 15.1 us: 1.5 us until the first tile lands, ~8 us at 8.5..9.7 TB/s, a 3 us tail in which the chunks run
// out (the 4-tile chunk ends at 8.5 us, the others at 11.6..13.2 median) and ~1.7 us of last-tile FMAs + window flush.
//
// A register-staged sibling (tiles in flight wait in VGPRs -- plain coalesced 16-byte loads, fully unrolled, the compiler's own
// vmcnt bookkeeping -- and one 7 KB LDS stage only re-tiles the landed tile) was built to put MORE tiles in flight per wave:
// 19.8 / 20.6 / 21.6 us with 2 / 3 / 4 tiles in flight against 18.3 us here.  More outstanding requests make it slower, not
// faster: at 8.6..9.7 TB/s the mid-phase of this kernel is already at what the fabric delivers for this pattern; the remaining
// time is start-up, tail and epilogue.
//
// Flushing the transposed window progressively (after tile t everything below (t + 1) * S1 is final: one ~4-granule atomic
// instruction per tile in the loop, 17 instead of 36 granules left for the epilogue) shortens the epilogue (0.92 -> 0.58 us per
// wave) but the atomics in the loop slow the stream (1 300 -> 1 050..1 170 tiles/us): 19.15 us against 18.45 us back to back.
//
// Round 4: XCD-contiguous row blocks.  Workgroups are dealt to the 8 XCDs round-robin by linear id; with blockIdx.x = row block every XCD
// streamed every 8th row block and pulled ALL of v through its own L2 (8 x 0.5 MB per launch).  The grid is now padded to 8 * ceil(nrb / 8)
// and workgroup b takes row block (b % 8) * ceil(nrb / 8) + b / 8: an XCD owns one contiguous eighth of the rows for all parts, fetches only
// that eighth of v (+ the 3-plane halo of the windows) and its neighbouring row blocks share window lines in ITS L2: 18.1 -> 17.7 us back
// to back (tools/spmv_probe.py, 200 launches), 21.3 -> 20.5 us per dispatch by bench.py's events; the delay / parts / ring-depth optima do
// not move (delay 0..18: 17.7..18.1; 5 / 6 parts 19.1 / 19.9; ring depth 3: 26.5); issuing the v windows before the first A_h tile instead of after
// it changes nothing, back to back or inside the solver (17.5..17.9 either way on one box, trace means 17.9..18.25).  Taken while A_h is Infinity-Cache resident (<= 192 MB; solve.hip: sym_xcd_map -- a stencil that streams from HBM prefers ONE interleaved address stream); WISKI_SYM_XCD=0 / 1 forces the plain / contiguous mapping, also for the LDS-window kernel k_stencil_spmv4_sym.
//
// Round 5: the four parts of a row block as the four waves of ONE workgroup (489 workgroups to dispatch and retire instead of 1 956; waves still
// independent, own LDS slices): 17.65 / 17.87 us against 17.63 / 18.33 back to back, 20.10 against 20.19 us per dispatch in bench.py -- nothing; not kept.
//
// Requires d == 3, m % 4 == 0.  part holds (nparts + 1) * m reals: part[y] = direct term of part y (plain stores),
// part[nparts] += transposed terms (must be zero on entry; re-zeroed by the consumer).
#pragma once

#include "lds_dma.h"   // glds_b128 / glds_b128_stream / glds_b32, wave_lgkm_fence, lds_addr

// counted wait on the vector-memory queue; n is wave-uniform and a multiple of 7 (instructions per tile)
__device__ __forceinline__ void wait_vm_tiles(int tiles_ahead) {
  if (tiles_ahead >= 3) asm volatile("s_waitcnt vmcnt(21)" ::: "memory");
  else if (tiles_ahead == 2) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
  else if (tiles_ahead == 1) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
constexpr int SYMDMA_TILE = 7 * 256;   // reals of one group tile

#ifndef WISKI_DMA_ABLATE
#define WISKI_DMA_ABLATE 0   // timing ablations (wrong results): 1 no flush atomics, 2 also no LDS window updates, 3 no arithmetic at all, 4 stores for atomics
#endif
#ifdef WISKI_DMA_TIMING   // per-wave phase stamps (100 MHz wall clock) of every wave: tools/spmv_probe.py --dma-timing
__device__ long long g_dma_dbg[4096 * 16];
#define DMA_STAMP(i) do { dma_ts[i] = wall_clock64(); } while (0)   // kept in registers, stored once at the end
extern "C" int wiski_dma_dbg(long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dma_dbg), sizeof(long long) * (size_t)n) == hipSuccess ? 0 : -1;
}
#else
#define DMA_STAMP(i) do {} while (0)
#endif

static inline int symdma_w4(int g2) { return ((256 + 6 * g2 + 10 + 3) / 4) | 1; }
static inline int symdma_wp(int g2) { return (4 * symdma_w4(g2) + 63) & ~63; }
// floats of the linear v-window image: 256-float (1 KiB) copies plus 64-float ones for the remainder
__host__ __device__ static inline int symdma_vwf(int g2) {
  const int need = 256 + 6 * g2 + 10 + 3, n128 = need / 256;
  return 256 * n128 + 64 * ((need - 256 * n128 + 63) / 64);
}
static inline size_t symdma_lds_bytes(int g2, int nst) {
  return (size_t)(nst * SYMDMA_TILE + 256 + symdma_vwf(g2) + symdma_wp(g2)) * sizeof(float);
}

// Explicit work split of a launch (stencil-sharded replicas, wiski_shard): part y covers the groups
// 7 d0[y] + p1lo[y] - 3 ... + ntile[y] of ONE leading stencil digit d0.  n == 0: the built-in splits below.
struct SymDmaParts {
  unsigned char d0[8], p1lo[8], ntile[8];
  int n;
};

template <int NST, bool DOT>
__global__ __launch_bounds__(64) void k_spmv_sym_dma(GridDev<float> G, const float* __restrict__ A_h, const float* __restrict__ V, int W4, int WP,
                                                     int nparts, float* __restrict__ part, const float* __restrict__ add, float beta,
                                                     double* __restrict__ dots, int delay, SymDmaParts tab, int xcd_rb,
                                                     unsigned long long* __restrict__ stamp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x;
  // measurement hook (wiski_prof_*; NULL otherwise): this wave's start and end by the 100 MHz wall clock.  The start is read here
  // and kept in a register; the pair leaves as ONE plain 16-byte store at the very end (no atomics: they cost 1.5 us per dispatch;
  // an atomic up here under `lane == 0` also costs the compiler its proof that the LDS-DMA destinations below are wave-uniform).
  // The host takes max(end) - min(start) over the waves of the dispatch (wiski_prof_stamps).
  const unsigned long long t_begin = stamp ? (unsigned long long)wall_clock64() : 0ull;
  const int m = G.m, S0 = G.stride[0], S1 = G.stride[1];
  // work units ("parts"), heaviest first in dispatch order.  nparts = 4: whole chunks d0 = 1, 2, 3, 0.
  // nparts = 7: the 7-group chunks split into digits 0..3 / 4..6 of the middle stencil digit (4 + 3 tiles).
  int d0, p1lo, ntile;
  {
    const int y = blockIdx.y;
    if (tab.n) { d0 = tab.d0[y]; p1lo = tab.p1lo[y]; ntile = tab.ntile[y]; }
    else if (nparts == 4) { d0 = (y + 1) & 3; p1lo = d0 == 0 ? 3 : 0; ntile = d0 == 0 ? 4 : 7; }
    else if (nparts == 5) {   // 7, 7, 4 (d0 = 3, digits 0..3), 4 (centre chunk), 3 (d0 = 3, digits 4..6)
      if (y < 2) { d0 = y + 1; p1lo = 0; ntile = 7; }
      else if (y == 2) { d0 = 3; p1lo = 0; ntile = 4; }
      else if (y == 3) { d0 = 0; p1lo = 3; ntile = 4; }
      else { d0 = 3; p1lo = 4; ntile = 3; }
    }
    else if (nparts == 6) {   // 7, 4, 4, 4 (centre), 3, 3
      if (y == 0) { d0 = 1; p1lo = 0; ntile = 7; }
      else if (y < 3) { d0 = y + 1; p1lo = 0; ntile = 4; }
      else if (y == 3) { d0 = 0; p1lo = 3; ntile = 4; }
      else { d0 = y - 2; p1lo = 4; ntile = 3; }
    }
    else if (y < 3) { d0 = y + 1; p1lo = 0; ntile = 4; }
    else if (y == 3) { d0 = 0; p1lo = 3; ntile = 4; }
    else { d0 = y - 3; p1lo = 4; ntile = 3; }
  }
  // All waves of a 4-part launch are resident at once (1 956 <= 8 per CU) and the CU arbitrates oldest-first: the chunks
  // dispatched last (y = 2: 7 tiles, y = 3: 4 tiles) get their first DMA issued 2.5 us after the others and finish last.
  // Raising their priority evens the finish times (19.3 -> 18.4..18.7 us; graded maps 0-1-2-3 / 0-1-2-0: no better).
  if (!tab.n && nparts == 4 && blockIdx.y >= 2) __builtin_amdgcn_s_setprio(3);
  // The 4-tile chunk (d0 = 0) carries half the bytes of the others: started with them it is gone after 8.7 us and the 7-tile
  // chunks stream on alone, then everybody's epilogue is exposed.  Started `delay` x 0.43 us late it joins when the others
  // are about three tiles in (50^3: delay = 12, 18.5 -> 18.1 us back to back; 6..8 and 14..20 are no better than 0, 24+ worse).
  if (!tab.n && nparts == 4 && blockIdx.y == 3)
    for (int i = 0; i < delay; ++i) __builtin_amdgcn_s_sleep(16);
  // xcd_rb != 0: grid.x is padded to 8 * xcd_rb (xcd_rb = ceil(row blocks / 8)) and workgroup b (on XCD b % 8) takes the (b / 8)-th row
  // block of ITS contiguous range, so every XCD sweeps one contiguous eighth of the rows (for all parts) and fetches only that eighth of v
  // (+ the 3-plane halo) into its L2.  The ranges are BALANCED -- nrb / 8 blocks each, the first nrb % 8 XCDs one more -- not xcd_rb
  // each with the remainder on the last XCD: at 50^3 that was 62 x 7 + 55 and the launch ended with the seven full XCDs.
  int rb = (int)blockIdx.x;
  if (xcd_rb) {
    const int nrb = (G.m + 255) >> 8, base = nrb >> 3, rem = nrb & 7;
    const int x = (int)(blockIdx.x & 7), sl = (int)(blockIdx.x >> 3);
    if (sl >= base + (x < rem ? 1 : 0)) return;
    rb = x * base + (x < rem ? x : rem) + sl;
  }
  const int iw0 = rb * 256;
  const int i4 = iw0 + 4 * lane;
  const bool live = i4 < m;
  const int nrows = m - iw0 < 256 ? m - iw0 : 256;
  float* __restrict__ stage = reinterpret_cast<float*>(smem);   // [NST][SYMDMA_TILE]
  float* __restrict__ xo_l = stage + NST * SYMDMA_TILE;         // [256]     v on this wave's own rows
  float* __restrict__ vw = xo_l + 256;                          // v on the chunk's window (linear image)
  float* __restrict__ tw = vw + symdma_vwf(S1);                 // [4][W4]   transposed-term window, plane-permuted ([j & 3][j >> 2])
  const int g0 = 7 * d0 + p1lo - 3;             // first group of the part (group = prefix code - centre code)
  const int wb = d0 * S0 + (p1lo - 3) * S1;     // window origin: row j of the window array is iw0 + wb - 3 + idx
  const int WL = 256 + (ntile - 1) * S1 + 10;
  const int vsh = (iw0 + wb - 3) & 3;           // the v window is copied from a 16-byte aligned origin
  const int vw128 = (WL + 3) / 256, vw32 = (WL + 3 - 256 * vw128 + 63) / 64;
  const unsigned stage_a = __builtin_amdgcn_readfirstlane(lds_addr(stage));
  const unsigned xo_a = __builtin_amdgcn_readfirstlane(lds_addr(xo_l));
  const unsigned vw_a = __builtin_amdgcn_readfirstlane(lds_addr(vw));

  auto issue_tile = [&](int t) {      // always 7 wave instructions (the counted waits rely on it)
    const int g = g0 + t;
    const float* __restrict__ src = g == 0 ? A_h + (int64_t)4 * iw0 : A_h + (int64_t)(7 * g - 3) * m + (int64_t)7 * iw0;
    const int lim = (g == 0 ? 4 : 7) * nrows;
    const unsigned dst = stage_a + (unsigned)((t % NST) * SYMDMA_TILE * sizeof(float));
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const int e = 4 * (64 * j + lane);
      glds_b128_stream(src + (e < lim ? e : 0), dst + 1024u * j);
    }
  };

#ifdef WISKI_DMA_TIMING
  long long dma_ts[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  DMA_STAMP(0);
  // ---- prologue: the first tile (the A_h stream is the critical path: its first request goes out before anything else),
  // own rows of v, the chunk's v window, the other NST - 1 tiles; zero the transposed window
  issue_tile(0);
  glds_b128(V + (live ? i4 : 0), xo_a);
  // v window, linear image: element c of the LDS array is v[jal + c], jal = (iw0 + wb - 3) rounded down to a
  // multiple of 4, so every lane copies one aligned 16-byte group (groups lie wholly inside or outside [0, m))
  // (1 KiB copies, then 256-byte ones for the remainder)
  const int jal = (iw0 + wb - 3) & ~3;
  for (int n = 0; n < vw128; ++n) {
    int j = jal + 4 * (64 * n + lane);
    j = j < 0 ? 0 : (j > m - 4 ? m - 4 : j);
    glds_b128(V + j, vw_a + 1024u * n);
  }
  for (int n = 0; n < vw32; ++n) {
    int j = jal + 256 * vw128 + 64 * n + lane;
    j = j < 0 ? 0 : (j > m - 1 ? m - 1 : j);
    glds_b32(V + j, vw_a + 1024u * vw128 + 256u * n);
  }
  DMA_STAMP(13);
#pragma unroll
  for (int t = 1; t < NST; ++t)
    if (t < ntile) issue_tile(t);
  DMA_STAMP(14);
  for (int e = lane; e < WP; e += 64) tw[e] = 0.f;
  // tile 0 and the windows have landed once at most the NST - 1 younger tiles are outstanding
  DMA_STAMP(1);
  wait_vm_tiles(NST - 1 < ntile - 1 ? NST - 1 : ntile - 1);
  DMA_STAMP(2);
  float xo[4];
  {
    const float4 x4 = *reinterpret_cast<const float4*>(xo_l + 4 * lane);
    xo[0] = live ? x4.x : 0.f; xo[1] = live ? x4.y : 0.f; xo[2] = live ? x4.z : 0.f; xo[3] = live ? x4.w : 0.f;
  }
  float acc[4] = {0.f, 0.f, 0.f, 0.f}, dg[4] = {0.f, 0.f, 0.f, 0.f};

  for (int t = 0; t < ntile; ++t) {
    const int g = g0 + t;
    const int rem = ntile - 1 - t;
    wait_vm_tiles(rem < NST - 1 ? rem : NST - 1);     // tile t has landed; younger tiles stay in flight
    DMA_STAMP(3 + t);
    const float* __restrict__ st = stage + (t % NST) * SYMDMA_TILE;
    float a[7][4];                                    // a[s][r]: row i4 + r, innermost offset digit s
    if (g == 0) {
      // centre group: 4 reals per row (digits 3..6); rows i4..i4+3 are 16 contiguous reals
      float4 q[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) q[r] = *reinterpret_cast<const float4*>(st + 16 * lane + 4 * r);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        a[0][r] = a[1][r] = a[2][r] = 0.f;
        a[3][r] = q[r].x; a[4][r] = q[r].y; a[5][r] = q[r].z; a[6][r] = q[r].w;
      }
    } else {
      float v[28];                                    // 7 x ds_read_b128 at a 112-byte lane stride: conflict-free
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        const float4 t4 = *reinterpret_cast<const float4*>(st + 28 * lane + 4 * j);
        v[4 * j + 0] = t4.x; v[4 * j + 1] = t4.y; v[4 * j + 2] = t4.z; v[4 * j + 3] = t4.w;
      }
#pragma unroll
      for (int s = 0; s < 7; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r) a[s][r] = v[7 * r + s];
    }
    const int w0 = 4 * lane + t * S1;                 // window index of row i4 + f - 3 (f - wb = t * S1)
    float win[10];
    {
      const float* __restrict__ wsrc = vw + (w0 + vsh);   // 4-way bank conflicts (lane stride 16 B): 10 x 8 LDS cycles per tile
#pragma unroll
      for (int e = 0; e < 10; ++e) win[e] = wsrc[e];
    }
    wave_lgkm_fence();                                 // the stage has been copied out: refill it
    if (t + NST < ntile) issue_tile(t + NST);
#if WISKI_DMA_ABLATE == 3
    if (a[0][0] + a[6][3] + win[0] + win[9] == 1234.5f) acc[0] += 1.f;
    if (false) {
#else
    if (live) {
#endif
      float tr[10];
#pragma unroll
      for (int e = 0; e < 10; ++e) tr[e] = 0.f;
#pragma unroll
      for (int s = 0; s < 7; ++s) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          acc[r] += a[s][r] * win[s + r];
          if (g == 0 && s == 3) dg[r] = a[s][r] * xo[r];          // the diagonal: counted once
          else tr[s + r] += a[s][r] * xo[r];
        }
      }
#if WISKI_DMA_ABLATE != 2 && WISKI_DMA_ABLATE != 3
      // transposed term -> wave-private LDS window, three lane-disjoint phases (window elements 0..3 / 4..7 /
      // 8..9 of lane l are the row quads of lanes l / l+1 / l+2; a wave's LDS operations execute in order)
#pragma unroll
      for (int ph = 0; ph < 3; ++ph) {
        // the 4 (2) cells of a phase sit in different planes: read them all, then write them all
        // (written as `*cell += x` the compiler serialises 10 LDS round trips per tile)
        constexpr int NE[3] = {4, 4, 2};
        float* cell[4];
        float old[4];
#pragma unroll
        for (int u = 0; u < NE[ph]; ++u) {
          const int idx = w0 + 4 * ph + u;
          cell[u] = tw + (idx & 3) * W4 + (idx >> 2);
          old[u] = *cell[u];
        }
#pragma unroll
        for (int u = 0; u < NE[ph]; ++u) *cell[u] = old[u] + tr[4 * ph + u];
        wave_lgkm_fence();
      }
#else
      if (tr[0] + tr[1] + tr[2] + tr[3] + tr[4] + tr[5] + tr[6] + tr[7] + tr[8] + tr[9] == 1234.5f) tw[lane] = 1.f;
#endif
    }
  }
  DMA_STAMP(10);
  // ---- epilogue (every DMA has been waited for): direct partial, window flush, p.Hp
  if (live) *reinterpret_cast<float4*>(part + (int64_t)blockIdx.y * m + i4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  {
    float* __restrict__ tacc = part + (int64_t)nparts * m;
    const int jbase = iw0 + wb - 3;
    for (int idx = lane; idx < WL; idx += 64) {
      const float v = tw[(idx & 3) * W4 + (idx >> 2)];
      const int j = jbase + idx;
#if WISKI_DMA_ABLATE != 1 && WISKI_DMA_ABLATE != 2 && WISKI_DMA_ABLATE != 4
      if (v != 0.f && j >= 0 && j < m) atomic_add_real(tacc + j, v);
#elif WISKI_DMA_ABLATE == 4     /* plain coalesced stores instead of atomics (wrong result, timing only) */
      if (j >= 0 && j < m) tacc[j] = v;
#else
      if (v == 1234.5f) tacc[j] = v;
#endif
    }
  }
  DMA_STAMP(11);
  if (DOT) {
    double pd = 0;
    if (live) {
#pragma unroll
      for (int r = 0; r < 4; ++r) pd += (double)xo[r] * (2.0 * (double)acc[r] - (double)dg[r]);
      if (g0 == 0 && add) {   // exactly one part starts at the centre group (in a sharded launch: on exactly one rank)
        const float4 ad = *reinterpret_cast<const float4*>(add + i4);
        pd += (double)beta * ((double)xo[0] * ad.x + (double)xo[1] * ad.y + (double)xo[2] * ad.z + (double)xo[3] * ad.w);
      }
    }
    pd = wave_reduce_sum<double>(pd);
    if (lane == 0) pcg_dot_add(dots, 0, pd);
  }
  DMA_STAMP(12);
  if (stamp && lane == 0) {
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    u64x2 pr;
    // where the wave ran, in the 16 bits above the 48-bit clock of the START word (wiski_prof_stamps masks them; read back by
    // wiski_prof_stamps_raw): HW_ID[15:4] = simd(2) pipe(2) cu(4) sh(1) se(3), XCC_ID[3:0] above them
    const unsigned hw = (__builtin_amdgcn_s_getreg(4 | (31 << 11)) >> 4) & 0xfffu, xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11)) & 0xfu;
    pr[0] = (t_begin & 0xffffffffffffull) | ((unsigned long long)(hw | (xcc << 12)) << 48);
    pr[1] = (unsigned long long)wall_clock64() & 0xffffffffffffull;
    *reinterpret_cast<u64x2*>(stamp + 2 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x)) = pr;
  }
#ifdef WISKI_DMA_TIMING
  if (lane == 0)
    for (int i = 0; i < 16; ++i) g_dma_dbg[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + i] = dma_ts[i];
#endif
}
