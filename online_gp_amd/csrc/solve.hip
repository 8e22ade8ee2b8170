// Inducing-space operators and the preconditioned-CG posterior solve.
//   stencil SpMV  : WtW @ V                (URLT:47-48) on the block-stencil form
//   Kron-Toeplitz : Kuu @ V                (BFN:334-348,363-366)
//   PCG           : (Kt^-1 + A)^-1 RHS     (CG branch of BFN:368-383)
#include "wiski_common.h"

#include <hip/hip_ext.h>

#include <chrono>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

// ------------------------------------------------------- profiling hook ---
// Optional HIP-event bracket around every stencil-SpMV launch, recorded on the
// stream the kernel runs on (bench.py's roofline leg).  Off by default.
static struct WiskiProf {
  bool on = false;       // events are attached to the dispatches (wiski_prof_enable)
  bool armed = false;    // between wiski_prof_start and wiski_prof_stop: every k_spmv_sym_dma dispatch is stamped, events or not
  std::vector<hipEvent_t> ev;
  size_t used = 0;
  // in-kernel stamps (k_spmv_sym_dma): every wave of a recorded dispatch stores its own (start, end) pair of the 100 MHz wall
  // clock with ONE plain 16-byte store at its very end -- no atomics (a first version funnelled min / max atomics into 64 words
  // per dispatch: +1.5 us per dispatch by the events).  The events above bracket [predecessor complete -> this kernel complete]
  // and so contain the dispatch latency in front of the first wave; max(end) - min(start) over the waves does not.
  unsigned long long* d_stamp = nullptr;     // arena of pairs
  size_t stamp_pairs = 0, stamp_used = 0;
  struct Span { size_t off, n; };
  std::vector<Span> spans;                   // per stamped dispatch, in launch order
} g_prof;
constexpr unsigned long long PROF_CLOCK_MASK = 0xffffffffffffull;   // the stamps keep 48 bits of the 100 MHz clock
constexpr size_t PROF_STAMP_ARENA_PAIRS = (size_t)1 << 20;   // 16 MB: 256 dispatches of 4096 waves

extern "C" int wiski_prof_start(int max_launches) {
  if (max_launches < 1) return WISKI_E_BADARG;
  while (g_prof.ev.size() < (size_t)max_launches * 2) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return WISKI_E_LAUNCH;
    g_prof.ev.push_back(e);
  }
  if (!g_prof.d_stamp) {
    if (hipMalloc(&g_prof.d_stamp, PROF_STAMP_ARENA_PAIRS * 2 * sizeof(unsigned long long)) == hipSuccess) g_prof.stamp_pairs = PROF_STAMP_ARENA_PAIRS;
    else { (void)hipGetLastError(); g_prof.d_stamp = nullptr; }
  }
  // (synchronous: this call sits outside every timed region) a pair whose end word is still 0 was never written
  if (g_prof.d_stamp && g_prof.stamp_used &&
      hipMemset(g_prof.d_stamp, 0, g_prof.stamp_used * 2 * sizeof(unsigned long long)) != hipSuccess) return WISKI_E_LAUNCH;
  if (g_prof.d_stamp && !g_prof.stamp_used && hipMemset(g_prof.d_stamp, 0, g_prof.stamp_pairs * 2 * sizeof(unsigned long long)) != hipSuccess) return WISKI_E_LAUNCH;
  g_prof.stamp_used = 0;
  g_prof.spans.clear();
  g_prof.used = 0;
  g_prof.on = true;
  g_prof.armed = true;
  return WISKI_OK;
}

extern "C" int wiski_prof_enable(int32_t on) {
  g_prof.on = on != 0;
  return WISKI_OK;
}

// the stamp pairs (one per wave) of the dispatch launch_timed is about to record, or NULL: not recording / no room
static inline unsigned long long* prof_stamp_pairs(size_t nwaves) {
  static const bool off = getenv("WISKI_PROF_NOSTAMP") != nullptr;   // A/B of what the stamps themselves cost
  if (off || !g_prof.armed || !g_prof.d_stamp || g_prof.stamp_used + nwaves > g_prof.stamp_pairs) return nullptr;
  g_prof.spans.push_back(WiskiProf::Span{g_prof.stamp_used, nwaves});
  g_prof.stamp_used += nwaves;
  return g_prof.d_stamp + 2 * g_prof.spans.back().off;
}

// Stops recording; the caller must have synchronised the stream(s).
extern "C" int wiski_prof_stop(double* total_ms, int64_t* launches) {
  g_prof.on = false;
  g_prof.armed = false;
  double tot = 0;
  for (size_t i = 0; i + 1 < g_prof.used; i += 2) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, g_prof.ev[i], g_prof.ev[i + 1]) != hipSuccess) return WISKI_E_LAUNCH;
    tot += ms;
  }
  if (total_ms) *total_ms = tot;
  if (launches) *launches = (int64_t)(g_prof.used / 2);
  g_prof.used = 0;
  return WISKI_OK;
}

// By in-kernel stamps: sum over the stamped dispatches of (latest "wave finished" - earliest "wave started"), 100 MHz wall clock.
// EVERY k_spmv_sym_dma dispatch between wiski_prof_start and wiski_prof_stop is stamped (the stamps cost nothing measurable), whether
// wiski_prof_enable has the events attached or not -- an unbiased sample where the events, which do cost, cover a subset.  The stamps
// stay readable until the next wiski_prof_start; the stream must be synchronised.
extern "C" int wiski_prof_stamps(double* total_ms, int64_t* launches, double* each_us, int64_t each_cap) {
  const size_t nd = g_prof.spans.size();
  double tot = 0;
  int64_t cnt = 0;
  if (nd && g_prof.d_stamp && g_prof.stamp_used) {
    std::vector<unsigned long long> h(2 * g_prof.stamp_used);
    if (hipMemcpy(h.data(), g_prof.d_stamp, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return WISKI_E_LAUNCH;
    for (size_t i = 0; i < nd; ++i) {
      const WiskiProf::Span sp = g_prof.spans[i];
      unsigned long long lo = ~0ull, hi = 0;
      for (size_t w = 0; w < sp.n; ++w) {
        unsigned long long b = h[2 * (sp.off + w)] & PROF_CLOCK_MASK, e = h[2 * (sp.off + w) + 1];   // (bits 48.. of the start word: placement)
        if (!e) continue;                      // (a padded workgroup that returned at once)
        if (e < b) e += PROF_CLOCK_MASK + 1;   // the 48-bit clock wrapped between the two reads (once in 32 days)
        if (b < lo) lo = b;
        if (e > hi) hi = e;
      }
      if (hi > lo) {
        if (each_us && cnt < each_cap) each_us[cnt] = (double)(hi - lo) * 1e-2;   // 10 ns ticks -> us
        tot += (double)(hi - lo) * 1e-5;                                         // -> ms
        ++cnt;
      }
    }
  }
  if (total_ms) *total_ms = tot;
  if (launches) *launches = cnt;
  return WISKI_OK;
}

// The raw pairs of recorded dispatch i (same lifetime as wiski_prof_stamps): pairs[2w] = start | placement << 48 (simd(2) pipe(2)
// cu(4) sh(1) se(3) xcc(4), low to high), pairs[2w+1] = end (0: the workgroup was padding), w = blockIdx.y * gridDim.x +
// blockIdx.x.  *nwaves: the dispatch's wave count (0: not stamped).  For tools/stamp_report.py; nothing in the product reads it.
extern "C" int wiski_prof_stamps_raw(int64_t i, uint64_t* pairs, int64_t cap, int64_t* nwaves) {
  if (!nwaves || i < 0) return WISKI_E_BADARG;
  *nwaves = 0;
  if ((size_t)i >= g_prof.spans.size() || !g_prof.d_stamp) return WISKI_OK;
  const WiskiProf::Span sp = g_prof.spans[(size_t)i];
  *nwaves = (int64_t)sp.n;
  if (!pairs || !sp.n) return WISKI_OK;
  const size_t n = (size_t)cap < sp.n ? (size_t)cap : sp.n;
  if (hipMemcpy(pairs, g_prof.d_stamp + 2 * sp.off, n * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return WISKI_E_LAUNCH;
  return WISKI_OK;
}

// What an EMPTY dispatch costs by the same clock (start / stop events attached to the dispatch packet): the floor every
// per-dispatch kernel time measured through launch_timed contains.  Average over n launches, microseconds.
__global__ void k_prof_empty() {}
extern "C" int wiski_prof_empty(int32_t n, double* avg_us, void* stream) {
  if (n < 1 || !avg_us) return WISKI_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  std::vector<hipEvent_t> ev(2 * (size_t)n);
  for (auto& e : ev)
    if (hipEventCreate(&e) != hipSuccess) return WISKI_E_LAUNCH;
  for (int i = 0; i < n; ++i) hipExtLaunchKernelGGL(k_prof_empty, dim3(1), dim3(64), 0, s, ev[2 * i], ev[2 * i + 1], 0);
  int rc = hipStreamSynchronize(s) == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
  double tot = 0;
  for (int i = 0; i < n && rc == WISKI_OK; ++i) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]) != hipSuccess) rc = WISKI_E_LAUNCH;
    tot += ms;
  }
  for (auto& e : ev) (void)hipEventDestroy(e);
  *avg_us = tot * 1e3 / n;
  return rc;
}

// ---------------------------------------------------------- stencil SpMV ---
// out[c][i] = beta*add[c][i] + sum_o A_st[o][i] * V[c][clamp(i + off(o))]
// HBM-bound: A_st (R*m reals) is streamed exactly once, fully coalesced along
// i; V (k*m reals) is re-read R times but stays L2 / MALL resident.  Entries
// whose neighbour would leave the grid were never scattered (exact zeros), so
// clamping the flat neighbour index keeps the loads legal without a branch.
// DOT: also accumulates dots[c][blockIdx.x % PCG_DOT_SLOTS] += sum_i V[c][i] * out[c][i] (CG's p.Hp, slotted:
// see PcgScal).
template <typename real, int KC, bool DOT>
__global__ __launch_bounds__(256) void k_stencil_spmv(GridDev<real> G, const real* __restrict__ A_st, const real* __restrict__ V,
                                                      int k, const real* __restrict__ add, real beta, real* __restrict__ out,
                                                      double* __restrict__ dots) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* s_off = reinterpret_cast<int*>(smem);
  const int R = G.R, m = G.m, d = G.d;
  for (int o = threadIdx.x; o < R; o += blockDim.x) {
    int rem = o, f = 0;
    for (int q = d - 1; q >= 0; --q) {
      int c = rem % 7;
      rem /= 7;
      f += (c - 3) * G.stride[q];
    }
    s_off[o] = f;
  }
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int c0 = blockIdx.y * KC;
  real acc[KC];
#pragma unroll
  for (int c = 0; c < KC; ++c) acc[c] = (real)0;
  if (i < m) {
    const real* __restrict__ a_ptr = A_st + i;
#pragma unroll 8
    for (int o = 0; o < R; ++o) {
      const real a = a_ptr[(int64_t)o * m];
      int j = i + s_off[o];
      j = j < 0 ? 0 : (j >= m ? m - 1 : j);
#pragma unroll
      for (int c = 0; c < KC; ++c)
        if (c0 + c < k) acc[c] += a * V[(int64_t)(c0 + c) * m + j];
    }
  }
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    double part = 0;
    if (i < m && c0 + c < k) {
      const int64_t e = (int64_t)(c0 + c) * m + i;
      real r = acc[c];
      if (add) r += beta * add[e];
      out[e] = r;
      if (DOT) part = (double)V[e] * (double)r;
    }
    if (DOT) {
      double* s_red = reinterpret_cast<double*>(smem);
      __syncthreads();
      double tot = block_reduce_sum(part, s_red);
      if (threadIdx.x == 0 && c0 + c < k) pcg_dot_add(dots, c0 + c, tot);
    }
  }
}

template <typename real>
static int launch_spmv(const GridDev<real>& G, const real* A_st, const real* V, int k, const real* add, real beta, real* out, double* dots,
                       hipStream_t s) {
  const int kc = k >= 4 ? 4 : (k >= 2 ? 2 : 1);
  dim3 grd((unsigned)((G.m + 255) / 256), (unsigned)((k + kc - 1) / kc));
  size_t sh = (size_t)G.R * sizeof(int);
  if (sh < 16 * sizeof(double)) sh = 16 * sizeof(double);
  const bool prof = g_prof.on && g_prof.used + 2 <= g_prof.ev.size();
  if (prof) (void)hipEventRecord(g_prof.ev[g_prof.used++], s);
#define SPMV(KC)                                                                                                            \
  do {                                                                                                                      \
    if (dots) hipLaunchKernelGGL((k_stencil_spmv<real, KC, true>), grd, dim3(256), sh, s, G, A_st, V, k, add, beta, out, dots); \
    else hipLaunchKernelGGL((k_stencil_spmv<real, KC, false>), grd, dim3(256), sh, s, G, A_st, V, k, add, beta, out, dots);   \
  } while (0)
  if (kc == 4) SPMV(4);
  else if (kc == 2) SPMV(2);
  else SPMV(1);
#undef SPMV
  if (prof) (void)hipEventRecord(g_prof.ev[g_prof.used++], s);
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}

// ------------------------------------------------- stencil SpMV, wide form ---
// Fast path for m % 4 == 0.  Each thread owns 4 consecutive rows and streams
// A_st with 16-byte loads; the grid is additionally split over the leading
// stencil digit (NCH = 7 chunks for d >= 2) so that >= 13 waves per CU keep
// ~90 KB of HBM requests in flight.  For one "mid" offset combination the 7
// innermost offsets of 4 rows read a shared 10-wide window of v (10 loads for
// 28 FMAs).  Each chunk writes its own partial vector (plain coalesced 16-byte
// stores, no atomics); the consumer sums the NCH partials (k_spmv_reduce, or
// fused into k_pcg_update_x / k_pcg_init).
//   part[ch][c][i] = sum_{o in chunk ch} A_st[o][i] * V[c][clamp(i+off(o))]
//   DOT: dots[c] += sum_i V[c][i] * part[ch][c][i]  (+ beta * V.add for ch == 0)
template <typename real>
struct Vec4 { real x, y, z, w; };

template <typename real>
__device__ __forceinline__ Vec4<real> load4(const real* __restrict__ p) {
  Vec4<real> r;
  if constexpr (sizeof(real) == 4) {
    float4 v = *reinterpret_cast<const float4*>(p);
    r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w;
  } else {
    double2 a = *reinterpret_cast<const double2*>(p);
    double2 b = *reinterpret_cast<const double2*>(p + 2);
    r.x = a.x; r.y = a.y; r.z = b.x; r.w = b.y;
  }
  return r;
}

// Same, with the non-temporal cache policy: A_st is read exactly once per launch, so
// it should not displace v / the partial vectors from L2.
template <typename real>
__device__ __forceinline__ Vec4<real> load4_nt(const real* __restrict__ p) {
  Vec4<real> r;
  if constexpr (sizeof(real) == 4) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p));
    r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w;
  } else {
    typedef double d2 __attribute__((ext_vector_type(2)));
    d2 a = __builtin_nontemporal_load(reinterpret_cast<const d2*>(p));
    d2 b = __builtin_nontemporal_load(reinterpret_cast<const d2*>(p + 2));
    r.x = a.x; r.y = a.y; r.z = b.x; r.w = b.y;
  }
  return r;
}

template <typename real>
__device__ __forceinline__ void store4(real* __restrict__ p, real a, real b, real c, real d) {
  if constexpr (sizeof(real) == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
  } else {
    *reinterpret_cast<double2*>(p) = make_double2(a, b);
    *reinterpret_cast<double2*>(p + 2) = make_double2(c, d);
  }
}

static inline int spmv_nch(int d) { return d >= 2 ? 7 : 1; }

template <typename real, int KC, bool DOT>
__global__ __launch_bounds__(256) void k_stencil_spmv4(GridDev<real> G, const real* __restrict__ A_st, const real* __restrict__ V, int k,
                                                       int nmid, real* __restrict__ part, const real* __restrict__ add, real beta,
                                                       double* __restrict__ dots) {
  __shared__ int s_off[352];
  __shared__ double s_red[16];
  const int m = G.m, d = G.d;
  const int ch = blockIdx.y;
  const int c0 = blockIdx.z * KC;
  for (int mid = threadIdx.x; mid < nmid; mid += blockDim.x) {
    int rem = mid, f = d >= 2 ? (ch - 3) * G.stride[0] : 0;
    for (int q = d - 2; q >= 1; --q) {
      const int c = rem % 7;
      rem /= 7;
      f += (c - 3) * G.stride[q];
    }
    s_off[mid] = f;
  }
  __syncthreads();
  const int i4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const bool live = i4 < m;
  real acc[KC][4];
#pragma unroll
  for (int c = 0; c < KC; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[c][r] = (real)0;
  if (live) {
    const real* __restrict__ a_base = A_st + (int64_t)ch * nmid * 7 * m + i4;
#ifdef WISKI_SPMV_UNROLL
#pragma unroll WISKI_SPMV_UNROLL
#endif
    for (int mid = 0; mid < nmid; ++mid) {
      const int base = i4 + s_off[mid] - 3;
      real win[KC][10];
#pragma unroll
      for (int e = 0; e < 10; ++e) {
        int j = base + e;
        j = j < 0 ? 0 : (j >= m ? m - 1 : j);
#pragma unroll
        for (int c = 0; c < KC; ++c) win[c][e] = (c0 + c < k) ? V[(int64_t)(c0 + c) * m + j] : (real)0;
      }
      const real* __restrict__ a_mid = a_base + (int64_t)mid * 7 * m;
#pragma unroll
      for (int c7 = 0; c7 < 7; ++c7) {
#ifdef WISKI_SPMV_NT
        const Vec4<real> a = load4_nt<real>(a_mid + (int64_t)c7 * m);
#else
        const Vec4<real> a = load4<real>(a_mid + (int64_t)c7 * m);
#endif
#pragma unroll
        for (int c = 0; c < KC; ++c) {
          acc[c][0] += a.x * win[c][c7 + 0];
          acc[c][1] += a.y * win[c][c7 + 1];
          acc[c][2] += a.z * win[c][c7 + 2];
          acc[c][3] += a.w * win[c][c7 + 3];
        }
      }
    }
  }
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    double pd = 0;
    if (live && c0 + c < k) {
      const int64_t e = (int64_t)(c0 + c) * m + i4;
      store4<real>(part + ((int64_t)ch * k) * m + e, acc[c][0], acc[c][1], acc[c][2], acc[c][3]);
      if (DOT) {
        const Vec4<real> v = load4<real>(V + e);
        pd = (double)v.x * acc[c][0] + (double)v.y * acc[c][1] + (double)v.z * acc[c][2] + (double)v.w * acc[c][3];
        if (ch == 0 && add) {
          const Vec4<real> ad = load4<real>(add + e);
          pd += (double)beta * ((double)v.x * ad.x + (double)v.y * ad.y + (double)v.z * ad.z + (double)v.w * ad.w);
        }
      }
    }
    if (DOT) {
      const double tot = block_reduce_sum(pd, s_red);
      if (threadIdx.x == 0 && c0 + c < k) pcg_dot_add(dots, c0 + c, tot);
    }
  }
}

// out[c][i] = beta * add[c][i] + sum_ch part[ch][c][i]
// zl: part[nch-1] is the atomically accumulated partial of the symmetric SpMV; re-zero it once consumed.
template <typename real>
__global__ __launch_bounds__(256) void k_spmv_reduce(int64_t km, int nch, real* __restrict__ part, const real* __restrict__ add, real beta,
                                                     real* __restrict__ out, int zl) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < km; e += (int64_t)gridDim.x * blockDim.x) {
    real r = add ? beta * add[e] : (real)0;
    for (int ch = 0; ch < nch; ++ch) r += part[(int64_t)ch * km + e];
    if (zl) part[(int64_t)(nch - 1) * km + e] = (real)0;
    out[e] = r;
  }
}

// ------------------------------------------- stencil SpMM (many right-hand sides) ---
// Same product for KC columns at once (predictive variances, MLL probes: k = 8..64 columns).
// The wide SpMV re-reads its v windows from L2 per thread and per "mid" offset; with many
// columns that traffic (and the A re-read per column group) dominates.  Here a block stages, ONCE per
// (row block, chunk, column group), the union of all v windows of the chunk -- rows
// [i0 + off(c0) - Wd, i0 + 1024 + off(c0) + Wd), Wd = 3 * sum_{q>=1} stride_q -- for KC columns
// in LDS, in a [column][j & 3][j >> 2] layout so that the threads' 4-row groups read it
// conflict-free at any shift.  A is then streamed once per column group with 16-byte loads
// (KC = 16: 4 passes over A for 64 columns instead of 8, v traffic cut ~5x).
template <typename real, int KC, bool DOT>
__global__ __launch_bounds__(256) void k_stencil_spmm(GridDev<real> G, const real* __restrict__ A_st, const real* __restrict__ V, int k, int nmid,
                                                      int Wd, int W4, real* __restrict__ part, const real* __restrict__ add, real beta,
                                                      double* __restrict__ dots) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  real* win = reinterpret_cast<real*>(smem);   // [KC][4][W4]
  __shared__ int s_off[64];
  __shared__ double s_red[16];
  const int m = G.m, d = G.d;
  const int ch = blockIdx.y;
  const int c0 = blockIdx.z * KC;
  const int off_c0 = (ch - 3) * G.stride[0];
  for (int mid = threadIdx.x; mid < nmid; mid += blockDim.x) {
    int rem = mid, f = 0;
    for (int q = d - 2; q >= 1; --q) {
      const int c = rem % 7;
      rem /= 7;
      f += (c - 3) * G.stride[q];
    }
    s_off[mid] = f;
  }
  const int i0 = blockIdx.x * 1024;
  const int ws = i0 + off_c0 - Wd;          // flat index of window element 0
  const int W = 4 * W4;
  for (int idx = threadIdx.x; idx < W; idx += blockDim.x) {
    int j = ws + idx;
    j = j < 0 ? 0 : (j >= m ? m - 1 : j);   // clamped reads only ever meet exact-zero A entries
    real* dstp = win + (idx & 3) * W4 + (idx >> 2);
#pragma unroll
    for (int c = 0; c < KC; ++c) dstp[c * W] = (c0 + c < k) ? V[(int64_t)(c0 + c) * m + j] : (real)0;
  }
  __syncthreads();
  const int t = threadIdx.x;
  const int i4 = i0 + 4 * t;
  const bool live = i4 < m;
  real acc[KC][4];
#pragma unroll
  for (int c = 0; c < KC; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[c][r] = (real)0;
  if (live) {
    const real* __restrict__ a_base = A_st + (int64_t)ch * nmid * 7 * m + i4;
    for (int mid = 0; mid < nmid; ++mid) {
      const int sh = Wd + s_off[mid] - 3;   // window index of this thread's first needed value is 4t + sh
      const real* __restrict__ a_mid = a_base + (int64_t)mid * 7 * m;
      Vec4<real> a[7];
#pragma unroll
      for (int c7 = 0; c7 < 7; ++c7) a[c7] = load4<real>(a_mid + (int64_t)c7 * m);
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        real wv[10];
#pragma unroll
        for (int e = 0; e < 10; ++e) {
          const int j = 4 * t + sh + e;
          wv[e] = win[c * W + (j & 3) * W4 + (j >> 2)];
        }
#pragma unroll
        for (int c7 = 0; c7 < 7; ++c7) {
          acc[c][0] += a[c7].x * wv[c7 + 0];
          acc[c][1] += a[c7].y * wv[c7 + 1];
          acc[c][2] += a[c7].z * wv[c7 + 2];
          acc[c][3] += a[c7].w * wv[c7 + 3];
        }
      }
    }
  }
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    double pd = 0;
    if (live && c0 + c < k) {
      const int64_t e = (int64_t)(c0 + c) * m + i4;
      store4<real>(part + ((int64_t)ch * k) * m + e, acc[c][0], acc[c][1], acc[c][2], acc[c][3]);
      if (DOT) {
        const Vec4<real> v = load4<real>(V + e);
        pd = (double)v.x * acc[c][0] + (double)v.y * acc[c][1] + (double)v.z * acc[c][2] + (double)v.w * acc[c][3];
        if (ch == 0 && add) {
          const Vec4<real> ad = load4<real>(add + e);
          pd += (double)beta * ((double)v.x * ad.x + (double)v.y * ad.y + (double)v.z * ad.z + (double)v.w * ad.w);
        }
      }
    }
    if (DOT) {
      const double tot = block_reduce_sum(pd, s_red);
      if (threadIdx.x == 0 && c0 + c < k) pcg_dot_add(dots, c0 + c, tot);
    }
  }
}

template <typename real, int KC>
static int launch_spmm_kc(const GridDev<real>& G, const real* A_st, const real* V, int k, int nmid, real* part, const real* add, real beta,
                          double* dots, hipStream_t s) {
  int Wd = 3;
  for (int q = 1; q < G.d - 1; ++q) Wd += 3 * G.stride[q];
  const int W4 = ((1024 + 2 * Wd + 3) / 4 + 16) | 1;   // odd word stride between the four residue planes
  const size_t sh = (size_t)KC * 4 * W4 * sizeof(real);
  static size_t lds_set[2] = {0, 0};
  dim3 grd((unsigned)((G.m + 1023) / 1024), (unsigned)spmv_nch(G.d), (unsigned)((k + KC - 1) / KC));
  if (dots) {
    if (sh > 48 * 1024 && sh > lds_set[0]) {
      if (hipFuncSetAttribute((const void*)k_stencil_spmm<real, KC, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh) != hipSuccess)
        return WISKI_E_LAUNCH;
      lds_set[0] = sh;
    }
    hipLaunchKernelGGL((k_stencil_spmm<real, KC, true>), grd, dim3(256), sh, s, G, A_st, V, k, nmid, Wd, W4, part, add, beta, dots);
  } else {
    if (sh > 48 * 1024 && sh > lds_set[1]) {
      if (hipFuncSetAttribute((const void*)k_stencil_spmm<real, KC, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh) != hipSuccess)
        return WISKI_E_LAUNCH;
      lds_set[1] = sh;
    }
    hipLaunchKernelGGL((k_stencil_spmm<real, KC, false>), grd, dim3(256), sh, s, G, A_st, V, k, nmid, Wd, W4, part, add, beta, dots);
  }
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}

// partial SpMV launcher (requires m % 4 == 0); part holds spmv_nch(d)*k*m reals.
template <typename real>
static int launch_spmv4(const GridDev<real>& G, const real* A_st, const real* V, int k, real* part, const real* add, real beta, double* dots,
                        hipStream_t s) {
  const int nch = spmv_nch(G.d);
  int nmid = 1;
  for (int q = 1; q < G.d - 1; ++q) nmid *= 7;
  const bool prof = g_prof.on && g_prof.used + 2 <= g_prof.ev.size();
  if (k >= 8 && (G.d == 2 || G.d == 3) && nmid <= 64 && !prof) {
    // many right-hand sides: LDS-staged SpMM (fp32: 16 columns per pass, fp64: 8)
    if constexpr (sizeof(real) == 4) {
      if (k >= 16) return launch_spmm_kc<real, 16>(G, A_st, V, k, nmid, part, add, beta, dots, s);
    }
    return launch_spmm_kc<real, 8>(G, A_st, V, k, nmid, part, add, beta, dots, s);
  }
  const int kc = k >= 8 ? 8 : (k >= 4 ? 4 : (k >= 2 ? 2 : 1));
  dim3 grd((unsigned)((G.m / 4 + 255) / 256), (unsigned)nch, (unsigned)((k + kc - 1) / kc));
  if (prof) (void)hipEventRecord(g_prof.ev[g_prof.used++], s);
#define SPMV4(KC)                                                                                                              \
  do {                                                                                                                         \
    if (dots) hipLaunchKernelGGL((k_stencil_spmv4<real, KC, true>), grd, dim3(256), 0, s, G, A_st, V, k, nmid, part, add, beta, dots); \
    else hipLaunchKernelGGL((k_stencil_spmv4<real, KC, false>), grd, dim3(256), 0, s, G, A_st, V, k, nmid, part, add, beta, dots);   \
  } while (0)
  if (kc == 8) SPMV4(8);
  else if (kc == 4) SPMV4(4);
  else if (kc == 2) SPMV4(2);
  else SPMV4(1);
#undef SPMV4
  if (prof) (void)hipEventRecord(g_prof.ev[g_prof.used++], s);
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}

// ------------------------------------- symmetric half-stencil SpMV (native WtW storage) ---
// A = W^T D^-1 W is symmetric, so only the offsets o >= centre are stored, a(oh, i) = A[i, i + off(c + oh)],
// oh = 0 .. (R-1)/2, in the row-interleaved layout k_scatter_stats_sym accumulates (scatter_stats.hip):
//   oh < 4            : A_h[4 i + oh]                          (group 0: innermost digits 3..6 of the centre prefix)
//   oh = 7 g - 3 + s  : A_h[(7 g - 3) m + 7 i + s]             (group g >= 1, s = 0..6)
// and every stored entry is used twice,
//   out[i]          += a(oh, i) * v[i + off]        ("direct")
//   out[i + off]    += a(oh, i) * v[i]              ("transposed", oh > 0)
// which halves the HBM bytes of the product (and of the model state).
//
// Generic form (any m, any d): one thread per output row gathers both terms; the transposed one re-reads
// A_h at row i - off (L2 hits for the small offsets).  Used for m % 4 != 0.
template <typename real, int KC, bool DOT>
__global__ __launch_bounds__(256) void k_stencil_spmv_sym(GridDev<real> G, const real* __restrict__ A_h, const real* __restrict__ V, int k,
                                                          const real* __restrict__ add, real beta, real* __restrict__ out,
                                                          double* __restrict__ dots) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int R = G.R, m = G.m, d = G.d;
  const int H = (R + 1) / 2, ctr = (R - 1) / 2;
  int64_t* s_base = reinterpret_cast<int64_t*>(smem);        // a(oh, i) = A_h[s_base[oh] + i * s_str[oh]]
  int* s_off = reinterpret_cast<int*>(s_base + H);
  int* s_str = s_off + H;
  for (int oh = threadIdx.x; oh < H; oh += blockDim.x) {
    int rem = ctr + oh, f = 0;
    for (int q = d - 1; q >= 0; --q) {
      int c = rem % 7;
      rem /= 7;
      f += (c - 3) * G.stride[q];
    }
    s_off[oh] = f;
    const int g = oh < 4 ? 0 : (oh - 4) / 7 + 1;
    s_base[oh] = oh < 4 ? (int64_t)oh : (int64_t)(7 * g - 3) * m + (oh - 4) % 7;
    s_str[oh] = oh < 4 ? 4 : 7;
  }
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int c0 = blockIdx.y * KC;
  real acc[KC];
#pragma unroll
  for (int c = 0; c < KC; ++c) acc[c] = (real)0;
  if (i < m) {
#pragma unroll 4
    for (int oh = 0; oh < H; ++oh) {
      const int off = s_off[oh];
      const real* __restrict__ a_row = A_h + s_base[oh];
      const int64_t st = s_str[oh];
      const real a = a_row[i * st];
      int j = i + off;
      j = j < 0 ? 0 : (j >= m ? m - 1 : j);
      const int jt = i - off;
      const bool tr = oh > 0 && jt >= 0 && jt < m;
      const real at = tr ? a_row[jt * st] : (real)0;
      const int jtc = tr ? jt : i;
#pragma unroll
      for (int c = 0; c < KC; ++c)
        if (c0 + c < k) acc[c] += a * V[(int64_t)(c0 + c) * m + j] + at * V[(int64_t)(c0 + c) * m + jtc];
    }
  }
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    double part = 0;
    if (i < m && c0 + c < k) {
      const int64_t e = (int64_t)(c0 + c) * m + i;
      real r = acc[c];
      if (add) r += beta * add[e];
      out[e] = r;
      if (DOT) part = (double)V[e] * (double)r;
    }
    if (DOT) {
      __shared__ double s_red[16];
      double tot = block_reduce_sum(part, s_red);
      if (threadIdx.x == 0 && c0 + c < k) pcg_dot_add(dots, c0 + c, tot);
    }
  }
}

template <typename real>
static int launch_spmv_sym(const GridDev<real>& G, const real* A_h, const real* V, int k, const real* add, real beta, real* out, double* dots,
                           hipStream_t s) {
  const int kc = k >= 4 ? 4 : (k >= 2 ? 2 : 1);
  dim3 grd((unsigned)((G.m + 255) / 256), (unsigned)((k + kc - 1) / kc));
  const size_t sh = (size_t)((G.R + 1) / 2) * (sizeof(int64_t) + 2 * sizeof(int));
#define SPMV(KC)                                                                                                                 \
  do {                                                                                                                           \
    if (dots) hipLaunchKernelGGL((k_stencil_spmv_sym<real, KC, true>), grd, dim3(256), sh, s, G, A_h, V, k, add, beta, out, dots); \
    else hipLaunchKernelGGL((k_stencil_spmv_sym<real, KC, false>), grd, dim3(256), sh, s, G, A_h, V, k, add, beta, out, dots);   \
  } while (0)
  if (kc == 4) SPMV(4);
  else if (kc == 2) SPMV(2);
  else SPMV(1);
#undef SPMV
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}

// Wide form (m % 4 == 0).  As k_stencil_spmv4, a thread owns 4 consecutive rows and streams A_h with
// 16-byte loads, one "group" (= one value of the leading d-1 stencil digits, 7 innermost offsets; the
// centre group holds only its 4 non-negative ones) at a time.  The direct term accumulates in registers
// and is written as one partial vector per chunk of groups (blockIdx.y).  The transposed term of a group
// lands on the 10-wide row window [i4 + f - 3, i4 + f + 7): it is accumulated in registers across the 7
// offsets and then added to a *wave-private* LDS window ([j & 3][j >> 2] planes, so the lanes' 4-row
// groups hit distinct banks at any shift) with plain read-modify-writes in three lane-disjoint phases
// (window elements 0..3 / 4..7 / 8..9 of lane t are the quads of lanes t / t+1 / t+2: within a phase no
// two lanes touch the same cell, and a wave's LDS operations execute in order) -- LDS float atomics
// (ds_add_f32) measured ~0.4 lanes/clk/CU here and would triple the kernel time.  The window is flushed
// with *coalesced* fire-and-forget global atomics (measured ~300 G lane-atomics/s when the 64 lanes cover
// 256 contiguous bytes, vs ~40 G/s for scattered ones) into the extra partial vector part[nch], which
// must be zero on entry and is re-zeroed by whoever consumes the partials.  The window follows the
// groups: it is flushed whenever the next group's offset leaves [wb, wb + span].  No block barrier is
// needed in the main loop.
//   DOT: dots[c] += v . (A v) = sum_i v_i (2 direct_i - diag_i v_i)   (+ beta v.add once)
static inline int sym_groups(int d) {
  int np = 1;
  for (int q = 0; q < d - 1; ++q) np *= 7;
  return (np + 1) / 2;
}
// XCD-contiguous row ranges for the half-stencil SpMV kernels (spmv_sym_dma.h).  Taken where A_h is served by the 256 MB Infinity
// Cache: 50^3 fp32 (LDS-DMA kernel, 86 MB) 18.1 -> 17.7 us back to back, 21.3 -> 20.5 us per dispatch inside bench.py; 50^3 fp64
// (LDS-window kernel, 172 MB) 36.8 -> 36.2 us.  NOT where A_h streams from HBM: at 30^4 (7.8 GB fp64 / 3.9 GB fp32) eight separate
// address streams are slower than one interleaved stream (1342 -> 1409 us, 738 -> 754 us: DRAM page locality), so the plain
// blockIdx.x = row block mapping stays there.  WISKI_SYM_XCD=0 / 1 forces it off / on.
static int g_sym_xcd = -2;
static inline bool sym_xcd_map(int64_t a_bytes) {
  if (g_sym_xcd == -2) {
    const char* e = getenv("WISKI_SYM_XCD");
    g_sym_xcd = e ? atoi(e) : -1;
  }
  if (g_sym_xcd >= 0) return g_sym_xcd != 0;
  return a_bytes <= (int64_t)192 << 20;
}
static inline int sym_block() { return 128; }    // threads per block of the wide symmetric SpMV (64 / 256 measured: no better)
static inline int sym_nch_lds(int d) {           // chunks of the LDS-window kernel: 7 (or every group of a small stencil)
  const int ng = sym_groups(d);
  return ng < 7 ? ng : 7;
}

__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

#ifdef WISKI_SYM_TIMING   // phase stamps of one mid-grid wave (tools/spmv_probe.py --timing)
__device__ long long g_sym_dbg[32];
#define SYM_STAMP(i) do { if (lane == 0 && wave == 0 && blockIdx.x == gridDim.x / 2 && blockIdx.y == gridDim.y / 2 && blockIdx.z == 0) g_sym_dbg[i] = wall_clock64(); } while (0)
extern "C" int wiski_sym_dbg(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sym_dbg), sizeof(long long) * 32) == hipSuccess ? 0 : -1; }
#else
#define SYM_STAMP(i) do {} while (0)
#endif
#ifndef WISKI_SYM_ABLATE
#define WISKI_SYM_ABLATE 0   // timing ablations (tools/spmv_probe.py): 1 no LDS accumulation, 3 no global flush atomics
#endif
template <typename real, int KC, bool DOT>
__global__ __launch_bounds__(256) void k_stencil_spmv4_sym(GridDev<real> G, const real* __restrict__ A_h, const real* __restrict__ V, int k,
                                                           int ng, int nch, int span, int W4, real* __restrict__ part,
                                                           const real* __restrict__ add, real beta, double* __restrict__ dots, int g_lo, int g_hi,
                                                           int xcd_rb) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int s_off[176];
  __shared__ double s_red[16];
  const int m = G.m, d = G.d;
  // xcd_rb != 0: XCD-contiguous row blocks (see spmv_sym_dma.h): grid.x = 8 * xcd_rb, workgroup b (on XCD b % 8) takes row block
  // (b % 8) * xcd_rb + b / 8, so the v windows an XCD's L2 has to hold are those of ONE contiguous range of rows
  const int rbx = xcd_rb ? (int)(blockIdx.x & 7) * xcd_rb + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  if (xcd_rb && (int64_t)rbx * (blockDim.x >> 6) * 256 >= m) return;      // block-uniform, before the first barrier
  const int ch = blockIdx.y;
  const int c0 = blockIdx.z * KC;
  // the groups of this launch, [g_lo, g_hi) -- all ng of them, or the share of a stencil-sharded replica (wiski_shard) --
  // dealt to the nch chunks (a chunk may be empty: it then writes a zero partial)
  const int gA = g_lo + (int)((int64_t)ch * (g_hi - g_lo) / nch), gB = g_lo + (int)((int64_t)(ch + 1) * (g_hi - g_lo) / nch);
  const int cP = ng - 1;                        // prefix code of the centre: (7^(d-1) - 1) / 2
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  constexpr int STAGE = 7 * 256;                // reals of one group tile of a wave
  real* __restrict__ stage = reinterpret_cast<real*>(smem) + (size_t)wave * (STAGE + KC * 4 * W4);   // this wave's A tile
  real* __restrict__ tw = stage + STAGE;                                                           // and its [KC][4][W4] window
  for (int g = gA + t; g < gB; g += blockDim.x) {
    int rem = cP + g, f = 0;
    for (int q = d - 2; q >= 0; --q) {
      f += (rem % 7 - 3) * G.stride[q];
      rem /= 7;
    }
    s_off[g - gA] = f;
  }
  SYM_STAMP(0);
  for (int e = lane; e < KC * 4 * W4; e += 64) tw[e] = (real)0;
  __syncthreads();
  SYM_STAMP(1);
  const int iw0 = (rbx * (blockDim.x >> 6) + wave) * 256;   // first row of this wave
  const int i4 = iw0 + 4 * lane;
  const bool live = i4 < m;
  const int64_t km = (int64_t)k * m;
  real* __restrict__ tacc_out = part + (int64_t)nch * km;   // the atomically accumulated partial
  const int wlen = 256 + span + 10;

  auto flush = [&](int wb) {
    for (int idx = lane; idx < wlen; idx += 64) {
      const int j = iw0 + wb - 3 + idx;
      real* cell = tw + (idx & 3) * W4 + (idx >> 2);
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        const real v = cell[c * 4 * W4];
        if (v != (real)0) {
          cell[c * 4 * W4] = (real)0;
#if WISKI_SYM_ABLATE != 3
          if (j >= 0 && j < m && c0 + c < k) atomic_add_real(tacc_out + (int64_t)(c0 + c) * m + j, v);
#endif
        }
      }
    }
    wave_lds_fence();
  };

  real acc[KC][4], dg[KC][4], xo[KC][4];
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    Vec4<real> x4;
    x4.x = x4.y = x4.z = x4.w = (real)0;
    if (live && c0 + c < k) x4 = load4<real>(V + (int64_t)(c0 + c) * m + i4);
    xo[c][0] = x4.x; xo[c][1] = x4.y; xo[c][2] = x4.z; xo[c][3] = x4.w;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[c][r] = dg[c][r] = (real)0;
  }
  // one group's operands: 4 rows x 7 innermost offsets of A_h and the 10-wide window of v.  A_h keeps a
  // row's 7 offsets adjacent (the scatter's layout), so the wave reads its 256 rows x 7 reals as one
  // contiguous span with fully coalesced 16-byte loads, parks it in a wave-private LDS tile and reads it
  // back row-wise: lane t gets the 28 reals of rows 4t..4t+3 (7 x ds_read_b128, stride 112 B: bank-
  // conflict free).  (Issuing the next group's loads before consuming the current one was measured twice -- on the
  // offset-major predecessor and on this kernel, v window first so that vmcnt can retire it alone: 158 VGPRs,
  // 23.3 us at best (5 chunks) against 23.9 us without; not kept.  Mind the register budget: 120 VGPRs keep 4 waves
  // per SIMD = 16 per CU for the 13.3 this grid wants; a refactor that cost 130 made the kernel 28% slower.)
  struct GroupData {
    Vec4<real> a[7];       // a[s].{x,y,z,w} = rows i4..i4+3 at innermost offset digit s
    real win[KC][10];
  };
  const int nrows = m - iw0 < 256 ? m - iw0 : 256;      // rows of this wave inside the grid (multiple of 4, may be <= 0)
  auto fetch = [&](int g, GroupData& D) {
    const int f = s_off[g - gA];
    if (g == 0) {
      // centre group: 4 reals per row (digits 3..6), rows i4..i4+3 are 16 contiguous reals
      Vec4<real> q[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (live) q[r] = load4<real>(A_h + (int64_t)4 * i4 + 4 * r);
        else q[r].x = q[r].y = q[r].z = q[r].w = (real)0;
      }
#pragma unroll
      for (int l = 0; l < 3; ++l) D.a[l].x = D.a[l].y = D.a[l].z = D.a[l].w = (real)0;
      D.a[3].x = q[0].x; D.a[3].y = q[1].x; D.a[3].z = q[2].x; D.a[3].w = q[3].x;
      D.a[4].x = q[0].y; D.a[4].y = q[1].y; D.a[4].z = q[2].y; D.a[4].w = q[3].y;
      D.a[5].x = q[0].z; D.a[5].y = q[1].z; D.a[5].z = q[2].z; D.a[5].w = q[3].z;
      D.a[6].x = q[0].w; D.a[6].y = q[1].w; D.a[6].z = q[2].w; D.a[6].w = q[3].w;
    } else {
      const real* __restrict__ src = A_h + (int64_t)(7 * g - 3) * m + (int64_t)7 * iw0;
      Vec4<real> ld[7];
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        const int e = 4 * (64 * j + lane);
        if (e < 7 * nrows) ld[j] = load4<real>(src + e);
        else ld[j].x = ld[j].y = ld[j].z = ld[j].w = (real)0;
      }
#pragma unroll
      for (int j = 0; j < 7; ++j) store4<real>(stage + 4 * (64 * j + lane), ld[j].x, ld[j].y, ld[j].z, ld[j].w);
      wave_lds_fence();
      real v[28];
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        const Vec4<real> t4 = load4<real>(stage + 28 * lane + 4 * j);
        v[4 * j + 0] = t4.x; v[4 * j + 1] = t4.y; v[4 * j + 2] = t4.z; v[4 * j + 3] = t4.w;
      }
      wave_lds_fence();
#pragma unroll
      for (int l = 0; l < 7; ++l) { D.a[l].x = v[l]; D.a[l].y = v[7 + l]; D.a[l].z = v[14 + l]; D.a[l].w = v[21 + l]; }
    }
    const int base = i4 + f - 3;
#pragma unroll
    for (int e = 0; e < 10; ++e) {
      int j = base + e;
      j = j < 0 ? 0 : (j >= m ? m - 1 : j);
#pragma unroll
      for (int c = 0; c < KC; ++c) D.win[c][e] = (c0 + c < k) ? V[(int64_t)(c0 + c) * m + j] : (real)0;
    }
  };
  GroupData cur;
  int wb = gB > gA ? s_off[0] : 0;
  for (int g = gA; g < gB; ++g) {
    const int f = s_off[g - gA];
    if (f < wb || f - wb > span) {          // uniform: the window moves on
      flush(wb);
      wb = f;
    }
    if (nrows > 0) fetch(g, cur);         // wave-cooperative (all lanes), rows past the grid read as zeros
    SYM_STAMP(2 + 3 * (g - gA));
    if (live) {
      real tr[KC][10];
#pragma unroll
      for (int c = 0; c < KC; ++c)
#pragma unroll
        for (int e = 0; e < 10; ++e) tr[c][e] = (real)0;
#pragma unroll
      for (int l = 0; l < 7; ++l) {
        const Vec4<real> a = cur.a[l];       // exact zeros for the digits the centre group does not store
#pragma unroll
        for (int c = 0; c < KC; ++c) {
          acc[c][0] += a.x * cur.win[c][l + 0];
          acc[c][1] += a.y * cur.win[c][l + 1];
          acc[c][2] += a.z * cur.win[c][l + 2];
          acc[c][3] += a.w * cur.win[c][l + 3];
          if (g == 0 && l == 3) {           // the diagonal: counted once
            dg[c][0] = a.x * xo[c][0]; dg[c][1] = a.y * xo[c][1]; dg[c][2] = a.z * xo[c][2]; dg[c][3] = a.w * xo[c][3];
          } else {
            tr[c][l + 0] += a.x * xo[c][0];
            tr[c][l + 1] += a.y * xo[c][1];
            tr[c][l + 2] += a.z * xo[c][2];
            tr[c][l + 3] += a.w * xo[c][3];
          }
        }
      }
      SYM_STAMP(3 + 3 * (g - gA));
#if WISKI_SYM_ABLATE != 1
      const int w0 = 4 * lane + (f - wb);
#pragma unroll
      for (int ph = 0; ph < 3; ++ph) {      // lane-disjoint phases, see the header comment
        // the cells of a phase are distinct: read them all, then write them all (written as `cell += x` the
        // compiler serialises one LDS round trip per element)
        constexpr int NE[3] = {4, 4, 2};
        real* cell[4];
        real old[KC][4];
#pragma unroll
        for (int u = 0; u < NE[ph]; ++u) {
          const int idx = w0 + 4 * ph + u;
          cell[u] = tw + (idx & 3) * W4 + (idx >> 2);
#pragma unroll
          for (int c = 0; c < KC; ++c) old[c][u] = cell[u][c * 4 * W4];
        }
#pragma unroll
        for (int u = 0; u < NE[ph]; ++u)
#pragma unroll
          for (int c = 0; c < KC; ++c) cell[u][c * 4 * W4] = old[c][u] + tr[c][4 * ph + u];
        wave_lds_fence();
      }
#endif
      SYM_STAMP(4 + 3 * (g - gA));
    }
  }
  SYM_STAMP(20);
  flush(wb);
  SYM_STAMP(21);
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    double pd = 0;
    if (live && c0 + c < k) {
      const int64_t e = (int64_t)(c0 + c) * m + i4;
      store4<real>(part + (int64_t)ch * km + e, acc[c][0], acc[c][1], acc[c][2], acc[c][3]);
      if (DOT) {
#pragma unroll
        for (int r = 0; r < 4; ++r) pd += (double)xo[c][r] * (2.0 * (double)acc[c][r] - (double)dg[c][r]);
        if (gA == 0 && gB > 0 && add) {        // the chunk that holds the centre group (sharded: on exactly one rank)
          const Vec4<real> ad = load4<real>(add + e);
          pd += (double)beta * ((double)xo[c][0] * ad.x + (double)xo[c][1] * ad.y + (double)xo[c][2] * ad.z + (double)xo[c][3] * ad.w);
        }
      }
    }
    if (DOT) {
      const double tot = block_reduce_sum(pd, s_red);
      if (threadIdx.x == 0 && c0 + c < k) pcg_dot_add(dots, c0 + c, tot);
    }
  }
  SYM_STAMP(22);
}

#include "spmv_sym_dma.h"
#include "spmv_sym_dma_mc.h"
#include "spmm_sym_cols.h"
#include "spmm_sym_bcast.h"

// Which wide half-stencil kernel serves (G, k): the LDS-DMA pipelined one (d = 3, fp32, one right-hand side; 4 chunks)
// or the LDS-window one (anything else with m % 4 == 0; up to 7 chunks).  WISKI_SYM_DMA=0 forces the latter,
// Ring depth 2, the 4-part work split and the late start of the light chunk (12 x 0.43 us at 50^3, scaled with m) are what the
// sweeps of rounds 2-4 settled on (spmv_sym_dma.h: 3-deep ring 22.8 us, 5 / 6 / 7 parts 19.1 / 19.9 / 20.5, delay 0..28: 18.0 at 12).
static int g_sym_dma = -1;
constexpr int g_sym_dma_nst = 2, g_sym_dma_parts = 4, g_sym_dma_delay = 12;
template <typename real>
static inline bool sym_use_dma(const GridDev<real>& G, int k) {
  if constexpr (sizeof(real) != 4) return false;
  if (g_sym_dma < 0) {
    const char* e = getenv("WISKI_SYM_DMA");
    g_sym_dma = e ? atoi(e) : 1;
  }
  return g_sym_dma != 0 && G.d == 3 && k == 1 && (G.m % 4) == 0 && symdma_lds_bytes(G.g[2], g_sym_dma_nst) <= 64 * 1024;
}
// Many right-hand sides (k >= 32): the lanes-are-columns SpMM (spmm_sym_cols.h) reads A_h once per 64 columns and leaves
// ONE finished vector per column in part[0] (no atomically accumulated partial).
static inline bool sym_use_bcast();
static inline bool sym_use_cols(int k, size_t es) {
  // (the row-major images, 64 columns per slice, must fit behind part[0]: 64 m <= 7 k m, i.e. k >= 16)
  // Scalar-path kernel, measured at 50^3: 288 us at k = 16 (the 4-column kernel: 180), 240 us at k = 64 (670) -> from 32 columns.
  // The broadcast kernel costs the same for any k <= 64 (fp32 131..140 us, fp64 322..369 us) against 158 us (fp32, 4-column LDS-DMA
  // kernel) / 290 us (fp64, LDS-window kernel) at k = 16 and 229 / 429 us at k = 24: from 16 (fp32) / 24 (fp64) columns.
  return k >= (sym_use_bcast() ? (es == 4 ? 16 : 24) : 32);
}
// The DPP-broadcast form of the many-column product (spmm_sym_bcast.h), both precisions, any d.  WISKI_SPMM_BCAST=0 keeps the scalar-path kernel.
static int g_spmm_bcast = -1;
static inline bool sym_use_bcast() {
  if (g_spmm_bcast < 0) {
    const char* e = getenv("WISKI_SPMM_BCAST");
    g_spmm_bcast = e ? atoi(e) : 1;
  }
  return g_spmm_bcast != 0;
}
static inline int spmmc_kp(int k) { return (k + 15) / 16 * 16; }
// A few right-hand sides (2 <= k < 32; d = 3, fp32): the multi-column LDS-DMA kernel (spmv_sym_dma_mc.h), 4 (2) columns per pass
// over A_h with the built-in 4-part split (WISKI_SYM_DMA=0 also sends these to the LDS-window kernel).
template <typename real>
static inline bool sym_use_dma_mc(const GridDev<real>& G, int k) {
  if constexpr (sizeof(real) != 4) return false;
  if (g_sym_dma < 0) (void)sym_use_dma<real>(G, 1);
  return g_sym_dma != 0 && G.d == 3 && k >= 2 && !sym_use_cols(k, sizeof(real)) && (G.m % 4) == 0 && G.g[2] >= 4 &&
         symdma_mc_lds_bytes(G.g[2], k >= 4 ? 4 : 2) <= 64 * 1024;
}
// number of direct partial vectors the wide half-stencil SpMV writes for (G, k); one more is accumulated atomically
template <typename real>
static inline int sym_nch(const GridDev<real>& G, int k) {
  if (sym_use_dma<real>(G, k)) return g_sym_dma_parts;
  return sym_use_dma_mc<real>(G, k) ? 4 : sym_nch_lds(G.d);
}
// total number of partial vectors the consumers of launch_spmv4_sym have to sum, and whether the last one is the
// atomically accumulated one (to be re-zeroed once consumed)
template <typename real>
static inline int sym_partials(const GridDev<real>& G, int k, int* zl) {
  if (sym_use_cols(k, sizeof(real))) { *zl = 0; return 1; }
  *zl = 1;
  return sym_nch<real>(G, k) + 1;
}

// One launch with optional per-dispatch timing: in profiling mode the start/stop events are attached to the
// kernel's own dispatch packet (hipExtLaunchKernelGGL), so their difference is the kernel's execution time --
// a hipEventRecord bracket adds ~3 us of marker-packet latency to a 20 us kernel.
template <typename F, typename... Args>
static inline void launch_timed(F kern, dim3 grd, dim3 blk, size_t sh, hipStream_t s, Args... args) {
  if (g_prof.on && g_prof.used + 2 <= g_prof.ev.size()) {
    hipEvent_t e0 = g_prof.ev[g_prof.used], e1 = g_prof.ev[g_prof.used + 1];
    g_prof.used += 2;
    hipExtLaunchKernelGGL(kern, grd, blk, (unsigned)sh, s, e0, e1, 0, args...);
  } else {
    hipLaunchKernelGGL(kern, grd, blk, sh, s, args...);
  }
}

// partial SpMV launcher on the half stencil (requires m % 4 == 0).  part holds (sym_nch(G, k) + 1) * k * m reals;
// part[sym_nch(G, k)] must be zero on entry (see the kernel comments).
// Stencil-sharded replicas (wiski_shard): rank r of n owns the groups [r ng / n, (r + 1) ng / n) of the half stencil.
static inline void shard_group_range(int ng, int rank, int nranks, int* lo, int* hi) {
  *lo = (int)((int64_t)rank * ng / nranks);
  *hi = (int)((int64_t)(rank + 1) * ng / nranks);
}
// d = 3: the parts (runs of groups inside one leading digit d0) that cover [lo, hi); returns their number (<= 8)
static inline int shard_parts_d3(int lo, int hi, SymDmaParts* tab) {
  int n = 0;
  for (int g = lo; g < hi && n < 8;) {
    const int d0 = (g + 3) / 7, p1 = (g + 3) % 7;
    int run = 7 - p1;
    if (g + run > hi) run = hi - g;
    tab->d0[n] = (unsigned char)d0; tab->p1lo[n] = (unsigned char)p1; tab->ntile[n] = (unsigned char)run;
    ++n;
    g += run;
  }
  tab->n = n;
  return n;
}

template <typename real>
static int launch_spmv4_sym(const GridDev<real>& G, const real* A_h, const real* V, int k, real* part, const real* add, real beta, double* dots,
                            hipStream_t s, const SymDmaParts* shard_tab = nullptr, int g_lo = 0, int g_hi = -1) {
  // a stencil shard is given as a part table (the LDS-DMA kernel: d = 3, fp32, k = 1) or as a group range (the LDS-window kernel)
  if (shard_tab && !(sizeof(real) == 4 && sym_use_dma<real>(G, k))) return WISKI_E_BADARG;
  const bool ranged = g_hi >= 0;
  if (ranged && (shard_tab || k != 1 || g_lo < 0 || g_hi < g_lo || g_hi > sym_groups(G.d))) return WISKI_E_BADARG;
  if (sym_use_cols(k, sizeof(real)) && !ranged) {
    // part[0] <- A V (column-major, written by the product itself); the row-major copy of V lives behind it
    const int m = G.m, kp = spmmc_kp(k);
    const int64_t km = (int64_t)k * m;
    real* Vt = part + km;
    real* Ot = part;
    if (sym_use_bcast()) {   // coefficients on the vector path, DPP row broadcast (spmm_sym_bcast.h); operands in 64-column slices
      const int ns = (k + 63) / 64;
      dim3 tg((unsigned)(8 * (((m + 63) / 64 + 7) / 8)), (unsigned)ns);            // (padded: XCD-contiguous row tiles, see the kernel)
      if (dots && add) hipLaunchKernelGGL((k_transpose_cm_rm<real, true, true>), tg, dim3(256), 0, s, m, k, 64 * ns, V, Vt, add, beta, dots);
      else hipLaunchKernelGGL((k_transpose_cm_rm<real, false, true>), tg, dim3(256), 0, s, m, k, 64 * ns, V, Vt, (const real*)nullptr, (real)0, (double*)nullptr);
      const int ng = sym_groups(G.d);
      const int64_t a_len = (int64_t)(7 * ng - 3) * m;
      constexpr int RT = 16;   // rows per wave tile (measured at 50^3, fp32 / fp64: RT = 8 155 / 452 us, 16 137 / 372, 24 150 / 379, 32 149 / 411)
      const int ntb = (m + RT - 1) / RT, nbb = (ntb + 3) / 4;                  // 4 tiles (waves) per block
      // grid.x is padded to a multiple of 8: workgroup b runs on XCD b % 8 and takes tile (b % 8) * (grid.x / 8) + b / 8 (see below)
      dim3 gb((unsigned)((nbb + 7) / 8 * 8), (unsigned)ns);
#define SPMMB_LAUNCH(VAR)                                                                                                                  \
  do {                                                                                                                                     \
    if (dots) launch_timed(k_spmm_sym_bcast<real, true, RT, VAR>, gb, dim3(256), 0, s, G, A_h, a_len, (const real*)Vt, k, ng, Ot, dots);     \
    else launch_timed(k_spmm_sym_bcast<real, false, RT, VAR>, gb, dim3(256), 0, s, G, A_h, a_len, (const real*)Vt, k, ng, Ot, dots);        \
  } while (0)
#ifdef WISKI_SPMMB_ABLATE
      if (g_spmm_bcast == 3) SPMMB_LAUNCH(3);
      else if (g_spmm_bcast == 4) SPMMB_LAUNCH(4);
      else
#endif
        SPMMB_LAUNCH(1);
#undef SPMMB_LAUNCH
      return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
    }
    dim3 tg((unsigned)(8 * (((m + 63) / 64 + 7) / 8)), (unsigned)((kp + 63) / 64));
    if (dots && add) hipLaunchKernelGGL((k_transpose_cm_rm<real, true>), tg, dim3(256), 0, s, m, k, kp, V, Vt, add, beta, dots);
    else hipLaunchKernelGGL((k_transpose_cm_rm<real, false>), tg, dim3(256), 0, s, m, k, kp, V, Vt, (const real*)nullptr, (real)0, (double*)nullptr);
    // grid.x is padded to a multiple of 8: workgroup b runs on XCD b % 8 and takes tile (b % 8) * (grid.x / 8) + b / 8, so
    // each XCD sweeps ONE contiguous eighth of the rows and its L2 holds the v windows of neighbouring tiles
    const int ntile = (m + SPMMC_RT - 1) / SPMMC_RT, nblk = (ntile + 3) / 4;    // 4 tiles (waves) per block
    dim3 grd((unsigned)((nblk + 7) / 8 * 8), (unsigned)((kp + 63) / 64));
    const int ng = sym_groups(G.d);
    if (kp == 64) {
      if (dots) launch_timed(k_spmm_sym_cols<real, true, 64>, grd, dim3(256), 0, s, G, A_h, (const real*)Vt, k, kp, ng, Ot, dots);
      else launch_timed(k_spmm_sym_cols<real, false, 64>, grd, dim3(256), 0, s, G, A_h, (const real*)Vt, k, kp, ng, Ot, dots);
    } else {
      if (dots) launch_timed(k_spmm_sym_cols<real, true, 0>, grd, dim3(256), 0, s, G, A_h, (const real*)Vt, k, kp, ng, Ot, dots);
      else launch_timed(k_spmm_sym_cols<real, false, 0>, grd, dim3(256), 0, s, G, A_h, (const real*)Vt, k, kp, ng, Ot, dots);
    }
    return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
  }
  if constexpr (sizeof(real) == 4) {
    if (sym_use_dma<real>(G, k) && !ranged) {
      const int W4 = symdma_w4(G.g[2]), WP = symdma_wp(G.g[2]);
      const size_t sh = symdma_lds_bytes(G.g[2], g_sym_dma_nst);
      SymDmaParts tab{};
      if (shard_tab) tab = *shard_tab;
      // partial vectors written: part[0 .. np); the accumulated one always lives at part[g_sym_dma_parts], whatever np is
      // (the zero regions of wiski_pcg_zero_regions must not depend on the sharding)
      const int np = tab.n ? tab.n : g_sym_dma_parts;
      if (np > g_sym_dma_parts) return WISKI_E_BADARG;
      const int nrb = (G.m + 255) / 256;
      const int xcd_rb = sym_xcd_map((int64_t)(7 * sym_groups(G.d) - 3) * G.m * (int64_t)sizeof(real)) ? (nrb + 7) / 8 : 0;
      dim3 grd((unsigned)(xcd_rb ? 8 * xcd_rb : nrb), (unsigned)np);
      // the light chunk joins when the heavy ones are ~3 / 7 through their stream: proportional to the stream's length
      const int delay = (int)((int64_t)g_sym_dma_delay * G.m / 125000);
#define SYMDMA(NST, DOT)                                                                                                          \
  do {                                                                                                                            \
    static size_t lds_set = 0;                                                                                                    \
    if (sh > 48 * 1024 && sh > lds_set) {                                                                                         \
      if (hipFuncSetAttribute((const void*)k_spmv_sym_dma<NST, DOT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh) != hipSuccess) \
        return WISKI_E_LAUNCH;                                                                                                    \
      lds_set = sh;                                                                                                               \
    }                                                                                                                             \
    unsigned long long* st_p = prof_stamp_pairs((size_t)grd.x * grd.y);                                                           \
    launch_timed(k_spmv_sym_dma<NST, DOT>, grd, dim3(64), sh, s, G, A_h, V, W4, WP, g_sym_dma_parts, part, add, beta, dots, delay, tab, xcd_rb, st_p); \
  } while (0)
      if (dots) SYMDMA(2, true); else SYMDMA(2, false);
#undef SYMDMA
      return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
    }
  }
  if constexpr (sizeof(real) == 4) {
    if (sym_use_dma_mc<real>(G, k) && !ranged) {
      const int kc = k >= 4 ? 4 : 2;
      const int W4 = symdma_w4(G.g[2]), WP = symdma_wp(G.g[2]);
      const size_t sh = symdma_mc_lds_bytes(G.g[2], kc);
      const int nrb = (G.m + 255) / 256;
      const int xcd_rb = sym_xcd_map((int64_t)(7 * sym_groups(G.d) - 3) * G.m * (int64_t)sizeof(real)) ? (nrb + 7) / 8 : 0;
      dim3 grd((unsigned)(xcd_rb ? 8 * xcd_rb : nrb), 4u, (unsigned)((k + kc - 1) / kc));
#define SYMDMAMC(KC, DOT)                                                                                                          \
  do {                                                                                                                             \
    static size_t lds_set = 0;                                                                                                     \
    if (sh > 48 * 1024 && sh > lds_set) {                                                                                          \
      if (hipFuncSetAttribute((const void*)k_spmv_sym_dma_mc<KC, DOT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh) != hipSuccess) \
        return WISKI_E_LAUNCH;                                                                                                     \
      lds_set = sh;                                                                                                                \
    }                                                                                                                              \
    launch_timed(k_spmv_sym_dma_mc<KC, DOT>, grd, dim3(64), sh, s, G, A_h, V, k, W4, WP, part, add, beta, dots, xcd_rb);           \
  } while (0)
      if (kc == 4) { if (dots) SYMDMAMC(4, true); else SYMDMAMC(4, false); }
      else { if (dots) SYMDMAMC(2, true); else SYMDMAMC(2, false); }
#undef SYMDMAMC
      return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
    }
  }
  const int ng = sym_groups(G.d), nch = sym_nch_lds(G.d);
  if (ng - (int)((int64_t)(nch - 1) * ng / nch) > 176 || ng > 176 * nch) return WISKI_E_BADARG;
  // columns per pass over A_h (8 per pass, with the v / transposed windows of 4 live at a time, was measured:
  // 344 / 488 VGPRs leave one wave per SIMD, 1.3x / 2.2x slower)
  const int kc = k >= 4 ? 4 : (k >= 2 ? 2 : 1);
  // window span: one full cycle of the second-to-last stencil digit, shrunk to fit 48 KB of LDS
  int span = G.d >= 2 ? 6 * G.stride[G.d - 2] : 0;
  const int bs = sym_block(), nw = bs / 64;
  // per wave: a 7 x 256 tile of A_h plus kc windows; the windows are shrunk to keep a block within 64 KB
  const int cap = (int)((64 * 1024 / (nw * sizeof(real)) - 7 * 256) / kc) - 256 - 32;
  if (span > cap) span = cap > 0 ? cap : 0;
  const int W4 = ((256 + span + 10 + 3) / 4) | 1;
  const size_t sh = (size_t)nw * (7 * 256 + kc * 4 * W4) * sizeof(real);
  const int nrb = (G.m + 4 * bs - 1) / (4 * bs);
  const int xcd_rb = sym_xcd_map((int64_t)(7 * ng - 3) * G.m * (int64_t)sizeof(real)) ? (nrb + 7) / 8 : 0;
  dim3 grd((unsigned)(xcd_rb ? 8 * xcd_rb : nrb), (unsigned)nch, (unsigned)((k + kc - 1) / kc));
#define SPMV4S(KC)                                                                                                                              \
  do {                                                                                                                                          \
    static size_t lds_set[2] = {0, 0};   /* > 48 KB of dynamic LDS needs an opt-in per kernel */                                                \
    const int di = dots ? 1 : 0;                                                                                                                \
    if (sh > 48 * 1024 && sh > lds_set[di]) {                                                                                                   \
      const void* fn = dots ? (const void*)k_stencil_spmv4_sym<real, KC, true> : (const void*)k_stencil_spmv4_sym<real, KC, false>;             \
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh) != hipSuccess) return WISKI_E_LAUNCH;                   \
      lds_set[di] = sh;                                                                                                                         \
    }                                                                                                                                           \
    if (dots) launch_timed(k_stencil_spmv4_sym<real, KC, true>, grd, dim3(bs), sh, s, G, A_h, V, k, ng, nch, span, W4, part, add, beta, dots,   \
                           ranged ? g_lo : 0, ranged ? g_hi : ng, xcd_rb);                                                                      \
    else launch_timed(k_stencil_spmv4_sym<real, KC, false>, grd, dim3(bs), sh, s, G, A_h, V, k, ng, nch, span, W4, part, add, beta, dots,       \
                      ranged ? g_lo : 0, ranged ? g_hi : ng, xcd_rb);                                                                           \
  } while (0)
  if (kc == 4) SPMV4S(4);
  else if (kc == 2) SPMV4S(2);
  else SPMV4S(1);
#undef SPMV4S
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}

// -------------------------------------------------- Kronecker-Toeplitz MVM --
// One mode product of the d-way tensor view (pre, g, post):
//   out[pp, i, s] = scale * sum_j tcol[|i-j|] * in[pp, j, s]
// One thread per output element; the g reads per output are coalesced along s
// (or broadcast along i for the last mode) and L2-resident (a column is m reals).
// DOT: dots[c] += sum_e w[c][e] * out[c][e]   (CG's r.y on the last mode).
template <typename real, bool DOT>
__global__ __launch_bounds__(256) void k_toeplitz_mode(const real* __restrict__ tcol, int g, int post, int m, const real* __restrict__ in,
                                                       real scale, real* __restrict__ out, const real* __restrict__ wvec,
                                                       double* __restrict__ dots) {
  __shared__ real s_t[256];
  __shared__ double s_red[16];
  for (int j = threadIdx.x; j < g; j += blockDim.x) s_t[j] = tcol[j];
  __syncthreads();
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = blockIdx.y;
  double part = 0;
  if (e < m) {
    const int i = (e / post) % g;
    const real* __restrict__ src = in + (int64_t)c * m + (e - i * post);
    real acc = (real)0;
#pragma unroll 4
    for (int j = 0; j < g; ++j) {
      const int lag = i > j ? i - j : j - i;
      acc += s_t[lag] * src[(int64_t)j * post];
    }
    acc *= scale;
    out[(int64_t)c * m + e] = acc;
    if (DOT) part = (double)wvec[(int64_t)c * m + e] * (double)acc;
  }
  if (DOT) {
    double tot = block_reduce_sum(part, s_red);
    if (threadIdx.x == 0) unsafeAtomicAdd(dots + c, tot);
  }
}

// Innermost mode (post == 1, g <= 64): one 64-lane wave per grid line, lane = position in the line.  The line is loaded once
// (coalesced) and the symmetric-Toeplitz product is a running pair of wavefront shifts:
//   out_i = t_0 v_i + sum_{s >= 1} t_s (v_{i-s} + v_{i+s}),
// v_{i-s} / v_{i+s} obtained from the previous step by one-lane shuffles (zeros shifted in at the line ends); the
// coefficient t_s is wave-uniform.  g shuffles-up, g shuffles-down and 2 g VALU per line instead of g dependent L2 loads per
// output.  DOT: dots[c] += sum_e w[c][e] * out[c][e].
template <typename real, bool DOT>
__global__ __launch_bounds__(256) void k_toeplitz_line_shfl(const real* __restrict__ tcol, int g, int m, const real* __restrict__ in, real scale,
                                                            real* __restrict__ out, const real* __restrict__ wvec, double* __restrict__ dots) {
  __shared__ double s_red[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nlines = m / g;
  const int c = blockIdx.y;
  const int64_t cm = (int64_t)c * m;
  double part = 0;
  for (int line = blockIdx.x * 4 + wave; line < nlines; line += gridDim.x * 4) {
    const int64_t e = cm + (int64_t)line * g + lane;
    const bool live = lane < g;
    const real v = live ? in[e] : (real)0;
    real acc = tcol[0] * v;
    real up = v, dn = v;                       // v_{i-s}, v_{i+s}
    for (int sft = 1; sft < g; ++sft) {
      up = __shfl_up(up, 1, 64);
      dn = __shfl_down(dn, 1, 64);
      if (lane == 0) up = (real)0;             // zeros enter at the line ends (lanes >= g hold zeros already)
      if (lane == 63) dn = (real)0;
      acc += tcol[sft] * (up + dn);
    }
    if (live) {
      acc *= scale;
      out[e] = acc;
      if (DOT) part += (double)wvec[e] * (double)acc;
    }
  }
  if (DOT) {
    const double tot = block_reduce_sum(part, s_red);
    if (threadIdx.x == 0) unsafeAtomicAdd(dots + c, tot);
  }
}

// out = scale * Kuu V ; tmp is a k*m scratch; out must not alias V.
template <typename real>
static int launch_kron(const GridDev<real>& G, const real* tcol, const real* V, int k, real scale, real* tmp, real* out, const real* wvec,
                       double* dots, hipStream_t s) {
  if (G.d > 1 && !tmp) return WISKI_E_BADARG;
  for (int q = 0; q < G.d; ++q)
    if (G.g[q] > 256) return WISKI_E_BADARG;
  dim3 grd((unsigned)((G.m + 255) / 256), (unsigned)k);
  // ping-pong so that the last mode lands in `out`
  const real* src = V;
  int toff = 0;
  for (int q = 0; q < G.d; ++q) {
    const bool last = q == G.d - 1;
    real* dst = ((G.d - 1 - q) % 2 == 0) ? out : tmp;
    const real sc = last ? scale : (real)1;
    const int gq = G.g[q], post = G.stride[q];
    // innermost mode: wavefront shuffles along the line; the strided modes keep the one-thread-per-output kernel (its g reads per
    // output are coalesced along the fast index and L2-resident; an LDS-tiled thread-per-line variant measured 2x slower)
    if (post == 1 && gq <= 64) {
      int blocks = (G.m / gq + 3) / 4;
      if (blocks > 2048) blocks = 2048;
      dim3 lg((unsigned)blocks, (unsigned)k);
      if (last && dots) hipLaunchKernelGGL((k_toeplitz_line_shfl<real, true>), lg, dim3(256), 0, s, tcol + toff, gq, G.m, src, sc, dst, wvec, dots);
      else hipLaunchKernelGGL((k_toeplitz_line_shfl<real, false>), lg, dim3(256), 0, s, tcol + toff, gq, G.m, src, sc, dst, wvec, dots);
    } else if (last && dots)
      hipLaunchKernelGGL((k_toeplitz_mode<real, true>), grd, dim3(256), 0, s, tcol + toff, G.g[q], G.stride[q], G.m, src, sc, dst, wvec, dots);
    else
      hipLaunchKernelGGL((k_toeplitz_mode<real, false>), grd, dim3(256), 0, s, tcol + toff, G.g[q], G.stride[q], G.m, src, sc, dst, wvec, dots);
    src = dst;
    toff += G.g[q];
  }
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}

// ------------------------------------------- dense Kronecker mode products ---
// out[pp, i, s] = sum_j F[i, j] * in[pp, j, s] for a general g x g factor F (held
// in LDS).  Used for the Kronecker eigenbasis of Kuu (Kuu = (kron V_q) Lam (kron V_q)^T):
//   MODE 0  plain
//   MODE 1  spectral scaling epilogue of the forward transform: with
//           lam = kscale * prod_q eval_q[i_q],  f1 = 1/(1 + shift*lam):
//           out = f1 * acc  and  out[out2_off + .] = lam * f1 * acc
//   MODE 2  dot epilogue: dots[c - dot_c0] += sum_e wvec[c - dot_c0][e] * acc for c >= dot_c0
template <typename real, int MODE>
__global__ __launch_bounds__(256) void k_dense_mode(GridDev<real> G, int q, const real* __restrict__ Fmain, const real* __restrict__ Falt, int split,
                                                    int transposed,
                                                    const real* __restrict__ in, real* __restrict__ out, int64_t out2_off,
                                                    const real* __restrict__ evals, real kscale, real shift,
                                                    const real* __restrict__ wvec, int dot_c0, double* __restrict__ dots) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  real* sF = reinterpret_cast<real*>(smem);
  __shared__ double s_red[16];
  const int g = G.g[q], post = G.stride[q], m = G.m;
  const real* __restrict__ F = (Falt != nullptr && (int)blockIdx.y < split) ? Falt : Fmain;   // columns [0, split) may use another factor
  for (int t = threadIdx.x; t < g * g; t += blockDim.x) {
    const int i = t / g, j = t - i * g;
    sF[t] = transposed ? F[j * g + i] : F[t];
  }
  __syncthreads();
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = blockIdx.y;
  double part = 0;
  if (e < m) {
    const int i = (e / post) % g;
    const real* __restrict__ src = in + (int64_t)c * m + (e - i * post);
    const real* __restrict__ frow = sF + i * g;
    real acc = (real)0;
#pragma unroll 4
    for (int j = 0; j < g; ++j) acc += frow[j] * src[(int64_t)j * post];
    if (MODE == 1) {
      real lam = kscale;
      int eoff = 0;
      for (int qq = 0; qq < G.d; ++qq) {
        lam *= evals[eoff + (e / G.stride[qq]) % G.g[qq]];
        eoff += G.g[qq];
      }
      const real f1 = (real)1 / ((real)1 + shift * lam);
      out[(int64_t)c * m + e] = f1 * acc;
      out[out2_off + (int64_t)c * m + e] = lam * f1 * acc;
    } else {
      out[(int64_t)c * m + e] = acc;
      if (MODE == 2 && c >= dot_c0) part = (double)wvec[(int64_t)(c - dot_c0) * m + e] * (double)acc;
    }
  }
  if (MODE == 2 && c >= dot_c0) {
    const double tot = block_reduce_sum(part, s_red);
    if (threadIdx.x == 0) unsafeAtomicAdd(dots + (c - dot_c0), tot);
  }
}

// Spectral preconditioner apply:  t = (I + a Kt)^-1 r,  y = Kt t = (Kt^-1 + a I)^-1 r
// via Kt = V diag(lam) V^T.  ty = [t (k cols) | y (k cols)]; sa/sb: 2k-column scratch.
// Also accumulates rho[c] += r[c] . y[c].
template <typename real>
static int launch_spectral_precond(const GridDev<real>& G, const real* evec, const real* evec2, const real* evals, real kscale, real shift,
                                   const real* r, int k,
                                   real* sa, real* sb, real* ty, double* rho, hipStream_t s) {
  const int m = G.m, d = G.d;
  const int64_t km = (int64_t)k * m;
  if (spectral_fused_ok<real>(G)) return launch_spectral_fused<real>(G, evec, evec2, evals, kscale, shift, r, k, sa, sb, ty, rho, s);
  int eoff[WISKI_MAX_DIM + 1];
  eoff[0] = 0;
  for (int q = 0; q < d; ++q) {
    if (G.g[q] > 128) return WISKI_E_BADARG;
    eoff[q + 1] = eoff[q] + G.g[q] * G.g[q];
  }
  const unsigned gx = (unsigned)((m + 255) / 256);
  const real* cur = r;
  // forward: w = (kron V^T) r, scaled on the last mode into two column groups
  for (int q = 0; q < d; ++q) {
    real* dst = (cur == sa) ? sb : sa;
    const size_t sh = (size_t)G.g[q] * G.g[q] * sizeof(real);
    if (q < d - 1)
      hipLaunchKernelGGL((k_dense_mode<real, 0>), dim3(gx, (unsigned)k), dim3(256), sh, s, G, q, evec + eoff[q], (const real*)nullptr, 0, 1, cur, dst, (int64_t)0, evals,
                         kscale, shift, (const real*)nullptr, 0, (double*)nullptr);
    else
      hipLaunchKernelGGL((k_dense_mode<real, 1>), dim3(gx, (unsigned)k), dim3(256), sh, s, G, q, evec + eoff[q], (const real*)nullptr, 0, 1, cur, dst, km, evals, kscale,
                         shift, (const real*)nullptr, 0, (double*)nullptr);
    cur = dst;
  }
  // backward on 2k columns: [t | y] = (kron V) [f1 w | f2 w]; rho += r . y on the last mode
  for (int q = 0; q < d; ++q) {
    const bool last = q == d - 1;
    real* dst = last ? ty : ((cur == sa) ? sb : sa);
    const size_t sh = (size_t)G.g[q] * G.g[q] * sizeof(real);
    if (!last)
      hipLaunchKernelGGL((k_dense_mode<real, 0>), dim3(gx, (unsigned)(2 * k)), dim3(256), sh, s, G, q, evec + eoff[q], evec2 ? evec2 + eoff[q] : (const real*)nullptr, k, 0, cur, dst, (int64_t)0,
                         evals, kscale, shift, (const real*)nullptr, 0, (double*)nullptr);
    else
      hipLaunchKernelGGL((k_dense_mode<real, 2>), dim3(gx, (unsigned)(2 * k)), dim3(256), sh, s, G, q, evec + eoff[q], evec2 ? evec2 + eoff[q] : (const real*)nullptr, k, 0, cur, dst, (int64_t)0,
                         evals, kscale, shift, r, k, rho);
    cur = dst;
  }
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}

// ------------------------------------------- d/d tcol of bilinear Kron forms ---
// For symmetric-Toeplitz Kronecker factors K = kron_q T(tcol_q):
//   d/d tcol_q[l]  sum_c x_c^T K y_c  =  sum_c sum_{pp,s} sum_{|i-j| = l} X_c[pp,i,s] * Yq_c[pp,j,s]
// with Yq = (kron_{r != q} T_r) Y (all modes but q applied; done by the caller with
// a delta first column for dim q).  One thread per element of X; lag bins are
// accumulated with LDS fp64 atomics, then one global fp64 atomic per bin per block.
// Used by the Woodbury MLL backward (BWM:19-51 gradients; not on the streaming path).
template <typename real>
__global__ __launch_bounds__(256) void k_lag_correlate(int g, int post, int m, int k, const real* __restrict__ X, const real* __restrict__ Yq,
                                                       double* __restrict__ out) {
  // out[lag] += sum_c sum_{e} X[c][e] * Yq[c][e with mode-q index j],  lag = |i(e) - j|.
  // One thread per element e and block.y per chunk of LAG_CK columns: the thread keeps its x values of the chunk in
  // registers, reduces over the columns in fp64 registers and issues ONE LDS atomic per (e, j) -- not one per (e, j, c):
  // the dense-regime MLL calls this with k = m + 1 columns, where the per-column atomics cost 2.5 ms per call.
  constexpr int LAG_CK = 32;
  __shared__ double s_bins[256];
  for (int l = threadIdx.x; l < g; l += blockDim.x) s_bins[l] = 0.0;
  __syncthreads();
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int c0 = blockIdx.y * LAG_CK;
  if (e < m) {
    const int i = (e / post) % g;
    real xr[LAG_CK];
#pragma unroll
    for (int cc = 0; cc < LAG_CK; ++cc) xr[cc] = c0 + cc < k ? X[(int64_t)(c0 + cc) * m + e] : (real)0;
    const real* __restrict__ src = Yq + (int64_t)c0 * m + (e - i * post);
    for (int j = 0; j < g; ++j) {
      double acc = 0.0;
#pragma unroll
      for (int cc = 0; cc < LAG_CK; ++cc)
        if (c0 + cc < k) acc += (double)xr[cc] * (double)src[(int64_t)cc * m + (int64_t)j * post];
      const int lag = i > j ? i - j : j - i;
      atomicAdd(&s_bins[lag], acc);
    }
  }
  __syncthreads();
  for (int l = threadIdx.x; l < g; l += blockDim.x)
    if (s_bins[l] != 0.0) unsafeAtomicAdd(out + l, s_bins[l]);
}

// The same sums through tile Gram matrices (round 3; used for g <= 64): a block takes a g x 64 tile of X and of Yq -- one
// (column, outer index) slab and 64 consecutive elements of the unit-stride direction -- into LDS with coalesced loads, forms the
// g x g products sum_s X[i][s] Yq[j][s] from LDS (fp32 partial sums of 64 terms, accumulated in fp64) and bins THEM by lag:
// g^2 LDS atomics per 64 g elements instead of g per element, and no strided global re-reads (k_lag_correlate re-reads each
// Yq line g times: 125 us per call at 50^3 with 11 columns; this form: see DESIGN.md 3.6).
//   element (i, s) of a tile = base + i * stride_i + s * stride_s with (stride_i, stride_s) = (post, 1) when post > 1 and (1, g)
//   for the innermost mode (the tile then spans 64 consecutive lines).
template <typename real>
__global__ __launch_bounds__(256) void k_lag_gram(int g, int post, int m, int k, const real* __restrict__ X, const real* __restrict__ Yq,
                                                  double* __restrict__ out) {
  constexpr int TS = 64;
  __shared__ real sX[64][TS + 1], sY[64][TS + 1];
  __shared__ double s_bins[64];
  const int t = threadIdx.x;
  if (t < 64) s_bins[t] = 0.0;
  const int c = blockIdx.y;
  const real* __restrict__ Xc = X + (int64_t)c * m;
  const real* __restrict__ Yc = Yq + (int64_t)c * m;
  // tiles of this column: post > 1: (outer index pp, chunk of s); post == 1: chunks of 64 lines
  const int nchunk = post > 1 ? (post + TS - 1) / TS : 1;
  const int nouter = post > 1 ? m / (g * post) : (m / g + TS - 1) / TS;
  for (int tile = blockIdx.x; tile < nouter * nchunk; tile += gridDim.x) {
    int64_t base;
    int si, ss, ns;
    if (post > 1) {
      const int pp = tile / nchunk, s0 = (tile % nchunk) * TS;
      base = (int64_t)pp * g * post + s0;
      si = post; ss = 1;
      ns = post - s0 < TS ? post - s0 : TS;
    } else {
      const int l0 = tile * TS;
      base = (int64_t)l0 * g;
      si = 1; ss = g;
      ns = m / g - l0 < TS ? m / g - l0 : TS;
    }
    __syncthreads();                                       // previous tile's readers are done
    if (post > 1) {                                        // unit stride along s
      for (int e = t; e < g * TS; e += 256) {
        const int i = e / TS, sidx = e % TS;
        const bool ok = sidx < ns;
        sX[i][sidx] = ok ? Xc[base + (int64_t)i * si + sidx] : (real)0;
        sY[i][sidx] = ok ? Yc[base + (int64_t)i * si + sidx] : (real)0;
      }
    } else {                                               // unit stride along i (g contiguous reals per line)
      for (int e = t; e < g * TS; e += 256) {
        const int sidx = e / g, i = e % g;
        const bool ok = sidx < ns;
        sX[i][sidx] = ok ? Xc[base + (int64_t)sidx * ss + i] : (real)0;
        sY[i][sidx] = ok ? Yc[base + (int64_t)sidx * ss + i] : (real)0;
      }
    }
    __syncthreads();
    for (int p = t; p < g * g; p += 256) {
      const int i = p / g, j = p % g;
      real acc = (real)0;
#pragma unroll 16
      for (int sidx = 0; sidx < TS; ++sidx) acc += sX[i][sidx] * sY[j][sidx];
      const int lag = i > j ? i - j : j - i;
      atomicAdd(&s_bins[lag], (double)acc);
    }
  }
  __syncthreads();
  if (t < g && s_bins[t] != 0.0) unsafeAtomicAdd(out + t, s_bins[t]);
}

// d_grad[sum g] (double, accumulated into) ; d_tmp: 2*k*m reals of scratch.
template <typename real>
static int kron_grad_impl(const wiski_grid* grid, const real* d_tcol, const real* d_X, const real* d_Y, int32_t k, real* d_tmp,
                          double* d_grad, void* stream) {
  GridDev<real> G;
  int rc = make_grid_dev<real>(grid, &G);
  if (rc) return rc;
  if (!d_tcol || !d_X || !d_Y || !d_tmp || !d_grad || k < 1) return WISKI_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  const int m = G.m;
  real* yq = d_tmp;
  real* scratch = d_tmp + (int64_t)k * m;
  dim3 grd((unsigned)((m + 255) / 256), (unsigned)k);
  int toff = 0;
  for (int q = 0; q < G.d; ++q) {
    if (G.g[q] > 256) return WISKI_E_BADARG;
    // Yq = product of all modes r != q applied to Y (ping-pong yq <-> scratch)
    const real* src = d_Y;
    int nmodes = G.d - 1, done = 0, off2 = 0;
    for (int r = 0; r < G.d; ++r) {
      if (r != q) {
        // land the last applied mode in yq
        real* dst = ((nmodes - 1 - done) % 2 == 0) ? yq : scratch;
        hipLaunchKernelGGL((k_toeplitz_mode<real, false>), grd, dim3(256), 0, s, d_tcol + off2, G.g[r], G.stride[r], m, src, (real)1, dst,
                           (const real*)nullptr, (double*)nullptr);
        src = dst;
        ++done;
      }
      off2 += G.g[r];
    }
    if (G.g[q] <= 64 && k <= 4096 && m % G.g[q] == 0) {
      const int post = G.stride[q];
      const int ntile = post > 1 ? (m / (G.g[q] * post)) * ((post + 63) / 64) : (m / G.g[q] + 63) / 64;
      int bx = ntile < 1 ? 1 : ntile;
      if ((int64_t)bx * k > 4096) bx = (int)(4096 / k) > 0 ? (int)(4096 / k) : 1;      // grid-stride over the tiles of a column
      hipLaunchKernelGGL((k_lag_gram<real>), dim3((unsigned)bx, (unsigned)k), dim3(256), 0, s, G.g[q], post, m, k, d_X, src, d_grad + toff);
    } else {
      hipLaunchKernelGGL((k_lag_correlate<real>), dim3((unsigned)((m + 255) / 256), (unsigned)((k + 31) / 32)), dim3(256), 0, s, G.g[q], G.stride[q],
                         m, k, d_X, src, d_grad + toff);
    }
    toff += G.g[q];
  }
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}

// ------------------------------------------------ general spectral Kron apply ---
// out = (kron V_q) diag(f(lam)) (kron V_q)^T v,  lam = kscale * prod_q eval_q[i_q],
// f(lam) = lam^p / (1 + shift*lam)^r : p = 1/2, r = 0 is Kt^{1/2} (symmetric root
// used by the stochastic-Lanczos logdet); p = 1, r = 1 the CG preconditioner.
template <typename real>
__global__ __launch_bounds__(256) void k_spectral_scale(GridDev<real> G, const real* __restrict__ evals, real kscale, real shift, real pw, real rw,
                                                        real* __restrict__ v) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = blockIdx.y;
  if (e >= G.m) return;
  real lam = kscale;
  int eoff = 0;
  for (int qq = 0; qq < G.d; ++qq) {
    lam *= evals[eoff + (e / G.stride[qq]) % G.g[qq]];
    eoff += G.g[qq];
  }
  real f = (real)1;
  if (pw != (real)0) f = lam > (real)0 ? (real)pow((double)lam, (double)pw) : (real)0;
  if (rw != (real)0) f *= (real)pow(1.0 + (double)shift * (double)lam, -(double)rw);
  v[(int64_t)c * G.m + e] *= f;
}

template <typename real>
static int spectral_mm_impl(const wiski_grid* grid, const real* d_evec, const real* d_eval, real kscale, real shift, real pw, real rw,
                            const real* d_V, int32_t k, real* d_tmp, real* d_out, void* stream) {
  GridDev<real> G;
  int rc = make_grid_dev<real>(grid, &G);
  if (rc) return rc;
  if (!d_evec || !d_eval || !d_V || !d_tmp || !d_out || k < 1 || d_out == d_V) return WISKI_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  const int m = G.m, d = G.d;
  int eoff[WISKI_MAX_DIM + 1];
  eoff[0] = 0;
  for (int q = 0; q < d; ++q) {
    if (G.g[q] > 128) return WISKI_E_BADARG;
    eoff[q + 1] = eoff[q] + G.g[q] * G.g[q];
  }
  dim3 grd((unsigned)((m + 255) / 256), (unsigned)k);
  const real* cur = d_V;
  // 2d mode products; ping-pong so that the last one lands in d_out
  for (int step = 0; step < 2 * d; ++step) {
    const int q = step < d ? step : step - d;
    const int transposed = step < d ? 1 : 0;
    real* dst = ((2 * d - 1 - step) % 2 == 0) ? d_out : d_tmp;
    const size_t sh = (size_t)G.g[q] * G.g[q] * sizeof(real);
    hipLaunchKernelGGL((k_dense_mode<real, 0>), grd, dim3(256), sh, s, G, q, d_evec + eoff[q], (const real*)nullptr, 0, transposed, cur, dst, (int64_t)0, d_eval, kscale,
                       shift, (const real*)nullptr, 0, (double*)nullptr);
    cur = dst;
    if (step == d - 1) hipLaunchKernelGGL((k_spectral_scale<real>), grd, dim3(256), 0, s, G, d_eval, kscale, shift, pw, rw, dst);
  }
  return hipGetLastError() == hipSuccess ? WISKI_OK : WISKI_E_LAUNCH;
}

// -------------------------------------------------------------------- PCG ---
// scalar slots (double): S[0..k) = ||rhs||^2 ; then per iteration slot
// it in [0, max_iter]: rho[k], php[k][PCG_DOT_SLOTS], rn[k] (struct PcgScal).  rn of slot 0 = ||r0||^2.
// Convergence polling without a stream synchronisation: a one-wave kernel queued behind the
// iteration copies the residual norms (and the caller's out-of-grid flag) into a host-mapped,
// coherent buffer and then releases a sequence number; the host spins on it.  A poll costs a few
// microseconds instead of the ~80 us of hipMemcpyAsync + hipStreamSynchronize + relaunch bubble.
struct WiskiPoll {
  double* h = nullptr;       // host view:   [0] = sequence (as int64), [1..k] = rn0, [k+1..2k] = rn, [2k+1] = err flag, [2k+2] = guard
  double* d = nullptr;       // device view of the same pinned allocation
  long long* g = nullptr;    // device memory: +seq when the poll `seq` found every column converged and no error flag, else -seq --
                             // the guard of a speculatively queued absorb (wiski_scatter_stats_step); mirrored in h[2k+2]
  int cap = 0;
  long long seq = 0;
};
// One poll buffer per device, handed out under a mutex and owned by the calling solve until it returns (a
// second host thread solving on the same device waits its turn instead of interleaving sequence numbers).
static std::mutex g_poll_mu;
static std::map<int, WiskiPoll> g_polls;

static int poll_reserve(WiskiPoll& P, int k) {
  if (P.cap >= k) return WISKI_OK;
  if (P.h) (void)hipHostFree(P.h);
  const int cap = k < 64 ? 64 : k;
  if (hipHostMalloc((void**)&P.h, (size_t)(2 * cap + 3) * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent | hipHostMallocPortable) !=
      hipSuccess)
    return WISKI_E_LAUNCH;
  if (hipHostGetDevicePointer((void**)&P.d, P.h, 0) != hipSuccess) return WISKI_E_LAUNCH;
  if (!P.g) {
    if (hipMalloc((void**)&P.g, sizeof(long long)) != hipSuccess) return WISKI_E_LAUNCH;
    if (hipMemset(P.g, 0, sizeof(long long)) != hipSuccess) return WISKI_E_LAUNCH;
  }
  P.h[0] = 0;
  P.cap = cap;
  return WISKI_OK;
}

// spin until the device has released sequence number `seq` into the poll buffer (20 s fallback: stream sync)
#ifdef WISKI_STEP_TIMING   // host time spent waiting for polls (tools/spec_probe.py): is a streaming loop GPU- or host-bound?
static double g_poll_wait_us = 0;
static long long g_poll_waits = 0, g_poll_waits_immediate = 0;
extern "C" void wiski_debug_poll_wait(double* us, long long* n, long long* immediate) {
  *us = g_poll_wait_us; *n = g_poll_waits; *immediate = g_poll_waits_immediate;
  g_poll_wait_us = 0; g_poll_waits = 0; g_poll_waits_immediate = 0;
}
struct PollWaitTimer {
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  bool immediate;
  explicit PollWaitTimer(bool imm) : immediate(imm) {}
  ~PollWaitTimer() {
    g_poll_wait_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    ++g_poll_waits;
    if (immediate) ++g_poll_waits_immediate;
  }
};
#endif
static int poll_wait(WiskiPoll& P, long long seq, hipStream_t s) {
  volatile long long* flag = reinterpret_cast<volatile long long*>(P.h);
#ifdef WISKI_STEP_TIMING
  PollWaitTimer pwt(__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq);
#endif
  const auto t0 = std::chrono::steady_clock::now();
  long spins = 0;
  while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {
    if ((++spins & 0xFFF) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) {
      if (hipStreamSynchronize(s) != hipSuccess) return WISKI_E_LAUNCH;   // fallback: should not happen
      if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) return WISKI_E_LAUNCH;
      break;
    }
  }
  return WISKI_OK;
}

__global__ void k_publish_flag(const int32_t* __restrict__ flag, double* __restrict__ poll, long long seq) {
  poll[1] = (double)*flag;
  __threadfence_system();
  __hip_atomic_store(reinterpret_cast<long long*>(poll), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Value of a device int32 flag once everything queued on `stream` before this call has run, without a stream
// synchronisation (the caller's out-of-grid flag after a gather: models/batched_fixed_noise_online_gp.py).
extern "C" int wiski_read_flag(const int32_t* d_flag, int32_t* h_value, void* stream) {
  if (!d_flag || !h_value) return WISKI_E_BADARG;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return WISKI_E_LAUNCH;
  std::lock_guard<std::mutex> lock(g_poll_mu);
  WiskiPoll& P = g_polls[dev];
  if (poll_reserve(P, 1) != WISKI_OK) return WISKI_E_LAUNCH;
  const long long seq = ++P.seq;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_publish_flag, dim3(1), dim3(1), 0, s, d_flag, P.d, seq);
  if (hipGetLastError() != hipSuccess) return WISKI_E_LAUNCH;
  if (int rc = poll_wait(P, seq, s)) return rc;
  *h_value = (int32_t)P.h[1];
  return WISKI_OK;
}

__global__ void k_pcg_publish(PcgScal S, int slot, double* __restrict__ poll, const int32_t* __restrict__ err, long long seq, double tol2,
                              long long* __restrict__ guard) {
  const int k = S.k;
  int ok = 1;
  for (int c = threadIdx.x; c < k; c += blockDim.x) {
    const double r0 = S.rn0()[c], r1 = S.rn(slot)[c];
    poll[1 + c] = r0;
    poll[1 + k + c] = r1;
    if (r0 > 0 && r1 > tol2 * r0) ok = 0;        // the host's convergence test, on the same doubles
  }
  ok = __syncthreads_and(ok);
  if (threadIdx.x == 0) {
    const double e = err ? (double)*err : 0.0;
    const long long gv = (ok && e == 0.0) ? seq : -seq;
    poll[1 + 2 * k] = e;
    poll[2 + 2 * k] = (double)gv;
    if (guard) *guard = gv;
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(reinterpret_cast<long long*>(poll), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// One launch instead of two hipMemsetAsync (each costs ~5 us of stream time): zero the scalar block and,
// on the half-stencil path, the atomically accumulated partial vector.
template <typename real>
__global__ __launch_bounds__(256) void k_pcg_zero(double* __restrict__ scal, int64_t nscal, real* __restrict__ vec, int64_t nvec) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = i; e < nscal; e += stride) scal[e] = 0.0;
  for (int64_t e = i; e < nvec; e += stride) vec[e] = (real)0;
}

// r = rhs - t - sum_ch part[ch] (t / part may be NULL); rn0 += rhs^2 ; rn(0) += r^2
// VEC = 4: 16-byte accesses (requires m % 4 == 0), all loads of a thread independent.
// given != 0: r already holds the residual (carried over by the caller, warm = 2); only the norms are formed.
template <typename real, int VEC>
__global__ __launch_bounds__(256) void k_pcg_init(int m, const real* __restrict__ rhs, const real* __restrict__ t, real* __restrict__ part,
                                                  int nch, int zl, int given, real* __restrict__ r, PcgScal S) {
  __shared__ double s_red[16];
  const int c = blockIdx.y;
  const int64_t km = (int64_t)S.k * m;
  double a = 0, bsum = 0;
  for (int i = (blockIdx.x * blockDim.x + threadIdx.x) * VEC; i < m; i += gridDim.x * blockDim.x * VEC) {
    const int64_t e = (int64_t)c * m + i;
    real f[VEC], rr[VEC];
    if constexpr (VEC == 4) {
      const Vec4<real> f4 = load4<real>(rhs + e);
      f[0] = f4.x; f[1] = f4.y; f[2] = f4.z; f[3] = f4.w;
      if (given) {
        const Vec4<real> r4 = load4<real>(r + e);
        rr[0] = r4.x; rr[1] = r4.y; rr[2] = r4.z; rr[3] = r4.w;
      } else {
      Vec4<real> t4;
      t4.x = t4.y = t4.z = t4.w = (real)0;
      if (t) t4 = load4<real>(t + e);
      Vec4<real> pp[8];
#pragma unroll
      for (int ch = 0; ch < 8; ++ch)
        if (ch < nch) pp[ch] = load4<real>(part + (int64_t)ch * km + e);
      if (zl) store4<real>(part + (int64_t)(nch - 1) * km + e, (real)0, (real)0, (real)0, (real)0);
      rr[0] = f[0] - t4.x; rr[1] = f[1] - t4.y; rr[2] = f[2] - t4.z; rr[3] = f[3] - t4.w;
#pragma unroll
      for (int ch = 0; ch < 8; ++ch)
        if (ch < nch) { rr[0] -= pp[ch].x; rr[1] -= pp[ch].y; rr[2] -= pp[ch].z; rr[3] -= pp[ch].w; }
      store4<real>(r + e, rr[0], rr[1], rr[2], rr[3]);
      }
    } else {
      f[0] = rhs[e];
      if (given) {
        rr[0] = r[e];
      } else {
        rr[0] = t ? f[0] - t[e] : f[0];
        for (int ch = 0; ch < nch; ++ch) rr[0] -= part[(int64_t)ch * km + e];
        if (zl) part[(int64_t)(nch - 1) * km + e] = (real)0;
        r[e] = rr[0];
      }
    }
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
      a += (double)f[q] * f[q];
      bsum += (double)rr[q] * rr[q];
    }
  }
  a = block_reduce_sum(a, s_red);
  bsum = block_reduce_sum(bsum, s_red);
  if (threadIdx.x == 0) {
    unsafeAtomicAdd(S.rn0() + c, a);
    unsafeAtomicAdd(S.rn(0) + c, bsum);
  }
}


// p = y + beta p ; pt = t + beta pt ; beta = rho(it)/rho(it-1)   (t = Kt^-1 y: r for
// the plain Kt preconditioner, (I + a Kt)^-1 r for the spectral one)
// VEC = 4: 16-byte accesses (requires m % 4 == 0).
template <typename real, int VEC>
__global__ __launch_bounds__(256) void k_pcg_update_p(int m, int it, double tol2, const real* __restrict__ y, const real* __restrict__ r,
                                                      real* __restrict__ p, real* __restrict__ pt, PcgScal S) {
  const int c = blockIdx.y;
  double beta = 0;
  if (it > 0) {
    const double den = S.rho(it - 1)[c];
    beta = den > 0 ? S.rho(it)[c] / den : 0;
  }
  const real bt = (real)beta;
  const int64_t cm = (int64_t)c * m;
  for (int i = (blockIdx.x * blockDim.x + threadIdx.x) * VEC; i < m; i += gridDim.x * blockDim.x * VEC) {
    real yv[VEC], rv[VEC], pv[VEC], ptv[VEC];
    if constexpr (VEC == 4) {
      const Vec4<real> a = load4<real>(y + cm + i), b = load4<real>(r + cm + i);
      yv[0] = a.x; yv[1] = a.y; yv[2] = a.z; yv[3] = a.w;
      rv[0] = b.x; rv[1] = b.y; rv[2] = b.z; rv[3] = b.w;
      if (it > 0) {
        const Vec4<real> cpp = load4<real>(p + cm + i), dpp = load4<real>(pt + cm + i);
        pv[0] = cpp.x; pv[1] = cpp.y; pv[2] = cpp.z; pv[3] = cpp.w;
        ptv[0] = dpp.x; ptv[1] = dpp.y; ptv[2] = dpp.z; ptv[3] = dpp.w;
      }
    } else {
      yv[0] = y[cm + i]; rv[0] = r[cm + i];
      if (it > 0) { pv[0] = p[cm + i]; ptv[0] = pt[cm + i]; }
    }
#pragma unroll
    for (int u = 0; u < VEC; ++u) {
      pv[u] = it == 0 ? yv[u] : yv[u] + bt * pv[u];
      ptv[u] = it == 0 ? rv[u] : rv[u] + bt * ptv[u];
    }
    if constexpr (VEC == 4) {
      store4<real>(p + cm + i, pv[0], pv[1], pv[2], pv[3]);
      store4<real>(pt + cm + i, ptv[0], ptv[1], ptv[2], ptv[3]);
    } else {
      p[cm + i] = pv[0]; pt[cm + i] = ptv[0];
    }
  }
}

// alpha = rho/php ; u += alpha p ; z += alpha pt ; r -= alpha hp ; rn(it+1) += r^2
// nch > 0: hp is not materialised, hp = pt + sum_ch part[ch]  (wide SpMV path).
// VEC = 4: 16-byte accesses (requires m % 4 == 0); all partial loads are independent.
template <typename real, int VEC>
__global__ __launch_bounds__(256) void k_pcg_update_x(int m, int it, double tol2, const real* __restrict__ p, const real* __restrict__ pt,
                                                      const real* __restrict__ hp, real* __restrict__ part, int nch, int zl,
                                                      real* __restrict__ u, real* __restrict__ z, real* __restrict__ r, PcgScal S,
                                                      double* __restrict__ poll = nullptr, const int32_t* __restrict__ err = nullptr,
                                                      long long seq = 0, long long* __restrict__ guard = nullptr) {
  __shared__ double s_red[16];
  const int c = blockIdx.y;
  double alpha = 0;
  const double den = S.php_sum(it, c);
  if (blockIdx.x == 0) pcg_dot_clear(S.php(it + 1), c, 1, S.k);   // ring entry of the next SpMV (see PcgScal)
  if (pcg_active(S, it, c, tol2) && den > 0) alpha = S.rho(it)[c] / den;
  const real al = (real)alpha;
  double acc = 0;
  const int64_t cm = (int64_t)c * m;
  const int64_t km = (int64_t)S.k * m;
  for (int i = (blockIdx.x * blockDim.x + threadIdx.x) * VEC; i < m; i += gridDim.x * blockDim.x * VEC) {
    const int64_t e = cm + i;
    real pv[VEC], ptv[VEC], hv[VEC], uv[VEC], zv[VEC], rv[VEC];
    if constexpr (VEC == 4) {
      const Vec4<real> a = load4<real>(p + e), b = load4<real>(pt + e), cu = load4<real>(u + e), cz = load4<real>(z + e), cr = load4<real>(r + e);
      pv[0] = a.x; pv[1] = a.y; pv[2] = a.z; pv[3] = a.w;
      ptv[0] = b.x; ptv[1] = b.y; ptv[2] = b.z; ptv[3] = b.w;
      uv[0] = cu.x; uv[1] = cu.y; uv[2] = cu.z; uv[3] = cu.w;
      zv[0] = cz.x; zv[1] = cz.y; zv[2] = cz.z; zv[3] = cz.w;
      rv[0] = cr.x; rv[1] = cr.y; rv[2] = cr.z; rv[3] = cr.w;
      if (nch > 0) {
        Vec4<real> pp[8];
#pragma unroll
        for (int ch = 0; ch < 8; ++ch)
          if (ch < nch) pp[ch] = load4<real>(part + (int64_t)ch * km + e);
        if (zl) store4<real>(part + (int64_t)(nch - 1) * km + e, (real)0, (real)0, (real)0, (real)0);
#pragma unroll
        for (int q = 0; q < 4; ++q) hv[q] = ptv[q];
#pragma unroll
        for (int ch = 0; ch < 8; ++ch)
          if (ch < nch) { hv[0] += pp[ch].x; hv[1] += pp[ch].y; hv[2] += pp[ch].z; hv[3] += pp[ch].w; }
      } else {
        const Vec4<real> h4 = load4<real>(hp + e);
        hv[0] = h4.x; hv[1] = h4.y; hv[2] = h4.z; hv[3] = h4.w;
      }
    } else {
      pv[0] = p[e]; ptv[0] = pt[e]; uv[0] = u[e]; zv[0] = z[e]; rv[0] = r[e];
      if (nch > 0) {
        hv[0] = ptv[0];
        for (int ch = 0; ch < nch; ++ch) hv[0] += part[(int64_t)ch * km + e];
        if (zl) part[(int64_t)(nch - 1) * km + e] = (real)0;
      } else {
        hv[0] = hp[e];
      }
    }
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
      uv[q] += al * pv[q];
      zv[q] += al * ptv[q];
      rv[q] -= al * hv[q];
      acc += (double)rv[q] * rv[q];
    }
    if constexpr (VEC == 4) {
      store4<real>(u + e, uv[0], uv[1], uv[2], uv[3]);
      store4<real>(z + e, zv[0], zv[1], zv[2], zv[3]);
      store4<real>(r + e, rv[0], rv[1], rv[2], rv[3]);
    } else {
      u[e] = uv[0]; z[e] = zv[0]; r[e] = rv[0];
    }
  }
  acc = block_reduce_sum(acc, s_red);
  if (poll == nullptr) {
    if (threadIdx.x == 0) unsafeAtomicAdd(S.rn(it + 1) + c, acc);
  } else {
    // Convergence poll folded into the update (one launch less per poll): the block that draws the last ticket knows every
    // block's ||r||^2 contribution has been performed at the memory side and publishes what k_pcg_publish would, reading the
    // norms around L2.  The contribution is a RETURNING atomic and the ticket is issued only once its result is back -- a
    // release fence here would write back the L2 lines of u, z, r the block has just dirtied (k = 64: 34 -> 113 us).
    __shared__ int s_last;
    if (threadIdx.x == 0) {
      const double prev = atomicAdd(S.rn(it + 1) + c, acc);
      unsigned one = 1u;
      asm volatile("" : "+v"(one) : "v"(prev));            // orders the ticket after the arrival of `prev`
      const unsigned t = atomicAdd(S.ticket(), one);
      s_last = t == gridDim.x * gridDim.y - 1;
    }
    __syncthreads();
    if (s_last) {
      const int k = S.k;
      int ok = 1;
      for (int c2 = threadIdx.x; c2 < k; c2 += blockDim.x) {
        const double r0 = __hip_atomic_load(S.rn0() + c2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double r1 = __hip_atomic_load(S.rn(it + 1) + c2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        poll[1 + c2] = r0;
        poll[1 + k + c2] = r1;
        if (r0 > 0 && r1 > tol2 * r0) ok = 0;    // the host's convergence test, on the same doubles
      }
      ok = __syncthreads_and(ok);                // s_last is block-uniform: every thread of the block is here
      if (threadIdx.x == 0) {
        // the guard of a speculatively queued absorb (wiski_scatter_stats_step): +seq iff the host will find this poll
        // converged and error-free
        const double e = err ? (double)*err : 0.0;
        const long long gv = (ok && e == 0.0) ? seq : -seq;
        poll[1 + 2 * k] = e;
        poll[2 + 2 * k] = (double)gv;
        if (guard) *guard = gv;
        __hip_atomic_store(S.ticket(), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // ready for the next poll of this solve
      }
      __threadfence_system();
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_store(reinterpret_cast<long long*>(poll), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// Stencil-sharded solve: this rank's share of A p = its direct partials + its accumulated (transposed-term) partial, summed
// into ONE vector that is then all-reduced over the ranks; the accumulated partial is re-zeroed on the way.
template <typename real>
__global__ __launch_bounds__(256) void k_shard_reduce(int m4, int m, const real* __restrict__ part, int np, real* __restrict__ acc,
                                                      real* __restrict__ out) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m4; i += gridDim.x * blockDim.x) {
    Vec4<real> t = load4<real>(acc + 4 * i);
    for (int p = 0; p < np; ++p) {
      const Vec4<real> v = load4<real>(part + (int64_t)p * m + 4 * i);
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    store4<real>(out + 4 * i, t.x, t.y, t.z, t.w);
    store4<real>(acc + 4 * i, (real)0, (real)0, (real)0, (real)0);
  }
}

static inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

static int64_t pcg_ws_bytes(int m, int k, int max_iter, int es) {
  int64_t vec = align_up((int64_t)k * m * es, 256);
  int64_t scal = align_up(PcgScal::doubles(k, max_iter) * 8, 256);
  return (6 + 8 + 6) * vec + scal;
}

template <typename real>
static int pcg_impl(const wiski_grid* grid, const real* d_A, const real* d_tcol, real kscale, const real* d_evec, const real* d_evec2, const real* d_eval,
                    real shift, const real* d_RHS, int32_t k, real* d_U, real* d_Z, int32_t warm, double tol, int32_t max_iter,
                    int32_t check_every, int32_t first_check, void* d_work, int64_t work_bytes, int32_t* h_iters, double* h_relres,
                    const int32_t* d_err, int32_t* h_err, int32_t a_sym, real* d_R, void* stream, wiski_pcg_async* as = nullptr,
                    int32_t amode = 0, const wiski_shard* shard = nullptr, const wiski_twolevel* two_level = nullptr) {
  // shard (wiski_shard, nranks > 1): d_A holds only this rank's groups of the half stencil; every A . v product is this rank's
  // share, summed over the ranks by ONE all-reduce of an m-vector (+ the p . Ap slots) per product on the solve's stream.
  // Everything else -- preconditioner, vector updates, scalars -- is replicated, so all ranks take identical iterations.
  // amode (with `as`): 0 = run to convergence; 1 = START: queue the iterations up to the first convergence poll, queue the
  // poll and return WISKI_PENDING without waiting for it; 2 = RESUME a started solve (same arguments): wait for that poll
  // (normally long over), finish with synchronous polls if it was not converged.  Nothing but the resume call may use
  // (U, Z, R, workspace) in between.
  GridDev<real> G;
  int rc = make_grid_dev<real>(grid, &G);
  if (rc) return rc;
  if (!d_A || !d_RHS || !d_U || !d_Z || !d_work || k < 1 || max_iter < 1) return WISKI_E_BADARG;
  if (amode != 0 && !as) return WISKI_E_BADARG;
  const bool resume = amode == 2;
  if (resume && as->state != 1) return WISKI_E_BADARG;
  const bool spectral = d_evec != nullptr && d_eval != nullptr;
  if (!spectral && !d_tcol) return WISKI_E_BADARG;
  if (work_bytes < pcg_ws_bytes(G.m, k, max_iter, (int)sizeof(real))) return WISKI_E_WORKSPACE;
  if (check_every < 1) check_every = 10;
  hipStream_t s = (hipStream_t)stream;
  const int m = G.m;
  const int64_t vec = align_up((int64_t)k * m * sizeof(real), 256);
  char* w = (char*)d_work;
  real* r = d_R ? d_R : (real*)(w + 0 * vec);   // caller-owned residual: survives the call (carry-over for warm = 2)
  real* y = (real*)(w + 1 * vec);
  real* p = (real*)(w + 2 * vec);
  real* pt = (real*)(w + 3 * vec);
  real* hp = (real*)(w + 4 * vec);
  real* tmp = (real*)(w + 5 * vec);
  real* part = (real*)(w + 6 * vec);
  real* ty = (real*)(w + 14 * vec);
  real* sa = (real*)(w + 16 * vec);
  real* sb = (real*)(w + 18 * vec);
  PcgScal S{(double*)(w + 20 * vec), k, (double*)(w + 20 * vec) + PcgScal::scalars(k, max_iter)};
  const bool wide = (m % 4) == 0;
  const bool sym = a_sym != 0;
  // number of partial vectors the wide SpMV leaves behind; on the half stencil the last one is the
  // atomically accumulated transposed term: zero here, re-zeroed by every consumer (zl)
  int zl = 0;
  const int nch = wide ? (sym ? sym_partials<real>(G, k, &zl) : spmv_nch(G.d)) : 0;
  const bool sharded = wiski_shard_active(shard);
  SymDmaParts stab{};
  int sh_lo = 0, sh_hi = 0;                      // this replica's groups (stencil-sharded solves)
  bool sh_dma = false;                           // its products run on the LDS-DMA kernel (part table) / on the LDS-window kernel (group range)
  if (sharded) {
    if (!sym || !wide || k != 1 || shard->rank < 0 || shard->rank >= shard->nranks) return WISKI_E_BADARG;
    if (!shard->comm && !shard->allreduce) return WISKI_E_BADARG;
    shard_group_range(sym_groups(G.d), shard->rank, shard->nranks, &sh_lo, &sh_hi);   // may be empty (more ranks than groups): zeros
    sh_dma = sym_use_dma<real>(G, k);
    if (sh_dma) shard_parts_d3(sh_lo, sh_hi, &stab);
  }
  // what the consumers of a product read: the partial vectors of the local launch, or the all-reduced sum
  real* cpart = sharded ? hp : part;
  const int cnch = sharded ? 1 : nch, czl = sharded ? 0 : zl;
  auto spmv_wide = [&](const real* v, const real* add, real beta, double* dots) -> int {
    if (!sharded)
      return sym ? launch_spmv4_sym<real>(G, d_A, v, k, part, add, beta, dots, s) : launch_spmv4<real>(G, d_A, v, k, part, add, beta, dots, s);
    int np = 0;                                   // partial vectors this rank's launch wrote
    if (sh_dma) {
      if (stab.n) {
        if (int r1 = launch_spmv4_sym<real>(G, d_A, v, k, part, add, beta, dots, s, &stab)) return r1;
      }
      np = stab.n;
    } else if (sh_hi > sh_lo) {
      if (int r1 = launch_spmv4_sym<real>(G, d_A, v, k, part, add, beta, dots, s, nullptr, sh_lo, sh_hi)) return r1;
      np = nch - 1;
    }
    const int m4 = m / 4;
    hipLaunchKernelGGL((k_shard_reduce<real>), dim3((unsigned)((m4 + 255) / 256)), dim3(256), 0, s, m4, m, (const real*)part, np,
                       part + (int64_t)(nch - 1) * k * m, hp);
    if (hipGetLastError() != hipSuccess) return WISKI_E_LAUNCH;
    const int64_t nd = dots ? (int64_t)k * PCG_DOT_COL : 0;
    if (shard->comm) {
      if constexpr (sizeof(real) == 4) return wiski_allreduce_stats_f32(shard->comm, nullptr, 0, (float*)hp, m, nullptr, 0, dots, nd, s);
      else return wiski_allreduce_stats_f64(shard->comm, nullptr, 0, (double*)hp, m, nullptr, 0, dots, nd, s);
    }
    return shard->allreduce(shard->ctx, hp, (int64_t)m, (int32_t)sizeof(real), dots, nd, s);
  };
  auto spmv_narrow = [&](const real* v, const real* add, real beta, real* out, double* dots) {
    return sym ? launch_spmv_sym<real>(G, d_A, v, k, add, beta, out, dots, s) : launch_spmv<real>(G, d_A, v, k, add, beta, out, dots, s);
  };
  if (!resume && as && as->prezeroed) {
    as->prezeroed = 0;      // wiski_stream_step had the step's first kernel zero both regions (wiski_pcg_zero_regions)
  } else if (!resume) {
    const int64_t nscal = PcgScal::doubles(k, max_iter), nvec = zl ? (int64_t)k * m : 0;
    int64_t zb = ((nscal > nvec ? nscal : nvec) + 255) / 256;
    if (zb > 1024) zb = 1024;
    hipLaunchKernelGGL((k_pcg_zero<real>), dim3((unsigned)zb), dim3(256), 0, s, S.base, nscal, part + (int64_t)(nch > 0 ? nch - 1 : 0) * k * m, nvec);
  }
  const double tol2 = tol * tol;
  // Blocks per column of the vector kernels (grid-stride loops).  Every block ends with one fp64 atomic per norm on the
  // column's scalar, the scalars of 16 columns share a 128-byte line, and same-line atomics serialise at the memory side
  // (~12 ns each): 489 blocks x 64 columns made k_pcg_init 138 us on a 32 MB sweep.  m / 4096 blocks, 32..256.
  int bcap = m / 4096;
  bcap = bcap < 32 ? 32 : (bcap > 256 ? 256 : bcap);
  if (k < 16 && bcap < 1024 / k) bcap = 1024 / k;     // few columns: the sweep wants the parallelism more than the atomics cost
  int eb = (m + 255) / 256;
  if (eb > bcap) eb = bcap;
  dim3 egrid((unsigned)eb, (unsigned)k);
  int vb = (m / 4 + 255) / 256;
  if (vb < 1) vb = 1;
  if (vb > bcap) vb = bcap;
  dim3 vgrid((unsigned)vb, (unsigned)k);   // 16-byte groups, grid-stride

  // fp32 fused path with a carried residual: the norms are formed by the first forward transform of r (apply = 2 below)
  const bool init_in_fwd = !resume && warm == 2 && sizeof(real) == 4 && spectral && wide && spectral_fused_ok<real>(G);
  if (resume) {
    // the start call queued everything up to (and including) the poll of iteration as->it
  } else if (warm == 2 && init_in_fwd) {
    if (!d_R) return WISKI_E_BADARG;
  } else if (warm == 2) {
    // r0 carried over by the caller in d_R (wiski_scatter_stats_cnt's d_res): no A u product
    if (!d_R) return WISKI_E_BADARG;
    if (wide) hipLaunchKernelGGL((k_pcg_init<real, 4>), vgrid, dim3(256), 0, s, m, d_RHS, (const real*)nullptr, (real*)nullptr, 0, 0, 1, r, S);
    else hipLaunchKernelGGL((k_pcg_init<real, 1>), egrid, dim3(256), 0, s, m, d_RHS, (const real*)nullptr, (real*)nullptr, 0, 0, 1, r, S);
  } else if (warm) {
    // r0 = rhs - (z + A u)
    if (wide) {
      rc = spmv_wide(d_U, nullptr, (real)0, nullptr);
      if (rc) return rc;
      hipLaunchKernelGGL((k_pcg_init<real, 4>), vgrid, dim3(256), 0, s, m, d_RHS, (const real*)d_Z, cpart, cnch, czl, 0, r, S);
    } else {
      rc = spmv_narrow(d_U, d_Z, (real)1, hp, nullptr);
      if (rc) return rc;
      hipLaunchKernelGGL((k_pcg_init<real, 1>), egrid, dim3(256), 0, s, m, d_RHS, (const real*)hp, (real*)nullptr, 0, 0, 0, r, S);
    }
  } else {
    if (hipMemsetAsync(d_U, 0, (size_t)k * m * sizeof(real), s) != hipSuccess) return WISKI_E_LAUNCH;
    if (hipMemsetAsync(d_Z, 0, (size_t)k * m * sizeof(real), s) != hipSuccess) return WISKI_E_LAUNCH;
    hipLaunchKernelGGL((k_pcg_init<real, 1>), egrid, dim3(256), 0, s, m, d_RHS, (const real*)nullptr, (real*)nullptr, 0, 0, 0, r, S);
  }

  std::vector<double> h_rn0(k), h_rn(k);
  double err_seen = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return WISKI_E_LAUNCH;
  // a deferred solve owns a poll buffer of its own (nobody else may publish into it between start and resume)
  std::unique_lock<std::mutex> poll_lock(g_poll_mu, std::defer_lock);
  WiskiPoll* Pp = nullptr;
  if (as) {
    if (!as->poll) as->poll = new WiskiPoll();
    Pp = static_cast<WiskiPoll*>(as->poll);
  } else {
    poll_lock.lock();
    Pp = &g_polls[dev];
  }
  WiskiPoll& P = *Pp;
  if (poll_reserve(P, k) != WISKI_OK) return WISKI_E_LAUNCH;
  auto collect = [&]() {
    for (int c = 0; c < k; ++c) {
      h_rn0[c] = P.h[1 + c];
      h_rn[c] = P.h[1 + k + c];
    }
    err_seen = P.h[1 + 2 * k];
  };
  bool deferred = false;      // START mode: the first poll has been queued but not waited for
  // after_publish: the poll with sequence number `seq` (iteration slot `slot`) has been queued
  auto fetch = [&](int slot, long long queued_seq = 0) -> int {
    long long seq = queued_seq;
    if (!seq) {
      seq = ++P.seq;
      hipLaunchKernelGGL(k_pcg_publish, dim3(1), dim3(64), 0, s, S, slot, P.d, d_err, seq, tol2, P.g);
    }
    if (hipGetLastError() != hipSuccess) return WISKI_E_LAUNCH;
    if (amode == 1) {
      as->state = 1;
      as->it = slot;
      as->seq = seq;
      deferred = true;
      return WISKI_OK;
    }
    if (int prc = poll_wait(P, seq, s)) return prc;
    collect();
    return WISKI_OK;
  };
  auto converged = [&]() {
    for (int c = 0; c < k; ++c)
      if (h_rn0[c] > 0 && h_rn[c] > tol2 * h_rn0[c]) return false;
    return true;
  };

  int it = 0;
  bool done = false;
  if (resume) {
    it = as->it;
    as->state = 0;
    if (int prc = poll_wait(P, as->seq, s)) return prc;
    collect();
    as->guard_ok = (long long)P.h[2 + 2 * k] == as->seq ? 1 : 0;   // did an absorb guarded by this poll run? (wiski_pcg_async_guard)
    done = converged();
    first_check = it;          // from here on: a poll after every check_every-th further iteration
    amode = 0;
  }
  if (first_check < 1) first_check = check_every;
  if (first_check > max_iter) first_check = max_iter;
  // host convergence polls: after `first_check` iterations, then every `check_every`
  auto due = [&](int i) { return i == max_iter || i == first_check || (i > first_check && (i - first_check) % check_every == 0); };
  const bool fused_cg = spectral && wide && spectral_fused_ok<real>(G);
  if (two_level && !(fused_cg && sizeof(real) == 4)) return WISKI_E_BADARG;   // the exact block lives in the fused fp32 slab kernels (k > 1: launch_slab checks the scratch)
  bool pending = false;   // fused path: update_x of iteration it-1 not applied yet
  // publish: fold the convergence poll of iteration `it` into this update; returns its sequence number (0: nothing launched)
  auto flush_update = [&](bool publish = false) -> long long {
    long long seq = 0;
    if (pending) {
      if (publish) seq = ++P.seq;
      hipLaunchKernelGGL((k_pcg_update_x<real, 4>), vgrid, dim3(256), 0, s, m, it - 1, tol2, (const real*)p, (const real*)pt, (const real*)hp,
                         cpart, cnch, czl, d_U, d_Z, r, S, publish ? P.d : (double*)nullptr, d_err, seq, P.g);
      pending = false;
    }
    return seq;
  };
  while (!done && it < max_iter) {
    if (fused_cg) {
      // fp32: 4 launches per iteration: [update_x(it-1) + mode-0 fwd] -> slab (+rho) -> [mode-0 bwd + update_p] -> SpMV (+p.Hp)
      // fp64: the update is its own launch (5 per iteration)
      constexpr bool fuse_upd = sizeof(real) == 4;
      if (!fuse_upd) flush_update();
      const bool init_now = init_in_fwd && it == 0;
      rc = launch_spectral_fused_cg<real>(G, d_evec, d_evec2, d_eval, kscale, shift, r, k, sa, sb, it, pending ? 1 : 0, tol2, p, pt, cpart, cnch, czl,
                                          d_U, d_Z, S, s, init_now ? d_RHS : (const real*)nullptr, two_level);
      pending = false;
      if (rc) return rc;
      rc = spmv_wide(p, pt, (real)1, S.php(it));
      if (rc) return rc;
      pending = true;
      ++it;
      if (due(it)) {
        rc = fetch(it, flush_update(true));
        if (rc) return rc;
        if (deferred) return WISKI_PENDING;
        done = converged();
      }
      continue;
    }
    // y = Kt r, rho(it) = r.y
    if (spectral) {
      // [t | y] = spectral preconditioner applied to r, rho(it) = r.y
      rc = launch_spectral_precond<real>(G, d_evec, d_evec2, d_eval, kscale, shift, r, k, sa, sb, ty, S.rho(it), s);
      if (rc) return rc;
      if (wide)
        hipLaunchKernelGGL((k_pcg_update_p<real, 4>), vgrid, dim3(256), 0, s, m, it, tol2, (const real*)(ty + (int64_t)k * m), (const real*)ty, p,
                           pt, S);
      else
        hipLaunchKernelGGL((k_pcg_update_p<real, 1>), egrid, dim3(256), 0, s, m, it, tol2, (const real*)(ty + (int64_t)k * m), (const real*)ty, p,
                           pt, S);
    } else {
      // y = Kt r, rho(it) = r.y
      rc = launch_kron<real>(G, d_tcol, r, k, kscale, tmp, y, r, S.rho(it), s);
      if (rc) return rc;
      if (wide) hipLaunchKernelGGL((k_pcg_update_p<real, 4>), vgrid, dim3(256), 0, s, m, it, tol2, (const real*)y, (const real*)r, p, pt, S);
      else hipLaunchKernelGGL((k_pcg_update_p<real, 1>), egrid, dim3(256), 0, s, m, it, tol2, (const real*)y, (const real*)r, p, pt, S);
    }
    // hp = pt + A p, php(it) = p.hp
    if (wide) rc = spmv_wide(p, pt, (real)1, S.php(it));
    else rc = spmv_narrow(p, pt, (real)1, hp, S.php(it));
    if (rc) return rc;
    const bool pub = due(it + 1);
    const long long useq = pub ? ++P.seq : 0;
    if (wide)
      hipLaunchKernelGGL((k_pcg_update_x<real, 4>), vgrid, dim3(256), 0, s, m, it, tol2, (const real*)p, (const real*)pt, (const real*)hp,
                         cpart, cnch, czl, d_U, d_Z, r, S, pub ? P.d : (double*)nullptr, d_err, useq, P.g);
    else
      hipLaunchKernelGGL((k_pcg_update_x<real, 1>), egrid, dim3(256), 0, s, m, it, tol2, (const real*)p, (const real*)pt, (const real*)hp,
                         cpart, cnch, czl, d_U, d_Z, r, S, pub ? P.d : (double*)nullptr, d_err, useq, P.g);
    ++it;
    if (due(it)) {
      rc = fetch(it, useq);
      if (rc) return rc;
      if (deferred) return WISKI_PENDING;
      done = converged();
    }
  }
  flush_update();
  if (hipGetLastError() != hipSuccess) return WISKI_E_LAUNCH;
  if (it == 0) {
    rc = fetch(0);
    if (rc) return rc;
  }
  if (h_iters) *h_iters = it;
  if (h_err) *h_err = (int32_t)err_seen;   // raw flag word: bit 0 = any point outside the grid, bits 1.. = count of such training points
  if (h_relres)
    for (int c = 0; c < k; ++c) h_relres[c] = h_rn0[c] > 0 ? sqrt(h_rn[c] / h_rn0[c]) : 0.0;
  return done ? WISKI_OK : WISKI_E_NOTCONV;
}

template <typename real>
static int spmv_impl(const wiski_grid* grid, const real* d_A, const real* d_V, int32_t k, const real* d_add, real beta, real* d_out, void* stream) {
  GridDev<real> G;
  int rc = make_grid_dev<real>(grid, &G);
  if (rc) return rc;
  if (!d_A || !d_V || !d_out || k < 1) return WISKI_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  if (G.m % 4 != 0) return launch_spmv<real>(G, d_A, d_V, k, d_add, beta, d_out, nullptr, s);
  const int nch = spmv_nch(G.d);
  const int64_t km = (int64_t)k * G.m;
  real* part = nullptr;
  if (hipMallocAsync((void**)&part, (size_t)nch * km * sizeof(real), s) != hipSuccess) return WISKI_E_LAUNCH;  // stream-ordered scratch
  rc = launch_spmv4<real>(G, d_A, d_V, k, part, nullptr, (real)0, nullptr, s);
  if (rc == WISKI_OK) {
    int64_t blocks = (km + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL((k_spmv_reduce<real>), dim3((unsigned)blocks), dim3(256), 0, s, km, nch, part, d_add, beta, d_out, 0);
    if (hipGetLastError() != hipSuccess) rc = WISKI_E_LAUNCH;
  }
  (void)hipFreeAsync(part, s);
  return rc;
}

// Same product on the symmetric half stencil A_h [(R+1)/2][m].
template <typename real>
static int spmv_sym_impl(const wiski_grid* grid, const real* d_A, const real* d_V, int32_t k, const real* d_add, real beta, real* d_out, void* stream) {
  GridDev<real> G;
  int rc = make_grid_dev<real>(grid, &G);
  if (rc) return rc;
  if (!d_A || !d_V || !d_out || k < 1 || d_out == d_V) return WISKI_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  if (G.m % 4 != 0) return launch_spmv_sym<real>(G, d_A, d_V, k, d_add, beta, d_out, nullptr, s);
  int zl = 0;
  const int nch = sym_partials<real>(G, k, &zl);
  const int64_t km = (int64_t)k * G.m;
  real* part = nullptr;
  const int64_t slots = zl ? nch : 8;           // the many-column SpMM keeps its row-major copies behind part[0]
  if (hipMallocAsync((void**)&part, (size_t)slots * km * sizeof(real), s) != hipSuccess) return WISKI_E_LAUNCH;  // stream-ordered scratch
  if (zl && hipMemsetAsync(part + (int64_t)(nch - 1) * km, 0, (size_t)km * sizeof(real), s) != hipSuccess) rc = WISKI_E_LAUNCH;
#ifdef WISKI_PROBE_DOTS     /* timing builds: exercise the DOT variant of the kernels from tools/spmv_probe.py */
  static double* probe_dots = nullptr;
  if (!probe_dots) { (void)hipMalloc((void**)&probe_dots, sizeof(double) * 2 * 64 * PCG_DOT_COL); (void)hipMemset(probe_dots, 0, sizeof(double) * 2 * 64 * PCG_DOT_COL); }
  if (rc == WISKI_OK) rc = launch_spmv4_sym<real>(G, d_A, d_V, k, part, d_V, (real)1, k <= 64 ? probe_dots : nullptr, s);
#else
  if (rc == WISKI_OK) rc = launch_spmv4_sym<real>(G, d_A, d_V, k, part, nullptr, (real)0, nullptr, s);
#endif
  if (rc == WISKI_OK) {
    int64_t blocks = (km + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL((k_spmv_reduce<real>), dim3((unsigned)blocks), dim3(256), 0, s, km, nch, part, d_add, beta, d_out, 0);
    if (hipGetLastError() != hipSuccess) rc = WISKI_E_LAUNCH;
  }
  (void)hipFreeAsync(part, s);
  return rc;
}

template <typename real>
static int kron_impl(const wiski_grid* grid, const real* d_tcol, const real* d_V, int32_t k, real scale, real* d_tmp, real* d_out, void* stream) {
  GridDev<real> G;
  int rc = make_grid_dev<real>(grid, &G);
  if (rc) return rc;
  if (!d_tcol || !d_V || !d_out || k < 1 || d_out == d_V) return WISKI_E_BADARG;
  return launch_kron<real>(G, d_tcol, d_V, k, scale, d_tmp, d_out, nullptr, nullptr, (hipStream_t)stream);
}

// The two regions a solve zeroes before its first kernel (scalars + dot-slot ring; the atomically accumulated partial
// vector of the half-stencil SpMV), for a caller that has an earlier kernel do it: same layout arithmetic as pcg_impl.
template <typename real>
static int pcg_zero_regions_impl(const wiski_grid* grid, int32_t k, int32_t max_iter, void* d_work, int32_t a_sym, void** p1, int64_t* n1, void** p2,
                                 int64_t* n2) {
  GridDev<real> G;
  int rc = make_grid_dev<real>(grid, &G);
  if (rc) return rc;
  if (!d_work || !p1 || !n1 || !p2 || !n2 || k < 1 || max_iter < 1) return WISKI_E_BADARG;
  const int m = G.m;
  const int64_t vec = align_up((int64_t)k * m * sizeof(real), 256);
  char* w = (char*)d_work;
  real* part = (real*)(w + 6 * vec);
  int zl = 0;
  const bool wide = (m % 4) == 0;
  const int nch = wide ? (a_sym ? sym_partials<real>(G, k, &zl) : spmv_nch(G.d)) : 0;
  *p1 = w + 20 * vec;
  *n1 = PcgScal::doubles(k, max_iter) * (int64_t)sizeof(double);
  *p2 = part + (int64_t)(nch > 0 ? nch - 1 : 0) * k * m;
  *n2 = zl ? (int64_t)k * m * (int64_t)sizeof(real) : 0;
  return WISKI_OK;
}

extern "C" {
int wiski_pcg_zero_regions_f32(const wiski_grid* g, int32_t k, int32_t max_iter, void* work, int32_t a_sym, void** p1, int64_t* n1, void** p2, int64_t* n2) { return pcg_zero_regions_impl<float>(g, k, max_iter, work, a_sym, p1, n1, p2, n2); }
int wiski_pcg_zero_regions_f64(const wiski_grid* g, int32_t k, int32_t max_iter, void* work, int32_t a_sym, void** p1, int64_t* n1, void** p2, int64_t* n2) { return pcg_zero_regions_impl<double>(g, k, max_iter, work, a_sym, p1, n1, p2, n2); }
int wiski_stencil_spmv_f32(const wiski_grid* g, const float* A, const float* V, int32_t k, const float* add, float beta, float* out, void* s) { return spmv_impl<float>(g, A, V, k, add, beta, out, s); }
int wiski_stencil_spmv_f64(const wiski_grid* g, const double* A, const double* V, int32_t k, const double* add, double beta, double* out, void* s) { return spmv_impl<double>(g, A, V, k, add, beta, out, s); }
int wiski_stencil_spmv_sym_f32(const wiski_grid* g, const float* A, const float* V, int32_t k, const float* add, float beta, float* out, void* s) { return spmv_sym_impl<float>(g, A, V, k, add, beta, out, s); }
int wiski_stencil_spmv_sym_f64(const wiski_grid* g, const double* A, const double* V, int32_t k, const double* add, double beta, double* out, void* s) { return spmv_sym_impl<double>(g, A, V, k, add, beta, out, s); }
int wiski_kron_toeplitz_mm_f32(const wiski_grid* g, const float* tcol, const float* V, int32_t k, float scale, float* tmp, float* out, void* s) { return kron_impl<float>(g, tcol, V, k, scale, tmp, out, s); }
int wiski_kron_toeplitz_mm_f64(const wiski_grid* g, const double* tcol, const double* V, int32_t k, double scale, double* tmp, double* out, void* s) { return kron_impl<double>(g, tcol, V, k, scale, tmp, out, s); }
int wiski_kron_toeplitz_grad_f32(const wiski_grid* g, const float* tcol, const float* X, const float* Y, int32_t k, float* tmp, double* grad, void* s) { return kron_grad_impl<float>(g, tcol, X, Y, k, tmp, grad, s); }
int wiski_kron_toeplitz_grad_f64(const wiski_grid* g, const double* tcol, const double* X, const double* Y, int32_t k, double* tmp, double* grad, void* s) { return kron_grad_impl<double>(g, tcol, X, Y, k, tmp, grad, s); }
int wiski_kron_spectral_mm_f32(const wiski_grid* g, const float* evec, const float* eval, float kscale, float shift, float pw, float rw, const float* V, int32_t k, float* tmp, float* out, void* s) { return spectral_mm_impl<float>(g, evec, eval, kscale, shift, pw, rw, V, k, tmp, out, s); }
int wiski_kron_spectral_mm_f64(const wiski_grid* g, const double* evec, const double* eval, double kscale, double shift, double pw, double rw, const double* V, int32_t k, double* tmp, double* out, void* s) { return spectral_mm_impl<double>(g, evec, eval, kscale, shift, pw, rw, V, k, tmp, out, s); }
int64_t wiski_pcg_workspace_bytes(const wiski_grid* grid, int32_t k, int32_t max_iter, int32_t elem_size) {
  if (!grid || grid->d < 1 || grid->d > WISKI_MAX_DIM || k < 1 || max_iter < 1) return WISKI_E_BADARG;
  int64_t m = 1;
  for (int q = 0; q < grid->d; ++q) m *= grid->g[q];
  return pcg_ws_bytes((int)m, k, max_iter, elem_size);
}
int wiski_pcg_async_f32(const wiski_grid* g, const float* A, const float* tcol, float kscale, const float* evec, const float* evec2, const float* eval, float shift, const float* RHS, int32_t k, float* U, float* Z, int32_t warm, double tol, int32_t max_iter, int32_t check_every, int32_t first_check, void* work, int64_t wb, int32_t* iters, double* relres, const int32_t* d_err, int32_t* h_err, int32_t a_sym, float* R, void* s, wiski_pcg_async* as, int32_t amode) {
  return pcg_impl<float>(g, A, tcol, kscale, evec, evec2, eval, shift, RHS, k, U, Z, warm, tol, max_iter, check_every, first_check, work, wb, iters, relres, d_err, h_err, a_sym, R, s, as, amode);
}
int wiski_pcg_async_f64(const wiski_grid* g, const double* A, const double* tcol, double kscale, const double* evec, const double* evec2, const double* eval, double shift, const double* RHS, int32_t k, double* U, double* Z, int32_t warm, double tol, int32_t max_iter, int32_t check_every, int32_t first_check, void* work, int64_t wb, int32_t* iters, double* relres, const int32_t* d_err, int32_t* h_err, int32_t a_sym, double* R, void* s, wiski_pcg_async* as, int32_t amode) {
  return pcg_impl<double>(g, A, tcol, kscale, evec, evec2, eval, shift, RHS, k, U, Z, warm, tol, max_iter, check_every, first_check, work, wb, iters, relres, d_err, h_err, a_sym, R, s, as, amode);
}
int wiski_pcg_sharded_f32(const wiski_grid* g, const float* A, const float* tcol, float kscale, const float* evec, const float* evec2, const float* eval, float shift, const float* RHS, int32_t k, float* U, float* Z, int32_t warm, double tol, int32_t max_iter, int32_t check_every, int32_t first_check, void* work, int64_t wb, int32_t* iters, double* relres, const int32_t* d_err, int32_t* h_err, int32_t a_sym, float* R, void* s, wiski_pcg_async* as, int32_t amode, const wiski_shard* shard) {
  return pcg_impl<float>(g, A, tcol, kscale, evec, evec2, eval, shift, RHS, k, U, Z, warm, tol, max_iter, check_every, first_check, work, wb, iters, relres, d_err, h_err, a_sym, R, s, as, amode, shard);
}
int wiski_pcg_twolevel_f32(const wiski_grid* g, const float* A, const float* tcol, float kscale, const float* evec, const float* evec2, const float* eval, float shift, const float* RHS, int32_t k, float* U, float* Z, int32_t warm, double tol, int32_t max_iter, int32_t check_every, int32_t first_check, void* work, int64_t wb, int32_t* iters, double* relres, const int32_t* d_err, int32_t* h_err, int32_t a_sym, float* R, void* s, wiski_pcg_async* as, int32_t amode, const wiski_shard* shard, const wiski_twolevel* two_level) {
  return pcg_impl<float>(g, A, tcol, kscale, evec, evec2, eval, shift, RHS, k, U, Z, warm, tol, max_iter, check_every, first_check, work, wb, iters, relres, d_err, h_err, a_sym, R, s, as, amode, shard, two_level);
}
// One application of the fused preconditioner (iteration 0 of the CG front end without an update to fold in): p = y = P r,
// pt = t = Kt^-1 y, rho += r . y.
int wiski_precond_apply_f32(const wiski_grid* grid, const float* evec, const float* evec2, const float* eval, float kscale, float shift, const float* d_r, float* w0, float* w1, float* d_y, float* d_t, double* d_rho, const wiski_twolevel* two_level, void* stream) {
  GridDev<float> G;
  if (int rc = make_grid_dev<float>(grid, &G)) return rc;
  if (!evec || !eval || !d_r || !w0 || !w1 || !d_y || !d_t || !d_rho || !spectral_fused_ok<float>(G) || G.m % 4) return WISKI_E_BADARG;
  PcgScal S{d_rho - 1, 1, nullptr};          // rho(0) = base + k (1 + 2 * 0) = d_rho; nothing else of S is touched at it = 0, apply = 0
  return launch_spectral_fused_cg<float>(G, evec, evec2, eval, kscale, shift, const_cast<float*>(d_r), 1, w0, w1, 0, 0, 0.0, d_y, d_t, (float*)nullptr, 0, 0,
                                         (float*)nullptr, (float*)nullptr, S, (hipStream_t)stream, (const float*)nullptr, two_level);
}
// ... of k grid vectors at once (the multi-column kernels): d_r, d_y, d_t [k][m], w0 k m reals, w1 2 k m reals, d_rho [k] doubles (+=).
int wiski_precond_apply_cols_f32(const wiski_grid* grid, const float* evec, const float* evec2, const float* eval, float kscale, float shift, const float* d_r, int32_t k, float* w0, float* w1, float* d_y, float* d_t, double* d_rho, const wiski_twolevel* two_level, void* stream) {
  GridDev<float> G;
  if (int rc = make_grid_dev<float>(grid, &G)) return rc;
  if (!evec || !eval || !d_r || !w0 || !w1 || !d_y || !d_t || !d_rho || k < 1 || !spectral_fused_ok<float>(G) || G.m % 4) return WISKI_E_BADARG;
  PcgScal S{d_rho - k, k, nullptr};          // rho(0) = base + k
  return launch_spectral_fused_cg<float>(G, evec, evec2, eval, kscale, shift, const_cast<float*>(d_r), k, w0, w1, 0, 0, 0.0, d_y, d_t, (float*)nullptr, 0, 0,
                                         (float*)nullptr, (float*)nullptr, S, (hipStream_t)stream, (const float*)nullptr, two_level);
}
int wiski_pcg_sharded_f64(const wiski_grid* g, const double* A, const double* tcol, double kscale, const double* evec, const double* evec2, const double* eval, double shift, const double* RHS, int32_t k, double* U, double* Z, int32_t warm, double tol, int32_t max_iter, int32_t check_every, int32_t first_check, void* work, int64_t wb, int32_t* iters, double* relres, const int32_t* d_err, int32_t* h_err, int32_t a_sym, double* R, void* s, wiski_pcg_async* as, int32_t amode, const wiski_shard* shard) {
  return pcg_impl<double>(g, A, tcol, kscale, evec, evec2, eval, shift, RHS, k, U, Z, warm, tol, max_iter, check_every, first_check, work, wb, iters, relres, d_err, h_err, a_sym, R, s, as, amode, shard);
}
int wiski_shard_groups(int32_t d, int32_t rank, int32_t nranks, int32_t* g_lo, int32_t* g_hi) {
  if (d < 1 || d > WISKI_MAX_DIM || nranks < 1 || rank < 0 || rank >= nranks || !g_lo || !g_hi) return WISKI_E_BADARG;
  int lo, hi;
  shard_group_range(sym_groups(d), rank, nranks, &lo, &hi);
  *g_lo = lo;
  *g_hi = hi;
  return WISKI_OK;
}
int wiski_pcg_async_free(wiski_pcg_async* as) {
  if (!as) return WISKI_E_BADARG;
  if (as->poll) {
    WiskiPoll* P = static_cast<WiskiPoll*>(as->poll);
    if (P->h) (void)hipHostFree(P->h);
    if (P->g) (void)hipFree(P->g);
    delete P;
    as->poll = nullptr;
  }
  as->state = 0;
  return WISKI_OK;
}
int wiski_pcg_async_guard(const wiski_pcg_async* as, const void** d_guard, int64_t* expect) {
  if (!as || !d_guard || !expect || as->state != 1 || !as->poll) return WISKI_E_BADARG;
  const WiskiPoll* P = static_cast<const WiskiPoll*>(as->poll);
  if (!P->g) return WISKI_E_BADARG;
  *d_guard = P->g;
  *expect = as->seq;
  return WISKI_OK;
}
int wiski_pcg_f32(const wiski_grid* g, const float* A, const float* tcol, float kscale, const float* evec, const float* evec2, const float* eval, float shift, const float* RHS, int32_t k, float* U, float* Z, int32_t warm, double tol, int32_t max_iter, int32_t check_every, int32_t first_check, void* work, int64_t wb, int32_t* iters, double* relres, const int32_t* d_err, int32_t* h_err, int32_t a_sym, float* R, void* s) {
  return pcg_impl<float>(g, A, tcol, kscale, evec, evec2, eval, shift, RHS, k, U, Z, warm, tol, max_iter, check_every, first_check, work, wb, iters, relres, d_err, h_err, a_sym, R, s);
}
int wiski_pcg_f64(const wiski_grid* g, const double* A, const double* tcol, double kscale, const double* evec, const double* evec2, const double* eval, double shift, const double* RHS, int32_t k, double* U, double* Z, int32_t warm, double tol, int32_t max_iter, int32_t check_every, int32_t first_check, void* work, int64_t wb, int32_t* iters, double* relres, const int32_t* d_err, int32_t* h_err, int32_t a_sym, double* R, void* s) {
  return pcg_impl<double>(g, A, tcol, kscale, evec, evec2, eval, shift, RHS, k, U, Z, warm, tol, max_iter, check_every, first_check, work, wb, iters, relres, d_err, h_err, a_sym, R, s);
}
}
