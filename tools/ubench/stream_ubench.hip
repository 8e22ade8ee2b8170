// Micro-benchmark: what does a pure read of an 86 MB (half-stencil sized) buffer cost on MI355X, per launch, as the
// dispatch-packet timestamps see it (hipExtLaunchKernelGGL start/stop events = what rocprofv3 reports)?  Sets the
// practical ceiling of k_spmv_sym_dma's roofline fraction.  Variants: plain 16-byte loads (grid-stride) and the
// LDS-DMA ring of the SpMV without any arithmetic; buffer re-read every launch (MALL-warm) or with a 1 GB sweep in
// between (MALL-cold); an empty kernel gives the fixed launch cost.
//   hipcc --offload-arch=gfx950 -O3 stream_ubench.hip -o stream_ubench
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

__global__ void k_empty() {}

__global__ __launch_bounds__(256) void k_read16(const float4* __restrict__ a, size_t n16, float* __restrict__ sink) {
  float s = 0.f;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    const float4 v0 = a[i], v1 = a[i + stride], v2 = a[i + 2 * stride], v3 = a[i + 3 * stride];
    s += v0.x + v0.y + v0.z + v0.w + v1.x + v1.y + v1.z + v1.w + v2.x + v2.y + v2.z + v2.w + v3.x + v3.y + v3.z + v3.w;
  }
  for (; i < n16; i += stride) { const float4 v = a[i]; s += v.x + v.y + v.z + v.w; }
  if (s == 1234.5f) sink[0] = s;
}

__device__ __forceinline__ void glds_b128(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int MOD>
__device__ __forceinline__ void glds_b128_mod(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  if (MOD == 1)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
  else if (MOD == 2)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
  else if (MOD == 3)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc0 sc1\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc0 nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int MOD>
__global__ __launch_bounds__(64) void k_dma_ring_mod(const float* __restrict__ a, int ntile, size_t tile_stride, int nrb, float* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* st = reinterpret_cast<float*>(smem);
  const int lane = threadIdx.x;
  const unsigned sa = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)st);
  const int rb = blockIdx.x % nrb, part = blockIdx.x / nrb;
  const float* base = a + (size_t)part * ntile * tile_stride + (size_t)rb * 1792;
  auto issue = [&](int t) {
    const float* src = base + (size_t)t * tile_stride;
#pragma unroll
    for (int j = 0; j < 7; ++j) glds_b128_mod<MOD>(src + 4 * (64 * j + lane), sa + (unsigned)((t & 1) * 7168 + 1024 * j));
  };
  issue(0);
  if (ntile > 1) issue(1);
  float s = 0.f;
  for (int t = 0; t < ntile; ++t) {
    if (t + 1 < ntile) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const float* q = st + (t & 1) * 1792 + 28 * lane;
#pragma unroll
    for (int j = 0; j < 7; ++j) { const float4 v = *reinterpret_cast<const float4*>(q + 4 * j); s += v.x + v.y + v.z + v.w; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (t + 2 < ntile) issue(t + 2);
  }
  if (s == 1234.5f) sink[0] = s;
}

// one wave per block; each wave streams `ntile` 7 KiB tiles through a 2-deep LDS ring and reads them back (b128)
__global__ __launch_bounds__(64) void k_dma_ring(const float* __restrict__ a, int ntile, size_t tile_stride, int nrb, float* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* st = reinterpret_cast<float*>(smem);
  const int lane = threadIdx.x;
  const unsigned sa = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)st);
  // wave = (row block rb, part): its tile t is rows [256 rb, 256 rb + 256) of group part * ntile + t (a group = 7 m floats)
  const int rb = blockIdx.x % nrb, part = blockIdx.x / nrb;
  const float* base = a + (size_t)part * ntile * tile_stride + (size_t)rb * 1792;
  auto issue = [&](int t) {
    const float* src = base + (size_t)t * tile_stride;
#pragma unroll
    for (int j = 0; j < 7; ++j) glds_b128(src + 4 * (64 * j + lane), sa + (unsigned)((t & 1) * 7168 + 1024 * j));
  };
  issue(0);
  if (ntile > 1) issue(1);
  float s = 0.f;
  for (int t = 0; t < ntile; ++t) {
    if (t + 1 < ntile) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const float* q = st + (t & 1) * 1792 + 28 * lane;
#pragma unroll
    for (int j = 0; j < 7; ++j) { const float4 v = *reinterpret_cast<const float4*>(q + 4 * j); s += v.x + v.y + v.z + v.w; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (t + 2 < ntile) issue(t + 2);
  }
  if (s == 1234.5f) sink[0] = s;
}

static double med(std::vector<float>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
  const size_t m = 125000, H = 172;
  const size_t nfl = m * H;              // 86 MB
  float *a, *big, *sink;
  hipMalloc(&a, nfl * 4 + (1 << 20));
  hipMalloc(&big, (size_t)1 << 30);
  hipMalloc(&sink, 64);
  hipMemset(a, 0, nfl * 4 + (1 << 20));
  hipMemset(big, 0, (size_t)1 << 30);
  const int reps = 60;
  std::vector<hipEvent_t> ev(2 * reps);
  for (auto& e : ev) hipEventCreate(&e);
  auto run = [&](const char* name, bool cold, auto launch) {
    std::vector<float> t;
    for (int r = 0; r < reps; ++r) {
      if (cold) hipMemsetAsync(big, r & 1, (size_t)1 << 30, 0);
      launch(ev[2 * r], ev[2 * r + 1]);
    }
    hipDeviceSynchronize();
    for (int r = 5; r < reps; ++r) { float ms; hipEventElapsedTime(&ms, ev[2 * r], ev[2 * r + 1]); t.push_back(ms * 1e3f); }
    const double us = med(t);
    printf("%-44s %s : median %7.2f us  min %7.2f  -> %6.2f TB/s of 86 MB\n", name, cold ? "MALL-cold" : "MALL-warm", us, t[0], nfl * 4 / us / 1e6);
  };
  run("empty kernel (1 block)", false, [&](hipEvent_t s, hipEvent_t e) { hipExtLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0, s, e, 0); });
  run("empty kernel (2048 x 64)", false, [&](hipEvent_t s, hipEvent_t e) { hipExtLaunchKernelGGL(k_empty, dim3(2048), dim3(64), 0, 0, s, e, 0); });
  for (int cold = 0; cold < 1; ++cold) {
    for (int blocks : {1024, 2048, 4096}) {
      char nm[64]; snprintf(nm, 64, "plain 16 B loads, %d x 256 threads", blocks);
      run(nm, cold, [&](hipEvent_t s, hipEvent_t e) { hipExtLaunchKernelGGL(k_read16, dim3(blocks), dim3(256), 0, 0, s, e, 0, (const float4*)a, nfl / 4, sink); });
    }
    // DMA ring: 489 row blocks x P parts, each wave ntile tiles (tile stride = one group array = 7 m floats)
    for (int lds : {14336, 19968, 27136})
    for (int parts : {4, 6, 8, 24}) {
      const int nt = 24 / parts;            // 24 groups of 7 m floats = 84 MB in every variant
      char nm[80]; snprintf(nm, 80, "LDS-DMA ring, %d waves x %d tiles, %d B LDS", 489 * parts, nt, lds);
      run(nm, cold, [&](hipEvent_t s, hipEvent_t e) {
        hipExtLaunchKernelGGL(k_dma_ring, dim3(489 * parts), dim3(64), lds, 0, s, e, 0, (const float*)a, nt, (size_t)(7 * m), 489, sink);
      });
    }
  }
  // cache-policy modifiers on the LDS-DMA loads (2934 waves x 4 tiles)
#define MODRUN(M, NAME) run("LDS-DMA ring 2934 x 4, " NAME, false, [&](hipEvent_t s, hipEvent_t e) { \
    hipExtLaunchKernelGGL((k_dma_ring_mod<M>), dim3(489 * 6), dim3(64), 19968, 0, s, e, 0, (const float*)a, 4, (size_t)(7 * m), 489, sink); });
  MODRUN(1, "nt") MODRUN(2, "sc1") MODRUN(3, "sc0 sc1") MODRUN(4, "sc0 nt")
  // a read that does not fit the 256 MB Infinity Cache: the practical HBM ceiling (what k_gather_ell's 541 MB stream can hope for)
  for (int blocks : {2048, 4096, 8192}) {
    const size_t n16 = ((size_t)541 << 20) / 16;
    std::vector<float> t;
    for (int r = 0; r < 20; ++r) hipExtLaunchKernelGGL(k_read16, dim3(blocks), dim3(256), 0, 0, ev[2 * r], ev[2 * r + 1], 0, (const float4*)big, n16, sink);
    hipDeviceSynchronize();
    for (int r = 3; r < 20; ++r) { float ms; hipEventElapsedTime(&ms, ev[2 * r], ev[2 * r + 1]); t.push_back(ms * 1e3f); }
    const double us = med(t);
    printf("plain 16 B loads of 541 MiB, %d x 256 threads : median %7.2f us -> %5.2f TB/s\n", blocks, us, n16 * 16 / us / 1e6);
  }
  return 0;
}
