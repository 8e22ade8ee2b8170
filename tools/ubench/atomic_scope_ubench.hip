// Micro-benchmark: does any cache-policy variant of global_atomic_add_f32 run faster than the default on a large (86 MB)
// target, and does XCD-local addressing help?  (If atomics could be made to execute in the issuing XCD's L2, the statistics
// scatter could be restructured owner-computes by XCD.)  hipcc --offload-arch=gfx950 -O3 atomic_scope_ubench.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MOD>
__device__ __forceinline__ void atom(float* p, float v) {
  if (MOD == 0) asm volatile("global_atomic_add_f32 %0, %1, off" ::"v"(p), "v"(v) : "memory");
  else if (MOD == 1) asm volatile("global_atomic_add_f32 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else if (MOD == 2) asm volatile("global_atomic_add_f32 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
  else if (MOD == 3) asm volatile("global_atomic_add_f32 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
}
// runs of 8 consecutive floats at random 32-byte aligned places (the scatter's access shape); local != 0: the block only
// touches the eighth of the array that belongs to "its" XCD (blockIdx.x % 8)
template <int MOD>
__global__ __launch_bounds__(256) void k_atom(float* __restrict__ y, long long n, int per_thread, int local, float v) {
  const int lane8 = threadIdx.x & 7;
  unsigned s = ((blockIdx.x * 256 + threadIdx.x) >> 3) * 2654435761u + 12345u;
  const long long part = n / 8, base = local ? (long long)(blockIdx.x & 7) * part : 0, span = local ? part : n;
  for (int r = 0; r < per_thread; ++r) {
    s = s * 1664525u + 1013904223u;
    const long long j = base + ((long long)(s >> 4) % (span / 8)) * 8 + lane8;
    atom<MOD>(y + j, v);
  }
}

int main() {
  const long long n = 21500000 / 8 * 8;
  float* y;
  hipMalloc(&y, (size_t)n * 4);
  hipMemset(y, 0, (size_t)n * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const char* names[] = {"default", "sc1", "nt", "sc1 nt"};
  for (int local = 0; local < 2; ++local)
    for (int mod = 0; mod < 4; ++mod) {
      const int blocks = 2048, per_thread = 16, reps = 10;
      auto launch = [&]() {
        if (mod == 0) hipLaunchKernelGGL(k_atom<0>, dim3(blocks), dim3(256), 0, 0, y, n, per_thread, local, 1.0f);
        if (mod == 1) hipLaunchKernelGGL(k_atom<1>, dim3(blocks), dim3(256), 0, 0, y, n, per_thread, local, 1.0f);
        if (mod == 2) hipLaunchKernelGGL(k_atom<2>, dim3(blocks), dim3(256), 0, 0, y, n, per_thread, local, 1.0f);
        if (mod == 3) hipLaunchKernelGGL(k_atom<3>, dim3(blocks), dim3(256), 0, 0, y, n, per_thread, local, 1.0f);
      };
      launch();
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int i = 0; i < reps; ++i) launch();
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double ops = (double)blocks * 256 * per_thread * reps;
      printf("%-8s %s : %8.2f us/launch  %7.1f G lane-atomics/s  (%5.1f G 32-byte runs/s)\n", names[mod], local ? "XCD-local eighths" : "whole array      ",
             ms * 1e3 / reps, ops / (ms * 1e-3) / 1e9, ops / 8 / (ms * 1e-3) / 1e9);
    }
  return 0;
}
