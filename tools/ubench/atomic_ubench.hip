// Micro-benchmark: rate of fire-and-forget global fp32 atomics on MI355X for different
// address patterns (used to decide how a symmetric half-stencil SpMV may write its
// transposed contributions).  hipcc --offload-arch=gfx950 -O3 atomic_ubench.hip -o atomic_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// pattern 0: lane l of global thread t adds to y[(t*rep + r) % n]   -> fully coalesced waves (64 consecutive floats)
// pattern 1: each thread owns 4 consecutive rows (stride-4 lanes)    -> 4 atomics, lanes 16 B apart
// pattern 2: random rows
// pattern 3: coalesced, but windows of neighbouring blocks overlap by 25% (contention between blocks)
__global__ __launch_bounds__(256) void k_atomic(float* __restrict__ y, int n, int per_thread, int pattern, float v) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (pattern == 0) {
    for (int r = 0; r < per_thread; ++r) {
      int j = (int)(((long long)r * gridDim.x * blockDim.x + t) % n);
      unsafeAtomicAdd(y + j, v);
    }
  } else if (pattern == 1) {
    for (int r = 0; r < per_thread; r += 4) {
      long long b = ((long long)(r / 4) * gridDim.x * blockDim.x + t) * 4 % n;
#pragma unroll
      for (int e = 0; e < 4; ++e) unsafeAtomicAdd(y + b + e, v);
    }
  } else if (pattern == 2) {
    unsigned s = t * 2654435761u + 12345u;
    for (int r = 0; r < per_thread; ++r) {
      s = s * 1664525u + 1013904223u;
      unsafeAtomicAdd(y + (s >> 8) % n, v);
    }
  } else if (pattern == 3) {
    for (int r = 0; r < per_thread; ++r) {
      long long j = ((long long)r * gridDim.x + blockIdx.x) * 192 + threadIdx.x;
      unsafeAtomicAdd(y + j % n, v);
    }
  } else if (pattern == 4) {   // plain coalesced read-modify-write (no atomic) for reference
    for (int r = 0; r < per_thread; ++r) {
      int j = (int)(((long long)r * gridDim.x * blockDim.x + t) % n);
      y[j] += v;
    }
  }
}

int main() {
  const int n = 1 << 21;   // 8 MB target
  float* y;
  hipMalloc(&y, (size_t)n * 4);
  hipMemset(y, 0, (size_t)n * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const char* names[] = {"coalesced", "4-row threads", "random", "coalesced, overlapping blocks", "plain RMW (no atomic)"};
  for (int target : {1 << 17, 1 << 21}) {
    for (int pattern = 0; pattern < 5; ++pattern) {
      for (int blocks : {512, 2048}) {
        const int per_thread = 16;
        hipLaunchKernelGGL(k_atomic, dim3(blocks), dim3(256), 0, 0, y, target, per_thread, pattern, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        const int reps = 20;
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_atomic, dim3(blocks), dim3(256), 0, 0, y, target, per_thread, pattern, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double ops = (double)blocks * 256 * per_thread * reps;
        printf("target %7d floats  %-32s blocks %4d : %8.2f us/launch  %7.1f G lane-atomics/s\n", target, names[pattern], blocks, ms * 1e3 / reps,
               ops / (ms * 1e-3) / 1e9);
      }
    }
  }
  return 0;
}
