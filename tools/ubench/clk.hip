#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(long long* out, int n, float* sink) {
  long long c0 = clock64(), w0 = wall_clock64();
  float a = threadIdx.x;
  for (int i = 0; i < n; ++i) a = a * 1.0001f + 0.5f;
  long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
  sink[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
int main() {
  long long* d; float* s; hipMalloc(&d, 16); hipMalloc(&s, 1 << 24);
  for (int blocks : {1, 100, 256, 2048}) for (int n : {2000, 20000, 2000000}) {
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, n, s); hipDeviceSynchronize();
    long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("blocks %d n %d: cycles %lld wall-ticks %lld -> %.0f MHz (wall clock 100MHz), %.2f cyc/iter\n", blocks, n, h[0], h[1], h[0] / (h[1] / 100.0), (double)h[0] / n);
  }
}
