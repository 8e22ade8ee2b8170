// What the shader clock is while a ONE-workgroup kernel runs (the small dense kernels of the reference step: eigen-update, diagonal
// steps of the Cholesky): s_memtime (shader clock) against s_memrealtime (100 MHz) around a dependent FMA chain, (a) on an idle GPU,
// (b) right behind a kernel that fills the chip.   hipcc --offload-arch=gfx950 -O3 clock_ubench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_spin(double* out, long long* stamps, int iters) {
  double x = 1.0 + threadIdx.x * 1e-9;
  const long long w0 = wall_clock64(), c0 = clock64();
  for (int i = 0; i < iters; ++i) x = x * 1.0000001 + 1e-12;
  const long long w1 = wall_clock64(), c1 = clock64();
  if (threadIdx.x == 0) { stamps[0] = w1 - w0; stamps[1] = c1 - c0; }
  out[threadIdx.x] = x;
}
template <typename T>
__global__ void k_chain(T* out, long long* stamps, int iters) {       // dependent FMAs, unrolled 32x: cycles per dependent operation
  T x = (T)1 + (T)threadIdx.x * (T)1e-6;
  const T a = (T)1.0000001, b = (T)1e-7;
  const long long c0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 32; ++u) x = x * a + b;
  }
  const long long c1 = clock64();
  if (threadIdx.x == 0) stamps[0] = c1 - c0;
  out[threadIdx.x] = x;
}
template <typename T>
__global__ void k_indep(T* out, long long* stamps, int iters) {       // 8 independent chains: cycles per issued operation
  T x[8];
  for (int u = 0; u < 8; ++u) x[u] = (T)1 + (T)(threadIdx.x + u) * (T)1e-6;
  const T a = (T)1.0000001, b = (T)1e-7;
  const long long c0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = x[u] * a + b;
  }
  const long long c1 = clock64();
  if (threadIdx.x == 0) stamps[0] = c1 - c0;
  T s_ = 0;
  for (int u = 0; u < 8; ++u) s_ += x[u];
  out[threadIdx.x] = s_;
}
__global__ void k_fill(float* y, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) y[i] = y[i] * 1.0001f + 1.f;
}
int main() {
  double* out; long long* st; float* y;
  hipMalloc(&out, 64 * 8); hipMalloc(&st, 16); hipMalloc(&y, 1ll << 30);
  long long h[2];
  for (int rep = 0; rep < 3; ++rep) {
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, 0, out, st, 20000);
    hipMemcpy(h, st, 16, hipMemcpyDeviceToHost);
    printf("idle GPU        : %6.1f us wall, %9lld shader cycles -> %.0f MHz, %.1f cycles per dependent fp64 FMA\n", h[0] / 100.0, h[1], h[1] / (h[0] / 100.0), (double)h[1] / 20000);
    for (int k = 0; k < 20; ++k) hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, y, (1ll << 28));
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, 0, out, st, 20000);
    hipMemcpy(h, st, 16, hipMemcpyDeviceToHost);
    printf("behind full load: %6.1f us wall, %9lld shader cycles -> %.0f MHz, %.1f cycles per dependent fp64 FMA\n", h[0] / 100.0, h[1], h[1] / (h[0] / 100.0), (double)h[1] / 20000);
  }
  float* outf; hipMalloc(&outf, 64 * 4);
  hipLaunchKernelGGL((k_chain<double>), dim3(1), dim3(64), 0, 0, out, st, 1000); hipMemcpy(h, st, 8, hipMemcpyDeviceToHost);
  printf("dependent fp64 FMA : %.1f cycles each\n", h[0] / 32000.0);
  hipLaunchKernelGGL((k_indep<double>), dim3(1), dim3(64), 0, 0, out, st, 1000); hipMemcpy(h, st, 8, hipMemcpyDeviceToHost);
  printf("independent fp64 FMA (8 chains): %.1f cycles each\n", h[0] / 32000.0);
  hipLaunchKernelGGL((k_chain<float>), dim3(1), dim3(64), 0, 0, outf, st, 1000); hipMemcpy(h, st, 8, hipMemcpyDeviceToHost);
  printf("dependent fp32 FMA : %.1f cycles each\n", h[0] / 32000.0);
  hipLaunchKernelGGL((k_indep<float>), dim3(1), dim3(64), 0, 0, outf, st, 1000); hipMemcpy(h, st, 8, hipMemcpyDeviceToHost);
  printf("independent fp32 FMA (8 chains): %.1f cycles each\n", h[0] / 32000.0);
  return 0;
}
