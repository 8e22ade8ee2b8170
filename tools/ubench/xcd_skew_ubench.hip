// When do the workgroups of a dispatch start on each XCD?  A probe launch of 1984 one-wave workgroups (the LDS-DMA SpMV's grid, 20 KB of
// LDS each, ~12 us of sleep) behind a kernel that fills the chip; every wave stamps its start (100 MHz clock) and its XCC_ID.  Reported
// per logical XCD (workgroup id % 8) and per physical XCC: earliest / mean start after the dispatch's first wave, over 30 launches.
//   hipcc --offload-arch=gfx950 -O3 xcd_skew_ubench.hip -o /tmp/xcd_skew && /tmp/xcd_skew
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_fill(float* p, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = p[i] * 1.0001f + 1.f;
}
__global__ __launch_bounds__(64) void k_probe(unsigned long long* st, int hold_ticks) {
  extern __shared__ char lds[];
  const long long t0 = wall_clock64();
  if (threadIdx.x == 0) lds[0] = 1;
  while (wall_clock64() - t0 < hold_ticks) __builtin_amdgcn_s_sleep(16);
  if (threadIdx.x == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11)) & 0xfu;
    st[blockIdx.x] = ((unsigned long long)t0 & 0xffffffffffffull) | ((unsigned long long)xcc << 48);
  }
}
int main() {
  const int NWG = 1984, REPS = 30, N = 1 << 24;
  float* p; unsigned long long* st;
  hipMalloc(&p, N * sizeof(float)); hipMemset(p, 0, N * sizeof(float));
  hipMalloc(&st, NWG * sizeof(unsigned long long));
  hipFuncSetAttribute((const void*)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 20480);
  std::vector<unsigned long long> h(NWG);
  for (int mode = 0; mode < 2; ++mode) {
    double lo_l[8] = {0}, mean_l[8] = {0}, lo_p[8] = {0}, mean_p[8] = {0}; int cnt_p[8] = {0};
    int l2p[8][8] = {{0}};
    for (int r = 0; r < REPS + 3; ++r) {
      if (mode == 0) hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, p, N);
      else hipDeviceSynchronize();
      hipLaunchKernelGGL(k_probe, dim3(NWG), dim3(64), 20480, 0, st, 1200);
      hipDeviceSynchronize();
      if (r < 3) continue;
      hipMemcpy(h.data(), st, NWG * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      unsigned long long t0 = ~0ull;
      for (int i = 0; i < NWG; ++i) { unsigned long long t = h[i] & 0xffffffffffffull; if (t < t0) t0 = t; }
      double mn_l[8], sm_l[8] = {0}, mn_p[8], sm_p[8] = {0}; int n_l[8] = {0}, n_p[8] = {0};
      for (int x = 0; x < 8; ++x) mn_l[x] = mn_p[x] = 1e9;
      for (int i = 0; i < NWG; ++i) {
        const double t = (double)((h[i] & 0xffffffffffffull) - t0) * 1e-2; const int x = i & 7, px = (int)(h[i] >> 48) & 7;
        if (t < mn_l[x]) mn_l[x] = t; sm_l[x] += t; ++n_l[x];
        if (t < mn_p[px]) mn_p[px] = t; sm_p[px] += t; ++n_p[px];
        ++l2p[x][px];
      }
      for (int x = 0; x < 8; ++x) { lo_l[x] += mn_l[x] / REPS; mean_l[x] += sm_l[x] / n_l[x] / REPS; lo_p[x] += mn_p[x] / REPS; mean_p[x] += sm_p[x] / (n_p[x] ? n_p[x] : 1) / REPS; cnt_p[x] += n_p[x]; }
    }
    printf("== probe %s\n", mode == 0 ? "behind a chip-filling kernel" : "on an idle device");
    for (int x = 0; x < 8; ++x) {
      int best = 0; for (int q = 1; q < 8; ++q) if (l2p[x][q] > l2p[x][best]) best = q;
      printf("  logical XCD %d (-> XCC %d): first wave +%5.2f us, mean start +%5.2f us   |  physical XCC %d: first +%5.2f, mean +%5.2f (%d waves / launch)\n", x, best,
             lo_l[x], mean_l[x], x, lo_p[x], mean_p[x], cnt_p[x] / REPS);
    }
  }
  return 0;
}
