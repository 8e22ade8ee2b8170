// LDS-DMA primitives of gfx950 shared by the streaming kernels (spmv_sym_dma.h, spmv_sym_dma_mc.h, gather_ell_dma.h):
// `global_load_lds_*` copies 64 lanes x 4 / 16 bytes from per-lane global addresses straight into LDS at M0 + 4 / 16 * lane --
// no VGPR destination, so a wave can keep many KiB in flight at the cost of LDS space only.  The copies count on vmcnt like any
// vector-memory load and land in issue order; a kernel that uses them keeps its own count (`s_waitcnt vmcnt(N)`, N = the loads it
// has issued SINCE the one it needs) and must hide every other vector-memory load from the compiler, whose own bookkeeping does
// not see these and would drain the queue (vmcnt(0)) at its first load.
#pragma once

__device__ __forceinline__ void glds_b128(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}
// the A_h stream: read once per launch, so "sc0 nt" (stream through the XCD's L2 without keeping the line; the bare ring of
// tools/ubench/stream_ubench.hip moves 86 MB in 10.8 us with it against 12.1 us plain; nt / sc1 / sc0 sc1 alone: no change)
__device__ __forceinline__ void glds_b128_stream(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
#ifdef WISKI_DMA_PLAIN_STREAM
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
#else
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc0 nt\n\ts_mov_b32 m0, %0"
#endif
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}
__device__ __forceinline__ void glds_b32(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}
__device__ __forceinline__ void wave_lgkm_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)p; }
