// Symmetric half-stencil SpMM for a FEW right-hand sides (2 <= k < 32), LDS-DMA pipelined (d = 3, fp32).
// Included by solve.hip after spmv_sym_dma.h (shares its DMA helpers, window geometry and partial-vector contract).
//
// The k = 1 kernel (spmv_sym_dma.h) spends its time streaming A_h; its arithmetic per tile -- 56 FMAs and ~30 LDS words per
// lane -- is a small fraction.  So one pass over A_h can serve several columns: a wave still owns 256 rows x one part of the
// half stencil and streams the same 7 KB tiles through the same 2-deep LDS ring, but every tile, once in registers, is applied
// to KC columns: per column one 10-wide window of v (LDS, fetched once per part by LDS-DMA), 28 + 28 FMAs, and the
// lane-disjoint read-modify-write of that column's transposed-term window.  A_h is read ceil(k / KC) times instead of k
// times (the 4-columns-per-pass LDS-window kernel it replaces, k_stencil_spmv4_sym<4>, needs 140 us per pass: its v windows
// come from L2 with 10 loads per tile and column, and its register budget leaves one wave per SIMD).
//
// LDS per wave: the tile ring (14 KB) + KC x (own rows 1 KB + v window + transposed window, 2 x 2.3 KB at g2 = 50):
// 37 KB at KC = 4, i.e. 4 waves per CU.  KC = 2 (25 KB, 6 waves) serves k = 2, 3.
//
// Partial vectors as in the wide kernels: part[(ch * k + c) * m + i] = direct term of part ch, column c (plain stores),
// part[(nparts * k + c) * m + i] += transposed terms (atomics; zero on entry, re-zeroed by the consumer).
#pragma once

template <int KC, bool DOT>
__global__ __launch_bounds__(64) void k_spmv_sym_dma_mc(GridDev<float> G, const float* __restrict__ A_h, const float* __restrict__ V, int k, int W4,
                                                        int WP, float* __restrict__ part, const float* __restrict__ add, float beta,
                                                        double* __restrict__ dots, int xcd_rb) {
  constexpr int NST = 2, NPARTS = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x;
  const int m = G.m, S0 = G.stride[0], S1 = G.stride[1];
  const int c0 = blockIdx.z * KC;                  // first column of this pass
  const int y = blockIdx.y;
  const int d0 = (y + 1) & 3, p1lo = d0 == 0 ? 3 : 0, ntile = d0 == 0 ? 4 : 7;
  // xcd_rb != 0: XCD-contiguous row blocks (spmv_sym_dma.h)
  const int rb = xcd_rb ? (int)(blockIdx.x & 7) * xcd_rb + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  if (xcd_rb && rb * 256 >= G.m) return;
  const int iw0 = rb * 256;
  const int i4 = iw0 + 4 * lane;
  const bool live = i4 < m;
  const int nrows = m - iw0 < 256 ? m - iw0 : 256;
  const int VWF = symdma_vwf(S1);
  float* __restrict__ stage = reinterpret_cast<float*>(smem);   // [NST][SYMDMA_TILE]
  float* __restrict__ xo_l = stage + NST * SYMDMA_TILE;         // [KC][256]   v on this wave's own rows
  float* __restrict__ vw = xo_l + KC * 256;                     // [KC][VWF]   v on the part's window (linear image)
  float* __restrict__ tw = vw + KC * VWF;                       // [KC][WP]    transposed-term windows, plane-permuted
  const int g0 = 7 * d0 + p1lo - 3;
  const int wb = d0 * S0 + (p1lo - 3) * S1;
  const int WL = 256 + (ntile - 1) * S1 + 10;
  const int vsh = (iw0 + wb - 3) & 3;
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

  // ---- prologue: first tile, then per column the own rows and the part's v window, then the second tile
  issue_tile(0);
  const int jal = (iw0 + wb - 3) & ~3;
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    const int col = c0 + c < k ? c0 + c : k - 1;     // padding columns of the last pass recompute the last column (never written)
    const float* __restrict__ Vc = V + (int64_t)col * m;
    glds_b128(Vc + (live ? i4 : 0), xo_a + (unsigned)(c * 256 * sizeof(float)));
    const unsigned vwc = vw_a + (unsigned)(c * VWF * sizeof(float));
    for (int n = 0; n < vw128; ++n) {
      int j = jal + 4 * (64 * n + lane);
      j = j < 0 ? 0 : (j > m - 4 ? m - 4 : j);
      glds_b128(Vc + j, vwc + 1024u * n);
    }
    for (int n = 0; n < vw32; ++n) {
      int j = jal + 256 * vw128 + 64 * n + lane;
      j = j < 0 ? 0 : (j > m - 1 ? m - 1 : j);
      glds_b32(Vc + j, vwc + 1024u * vw128 + 256u * n);
    }
  }
  if (1 < ntile) issue_tile(1);
  for (int e = lane; e < KC * WP; e += 64) tw[e] = 0.f;
  wait_vm_tiles(1 < ntile - 1 ? 1 : ntile - 1);      // tile 0 and every window have landed
  float xo[KC][4], acc[KC][4], dg[KC][4];
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    const float4 x4 = *reinterpret_cast<const float4*>(xo_l + c * 256 + 4 * lane);
    xo[c][0] = live ? x4.x : 0.f; xo[c][1] = live ? x4.y : 0.f; xo[c][2] = live ? x4.z : 0.f; xo[c][3] = live ? x4.w : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[c][r] = dg[c][r] = 0.f;
  }

  for (int t = 0; t < ntile; ++t) {
    const int g = g0 + t;
    const int rem = ntile - 1 - t;
    wait_vm_tiles(rem < NST - 1 ? rem : NST - 1);
    const float* __restrict__ st = stage + (t % NST) * SYMDMA_TILE;
    float a[7][4];                                    // a[s][r]: row i4 + r, innermost offset digit s
    if (g == 0) {
      float4 q[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) q[r] = *reinterpret_cast<const float4*>(st + 16 * lane + 4 * r);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        a[0][r] = a[1][r] = a[2][r] = 0.f;
        a[3][r] = q[r].x; a[4][r] = q[r].y; a[5][r] = q[r].z; a[6][r] = q[r].w;
      }
    } else {
      float v[28];
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
    wave_lgkm_fence();                                // the stage has been copied out: refill it
    if (t + NST < ntile) issue_tile(t + NST);
    const int w0 = 4 * lane + t * S1;                 // window index of row i4 + f - 3
    if (live) {
      // One wave per SIMD is all the LDS budget allows, so nothing hides an LDS round trip: the window reads of ALL columns
      // are issued together, then all FMAs, then each lane-disjoint read-modify-write phase once for all columns (3 round
      // trips per tile instead of 3 per column).
      float win[KC][10], tr[KC][10];
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        const float* __restrict__ wsrc = vw + c * VWF + (w0 + vsh);
#pragma unroll
        for (int e = 0; e < 10; ++e) { win[c][e] = wsrc[e]; tr[c][e] = 0.f; }
      }
#pragma unroll
      for (int c = 0; c < KC; ++c) {
#pragma unroll
        for (int s = 0; s < 7; ++s) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            acc[c][r] += a[s][r] * win[c][s + r];
            if (g == 0 && s == 3) dg[c][r] = a[s][r] * xo[c][r];      // the diagonal: counted once
            else tr[c][s + r] += a[s][r] * xo[c][r];
          }
        }
      }
#pragma unroll
      for (int ph = 0; ph < 3; ++ph) {                // lane-disjoint phases (see spmv_sym_dma.h): read all, then write all
        constexpr int NE[3] = {4, 4, 2};
        float* cell[KC][4];
        float old[KC][4];
#pragma unroll
        for (int c = 0; c < KC; ++c)
#pragma unroll
          for (int u = 0; u < NE[ph]; ++u) {
            const int idx = w0 + 4 * ph + u;
            cell[c][u] = tw + c * WP + (idx & 3) * W4 + (idx >> 2);
            old[c][u] = *cell[c][u];
          }
#pragma unroll
        for (int c = 0; c < KC; ++c)
#pragma unroll
          for (int u = 0; u < NE[ph]; ++u) *cell[c][u] = old[c][u] + tr[c][4 * ph + u];
        wave_lgkm_fence();
      }
    }
  }
  // ---- epilogue: direct partials, window flush, p.Hp
  const int64_t km = (int64_t)k * m;
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    const bool cw = c0 + c < k;                       // padding columns write nothing
    if (live && cw) *reinterpret_cast<float4*>(part + (int64_t)y * km + (int64_t)(c0 + c) * m + i4) = make_float4(acc[c][0], acc[c][1], acc[c][2], acc[c][3]);
    if (cw) {
      float* __restrict__ tacc = part + (int64_t)NPARTS * km + (int64_t)(c0 + c) * m;
      const float* __restrict__ twc = tw + c * WP;
      const int jbase = iw0 + wb - 3;
      for (int idx = lane; idx < WL; idx += 64) {
        const float v = twc[(idx & 3) * W4 + (idx >> 2)];
        const int j = jbase + idx;
        if (v != 0.f && j >= 0 && j < m) atomic_add_real(tacc + j, v);
      }
    }
    if (DOT) {
      double pd = 0;
      if (live && cw) {
#pragma unroll
        for (int r = 0; r < 4; ++r) pd += (double)xo[c][r] * (2.0 * (double)acc[c][r] - (double)dg[c][r]);
        if (g0 == 0 && add) {
          const float4 ad = *reinterpret_cast<const float4*>(add + (int64_t)(c0 + c) * m + i4);
          pd += (double)beta * ((double)xo[c][0] * ad.x + (double)xo[c][1] * ad.y + (double)xo[c][2] * ad.z + (double)xo[c][3] * ad.w);
        }
      }
      pd = wave_reduce_sum<double>(pd);
      if (lane == 0 && cw) pcg_dot_add(dots, c0 + c, pd);
    }
  }
}

static inline size_t symdma_mc_lds_bytes(int g2, int kc) {
  return (size_t)(2 * SYMDMA_TILE + kc * (256 + symdma_vwf(g2) + symdma_wp(g2))) * sizeof(float);
}
