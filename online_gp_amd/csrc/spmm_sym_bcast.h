// Symmetric half-stencil SpMM for 64 right-hand sides: the coefficients travel on the VECTOR path and are broadcast by DPP.
// Included by solve.hip after spmm_sym_cols.h (same tiling, same operands, same results up to the order of the fp additions).
//
// k_spmm_sym_cols keeps a lane = a column and feeds every wave-uniform stencil coefficient through the scalar unit (s_load ->
// SGPR operand).  That reads A_h (86 MB at 50^3, twice) through a path with a handful of requests in flight per wave: each
// 64-byte s_load is a scalar-cache miss served from the Infinity Cache / HBM at ~1..2 us, a wave cannot have more than two or
// three of them outstanding (SGPR budget, one all-or-nothing lgkmcnt), and hipcc's pairing of the FMAs into v_pk_fma_f32 costs
// more s_mov / v_mov than it saves (PMC: 42 M VALU + 56 M SALU instructions per launch, 225..240 us against a 70 us FMA bound).
//
// Here the same RT x 64 tile takes its coefficients with ordinary vector loads -- deep queue, counted vmcnt, hoisted by the
// compiler: lane l fetches the 16 bytes at span + 16 (l & 15), so ONE global_load_dwordx4 brings 64 coefficients (each 16-lane
// row of the wave holds the same 16 x 4) -- and every FMA picks its coefficient with the DPP row broadcast
//     v_fmac_f32_dpp acc, coef, win  row_newbcast:n        (acc += coef[lane n of my row] * win)
// so a coefficient costs neither an SGPR nor a scalar instruction: per tile and stencil group 224 FMAs, 44 window loads (a row
// of V per load, 256 B) and 5 coefficient loads (2 for the 112 reals of the direct span, 3 for the 154 of the transposed one).
// The windows are those of k_spmm_sym_cols: RT + 6 rows per term; rows outside the grid are clamped for the direct term (they only
// meet coefficients that are exactly zero) and zeroed for the transposed one (their "coefficients" are whatever lies before /
// behind the group's array, finite by construction); lanes whose 16 bytes would leave A_h are clamped into it -- they can only
// belong to rows outside the grid.
#pragma once

// fp64 (v_fmac_f64_dpp: the 64-bit DPP forms of gfx90a+ allow exactly this control, row_newbcast): a 16-byte load brings 2 coefficients per
// lane, 32 per instruction; a row of V is 512 B.

// 16 bytes of coefficients from an address that is only aligned to the element size
template <typename real> struct spmmb_vec;
template <> struct spmmb_vec<float> { typedef float type __attribute__((ext_vector_type(4), aligned(4))); };
template <> struct spmmb_vec<double> { typedef double type __attribute__((ext_vector_type(2), aligned(8))); };

template <typename F, int... I>
__device__ __forceinline__ void spmmb_for_impl(F&& fn, std::integer_sequence<int, I...>) {
  (fn(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void spmmb_for(F&& fn) {
  spmmb_for_impl(fn, std::make_integer_sequence<int, N>{});
}

// A VALU write of a VGPR needs two wait states before a DPP read of it, and the compiler's hazard recogniser does not look into
// inline asm.  The coefficient registers are written by loads, but the clamped / unclamped load branches merge in front of the
// FMAs and a merge may be a v_mov: every FMA block opens with the two wait states.  (Tile heights whose registers spill -- fp64
// at RT = 32 -- got copies in between the FMAs and wrong digits: RT stays where nothing spills, 94 / 173 VGPRs at RT = 16.)
__device__ __forceinline__ void spmmb_dpp_guard() { asm volatile("s_nop 1"); }

// acc += coef[lane BC of this lane's 16-lane row] * w
template <int BC>
__device__ __forceinline__ void spmmb_fma(float& acc, float coef, float w) {
  asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(coef), "v"(w), "n"(BC));
}
template <int BC>
__device__ __forceinline__ void spmmb_fma(double& acc, double coef, double w) {
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(coef), "v"(w), "n"(BC));
}

// One wave = RT rows x 64 columns, four waves per block; blockIdx.y = the 64-column slice.  Vt: one row-major [m][64] image per
// slice (k_transpose_cm_rm<.., SLICED>), Ot column-major [k][m].
// DOT: dots[c] += sum_j Vt[j][c] * Ot[c][j].  a_len = number of reals in A_h.
// VAR = 1: the product.  Timing ablations (wrong results; WISKI_SPMM_BCAST=3 / 4 in a -DWISKI_SPMMB_ABLATE build): 3 without the
// FMAs (loads only: 117 us at 50^3), 4 without the window loads (FMAs + coefficient loads: 97 us); the product takes 138 us.
// Requesting the operands of both terms before the first FMA (one memory latency per group) beats term-by-term order by 5 us;
// a barrier per group (to keep the block's four waves, whose windows overlap by 6 rows, in step for L1 hits) costs 2..5 us; term-by-term
// order ENFORCED with scheduling barriers (79 instead of 94 VGPRs, fp64 124 instead of 173) is 6 us slower in fp32 and no faster in fp64.
// Round 5: the block's four waves on tiles 48 rows apart (two short of a grid line at 50^3: the window of wave w + 1 for group g is 20 / 22 of the
// one wave w loads for group g + 1 a step later) instead of consecutive ones: 136-138 us against 135 -- sharing windows through L1 changes nothing:
// the ~1 200 row loads per tile are bound by the L1's 64 B / clk (4 clk per 256-byte row: 71 us; + 29 us of coefficient loads = the 117 us of the
// loads-only ablation), not by L2 fetches.  An LDS-staged window (128 B / clk) would halve the 71 us at the price of one block per CU.
template <typename real, bool DOT, int RT, int VAR>
__global__ __launch_bounds__(256) void k_spmm_sym_bcast(GridDev<real> G, const real* __restrict__ A_h, int64_t a_len, const real* __restrict__ Vt,
                                                        int k, int ng, real* __restrict__ Ot, double* __restrict__ dots) {
  static_assert(RT % 4 == 0, "column-major stores are 16-byte groups of rows");
  typedef typename spmmb_vec<real>::type cvec;
  constexpr int WN = RT + 6, KP = 64;
  constexpr int CPL = 16 / (int)sizeof(real);     // coefficients per lane and load
  constexpr int CL = 16 * CPL;                    // coefficients per load instruction (the 16 lanes of a row)
  const int m = G.m, d = G.d;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile = 4 * ((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) + wv;   // XCD-contiguous row ranges (see the launch)
  const bool active = tile * RT < m;               // padding waves recompute tile 0 and discard it (they must reach the barrier)
  const int j0 = active ? tile * RT : 0;
  const int c = blockIdx.y * 64 + lane;            // this lane's column (the last slice is padded to 64 columns with zeros)
  const real* __restrict__ vslice = Vt + (int64_t)blockIdx.y * m * KP;     // wave-uniform: the window loads below are SGPR base + lane + immediate
  const real* __restrict__ vcol = vslice + lane;
  const int l4 = CPL * (lane & 15);
  real acc[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r) acc[r] = (real)0;
  const int cP = ng - 1;                           // prefix code of the centre: (7^(d-1) - 1) / 2

  // NL 16-byte loads of the coefficient span that starts at real `base` of A_h; lanes that would leave A_h are clamped into it
  auto load_span = [&](auto nl_tag, int64_t base, cvec* cf) {
    constexpr int NL = decltype(nl_tag)::value;
    if (base >= 0 && base + CL * NL <= a_len) {    // wave-uniform: always, except at the two ends of A_h
      const real* __restrict__ p = A_h + base + l4;
#pragma unroll
      for (int n = 0; n < NL; ++n) cf[n] = *reinterpret_cast<const cvec*>(p + CL * n);
    } else {
#pragma unroll
      for (int n = 0; n < NL; ++n) {
        int64_t i = base + CL * n + l4;
        i = i < 0 ? 0 : (i > a_len - CPL ? a_len - CPL : i);
        cf[n] = *reinterpret_cast<const cvec*>(A_h + i);
      }
    }
  };

  auto group = [&](auto centre_tag, int64_t gbase, int f) {
    constexpr bool CENTRE = decltype(centre_tag)::value;
    constexpr int RS = CENTRE ? 4 : 7;               // reals per row
    constexpr int S0 = CENTRE ? 3 : 0;               // first stored digit
    constexpr int T0 = CENTRE ? 4 : 0;               // first digit of the transposed term (digit 3 of the centre group is the diagonal)
    constexpr int ND = (RT * RS + CL - 1) / CL, NT = (WN * RS + CL - 1) / CL;
    real win[WN], src[WN];
    cvec cd[ND], ct[NT];
    const int wb = j0 + f - 3, ib = j0 - f - 3;
    auto load_direct = [&]() {
      load_span(std::integral_constant<int, ND>{}, gbase + (int64_t)RS * j0, cd);
      if constexpr (VAR == 4) {
#pragma unroll
        for (int e = 0; e < WN; ++e) win[e] = (real)(1 + e);
      } else if (wb >= 0 && wb + WN <= m) {
        const real* __restrict__ wp = vslice + (int64_t)(wb + WN / 2) * KP;       // uniform base at the middle row: every row within the signed 13-bit immediate
#pragma unroll
        for (int e = 0; e < WN; ++e) win[e] = wp[(e - WN / 2) * KP + lane];
      } else {
#pragma unroll
        for (int e = 0; e < WN; ++e) {
          int j = wb + e;
          j = j < 0 ? 0 : (j >= m ? m - 1 : j);      // clamped rows only ever meet coefficients that are exactly zero
          win[e] = vcol[(int64_t)j * KP];
        }
      }
    };
    auto load_transposed = [&]() {
      load_span(std::integral_constant<int, NT>{}, gbase + (int64_t)RS * ib, ct);
      if constexpr (VAR == 4) {
#pragma unroll
        for (int e = 0; e < WN; ++e) src[e] = (real)(2 + e);
      } else if (ib >= 0 && ib + WN <= m) {
        const real* __restrict__ sp = vslice + (int64_t)(ib + WN / 2) * KP;
#pragma unroll
        for (int e = 0; e < WN; ++e) src[e] = sp[(e - WN / 2) * KP + lane];
      } else {
#pragma unroll
        for (int e = 0; e < WN; ++e) {
          const int i = ib + e;
          src[e] = (i >= 0 && i < m) ? vcol[(int64_t)i * KP] : (real)0;     // wave-uniform condition
        }
      }
    };
    // ---- direct term: out[j0 + r] += a(s, j0 + r) * v[j0 + r + f + s - 3]
    auto fma_direct = [&]() {
      if constexpr (VAR == 3) {
#pragma unroll
        for (int e = 0; e < WN; ++e) asm volatile("" ::"v"(win[e]));
#pragma unroll
        for (int n = 0; n < ND; ++n) asm volatile("" ::"v"(cd[n]));
      } else {
        spmmb_dpp_guard();
        spmmb_for<RT * RS>([&](auto it) {
          constexpr int idx = decltype(it)::value, r = idx / RS, s = S0 + idx % RS;
          spmmb_fma<(idx % CL) / CPL>(acc[r], cd[idx / CL][idx % CPL], win[r + s]);
        });
      }
    };
    // ---- transposed term: out[j] += a(s, i') * v[i'],  i' = j - f - (s - 3) = ib + e,  r = e + s - 6
    auto fma_transposed = [&]() {
      if constexpr (VAR == 3) {
#pragma unroll
        for (int e = 0; e < WN; ++e) asm volatile("" ::"v"(src[e]));
#pragma unroll
        for (int n = 0; n < NT; ++n) asm volatile("" ::"v"(ct[n]));
      } else {
        spmmb_dpp_guard();
        spmmb_for<WN * RS>([&](auto it) {
          constexpr int idx = decltype(it)::value, e = idx / RS, s = S0 + idx % RS, r = e + s - 6;
          if constexpr (s >= T0 && r >= 0 && r < RT) spmmb_fma<(idx % CL) / CPL>(acc[r], ct[idx / CL][idx % CPL], src[e]);
        });
      }
    };
    load_direct();        // operands of both terms first (one memory latency per group, not two)
    load_transposed();
    fma_direct();
    fma_transposed();
  };

  group(std::true_type{}, (int64_t)0, 0);
  for (int g = 1; g < ng; ++g) {
    int f = 0, rem = cP + g;                         // flat offset of the group's centre digit (leading d-1 stencil digits)
    for (int q = d - 2; q >= 0; --q) {
      f += (rem % 7 - 3) * G.stride[q];
      rem /= 7;
    }
    group(std::false_type{}, (int64_t)(7 * g - 3) * m, f);
  }

  double dot = 0;
  const bool cw = c < k && active;               // this lane writes (padding columns and padding waves do not)
  if (j0 + RT <= m) {
    real* __restrict__ op = Ot + (int64_t)(c < k ? c : 0) * m + j0;      // j0 and m are multiples of 4: 16-byte aligned
#pragma unroll
    for (int r = 0; r < RT; r += 4)
      if (cw) store4<real>(op + r, acc[r], acc[r + 1], acc[r + 2], acc[r + 3]);
    if (DOT) {
      const real* __restrict__ vp = vcol + (int64_t)j0 * KP;
#pragma unroll
      for (int r = 0; r < RT; ++r) dot += (double)vp[r * KP] * (double)acc[r];
    }
  } else {
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      const int j = j0 + r;
      if (j < m && cw) Ot[(int64_t)c * m + j] = acc[r];
      if (DOT && j < m && active) dot += (double)vcol[(int64_t)j * KP] * (double)acc[r];
    }
  }
  if constexpr (DOT) {
    __shared__ double s_dot[4][64];
    s_dot[wv][lane] = active ? dot : 0.0;
    __syncthreads();
    if (wv == 0 && c < k) pcg_dot_add(dots, c, s_dot[0][lane] + s_dot[1][lane] + s_dot[2][lane] + s_dot[3][lane]);
  }
}
