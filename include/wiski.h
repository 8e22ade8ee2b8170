/*
 * wiski.h -- C ABI of libwiski_hip.so: the MI355X (gfx950) implementation of the
 * WISKI streaming-update hot path of wjmaddox/online_gp.
 *
 * The reference has no FFI: its boundary is the Python object surface
 * (SURVEY.md section 8b).  Each entry point below replaces the torch/gpytorch
 * op sequence at the cited reference call site (paths relative to the
 * reference repo; BFN = online_gp/models/batched_fixed_noise_online_gp.py,
 * URLT = online_gp/lazy/updated_root_lazy_tensor.py,
 * BWM = online_gp/mlls/batched_woodbury_marginal_log_likelihood.py).
 *
 * Conventions
 *   - every function is `extern "C" int fn(..., void* stream)`; returns
 *     WISKI_OK or a negative WISKI_E_* code, never throws, never allocates
 *     memory the caller can see;
 *   - `stream` is a hipStream_t; work is enqueued asynchronously on it
 *     (wiski_pcg_* additionally synchronises the stream at its convergence
 *     checks);
 *   - pointers named d_* are DEVICE pointers borrowed for the call; all other
 *     pointers are HOST pointers (small per-dim arrays, outputs of wiski_pcg);
 *   - `_f32` / `_f64` suffix = scalar type of every `real` array;
 *   - dense vectors over the inducing grid are stored column-major as
 *     V[k][m] (k contiguous m-vectors); flat grid index has dim 0 slowest;
 *   - W^T D^-1 W is kept in block-stencil form A_st[o][i] = A[i, i+off(o)],
 *     o in 7^d relative offsets, i in [0, m)  (layout: o-major, m contiguous).
 */
#ifndef WISKI_H
#define WISKI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WISKI_OK 0
#define WISKI_E_BADARG (-1)   /* unsupported d / k / null pointer          */
#define WISKI_E_LAUNCH (-2)   /* hip launch or runtime error               */
#define WISKI_E_WORKSPACE (-3) /* workspace too small                      */
#define WISKI_E_NOTCONV (-4)  /* wiski_pcg hit max_iter (result still written) */
#define WISKI_PENDING 1       /* wiski_pcg_async / wiski_stream_step: the solve was started, its first poll is in flight */

#define WISKI_MAX_DIM 4

/* Inducing-grid geometry (host struct, passed by pointer, read at call time).
 * g0/h follow gpytorch's GridInterpolationKernel grid (BFN:114-120):
 * delta=(hi-lo)/(g-2), grid=linspace(lo-delta, hi+delta, g). */
typedef struct wiski_grid {
  int32_t d;                      /* 1..WISKI_MAX_DIM                      */
  int32_t g[WISKI_MAX_DIM];       /* points per dim (incl. 2 extension pts) */
  double g0[WISKI_MAX_DIM];       /* first grid point per dim               */
  double h[WISKI_MAX_DIM];        /* grid spacing per dim                   */
} wiski_grid;

int wiski_version(void);
/* A hipStream_t for work off the critical path (lazy/two_level.py: the refresh of the two-level block), at the device's lowest priority
 * when asked: beside a kernel of the caller's stream that fills every wave slot the dispatcher then serves the caller first. */
int wiski_side_stream_create(int32_t lowest_priority, void** out);

/* a1 -- replaces covar_module(X).evaluate_kernel() (BFN:143,205,261,421):
 * cubic (Keys a=-0.5) interpolation indices/values, T=4^d taps per point,
 * d_idx int32 [n][T], d_val real [n][T].  Out-of-grid points set *d_err != 0
 * (the reference raises RuntimeError there). */
int wiski_interp_f32(const wiski_grid* grid, const float* d_x, int64_t n, int32_t* d_idx, float* d_val, int32_t* d_err, void* stream);
int wiski_interp_f64(const wiski_grid* grid, const double* d_x, int64_t n, int32_t* d_idx, double* d_val, int32_t* d_err, void* stream);

/* a14 -- replaces left_interp(idx, val, pred_mean) (BFN:206-210,235), fused:
 * weights are recomputed from x on the fly, nothing but x and V is read.
 * d_out[n][k] = W(x) V,  V given as [k][m].  diag != 0: requires k == n and
 * writes d_out[p] = W(x_p) . V[p]  (per-query quadratic forms, BFN:222-228). */
int wiski_gather_f32(const wiski_grid* grid, const float* d_x, int64_t n, const float* d_V, int32_t k, int32_t diag, float* d_out, int32_t* d_err, void* stream);
int wiski_gather_f64(const wiski_grid* grid, const double* d_x, int64_t n, const double* d_V, int32_t k, int32_t diag, double* d_out, int32_t* d_err, void* stream);

/* Input gradient of the interpolation row (learned stems; streaming_partial_mll.py:20-36
 * differentiates through W):  d_out[n][d] = d/dx ( W(x_p) . V_c ), c = 0 (diag == 0, one
 * column d_V[m]) or c = p (diag != 0, d_V[n][m]). */
int wiski_gather_grad_f32(const wiski_grid* grid, const float* d_x, int64_t n, const float* d_V, int32_t diag, float* d_out, void* stream);
int wiski_gather_grad_f64(const wiski_grid* grid, const double* d_x, int64_t n, const double* d_V, int32_t diag, double* d_out, void* stream);

/* a14 with the dense operand stored ROW-major, d_Vr[m][ncols] (left_interp's own layout,
 * BFN:206-210): d_out[n][ncols] = W(x) Vr.  Every tap reads a contiguous row segment;
 * used for W* M with the cached dense posterior M of small grids (BFN:222-225). */
int wiski_gather_rows_f32(const wiski_grid* grid, const float* d_x, int64_t n, const float* d_Vr, int32_t ncols, float* d_out, int32_t* d_err, void* stream);
int wiski_gather_rows_f64(const wiski_grid* grid, const double* d_x, int64_t n, const double* d_Vr, int32_t ncols, double* d_out, int32_t* d_err, void* stream);

/* a14, ELL form -- same product from materialised (idx, val) rows of width T
 * (the layout InterpolatedLazyTensor keeps; BFN:206-210). k == 1 only. */
int wiski_gather_ell_f32(const int32_t* d_idx, const float* d_val, int64_t n, int32_t T, const float* d_v, float* d_out, void* stream);
int wiski_gather_ell_f64(const int32_t* d_idx, const double* d_val, int64_t n, int32_t T, const double* d_v, double* d_out, void* stream);
/* The same product for rows written by wiski_interp on `grid` (idx[tap] = base + sum_q c_q stride_q; T = 4^d): d_v is first
 * copied into a blocked layout (second-to-last dim in groups of 4, once per alignment of the stencil: csrc/gather_ell_dma.h) in which
 * every lane's four taps are one aligned group of four reals and a row touches ~5.5 cache lines instead of ~17.5 -- the gathers of v, not the
 * idx / val stream, bound the plain form.  d_vpack: scratch of
 * wiski_gather_ell_pack_elems(grid) reals, 16-byte aligned (0 elements: d = 1 or a grid too large -- pass NULL).  Small row
 * counts, d = 1, a NULL or misaligned scratch take wiski_gather_ell. */
int wiski_gather_ell_grid_f32(const wiski_grid* grid, const int32_t* d_idx, const float* d_val, int64_t n, const float* d_v, float* d_vpack, float* d_out, void* stream);
int wiski_gather_ell_grid_f64(const wiski_grid* grid, const int32_t* d_idx, const double* d_val, int64_t n, const double* d_v, double* d_vpack, double* d_out, void* stream);
int64_t wiski_gather_ell_pack_elems(const wiski_grid* grid);

/* a2+a3+a4+URLT:58 -- replaces _initialize_caches / _update_cache_dicts /
 * UpdatedRootLazyTensor.update's `tensor + V V^T` (BFN:31-60,155-171):
 *   d_b[m]      += W^T (y * wb)            interpolation_cache
 *   d_A_st      += W^T diag(wa) W          WtW (block stencil; may be NULL)
 *   d_stats[0]  += sum y^2 wb              response_cache      (double)
 *   d_stats[1]  += sum log(noise)          D_logdet            (double)
 * wa/wb/noise are per-point device arrays [n] (wb = 1/noise; wa = 1/noise at
 * initialisation, 1/max(noise,1e-7) on updates -- BFN:163). */
int wiski_scatter_stats_f32(const wiski_grid* grid, const float* d_x, const float* d_y, const float* d_wa, const float* d_wb, const float* d_noise, int64_t n, float* d_b, float* d_A_st, double* d_stats, int32_t* d_err, void* stream);
int wiski_scatter_stats_f64(const wiski_grid* grid, const double* d_x, const double* d_y, const double* d_wa, const double* d_wb, const double* d_noise, int64_t n, double* d_b, double* d_A_st, double* d_stats, int32_t* d_err, void* stream);

/* Same statistics with W^T diag(wa) W accumulated into the SYMMETRIC HALF stencil d_A_half --
 * the model's native WtW storage (replaces the dense m x m tensor of URLT:42,58).  Only stencil
 * offsets >= the centre are kept ((7^d+1)/2 * m reals: half the atomics of the full form, half the
 * bytes per product, and the buffer the data-parallel path all-reduces), in the "row-interleaved"
 * layout: with P the leading d-1 base-7 digits of an offset, s its innermost digit and
 * g = P - P_centre >= 0,
 *     g == 0 :  d_A_half[4 i + (s - 3)]                s = 3..6  (s = 3: the diagonal A[i,i])
 *     g >= 1 :  d_A_half[(7 g - 3) m + 7 i + s]        s = 0..6
 * holds A[i, i + off(P, s)].  (The 16 innermost-digit combinations of a tap pair then fall into one
 * 88-byte span, which is what the transaction-bound memory-side atomics want: 72 us vs 208 us per 4096
 * points at 50^3 against an offset-major half stencil.)  wiski_stencil_expand_add unpacks it into a
 * full offset-major stencil: A_st += expand(d_A_half), d_A_half = 0. */
int wiski_scatter_stats_sym_f32(const wiski_grid* grid, const float* d_x, const float* d_y, const float* d_wa, const float* d_wb, const float* d_noise, int64_t n, float* d_b, float* d_A_half, double* d_stats, int32_t* d_err, void* stream);
int wiski_scatter_stats_sym_f64(const wiski_grid* grid, const double* d_x, const double* d_y, const double* d_wa, const double* d_wb, const double* d_noise, int64_t n, double* d_b, double* d_A_half, double* d_stats, int32_t* d_err, void* stream);
/* One-launch form used by the model: as above (half != 0: d_A is the symmetric half stencil,
 * else the full offset-major one) and additionally
 *   d_cnt[m] += W^T wa      the row sums of the increment (the preconditioner's data-density statistic;
 *                           d_cnt may be NULL)
 *   d_res[m] += W^T (wb y - wa (W d_u))      (d_u / d_res: both or neither; half form only)
 * the residual carry-over: if d_res held b - z - A u for the current solution (u, z) of
 * (Kt^-1 + A) u = b, it still does after the increment, so the next wiski_pcg warm start (warm = 2)
 * needs no A u product. */
int wiski_scatter_stats_cnt_f32(const wiski_grid* grid, const float* d_x, const float* d_y, const float* d_wa, const float* d_wb, const float* d_noise, int64_t n, float* d_b, float* d_A, int32_t half, float* d_cnt, const float* d_u, float* d_res, double* d_stats, int32_t* d_err, void* stream);
int wiski_scatter_stats_cnt_f64(const wiski_grid* grid, const double* d_x, const double* d_y, const double* d_wa, const double* d_wb, const double* d_noise, int64_t n, double* d_b, double* d_A, int32_t half, double* d_cnt, const double* d_u, double* d_res, double* d_stats, int32_t* d_err, void* stream);
int wiski_stencil_expand_add_f32(const wiski_grid* grid, float* d_A_half, float* d_A_st, void* stream);
int wiski_stencil_expand_add_f64(const wiski_grid* grid, double* d_A_half, double* d_A_st, void* stream);

/* Adds W(x)^T as k = n one-hot-interpolated columns: d_out[p][idx] += val
 * (the sparse `wmat` of BFN:22-28 for a query batch, kept column-dense only
 * for the k right-hand sides of a solve). d_out must be zeroed by the caller. */
int wiski_wt_columns_f32(const wiski_grid* grid, const float* d_x, int64_t n, float* d_out, int32_t* d_err, void* stream);
int wiski_wt_columns_f64(const wiski_grid* grid, const double* d_x, int64_t n, double* d_out, int32_t* d_err, void* stream);

/* replaces WtW._matmul (URLT:47-48) on the stencil form:
 * d_out[c] = beta * d_add[c] + A_st . d_V[c]   (d_add may be NULL). */
int wiski_stencil_spmv_f32(const wiski_grid* grid, const float* d_A_st, const float* d_V, int32_t k, const float* d_add, float beta, float* d_out, void* stream);
int wiski_stencil_spmv_f64(const wiski_grid* grid, const double* d_A_st, const double* d_V, int32_t k, const double* d_add, double beta, double* d_out, void* stream);
/* Same product on the symmetric half stencil d_A_half (layout: wiski_scatter_stats_sym); every
 * stored entry is used for A[i,j] and A[j,i], so a product streams half the HBM bytes of the full
 * form.  d_out must not alias d_V. */
int wiski_stencil_spmv_sym_f32(const wiski_grid* grid, const float* d_A_half, const float* d_V, int32_t k, const float* d_add, float beta, float* d_out, void* stream);
int wiski_stencil_spmv_sym_f64(const wiski_grid* grid, const double* d_A_half, const double* d_V, int32_t k, const double* d_add, double beta, double* d_out, void* stream);

/* a8/a9/a11 -- replaces Kuu @ V (BFN:334-348,363-366) with
 * Kuu = kron_i SymToeplitz(tcol_i): d_out[c] = scale * Kuu d_V[c].
 * d_tcol = concatenated first columns (sum_i g[i] reals).  d_tmp: scratch of
 * k*m reals (ping-pong); d_out must not alias d_V. */
int wiski_kron_toeplitz_mm_f32(const wiski_grid* grid, const float* d_tcol, const float* d_V, int32_t k, float scale, float* d_tmp, float* d_out, void* stream);
int wiski_kron_toeplitz_mm_f64(const wiski_grid* grid, const double* d_tcol, const double* d_V, int32_t k, double scale, double* d_tmp, double* d_out, void* stream);

/* a17 backward -- gradient of bilinear forms in Kuu w.r.t. the Toeplitz columns
 * (what autograd computes through Kuu in BWM:19-51 / BFN:334-366):
 *   d_grad[sum g] += d/d tcol  sum_c X[c]^T (kron_q SymToeplitz(tcol_q)) Y[c]
 * d_tmp: 2*k*m reals of scratch; d_grad is double and is accumulated into. */
int wiski_kron_toeplitz_grad_f32(const wiski_grid* grid, const float* d_tcol, const float* d_X, const float* d_Y, int32_t k, float* d_tmp, double* d_grad, void* stream);
int wiski_kron_toeplitz_grad_f64(const wiski_grid* grid, const double* d_tcol, const double* d_X, const double* d_Y, int32_t k, double* d_tmp, double* d_grad, void* stream);

/* Spectral functions of Kt = kscale*Kuu in its Kronecker eigenbasis (d_evec/d_eval
 * as for wiski_pcg): d_out[c] = V diag(lam^pw / (1 + shift*lam)^rw) V^T d_V[c].
 * pw = 0.5, rw = 0 gives the symmetric root Kt^(1/2) (logdet(Q) by stochastic
 * Lanczos quadrature, BWM:27 inv_quad_logdet).  d_tmp: k*m reals. */
int wiski_kron_spectral_mm_f32(const wiski_grid* grid, const float* d_evec, const float* d_eval, float kscale, float shift, float pw, float rw, const float* d_V, int32_t k, float* d_tmp, float* d_out, void* stream);
int wiski_kron_spectral_mm_f64(const wiski_grid* grid, const double* d_evec, const double* d_eval, double kscale, double shift, double pw, double rw, const double* d_V, int32_t k, double* d_tmp, double* d_out, void* stream);

/* CG branch of a12 (BFN:368-383; gpytorch linear_cg under Q.inv_matmul),
 * moved to inducing space: solves (Kt^-1 + A) U = RHS, Kt = kscale*Kuu, for k
 * columns by preconditioned CG (no inverse of Kt is ever applied):
 *   U = M RHS,  M = (Kt^-1 + A)^-1  (SURVEY 3.5).
 * Preconditioner:
 *   d_evec == NULL : P = Kt                      (needs d_tcol)
 *   d_evec != NULL : P = (Kt^-1 + shift * kron_q diag(t_q))^-1, a separable model of A
 *     (t_q = per-dim data-density profile; t_q = 1 gives shift * I), applied through the
 *     per-dim generalized eigenproblems  K_q = X_q D_q X_q^T,  X_q^T diag(t_q) X_q = I:
 *     d_evec = concatenated row-major g_q x g_q matrices X_q, d_evec2 = Z_q = diag(t_q) X_q
 *     (NULL when t_q = 1: Z_q = X_q orthogonal), d_eval = concatenated D_q (>= 0);
 *     shift = density scale (d_tcol unused).
 * d_U/d_Z [k][m]: solution and its pre-image (U = Kt Z). warm != 0 starts
 * from the given (U, Z) (must satisfy U = Kt Z), else from zero.
 * Stops when every column has ||r||/||rhs|| < tol or at max_iter; the host polls the
 * residual norms after first_check iterations (< 1: check_every) and then every
 * check_every iterations.  A poll is a tiny publish kernel writing to host-mapped memory that
 * the host spins on (no stream synchronisation).  d_err (may be NULL): the out-of-grid flag of
 * the interp/scatter/gather entry points; its raw value (bit 0: some point was outside the grid; bits 1..:
 * number of training points the scatter kernels dropped for that reason) rides on the last poll into
 * *h_err, so the caller needs no separate device-to-host read to raise the reference's RuntimeError.
 * Host threads: the poll buffer is per device and held under a mutex for the duration of a solve, so
 * concurrent solves on one device from several host threads serialise (the calling thread's current
 * device must be the one the pointers live on).
 * a_sym != 0: d_A_st is the symmetric half stencil (wiski_scatter_stats_sym layout) instead of
 * the full offset-major [7^d][m] one.
 * d_R (k*m reals, may be NULL): caller-owned residual buffer used instead of workspace scratch; on return
 * it holds rhs - Z - A U for the returned (U, Z).  warm = 2 (needs d_R): d_R already holds that residual
 * for the incoming (U, Z) -- kept current across streaming updates by wiski_scatter_stats_cnt's d_res --
 * so the solve starts without an A U product.
 * workspace: wiski_pcg_workspace_bytes(...) bytes of device scratch.
 * h_iters (host, may be NULL): iterations run; h_relres (host, k doubles, may
 * be NULL): final relative residuals. */
int64_t wiski_pcg_workspace_bytes(const wiski_grid* grid, int32_t k, int32_t max_iter, int32_t elem_size);
int wiski_pcg_f32(const wiski_grid* grid, const float* d_A_st, const float* d_tcol, float kscale, const float* d_evec, const float* d_evec2, const float* d_eval, float shift, const float* d_RHS, int32_t k, float* d_U, float* d_Z, int32_t warm, double tol, int32_t max_iter, int32_t check_every, int32_t first_check, void* d_work, int64_t work_bytes, int32_t* h_iters, double* h_relres, const int32_t* d_err, int32_t* h_err, int32_t a_sym, float* d_R, void* stream);
int wiski_pcg_f64(const wiski_grid* grid, const double* d_A_st, const double* d_tcol, double kscale, const double* d_evec, const double* d_evec2, const double* d_eval, double shift, const double* d_RHS, int32_t k, double* d_U, double* d_Z, int32_t warm, double tol, int32_t max_iter, int32_t check_every, int32_t first_check, void* d_work, int64_t work_bytes, int32_t* h_iters, double* h_relres, const int32_t* d_err, int32_t* h_err, int32_t a_sym, double* d_R, void* stream);

/* Stencil-sharded replicas (multi-GPU step that DIVIDES the work).  Rank r of n owns the groups
 * [r G / n, (r + 1) G / n) of the symmetric half stencil (G = (7^(d-1) + 1) / 2 groups, wiski_shard_groups): it scatters only
 * the tap pairs of its groups (wiski_scatter_stats_step_sharded: 1 / n of the atomics per point; b, cnt, the statistics and
 * the carried residual stay replicated, so every rank must see every point) and computes only its groups' share of A p
 * (wiski_pcg_sharded), which ONE all-reduce(SUM) of an m-vector plus the p . Ap slots per product turns into the full
 * product on the solve's stream.  Vectors, preconditioner and scalars are replicated: all ranks take identical iterations.
 * comm != NULL: RCCL (an ncclComm_t, see wiski_comm_*); else `allreduce(ctx, d_vec, n_vec, elem_bytes, d_dots, n_dots, stream)`
 * is called -- in place, SUM, d_dots fp64, must be ordered after the work already queued on `stream` and before what is
 * queued next (tests route it through another transport).  One right-hand side, half stencil, m % 4 == 0; any d, fp32 and fp64
 * (d = 3 fp32: the LDS-DMA kernel on a part table; otherwise the LDS-window kernel on the replica's group range); the RCCL route
 * (comm != NULL) sums the vector in its own precision (ncclFloat32 / ncclFloat64). */
typedef int (*wiski_allreduce_fn)(void* ctx, void* d_vec, int64_t n_vec, int32_t elem_bytes, double* d_dots, int64_t n_dots, void* stream);
typedef struct wiski_shard {
  int32_t rank, nranks;
  void* comm;
  wiski_allreduce_fn allreduce;
  void* ctx;
} wiski_shard;
int wiski_shard_groups(int32_t d, int32_t rank, int32_t nranks, int32_t* g_lo, int32_t* g_hi);

/* Two-level preconditioner of the fused fp32 solve (d = 3).  The separable model
 * P = (Kt^-1 + a kron_q diag(t_q))^-1 is exact only for a separable data density; on road-like (line-clustered) streams
 * W^T D^-1 W is far from that and a warm CG step needs 6 iterations instead of 2.5.  In the generalized eigenbasis X of the
 * separable model (the tables wiski_pcg already transforms with: X^T (kron diag t) X = I, Kt = X^-T D X^-1) the system
 * matrix is D^-1 + X^T A X; the two-level form keeps the EXACT block N = (D_S^-1 + X_S^T A X_S)^-1 on the r modes of largest
 * prior eigenvalue D (the directions the data inform) and the diagonal model D / (1 + a D) on all others -- block Jacobi in
 * spectral coordinates, SPD.  The caller owns N (r x r, fp32) and refreshes it as the stream grows (projection of the new
 * points on X_S + a GEMM + an r x r Cholesky, off the critical path: a stale block only costs iterations, never accuracy).
 * The coupling crosses the dim-0 slabs of the fused slab kernel: the blocks that hold selected modes exchange their r
 * coefficients through d_cs (self-validating 64-bit words stamped with a per-launch number drawn on the host: a launch sequence
 * that uses a wiski_twolevel must not be captured into a graph and replayed), every block then forms the rows of N c it needs.
 * Modes are identified by their eigen-indices (i0, x, y) in the order of the eigen tables handed to the solve. */
typedef struct wiski_twolevel {
  int32_t r;                 /* modes in the exact block, 1 <= r <= 512 */
  int32_t nslab;             /* number of dim-0 eigen-indices i0 that hold at least one selected mode */
  const uint64_t* d_mask;    /* [g0][64]: bit y of d_mask[i0 * 64 + x] set <=> mode (i0, x, y) is selected */
  const int32_t* d_off;      /* [g0 + 1]: the selected modes of slab i0 are d_off[i0] .. d_off[i0 + 1] - 1 in block order */
  const uint16_t* d_pos;     /* [r]: x << 8 | y of every selected mode, block order (sorted by i0) */
  const float* d_N;          /* [r][r] row-major, symmetric positive definite, block order */
  uint64_t* d_cs;            /* [r + 1]: r exchange words {application number << 32 | fp32 bits} + one sticky count of exchange words that
                              * never arrived (their writer block was not co-resident: the application then used a stale coefficient --
                              * drop the block); zeroed ONCE by the caller.  Refused (WISKI_E_BADARG) where 2 g0 exceeds the CU count */
  float* d_mc;               /* [2][mc_cols][r] scratch of the multi-column form, or NULL: solves with k > 1 columns apply the block around the
                              * slab launch (two small launches per application: c_S = X_S^T r per column from the mode-0 image, d = N c_S;
                              * the slab kernel then substitutes d for the selected entries) -- no exchange words, no co-residency needed */
  int32_t mc_cols;           /* columns d_mc has room for; k > mc_cols (or d_mc == NULL with k > 1): WISKI_E_BADARG */
} wiski_twolevel;

/* Deferred convergence poll.  wiski_pcg_async_* = wiski_pcg_* plus a host-side handle (zero-initialised by the caller,
 * released with wiski_pcg_async_free) and a mode: 0 = as wiski_pcg; 1 = START: queue the iterations up to the first poll
 * (first_check), queue the poll, return WISKI_PENDING without waiting -- the host gets its time back while the GPU iterates;
 * 2 = RESUME with the same arguments: wait for that poll (normally long over), finish with synchronous polls if it had not
 * converged, fill h_iters / h_relres / h_err, return 0 / WISKI_E_NOTCONV.  Between START and RESUME nothing else may touch
 * (d_U, d_Z, d_R, d_work) or the system (d_A, d_RHS); the handle owns its own poll buffer. */
typedef struct wiski_pcg_async {
  int32_t state;   /* 0 idle, 1 a started solve is waiting to be resumed */
  int32_t it;      /* iterations queued so far */
  int64_t seq;     /* sequence number of the poll in flight */
  void* poll;      /* opaque: the handle's pinned poll buffer */
  int32_t prezeroed; /* set by wiski_stream_step when an earlier kernel of the step has zeroed the solve's scalar block and
                        accumulated partial vector (wiski_gather_zero): the next START / run skips its own zero launch */
  int32_t guard_ok;  /* set by RESUME: 1 when the poll it waited for found every column converged and no error flag, i.e. when a
                        kernel guarded by that poll (wiski_pcg_async_guard) has run */
  double shift;      /* wiski_stream_step: the preconditioner shift the pending solve was STARTed with -- its RESUME continues
                        with it even if the caller has meanwhile set the next step's shift in the argument struct */
} wiski_pcg_async;
int wiski_pcg_async_f32(const wiski_grid* grid, const float* d_A_st, const float* d_tcol, float kscale, const float* d_evec, const float* d_evec2, const float* d_eval, float shift, const float* d_RHS, int32_t k, float* d_U, float* d_Z, int32_t warm, double tol, int32_t max_iter, int32_t check_every, int32_t first_check, void* d_work, int64_t work_bytes, int32_t* h_iters, double* h_relres, const int32_t* d_err, int32_t* h_err, int32_t a_sym, float* d_R, void* stream, wiski_pcg_async* handle, int32_t mode);
int wiski_pcg_async_f64(const wiski_grid* grid, const double* d_A_st, const double* d_tcol, double kscale, const double* d_evec, const double* d_evec2, const double* d_eval, double shift, const double* d_RHS, int32_t k, double* d_U, double* d_Z, int32_t warm, double tol, int32_t max_iter, int32_t check_every, int32_t first_check, void* d_work, int64_t work_bytes, int32_t* h_iters, double* h_relres, const int32_t* d_err, int32_t* h_err, int32_t a_sym, double* d_R, void* stream, wiski_pcg_async* handle, int32_t mode);
int wiski_pcg_async_free(wiski_pcg_async* handle);
/* Speculation across the poll of a started solve (state == 1): *d_guard is a device int64 that the poll's publishing block sets
 * to +*expect when it finds the solve converged and the error flag clear -- exactly when RESUME will report convergence from
 * that poll -- and to -*expect otherwise.  A kernel queued behind the solve that reads it can run or skip itself without the
 * host having seen the poll (wiski_scatter_stats_step); RESUME then tells through handle->guard_ok which of the two happened. */
int wiski_pcg_async_guard(const wiski_pcg_async* handle, const void** d_guard, int64_t* expect);
/* Support of wiski_stream_step: the two device regions (pointer, bytes) a solve with these parameters zeroes before its first
 * kernel, and a predictive-mean gather (k <= 4 columns, as wiski_gather with diag = 0) whose kernel zeroes them on the way when
 * it can (*zeroed = 1; d = 3 and n <= 65536) -- one launch less per streaming step. */
int wiski_pcg_zero_regions_f32(const wiski_grid* grid, int32_t k, int32_t max_iter, void* d_work, int32_t a_sym, void** p1, int64_t* n1_bytes, void** p2, int64_t* n2_bytes);
int wiski_pcg_zero_regions_f64(const wiski_grid* grid, int32_t k, int32_t max_iter, void* d_work, int32_t a_sym, void** p1, int64_t* n1_bytes, void** p2, int64_t* n2_bytes);
/* The absorb of a streaming step: wiski_scatter_stats_cnt (half-stencil form) that also (i) writes the predictive mean of every
 * point under the CURRENT posterior mean d_u into d_mean_out [n] (the w_p . u it forms for the residual carry anyway -- BFN:206-210
 * without a gather launch; d_res may be NULL when only the mean is wanted), and (ii) zeroes the two regions of
 * wiski_pcg_zero_regions on the way (n*_bytes = 0: none).  d_guard != NULL: the whole kernel (absorb, mean, zeroing) runs only
 * if *d_guard == guard_expect when it starts and is a no-op otherwise (wiski_pcg_async_guard: the absorb of the next batch is
 * queued behind a solve whose convergence poll the host has not read yet).
 * d_bin / bin_bytes (optional, d = 3): a workspace of at least wiski_scatter_bin_bytes(grid, n, sizeof(real)) bytes, zeroed ONCE
 * by the caller and then left alone.  With it, batches of 8192 points or more (WISKI_OWNER_MIN_POINTS) take the owner-computes
 * form -- points binned by cell, one block per grid line adds all contributions to its rows with plain read-modify-writes -- whose
 * cost is a sweep over the touched part of A_half instead of 19 ns of memory-side atomic transactions per point; results agree
 * with the atomic form up to the order of the fp additions. */
int64_t wiski_scatter_bin_bytes(const wiski_grid* grid, int64_t n, int32_t elem_size);
int wiski_scatter_stats_step_f32(const wiski_grid* grid, const float* d_x, const float* d_y, const float* d_wa, const float* d_wb, const float* d_noise, int64_t n, float* d_b, float* d_A_half, float* d_cnt, const float* d_u, float* d_res, float* d_mean_out, double* d_stats, int32_t* d_err, void* z1, int64_t n1_bytes, void* z2, int64_t n2_bytes, const void* d_guard, int64_t guard_expect, void* d_bin, int64_t bin_bytes, void* stream);
int wiski_scatter_stats_step_f64(const wiski_grid* grid, const double* d_x, const double* d_y, const double* d_wa, const double* d_wb, const double* d_noise, int64_t n, double* d_b, double* d_A_half, double* d_cnt, const double* d_u, double* d_res, double* d_mean_out, double* d_stats, int32_t* d_err, void* z1, int64_t n1_bytes, void* z2, int64_t n2_bytes, const void* d_guard, int64_t guard_expect, void* d_bin, int64_t bin_bytes, void* stream);
int wiski_gather_zero_f32(const wiski_grid* grid, const float* d_x, int64_t n, const float* d_V, int32_t k, float* d_out, int32_t* d_err, void* z1, int64_t n1_bytes, void* z2, int64_t n2_bytes, int32_t* zeroed, void* stream);
int wiski_gather_zero_f64(const wiski_grid* grid, const double* d_x, int64_t n, const double* d_V, int32_t k, double* d_out, int32_t* d_err, void* z1, int64_t n1_bytes, void* z2, int64_t n2_bytes, int32_t* zeroed, void* stream);

/* One streaming step in one call (single output, symmetric half stencil): the predictive mean of the incoming batch under
 * the CURRENT posterior mean d_U (written to d_mean_out [q]; skipped when NULL) and the absorb of the q points into b /
 * A_half / cnt / stats (with carry != 0 the residual d_R is kept equal to b - Z - A U) -- one kernel,
 * wiski_scatter_stats_step -- and wiski_pcg (refresh of (d_U, d_Z) for RHS = d_b, warm = 2 when carry != 0 else 1) are queued
 * back to back on `stream` -- what BFN:204-210, BFN:258-273 and BFN:368-383 do in three Python calls.  d_wa / d_wb /
 * d_noise [q]: the per-point weights 1/clamp(noise,1e-7), 1/noise and the noise itself (ones for unit noise).  The struct
 * carries the model-resident pointers and solver parameters (meaning as in wiski_pcg); first_check / h_iters / h_relres /
 * h_err as in wiski_pcg.
 * handle != NULL selects the deferred form: the call first RESUMEs the solve a previous call started (its iteration count,
 * residual and out-of-grid flag land in h_iters / h_relres / h_err; h_resumed = 1), then queues the absorb of the new
 * batch, then STARTs the new solve (defer != 0: returns WISKI_PENDING) or runs it to convergence (defer == 0: h_iters /
 * h_relres / h_err then describe THIS solve and h_resumed = 2 says a pending one was finished on the way).  q = 0 with
 * defer = 0 just finishes a pending solve.  So the host-language work between two steps overlaps the GPU's CG iterations.
 * (With a mean requested the absorb is in fact queued BEFORE the RESUME, guarded on the device by the verdict of the pending
 * poll -- wiski_pcg_async_guard -- and queued again unguarded only if that verdict was "not converged" or "error flag set";
 * the results are those of the order described above.) */
typedef struct wiski_stream_args_f32 {
  float* d_A_half; float* d_b; float* d_cnt; double* d_stats; int32_t* d_err;     /* statistics + out-of-grid flag      */
  float* d_U; float* d_Z; float* d_R;                                              /* posterior-mean state (in place)    */
  const float* d_tcol; float kscale; const float* d_evec; const float* d_evec2; const float* d_eval; float shift;
  double tol; int32_t max_iter; int32_t check_every; void* d_work; int64_t work_bytes;
  void* d_bin; int64_t bin_bytes;                 /* optional binning workspace of the absorb (wiski_scatter_bin_bytes), or NULL / 0 */
  const wiski_shard* shard;                      /* NULL, or: this replica owns a share of the half stencil (see wiski_shard) */
  const wiski_twolevel* two_level;               /* NULL, or: exact block on the dominant modes in the preconditioner (see wiski_twolevel) */
} wiski_stream_args_f32;
typedef struct wiski_stream_args_f64 {
  double* d_A_half; double* d_b; double* d_cnt; double* d_stats; int32_t* d_err;
  double* d_U; double* d_Z; double* d_R;
  const double* d_tcol; double kscale; const double* d_evec; const double* d_evec2; const double* d_eval; double shift;
  double tol; int32_t max_iter; int32_t check_every; void* d_work; int64_t work_bytes;
  void* d_bin; int64_t bin_bytes;
  const wiski_shard* shard;
  const wiski_twolevel* two_level;               /* must be NULL (the two-level block exists for the fused fp32 path only) */
} wiski_stream_args_f64;
int wiski_stream_step_f32(const wiski_grid* grid, const wiski_stream_args_f32* args, const float* d_x, const float* d_y, const float* d_wa, const float* d_wb, const float* d_noise, int64_t q, float* d_mean_out, int32_t carry, int32_t first_check, int32_t* h_iters, double* h_relres, int32_t* h_err, void* stream, wiski_pcg_async* handle, int32_t defer, int32_t* h_resumed);
int wiski_stream_step_f64(const wiski_grid* grid, const wiski_stream_args_f64* args, const double* d_x, const double* d_y, const double* d_wa, const double* d_wb, const double* d_noise, int64_t q, double* d_mean_out, int32_t carry, int32_t first_check, int32_t* h_iters, double* h_relres, int32_t* h_err, void* stream, wiski_pcg_async* handle, int32_t defer, int32_t* h_resumed);

/* Several independent outputs in ONE absorb launch (BFN:37-55 carries num_outputs as a batch dimension; the Dirichlet classifier has 2):
 * as wiski_scatter_stats_cnt on the half stencil, with output o reading d_y + o * y_stride, weights at + o * w_stride (0: one weight
 * vector shared by all outputs) and accumulating into d_b / d_cnt / d_res (+ d_u) at + o * m, d_A_half + o * A_stride, d_stats + 2 o.
 * A point outside the grid is dropped for every output and counted once. */
int wiski_scatter_stats_multi_f32(const wiski_grid* grid, const float* d_x, const float* d_y, const float* d_wa, const float* d_wb, const float* d_noise, int64_t n, int32_t nout, int64_t y_stride, int64_t w_stride, float* d_b, float* d_A_half, int64_t A_stride, float* d_cnt, const float* d_u, float* d_res, double* d_stats, int32_t* d_err, void* stream);
int wiski_scatter_stats_multi_f64(const wiski_grid* grid, const double* d_x, const double* d_y, const double* d_wa, const double* d_wb, const double* d_noise, int64_t n, int32_t nout, int64_t y_stride, int64_t w_stride, double* d_b, double* d_A_half, int64_t A_stride, double* d_cnt, const double* d_u, double* d_res, double* d_stats, int32_t* d_err, void* stream);

/* The two halves of a stencil-sharded step on their own (wiski_stream_step uses them when args->shard is set): the absorb
 * restricted to the stencil groups [g_lo, g_hi) (same arguments as wiski_scatter_stats_step; always the atomic form), and
 * wiski_pcg_async with every A . v product summed over the ranks of `shard`. */
int wiski_scatter_stats_step_sharded_f32(const wiski_grid* grid, const float* d_x, const float* d_y, const float* d_wa, const float* d_wb, const float* d_noise, int64_t n, float* d_b, float* d_A_half, float* d_cnt, const float* d_u, float* d_res, float* d_mean_out, double* d_stats, int32_t* d_err, void* z1, int64_t n1_bytes, void* z2, int64_t n2_bytes, const void* d_guard, int64_t guard_expect, int32_t g_lo, int32_t g_hi, void* stream);
int wiski_scatter_stats_step_sharded_f64(const wiski_grid* grid, const double* d_x, const double* d_y, const double* d_wa, const double* d_wb, const double* d_noise, int64_t n, double* d_b, double* d_A_half, double* d_cnt, const double* d_u, double* d_res, double* d_mean_out, double* d_stats, int32_t* d_err, void* z1, int64_t n1_bytes, void* z2, int64_t n2_bytes, const void* d_guard, int64_t guard_expect, int32_t g_lo, int32_t g_hi, void* stream);
int wiski_pcg_sharded_f32(const wiski_grid* grid, const float* d_A_st, const float* d_tcol, float kscale, const float* d_evec, const float* d_evec2, const float* d_eval, float shift, const float* d_RHS, int32_t k, float* d_U, float* d_Z, int32_t warm, double tol, int32_t max_iter, int32_t check_every, int32_t first_check, void* d_work, int64_t work_bytes, int32_t* h_iters, double* h_relres, const int32_t* d_err, int32_t* h_err, int32_t a_sym, float* d_R, void* stream, wiski_pcg_async* handle, int32_t mode, const wiski_shard* shard);
int wiski_pcg_sharded_f64(const wiski_grid* grid, const double* d_A_st, const double* d_tcol, double kscale, const double* d_evec, const double* d_evec2, const double* d_eval, double shift, const double* d_RHS, int32_t k, double* d_U, double* d_Z, int32_t warm, double tol, int32_t max_iter, int32_t check_every, int32_t first_check, void* d_work, int64_t work_bytes, int32_t* h_iters, double* h_relres, const int32_t* d_err, int32_t* h_err, int32_t a_sym, double* d_R, void* stream, wiski_pcg_async* handle, int32_t mode, const wiski_shard* shard);
/* wiski_pcg_async / wiski_pcg_sharded with the two-level block (two_level may be NULL: then exactly wiski_pcg_sharded).
 * two_level != NULL needs the fused path: d = 3, every g_q <= 64, eigen tables given, k = 1 or 2 <= k <= two_level->mc_cols;
 * f64: WISKI_E_BADARG. */
int wiski_pcg_twolevel_f32(const wiski_grid* grid, const float* d_A_st, const float* d_tcol, float kscale, const float* d_evec, const float* d_evec2, const float* d_eval, float shift, const float* d_RHS, int32_t k, float* d_U, float* d_Z, int32_t warm, double tol, int32_t max_iter, int32_t check_every, int32_t first_check, void* d_work, int64_t work_bytes, int32_t* h_iters, double* h_relres, const int32_t* d_err, int32_t* h_err, int32_t a_sym, float* d_R, void* stream, wiski_pcg_async* handle, int32_t mode, const wiski_shard* shard, const wiski_twolevel* two_level);
/* Refresh of the block in ONE call, everything queued on `stream`: G (r x r fp64, caller-owned, running sum) += F^T F with
 * F = diag(d_scale) W(d_x) X_S for the n points absorbed since the last refresh (d_scale [n] = sqrt of the per-point weights, or
 * NULL for unit weights; d_V / kw / d_S: the per-dim eigenvector tables [g_q][kw] and index set [3][r] of X_S as for
 * wiski_basis_project), then d_N (r x r fp32) = (D_S^-1 + gscale G)^-1 with D_S = kscale * d_lam_unit, through the Cholesky factor
 * and inverse of C = I + (gscale D_S)^1/2 G (gscale D_S)^1/2.  gscale = 1: the block for the statistics as they are; > 1: for a
 * stream expected to have grown by that factor while the block is in use (the block is applied some steps after it was computed
 * and until the next one arrives; for a stationary stream G grows in proportion to the absorbed weight).  n = 0 just re-derives
 * N from G.  d_work: scratch of wiski_twolevel_refresh_workspace_bytes(r) bytes.  The verdict travels without a launch of the caller's:
 * *d_bad (device or pinned host int32, zeroed by the caller; may be NULL) |= 1 if the factorisation failed or N has a non-finite entry
 * (N is then poisoned with NaNs), |= 2 if *d_sticky != 0 (d_sticky: the time-out word wiski_twolevel.d_cs + r of the block, or NULL). */
int64_t wiski_twolevel_refresh_workspace_bytes(int32_t r);
int wiski_twolevel_refresh_f32(const wiski_grid* grid, const float* d_x, int64_t n, const float* d_scale, const double* d_V, int32_t kw, const int32_t* d_S, int32_t r, const double* d_lam_unit, double kscale, double gscale, double* d_G, void* d_work, int64_t work_bytes, float* d_N, const uint64_t* d_sticky, int32_t* d_bad, void* stream);
/* One application of the fused preconditioner on its own (what a CG iteration does to its residual; test and tooling entry):
 * d_y = P r, d_t = Kt^-1 P r, *d_rho (device double) += r . P r, for one m-vector d_r; d_w0 (m reals) and d_w1 (2 m reals)
 * are scratch.  two_level as above or NULL.  d = 3, every g_q <= 64, g_1 g_2 % 4 == 0. */
int wiski_precond_apply_f32(const wiski_grid* grid, const float* d_evec, const float* d_evec2, const float* d_eval, float kscale, float shift, const float* d_r, float* d_w0, float* d_w1, float* d_y, float* d_t, double* d_rho, const wiski_twolevel* two_level, void* stream);
/* The same for k grid vectors at once (the multi-column kernels of the 64-column variance / probe solves): d_r, d_y, d_t [k][m],
 * d_w0 k m reals, d_w1 2 k m reals, d_rho [k] doubles (+=).  two_level with k > 1 needs its d_mc scratch (k <= mc_cols). */
int wiski_precond_apply_cols_f32(const wiski_grid* grid, const float* d_evec, const float* d_evec2, const float* d_eval, float kscale, float shift, const float* d_r, int32_t k, float* d_w0, float* d_w1, float* d_y, float* d_t, double* d_rho, const wiski_twolevel* two_level, void* stream);

/* Dense Woodbury-factor path for small grids (the reference's own regime, m <=
 * max_cholesky_size): a10 `Q = I + L^T Kuu L` GEMM (BFN:350-355), a12 Cholesky
 * solve `Q.inv_matmul` (BFN:375), a13 `pred_cov` (BFN:399-403), BWM:27 logdet.
 * Row-major matrices with explicit leading dimensions, all DEVICE pointers.
 *   wiski_gemm   C[M,N] = alpha op(A) op(B) + beta C   (ta/tb != 0: transposed operand), MFMA
 *   wiski_potrf  in-place lower Cholesky (upper triangle zeroed); *d_info |= 1 on a
 *                non-positive pivot (caller adds jitter and retries, like psd_safe_cholesky)
 *                n <= 512: ONE launch (cooperating workgroups: serial chain on one, one owner wave per trailing tile; 32-wide
 *                panels in LDS, MFMA trailing updates); 512 < n: two-level blocked -- diagonal blocks of 256..448 rows through that
 *                launch (which also inverts them), panel and trailing update as two MFMA GEMMs per block
 *   wiski_potrf_inverse  the same factorisation plus the explicit inverse X = L^-1 (n x n, ldx; n <= 480: the same launch, one
 *                workgroup per 32-column block trailing the factor; beyond: the diagonal blocks' inverses + two GEMMs per block
 *                row) -- what a consumer wants that solves against the factor many times (the spectral and the dense Woodbury
 *                factor: mean, variances, MLL terms are then single GEMM / GEMV launches)
 *   wiski_trsm   in-place solve  L X = B (trans = 0)  or  L^T X = B (trans = 1), B is n x nrhs
 *   wiski_logdiag  *d_out += sum_i log A[i,i]   (double) */
int wiski_gemm_f32(int32_t ta, int32_t tb, int32_t M, int32_t N, int32_t K, float alpha, const float* d_A, int32_t lda, const float* d_B, int32_t ldb, float beta, float* d_C, int32_t ldc, void* stream);
int wiski_gemm_f64(int32_t ta, int32_t tb, int32_t M, int32_t N, int32_t K, double alpha, const double* d_A, int32_t lda, const double* d_B, int32_t ldb, double beta, double* d_C, int32_t ldc, void* stream);
int wiski_potrf_f32(int32_t n, float* d_A, int32_t lda, int32_t* d_info, void* stream);
int wiski_potrf_f64(int32_t n, double* d_A, int32_t lda, int32_t* d_info, void* stream);
int wiski_potrf_inverse_f32(int32_t n, float* d_A, int32_t lda, float* d_X, int32_t ldx, int32_t* d_info, void* stream);
int wiski_potrf_inverse_f64(int32_t n, double* d_A, int32_t lda, double* d_X, int32_t ldx, int32_t* d_info, void* stream);
int wiski_trsm_f32(int32_t trans, int32_t n, int32_t nrhs, const float* d_L, int32_t ldl, float* d_B, int32_t ldb, void* stream);
int wiski_trsm_f64(int32_t trans, int32_t n, int32_t nrhs, const double* d_L, int32_t ldl, double* d_B, int32_t ldb, void* stream);
int wiski_logdiag_f32(int32_t n, const float* d_A, int32_t lda, double* d_out, void* stream);
int wiski_logdiag_f64(int32_t n, const double* d_A, int32_t lda, double* d_out, void* stream);

/* The dense regime's posterior factor in ONE call (lazy/dense_woodbury.py; BFN:343-404 with Kt^(1/2) as the root, BWM:27 for the logdet):
 *   G = Kt^(1/2) (the Kronecker eigenbasis d_evec / d_eval of wiski_kron_spectral_mm applied to the identity),  B = I + sym(G A G),
 *   B = C C^T with X = C^-1 (wiski_potrf_inverse),  T = X G,  M = T^T T = (Kt^-1 + A)^-1,  *d_logdiag += sum_i log C[i,i].
 * d_A_half: the native half stencil; d_work: 4 m^2 reals; d_chol, d_M: [m, m] row-major outputs; *d_info |= 1 on a non-positive pivot (the
 * caller escalates jitter as psd_safe_cholesky does).  m <= 4096.  All launches are queued on `stream`; nothing is read back. */
int wiski_dense_factor_f32(const wiski_grid* grid, const float* d_A_half, const float* d_evec, const float* d_eval, float kscale, float* d_work, int64_t work_elems, float* d_chol, float* d_M, double* d_logdiag, int32_t* d_info, void* stream);
int wiski_dense_factor_f64(const wiski_grid* grid, const double* d_A_half, const double* d_evec, const double* d_eval, double kscale, double* d_work, int64_t work_elems, double* d_chol, double* d_M, double* d_logdiag, int32_t* d_info, void* stream);

/* a6 -- replaces UpdatedRootLazyTensor.collect_vector (URLT:69-119): in-place rank-q update of a root /
 * inverse-root pair.  On entry L L^T = A and R^T L = I (R = L^-T), both [m][r] row-major with leading
 * dimensions ldl / ldr; V [m][q] (ldv) holds the new columns (W^T scaled by 1/sqrt(noise), BFN:163-168).  On
 * return L L^T = A + V V^T and R^T L = I.  O(m r q) on the MFMA GEMM through the thin factor of
 * p = R^T V (the reference builds a full r x r U: O(m r^2)); the q x q eigenproblem of p^T p is solved on the
 * device (one-workgroup parallel Jacobi, fp64) for q <= 32 -- the call is then asynchronous on `stream` like every
 * other entry point -- and on the host beyond (one small copy and a stream synchronisation).  L differs from the
 * reference's L U S~ by a right orthogonal factor.  d_ws: wiski_root_update_workspace_elems(m, r, q) reals of device scratch. */
int64_t wiski_root_update_workspace_elems(int32_t m, int32_t r, int32_t q);
int wiski_root_update_f32(int32_t m, int32_t r, int32_t q, float* d_L, int32_t ldl, float* d_R, int32_t ldr, const float* d_V, int32_t ldv, float* d_ws, int64_t ws_elems, void* stream);
int wiski_root_update_f64(int32_t m, int32_t r, int32_t q, double* d_L, int32_t ldl, double* d_R, int32_t ldr, const double* d_V, int32_t ldv, double* d_ws, int64_t ws_elems, void* stream);

/* a10 / a12 / a13 / a17 in a reduced Kronecker eigenbasis -- replaces the reference's rank-limited root space (BFN:343-404 with
 * gpytorch's root_decomposition capped at max_root_decomposition_size, URLT:74-76) for smooth kernels.  With
 * K_q = V_q diag(ev_q) V_q^T per dim, basis function j is b_j = kron_q V_q[:, S[q][j]]; the r x r problem
 * G = B^T A B, C = I + Lam^1/2 G Lam^1/2 runs on wiski_gemm / wiski_potrf / wiski_trsm (fp64).  Not GEMMs:
 *   wiski_basis_project   F[p][j] = scale_p * colscale_j * (W B)[p][j]  for n points (row-major, leading dimension ldf);
 *                         d_V: per-dim eigenvector tables [g_q][kmax] (fp64, row-major), concatenated over dims; d_S [d][r]:
 *                         per-dim eigenvector index of basis function j (< kmax <= 32); d_scale [n] / d_colscale [r] optional;
 *                         d_prior [n] optional (needs d_tcol [sum g], the Toeplitz columns): prod_q w_q^T K_q w_q, the prior
 *                         variance w^T Kuu w of each point -- its excess over sum_j lam_j F[p][j]^2 bounds the truncation error
 *                         of a predictive variance.  Points outside the grid give zero rows and set bit 0 of *d_err.
 *   wiski_basis_pair_reduce   D[q][a][a'] += sum over pairs (j, j') equal in every dim but q with S[q][j] = a, S[q][j'] = a'
 *                         of Wt[j][j'] * prod_{p != q} ev[p][S[p][j]]  (Wt r x r row-major, ev [d][kmax], D [d][kmax][kmax] zeroed by
 *                         the caller): V_q D_q V_q^T summed along its lag diagonals is the gradient of
 *                         sum_jj' Wt[j][j'] b_j^T (kron_q SymToeplitz(tcol_q)) b_j' w.r.t. tcol_q -- the MLL backward (BWM:19-51). */
int wiski_basis_project_f32(const wiski_grid* grid, const float* d_x, int64_t n, const double* d_V, int32_t kmax, const int32_t* d_S, int32_t r, const float* d_scale, const double* d_colscale, const double* d_tcol, double* d_F, int64_t ldf, double* d_prior, int32_t* d_err, void* stream);
int wiski_basis_project_f64(const wiski_grid* grid, const double* d_x, int64_t n, const double* d_V, int32_t kmax, const int32_t* d_S, int32_t r, const double* d_scale, const double* d_colscale, const double* d_tcol, double* d_F, int64_t ldf, double* d_prior, int32_t* d_err, void* stream);
/* h[j] += sum_p F[p][j] t_p with t_p = d_wby[p] (/ d_scale[p] when the rows of F carry that scale; NULL: none): W^T D^-1 y of a batch in the
 * basis of F (wiski_basis_project), the right-hand-side half of a streamed update of the spectral factor's statistics (BFN:160). */
int wiski_basis_absorb_h_f32(int64_t n, int32_t r, const double* d_F, int64_t ldf, const float* d_wby, const float* d_scale, double* d_h, void* stream);
int wiski_basis_absorb_h_f64(int64_t n, int32_t r, const double* d_F, int64_t ldf, const double* d_wby, const double* d_scale, double* d_h, void* stream);
/* The spectral factor's refresh after a hyper-parameter step in ONE host call: wiski_basis_eig_update_adaptive (niter = 2, resid_ok = 1e300:
 * wiski_basis_eig_update) -> wiski_basis_change -> G = T^T G_ref T -> wiski_woodbury_c -> wiski_potrf_inverse -> wiski_factor_tail, queued
 * back to back on `stream` (OSR:113-146 refreshes its caches after every optimiser step; here that is seven kernels' worth of launches and no
 * host work in between).  All products in one packed fp64 buffer d_out of off[14] doubles; off[0..13] (wiski_factor_refresh_layout) are the
 * offsets of Vout, ev, resid, Tq, TS, lam_kuu, GT, G, chol (C factorised in place), sqG, lam, sq, Linv, tail (as wiski_factor_tail's d_out).
 * nV = sum_q g_q kw; d_work / d_verdict as wiski_basis_change; d_info as wiski_potrf_inverse; verdict_event: NULL or a hipEvent_t recorded on
 * `stream` right behind the change of basis (the verdict can then be read without waiting for the factorisation). */
int wiski_factor_refresh_layout(int32_t d, int64_t nV, int32_t kw, int32_t r_ref, int32_t r, int64_t* off);
int wiski_factor_refresh(int32_t d, const int32_t* d_g, int64_t nV, const double* d_tcol, const double* d_Vin, int32_t kw, int32_t kuse, const double* d_Vref, int32_t kref, int32_t niter, double resid_ok, const int32_t* d_Sref, const int32_t* d_S, int32_t r_ref, int32_t r, double* d_work, double* d_verdict, const double* d_Gref, const double* d_href, double kscale, int32_t* d_info, double* d_out, void* verdict_event, void* stream);
int wiski_basis_pair_reduce(int32_t d, int32_t r, int32_t kmax, const double* d_Wt, const int32_t* d_S, const double* d_ev, double* d_D, void* stream);
/* Dominant eigenvectors of the d symmetric-Toeplitz factors after a small change of their first columns, refined ON THE DEVICE from
 * the previous ones (no host eigh, no device-to-host copy): d_tcol [sum g] the new columns, d_Vin / d_Vout per-dim tables [g_q][kw]
 * (row-major, concatenated; kw even, <= 32; g_q <= 64; d_g [d] on the device), d_ev [d][kw] Ritz values (descending), d_resid [d]
 * the largest residual |K v - theta v|_inf / theta_1 over the first kuse vectors.  Two subspace-iteration steps + Rayleigh-Ritz
 * (parallel Jacobi), fp64, one workgroup per dim. */
int wiski_basis_eig_update(int32_t d, const int32_t* d_g, const double* d_tcol, const double* d_Vin, int32_t kw, int32_t kuse, double* d_Vout, double* d_ev, double* d_resid,
                           const double* d_Vref, int32_t kref, double* d_Tq, void* stream);   /* d_Vref / d_Tq (both or neither): also T_q = Vref_q^T Vnew_q, [d][32][32] */
/* The same refresh, adaptive: Rayleigh-Ritz in the span of the previous vectors (after niter steps of subspace iteration, normally 0 -- one Adam step
 * leaves the new eigenvectors inside that span to ~1e-14); only if the relative residual exceeds resid_ok a second pass with two steps re-centres the span. */
int wiski_basis_eig_update_adaptive(int32_t d, const int32_t* d_g, const double* d_tcol, const double* d_Vin, int32_t kw, int32_t kuse, double* d_Vout, double* d_ev, double* d_resid, const double* d_Vref, int32_t kref, double* d_Tq, int32_t niter, double resid_ok, void* stream);
/* Companion of wiski_basis_eig_update, one launch: the Kronecker-structured change of basis d_TS [r_ref, r] from the reference basis
 * (index set d_Sref [d, r_ref]) to the refreshed one (d_S [d, r]) given d_Tq [d][32][32] = Vref_q^T Vnew_q from wiski_basis_eig_update, the eigenvalues d_lam [r]
 * of Kuu on the kept index set, and d_verdict [3] = { max eigen-residual, trace fraction the index set leaves out, eigenvalue-weighted
 * defect of the reference span }.  d_work: r + 1 doubles, ZERO on first use (the kernel leaves it zero). */
int wiski_basis_change(int32_t d, const int32_t* d_g, int32_t kref, int32_t kw, int32_t r_ref, int32_t r, const double* d_Tq, const int32_t* d_Sref, const int32_t* d_S, const double* d_ev, const double* d_tcol, const double* d_resid, double* d_TS, double* d_lam, double* d_work, double* d_verdict, void* stream);
/* d_out [sum g] = scale * lag sums of V_q D_q V_q^T (d_V tables [g_q][kw], d_D [d][kw][kw] from wiski_basis_pair_reduce): the gradient
 * w.r.t. the Toeplitz columns, one launch; g_q <= 64. */
int wiski_basis_lag_grad(int32_t d, const int32_t* d_g, int32_t kw, const double* d_V, const double* d_D, double scale, const double* d_scale, double* d_out, void* stream);   /* d_scale != NULL: the scale is read from the device (captured graphs) */
/* d_diag[j] = |d_Y[:, j]|^2 (d_Y [r, n]), d_tail[j] = max(d_prior[j] * kscale - |d_F[j, :]|^2, 0) (d_F [n, r]): the two parts of the
 * predictive variances of n queries from the spectral factor, one launch. */
int wiski_spectral_var(int32_t n, int32_t r, const double* d_Y, const double* d_F, const double* d_prior, double kscale, double* d_diag, double* d_tail, void* stream);
/* ---- the Adam step on the MLL for the standard parameterisation, without the framework's autograd (csrc/hyper_step.hip; recorded into a
 * HIP graph by models/_graphed_step.py).  Replaces, for (Scale of)* RBF | Matern kernels with a homoskedastic second noise, what
 * /root/reference/online_gp/models/online_ski_regression.py:135-147 does through torch autograd + torch.optim.Adam. */
#define WISKI_HYPER_MAX_PARAMS 6
typedef struct {
  void* raw;          /* the raw (unconstrained) parameter, numel elements of the parameter dtype */
  void* exp_avg;      /* Adam's first / second moments (parameter dtype) and step counter(s) (fp32; step_numel = 1 or numel) */
  void* exp_avg_sq;
  void* step;
  int32_t numel;
  int32_t step_numel;
  int32_t role;       /* 0 lengthscale (numel 1 or d), 1 a factor of the output scale, 2 the second noise sigma2 */
  int32_t kind;       /* 0: value = lower + softplus(raw); 1: value = lower + (upper - lower) sigmoid(raw) */
  double lower, upper;
} wiski_hyper_param;
typedef struct {
  int32_t count;
  int32_t reserved;
  wiski_hyper_param p[WISKI_HYPER_MAX_PARAMS];
} wiski_hyper_plan;
/* d_ell [1 or d], d_scale [1] (NULL without output scales), d_s2 [1] (and d_s2_f64, may be NULL) from the raw parameters; with d_tcol64 != NULL also
 * the Toeplitz columns [sum g] (kind 0 RBF, 1-3 Matern 1/2, 3/2, 5/2) in fp64 and, if d_tcol != NULL, in the parameter dtype. */
int wiski_hyper_columns_f32(const wiski_hyper_plan* plan, const wiski_grid* grid, int32_t kind, float* d_ell, float* d_scale, float* d_s2, double* d_s2_f64, double* d_tcol64, float* d_tcol, void* stream);
int wiski_hyper_columns_f64(const wiski_hyper_plan* plan, const wiski_grid* grid, int32_t kind, double* d_ell, double* d_scale, double* d_s2, double* d_s2_f64, double* d_tcol64, double* d_tcol, void* stream);
/* d_out [9] = { val, coef0, coef1, coef2, g = -1/n, g coef0, g coef1, loss = -val/n, 1/s2 } (wiski_mll_value's arithmetic; d_n the data count on the
 * device); d_loss (may be NULL) receives the loss as well. */
int wiski_hyper_mid_f32(const double* d_bMb, const double* d_logdet, const float* d_s2, const double* d_c, const double* d_ld, const double* d_n, double* d_out, double* d_loss, void* stream);
int wiski_hyper_mid_f64(const double* d_bMb, const double* d_logdet, const double* d_s2, const double* d_c, const double* d_ld, const double* d_n, double* d_out, double* d_loss, void* stream);
/* Chain rule to the raw parameters (d_gell, d_gscale from wiski_stationary_columns_grad, sigma2's from d_mid / d_gkap) and torch.optim.Adam's update
 * (no weight decay, no amsgrad) of every parameter of the plan, its moments and step counters. */
int wiski_hyper_adam_f32(const wiski_hyper_plan* plan, const float* d_scale, const float* d_s2, const float* d_gell, const float* d_gscale, const double* d_mid, const double* d_gkap, const double* d_n, double lr, double beta1, double beta2, double eps, void* stream);
int wiski_hyper_adam_f64(const wiski_hyper_plan* plan, const double* d_scale, const double* d_s2, const double* d_gell, const double* d_gscale, const double* d_mid, const double* d_gkap, const double* d_n, double lr, double beta1, double beta2, double eps, void* stream);
/* count <= 12 fp64 segments dst[s][0 .. n[s]) = src[s][..] and one scalar store, in ONE launch (the staging of a factor state into the static
 * buffers of a captured hyper-parameter step: models/_graphed_step.py). */
#define WISKI_COPY_MAX_SEGMENTS 12
typedef struct {
  const void* src[WISKI_COPY_MAX_SEGMENTS];
  void* dst[WISKI_COPY_MAX_SEGMENTS];
  int64_t n[WISKI_COPY_MAX_SEGMENTS];
  int32_t count;
  int32_t reserved;
  double scalar;        /* written to scalar_dst[0] when scalar_dst != NULL */
  void* scalar_dst;
} wiski_copy_plan;
int wiski_multi_copy_f64(const wiski_copy_plan* plan, void* stream);
/* evaluate() of n <= 64 queries from the spectral factor (r <= 1024) in one launch (the reference loop scores every batch before absorbing it,
 * /root/reference/online_gp/models/online_ski_regression.py:56-78): d_F [n, r] = W B Lam^1/2 and d_prior [n] from wiski_basis_project,
 * d_Linv = chol^-1 [r, r] (ld ldl), d_t [r] = chol^-T chol^-1 Lam^1/2 h, d_s2 [1] the observation noise, d_y [n] the targets, d_err the
 * out-of-grid flag (may be NULL).  d_out (fp64 [4]) = { rmse, mean Gaussian nll, flag, max |mean| }; d_mean / d_var (may be NULL): the
 * means and LATENT variances in the data dtype.  d_ws: 200 doubles, zero on first use (left zero). */
int wiski_spectral_evaluate_f32(int32_t n, int32_t r, const double* d_F, const double* d_prior, const double* d_Linv, int32_t ldl, const double* d_t, double kscale, const float* d_s2, const float* d_y, const int32_t* d_err, double* d_ws, double* d_out, float* d_mean, float* d_var, void* stream);
int wiski_spectral_evaluate_f64(int32_t n, int32_t r, const double* d_F, const double* d_prior, const double* d_Linv, int32_t ldl, const double* d_t, double kscale, const double* d_s2, const double* d_y, const int32_t* d_err, double* d_ws, double* d_out, double* d_mean, double* d_var, void* stream);
/* The same for a larger batch (any n; one evaluate() chunk is <= 1024): d_Y [r, n] = chol^-1 d_F^T from wiski_gemm_f64; same outputs and workspace. */
int wiski_spectral_evaluate_y_f32(int32_t n, int32_t r, const double* d_Y, const double* d_F, const double* d_prior, const double* d_t, double kscale, const float* d_s2, const float* d_y, const int32_t* d_err, double* d_ws, double* d_out, float* d_mean, float* d_var, void* stream);
int wiski_spectral_evaluate_y_f64(int32_t n, int32_t r, const double* d_Y, const double* d_F, const double* d_prior, const double* d_t, double kscale, const double* d_s2, const double* d_y, const int32_t* d_err, double* d_ws, double* d_out, double* d_mean, double* d_var, void* stream);
/* After the factorisation, three launches: d_out (packed fp64, 6 r + 2) = hr [r] | c [r] | t [r] | coef [r] | zeta [r] | bMb | logdet | scratch [r]
 * with hr = T^T h_ref (d_TS [r_ref, r]), c = chol^-1 (sq o hr) (d_Linv = chol^-1, [r, r]), bMb = |c|^2, t = chol^-T c, coef = sq o t,
 * zeta = t / sq, logdet = 2 sum log diag d_chol. */
int wiski_factor_tail(int32_t r_ref, int32_t r, const double* d_TS, const double* d_href, const double* d_sq, const double* d_Linv, const double* d_chol, double* d_out, void* stream);
/* C = I + Lam^1/2 G Lam^1/2, lam = lam_kuu * kscale, sq = sqrt(lam)  (r x r, contiguous): the matrix the spectral factor factorises. */
int wiski_woodbury_c(int32_t r, const double* d_G, const double* d_lam_kuu, double kscale, double* d_C, double* d_lam, double* d_sq, double* d_sqG, void* stream);   /* d_sqG (may be NULL): Lam^1/2 G */
/* MLL backward: d_Wt = g_ld (G - P) + g_b zeta zeta^T (r x r), d_gkap = sum_i Wt[i,i] lam_kuu[i]; d_gb, d_gld device scalars.  One launch. */
int wiski_mll_weights(int32_t r, const double* d_G, const double* d_P, const double* d_zeta, const double* d_lam_kuu, const double* d_gb, const double* d_gld, double* d_Wt, double* d_gkap, void* stream);

/* (e) -- the one collective of the path: in-place RCCL all-reduce(SUM), grouped into one launch, of the
 * statistics that are sums over data points: the half-stencil delta of W^T D^-1 W (n_half reals), W^T D^-1 y (n_b), the
 * row-sum vector W^T D^-1 1 (n_cnt) and fp64 scalars ([y^T D^-1 y, log|D|] per output, point count, weight sums: n_scal
 * doubles).  Any pointer may be NULL / count 0.  `comm` is an ncclComm_t owned by the caller (one rank per GPU); librccl
 * is resolved with dlopen at the first call (WISKI_RCCL_LIB overrides its name), so the library has no link-time
 * dependency on it.  wiski_comm_* are thin bootstrap helpers for hosts without their own RCCL setup: rank 0 creates
 * wiski_comm_unique_id_bytes() bytes with wiski_comm_unique_id, ships them to the other ranks by any means, and every
 * rank calls wiski_comm_init_rank.  The reference has no distributed code. */
int wiski_comm_unique_id_bytes(void);
int wiski_comm_unique_id(void* id_out);
int wiski_comm_init_rank(const void* id_bytes, int32_t nranks, int32_t rank, void** comm_out);
int wiski_comm_destroy(void* comm);
/* What the loaded librccl reports: its version code (ncclGetVersion; e.g. 22203) and, for a communicator, the number of ranks
 * it joins and this rank's index (comm may be NULL: version only; any out pointer may be NULL).  bench.py prints these in
 * its JSON line so that a multi-GPU number says which RCCL carried it and how many ranks the communicator really saw. */
int wiski_comm_info(void* comm, int32_t* version_out, int32_t* nranks_out, int32_t* rank_out);
int wiski_allreduce_stats_f32(void* comm, float* d_half, int64_t n_half, float* d_b, int64_t n_b, float* d_cnt, int64_t n_cnt, double* d_scal, int64_t n_scal, void* stream);
int wiski_allreduce_stats_f64(void* comm, double* d_half, int64_t n_half, double* d_b, int64_t n_b, double* d_cnt, int64_t n_cnt, double* d_scal, int64_t n_scal, void* stream);

/* Value of a device int32 flag after everything already queued on `stream`, without a stream
 * synchronisation (a one-thread publish kernel + a host spin on pinned memory, a few microseconds once the
 * queue has drained).  Used for the out-of-grid flag after a query gather, where the reference raises
 * immediately (gpytorch's grid bounds check behind BFN:205). */
int wiski_read_flag(const int32_t* d_flag, int32_t* h_value, void* stream);

/* Measurement hook (bench.py roofline leg): every half-stencil SpMV launch gets a start/stop event pair
 * attached to its own dispatch packet (hipExtLaunchKernel), i.e. the kernel's execution time as rocprofv3
 * reports it, on the stream it runs on.  wiski_prof_stop returns the summed kernel time and launch count;
 * synchronise the stream before calling it. */
int wiski_prof_start(int32_t max_launches);
int wiski_prof_stop(double* total_ms, int64_t* launches);
/* Stamps taken INSIDE the kernel (k_spmv_sym_dma: 100 MHz wall clock, earliest wave start / latest wave end of each dispatch):
 * summed [first wave started -> last wave finished] time and the number of stamped dispatches.  EVERY such dispatch between
 * wiski_prof_start and wiski_prof_stop is stamped (no measurable cost), with or without the events of wiski_prof_enable.  The
 * event pair of wiski_prof_stop brackets [predecessor complete -> kernel complete] and so contains the dispatch latency in
 * front of the first wave (~2.4 us); this figure is the kernel alone.  Readable until the next wiski_prof_start, stream
 * synchronised. */
int wiski_prof_stamps(double* total_ms, int64_t* launches, double* each_us, int64_t each_cap);   /* each_us (may be NULL): the first each_cap dispatches one by one, microseconds */
/* the raw per-wave pairs of recorded dispatch i (tools/stamp_report.py): pairs[2w] = start | placement << 48 (simd(2) pipe(2)
 * cu(4) sh(1) se(3) xcc(4), low to high), pairs[2w+1] = end (0: padding workgroup), w = blockIdx.y * gridDim.x + blockIdx.x;
 * *nwaves = the dispatch's wave count (0: not stamped); at most cap pairs are copied (pairs may be NULL) */
int wiski_prof_stamps_raw(int64_t i, uint64_t* pairs, int64_t cap, int64_t* nwaves);
/* between start and stop: switch the event attachment off / on again without touching what has been recorded (sample some
 * steps of a pipelined loop, read all events once the loop has drained) */
int wiski_prof_enable(int32_t on);
/* the same clock on an EMPTY dispatch (average of n launches, microseconds; synchronises the stream): the per-dispatch
 * floor contained in every kernel time measured this way */
int wiski_prof_empty(int32_t n, double* avg_us, void* stream);

/* ---- hyper-parameter step glue (hyper_columns.hip): what the framework spends ~10 tiny launches each on, as single launches ----
 * wiski_stationary_columns      d_out [sum g] (fp64) = S * k(l h_q / ell_q): the Toeplitz columns of a Scale(RBF | Matern) kernel on the
 *                               grid (gpytorch: kernel(x1, x2, last_dim_is_batch=True) on the grid points, GridKernel / GridInterpolationKernel).
 *                               kind 0 RBF, 1 / 2 / 3 Matern-1/2, -3/2, -5/2; d_ell [nell] (nell = 1 isotropic or = d ARD), d_scale [1] or NULL,
 *                               both DEVICE pointers in the model's dtype (no host read of a hyper-parameter).
 * wiski_stationary_columns_grad d_gell [nell], d_gscale [1] (or NULL) = gradient of <d_gout, columns> w.r.t. lengthscales / outputscale.
 * wiski_gaussian_metrics        d_out [2] = { sqrt(mean (mu - y)^2), mean 1/2 ((mu - y)^2 / v + log v + log 2 pi) }, v = var + d_add_var[0]
 *                               (d_add_var may be NULL): the per-batch test metrics of the reference's regression loop
 *                               (online_ski_regression.py:88-111), n points, one launch. */
int wiski_stationary_columns_f32(const wiski_grid* grid, int32_t kind, const float* d_ell, int32_t nell, const float* d_scale, double* d_out, void* stream);
int wiski_stationary_columns_f64(const wiski_grid* grid, int32_t kind, const double* d_ell, int32_t nell, const double* d_scale, double* d_out, void* stream);
int wiski_stationary_columns_grad_f32(const wiski_grid* grid, int32_t kind, const float* d_ell, int32_t nell, const float* d_scale, const double* d_gout, float* d_gell, float* d_gscale, void* stream);
int wiski_stationary_columns_grad_f64(const wiski_grid* grid, int32_t kind, const double* d_ell, int32_t nell, const double* d_scale, const double* d_gout, double* d_gell, double* d_gscale, void* stream);
/* Scalar tail of the Woodbury MLL of one output (BWM:34-47), value and gradient, device scalars in and out:
 *   d_val  = -1/2 ((c - bMb) / s2 + logdet + ld + n log(2 pi) + n log s2)  (d_logdet may be NULL), d_coef [3] = { d val / d bMb, d val / d logdet, c - bMb };
 *   d_gs2  = g (1/2 coef[2] / s2^2 - 1/2 n / s2) - g_kap / s2^2   (d_gkap: gradient w.r.t. 1 / s2 through the factor, or NULL).
 * n: the data count by value, or from the device scalar d_n when that is not NULL (calls recorded into a hipGraph: the count moves on). */
int wiski_mll_value_f32(const double* d_bMb, const double* d_logdet, const float* d_s2, const double* d_c, const double* d_ld, double n, const double* d_n, double* d_val, double* d_coef, void* stream);
int wiski_mll_value_f64(const double* d_bMb, const double* d_logdet, const double* d_s2, const double* d_c, const double* d_ld, double n, const double* d_n, double* d_val, double* d_coef, void* stream);
int wiski_mll_s2_grad_f32(const double* d_g, const double* d_coef, const float* d_s2, double n, const double* d_n, const double* d_gkap, float* d_gs2, void* stream);
int wiski_mll_s2_grad_f64(const double* d_g, const double* d_coef, const double* d_s2, double n, const double* d_n, const double* d_gkap, double* d_gs2, void* stream);
int wiski_gaussian_metrics_f32(int64_t n, const float* d_mu, const float* d_var, const float* d_y, const float* d_add_var, float* d_out, void* stream);
int wiski_gaussian_metrics_f64(int64_t n, const double* d_mu, const double* d_var, const double* d_y, const double* d_add_var, double* d_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* WISKI_H */
