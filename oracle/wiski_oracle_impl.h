/*
 * TEST INFRASTRUCTURE -- CPU oracle body, included twice by wiski_oracle.c
 * (REAL=double / SUFFIX=_f64 and REAL=float / SUFFIX=_f32).
 *
 * Plain-C, single-threaded, matrix-free restatement of the WISKI streaming
 * hot path of wjmaddox/online_gp.  Nothing here is product code: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call it.
 *
 * PARITY UNPINNED for the gpytorch-side pieces: the interpolation weights,
 * grid geometry and Kuu construction live in un-vendored gpytorch
 * (requirements.txt:7, unpinned) which is absent from this image; they are
 * restated from the published algorithm.  The WISKI
 * algebra itself (cache sums, Woodbury posterior) follows the reference
 * file:line cited at each function and is pinned by the identity
 * "WISKI == exact GP on kernel W Kuu W^T + sigma^2 D"
 * (reference tests/mlls/test_batched_woodbury_marginal_log_likelihood.py:55-73,
 *  tests/models/test_woodbury_gp_model.py:260-289).
 */

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUFFIX)

/* Keys cubic convolution kernel, a = -0.5 (gpytorch
 * Interpolation._cubic_interpolation_kernel; call site
 * online_gp/models/batched_fixed_noise_online_gp.py:143,205,261,421). */
static REAL FN(keys_cubic)(REAL s) {
  REAL a = s < 0 ? -s : s;
  if (a <= (REAL)1) return ((REAL)1.5 * a - (REAL)2.5) * a * a + (REAL)1;
  if (a < (REAL)2) return (((REAL)-0.5 * a + (REAL)2.5) * a - (REAL)4) * a + (REAL)2;
  return (REAL)0;
}

/* Per-dim 4-tap stencil for one coordinate.  Returns lowest tap index j0
 * (taps j0..j0+3) and weights w[4]; -1 when x is outside the grid.
 * Boundary rule (gpytorch Interpolation.interpolate): if floor(u)-1 < 0 or
 * > g-4, one-hot on the nearest of the first/last four grid points. */
static int FN(dim_stencil)(REAL x, REAL g0, REAL h, int g, REAL w[4]) {
  REAL u = (x - g0) / h;
  REAL fl = (REAL)floor((double)u);
  REAL t = u - fl;
  long j0 = (long)fl - 1;
  if (x < g0 || x > g0 + h * (REAL)(g - 1)) return -1;
  w[0] = FN(keys_cubic)(t + (REAL)1);
  w[1] = FN(keys_cubic)(t);
  w[2] = FN(keys_cubic)(t - (REAL)1);
  w[3] = FN(keys_cubic)(t - (REAL)2);
  if (j0 < 0 || j0 > g - 4) {
    long base = j0 < 0 ? 0 : g - 4;
    int best = 0;
    REAL bd = (REAL)1e300;
    for (int c = 0; c < 4; ++c) {
      REAL dd = g0 + h * (REAL)(base + c) - x;
      if (dd < 0) dd = -dd;
      if (dd < bd) { bd = dd; best = c; }
    }
    for (int c = 0; c < 4; ++c) w[c] = (c == best) ? (REAL)1 : (REAL)0;
    j0 = base;
  }
  return (int)j0;
}

/* a1: interpolation indices/values, T = 4^d taps per point, dim 0 slowest in
 * both the tap order and the flat grid index. */
int FN(wo_interp)(const REAL *x, long n, int d, const REAL *g0, const REAL *h,
                  const int *g, long *idx_out, REAL *val_out) {
  long T = 1;
  for (int k = 0; k < d; ++k) T *= 4;
  for (long p = 0; p < n; ++p) {
    int j0[8];
    REAL w[8][4];
    for (int k = 0; k < d; ++k) {
      j0[k] = FN(dim_stencil)(x[p * d + k], g0[k], h[k], g[k], w[k]);
      if (j0[k] < 0) return -1;
    }
    for (long a = 0; a < T; ++a) {
      long flat = 0;
      REAL v = 1;
      long rem = a;
      /* digit of dim k in base 4, dim 0 most significant */
      long div = T / 4;
      for (int k = 0; k < d; ++k) {
        int c = (int)(rem / div);
        rem -= c * div;
        div = div > 1 ? div / 4 : 1;
        flat = flat * g[k] + (j0[k] + c);
        v *= w[k][c];
      }
      idx_out[p * T + a] = flat;
      val_out[p * T + a] = v;
    }
  }
  return 0;
}

/* a14 left_interp (batched_fixed_noise_online_gp.py:206-210):
 * out[p, c] = sum_a val[p,a] * V[c, idx[p,a]],  V stored [k][m]. */
int FN(wo_gather)(const REAL *x, long n, int d, const REAL *g0, const REAL *h,
                  const int *g, const REAL *V, long m, int k, REAL *out) {
  long T = 1;
  for (int q = 0; q < d; ++q) T *= 4;
  long *idx = (long *)malloc(sizeof(long) * T);
  REAL *val = (REAL *)malloc(sizeof(REAL) * T);
  for (long p = 0; p < n; ++p) {
    if (FN(wo_interp)(x + p * d, 1, d, g0, h, g, idx, val)) { free(idx); free(val); return -1; }
    for (int c = 0; c < k; ++c) {
      REAL acc = 0;
      for (long a = 0; a < T; ++a) acc += val[a] * V[(long)c * m + idx[a]];
      out[p * k + c] = acc;
    }
  }
  free(idx);
  free(val);
  return 0;
}

/* a3/a4/a5: additive sufficient statistics
 * (batched_fixed_noise_online_gp.py:31-60 initial, :155-171 increments;
 *  updated_root_lazy_tensor.py:58 for A += V V^T).
 *   c  += sum y^2 * wb          (response_cache,  y^T D^-1 y)
 *   b  += W^T (y * wb)          (interpolation_cache)
 *   A  += W^T diag(wa) W        (WtW), stored as a block stencil:
 *         A_st[o][i] = A[i, i + off(o)],  o in 7^d relative offsets
 *   ld += sum log(noise)        (D_logdet)
 * wb = 1/noise, wa = 1/noise (init) or 1/max(noise,1e-7) (update, :163). */
int FN(wo_scatter_stats)(const REAL *x, const REAL *y, const REAL *wa,
                         const REAL *wb, const REAL *noise, long n, int d,
                         const REAL *g0, const REAL *h, const int *g, long m,
                         REAL *b, REAL *A_st, double *c_ld /* [2] */) {
  long T = 1, R = 1;
  for (int q = 0; q < d; ++q) { T *= 4; R *= 7; }
  long *idx = (long *)malloc(sizeof(long) * T);
  REAL *val = (REAL *)malloc(sizeof(REAL) * T);
  int *tapc = (int *)malloc(sizeof(int) * T * d);
  for (long a = 0; a < T; ++a) {
    long rem = a, div = T / 4;
    for (int q = 0; q < d; ++q) {
      tapc[a * d + q] = (int)(rem / div);
      rem %= div;
      div = div > 1 ? div / 4 : 1;
    }
  }
  for (long p = 0; p < n; ++p) {
    if (FN(wo_interp)(x + p * d, 1, d, g0, h, g, idx, val)) { free(idx); free(val); free(tapc); return -1; }
    c_ld[0] += (double)y[p] * (double)y[p] * (double)wb[p];
    c_ld[1] += log((double)noise[p]);
    for (long a = 0; a < T; ++a) {
      b[idx[a]] += val[a] * y[p] * wb[p];
      if (A_st) {
        REAL va = val[a] * wa[p];
        for (long bb = 0; bb < T; ++bb) {
          long o = 0;
          for (int q = 0; q < d; ++q) o = o * 7 + (tapc[bb * d + q] - tapc[a * d + q] + 3);
          A_st[o * m + idx[a]] += va * val[bb];
        }
      }
    }
  }
  (void)R;
  free(idx); free(val); free(tapc);
  return 0;
}

/* flat offsets of the 7^d stencil */
static void FN(stencil_offsets)(int d, const int *g, long *off) {
  long R = 1;
  for (int q = 0; q < d; ++q) R *= 7;
  for (long o = 0; o < R; ++o) {
    long rem = o, div = R / 7, f = 0;
    for (int q = 0; q < d; ++q) {
      int c = (int)(rem / div);
      rem %= div;
      div = div > 1 ? div / 7 : 1;
      f = f * g[q] + (c - 3);
    }
    off[o] = f;
  }
}

/* WtW @ V (updated_root_lazy_tensor.py:47-48), block-stencil form.
 * out[c][i] = beta*add[c][i] + sum_o A_st[o][i] * V[c][i+off(o)].
 * Entries whose neighbour falls outside the grid were never scattered
 * (exact zeros); the flat neighbour index is clamped so the read is legal. */
void FN(wo_stencil_spmv)(const REAL *A_st, int d, const int *g, long m,
                         const REAL *V, int k, const REAL *add, REAL beta, REAL *out) {
  long R = 1;
  for (int q = 0; q < d; ++q) R *= 7;
  long *off = (long *)malloc(sizeof(long) * R);
  FN(stencil_offsets)(d, g, off);
  for (int c = 0; c < k; ++c) {
    const REAL *v = V + (long)c * m;
    REAL *o_ = out + (long)c * m;
    for (long i = 0; i < m; ++i) o_[i] = add ? beta * add[(long)c * m + i] : (REAL)0;
    for (long o = 0; o < R; ++o) {
      const REAL *a = A_st + o * m;
      long f = off[o];
      long lo = f < 0 ? -f : 0, hi = f > 0 ? m - f : m;
      for (long i = lo; i < hi; ++i) o_[i] += a[i] * v[i + f];
    }
  }
  free(off);
}

/* a8/a9/a11: Kuu @ V with Kuu = kron_i Toeplitz(tcol_i)
 * (batched_fixed_noise_online_gp.py:334-348,363-366).  V, out stored [k][m].
 * tcol is the concatenation of the d first columns (length sum g). */
void FN(wo_kron_toeplitz_mm)(const REAL *tcol, int d, const int *g, long m,
                             const REAL *V, int k, REAL scale, REAL *out) {
  REAL *cur = (REAL *)malloc(sizeof(REAL) * m);
  REAL *nxt = (REAL *)malloc(sizeof(REAL) * m);
  for (int c = 0; c < k; ++c) {
    memcpy(cur, V + (long)c * m, sizeof(REAL) * m);
    long post = m;
    const REAL *tc = tcol;
    for (int q = 0; q < d; ++q) {
      int gq = g[q];
      post /= gq;
      long pre = m / (post * gq);
      for (long pp = 0; pp < pre; ++pp)
        for (int i = 0; i < gq; ++i) {
          REAL *dst = nxt + (pp * gq + i) * post;
          for (long s = 0; s < post; ++s) dst[s] = 0;
          for (int j = 0; j < gq; ++j) {
            REAL t = tc[i > j ? i - j : j - i];
            const REAL *src = cur + (pp * gq + j) * post;
            for (long s = 0; s < post; ++s) dst[s] += t * src[s];
          }
        }
      REAL *tmp = cur; cur = nxt; nxt = tmp;
      tc += gq;
    }
    for (long i = 0; i < m; ++i) out[(long)c * m + i] = scale * cur[i];
  }
  free(cur); free(nxt);
}

/* CG branch of a12 (batched_fixed_noise_online_gp.py:368-383), matrix-free:
 * solve (Kt^-1 + A) U = RHS with preconditioner Kt = kscale * Kuu, i.e.
 * U = (Kt^-1 + A)^-1 RHS = M RHS  (SURVEY 3.5 "inducing posterior").
 * Works on the transformed iteration (no Kt^-1 is ever applied):
 *   r = rhs - (z + A Kt z),  y = Kt r,  p = y + beta p,  pt = r + beta pt,
 *   Hp = pt + A p,  alpha = r.y / p.Hp,  U += alpha p,  r -= alpha Hp.
 * Columns are independent.  Returns max iterations used. */
int FN(wo_pcg)(const REAL *A_st, const REAL *tcol, int d, const int *g, long m,
               REAL kscale, const REAL *RHS, int k, double tol, int max_iter,
               REAL *U /* in: initial guess must be 0 */, double *rel_res_out) {
  REAL *r = (REAL *)malloc(sizeof(REAL) * m), *y = (REAL *)malloc(sizeof(REAL) * m);
  REAL *p = (REAL *)malloc(sizeof(REAL) * m), *pt = (REAL *)malloc(sizeof(REAL) * m);
  REAL *hp = (REAL *)malloc(sizeof(REAL) * m);
  int worst = 0;
  for (int c = 0; c < k; ++c) {
    REAL *u = U + (long)c * m;
    double rn0 = 0;
    for (long i = 0; i < m; ++i) { u[i] = 0; r[i] = RHS[(long)c * m + i]; rn0 += (double)r[i] * r[i]; p[i] = 0; pt[i] = 0; }
    double rho_old = 1, rn = rn0;
    int it = 0;
    if (rn0 > 0)
      for (; it < max_iter; ++it) {
        FN(wo_kron_toeplitz_mm)(tcol, d, g, m, r, 1, kscale, y);
        double rho = 0;
        for (long i = 0; i < m; ++i) rho += (double)r[i] * y[i];
        REAL beta = it == 0 ? (REAL)0 : (REAL)(rho / rho_old);
        for (long i = 0; i < m; ++i) { p[i] = y[i] + beta * p[i]; pt[i] = r[i] + beta * pt[i]; }
        FN(wo_stencil_spmv)(A_st, d, g, m, p, 1, pt, (REAL)1, hp);
        double php = 0;
        for (long i = 0; i < m; ++i) php += (double)p[i] * hp[i];
        REAL alpha = (REAL)(rho / php);
        rn = 0;
        for (long i = 0; i < m; ++i) { u[i] += alpha * p[i]; r[i] -= alpha * hp[i]; rn += (double)r[i] * r[i]; }
        rho_old = rho;
        if (sqrt(rn / rn0) < tol) { ++it; break; }
      }
    if (rel_res_out) rel_res_out[c] = rn0 > 0 ? sqrt(rn / rn0) : 0;
    if (it > worst) worst = it;
  }
  free(r); free(y); free(p); free(pt); free(hp);
  return worst;
}

#undef FN
#undef CAT
#undef CAT_
