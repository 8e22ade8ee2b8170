/* TEST INFRASTRUCTURE -- body of wiski_baseline_omp.c (included for REAL = double / float). */

static REAL FN(wb_cubic)(REAL s) {
  REAL a = s < 0 ? -s : s;
  if (a <= (REAL)1) return ((REAL)1.5 * a - (REAL)2.5) * a * a + (REAL)1;
  if (a < (REAL)2) return (((REAL)-0.5 * a + (REAL)2.5) * a - (REAL)4) * a + (REAL)2;
  return (REAL)0;
}

/* per-dim 4-tap stencil (same rule as wiski_oracle_impl.h dim_stencil); -1 outside the grid */
static int FN(wb_dim)(REAL x, REAL g0, REAL h, int g, REAL w[4]) {
  REAL u = (x - g0) / h;
  REAL fl = (REAL)floor((double)u);
  REAL t = u - fl;
  long j0 = (long)fl - 1;
  if (x < g0 || x > g0 + h * (REAL)(g - 1)) return -1;
  w[0] = FN(wb_cubic)(t + (REAL)1); w[1] = FN(wb_cubic)(t); w[2] = FN(wb_cubic)(t - (REAL)1); w[3] = FN(wb_cubic)(t - (REAL)2);
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

/* taps of one point: flat indices, values and per-dim tap digits; returns -1 outside the grid */
static int FN(wb_taps)(const REAL *xp, int d, const REAL *g0, const REAL *h, const int *g, long T, long *idx, REAL *val, int *dig) {
  int j0[4];
  REAL w[4][4];
  for (int k = 0; k < d; ++k) {
    j0[k] = FN(wb_dim)(xp[k], g0[k], h[k], g[k], w[k]);
    if (j0[k] < 0) return -1;
  }
  for (long a = 0; a < T; ++a) {
    long flat = 0, rem = a, div = T / 4;
    REAL v = 1;
    for (int k = 0; k < d; ++k) {
      int c = (int)(rem / div);
      rem -= c * div;
      div = div > 1 ? div / 4 : 1;
      flat = flat * g[k] + (j0[k] + c);
      v *= w[k][c];
      if (dig) dig[a * d + k] = c;
    }
    idx[a] = flat;
    val[a] = v;
  }
  return 0;
}

/* b += W^T (y/noise), A_st += W^T diag(1/noise) W (full 7^d block stencil), c += y^2/noise, ld += log noise */
int FN(wb_absorb)(const REAL *x, const REAL *y, const REAL *noise, long n, int d, const REAL *g0, const REAL *h, const int *g, long m,
                  REAL *b, REAL *A_st, double *c_ld) {
  long T = 1;
  for (int q = 0; q < d; ++q) T *= 4;
  int bad = 0;
  double c_acc = 0, ld_acc = 0;
#pragma omp parallel reduction(+ : c_acc, ld_acc) reduction(| : bad)
  {
    long *idx = (long *)malloc(sizeof(long) * T);
    REAL *val = (REAL *)malloc(sizeof(REAL) * T);
    int *dig = (int *)malloc(sizeof(int) * T * d);
#pragma omp for schedule(dynamic, 64)
    for (long p = 0; p < n; ++p) {
      if (FN(wb_taps)(x + p * d, d, g0, h, g, T, idx, val, dig)) { bad |= 1; continue; }
      const REAL w = (REAL)1 / noise[p];
      c_acc += (double)y[p] * (double)y[p] * (double)w;
      ld_acc += log((double)noise[p]);
      for (long a = 0; a < T; ++a) {
        const REAL vb = val[a] * y[p] * w;
#pragma omp atomic
        b[idx[a]] += vb;
        const REAL va = val[a] * w;
        for (long bb = 0; bb < T; ++bb) {
          long o = 0;
          for (int q = 0; q < d; ++q) o = o * 7 + (dig[bb * d + q] - dig[a * d + q] + 3);
          const REAL add = va * val[bb];
#pragma omp atomic
          A_st[o * m + idx[a]] += add;
        }
      }
    }
    free(idx); free(val); free(dig);
  }
  c_ld[0] += c_acc;
  c_ld[1] += ld_acc;
  return bad ? -1 : 0;
}

/* out[p] = W(x_p) . v */
int FN(wb_gather)(const REAL *x, long n, int d, const REAL *g0, const REAL *h, const int *g, const REAL *v, REAL *out) {
  long T = 1;
  for (int q = 0; q < d; ++q) T *= 4;
  int bad = 0;
#pragma omp parallel reduction(| : bad)
  {
    long *idx = (long *)malloc(sizeof(long) * T);
    REAL *val = (REAL *)malloc(sizeof(REAL) * T);
#pragma omp for schedule(static)
    for (long p = 0; p < n; ++p) {
      if (FN(wb_taps)(x + p * d, d, g0, h, g, T, idx, val, (int *)0)) { bad |= 1; out[p] = 0; continue; }
      REAL acc = 0;
      for (long a = 0; a < T; ++a) acc += val[a] * v[idx[a]];
      out[p] = acc;
    }
    free(idx); free(val);
  }
  return bad ? -1 : 0;
}

static void FN(wb_offsets)(int d, const int *g, long *off) {
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

/* out = add + A v on the full block stencil; row chunks in parallel, offsets streamed in order inside a chunk */
static void FN(wb_spmv)(const REAL *A_st, const long *off, long R, long m, const REAL *v, const REAL *add, REAL *out) {
  const long CH = 2048;
  const long nch = (m + CH - 1) / CH;
#pragma omp parallel for schedule(static)
  for (long c = 0; c < nch; ++c) {
    const long lo0 = c * CH, hi0 = lo0 + CH < m ? lo0 + CH : m;
    for (long i = lo0; i < hi0; ++i) out[i] = add ? add[i] : (REAL)0;
    for (long o = 0; o < R; ++o) {
      const REAL *a = A_st + o * m;
      const long f = off[o];
      long lo = lo0, hi = hi0;
      if (lo + f < 0) lo = -f;
      if (hi + f > m) hi = m - f;
#pragma omp simd
      for (long i = lo; i < hi; ++i) out[i] += a[i] * v[i + f];
    }
  }
}

/* out = scale * (kron_q Toeplitz(tcol_q)) v; tmp holds 2 m reals */
static void FN(wb_kron)(const REAL *tcol, int d, const int *g, long m, const REAL *v, REAL scale, REAL *tmp, REAL *out) {
  const REAL *cur = v;
  long post = m;
  const REAL *tc = tcol;
  for (int q = 0; q < d; ++q) {
    const int gq = g[q];
    post /= gq;
    const long pre = m / (post * gq);
    REAL *dst0 = (q == d - 1) ? out : (tmp + (q & 1) * m);
    const REAL sc = (q == d - 1) ? scale : (REAL)1;
    const long po = post;
#pragma omp parallel for collapse(2) schedule(static)
    for (long pp = 0; pp < pre; ++pp)
      for (int i = 0; i < gq; ++i) {
        REAL *dst = dst0 + (pp * gq + i) * po;
        for (long s = 0; s < po; ++s) dst[s] = 0;
        for (int j = 0; j < gq; ++j) {
          const REAL t = sc * tc[i > j ? i - j : j - i];
          const REAL *src = cur + (pp * gq + j) * po;
#pragma omp simd
          for (long s = 0; s < po; ++s) dst[s] += t * src[s];
        }
      }
    cur = dst0;
    tc += gq;
  }
}

/* Warm-started Kt-preconditioned CG on (Kt^-1 + A) u = rhs in the (u, z = Kt^-1 u) form of wiski_oracle_impl.h wo_pcg.
 * warm != 0: (u, z) hold the previous solution (u = Kt z); else they are zeroed.  Returns the iterations used. */
int FN(wb_pcg)(const REAL *A_st, const REAL *tcol, int d, const int *g, long m, REAL kscale, const REAL *rhs, int warm, double tol, int max_iter,
               REAL *u, REAL *z, double *rel_res_out) {
  long R = 1;
  for (int q = 0; q < d; ++q) R *= 7;
  long *off = (long *)malloc(sizeof(long) * R);
  FN(wb_offsets)(d, g, off);
  REAL *r = (REAL *)malloc(sizeof(REAL) * m), *y = (REAL *)malloc(sizeof(REAL) * m), *p = (REAL *)malloc(sizeof(REAL) * m);
  REAL *pt = (REAL *)malloc(sizeof(REAL) * m), *hp = (REAL *)malloc(sizeof(REAL) * m), *tmp = (REAL *)malloc(sizeof(REAL) * 2 * m);
  double rn0 = 0, rn = 0;
  if (!warm) {
    memset(u, 0, sizeof(REAL) * m);
    memset(z, 0, sizeof(REAL) * m);
  }
  if (warm) FN(wb_spmv)(A_st, off, R, m, u, z, hp);   /* hp = z + A u */
#pragma omp parallel for reduction(+ : rn0, rn) schedule(static)
  for (long i = 0; i < m; ++i) {
    r[i] = warm ? rhs[i] - hp[i] : rhs[i];
    p[i] = 0; pt[i] = 0;
    rn0 += (double)rhs[i] * rhs[i];
    rn += (double)r[i] * r[i];
  }
  double rho_old = 1;
  int it = 0;
  if (rn0 > 0 && sqrt(rn / rn0) >= tol)
    for (; it < max_iter;) {
      FN(wb_kron)(tcol, d, g, m, r, kscale, tmp, y);
      double rho = 0;
#pragma omp parallel for reduction(+ : rho) schedule(static)
      for (long i = 0; i < m; ++i) rho += (double)r[i] * y[i];
      const REAL beta = it == 0 ? (REAL)0 : (REAL)(rho / rho_old);
#pragma omp parallel for schedule(static)
      for (long i = 0; i < m; ++i) { p[i] = y[i] + beta * p[i]; pt[i] = r[i] + beta * pt[i]; }
      FN(wb_spmv)(A_st, off, R, m, p, pt, hp);
      double php = 0;
#pragma omp parallel for reduction(+ : php) schedule(static)
      for (long i = 0; i < m; ++i) php += (double)p[i] * hp[i];
      const REAL alpha = (REAL)(rho / php);
      rn = 0;
#pragma omp parallel for reduction(+ : rn) schedule(static)
      for (long i = 0; i < m; ++i) { u[i] += alpha * p[i]; z[i] += alpha * pt[i]; r[i] -= alpha * hp[i]; rn += (double)r[i] * r[i]; }
      rho_old = rho;
      ++it;
      if (sqrt(rn / rn0) < tol) break;
    }
  if (rel_res_out) *rel_res_out = rn0 > 0 ? sqrt(rn / rn0) : 0;
  free(off); free(r); free(y); free(p); free(pt); free(hp); free(tmp);
  return it;
}

/* out = (kron_q M_q) v (trans = 0) or (kron_q M_q^T) v (trans = 1); M_q dense g_q x g_q row-major, concatenated in `mats`;
 * tmp holds 2 m reals.  Same line-parallel loop as wb_kron with a dense factor in place of the Toeplitz one. */
static void FN(wb_modes)(const REAL *mats, int trans, int d, const int *g, long m, const REAL *v, REAL *tmp, REAL *out) {
  const REAL *cur = v;
  long post = m;
  const REAL *M = mats;
  for (int q = 0; q < d; ++q) {
    const int gq = g[q];
    post /= gq;
    const long pre = m / (post * gq);
    REAL *dst0 = (q == d - 1) ? out : (tmp + (q & 1) * m);
    const long po = post;
#pragma omp parallel for collapse(2) schedule(static)
    for (long pp = 0; pp < pre; ++pp)
      for (int i = 0; i < gq; ++i) {
        REAL *dst = dst0 + (pp * gq + i) * po;
        for (long s = 0; s < po; ++s) dst[s] = 0;
        for (int j = 0; j < gq; ++j) {
          const REAL t = trans ? M[(long)j * gq + i] : M[(long)i * gq + j];
          const REAL *src = cur + (pp * gq + j) * po;
#pragma omp simd
          for (long s = 0; s < po; ++s) dst[s] += t * src[s];
        }
      }
    cur = dst0;
    M += (long)gq * gq;
  }
}

/* The same warm-started CG as wb_pcg with the separable density-profile preconditioner of DESIGN.md 3.3,
 *   P = (Kt^-1 + a kron_q diag(t_q))^-1 = X f2(D) X^T,   Kt^-1 P = Z f1(D) X^T,
 * from the d generalized eigenproblems K_q = X_q D_q X_q^T, X_q^T diag(t_q) X_q = I, Z_q = diag(t_q) X_q (host, numpy):
 *   f1 = 1 / (1 + a lam), f2 = lam f1, lam = kscale prod_q D_q[i_q].
 * X, Zm: concatenated g_q x g_q row-major factors; evals: concatenated D_q; shift = a. */
int FN(wb_pcg_profile)(const REAL *A_st, int d, const int *g, long m, REAL kscale, const REAL *X, const REAL *Zm, const REAL *evals, REAL shift,
                       const REAL *rhs, int warm, double tol, int max_iter, REAL *u, REAL *z, double *rel_res_out) {
  long R = 1;
  for (int q = 0; q < d; ++q) R *= 7;
  long *off = (long *)malloc(sizeof(long) * R);
  FN(wb_offsets)(d, g, off);
  REAL *r = (REAL *)malloc(sizeof(REAL) * m), *y = (REAL *)malloc(sizeof(REAL) * m), *t = (REAL *)malloc(sizeof(REAL) * m);
  REAL *p = (REAL *)malloc(sizeof(REAL) * m), *pt = (REAL *)malloc(sizeof(REAL) * m), *hp = (REAL *)malloc(sizeof(REAL) * m);
  REAL *c1 = (REAL *)malloc(sizeof(REAL) * m), *c2 = (REAL *)malloc(sizeof(REAL) * m), *tmp = (REAL *)malloc(sizeof(REAL) * 2 * m);
  REAL *f1 = (REAL *)malloc(sizeof(REAL) * m), *f2 = (REAL *)malloc(sizeof(REAL) * m);
#pragma omp parallel for schedule(static)
  for (long i = 0; i < m; ++i) {
    long rem = i;
    double lam = (double)kscale;
    long eo = 0;
    for (int q = 0; q < d; ++q) eo += g[q];
    for (int q = d - 1; q >= 0; --q) {
      eo -= g[q];
      lam *= (double)evals[eo + rem % g[q]];
      rem /= g[q];
    }
    const double a1 = 1.0 / (1.0 + (double)shift * lam);
    f1[i] = (REAL)a1;
    f2[i] = (REAL)(lam * a1);
  }
  double rn0 = 0, rn = 0;
  if (!warm) {
    memset(u, 0, sizeof(REAL) * m);
    memset(z, 0, sizeof(REAL) * m);
  }
  if (warm) FN(wb_spmv)(A_st, off, R, m, u, z, hp);   /* hp = z + A u */
#pragma omp parallel for reduction(+ : rn0, rn) schedule(static)
  for (long i = 0; i < m; ++i) {
    r[i] = warm ? rhs[i] - hp[i] : rhs[i];
    p[i] = 0; pt[i] = 0;
    rn0 += (double)rhs[i] * rhs[i];
    rn += (double)r[i] * r[i];
  }
  double rho_old = 1;
  int it = 0;
  if (rn0 > 0 && sqrt(rn / rn0) >= tol)
    for (; it < max_iter;) {
      FN(wb_modes)(X, 1, d, g, m, r, tmp, c1);          /* eigen-coefficients of r */
#pragma omp parallel for schedule(static)
      for (long i = 0; i < m; ++i) { const REAL c = c1[i]; c1[i] = f1[i] * c; c2[i] = f2[i] * c; }
      FN(wb_modes)(X, 0, d, g, m, c2, tmp, y);          /* y = P r */
      FN(wb_modes)(Zm, 0, d, g, m, c1, tmp, t);         /* t = Kt^-1 y */
      double rho = 0;
#pragma omp parallel for reduction(+ : rho) schedule(static)
      for (long i = 0; i < m; ++i) rho += (double)r[i] * y[i];
      const REAL beta = it == 0 ? (REAL)0 : (REAL)(rho / rho_old);
#pragma omp parallel for schedule(static)
      for (long i = 0; i < m; ++i) { p[i] = y[i] + beta * p[i]; pt[i] = t[i] + beta * pt[i]; }
      FN(wb_spmv)(A_st, off, R, m, p, pt, hp);
      double php = 0;
#pragma omp parallel for reduction(+ : php) schedule(static)
      for (long i = 0; i < m; ++i) php += (double)p[i] * hp[i];
      const REAL alpha = (REAL)(rho / php);
      rn = 0;
#pragma omp parallel for reduction(+ : rn) schedule(static)
      for (long i = 0; i < m; ++i) { u[i] += alpha * p[i]; z[i] += alpha * pt[i]; r[i] -= alpha * hp[i]; rn += (double)r[i] * r[i]; }
      rho_old = rho;
      ++it;
      if (sqrt(rn / rn0) < tol) break;
    }
  if (rel_res_out) *rel_res_out = rn0 > 0 ? sqrt(rn / rn0) : 0;
  free(off); free(r); free(y); free(t); free(p); free(pt); free(hp); free(c1); free(c2); free(tmp); free(f1); free(f2);
  return it;
}
