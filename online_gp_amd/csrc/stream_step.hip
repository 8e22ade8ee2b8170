// One streaming step behind ONE C-ABI call: the launches of
//   predictive mean of the incoming batch + absorb   wiski_scatter_stats_step (BFN:206-210, BFN:155-171 + URLT:58; carries the residual)
//   warm-started refresh of the mean                 wiski_pcg                (CG branch of BFN:368-383)
// are queued back to back on the caller's stream without returning to the host language in between (the Python
// front-end left the GPU idle for ~30 us between the gather and the scatter and ~40 us after the solver's last poll).
// With a deferred solve pending, the absorb of the NEXT batch is queued before the host has read that solve's convergence
// poll, guarded on the device by the poll's own verdict (wiski_pcg_async_guard): the GPU goes from the last iteration
// straight into the absorb instead of idling through the host's poll round trip (~20 us of a 220 us step).
// Pure host code: it only sequences the entry points above.
#include "wiski_common.h"

template <typename real>
struct StreamArgs;
template <>
struct StreamArgs<float> { using type = wiski_stream_args_f32; };
template <>
struct StreamArgs<double> { using type = wiski_stream_args_f64; };

static int scatter1(const wiski_grid* g, const float* x, const float* y, const float* wa, const float* wb, const float* nz, int64_t n, float* b, float* A, float* cnt, const float* u, float* res, double* st, int32_t* err, void* s) {
  return wiski_scatter_stats_cnt_f32(g, x, y, wa, wb, nz, n, b, A, 1, cnt, u, res, st, err, s);
}
static int scatter1(const wiski_grid* g, const double* x, const double* y, const double* wa, const double* wb, const double* nz, int64_t n, double* b, double* A, double* cnt, const double* u, double* res, double* st, int32_t* err, void* s) {
  return wiski_scatter_stats_cnt_f64(g, x, y, wa, wb, nz, n, b, A, 1, cnt, u, res, st, err, s);
}
// absorb + predictive mean of the batch (+ the zeroing the following solve would do in a launch of its own) in ONE kernel
static int scatter_step1(const wiski_grid* g, const wiski_stream_args_f32* a, const float* x, const float* y, const float* wa, const float* wb, const float* nz, int64_t n, int carry, float* mean_out, int zero, const void* guard, int64_t expect, void* s) {
  void *p1 = nullptr, *p2 = nullptr;
  int64_t n1 = 0, n2 = 0;
  if (zero)
    if (int rc = wiski_pcg_zero_regions_f32(g, 1, a->max_iter, a->d_work, 1, &p1, &n1, &p2, &n2)) return rc;
  if (wiski_shard_active(a->shard)) {
    int32_t glo = 0, ghi = 0;
    if (int rc = wiski_shard_groups(g->d, a->shard->rank, a->shard->nranks, &glo, &ghi)) return rc;
    return wiski_scatter_stats_step_sharded_f32(g, x, y, wa, wb, nz, n, a->d_b, a->d_A_half, a->d_cnt, a->d_U, carry ? a->d_R : nullptr, mean_out, a->d_stats, a->d_err, p1, n1, p2, n2, guard, expect, glo, ghi, s);
  }
  return wiski_scatter_stats_step_f32(g, x, y, wa, wb, nz, n, a->d_b, a->d_A_half, a->d_cnt, a->d_U, carry ? a->d_R : nullptr, mean_out, a->d_stats, a->d_err, p1, n1, p2, n2, guard, expect, a->d_bin, a->bin_bytes, s);
}
static int scatter_step1(const wiski_grid* g, const wiski_stream_args_f64* a, const double* x, const double* y, const double* wa, const double* wb, const double* nz, int64_t n, int carry, double* mean_out, int zero, const void* guard, int64_t expect, void* s) {
  void *p1 = nullptr, *p2 = nullptr;
  int64_t n1 = 0, n2 = 0;
  if (zero)
    if (int rc = wiski_pcg_zero_regions_f64(g, 1, a->max_iter, a->d_work, 1, &p1, &n1, &p2, &n2)) return rc;
  if (wiski_shard_active(a->shard)) {
    int32_t glo = 0, ghi = 0;
    if (int rc = wiski_shard_groups(g->d, a->shard->rank, a->shard->nranks, &glo, &ghi)) return rc;
    return wiski_scatter_stats_step_sharded_f64(g, x, y, wa, wb, nz, n, a->d_b, a->d_A_half, a->d_cnt, a->d_U, carry ? a->d_R : nullptr, mean_out, a->d_stats, a->d_err, p1, n1, p2, n2, guard, expect, glo, ghi, s);
  }
  return wiski_scatter_stats_step_f64(g, x, y, wa, wb, nz, n, a->d_b, a->d_A_half, a->d_cnt, a->d_U, carry ? a->d_R : nullptr, mean_out, a->d_stats, a->d_err, p1, n1, p2, n2, guard, expect, a->d_bin, a->bin_bytes, s);
}
static int pcg1(const wiski_grid* g, const wiski_stream_args_f32* a, int warm, int first_check, int32_t* it, double* rr, int32_t* herr, void* s,
                wiski_pcg_async* as, int mode) {
  return wiski_pcg_twolevel_f32(g, a->d_A_half, a->d_tcol, a->kscale, a->d_evec, a->d_evec2, a->d_eval, a->shift, a->d_b, 1, a->d_U, a->d_Z, warm, a->tol,
                                a->max_iter, a->check_every, first_check, a->d_work, a->work_bytes, it, rr, a->d_err, herr, 1, a->d_R, s, as, mode, a->shard,
                                a->two_level);
}
static int pcg1(const wiski_grid* g, const wiski_stream_args_f64* a, int warm, int first_check, int32_t* it, double* rr, int32_t* herr, void* s,
                wiski_pcg_async* as, int mode) {
  if (a->two_level) return WISKI_E_BADARG;
  return wiski_pcg_sharded_f64(g, a->d_A_half, a->d_tcol, a->kscale, a->d_evec, a->d_evec2, a->d_eval, a->shift, a->d_b, 1, a->d_U, a->d_Z, warm, a->tol,
                               a->max_iter, a->check_every, first_check, a->d_work, a->work_bytes, it, rr, a->d_err, herr, 1, a->d_R, s, as, mode, a->shard);
}

template <typename real>
static int stream_step_impl(const wiski_grid* grid, const typename StreamArgs<real>::type* a, const real* d_x, const real* d_y, const real* d_wa,
                            const real* d_wb, const real* d_noise, int64_t q, real* d_mean_out, int32_t carry, int32_t first_check, int32_t* h_iters,
                            double* h_relres, int32_t* h_err, void* stream, wiski_pcg_async* as, int32_t defer, int32_t* h_resumed) {
  if (!grid || !a || q < 0 || !a->d_A_half || !a->d_b || !a->d_U || !a->d_Z || !a->d_R) return WISKI_E_BADARG;
  if (q > 0 && (!d_x || !d_y || !d_wa || !d_wb || !d_noise)) return WISKI_E_BADARG;
  if (wiski_shard_active(a->shard) && q > 0 && !d_mean_out) return WISKI_E_BADARG;   // the sharded absorb is the mean-emitting kernel
  int rc = WISKI_OK;
  if (h_resumed) *h_resumed = 0;
  bool absorbed = false;                     // the speculative absorb below has run
  if (as && as->state == 1) {
    // A solve the previous call started is pending.  It has to finish BEFORE anything reads U (the batch mean) or changes the
    // system (the absorb) -- but the host need not have SEEN it finish: queue this batch's absorb now, guarded on the device
    // by the verdict of the pending poll (it runs iff that poll finds the solve converged and the error flag clear, which
    // is exactly when RESUME below queues nothing more), and only then wait for the poll.  The absorb then executes while
    // the host reads the poll and queues the next solve.
    bool spec = false;
    if (q > 0 && d_mean_out) {
      const void* guard = nullptr;
      int64_t expect = 0;
      if (wiski_pcg_async_guard(as, &guard, &expect) == WISKI_OK) {
        rc = scatter_step1(grid, a, d_x, d_y, d_wa, d_wb, d_noise, q, carry, d_mean_out, 1, guard, expect, stream);
        if (rc) return rc;
        spec = true;
      }
    }
    {
      typename StreamArgs<real>::type ar = *a;     // RESUME with the shift the solve was started with (a->shift is the NEXT solve's)
      ar.shift = (real)as->shift;
      rc = pcg1(grid, &ar, 2, first_check, h_iters, h_relres, h_err, stream, as, 2);
    }
    if (h_resumed) *h_resumed = 1;
    if (rc != WISKI_OK && rc != WISKI_E_NOTCONV) return rc;
    absorbed = spec && as->guard_ok;          // else the guarded kernel was a no-op (and RESUME may have queued more iterations)
    if (q == 0) return rc;
    if (!absorbed && h_err && *h_err) return rc;   // out-of-grid points in the previous batch: let the caller deal with them first
  } else if (q == 0 && as) {
    return WISKI_OK;
  }
  const int resumed_rc = rc;
  if (absorbed) {
    as->prezeroed = 1;
  } else if (q > 0) {
    if (d_mean_out) {
      // the absorb kernel forms w_p . U for the residual carry anyway: it is the predictive mean of the batch, so there is no
      // gather launch; with a handle the same kernel also zeroes what the solve below would zero in a launch of its own
      rc = scatter_step1(grid, a, d_x, d_y, d_wa, d_wb, d_noise, q, carry, d_mean_out, as ? 1 : 0, nullptr, 0, stream);
      if (rc) return rc;
      if (as) as->prezeroed = 1;           // nothing touches the solve's workspace between this kernel and the solve below
    } else {
      rc = scatter1(grid, d_x, d_y, d_wa, d_wb, d_noise, q, a->d_b, a->d_A_half, a->d_cnt, carry ? a->d_U : nullptr, carry ? a->d_R : nullptr, a->d_stats,
                    a->d_err, stream);
      if (rc) return rc;
    }
  }
  if (!as) return pcg1(grid, a, carry ? 2 : 1, first_check, h_iters, h_relres, h_err, stream, nullptr, 0);
  if (defer) {
    int32_t it2 = 0, e2 = 0;
    double r2 = 0;
    as->shift = (double)a->shift;
    rc = pcg1(grid, a, carry ? 2 : 1, first_check, &it2, &r2, &e2, stream, as, 1);      // outputs of THIS solve arrive with the next call
    return rc == WISKI_PENDING ? (resumed_rc == WISKI_E_NOTCONV ? WISKI_E_NOTCONV : WISKI_PENDING) : rc;
  }
  // deferral switched off with a solve pending: the outputs below describe THIS step's synchronous solve; the resumed
  // one (finished above) is reported as *h_resumed = 2 -- "resumed, but its numbers were overwritten" -- and a resumed
  // solve that stopped at max_iter still surfaces through the return code
  if (h_resumed && *h_resumed == 1) *h_resumed = 2;
  rc = pcg1(grid, a, carry ? 2 : 1, first_check, h_iters, h_relres, h_err, stream, as, 0);
  return rc == WISKI_OK && resumed_rc == WISKI_E_NOTCONV ? WISKI_E_NOTCONV : rc;
}

extern "C" {
int wiski_stream_step_f32(const wiski_grid* grid, const wiski_stream_args_f32* a, const float* d_x, const float* d_y, const float* d_wa, const float* d_wb, const float* d_noise, int64_t q, float* d_mean_out, int32_t carry, int32_t first_check, int32_t* h_iters, double* h_relres, int32_t* h_err, void* stream, wiski_pcg_async* as, int32_t defer, int32_t* h_resumed) {
  return stream_step_impl<float>(grid, a, d_x, d_y, d_wa, d_wb, d_noise, q, d_mean_out, carry, first_check, h_iters, h_relres, h_err, stream, as, defer, h_resumed);
}
int wiski_stream_step_f64(const wiski_grid* grid, const wiski_stream_args_f64* a, const double* d_x, const double* d_y, const double* d_wa, const double* d_wb, const double* d_noise, int64_t q, double* d_mean_out, int32_t carry, int32_t first_check, int32_t* h_iters, double* h_relres, int32_t* h_err, void* stream, wiski_pcg_async* as, int32_t defer, int32_t* h_resumed) {
  return stream_step_impl<double>(grid, a, d_x, d_y, d_wa, d_wb, d_noise, q, d_mean_out, carry, first_check, h_iters, h_relres, h_err, stream, as, defer, h_resumed);
}
}
