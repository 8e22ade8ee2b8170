/*
 * TEST INFRASTRUCTURE -- multi-threaded CPU baseline of the WISKI streaming step (bench.py `cpu_baseline`, kind "port").
 *
 * The scalar checker (wiski_oracle.c) stays single-threaded and is what the parity tests use; this file is the
 * same matrix-free algorithm with OpenMP over all host cores and a warm-started solve, so that the baseline runs the
 * workload of the GPU leg (same grid, q, init, tolerance, warm starts) instead of a cold scalar solve:
 *   absorb   (BFN:31-60,155-171 + URLT:58)   points in parallel, `omp atomic` adds into b and the block stencil
 *   A . v    (URLT:47-48)                    rows in parallel, stencil offsets streamed in order (full 7^d stencil)
 *   Kt . v   (BFN:334-348)                   Kronecker-Toeplitz mode products, lines in parallel
 *   refresh  (CG branch of BFN:368-383)      preconditioned CG on (Kt^-1 + A) u = b, warm-started from the previous
 *                                            (u, z = Kt^-1 u): wb_pcg with Kt as the preconditioner, wb_pcg_profile with the
 *                                            separable density-profile preconditioner of the GPU library (DESIGN.md 3.3;
 *                                            its small generalized eigenproblems are solved by numpy in baseline.py) --
 *                                            the one bench.py times, so that both legs run the same algorithm
 *   predict  (BFN:206-210)                   fused gather, queries in parallel
 * Checked against the scalar oracle in tests/test_oracle.py.  Never imported by online_gp_amd.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int wb_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* threads of the parallel regions that follow (the caller picks the CPUs it is actually allowed to use: a container's CPU
 * quota can be far below the logical CPU count, and over-subscribing it gets the whole process throttled) */
void wb_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUFFIX)

#define REAL double
#define SUFFIX _f64
#include "wiski_baseline_omp_impl.h"
#undef REAL
#undef SUFFIX

#define REAL float
#define SUFFIX _f32
#include "wiski_baseline_omp_impl.h"
#undef REAL
#undef SUFFIX
