"""Feature flags / numeric settings as context managers.

``check_decomposition`` and ``detach_interp_coeff`` mirror the reference's
online_gp/settings.py:3-7; the rest restate the gpytorch.settings the reference
reads or its drivers set (config/regression.yaml:24-27,
experiments/bayesopt/bayesopt.py:278-291; batched_fixed_noise_online_gp.py:15).
"""


class _feature_flag:
    _state = False

    @classmethod
    def on(cls):
        return cls._state

    @classmethod
    def off(cls):
        return not cls._state

    @classmethod
    def _set_state(cls, state):
        cls._state = state

    def __init__(self, state=True):
        self.prev = self.__class__.on()
        self.state = state

    def __enter__(self):
        self.__class__._set_state(self.state)

    def __exit__(self, *args):
        self.__class__._set_state(self.prev)
        return False


class _value_context:
    _global_value = None

    @classmethod
    def value(cls):
        return cls._global_value

    @classmethod
    def _set_value(cls, value):
        cls._global_value = value

    def __init__(self, value):
        self._orig_value = self.__class__.value()
        self._instance_value = value

    def __enter__(self):
        self.__class__._set_value(self._instance_value)

    def __exit__(self, *args):
        self.__class__._set_value(self._orig_value)
        return False


class check_decomposition(_feature_flag):
    _state = False


class detach_interp_coeff(_feature_flag):
    _state = False


class skip_posterior_variances(_feature_flag):
    _state = False


class fast_pred_var(_feature_flag):
    """gpytorch.settings.fast_pred_var: upstream then builds the predictive covariance from a Lanczos root of rank
    <= max_root_decomposition_size (BFN:229-243, 393-397).  Here: predictive variances from the spectral Woodbury factor
    (lazy/spectral_woodbury.py) with its basis CAPPED at min(spectral_max_rank, max_root_decomposition_size) vectors even where
    that leaves out more than ``spectral_tail`` of the prior trace (rough kernels) -- a rank-limited approximation like
    upstream's, except that the left-out prior variance of every query is added back and reported (``rel_bound``)."""

    _state = False


class fast_pred_samples(_feature_flag):
    """gpytorch.settings.fast_pred_samples: upstream then returns the predictive covariance as a RootLazyTensor built from a
    Lanczos root of the inducing posterior (BFN:229-243).  Here: where a factor provides a root without a solve -- the spectral
    Woodbury factor on large grids, the dense factor on small ones -- ``model(X)`` returns ``distributions.RootLazyTensor`` (root
    [n*, r] + the left-out prior variance as a diagonal term: the SAME covariance the default path returns, no Lanczos
    truncation); on the PCG path (rough kernels on large grids) the exact covariance is returned as before.  Independently of this
    flag ``MultivariateNormal.rsample`` samples through such a root whenever the covariance has one."""

    _state = False


class skip_logdet_forward(_feature_flag):
    _state = False


class deferred_refresh(_feature_flag):
    """``stream_step`` returns as soon as its launches are queued -- the first convergence poll of the warm-started solve is
    left in flight and looked at by the NEXT call that needs the posterior (a following ``stream_step``, ``prediction_cache``,
    ``posterior`` ...), which also finishes the solve should that poll say "not converged".  The host-side work between two
    steps then overlaps the GPU's CG iterations.  Results are identical; the iteration count and the out-of-grid error of a
    step surface one call later."""

    _state = False


class float32_grid(_feature_flag):
    """gpytorch builds the inducing grid in float32 and promotes it; the spec'd geometry here is float64.
    On = reproduce the quirk: first grid point and spacing come from a float32 ``linspace(lo - delta, hi + delta, g)``.
    The difference is ~1e-7 relative in g0 / h; tests/test_model_gpu.py shows the posterior moves by far less than the
    parity tolerance either way.  Read when a GridSpec / kernel is constructed."""

    _state = False


class deferred_bounds_check(_feature_flag):
    """Query points outside the inducing grid: off (default) = the posterior call itself raises, as gpytorch's grid
    check does (one flag read per call: a publish kernel + a host spin behind the gather).  On = the device flag is
    only looked at with the next solver poll / ``check_bounds()`` -- for streaming loops that evaluate and update in
    lock-step and would rather not wait for the gather before queueing the update."""

    _state = False


class use_toeplitz(_feature_flag):
    _state = True


class cg_tolerance(_value_context):
    """Relative residual ||r||/||rhs|| at which wiski_pcg stops.  None = dtype
    default (1e-6 for fp32, 1e-10 for fp64): tighter than gpytorch's eval
    default (1e-2) because the north-star parity bar is rtol 1e-4."""

    _global_value = None


class variance_cg_tolerance(_value_context):
    """Relative-residual tolerance of the PCG solves behind predictive *variances* (diagonal requests only).  None = the
    same as ``cg_tolerance``.  A variance is the quadratic form w^T M w = w^T u of the solve's own right-hand side, and CG's
    error in that form is the squared energy norm of the error -- second order in the residual -- so a residual tolerance of
    1e-2 already gives variances good to ~1e-4 (bench.py measures the deviation from a tight solve next to the latency).  Full
    covariance requests keep ``cg_tolerance`` (bilinear forms of different vectors are first order)."""

    _global_value = None


class max_cg_iterations(_value_context):
    _global_value = 2000


class max_cholesky_size(_value_context):
    """Grids with m <= this use the dense MFMA Cholesky path, larger ones PCG."""

    _global_value = 2048


class max_root_decomposition_size(_value_context):
    _global_value = 512


class cholesky_jitter(_value_context):
    _global_value = None


class variance_chunk(_value_context):
    """Right-hand sides per batched PCG solve when predictive variances are requested."""

    _global_value = 64


class spectral_preconditioner(_feature_flag):
    """Precondition wiski_pcg with (Kt^-1 + a I)^-1 in the Kronecker eigenbasis of
    Kuu (a = mean row sum of W^T D^-1 W) instead of Kt alone."""

    _state = True


class cg_check_every(_value_context):
    """Iterations between host-side convergence checks of wiski_pcg (warm-started refreshes poll
    first at the iteration count of the previous refresh).  Wide solves (>= 16 columns and >= 2^21
    entries: the variance / probe solves on large grids) poll after EVERY iteration whatever this
    says -- an iteration there costs ~50 polls (lazy/operators.py: solve_columns)."""

    _global_value = 3


class dense_small_grids(_feature_flag):
    """Grids with m <= max_cholesky_size use the dense MFMA Cholesky factor (the
    reference's own regime) instead of PCG; the posterior matrix M is then cached."""

    _state = True


class dense_rank_updates(_feature_flag):
    """Dense regime: condition_on_observations / fantasies update the cached posterior matrix by a rank-q Woodbury
    step (O(m^2 q)) instead of re-factorising (O(m^3)); a fresh factor is taken after 64 stacked updates."""

    _state = True


class residual_carry_over(_feature_flag):
    """Keep the posterior-mean residual b - z - A u current inside the scatter launches of streaming updates, so
    that the warm-started refresh needs no A u product (recomputed from scratch every 16th refresh)."""

    _state = True


class precond_profile_drift(_value_context):
    """Largest change of the normalised per-dimension data-density profile (entries in [0.01, 1]) the
    CG preconditioner tolerates before its generalized eigenbasis is recomputed (the estimate itself
    carries ~10 % sampling noise at 2e4 points on a 50-node axis)."""

    _global_value = 0.2


class density_profile_preconditioner(_feature_flag):
    """Model W^T D^-1 W in the CG preconditioner as a kron_q diag(t_q) with t_q the per-dim
    data-density profile (instead of a I): the inducing nodes outside the data box carry no
    data, which the constant model gets wrong (4-5x more CG iterations)."""

    _state = True


class two_level_preconditioner(_feature_flag):
    """One-call streaming steps on large 3-D fp32 grids (``FixedNoiseOnlineSKIGP.stream_step``): keep the EXACT block
    (D_S^-1 + X_S^T W^T D^-1 W X_S)^-1 of the system matrix on the ``two_level_rank`` dominant modes of the preconditioner's
    generalized eigenbasis and the separable density model on the rest (``lazy/two_level.py``, include/wiski.h: wiski_twolevel).
    The separable model alone needs 6 CG iterations per step on road-like (line-clustered) streams, 2.5 on uniform ones.
    The block follows the stream on a side stream; its age costs iterations, never accuracy."""

    _state = True


class two_level_rank(_value_context):
    """Modes in the exact block of the two-level preconditioner (<= 480).  50^3, road-like stream: 128 / 192 / 256 / 384 modes
    need 3.28 / 3.02 / 3.02 / 3.02 iterations per step (2.0 late in the stream); the refresh costs grow with the square."""

    _global_value = 192


class fused_factor_refresh(_feature_flag):
    """The spectral factor's refresh after a hyper-parameter step (eigenvector update -> change of basis -> G -> C -> Cholesky + inverse -> tail
    products) queued by ONE C call (``wiski_factor_refresh``) instead of seven wrapped ones: the same launches, ~40 us less interpreter time
    in a step that is bound by it.  Off: the call-by-call form (identical results)."""

    _state = True


class two_level_rebuild(_feature_flag):
    """Where the stream's two-level block was lost -- a hyper-parameter step or a density-profile re-solve moved the eigenbasis, points
    reached the statistics behind the tracker's back -- rebuild it from the statistics themselves (X_S^T A X_S: r stencil-product columns,
    ~1.5 ms at 50^3) the next time a WIDE solve (>= 16 columns: variances, probes, fantasies) asks for it.  The reference's per-batch
    loop on the PCG path (evaluate -> Adam step -> condition) then pays one rebuild per step and saves two thirds of the iterations of
    every 64-column solve in it."""

    _state = True


class two_level_min_iters(_value_context):
    """The two-level block is built only for streams whose first warm steps need at least this many CG iterations under the
    separable density model alone (uniform-like streams converge in 2-3 and never pay for it)."""

    _global_value = 4.0



class two_level_subsample(_value_context):
    """The two-level block's Gram matrix is accumulated from every `n`-th absorbed point (weighted by n): an unbiased estimate
    of X_S^T W^T D^-1 W X_S at 1/n of the refresh work; 1 = every point."""

    _global_value = 1


class two_level_lockstep(_feature_flag):
    """Switch a refreshed two-level block in exactly ``two_level_lag`` steps after its refresh was started even on one GPU (what
    stencil-sharded replicas always do): iteration counts then do not depend on how fast the side stream ran (tests)."""

    _state = False


class two_level_growth(_value_context):
    """A refresh of the two-level block is started when the absorbed weight has grown by this factor since the last one
    (50^3 road-like stream: 1.1 -> 22 refreshes per 3droad-sized pass, 2.57 iterations per step; 1.2 -> 13 and 2.62)."""

    _global_value = 1.2


class two_level_lag(_value_context):
    """Stencil-sharded multi-GPU steps (replicas must take identical iterations): streaming steps between the start of a block
    refresh (side stream) and the step that switches it in -- fixed, so that every rank switches at the same step.  One GPU: a
    finished refresh is switched in by the first step that finds it complete (at the latest 4 x this many steps on)."""

    _global_value = 2


class spectral_factor(_feature_flag):
    """Large grids, smooth kernels: serve predictive variances and the marginal log-likelihood from a dense Woodbury
    factor in the dominant Kronecker eigenspace of Kuu (``lazy/spectral_woodbury.py``) whenever that space is small
    (rank <= ``spectral_max_rank`` for the trace tail ``spectral_tail``) -- the reference's own rank-limited root
    space (BFN:343-404 under ``max_root_decomposition_size``) with a known truncation bound.  Off: everything
    goes through wiski_pcg.  The posterior mean always does."""

    _state = True


class spectral_dense_regime(_feature_flag):
    """Small inducing grids (m <= max_cholesky_size: the reference's own configurations) take the reference's per-batch loop
    -- evaluate -> Adam step on the MLL -> condition (experiments/regression.py:48-54) -- through the spectral factor's device
    pipeline as well (at FULL rank for rough kernels: no truncation at all), where the owner of the model asks for it
    (the streaming wrappers do: ``model._stream_owner``).  Off: every dense-regime request builds the nodal dense factor
    (lazy/dense_woodbury.py), one framework op at a time."""

    _state = True


class spectral_tail(_value_context):
    """Fraction of trace(Kuu) the reduced eigenbasis may leave out (None: 1e-6 in fp32 -- 4e-4 of a variance at 50^3, against the fp32 parity bar of 1e-2 -- and 1e-9 in fp64).  The left-out
    prior variance of each query is added back to its predictive variance and bounds the remaining error."""

    _global_value = None


class spectral_mean_tolerance(_value_context):
    """Largest admissible bound on the truncation error of a predictive MEAN taken from the spectral factor, relative to the
    largest mean of the batch (None: the parity bars themselves, 1e-2 in fp32 and 1e-4 in fp64 -- the bound is rigorous
    (Cauchy-Schwarz) and loose: measured at 50^3 fp32 it reads 3e-4 .. 1e-3 where the actual deviation from a tight PCG solve is
    1-4e-5).  The factor evaluates sqrt(tail(w) * b^T (Kt - Kt_B) b) every few states
    (``SpectralWoodburyFactor.mean_monitor``); above the limit the model answers means from its PCG state instead."""

    _global_value = None


class graphed_hyper_step(_feature_flag):
    """Streaming wrappers (OnlineSKIRegression ...): run the per-batch Adam step on the MLL as one captured HIP graph when the spectral
    factor serves the MLL (models/_graphed_step.py); off: the op-by-op step."""

    _state = True


class fused_hyper_columns(_feature_flag):
    """Toeplitz columns of (Scale of) RBF / Matern kernels and their gradient w.r.t. lengthscales and outputscale by one HIP launch
    each (``wiski_stationary_columns``) instead of the broadcasting-op graph; off: the op graph (what foreign kernels always use)."""

    _state = True


class fused_hyper_step(_feature_flag):
    """The captured Adam step (``graphed_hyper_step``) recorded WITHOUT autograd where the model has the standard parameterisation --
    (Scale of)* RBF | Matern, homoskedastic second noise, no registered priors, plain Adam: constraint transforms, MLL tail, chain rule
    and the Adam update are three kernels (csrc/hyper_step.hip) around the factor's backward launches, 11 graph nodes instead of 40; the
    same graph leaves the NEXT step's Toeplitz columns and sigma2 behind, so the following refresh starts without a host read.  Off: the
    autograd recording (what every other parameterisation takes)."""

    _state = True



class fused_evaluate(_feature_flag):
    """``OnlineSKIRegression.evaluate`` of a batch of <= 64 points from the spectral factor as one projection launch + ONE launch for
    means, variances and both metrics (``wiski_spectral_evaluate``) instead of the posterior object and ~18 small launches; off: the
    general path (what larger batches and several outputs always take)."""

    _state = True


class spectral_device_refresh(_feature_flag):
    """After a hyper-parameter step, refine the spectral factor's per-dim eigenvectors on the device (subspace iteration from the
    previous ones, ``wiski_basis_eig_update``) and keep the index set, instead of a host eigh + re-selection; the refinement's
    residual, the trace the kept index set leaves out and the reference-span defect are read back before the state is handed out
    (the copy is queued ahead of the factorisation, so the wait is short); a failed check (or every 64th step) takes the host path."""

    _state = True


class spectral_max_rank(_value_context):
    """Largest reduced basis the spectral factor is built for; beyond it the PCG path serves the request."""

    _global_value = 1024
