"""CPU: host logic, the C-ABI library loads and exports every symbol include/wiski.h
declares, and the product path fails loudly without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    from online_gp_amd import _hip

    if not os.path.exists(_hip._SO):
        _hip.build()
    return ctypes.CDLL(_hip._SO)


def test_cabi_exports_every_declared_symbol(built_lib):
    hdr = open(os.path.join(ROOT, "include", "wiski.h")).read()
    names = set(re.findall(r"^\s*(?:int|int64_t)\s+(wiski_\w+)\s*\(", hdr, flags=re.M))
    assert len(names) >= 20, names
    for n in sorted(names):
        assert hasattr(built_lib, n), f"libwiski_hip.so does not export {n}"
    assert built_lib.wiski_version() >= 1


def test_ctypes_structs_have_the_layout_of_the_header(tmp_path):
    """The by-pointer structs of the C ABI (wiski_grid, wiski_hyper_plan, wiski_copy_plan) as gcc lays them out from include/wiski.h
    against their ctypes mirrors in online_gp_amd/_hip.py: sizes and the offsets of the last members."""
    import ctypes
    import subprocess

    from online_gp_amd import _hip, grid_ops

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "wiski.h"\n'
                   'int main(void) { printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(wiski_grid), sizeof(wiski_hyper_param), sizeof(wiski_hyper_plan), '
                   'offsetof(wiski_hyper_param, upper), sizeof(wiski_copy_plan), offsetof(wiski_copy_plan, scalar), offsetof(wiski_copy_plan, scalar_dst), '
                   'sizeof(wiski_twolevel), offsetof(wiski_twolevel, d_mc), offsetof(wiski_twolevel, mc_cols)); return 0; }\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    want = [ctypes.sizeof(_hip.wiski_grid), ctypes.sizeof(_hip.wiski_hyper_param), ctypes.sizeof(_hip.wiski_hyper_plan), _hip.wiski_hyper_param.upper.offset,
            ctypes.sizeof(_hip.wiski_copy_plan), _hip.wiski_copy_plan.scalar.offset, _hip.wiski_copy_plan.scalar_dst.offset,
            ctypes.sizeof(grid_ops.TwoLevelStruct), grid_ops.TwoLevelStruct.d_mc.offset, grid_ops.TwoLevelStruct.mc_cols.offset]
    assert got == want, (got, want)


def test_workspace_query_runs_without_gpu(built_lib):
    from online_gp_amd import grid_ops

    g = grid_ops.GridSpec([[-1.1, 1.1]] * 3, 50)
    built_lib.wiski_pcg_workspace_bytes.restype = ctypes.c_int64
    need = built_lib.wiski_pcg_workspace_bytes(g.ref, 1, 100, 4)
    assert need >= 19 * 125000 * 4
    assert built_lib.wiski_pcg_workspace_bytes(g.ref, 0, 100, 4) < 0   # bad k -> WISKI_E_BADARG


def test_grid_spec_matches_oracle_spec():
    from online_gp_amd import grid_ops
    from oracle import spec

    for gb, gs in [([[-1.1, 1.1]] * 3, 50), ([[0.0, 1.0], [-2.0, 3.0]], [5, 9]), ([[-4.0, 14.0]], 20)]:
        g = grid_ops.GridSpec(gb, gs)
        g0, h, gg = spec.make_grid(gb, gs)
        assert np.allclose(g.g0, g0, rtol=0, atol=1e-15) and np.allclose(g.h, h, rtol=0, atol=1e-15) and list(gg) == g.g
        assert g.m == int(np.prod(gg)) and g.T == 4 ** g.d and g.R == 7 ** g.d
        pts = g.grid_points()
        delta = (gb[0][1] - gb[0][0]) / (g.g[0] - 2)               # linspace(lo - delta, hi + delta, g)
        assert abs(float(pts[0][0]) - (gb[0][0] - delta)) < 1e-12 and abs(float(pts[0][-1]) - (gb[0][1] + delta)) < 1e-12
    with pytest.raises(ValueError):
        grid_ops.GridSpec([[0, 1]] * 5, 8)
    with pytest.raises(ValueError):
        grid_ops.GridSpec([[0, 1]], 3)


def test_kernel_modules_match_oracle_columns():
    from online_gp_amd.kernels import GridInterpolationKernel, MaternKernel, RBFKernel, ScaleKernel
    from oracle import spec

    for kind, mk in [("rbf", lambda: RBFKernel(ard_num_dims=2)), ("matern52", lambda: MaternKernel(nu=2.5, ard_num_dims=2)),
                     ("matern12", lambda: MaternKernel(nu=0.5, ard_num_dims=2))]:
        k = GridInterpolationKernel(ScaleKernel(mk()), grid_size=[7, 9], num_dims=2, grid_bounds=[[-1, 1], [0, 2]])
        k.base_kernel.outputscale = 1.7
        k.base_kernel.base_kernel.lengthscale = torch.tensor([0.4, 1.3])
        g0, h, g = spec.make_grid([[-1, 1], [0, 2]], [7, 9])
        ref = np.concatenate(spec.toeplitz_columns(kind, h, g, [0.4, 1.3], 1.7))
        assert np.allclose(k.toeplitz_columns().detach().numpy(), ref, rtol=1e-6)
    # default raw parameters -> softplus(0)
    k = ScaleKernel(RBFKernel(ard_num_dims=3))
    assert abs(float(k.outputscale) - spec.SOFTPLUS0) < 1e-6
    assert torch.allclose(k.base_kernel.lengthscale, torch.full((1, 3), spec.SOFTPLUS0))
    # gradients flow to the raw hyper-parameters
    gk = GridInterpolationKernel(k, grid_size=6, num_dims=3, grid_bounds=[[-1, 1]] * 3)
    gk.toeplitz_columns().sum().backward()
    assert k.raw_outputscale.grad is not None and k.base_kernel.raw_lengthscale.grad.abs().sum() > 0


def test_likelihood_noise_product():
    from online_gp_amd.likelihoods import FNMGLikelihood

    lk = FNMGLikelihood(noise=torch.full((1, 5), 2.0), learn_additional_noise=True)
    assert torch.allclose(lk.noise, 2.0 * lk.second_noise)          # fnmg_likelihood.py:16-18
    lk.second_noise = 0.3
    assert abs(float(lk.second_noise) - 0.3) < 1e-6
    lk2 = FNMGLikelihood(noise=torch.ones(1, 5), learn_additional_noise=False)
    assert lk2.second_noise == 0 and lk2.second_noise_covar is None
    with pytest.raises(RuntimeError):
        lk2.second_noise = 1.0


def test_settings_context_managers():
    from online_gp_amd import settings

    assert settings.skip_posterior_variances.off()
    with settings.skip_posterior_variances(True):
        assert settings.skip_posterior_variances.on()
    assert settings.skip_posterior_variances.off()
    assert settings.cg_tolerance.value() is None
    with settings.cg_tolerance(1e-3):
        assert settings.cg_tolerance.value() == 1e-3
    assert settings.detach_interp_coeff.off() and settings.check_decomposition.off()


def test_product_path_refuses_cpu_tensors():
    """There is no CPU fallback: constructing the model on CPU tensors must raise."""
    from online_gp_amd import _hip
    from online_gp_amd.models import FixedNoiseOnlineSKIGP

    with pytest.raises(_hip.WiskiError):
        FixedNoiseOnlineSKIGP(torch.rand(5, 2), torch.rand(5, 1), None, grid_size=8)


def test_product_code_never_imports_the_oracle():
    import glob

    for f in glob.glob(os.path.join(ROOT, "online_gp_amd", "**", "*.py"), recursive=True):
        src = open(f).read()
        assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f


def test_half_stencil_layout_converters_roundtrip_on_cpu():
    """The row-interleaved half-stencil layout (include/wiski.h) against its definition, on the host:
    group 0 is A_h[4 i + (s - 3)], group g >= 1 is A_h[(7 g - 3) m + 7 i + s] for the offset-major row oh = 7 g - 3 + s."""
    import torch

    from online_gp_amd import grid_ops

    for d, g in ((1, 9), (2, 6), (3, 5)):
        grid = grid_ops.GridSpec([[-1.0, 1.0]] * d, g)
        H, m = (grid.R + 1) // 2, grid.m
        om = torch.arange(H * m, dtype=torch.float64).reshape(H, m)            # om[oh, i] = oh * m + i
        nat = grid_ops.half_stencil_from_offset_major(grid, om)
        assert nat.shape == (H, m) and torch.equal(grid_ops.half_stencil_to_offset_major(grid, nat), om)
        flat = nat.reshape(-1)
        for i in (0, 1, m // 2, m - 1):
            for oh in range(H):
                pos = 4 * i + oh if oh < 4 else (7 * ((oh - 4) // 7 + 1) - 3) * m + 7 * i + (oh - 4) % 7
                assert flat[pos] == oh * m + i
        assert grid_ops.is_half_stencil(grid, nat) and not grid_ops.is_half_stencil(grid, torch.zeros(grid.R, m))


def test_constraints_and_priors_follow_gpytorch_parameterisation():
    """Advisor finding r1: kernel kwargs must not be swallowed.  Interval / GreaterThan / Positive transforms, prior
    log-densities, registration through the kernel constructors as the reference's BO driver passes them
    (experiments/bayesopt/bayesopt.py:69-77), and a TypeError for anything unknown."""
    import math

    import pytest
    import torch

    from online_gp_amd.constraints import GreaterThan, Interval, Positive
    from online_gp_amd.kernels import GridInterpolationKernel, MaternKernel, RBFKernel, ScaleKernel
    from online_gp_amd.priors import GammaPrior, named_priors

    raw = torch.zeros(3, dtype=torch.float64)
    assert torch.allclose(Positive().transform(raw), torch.full((3,), math.log(2.0), dtype=torch.float64))
    assert torch.allclose(GreaterThan(1e-4).transform(raw), torch.full((3,), math.log(2.0) + 1e-4, dtype=torch.float64))
    iv = Interval(1e-4, 12.0)
    assert torch.allclose(iv.transform(raw), torch.full((3,), 0.5 * (12.0 + 1e-4), dtype=torch.float64))
    for c in (Positive(), GreaterThan(0.3), iv):
        v = torch.tensor([0.5, 1.7, 11.0], dtype=torch.float64)
        assert torch.allclose(c.transform(c.inverse_transform(v)), v, rtol=1e-12)
    with pytest.raises(RuntimeError):
        iv.inverse_transform(torch.tensor(13.0))
    # Gamma(3, 6) log-density at 0.5: 3 log 6 - lgamma(3) + 2 log 0.5 - 3
    lp = float(GammaPrior(3.0, 6.0).log_prob(torch.tensor(0.5, dtype=torch.float64)))
    assert abs(lp - (3 * math.log(6.0) - math.lgamma(3.0) + 2 * math.log(0.5) - 3.0)) < 1e-12
    k = ScaleKernel(MaternKernel(nu=2.5, ard_num_dims=3, lengthscale_prior=GammaPrior(3.0, 6.0), lengthscale_constraint=Interval(1e-4, 12.0)),
                    outputscale_prior=GammaPrior(2.0, 0.15), outputscale_constraint=Interval(1e-4, 12.0))
    cov = GridInterpolationKernel(k, grid_size=10, num_dims=3, grid_bounds=[[0.0, 1.0]] * 3)
    assert abs(float(k.outputscale.detach()) - 6.00005) < 1e-5 and abs(float(k.base_kernel.lengthscale.detach()[0, 0]) - 6.00005) < 1e-5
    k.base_kernel.lengthscale = 0.7
    assert torch.allclose(k.base_kernel.lengthscale.double(), torch.full((1, 3), 0.7, dtype=torch.float64), rtol=1e-6)
    names = sorted(n for n, _, _ in named_priors(cov))
    assert names == ["base_kernel.base_kernel.lengthscale_prior", "base_kernel.outputscale_prior"]
    total = sum(float(p.log_prob(c()).sum()) for _, p, c in named_priors(cov))
    assert math.isfinite(total)
    with pytest.raises(TypeError):
        RBFKernel(ard_num_dims=2, lengthscale_prio=GammaPrior(3.0, 6.0))
    with pytest.raises(TypeError):
        ScaleKernel(RBFKernel(), outputscale_priors=None, foo=1)


def test_stencil_shard_group_arithmetic_is_a_partition():
    """wiski_shard_groups (pure host arithmetic, no GPU): the ranks' group ranges tile [0, G) without gaps or overlap for
    every world size, and half_stencil_group_slices maps them onto disjoint element ranges that cover the whole half stencil."""
    import torch

    from online_gp_amd import grid_ops

    for d, G in ((1, 1), (2, 4), (3, 25), (4, 172)):
        grid = grid_ops.GridSpec(torch.tensor([[-1.0, 1.0]] * d), 6)
        for world in (1, 2, 3, 4, 8):
            ranges = [grid_ops.shard_groups(d, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == G
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
            covered = sorted(sl for lo, hi in ranges for sl in grid_ops.half_stencil_group_slices(grid, lo, hi))
            assert covered[0][0] == 0 and covered[-1][1] == (grid.R + 1) // 2 * grid.m
            assert all(covered[i][1] == covered[i + 1][0] for i in range(len(covered) - 1))


def test_dpp_broadcast_spmm_has_no_valu_to_dpp_hazard():
    """The 64-column half-stencil product issues its FMAs as inline-asm v_fmac_f32_dpp / v_fmac_f64_dpp (csrc/spmm_sym_bcast.h), which
    hipcc's hazard recogniser does not see: the gfx950 ISA of every instantiation must not read a coefficient register through DPP
    within two wait states of a VALU write to it, and must not spill (tools/check_dpp_hazards.py)."""
    import shutil
    import subprocess
    import sys

    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_dpp_hazards.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("0 hazards") == 4, r.stdout          # fp32 / fp64, with and without the p . Ap epilogue


def test_two_level_mode_selection_is_the_top_of_the_tensor_product_spectrum():
    """lazy/two_level.select_modes (host arithmetic): the r tensor-product modes of largest eigenvalue among the KMAX largest per dim,
    returned in block order (sorted by i0, then i1, then i2) with their eigenvalues -- against a brute-force sort."""
    from online_gp_amd.lazy import two_level as tlm

    rng = np.random.default_rng(4)
    D = [np.sort(rng.uniform(1e-6, 1.0, g) ** 3) for g in (12, 9, 40)]          # table order: ascending
    for rank in (1, 17, 200):
        idx, lam = tlm.select_modes(D, 2.5, rank)
        r = idx.shape[1]
        assert r == min(rank, 12 * 9 * min(40, tlm.KMAX), tlm.MAXR)
        full = 2.5 * np.einsum("i,j,k->ijk", D[0], D[1], D[2])
        assert np.allclose(lam, full[idx[0], idx[1], idx[2]], rtol=1e-14)
        top = np.sort(full.reshape(-1))[::-1][:r]
        assert np.allclose(np.sort(lam)[::-1], top, rtol=1e-14)
        keys = idx[0] * 10 ** 6 + idx[1] * 10 ** 3 + idx[2]
        assert np.all(np.diff(keys) > 0)                                          # block order, no mode twice


def test_backend_selection_of_multi_rank_runs():
    """One device per rank over nccl (= RCCL) whenever the box has a device for every rank; the 1-GPU boxes' self-test
    arrangement (gloo, every rank on device 0) otherwise; a forced nccl without the devices is refused.  bench.py and
    tests/test_distributed_gpu.py both take their backend from this function."""
    from online_gp_amd.distributed import pick_backend

    b, dev_of = pick_backend(2, 8)
    assert b == "nccl" and [dev_of(r) for r in range(2)] == [0, 1]
    b, dev_of = pick_backend(8, 8)
    assert b == "nccl" and [dev_of(r) for r in range(8)] == list(range(8))
    b, dev_of = pick_backend(2, 1)
    assert b == "gloo" and [dev_of(r) for r in range(2)] == [0, 0]
    b, dev_of = pick_backend(2, 8, "gloo")
    assert b == "gloo" and dev_of(1) == 0
    with pytest.raises(RuntimeError):
        pick_backend(2, 1, "nccl")
    with pytest.raises(ValueError):
        pick_backend(2, 2, "mpi")


def test_bench_spawns_its_own_ranks_without_a_launcher(monkeypatch):
    """`python bench.py --gpus N` with WORLD_SIZE unset re-executes itself under torch.distributed.run with N ranks on
    127.0.0.1 (the driver's own launch line) instead of benchmarking one GPU and printing n_gpus = 1."""
    import subprocess
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "10", "--warmup", "3"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as exc:
        bench.main()
    assert exc.value.code == 7                                   # the launcher's exit code is relayed
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert os.path.basename(cmd[cmd.index("--master-port") + 2]) == "bench.py"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "10", "--warmup", "3"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # a launcher whose world size disagrees with --gpus is refused (the line's n_gpus must be the world size)
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit):
        bench.main()


def test_stems_shapes_and_parameter_free_identity():
    """Identity owns nothing to learn (the streaming wrapper gives it a no-op optimiser); LinearStem / MLP map into (-1, 1)
    (reference online_gp/models/stems.py: BatchNorm without affine, tanh(z / 2)); hidden widths may come as the reference's
    comma-separated configuration string."""
    from online_gp_amd.models import MLP, Identity, LinearStem

    ident = Identity(3)
    x = torch.randn(17, 3)
    assert ident(x) is x and list(ident.parameters()) == [] and (ident.input_dim, ident.output_dim) == (3, 3)
    for stem in (LinearStem(3, 2), MLP(3, 2, 2, "16,8"), MLP(3, 2, 1, [5])):
        z = stem(x)
        assert z.shape == (17, 2) and float(z.abs().max()) < 1.0 and (stem.input_dim, stem.output_dim) == (3, 2)
        assert any(isinstance(m, torch.nn.BatchNorm1d) and not m.affine for m in stem.modules())
    assert [m.out_features for m in MLP(3, 2, 2, "16,8") if isinstance(m, torch.nn.Linear)] == [16, 8, 2]
    with pytest.raises(ValueError):
        MLP(3, 2, 3, "16,8")


def test_every_environment_switch_of_the_library_is_documented():
    """The library's environment switches (getenv("WISKI_...") in csrc/, os.environ in the package) are the ones INTEGRATION.md 3 lists -- and no more
    than twelve: a knob of a measured-and-rejected variant becomes a constant, not a switch (round 6)."""
    import glob

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    found = set()
    for f in glob.glob(os.path.join(root, "online_gp_amd", "csrc", "*")):
        if f.endswith((".hip", ".h")):
            found |= set(re.findall(r'getenv\("(WISKI_[A-Z0-9_]+)"\)', open(f).read()))
    for f in glob.glob(os.path.join(root, "online_gp_amd", "**", "*.py"), recursive=True):
        found |= set(re.findall(r'environ(?:\.get)?\(?\[?"(WISKI_[A-Z0-9_]+)"', open(f).read()))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    assert len(found) <= 12, sorted(found)
    for name in sorted(found):
        assert "`" + name + "`" in doc, f"{name} is read by the library but not documented in INTEGRATION.md"
