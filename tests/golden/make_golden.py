#!/usr/bin/env python
"""Generates tests/golden/*.npz.

The reference (wjmaddox/online_gp) cannot be imported in this image: every
hot-path module imports gpytorch at top level and gpytorch is not installed
(SURVEY.md 8c).  The vectors are therefore produced by the repo's own oracle
-- the data-space exact SKI GP (oracle/dataspace.py), which is the identity the
reference's tests pin WISKI against -- on the INPUT sets the reference's tests
define.  Each file stores inputs and expected outputs only.

  case 1  tests/models/test_woodbury_gp_model.py:62-101 (dead test): 1-D stream
  case 2  tests/mlls/test_batched_woodbury_marginal_log_likelihood.py:20-44 (live)
  case 3  tests/models/test_batched_online_ski_gp_model.py:50-64 (2-output batch)
  case 4  notebooks/regression_viz_1D.ipynb cells 86-88,184,683-693 (C1 plumbing)
  case 5  small-m slices of BASELINE configs C2..C5

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import dataspace, dense_reference, spec  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def stream_case(name, gb, g, kind, ell, osc, s2, chunks_x, chunks_y, chunks_noise, Xs):
    """Expected posterior (mean, var, full cov, mll) after every chunk of the stream."""
    res = {"grid_bounds": np.asarray(gb, float), "grid_size": np.asarray(g), "kind": kind, "lengthscale": np.asarray(ell, float),
           "outputscale": float(osc), "sigma2": float(s2), "test_x": np.asarray(Xs, float), "n_chunks": len(chunks_x)}
    X = np.zeros((0, np.asarray(gb).reshape(-1, 2).shape[0])); y = np.zeros(0); nz = np.zeros(0)
    for i, (cx, cy, cn) in enumerate(zip(chunks_x, chunks_y, chunks_noise)):
        cx = np.asarray(cx, float).reshape(len(cy), -1)
        X = np.concatenate([X, cx]); y = np.concatenate([y, np.asarray(cy, float)]); nz = np.concatenate([nz, np.asarray(cn, float)])
        O = dataspace.DataSpaceGP(gb, g, kind, ell, osc, s2).fit(X, y, nz)
        mean, cov = O.predict(Xs, full_cov=True)
        res[f"x_{i}"], res[f"y_{i}"], res[f"noise_{i}"] = cx, np.asarray(cy, float), np.asarray(cn, float)
        res[f"mean_{i}"], res[f"cov_{i}"], res[f"mll_{i}"] = mean, cov, O.mll()
    np.savez(os.path.join(OUT, name + ".npz"), **res)
    return res


def main():
    sp0 = spec.SOFTPLUS0
    # ---- case 1: dead-test 1-D known-input vectors
    xs = np.array([2.0, 3.0, 4.0, 1.0, 7.0])
    labels = np.sin(xs) + np.array([0.1, 0.2, -0.1, -0.2, -0.2])
    newp = np.array([2.4, 4.7])
    cx = [xs, newp, np.array([2.3]), np.array([4.1]), np.array([4.3])]
    cy = [labels, np.sin(newp) + np.array([0.1, -0.15]), np.sin(np.array([2.3])), np.sin(np.array([4.1])) + 1, np.sin(np.array([4.3]))]
    cn = [np.ones_like(c) for c in cy]
    stream_case("case1_1d_stream", [[-4.0, 14.0]], 20, "rbf", [10.0], 1.0, 0.01, cx, cy, cn, np.array([[5.0], [8.0]]))

    # ---- case 2: live MLL test inputs (fp64, seed 10), 1 and 3 outputs, fixed noise 0.1, sigma2 = 1
    torch.set_default_dtype(torch.float64)
    torch.random.manual_seed(10)
    train_x = torch.rand(10, 2)
    train_y = torch.sin(2 * train_x[:, 0] + 3 * train_x[:, 1]).unsqueeze(-1)
    train_y3 = torch.cat((train_y, train_y + 0.3 * torch.randn_like(train_y), train_y + 0.3 * torch.randn_like(train_y)), dim=1)
    X = train_x.numpy(); Y3 = train_y3.numpy()
    res = {"x": X, "y": Y3, "noise": 0.1 * np.ones_like(Y3), "grid_bounds": np.array([[0.0, 1.0], [0.0, 1.0]]), "grid_size": 5,
           "lengthscale": sp0, "outputscale": sp0}
    Xs = np.random.default_rng(5).uniform(0, 1, (6, 2))
    res["test_x"] = Xs
    for o in range(3):
        O = dataspace.DataSpaceGP(res["grid_bounds"], 5, "rbf", sp0, sp0, 1.0).fit(X, Y3[:, o], 0.1 * np.ones(10))
        res[f"mll_{o}"] = O.mll()
        res[f"mean_{o}"], res[f"cov_{o}"] = O.predict(Xs, full_cov=True)
        # finite-difference gradients of the MLL w.r.t. (log lengthscale_0, log lengthscale_1, log outputscale)
        grads = []
        for k in range(3):
            vals = []
            for sgn in (+1, -1):
                th = np.log(np.array([sp0, sp0, sp0])); th[k] += sgn * 1e-5
                e = np.exp(th)
                vals.append(dataspace.DataSpaceGP(res["grid_bounds"], 5, "rbf", e[:2], e[2], 1.0).fit(X, Y3[:, o], 0.1 * np.ones(10)).mll())
            grads.append((vals[0] - vals[1]) / 2e-5)
        res[f"dmll_dlog_{o}"] = np.array(grads)
    np.savez(os.path.join(OUT, "case2_mll_2d.npz"), **res)

    # ---- case 3: 2-output batch with heteroscedastic noise (shape test set-up, seeded)
    rng = np.random.default_rng(3)
    tx = rng.uniform(0, 1, (10, 1))
    ty = np.stack([np.sin(3 * tx[:, 0]), np.sin(5 * tx[:, 0])], 1)
    tv = 0.01 * ty ** 2 + 1e-3
    res = {"x": tx, "y": ty, "noise": tv, "grid_bounds": np.array([[0.0, 1.0]]), "grid_size": 10, "test_x": rng.uniform(0, 1, (5, 1))}
    for o in range(2):
        O = dataspace.DataSpaceGP(res["grid_bounds"], 10, "rbf", sp0, sp0, 1.0).fit(tx, ty[:, o], tv[:, o])
        res[f"mean_{o}"], res[f"cov_{o}"] = O.predict(res["test_x"], full_cov=True)
        res[f"mll_{o}"] = O.mll()
    np.savez(os.path.join(OUT, "case3_batch_1d.npz"), **res)

    # ---- case 4: C1 plumbing: sin(4x) + 0.4 N(0,1) on linspace(-1,1,39), init 10, chunks of 10
    rng = np.random.default_rng(4)
    x = np.linspace(-1, 1, 39)
    perm = rng.permutation(39)
    x = x[perm]
    y = np.sin(4 * x) + 0.4 * rng.standard_normal(39)
    s2 = sp0 + 1e-4
    for g in (12, 64):
        cx = [x[:10], x[10:20], x[20:30], x[30:]]
        cy = [y[:10], y[10:20], y[20:30], y[30:]]
        stream_case(f"case4_c1_g{g}", [[-1.1, 1.1]], g, "rbf", [sp0], sp0, s2, cx, cy, [np.ones_like(c) for c in cy],
                    np.linspace(-1, 1, 9)[:, None])

    # ---- case 5: small-m slices of C2..C5, 256 streamed points each
    def synth(d, n, seed, lo=-1.0, hi=1.0):
        r = np.random.default_rng(seed)
        X = r.uniform(lo, hi, (n, d))
        y = np.sin(2 * np.pi * X[:, 0]) * np.cos(np.pi * X[:, 1 % d]) + 0.5 * X[:, 2 % d] + 0.1 * r.standard_normal(n)
        return X, (y - y.mean()) / y.std()
    for name, d, g, kind, gb, lo, hi in [("c3_d3_g8", 3, 8, "rbf", [[-1.1, 1.1]] * 3, -1, 1), ("c2_d4_g6", 4, 6, "rbf", [[-1.1, 1.1]] * 4, -1, 1),
                                         ("c4_bo_quirk", 3, 10, "matern52", [[-32.768, 32.768]] * 3, 0, 1),
                                         ("c5_d2_g30", 2, 30, "matern12", [[0.0, 1.0]] * 2, 0, 1)]:
        X, y = synth(d, 256 + 16, 50 + d, lo, hi)
        r = np.random.default_rng(7)
        noise = r.uniform(1e-3, 0.05, 256) if name.startswith("c5") else np.ones(256)
        cx = [X[:64], X[64:128], X[128:256]]
        cy = [y[:64], y[64:128], y[128:256]]
        cn = [noise[:64], noise[64:128], noise[128:256]]
        stream_case("case5_" + name, gb, g, kind, [sp0] * d, sp0, sp0 + 1e-4, cx, cy, cn, X[256:])
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
