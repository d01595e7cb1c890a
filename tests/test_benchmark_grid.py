"""The experiment-grid driver: data generator pinned by the reference-made fixture (CPU), one
scenario through the HIP solvers judged against the oracle (GPU)."""
import json
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import fos_oracle as orc  # noqa: E402
from tests import _data  # noqa: E402


def test_data_generator_reproduces_the_reference_stream():
    """tests/golden/boston.npz holds A, b made by the reference's generate_correlated_boston_like_data() (defaults)."""
    from fastoptsolver_amd.easy_boston_data import generate_correlated_boston_like_data
    fx = _data.load("boston")
    A, b, x_true = generate_correlated_boston_like_data()
    assert A.shape == (1000, 5) and A.dtype == np.float64
    assert np.array_equal(A, fx["boston/A"]) and np.array_equal(b, fx["boston/b"])
    assert np.array_equal(x_true, [5.0, 0.0, -0.02, -0.05, 1.5])
    # SURVEY 8(c) known answers of the reference
    assert np.allclose(A[0], [6.01989592, -0.108975927, 305.894021, 62.9161254, 4.25320451], rtol=1e-8)
    assert b[0] == pytest.approx(26.699695501983456, rel=1e-14)
    # other arguments follow the oracle's restatement (itself pinned by the same fixture)
    A2, b2, _ = generate_correlated_boston_like_data(m=257, seed=3, noise_std=0.5, rho1=0.5, rho2=0.7)
    Ao, bo, _ = orc.boston_like_data(m=257, seed=3, noise_std=0.5, rho1=0.5, rho2=0.7)
    assert np.array_equal(A2, Ao) and np.array_equal(b2, bo)


def test_grid_is_the_reference_figure_set():
    from fastoptsolver_amd import benchmark as bm
    g = bm.grid()
    assert len(g) == 80 and len(set(g)) == 80
    assert bm.scenario_name(0, 0.5, 0.5, 0.7) == "benchmark_s0_n0.5_r10.5_r20.7"      # a file name under figures/
    assert bm.scenario_name(4, 5.0, 0.8, 0.9) == "benchmark_s4_n5.0_r10.8_r20.9"
    labels = [f"{r}-{s[0]}" for r in bm.REGS for s in bm.STEP_RULES]
    assert sorted(labels) == sorted(["lasso-fixed-t1.0", "elasticnet-fixed-t1.0", "lasso-armijo-t1.0",
                                     "elasticnet-armijo-t1.0", "lasso-armijo-t2.0", "elasticnet-armijo-t2.0"])


@pytest.mark.gpu
def test_one_scenario_on_the_gpu_matches_the_oracle(tmp_path):
    from fastoptsolver_amd import benchmark as bm
    res = bm.run_scenario(1, 2.0, 0.8, 0.7, max_iter=60, tol=0.0, raw=True)
    assert res["name"] == "benchmark_s1_n2.0_r10.8_r20.7"
    assert set(res["curves"]) == set(bm.PANELS)
    for panel in bm.PANELS[1:]:
        assert len(res["curves"][panel]) == 6
        for label, c in res["curves"][panel].items():
            assert len(c) == 60 and np.isfinite(c).all(), (panel, label)
            assert c[-1] < c[0], (panel, label)
            sub = bm.suboptimality(res, panel, label)
            assert (sub >= 0).all() and sub.min() <= sub[0]
    assert all(len(c) > 3 for c in res["curves"]["L-BFGS"].values())
    # the oracle on the same data: fixed-step FISTA objective history, lasso and elastic net
    A, b, _ = orc.boston_like_data(m=1000, seed=1, noise_std=2.0, rho1=0.8, rho2=0.7)
    for reg, a1, a2 in (("lasso", 1.0, 0.0), ("elasticnet", 1.0, 0.5)):
        _, h = orc.fista(A, b, reg, a1, a2, max_iter=60, return_history=True, L=res["params"]["L"])
        ours = np.asarray(res["curves"]["FISTA"][f"{reg}-fixed-t1.0"])
        assert np.allclose(ours, h["obj"], rtol=1e-5), reg
    # ISTA (fixed step) against the oracle's ista on the same closures
    prob = orc.FistaProblem(A, b, 1.0, 0.0)
    g = lambda x: orc.smooth_value(A, b, x, 0.0)                                   # noqa: E731
    grad = lambda x: orc.gram_gradient(A, x, b, 0.0)[0]                            # noqa: E731
    prox = lambda v, t: orc.prox_l1(v, t * 1.0)                                    # noqa: E731
    _, log = orc.ista(np.zeros(5), g, grad, prox, res["params"]["L"], max_iter=60, return_history=True)
    ref = [prob.objective_inline(x) for x in log["x"][1:]]
    assert np.allclose(res["curves"]["ISTA"]["lasso-fixed-t1.0"], ref, rtol=1e-5)
    # artefacts: JSON round trip and the four-panel figure
    p = tmp_path / (res["name"] + ".json")
    p.write_text(json.dumps(res))
    assert json.loads(p.read_text())["params"]["seed"] == 1
    bm.plot_scenario(res, str(tmp_path / (res["name"] + ".png")))
    assert (tmp_path / (res["name"] + ".png")).stat().st_size > 10_000


@pytest.mark.gpu
def test_standardised_scenario_converges_like_the_reference_figures():
    """Default preprocessing: every variant reaches 1e-5 suboptimality within tens of iterations (the reference's
    figures: ~35 ISTA / ~25 FISTA iterations), and the tolerance stop ends the runs long before max_iter."""
    from fastoptsolver_amd import benchmark as bm
    res = bm.run_scenario(0, 0.5, 0.5, 0.7)
    A, b, _ = orc.boston_like_data(m=1000, seed=0, noise_std=0.5, rho1=0.5, rho2=0.7)
    A, b = bm.standardize(A, b)
    assert np.allclose(A.mean(axis=0), 0, atol=1e-12) and np.allclose(A.std(axis=0), 1) and abs(b.mean()) < 1e-12
    for panel in bm.PANELS[1:]:
        for label, c in res["curves"][panel].items():
            sub = bm.suboptimality(res, panel, label)
            assert len(c) < 200, (panel, label, len(c))
            assert sub[-1] < 1e-3 * sub[0], (panel, label)
    # every FISTA / FISTA-delta / ISTA curve of the scenario against the oracle: same stopping iteration, same values
    L = res["params"]["L"]
    for reg, a1, a2 in (("lasso", 1.0, 0.0), ("elasticnet", 1.0, 0.5)):
        ref_prob = orc.FistaProblem(A, b, a1, a2)
        g = lambda x: orc.smooth_value(A, b, x, a2)                                # noqa: E731
        grad = lambda x: orc.gram_gradient(A, x, b, a2)[0]                         # noqa: E731
        prox = lambda v, t: orc.prox_l1(v, t * a1)                                 # noqa: E731
        for label, bt, tf in bm.STEP_RULES:
            kw = dict(backtracking=bt, t_init_factor=tf, max_iter=500, tol=1e-6, return_history=True)
            refs = {"FISTA": orc.fista(A, b, reg, a1, a2, L=L, **kw)[1]["obj"],
                    "FISTA-Δ": orc.fista_delta(A, b, reg, a1, a2, 3.0, L=L, **kw)[1]["obj"],
                    "ISTA": [ref_prob.objective_inline(x) for x in orc.ista(np.zeros(5), g, grad, prox, L + a2, **kw)[1]["x"][1:]]}
            for panel, ref in refs.items():
                ours = res["curves"][panel][f"{reg}-{label}"]
                assert len(ours) == len(ref), (panel, reg, label, len(ours), len(ref))
                assert np.allclose(ours, ref, rtol=1e-5), (panel, reg, label)
