"""The experiment grid of the reference's (missing) benchmark notebook, run on the HIP solvers.

What is recoverable from the reference's artefacts (`figures/benchmark_s{seed}_n{noise}_r1{rho1}_r2{rho2}.{png,pdf}`):
  * 80 scenarios: seeds 0-4 x noise_std {0.5, 1, 2, 5} x rho1 {0.5, 0.8} x rho2 {0.7, 0.9} of
    generate_correlated_boston_like_data;
  * four panels per scenario - L-BFGS, ISTA variants, FISTA variants, FISTA-delta variants - each solver panel with
    six curves {lasso, elasticnet} x {fixed-t1.0, armijo-t1.0, armijo-t2.0};
  * y = suboptimality f(x_k) - f*, x = iteration k, both log-scaled.
Not recoverable (the notebook is absent): the regularisation weights, iteration budget, tolerances, delta, the
preprocessing and the definition of f*.  They are parameters here and the defaults are this build's choice: f* is the
lowest objective any curve of the same regulariser reaches in the scenario, and the features are standardised
(zero mean, unit variance; b centred) - the reference's curves reach 1e-5 suboptimality in ~25-35 iterations from
~1e4, which the raw features (cond(A^T A) ~ 1e9, L ~ 9e7) cannot do with any first-order method, while standardised
features (cond ~ 6-20, f(0) = m*var(b)/2 ~ 1e4) do exactly that.  --raw runs the unstandardised data.

    python -m fastoptsolver_amd.benchmark --out DIR [--seeds 0 1] [--noise 0.5] [--plot] ...

writes DIR/benchmark_s{seed}_n{noise}_r1{rho1}_r2{rho2}.json (curves + timings + solver metrics) and, with --plot,
the .png next to it.  Every solver call goes through the package's public functions, i.e. the HIP path."""
import argparse
import itertools
import json
import os
import time

import numpy as np

from . import iterative_solvers as its
from .easy_boston_data import generate_correlated_boston_like_data
from .lbfgs import LBFGSSolver
from .objective_functions import compute_objective
from .operators import L1Prox, LeastSquares

SEEDS = (0, 1, 2, 3, 4)
NOISE = (0.5, 1.0, 2.0, 5.0)
RHO1 = (0.5, 0.8)
RHO2 = (0.7, 0.9)
REGS = ("lasso", "elasticnet")
STEP_RULES = (("fixed-t1.0", False, 1.0), ("armijo-t1.0", True, 1.0), ("armijo-t2.0", True, 2.0))
PANELS = ("L-BFGS", "ISTA", "FISTA", "FISTA-Δ")


def scenario_name(seed, noise, rho1, rho2):
    """The reference's file stem: benchmark_s0_n0.5_r10.5_r20.7."""
    return f"benchmark_s{seed}_n{noise}_r1{rho1}_r2{rho2}"


def grid(seeds=SEEDS, noise=NOISE, rho1=RHO1, rho2=RHO2):
    return list(itertools.product(seeds, noise, rho1, rho2))


def _weights(reg, alpha1, alpha2):
    return (alpha1, 0.0) if reg == "lasso" else (alpha1, alpha2)


def standardize(A, b):
    """Zero-mean / unit-variance columns, centred targets (population std, as sklearn's StandardScaler)."""
    return (A - A.mean(axis=0)) / A.std(axis=0), b - b.mean()


def run_scenario(seed, noise, rho1, rho2, *, m=1000, alpha1=1.0, alpha2=0.5, max_iter=500, tol=1e-6, delta=3.0,
                 eta=0.5, lipschitz_seed=0, raw=False):
    """All curves of one scenario.  Returns {"name", "params", "curves": {panel: {label: [f(x_k)...]}}, "fstar":
    {reg: f*}, "seconds": {panel/label: wall}, "metrics": {...}}."""
    A, b, _ = generate_correlated_boston_like_data(m=m, seed=seed, noise_std=noise, rho1=rho1, rho2=rho2)
    if not raw:
        A, b = standardize(A, b)
    prob = _core_problem(A, b)
    curves = {p: {} for p in PANELS}
    seconds, metrics = {}, {}

    def objective_curve(xs, reg, a1, a2):
        return [float(compute_objective(x, prob, None, reg, a1, a2)) for x in xs]

    # the power iteration draws from the global legacy stream like the reference; it runs once per scenario (seeded)
    # and every variant is handed the same L through the solvers' L= keyword
    np.random.seed(lipschitz_seed)
    L = its.estimate_lipschitz(prob)

    for reg in REGS:
        a1, a2 = _weights(reg, alpha1, alpha2)
        t0 = time.perf_counter()
        s = LBFGSSolver(reg, a1, a2, max_iter=max_iter).fit(prob, None)
        seconds[f"L-BFGS/{reg}"] = time.perf_counter() - t0
        curves["L-BFGS"][reg] = [float(v) for v in s.history_]
        for label, bt, tf in STEP_RULES:
            key = f"{reg}-{label}"
            kw = dict(backtracking=bt, eta=eta, t_init_factor=tf, max_iter=max_iter, tol=tol)
            t0 = time.perf_counter()
            ls = LeastSquares(prob, None, a2)
            _, log = its.ista(np.zeros(A.shape[1]), ls, ls.grad, L1Prox(a1), L + a2, return_history=True, **kw)
            seconds[f"ISTA/{key}"] = time.perf_counter() - t0
            metrics[f"ISTA/{key}"] = its.get_metrics()
            curves["ISTA"][key] = objective_curve(log["x"][1:], reg, a1, a2)
            t0 = time.perf_counter()
            _, h = its.fista(prob, None, reg, a1, a2, return_history=True, L=L, **kw)
            seconds[f"FISTA/{key}"] = time.perf_counter() - t0
            metrics[f"FISTA/{key}"] = its.get_metrics()
            curves["FISTA"][key] = [float(v) for v in h["obj"]]
            t0 = time.perf_counter()
            _, h = its.fista_delta(prob, None, reg, a1, a2, delta, return_history=True, L=L, **kw)
            seconds[f"FISTA-Δ/{key}"] = time.perf_counter() - t0
            metrics[f"FISTA-Δ/{key}"] = its.get_metrics()
            curves["FISTA-Δ"][key] = [float(v) for v in h["obj"]]

    fstar = {}
    for reg in REGS:
        vals = [min(c) for p in PANELS[1:] for k, c in curves[p].items() if k.startswith(reg) and len(c)]
        fstar[reg] = min(vals)
    # L-BFGS ignores the l1 term when it optimises (lbfgs.py:43-54) while its history includes it (:56-61): its
    # curve is measured against its own best value, like the single-curve panel of the reference's figures
    fstar["L-BFGS"] = {reg: min(curves["L-BFGS"][reg]) for reg in REGS if curves["L-BFGS"][reg]}
    return {"name": scenario_name(seed, noise, rho1, rho2),
            "params": dict(seed=seed, noise_std=noise, rho1=rho1, rho2=rho2, m=m, alpha1=alpha1, alpha2=alpha2,
                           max_iter=max_iter, tol=tol, delta=delta, eta=eta, raw=bool(raw), L=float(L)),
            "curves": curves, "fstar": fstar, "seconds": seconds,
            "metrics": {k: {kk: (float(vv) if isinstance(vv, (int, float, np.floating, np.integer)) else vv)
                            for kk, vv in v.items()} for k, v in metrics.items()}}


def _core_problem(A, b):
    from . import _core
    return _core.prepare(A, b)


def suboptimality(result, panel, label):
    """f(x_k) - f* of one curve (k = 1, 2, ...), clipped at 0."""
    c = np.asarray(result["curves"][panel][label], dtype=np.float64)
    if panel == "L-BFGS":
        return np.maximum(c - result["fstar"]["L-BFGS"][label], 0.0)
    reg = label.split("-", 1)[0]
    return np.maximum(c - result["fstar"][reg], 0.0)


def plot_scenario(result, path):
    """The reference's four-panel layout (log-log suboptimality vs k)."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    fig, axes = plt.subplots(1, 4, figsize=(20, 4))
    fig.suptitle("Scenario " + result["name"].replace("benchmark_", ""))
    titles = ("L-BFGS", "ISTA Variants", "FISTA Variants", "FISTA-Δ Variants")
    for ax, panel, title in zip(axes, PANELS, titles):
        for label in result["curves"][panel]:
            sub = suboptimality(result, panel, label)
            if len(sub):
                ax.loglog(np.arange(1, len(sub) + 1), sub, label=("L-BFGS " + label) if panel == "L-BFGS" else label)
        ax.set_title(title)
        ax.set_xlabel("Iteration k")
        ax.set_ylabel("Suboptimality")
        ax.grid(True, which="both", linestyle="--")
        ax.legend(fontsize=8)
    fig.tight_layout(rect=(0, 0, 1, 0.93))
    fig.savefig(path, dpi=120)
    plt.close(fig)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--out", default="benchmark_out")
    ap.add_argument("--seeds", type=int, nargs="*", default=list(SEEDS))
    ap.add_argument("--noise", type=float, nargs="*", default=list(NOISE))
    ap.add_argument("--rho1", type=float, nargs="*", default=list(RHO1))
    ap.add_argument("--rho2", type=float, nargs="*", default=list(RHO2))
    ap.add_argument("--m", type=int, default=1000)
    ap.add_argument("--alpha1", type=float, default=1.0)
    ap.add_argument("--alpha2", type=float, default=0.5)
    ap.add_argument("--max-iter", type=int, default=500)
    ap.add_argument("--tol", type=float, default=1e-6)
    ap.add_argument("--raw", action="store_true", help="do not standardise the features")
    ap.add_argument("--delta", type=float, default=3.0)
    ap.add_argument("--plot", action="store_true")
    a = ap.parse_args(argv)
    os.makedirs(a.out, exist_ok=True)
    todo = grid(a.seeds, a.noise, a.rho1, a.rho2)
    t_all = time.perf_counter()
    for i, (seed, noise, r1, r2) in enumerate(todo):
        t0 = time.perf_counter()
        res = run_scenario(seed, noise, r1, r2, m=a.m, alpha1=a.alpha1, alpha2=a.alpha2, max_iter=a.max_iter,
                           tol=a.tol, delta=a.delta, raw=a.raw)
        stem = os.path.join(a.out, res["name"])
        with open(stem + ".json", "w") as f:
            json.dump(res, f)
        if a.plot:
            plot_scenario(res, stem + ".png")
        print(f"[{i + 1}/{len(todo)}] {res['name']}: {time.perf_counter() - t0:.2f} s", flush=True)
    print(f"{len(todo)} scenarios in {time.perf_counter() - t_all:.1f} s -> {a.out}", flush=True)


if __name__ == "__main__":
    main()
