#!/usr/bin/env python3
"""Generate tests/golden/*.npz by importing and RUNNING the reference (ElBaldo1/FastOptSolver).

Run in the build container only (the reference does not travel to the GPU box):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python3 /root/repo/tests/golden/make_golden.py

Environment when the committed fixtures were made: Python 3.10.12, NumPy 2.2.6, SciPy 1.15.3.
Each fixture holds inputs (A, b, scalars, the power-iteration start vector v0 that the reference
drew from the global legacy NumPy stream) and the reference's outputs.  Nothing from the
reference's source text is stored.
"""
import json
import os
import sys

import numpy as np

REF = os.environ.get("FOS_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

import iterative_solvers as ref_its  # noqa: E402
import lbfgs as ref_lbfgs  # noqa: E402
import objective_functions as ref_obj  # noqa: E402
import prox_operators as ref_prox  # noqa: E402
import easy_boston_data as ref_data  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
KEEP = (1, 2, 10, 50)


def synth(m, n, seed, noise=0.1, density=0.05):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((m, n))
    xt = np.zeros(n)
    nz = max(1, int(round(density * n)))
    idx = rng.choice(n, size=nz, replace=False)
    xt[idx] = rng.standard_normal(nz)
    b = A @ xt + noise * rng.standard_normal(m)
    return A, b, xt


def seeded_v0(n, seed):
    """What estimate_lipschitz (iterative_solvers.py:50) will draw after np.random.seed(seed)."""
    np.random.seed(seed)
    v0 = np.random.randn(n)
    np.random.seed(seed)
    return v0


def pick(history_x, offset):
    """history_x[i] for the kept iteration numbers; offset = index of x_1 in the list."""
    last = len(history_x) - offset
    ks = sorted({k for k in KEEP if k <= last} | {last})
    return np.array(ks), np.stack([history_x[k - 1 + offset] for k in ks])


def run_fista_family(tag, A, b, store, seed=0, max_iter=60):
    n = A.shape[1]
    lam = float(np.max(np.abs(A.T @ b)))
    regs = {
        "lasso": ("lasso", 0.1 * lam, 0.0),
        "ridge": ("ridge", 0.0, 0.5),
        "enet": ("elasticnet", 0.05 * lam, 0.5),
    }
    variants = {
        "fixed": dict(),
        "bt1": dict(backtracking=True, t_init_factor=1.0),
        "bt2": dict(backtracking=True, t_init_factor=2.0),
        "fixed_rs": dict(adaptive_restart=True),
        "bt2_rs": dict(backtracking=True, t_init_factor=2.0, adaptive_restart=True, restart_threshold=1.0),
    }
    store[f"{tag}/v0"] = seeded_v0(n, seed)
    np.random.seed(seed)
    store[f"{tag}/L"] = ref_its.estimate_lipschitz(A)
    cases = []
    for rname, (reg, a1, a2) in regs.items():
        for vname, kw in variants.items():
            key = f"{tag}/fista/{rname}/{vname}"
            np.random.seed(seed)
            x, h = ref_its.fista(A, b, reg, a1, a2, max_iter=max_iter, return_history=True, **kw)
            met = ref_its.get_metrics()
            ks, xs = pick(h["x"], 1)
            store[key + "/ks"], store[key + "/xs"] = ks, xs
            store[key + "/obj"] = np.array(h["obj"])
            store[key + "/x"] = x
            store[key + "/counts"] = np.array([met["grad_num_calls"], met["ls_num_calls"], met["ls_iters_total"]])
            store[key + "/ls_iters"] = np.array(ref_its.ls_call_iters, dtype=np.int64)
            cases.append(dict(key=key, algo="fista", reg=reg, alpha1=a1, alpha2=a2, max_iter=max_iter, kw=kw))
            if "adaptive_restart" in kw:
                continue
            key = f"{tag}/fista_delta/{rname}/{vname}"
            np.random.seed(seed)
            kwd = {k: v for k, v in kw.items()}
            x, h = ref_its.fista_delta(A, b, reg, a1, a2, 3.0, max_iter=max_iter, return_history=True, **kwd)
            met = ref_its.get_metrics()
            ks, xs = pick(h["x"], 0)
            store[key + "/ks"], store[key + "/xs"] = ks, xs
            store[key + "/obj"] = np.array(h["obj"])
            store[key + "/x"] = x
            store[key + "/counts"] = np.array([met["grad_num_calls"], met["ls_num_calls"], met["ls_iters_total"]])
            store[key + "/ls_iters"] = np.array(ref_its.ls_call_iters, dtype=np.int64)
            cases.append(dict(key=key, algo="fista_delta", reg=reg, alpha1=a1, alpha2=a2, delta=3.0,
                              max_iter=max_iter, kw=kwd))
    # stopping rules (iterative_solvers.py:179, :238, :242)
    reg, a1, a2 = regs["lasso"]
    for sname, kw in {
        "tol_step": dict(tol=1e-3 * float(np.linalg.norm(b)) / lam),
        "tol_ratio": dict(tol_ratio=0.5),
        "tol_grad": dict(tol=1e9),
    }.items():
        key = f"{tag}/fista_stop/{sname}"
        np.random.seed(seed)
        x, h = ref_its.fista(A, b, reg, a1, a2, max_iter=200, return_history=True, **kw)
        store[key + "/x"] = x
        store[key + "/niter"] = np.array(len(h["obj"]))
        cases.append(dict(key=key, algo="fista", reg=reg, alpha1=a1, alpha2=a2, max_iter=200, kw=kw))
    key = f"{tag}/fista_delta_stop/tol_ratio"
    np.random.seed(seed)
    x, h = ref_its.fista_delta(A, b, reg, a1, a2, 4.0, max_iter=200, tol_ratio=0.5, return_history=True)
    store[key + "/x"] = x
    store[key + "/niter"] = np.array(len(h["obj"]))
    cases.append(dict(key=key, algo="fista_delta", reg=reg, alpha1=a1, alpha2=a2, delta=4.0, max_iter=200,
                      kw=dict(tol_ratio=0.5)))
    return cases


def run_ista(tag, A, b, store, max_iter=40):
    """ista (iterative_solvers.py:65) with least-squares callables and the reference's prox closures."""
    lam = float(np.max(np.abs(A.T @ b)))
    L = float(np.linalg.norm(A, 2) ** 2)
    store[f"{tag}/ista/L"] = L
    cases = []
    for pname, (a1, a2, in_smooth) in {
        "l1": (0.1 * lam, 0.0, False),
        "enet_prox": (0.05 * lam, 0.5, False),       # l2 inside prox_elastic_net (prox_operators.py:10)
        "enet_smooth": (0.05 * lam, 0.5, True),      # l2 inside g / grad_g, prox_l1 only
    }.items():
        s2 = a2 if in_smooth else 0.0

        def g(x, s2=s2):
            r = A @ x - b
            return 0.5 * r.dot(r) + 0.5 * s2 * x.dot(x)

        def grad_g(x, s2=s2):
            return A.T @ (A @ x - b) + s2 * x

        if pname == "enet_prox":
            prox_h = lambda v, t, a1=a1, a2=a2: ref_prox.prox_elastic_net(v, t, a1, a2)  # noqa: E731
        else:
            prox_h = lambda v, t, a1=a1: ref_prox.prox_l1(v, t * a1)  # noqa: E731
        for vname, kw in {"fixed": dict(), "bt2": dict(backtracking=True, t_init_factor=2.0)}.items():
            key = f"{tag}/ista/{pname}/{vname}"
            x, log = ref_its.ista(np.zeros(A.shape[1]), g, grad_g, prox_h, L + s2, max_iter=max_iter,
                                  return_history=True, **kw)
            met = ref_its.get_metrics()
            ks, xs = pick(log["x"], 1)
            store[key + "/ks"], store[key + "/xs"] = ks, xs
            store[key + "/t"] = np.array(log["t"])
            store[key + "/delta"] = np.array(log["delta"])
            store[key + "/x"] = x
            store[key + "/counts"] = np.array([met["grad_num_calls"], met["ls_num_calls"], met["ls_iters_total"]])
            store[key + "/ls_iters"] = np.array(ref_its.ls_call_iters, dtype=np.int64)
            cases.append(dict(key=key, algo="ista", prox=pname, alpha1=a1, alpha2=a2, in_smooth=in_smooth,
                              max_iter=max_iter, kw=kw))
    return cases


def run_lbfgs(tag, A, b, store):
    lam = float(np.max(np.abs(A.T @ b)))
    cases = []
    for rname, (reg, a1, a2) in {
        "ridge": ("ridge", 0.0, 1.0),
        "enet": ("elasticnet", 0.05 * lam, 0.5),
        "lasso": ("lasso", 0.1 * lam, 0.0),          # l1 ignored by the optimiser (lbfgs.py:49)
        "enet_tiny1": ("elasticnet", 1e-9, 0.5),     # -> ridge (lbfgs.py:21-25)
    }.items():
        key = f"{tag}/lbfgs/{rname}"
        iters = []
        orig = ref_lbfgs.compute_objective

        def spy(x, *a, **k):
            iters.append(np.array(x, copy=True))
            return orig(x, *a, **k)

        ref_lbfgs.compute_objective = spy
        try:
            s = ref_lbfgs.LBFGSSolver(reg, a1, a2).fit(A, b)
        finally:
            ref_lbfgs.compute_objective = orig
        met = ref_its.get_metrics()
        store[key + "/x"] = s.x_
        store[key + "/final_obj"] = np.array(s.final_obj_)
        store[key + "/history"] = np.array(s.history_)
        store[key + "/iterates"] = np.stack(iters)
        store[key + "/nit_nfev"] = np.array([len(s.history_), met["grad_num_calls"]])
        cases.append(dict(key=key, algo="lbfgs", reg=reg, alpha1=a1, alpha2=a2,
                          norm_reg=s.reg_type, norm_a1=s.alpha1, norm_a2=s.alpha2))
    return cases


def leaf_vectors(store):
    rng = np.random.default_rng(7)
    v = rng.standard_normal(257) * 3
    v[:5] = [0.0, -0.0, 1.5, -1.5, 1e-300]
    store["leaf/v"] = v
    store["leaf/prox_l1_0.7"] = ref_prox.prox_l1(v, 0.7)
    store["leaf/prox_l1_0"] = ref_prox.prox_l1(v, 0.0)
    store["leaf/prox_en"] = ref_prox.prox_elastic_net(v, 0.3, 2.0, 0.5)
    A = rng.standard_normal((33, 257))
    b = rng.standard_normal(33)
    store["leaf/A"], store["leaf/b"] = A, b
    store["leaf/obj"] = np.array([ref_obj.compute_objective(v, A, b, r, 0.3, 0.7)
                                  for r in ("lasso", "ridge", "elasticnet")])


def main():
    meta = {"numpy": np.__version__, "cases": {}}
    import scipy
    meta["scipy"] = scipy.__version__

    # ---- small / ragged problems, A stored ----
    for tag, (m, n, seed) in {"tiny": (64, 16, 1), "ragged": (777, 129, 2)}.items():
        store = {}
        A, b, xt = synth(m, n, seed)
        store[f"{tag}/A"], store[f"{tag}/b"] = A, b
        cases = run_fista_family(tag, A, b, store)
        cases += run_ista(tag, A, b, store)
        cases += run_lbfgs(tag, A, b, store)
        np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), **store)
        meta["cases"][tag] = dict(m=m, n=n, seed=seed, cases=cases)

    # ---- config 1: Boston-like data (easy_boston_data.py) ----
    store = {}
    A, b, xt = ref_data.generate_correlated_boston_like_data()
    store["boston/A"], store["boston/b"], store["boston/x_true"] = A, b, xt
    A2, b2, _ = ref_data.generate_correlated_boston_like_data(m=50, seed=3, noise_std=0.5, rho1=0.5, rho2=0.7)
    store["boston/A_alt"], store["boston/b_alt"] = A2, b2
    cases = []
    store["boston/v0"] = seeded_v0(5, 0)
    np.random.seed(0)
    store["boston/L"] = ref_its.estimate_lipschitz(A)
    for name, fn, args, kw in [
        ("fista_lasso", ref_its.fista, ("lasso", 1.0, 0.0), dict(max_iter=500)),
        ("fista_enet_bt", ref_its.fista, ("elasticnet", 1.0, 0.5), dict(max_iter=200, backtracking=True, t_init_factor=2.0)),
        ("fdelta_enet", ref_its.fista_delta, ("elasticnet", 1.0, 0.5, 3.0), dict(max_iter=500)),
    ]:
        key = f"boston/{name}"
        np.random.seed(0)
        x, h = fn(A, b, *args, return_history=True, **kw)
        met = ref_its.get_metrics()
        ks, xs = pick(h["x"], 1 if fn is ref_its.fista else 0)
        store[key + "/ks"], store[key + "/xs"] = ks, xs
        store[key + "/obj"] = np.array(h["obj"])
        store[key + "/x"] = x
        store[key + "/counts"] = np.array([met["grad_num_calls"], met["ls_num_calls"], met["ls_iters_total"]])
        store[key + "/ls_iters"] = np.array(ref_its.ls_call_iters, dtype=np.int64)
        cases.append(dict(key=key, algo=fn.__name__, args=list(args), kw=kw))
    cases += run_lbfgs("boston", A, b, store)
    np.savez_compressed(os.path.join(OUT, "boston.npz"), **store)
    meta["cases"]["boston"] = dict(m=1000, n=5, cases=cases)

    # ---- tile-aligned problem, A regenerated from its seed (too large to store) ----
    tag, m, n, seed = "aligned", 4096, 512, 3
    store = {}
    A, b, xt = synth(m, n, seed)
    store[f"{tag}/A_head"] = A[:2, :8].copy()
    store[f"{tag}/A_sum"] = np.array(A.sum())
    store[f"{tag}/b"] = b
    cases = run_fista_family(tag, A, b, store, max_iter=100)
    cases += run_lbfgs(tag, A, b, store)
    np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), **store)
    meta["cases"][tag] = dict(m=m, n=n, seed=seed, cases=cases)

    store = {}
    leaf_vectors(store)
    np.savez_compressed(os.path.join(OUT, "leaf.npz"), **store)

    with open(os.path.join(OUT, "cases.json"), "w") as fh:
        json.dump(meta, fh, indent=1, default=float)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
