"""Shared input builders for the tests (same generators tests/golden/make_golden.py used)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def synth(m, n, seed, noise=0.1, density=0.05, dtype=np.float64):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((m, n))
    xt = np.zeros(n)
    nz = max(1, int(round(density * n)))
    idx = rng.choice(n, size=nz, replace=False)
    xt[idx] = rng.standard_normal(nz)
    b = A @ xt + noise * rng.standard_normal(m)
    return A.astype(dtype), b.astype(dtype), xt


def load(tag):
    return np.load(os.path.join(GOLDEN, f"{tag}.npz"))


def cases(tag):
    with open(os.path.join(GOLDEN, "cases.json")) as fh:
        return json.load(fh)["cases"][tag]


def problem(tag):
    """(A, b, fixture) for a golden tag; the 'aligned' A is regenerated from its seed and checked."""
    fx = load(tag)
    if tag == "aligned":
        meta = cases(tag)
        A, b, _ = synth(meta["m"], meta["n"], meta["seed"])
        assert np.array_equal(A[:2, :8], fx["aligned/A_head"]) and np.isclose(A.sum(), fx["aligned/A_sum"], rtol=1e-13), \
            "NumPy Generator stream changed: regenerate goldens"
        assert np.array_equal(b, fx["aligned/b"])
        return A, b, fx
    return fx[f"{tag}/A"], fx[f"{tag}/b"], fx


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    den = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / den) if den > 0 else float(np.linalg.norm(a - b))
