"""GPU: seeded randomized parity sweep — many small random shapes / parameter combinations of every solver family
against the oracle.  Aimed at masks, clamps, pipeline tails (rows per workgroup < register tiles), ragged edges,
odd restart / tolerance settings.  Tolerance 1e-5 relative on iterates as everywhere else."""
import numpy as np
import pytest
import torch

from oracle import fos_oracle as orc
from tests import _data

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def fos():
    import fastoptsolver_amd as f
    torch.cuda.set_device(0)
    return f


def _problem(rng):
    # shapes that hit: 1 row, fewer rows than workgroups, rows % R != 0, n in every menu bucket, ragged n (fallback)
    m = int(rng.choice([1, 2, 3, 5, 17, 64, 255, 256, 257, 511, 1000, 2049]))
    n = int(rng.choice([4, 8, 12, 16, 100, 256, 1000, 1024, 1028, 2048, 4096, 4100, 7, 33, 129]))
    A = rng.standard_normal((m, n))
    xt = np.zeros(n)
    k = max(1, n // 10)
    xt[rng.choice(n, k, replace=False)] = rng.standard_normal(k)
    b = A @ xt + 0.1 * rng.standard_normal(m)
    A32, b32 = A.astype(np.float32), b.astype(np.float32)
    return A32, b32, A32.astype(np.float64), b32.astype(np.float64)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("FOS_FUZZ_SEEDS", "60"))))
def test_random_fista_family(fos, seed):
    rng = np.random.default_rng(1000 + seed)
    A32, b32, A, b = _problem(rng)
    m, n = A.shape
    lam = float(np.max(np.abs(A.T @ b))) or 1.0
    a1 = float(rng.choice([0.0, 0.3, 0.05, 1e-3])) * lam
    a2 = float(rng.choice([0.0, 0.5, 10.0]))
    L = float(np.linalg.norm(A, 2) ** 2) or 1.0
    iters = int(rng.integers(1, 30))
    kind = int(rng.integers(0, 5))
    prob = fos.prepare(A32, b32)
    if kind == 0:       # plain fista, with and without history
        x = fos.fista(prob, None, "elasticnet", a1, a2, max_iter=iters, L=L)
        x_ref = orc.fista(A, b, "elasticnet", a1, a2, max_iter=iters, L=L)
        xh, h = fos.fista(prob, None, "elasticnet", a1, a2, max_iter=iters, L=L, return_history=True)
        _, h_ref = orc.fista(A, b, "elasticnet", a1, a2, max_iter=iters, L=L, return_history=True)
        # Objective resolution of a float32 pass over A: r = Ax - b is formed with an absolute error e, |e| ~ c*eps32*|b|
        # (the products cancel against b), so 0.5|r|^2 carries |r||e| + 0.5|e|^2 = 2c*eps32*sqrt(obj*obj0) + c^2*eps32^2*
        # obj0.  It only shows when the fit becomes near-exact (seed 613: m = 256 < n = 4100, no regularisation, the
        # objective falls from 5e4 to 2e-6 in 26 iterations); the ITERATES keep the 1e-5 bound throughout.
        ref = np.asarray(h_ref["obj"])
        obj0 = 0.5 * float(b @ b)
        bound = TOL * np.abs(ref) + 1e-6 * np.sqrt(np.abs(ref) * obj0) + 1e-12 * obj0
        assert (np.abs(np.asarray(h["obj"]) - ref) <= bound).all() and _data.rel(xh, x_ref) < TOL
    elif kind == 1:     # adaptive restart with a random threshold, t_init_factor
        kw = dict(adaptive_restart=True, restart_threshold=float(rng.choice([0.5, 1.0, 2.0])),
                  t_init_factor=float(rng.choice([1.0, 0.5])))
        x = fos.fista(prob, None, "lasso", a1, a2, max_iter=iters, L=L, **kw)
        x_ref = orc.fista(A, b, "lasso", a1, a2, max_iter=iters, L=L, **kw)
    elif kind == 2:     # fista_delta
        d = float(rng.choice([2.5, 3.0, 10.0]))
        x = fos.fista_delta(prob, None, "elasticnet", a1, a2, d, max_iter=iters, L=L)
        x_ref = orc.fista_delta(A, b, "elasticnet", a1, a2, d, max_iter=iters, L=L)
    elif kind == 3:     # stopping rules on the device (tol_ratio) / host (tol)
        kw = dict(tol_ratio=float(rng.choice([0.3, 0.9]))) if rng.random() < 0.5 else dict(tol=1e-3 * float(np.linalg.norm(b)))
        x = fos.fista(prob, None, "lasso", a1, a2, max_iter=60, L=L, check_every=int(rng.integers(1, 7)), **kw)
        x_ref = orc.fista(A, b, "lasso", a1, a2, max_iter=60, L=L, **kw)
    else:               # ista with elastic-net prox from a random start
        x0 = rng.standard_normal(n)
        ls = fos.LeastSquares(prob, None, 0.0)
        x = fos.ista(x0, ls, ls.grad, fos.ElasticNetProx(a1, a2), L, max_iter=iters)
        x_ref = orc.ista(x0, lambda z: orc.smooth_value(A, b, z, 0.0), lambda z: orc.gram_gradient(A, z, b, 0.0)[0],
                         lambda v, t: orc.prox_elastic_net(v, t, a1, a2), L, max_iter=iters)
    den = max(np.linalg.norm(x_ref), 1e-12)
    assert np.linalg.norm(np.asarray(x) - x_ref) / den < TOL, (seed, kind, m, n, a1, a2, iters, prob.plan())


@pytest.mark.parametrize("seed", range(8))
def test_random_lbfgs(fos, seed):
    rng = np.random.default_rng(2000 + seed)
    m = int(rng.choice([64, 300, 1000]))
    n = int(rng.choice([8, 64, 129, 512]))
    A = rng.standard_normal((m, n)).astype(np.float32)
    b = rng.standard_normal(m).astype(np.float32)
    a2 = float(rng.choice([0.5, 5.0]))
    s = fos.LBFGSSolver("ridge", 0.0, a2, max_iter=12).fit(A, b)
    s_ref = orc.LBFGSSolver("ridge", 0.0, a2, max_iter=12).fit(A.astype(np.float64), b.astype(np.float64))
    # the oracle runs on the same fp32-representable A and b, so the two runs differ by summation order only
    assert (s.nit_, s.nfev_, s.task_) == (s_ref.nit_, s_ref.nfev_, s_ref.task_), (seed, s.task_, s_ref.task_)
    for i in range(s.nit_):
        assert _data.rel(s.iterates_[i], s_ref.iterates_[i]) < 1e-5, (seed, i)
    # closed form: (A^T A + a2 I)^-1 A^T b  (SURVEY 4 cross-check) when the run converged
    if s.task_.startswith("CONVERGENCE"):
        A64 = A.astype(np.float64)
        x_star = np.linalg.solve(A64.T @ A64 + a2 * np.eye(n), A64.T @ b.astype(np.float64))
        assert _data.rel(s.x_, x_star) < 1e-4      # distance of L-BFGS's own stopping point (factr 1e7) from the optimum


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("FOS_FUZZ_LBFGS_SEEDS", "10"))))
def test_random_lbfgs_wide(fos, seed):
    """Longer vectors: the whole-chip direction kernels (n >= 2048: Gram matrix + coefficient recursion), more than one
    column chunk per workgroup, ragged n (two-pass / padded passes), bf16 storage, the full 10-pair memory with ring
    wrap-around (25 iterations), the speculative first trial of every iteration - counts, exit and every iterate
    against the oracle on the stored A."""
    rng = np.random.default_rng(2500 + seed)
    n = int(rng.choice([2048, 2052, 3001, 4100, 8192, 16388]))
    m = int(rng.choice([n // 4, n // 2, n, 2 * n]))
    kind = str(rng.choice(["f32", "f32", "bf16"]))
    A = rng.standard_normal((m, n)).astype(np.float32)
    At = torch.as_tensor(A).to(torch.bfloat16 if kind == "bf16" else torch.float32).cuda()
    A64 = At.to(torch.float64).cpu().numpy()
    b32 = rng.standard_normal(m).astype(np.float32)
    a2 = float(rng.choice([1.0, 30.0, 1000.0]))
    reg = str(rng.choice(["ridge", "elasticnet"]))
    s = fos.LBFGSSolver(reg, 0.3, a2, max_iter=25).fit(At, torch.as_tensor(b32).cuda())
    s_ref = orc.LBFGSSolver(reg, 0.3, a2, max_iter=25).fit(A64, b32.astype(np.float64))
    assert (s.nit_, s.nfev_, s.task_) == (s_ref.nit_, s_ref.nfev_, s_ref.task_), (seed, m, n, kind, s.task_, s_ref.task_)
    for i in range(s.nit_):
        xi = s.iterates_[i].detach().cpu().numpy().astype(np.float64)
        assert _data.rel(xi, s_ref.iterates_[i]) < 1e-5, (seed, m, n, kind, i)
    assert np.allclose(s.history_, s_ref.history_, rtol=1e-6), (seed, m, n, kind)


_BT_COVERAGE = [0, 0]      # iterations with compared shrink counts / all iterations, over the sweep


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("FOS_FUZZ_BT_SEEDS", "40"))))
def test_random_backtracking(fos, seed):
    """Armijo backtracking (matrix-core batch where the shape allows, one candidate per pass otherwise) against the
    oracle: per-iteration shrink counts and iterates, judged up to the reference's first float64-noise event (a step
    underflow: one search shrinking t by more than 1e-7, after which the reference is frozen at t ~ 1e-17 and its
    own comparisons are rounding noise; or an iteration that changes the objective by < 1e-10 relative; see
    tests/test_gpu_parity.py::_check_linesearch_counts).

    Iterate errors are measured against the largest iterate norm the trajectory has reached: with t_init_factor > 1
    the iterates can grow geometrically until the first rejection, and the accepted step then cancels y against
    t*grad (seed 39: |y| = 2.1e3 -> |x| = 46), which amplifies the 2e-7 relative rounding of a float32 gradient by
    the cancellation factor.  The error stays at 2e-7 of the trajectory's scale."""
    from fastoptsolver_amd import iterative_solvers as its
    rng = np.random.default_rng(3000 + seed)
    A32, b32, A, b = _problem(rng)
    n = A.shape[1]
    lam = float(np.max(np.abs(A.T @ b))) or 1.0
    a1 = float(rng.choice([0.0, 0.3, 0.05, 1e-3])) * lam
    a2 = float(rng.choice([0.0, 0.5, 10.0]))
    L = float(np.linalg.norm(A, 2) ** 2) or 1.0
    iters = int(rng.integers(2, 25))
    eta = float(rng.choice([0.5, 0.7, 0.9]))
    tf = float(rng.choice([1.0, 2.0, 4.0]))
    delta = float(rng.choice([0.0, 3.0]))               # 0 -> fista, else fista_delta
    kw = dict(backtracking=True, eta=eta, t_init_factor=tf, max_iter=iters, return_history=True)
    prob = fos.prepare(A32, b32)
    if delta:
        x, h = fos.fista_delta(prob, None, "elasticnet", a1, a2, delta, L=L, **kw)
    else:
        x, h = fos.fista(prob, None, "elasticnet", a1, a2, L=L, **kw)
    ours = list(its.ls_call_iters)
    ref = orc.FistaProblem(A, b, a1, a2)
    st = ref.init_state(L, tf)
    robust = []                                   # per iteration: were all of its Armijo comparisons decidable?
    eps64 = float(np.finfo(np.float64).eps)
    compared = [0, 0]                             # iterations whose shrink count was compared / Armijo comparisons seen

    def backtrack(y, g, tau, eta_):
        """FistaProblem.backtrack (iterative_solvers.py:183-197) that also records whether each comparison is decidable
        on the device.  With d = x_tmp - y the test is  S = (1-C) g.d + 0.5|A d|^2 + 0.5 a2 |d|^2 <= 0.  Backtracking
        runs take the gradient from the fp64-accumulating pass at the unrounded y (fos_fista_set_precise), so g.d - the
        cancelling sum - is fp64-accurate; what is left of float32 is |A d|^2, one residual pass on the fp32-rounded d
        (relative error ~1e-7 of that term, no cancellation).  A comparison counts as decidable when |S| exceeds
          (a) 2e-6 of the size of its terms (round 1, fp32 gradient: 1e-5 of the terms AND 4*eps32*|A|(|Ay|+|b|)*|d|,
              which excluded every comparison near a fit - 26 % of all iterations of this sweep),
          (b) the reference's own float64 rounding of g.d, 64*eps64*|g||d|,
        and when d is not itself rounding noise (|d| > 1e-9|y|: seeds 611 / 657, a converged one-row problem where the
        reference's own lhs - rhs has the opposite sign of S)."""
        shrinks, ok = 0, True
        g_norm = float(np.linalg.norm(g))
        while True:
            cand = ref.prox(y - tau * g, tau)
            d = cand - y
            Ad = A @ d
            terms = [(1.0 - orc.ARMIJO_C) * float(g @ d), 0.5 * float(Ad @ Ad), 0.5 * a2 * float(d @ d)]
            nd = float(np.linalg.norm(d))
            ok = ok and abs(sum(terms)) > max(2e-6 * sum(abs(t) for t in terms), 64.0 * eps64 * g_norm * nd) \
                and nd > 1e-9 * float(np.linalg.norm(y))
            compared[1] += 1
            if orc.smooth_value(A, b, cand, a2) <= orc.smooth_value(A, b, y, a2) + orc.ARMIJO_C * float(g @ d):
                break
            tau *= eta_
            shrinks += 1
        ref.metrics.ls_iters.append(shrinks)
        ref.metrics.ls_times.append(0.0)
        robust.append(ok)
        return tau

    ref.backtrack = backtrack
    xs, objs = [], []
    for _ in range(iters):
        (ref.step_delta(st, delta, backtracking=True, eta=eta) if delta else ref.step(st, backtracking=True, eta=eta))
        xs.append(st.x.copy())
        objs.append(ref.objective_inline(st.x))
    counts = ref.metrics.ls_iters
    assert len(ours) == len(counts) == iters
    obj_prev, good = 0.5 * float(b @ b), iters
    for k in range(iters):
        progress = abs(objs[k] - obj_prev) / max(abs(objs[k]), 1e-300)
        obj_prev = objs[k]
        if eta ** counts[k] < 1e-7 or progress < 1e-10:
            assert progress < 1e-10 or eta ** ours[k] < 1e-5, (seed, k, ours[k], counts[k])   # an underflow here too
            good = k
            break
        if not robust[k]:          # a comparison below the resolution of a float32 pass over A: either outcome is
            good = k               # legitimate, and the runs may part ways from here on
            break
        assert ours[k] == counts[k], (seed, k, ours[:k + 1], counts[:k + 1], A.shape, a1, a2, eta, tf, delta)
        compared[0] += 1
    _BT_COVERAGE[0] += compared[0]
    _BT_COVERAGE[1] += iters
    if good > 0:
        xk = h["x"][good if not delta else good - 1]         # fista's history starts with x0, fista_delta's with x1
        den = max(max(np.linalg.norm(v) for v in xs[:good]), 1e-12)
        assert np.linalg.norm(np.asarray(xk) - xs[good - 1]) / den < TOL, (seed, good, A.shape, a1, a2, eta, tf, delta)


def test_backtracking_sweep_coverage_report():
    """Runs after the sweep below (file order): how much of it was compared."""
    print(f"backtracking sweep: shrink counts compared on {_BT_COVERAGE[0]} of {_BT_COVERAGE[1]} iterations")


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("FOS_FUZZ_SHAPE_SEEDS", "40"))))
def test_random_shapes_and_layouts_gemv_pair(fos, seed):
    """The single pass over A through whatever kernel the planner picks - resident / row-per-thread (n <= 64),
    chunk-per-lane (aligned rows up to 128 columns), one-wave-per-row (..512), streaming geometries of 1..8 chunks per
    thread, two-pass - on random shapes, row strides and alignments."""
    rng = np.random.default_rng(5000 + seed)
    n = int(rng.choice([1, 2, 3, 5, 8, 13, 16, 31, 32, 33, 63, 64, 65, 100, 128, 255, 256, 260, 511, 512, 516, 700, 1024, 1500,
                        68, 96, 120, 2052, 2560, 3000, 5000, 6148, 10000, 12288, 14000]))    # round 3: 3/5/7-chunk geometries, 65..128
    m = int(rng.choice([1, 2, 7, 64, 255, 256, 257, 1000, 4096, 4097, 10000, 30001]))
    pad = int(rng.choice([0, 0, 1, 3, 4, 8]))                 # extra floats between rows (lda = n + pad)
    off = int(rng.choice([0, 0, 1, 4]))                       # start offset in floats (16-byte alignment or not)
    flat = torch.randn(off + m * (n + pad) + 8, device="cuda")
    At = flat[off: off + m * (n + pad)].view(m, n + pad)[:, :n]
    A = At.cpu().numpy().astype(np.float64)
    b = rng.standard_normal(m).astype(np.float32)
    y = rng.standard_normal(n).astype(np.float32)
    prob = fos.prepare(At, b)
    g = prob.gemv_pair(torch.as_tensor(y).cuda(), alpha2=0.25).cpu().numpy()
    g_ref, rr_ref = orc.gram_gradient(A, y.astype(np.float64), b.astype(np.float64), 0.25)
    # float32 resolution of the pass: r_i = A_i.y - b_i is formed with an absolute error of a few eps32 * (|A_i|.|y| +
    # |b_i|) however small r_i itself is (m = 1, 2 rows with a nearly exact fit: seeds 438, 1409, 1444 of a 2000-case
    # soak), grad = A^T r inherits it times |A|, ||r||^2 inherits 2|r| times it
    eps32 = float(np.finfo(np.float32).eps)
    dr = 4.0 * eps32 * float(np.linalg.norm(np.abs(A) @ np.abs(y.astype(np.float64)) + np.abs(b)))
    r_norm = float(np.sqrt(rr_ref))
    g_tol = 2e-6 * float(np.linalg.norm(g_ref)) + float(np.linalg.norm(A, 2) if m * n <= 1 << 16 else np.linalg.norm(A)) * dr
    assert np.linalg.norm(g - g_ref) <= g_tol, (seed, m, n, pad, off, prob.plan())
    rr = prob.residual_objective(torch.as_tensor(y).cuda())[0]
    assert abs(rr - rr_ref) <= 5e-6 * rr_ref + 2.0 * r_norm * dr + dr * dr, (seed, m, n, pad, off, prob.plan())


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("FOS_FUZZ_PATH_SEEDS", "16"))))
def test_random_regularisation_paths(fos, seed):
    """fista_path on random shapes / storage types / weight counts: the multi-vector VALU pass (<= 4 weights, fp32,
    n <= 8192), the matrix-core pass (3..16 weights; partial tiles in rows and columns, several row splits), one-by-one
    fall-backs (ragged n, n <= 64) - every solution against the oracle on the stored A."""
    rng = np.random.default_rng(7000 + seed)
    m = int(rng.choice([3, 64, 130, 257, 1000, 2049, 5000]))
    n = int(rng.choice([8, 64, 72, 128, 264, 1000, 1024, 2056, 4096]))
    kind = str(rng.choice(["f32", "f32", "bf16"]))
    nlam = int(rng.choice([2, 3, 4, 5, 9, 16]))
    iters = int(rng.integers(3, 20))
    A = rng.standard_normal((m, n)).astype(np.float32)
    At = torch.as_tensor(A).to(torch.bfloat16 if kind == "bf16" else torch.float32).cuda()
    A64 = At.to(torch.float64).cpu().numpy()
    b32 = rng.standard_normal(m).astype(np.float32)
    b = b32.astype(np.float64)
    lam = float(np.max(np.abs(A64.T @ b))) or 1.0
    L = float(np.linalg.norm(A64, "fro") ** 2) or 1.0
    alphas = [(lam * float(rng.choice([0.5, 0.1, 0.01])) * 0.8 ** i, float(rng.choice([0.0, 0.5]))) for i in range(nlam)]
    delta = float(rng.choice([0.0, 0.0, 3.5]))
    # round 3: data-dependent control inside the lockstep (adaptive restart, ratio stop) on a third of the FISTA draws
    ctl = {}
    if not delta and rng.random() < 0.5:
        ctl = dict(adaptive_restart=bool(rng.random() < 0.7), restart_threshold=float(rng.choice([1.0, 0.9])),
                   tol_ratio=float(rng.choice([0.0, 0.6, 0.9])))
    xs, info = fos.fista_path(fos.prepare(At, b32), None, alphas, max_iter=iters, L=L, return_info=True, **ctl,
                              **({"delta": delta} if delta else {}))
    assert len(xs) == nlam
    for (a1, a2), x, (k_done, _code) in zip(alphas, xs, info):
        if delta:
            x_ref = orc.fista_delta(A64, b, "elasticnet", a1, a2, delta, max_iter=iters, L=L)
        else:
            x_ref, h_ref = orc.fista(A64, b, "elasticnet", a1, a2, max_iter=iters, L=L, return_history=True, **ctl)
            assert k_done == len(h_ref["obj"]), (seed, m, n, kind, nlam, iters, a1, a2, ctl, k_done, len(h_ref["obj"]))
        x = x.detach().cpu().numpy().astype(np.float64) if torch.is_tensor(x) else np.asarray(x)
        den = float(np.linalg.norm(x_ref))
        assert np.linalg.norm(x - x_ref) <= 1e-5 * den + 1e-12, (seed, m, n, kind, nlam, iters, a1, a2, delta, ctl)
