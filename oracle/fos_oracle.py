"""CPU oracle for the FISTA / ISTA / FISTA-delta / L-BFGS hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``fastoptsolver_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg use it, and there only as the checker / reported baseline.

It is an independent float64 NumPy restatement of what the reference computes,
organised as small state machines (``FistaState.step`` ...) rather than the
reference's monolithic loops, so that one oracle step can be compared with one
device step.  Every function cites the reference ``file:line`` it restates
(paths relative to the reference checkout).

Parity pin: ``tests/golden/*.npz`` were produced by importing the reference
itself (``tests/golden/make_golden.py``, NumPy 2.2.6 / SciPy 1.15.3) and
``tests/test_oracle_golden.py`` checks this file against every one of them.

The L-BFGS arithmetic is not in the reference: ``lbfgs.py:64`` calls
``scipy.optimize.fmin_l_bfgs_b`` (SciPy is unpinned by the reference; 1.15.3
here).  ``lbfgs_minimize`` restates the published unbounded L-BFGS-B iteration
(two-loop direction + MINPACK-2 ``dcsrch``/``dcstep`` line search with L-BFGS-B's
constants) and is pinned by goldens captured from SciPy 1.15.3 through the
reference's own call site.
"""
from __future__ import annotations

import math
import time
from dataclasses import dataclass, field

import numpy as np

# Armijo sufficient-decrease constant: iterative_solvers.py:11 (module global C).
ARMIJO_C = 1e-2


# --------------------------------------------------------------------------
# leaf operators
# --------------------------------------------------------------------------
def prox_l1(v, thr):
    """Soft threshold.  prox_operators.py:3-8."""
    v = np.asarray(v, dtype=np.float64)
    mag = np.abs(v) - thr
    return np.sign(v) * np.where(mag > 0.0, mag, 0.0)


def prox_elastic_net(v, tau, alpha1, alpha2):
    """prox of alpha1*|x|_1 + 0.5*alpha2*|x|^2 with step tau.  prox_operators.py:10-16."""
    return prox_l1(v, tau * alpha1) / (1.0 + tau * alpha2)


def compute_objective(x, A, b, reg_type, alpha1, alpha2):
    """f(x) by reg_type.  objective_functions.py:3-30 (ValueError at :28)."""
    if reg_type not in ("lasso", "ridge", "elasticnet"):
        raise ValueError(f"Unsupported reg_type='{reg_type}'")
    res = A @ x - b
    val = 0.5 * float(res @ res)
    if reg_type != "lasso":
        val += 0.5 * alpha2 * float(x @ x)
    if reg_type != "ridge":
        val += alpha1 * float(np.abs(x).sum())
    return val


def gram_gradient(A, y, b=None, alpha2=0.0):
    """(A^T (A y - b) [+ alpha2 y],  ||A y - b||^2).  iterative_solvers.py:173-175, lbfgs.py:46-51."""
    res = A @ y
    if b is not None:
        res = res - b
    grad = A.T @ res
    if alpha2 > 0:
        grad = grad + alpha2 * y
    return grad, float(res @ res)


def smooth_value(A, b, z, alpha2):
    """g_smooth.  iterative_solvers.py:163-168 / :282-287."""
    res = A @ z - b
    val = 0.5 * float(res @ res)
    if alpha2 > 0:
        val += 0.5 * alpha2 * float(z @ z)
    return val


# --------------------------------------------------------------------------
# metrics (iterative_solvers.py:16-40): same seven keys
# --------------------------------------------------------------------------
@dataclass
class Metrics:
    grad_times: list = field(default_factory=list)
    ls_times: list = field(default_factory=list)
    ls_iters: list = field(default_factory=list)

    def as_dict(self):
        gt, lt = self.grad_times, self.ls_times
        return {
            "grad_num_calls": len(gt),
            "grad_time_total": sum(gt),
            "grad_time_mean": float(np.mean(gt)) if gt else 0.0,
            "ls_num_calls": len(lt),
            "ls_time_total": sum(lt),
            "ls_time_mean": float(np.mean(lt)) if lt else 0.0,
            "ls_iters_total": sum(self.ls_iters),
        }


# --------------------------------------------------------------------------
# power iteration
# --------------------------------------------------------------------------
def estimate_lipschitz(A, n_iter=100, tol=1e-6, v0=None, return_calls=False):
    """lambda_max(A^T A) by power iteration.  iterative_solvers.py:45-60.

    ``v0`` None draws ``np.random.randn(n)`` from the *global legacy* stream
    exactly like :50 does, so seeding ``np.random.seed(s)`` before the call
    reproduces the reference's estimate.
    """
    n = A.shape[1]
    v = np.random.randn(n) if v0 is None else np.array(v0, dtype=np.float64)
    v = v / np.linalg.norm(v)
    last, L, calls = 0.0, None, 0
    for _ in range(n_iter):
        w = A.T @ (A @ v)
        calls += 1
        L = float(np.linalg.norm(w))
        v = w / L
        if abs(L - last) < tol:
            break
        last = L
    return (L, calls) if return_calls else L


# --------------------------------------------------------------------------
# FISTA / FISTA-delta as a state machine
# --------------------------------------------------------------------------
@dataclass
class FistaState:
    """Loop-carried state of iterative_solvers.py:149-161 (fista) / :269-280 (fista_delta)."""
    x: np.ndarray          # x_k
    x_old: np.ndarray      # x_{k-1}
    y: np.ndarray          # y_k
    t: float               # t_prev (FISTA momentum scalar)
    tau: float             # current step (persists through backtracking, :197)
    k: int = 0             # completed iterations
    stopped: bool = False


class FistaProblem:
    """One composite problem 0.5||Ax-b||^2 + 0.5 a2 ||x||^2 + a1 ||x||_1 and the FISTA family on it."""

    def __init__(self, A, b, alpha1, alpha2):
        self.A = np.asarray(A, dtype=np.float64)
        self.b = np.asarray(b, dtype=np.float64)
        self.a1 = float(alpha1)
        self.a2 = float(alpha2)
        self.metrics = Metrics()

    # -- pieces ----------------------------------------------------------
    def init_state(self, L, t_init_factor=1.0):
        """iterative_solvers.py:149-158: zeros, t=1, L += alpha2 if alpha2>0, tau=t_init_factor/L."""
        n = self.A.shape[1]
        if self.a2 > 0:
            L = L + self.a2
        z = np.zeros(n)
        return FistaState(x=z.copy(), x_old=z.copy(), y=z.copy(), t=1.0, tau=t_init_factor / L)

    def gradient(self, y):
        t0 = time.perf_counter()
        g, _ = gram_gradient(self.A, y, self.b, self.a2)
        self.metrics.grad_times.append(time.perf_counter() - t0)
        return g

    def prox(self, v, step):
        """:201 — prox only when alpha1 > 0."""
        return prox_l1(v, step * self.a1) if self.a1 > 0 else v

    def backtrack(self, y, g, tau, eta):
        """Armijo loop.  iterative_solvers.py:183-197 (fista) / :298-312 (fista_delta)."""
        t0 = time.perf_counter()
        shrinks = 0
        while True:
            cand = self.prox(y - tau * g, tau)
            lhs = smooth_value(self.A, self.b, cand, self.a2)
            rhs = smooth_value(self.A, self.b, y, self.a2) + ARMIJO_C * float(g @ (cand - y))
            if lhs <= rhs:
                break
            tau *= eta
            shrinks += 1
        self.metrics.ls_times.append(time.perf_counter() - t0)
        self.metrics.ls_iters.append(shrinks)
        return tau

    # -- one FISTA iteration (iterative_solvers.py:170-242) -----------------
    def step(self, s: FistaState, *, backtracking=False, eta=0.5, tol=0.0, tol_ratio=0.0,
             adaptive_restart=False, restart_threshold=1.0):
        g = self.gradient(s.y)
        if tol > 0.0 and np.linalg.norm(g) < tol:        # :179 — before the update
            s.stopped = True
            return None
        if backtracking:
            s.tau = self.backtrack(s.y, g, s.tau, eta)
        x_new = self.prox(s.y - s.tau * g, s.tau)          # :200-201
        move = float(np.linalg.norm(x_new - s.x))          # :204
        move_old = float(np.linalg.norm(s.x - s.x_old))    # :205
        ratio = move / move_old if move_old > 0 else math.inf
        if adaptive_restart and ratio > restart_threshold:  # :209-213
            t_new, y_new = 1.0, x_new.copy()
        else:                                               # :214-221
            t_new = 0.5 * (1.0 + math.sqrt(1.0 + 4.0 * s.t * s.t))
            y_new = x_new + ((s.t - 1.0) / t_new) * (x_new - s.x)
        s.x_old, s.x, s.y, s.t = s.x, x_new, y_new, t_new   # :235
        s.k += 1
        if tol > 0.0 and move < tol:                        # :238
            s.stopped = True
        if tol_ratio > 0.0 and ratio < tol_ratio:           # :242
            s.stopped = True
        return {"move": move, "ratio": ratio}

    # -- one FISTA-delta iteration (iterative_solvers.py:289-342) ------------
    def step_delta(self, s: FistaState, delta, *, backtracking=False, eta=0.5, tol=0.0, tol_ratio=0.0):
        g = self.gradient(s.y)
        if backtracking:
            s.tau = self.backtrack(s.y, g, s.tau, eta)
        x_new = self.prox(s.y - s.tau * g, s.tau)
        move = float(np.linalg.norm(x_new - s.x))
        move_old = float(np.linalg.norm(s.x - s.x_old))
        ratio = move / move_old if move_old > 0 else math.inf
        kk = s.k + 1                                         # reference k runs 1..max_iter (:289)
        theta = kk / (kk + 1 + delta)                        # :330
        s.y = x_new + theta * (x_new - s.x)
        s.x_old, s.x = s.x, x_new
        s.k = kk
        if tol > 0.0 and move < tol:
            s.stopped = True
        if tol_ratio > 0.0 and ratio < tol_ratio:
            s.stopped = True
        return {"move": move, "ratio": ratio}

    def objective_inline(self, x):
        """History objective of fista.  iterative_solvers.py:225-230 (driven by alpha>0, not reg_type)."""
        res = self.A @ x - self.b
        val = 0.5 * float(res @ res)
        if self.a2 > 0:
            val += 0.5 * self.a2 * float(x @ x)
        if self.a1 > 0:
            val += self.a1 * float(np.linalg.norm(x, 1))
        return val


def fista(A, b, reg_type, alpha1, alpha2, backtracking=False, eta=0.5, t_init_factor=1.0,
          max_iter=500, tol=0.0, tol_ratio=0.0, adaptive_restart=False, restart_threshold=1.0,
          return_history=False, *, L=None, v0=None, return_metrics=False):
    """iterative_solvers.py:132-245.  ``reg_type`` is unused there and here."""
    prob = FistaProblem(A, b, alpha1, alpha2)
    if L is None:
        L = estimate_lipschitz(prob.A, v0=v0)
    st = prob.init_state(L, t_init_factor)
    hist = {"x": [st.x.copy()], "obj": []} if return_history else None
    for _ in range(max_iter):
        info = prob.step(st, backtracking=backtracking, eta=eta, tol=tol, tol_ratio=tol_ratio,
                         adaptive_restart=adaptive_restart, restart_threshold=restart_threshold)
        if info is None:
            break
        if return_history:
            hist["obj"].append(prob.objective_inline(st.x))
            hist["x"].append(st.x.copy())
        if st.stopped:
            break
    out = (st.x, hist) if return_history else st.x
    return (out, prob.metrics.as_dict()) if return_metrics else out


def fista_delta(A, b, reg_type, alpha1, alpha2, delta, backtracking=False, eta=0.5, t_init_factor=1.0,
                max_iter=500, tol=0.0, tol_ratio=0.0, return_history=False, *, L=None, v0=None,
                return_metrics=False):
    """iterative_solvers.py:251-344 (assert at :268; history without x0, objective via compute_objective :321)."""
    assert delta > 2, "In FISTA-Δ, delta must be > 2 for convergence (course requirement)"
    prob = FistaProblem(A, b, alpha1, alpha2)
    if L is None:
        L = estimate_lipschitz(prob.A, v0=v0)
    st = prob.init_state(L, t_init_factor)
    hist = {"x": [], "obj": []} if return_history else None
    for _ in range(max_iter):
        prob.step_delta(st, delta, backtracking=backtracking, eta=eta, tol=tol, tol_ratio=tol_ratio)
        if return_history:
            hist["x"].append(st.x.copy())
            hist["obj"].append(compute_objective(st.x, prob.A, prob.b, reg_type, alpha1, alpha2))
        if st.stopped:
            break
    out = (st.x, hist) if return_history else st.x
    return (out, prob.metrics.as_dict()) if return_metrics else out


# --------------------------------------------------------------------------
# ISTA over caller-supplied callables
# --------------------------------------------------------------------------
def ista(x0, g, grad_g, prox_h, L, backtracking=False, eta=0.5, t_init_factor=1.0, max_iter=500,
         tol=0.0, return_history=False, *, return_metrics=False):
    """iterative_solvers.py:65-125.  prox_h receives the step t (not t*alpha)."""
    met = Metrics()
    x = np.array(x0, dtype=np.float64, copy=True)
    step = t_init_factor / L
    log = {"x": [x.copy()], "t": [step], "delta": []} if return_history else None
    for _ in range(max_iter):
        t0 = time.perf_counter()
        gr = grad_g(x)
        met.grad_times.append(time.perf_counter() - t0)
        if backtracking:                                   # :92-108
            t0 = time.perf_counter()
            trial, shrinks = step, 0
            while True:
                cand = prox_h(x - trial * gr, trial)
                if g(cand) <= g(x) + ARMIJO_C * float(gr @ (cand - x)):
                    break
                trial *= eta
                shrinks += 1
            met.ls_times.append(time.perf_counter() - t0)
            met.ls_iters.append(shrinks)
            step = trial
        else:
            cand = prox_h(x - step * gr, step)             # :110-111
        move = float(np.linalg.norm(cand - x))
        x = cand
        if return_history:
            log["x"].append(x.copy())
            log["t"].append(step)
            log["delta"].append(move)
        if tol > 0.0 and move < tol:
            break
    out = (x, log) if return_history else x
    return (out, met.as_dict()) if return_metrics else out


# --------------------------------------------------------------------------
# More'-Thuente line search (MINPACK-2 dcsrch / dcstep), as L-BFGS-B uses it
# --------------------------------------------------------------------------
def _mt_trial(lo, f_lo, d_lo, hi, f_hi, d_hi, t, f_t, d_t, bracketed, t_min, t_max):
    """dcstep: safeguarded cubic/quadratic trial-step selection.

    Returns the updated (lo, f_lo, d_lo, hi, f_hi, d_hi, t, bracketed); ``lo`` is the
    endpoint with the least function value so far, ``hi`` the other endpoint.
    """
    sign_change = d_t * (d_lo / abs(d_lo))

    def cubic_gamma(fa, da, fb, db, a, bpt, use_min_form):
        theta = 3.0 * (fa - fb) / (bpt - a) + da + db
        s = max(abs(theta), abs(da), abs(db))
        if use_min_form:
            gam = s * math.sqrt(max(0.0, (theta / s) ** 2 - (da / s) * (db / s)))
        else:
            gam = s * math.sqrt((theta / s) ** 2 - (da / s) * (db / s))
        return theta, gam

    if f_t > f_lo:
        # case 1: higher value -> minimum bracketed; cubic vs quadratic, closer to lo wins
        theta, gam = cubic_gamma(f_lo, d_lo, f_t, d_t, lo, t, False)
        if t < lo:
            gam = -gam
        p = (gam - d_lo) + theta
        q = ((gam - d_lo) + gam) + d_t
        r = p / q
        t_cubic = lo + r * (t - lo)
        t_quad = lo + ((d_lo / ((f_lo - f_t) / (t - lo) + d_lo)) / 2.0) * (t - lo)
        if abs(t_cubic - lo) < abs(t_quad - lo):
            t_new = t_cubic
        else:
            t_new = t_cubic + (t_quad - t_cubic) / 2.0
        bracketed = True
    elif sign_change < 0.0:
        # case 2: lower value, derivatives of opposite sign -> bracketed; cubic vs secant, farther from t wins
        theta, gam = cubic_gamma(f_lo, d_lo, f_t, d_t, lo, t, False)
        if t > lo:
            gam = -gam
        p = (gam - d_t) + theta
        q = ((gam - d_t) + gam) + d_lo
        r = p / q
        t_cubic = t + r * (lo - t)
        t_sec = t + (d_t / (d_t - d_lo)) * (lo - t)
        t_new = t_cubic if abs(t_cubic - t) > abs(t_sec - t) else t_sec
        bracketed = True
    elif abs(d_t) < abs(d_lo):
        # case 3: lower value, same-sign derivative that shrinks
        theta, gam = cubic_gamma(f_lo, d_lo, f_t, d_t, lo, t, True)
        if t > lo:
            gam = -gam
        p = (gam - d_t) + theta
        q = (gam + (d_lo - d_t)) + gam
        r = p / q
        if r < 0.0 and gam != 0.0:
            t_cubic = t + r * (lo - t)
        elif t > lo:
            t_cubic = t_max
        else:
            t_cubic = t_min
        t_sec = t + (d_t / (d_t - d_lo)) * (lo - t)
        if bracketed:
            t_new = t_cubic if abs(t_cubic - t) < abs(t_sec - t) else t_sec
            if t > lo:
                t_new = min(t + 0.66 * (hi - t), t_new)
            else:
                t_new = max(t + 0.66 * (hi - t), t_new)
        else:
            t_new = t_cubic if abs(t_cubic - t) > abs(t_sec - t) else t_sec
            t_new = max(t_min, min(t_max, t_new))
    else:
        # case 4: lower value, same-sign derivative that does not shrink
        if bracketed:
            theta, gam = cubic_gamma(f_t, d_t, f_hi, d_hi, t, hi, False)
            if t > hi:
                gam = -gam
            p = (gam - d_t) + theta
            q = ((gam - d_t) + gam) + d_hi
            r = p / q
            t_new = t + r * (hi - t)
        elif t > lo:
            t_new = t_max
        else:
            t_new = t_min

    # update the interval that contains a minimiser
    if f_t > f_lo:
        hi, f_hi, d_hi = t, f_t, d_t
    else:
        if sign_change < 0.0:
            hi, f_hi, d_hi = lo, f_lo, d_lo
        lo, f_lo, d_lo = t, f_t, d_t
    return lo, f_lo, d_lo, hi, f_hi, d_hi, t_new, bracketed


class MoreThuente:
    """Reverse-communication dcsrch.  ``start`` then repeated ``advance(f, d)``; ``task`` is
    'FG', 'CONVERGENCE', 'WARNING: ...' or 'ERROR: ...'."""

    def __init__(self, ftol=1e-3, gtol=0.9, xtol=0.1, stpmin=0.0, stpmax=1e10):
        self.ftol, self.gtol, self.xtol = ftol, gtol, xtol
        self.stpmin, self.stpmax = stpmin, stpmax
        self.task = "START"

    def start(self, stp, f0, d0):
        if stp < self.stpmin:
            self.task = "ERROR: STP .LT. STPMIN"
        elif stp > self.stpmax:
            self.task = "ERROR: STP .GT. STPMAX"
        elif d0 >= 0.0:
            self.task = "ERROR: INITIAL G .GE. ZERO"
        if self.task.startswith("ERROR"):
            return stp
        self.bracketed = False
        self.stage = 1
        self.f0, self.d0 = f0, d0
        self.dtest = self.ftol * d0
        self.width = self.stpmax - self.stpmin
        self.width1 = 2.0 * self.width
        self.lo, self.f_lo, self.d_lo = 0.0, f0, d0
        self.hi, self.f_hi, self.d_hi = 0.0, f0, d0
        self.tmin = 0.0
        self.tmax = stp + 4.0 * stp
        self.task = "FG"
        return stp

    def advance(self, stp, f, d):
        ftest = self.f0 + stp * self.dtest
        if self.stage == 1 and f <= ftest and d >= 0.0:
            self.stage = 2
        task = "FG"
        if self.bracketed and (stp <= self.tmin or stp >= self.tmax):
            task = "WARNING: ROUNDING ERRORS PREVENT PROGRESS"
        if self.bracketed and self.tmax - self.tmin <= self.xtol * self.tmax:
            task = "WARNING: XTOL TEST SATISFIED"
        if stp == self.stpmax and f <= ftest and d <= self.dtest:
            task = "WARNING: STP = STPMAX"
        if stp == self.stpmin and (f > ftest or d >= self.dtest):
            task = "WARNING: STP = STPMIN"
        if f <= ftest and abs(d) <= self.gtol * (-self.d0):
            task = "CONVERGENCE"
        self.task = task
        if task != "FG":
            return stp

        if self.stage == 1 and f <= self.f_lo and f > ftest:
            # work on the modified function psi(t) = f(t) - f0 - ftol*d0*t
            gt = self.dtest
            out = _mt_trial(self.lo, self.f_lo - self.lo * gt, self.d_lo - gt,
                            self.hi, self.f_hi - self.hi * gt, self.d_hi - gt,
                            stp, f - stp * gt, d - gt, self.bracketed, self.tmin, self.tmax)
            lo, fl, dl, hi, fh, dh, stp, self.bracketed = out
            self.lo, self.f_lo, self.d_lo = lo, fl + lo * gt, dl + gt
            self.hi, self.f_hi, self.d_hi = hi, fh + hi * gt, dh + gt
        else:
            out = _mt_trial(self.lo, self.f_lo, self.d_lo, self.hi, self.f_hi, self.d_hi,
                            stp, f, d, self.bracketed, self.tmin, self.tmax)
            (self.lo, self.f_lo, self.d_lo, self.hi, self.f_hi, self.d_hi, stp, self.bracketed) = out

        if self.bracketed:
            if abs(self.hi - self.lo) >= 0.66 * self.width1:
                stp = self.lo + 0.5 * (self.hi - self.lo)
            self.width1 = self.width
            self.width = abs(self.hi - self.lo)
            self.tmin = min(self.lo, self.hi)
            self.tmax = max(self.lo, self.hi)
        else:
            self.tmin = stp + 1.1 * (stp - self.lo)
            self.tmax = stp + 4.0 * (stp - self.lo)
        stp = max(stp, self.stpmin)
        stp = min(stp, self.stpmax)
        if (self.bracketed and (stp <= self.tmin or stp >= self.tmax)) or \
           (self.bracketed and self.tmax - self.tmin <= self.xtol * self.tmax):
            stp = self.lo
        return stp


# --------------------------------------------------------------------------
# L-BFGS (unbounded L-BFGS-B) — the arithmetic behind lbfgs.py:64
# --------------------------------------------------------------------------
def two_loop_direction(g, S, Y):
    """d = -H g for history lists S, Y (oldest first).  Identity scaling when empty."""
    q = np.array(g, dtype=np.float64, copy=True)
    k = len(S)
    rho = [1.0 / float(Y[i] @ S[i]) for i in range(k)]
    coef = [0.0] * k
    for i in range(k - 1, -1, -1):
        coef[i] = rho[i] * float(S[i] @ q)
        q -= coef[i] * Y[i]
    if k:
        q *= float(S[-1] @ Y[-1]) / float(Y[-1] @ Y[-1])
    for i in range(k):
        beta = rho[i] * float(Y[i] @ q)
        q += S[i] * (coef[i] - beta)
    return -q


def lbfgs_minimize(fg, x0, maxiter=15000, pgtol=1e-5, m=10, factr=1e7, maxls=20, callback=None):
    """Unbounded L-BFGS-B as SciPy's ``fmin_l_bfgs_b`` runs it (defaults of lbfgs.py:64-70 via
    scipy: m=10, factr=1e7, maxls=20).  Returns dict(x, f, g, nit, nfev, task, iterates)."""
    eps = np.finfo(np.float64).eps
    x = np.array(x0, dtype=np.float64, copy=True)
    f, g = fg(x)
    g = np.asarray(g, dtype=np.float64)
    nfev, nit = 1, 0
    S, Y = [], []
    task = None
    if np.max(np.abs(g)) <= pgtol:
        return dict(x=x, f=f, g=g, nit=0, nfev=nfev, task="CONVERGENCE: NORM_OF_PROJECTED_GRADIENT_<=_PGTOL")
    while True:
        d = two_loop_direction(g, S, Y)
        # ---- line search (lnsrlb) ----
        gd0 = float(g @ d)
        if gd0 >= 0.0:
            if not S:
                task = "ABNORMAL_TERMINATION_IN_LNSRCH"
                break
            S, Y = [], []
            continue
        dnorm = math.sqrt(float(d @ d))
        stp = min(1.0 / dnorm, 1e10) if nit == 0 else 1.0
        x_old, g_old, f_old = x, g, f
        ls = MoreThuente()
        stp = ls.start(stp, f_old, gd0)
        evals, failed = 0, False
        while True:
            if evals >= maxls:          # iback >= maxls
                failed = True
                break
            x = x_old + d if stp == 1.0 else stp * d + x_old
            f, g = fg(x)
            g = np.asarray(g, dtype=np.float64)
            nfev += 1
            evals += 1
            stp = ls.advance(stp, f, float(g @ d))
            if ls.task != "FG":
                break
        if failed or ls.task.startswith("ERROR"):
            x, g, f = x_old, g_old, f_old
            if not S:
                task = "ABNORMAL_TERMINATION_IN_LNSRCH"
                break
            S, Y = [], []
            continue
        # ---- new iterate ----
        nit += 1
        if callback is not None:
            callback(x.copy())
        if nit >= maxiter:
            task = "STOP: TOTAL NO. OF ITERATIONS REACHED LIMIT"
            break
        if np.max(np.abs(g)) <= pgtol:
            task = "CONVERGENCE: NORM_OF_PROJECTED_GRADIENT_<=_PGTOL"
            break
        if (f_old - f) <= eps * factr * max(abs(f_old), abs(f), 1.0):
            task = "CONVERGENCE: REL_REDUCTION_OF_F_<=_FACTR*EPSMCH"
            break
        gd1 = float(g @ d)
        yv = g - g_old
        sv = stp * d
        if stp == 1.0:
            sy, curv_floor = gd1 - gd0, -gd0
        else:
            sy, curv_floor = (gd1 - gd0) * stp, -gd0 * stp
        if sy > eps * curv_floor:
            S.append(sv)
            Y.append(yv)
            if len(S) > m:
                S.pop(0)
                Y.pop(0)
    return dict(x=x, f=f, g=g, nit=nit, nfev=nfev, task=task)


class LBFGSSolver:
    """lbfgs.py:7-73: reg-type normalisation (:9-39), fg closure (:43-54), callback history (:56-61)."""

    def __init__(self, reg_type, alpha1, alpha2, max_iter=500, tol=1e-6, eps=1e-8):
        if reg_type == "lasso":
            kind, a1, a2 = "lasso", alpha1, 0.0
        elif reg_type == "ridge":
            kind, a1, a2 = "ridge", 0.0, alpha2
        elif reg_type == "elasticnet":
            if alpha1 < eps:
                kind, a1, a2 = "ridge", 0.0, alpha2
            elif alpha2 < eps:
                kind, a1, a2 = "lasso", alpha1, 0.0
            else:
                kind, a1, a2 = "elasticnet", alpha1, alpha2
        else:
            raise ValueError(f"Unsupported reg_type='{reg_type}'")
        self.reg_type, self.alpha1, self.alpha2 = kind, a1, a2
        self.max_iter, self.tol = max_iter, tol
        self.history_ = []

    def fit(self, A, b):
        A = np.asarray(A, dtype=np.float64)
        b = np.asarray(b, dtype=np.float64)
        self.metrics = Metrics()
        smooth_l2 = self.reg_type in ("ridge", "elasticnet")
        self.iterates_ = []

        def fg(x):
            t0 = time.perf_counter()
            grad, rr = gram_gradient(A, x, b, self.alpha2 if smooth_l2 else 0.0)
            loss = 0.5 * rr + (0.5 * self.alpha2 * float(x @ x) if smooth_l2 else 0.0)
            self.metrics.grad_times.append(time.perf_counter() - t0)
            return loss, grad

        def cb(xk):
            self.iterates_.append(xk)
            self.history_.append(compute_objective(xk, A, b, self.reg_type, self.alpha1, self.alpha2))

        res = lbfgs_minimize(fg, np.zeros(A.shape[1]), maxiter=self.max_iter, pgtol=self.tol, callback=cb)
        self.x_, self.final_obj_ = res["x"], res["f"]
        self.nit_, self.nfev_, self.task_ = res["nit"], res["nfev"], res["task"]
        return self


# --------------------------------------------------------------------------
# config-1 input generator
# --------------------------------------------------------------------------
def boston_like_data(m=1000, seed=42, noise_std=2.0, rho1=0.8, rho2=0.9):
    """easy_boston_data.py:7-45.  Draw order: block1, block2, distance, noise (PCG64 default_rng)."""
    rng = np.random.default_rng(seed)
    blk1 = rng.multivariate_normal([6, 0.2], 0.25 * np.array([[1.0, rho1], [rho1, 1.0]]), size=m)
    blk2 = rng.multivariate_normal([300, 60], 100 * np.array([[1.0, rho2], [rho2, 1.0]]), size=m)
    dist = rng.normal(4, 1.0, size=(m, 1))
    A = np.hstack([blk1, blk2, dist])
    x_true = np.array([5.0, 0.0, -0.02, -0.05, 1.5])
    b = A @ x_true + rng.normal(0, noise_std, size=m)
    return A, b, x_true


# --------------------------------------------------------------------------
# sharded gradient (the CPU "fake collective" for the multi-GPU logic)
# --------------------------------------------------------------------------
def sharded_gram_gradient(A, y, b, alpha2, parts):
    """Sum over contiguous row shards in fixed rank order; alpha2*y added once after the sum."""
    m = A.shape[0]
    edges = np.linspace(0, m, parts + 1).astype(int)
    g = np.zeros(A.shape[1])
    rr = 0.0
    for p in range(parts):
        Ap, bp = A[edges[p]:edges[p + 1]], b[edges[p]:edges[p + 1]]
        gp, rp = gram_gradient(Ap, y, bp, 0.0)
        g += gp
        rr += rp
    if alpha2 > 0:
        g = g + alpha2 * y
    return g, rr
