"""MI355X drop-in for the reference's ``lbfgs.py``.

The reference delegates the optimiser to ``scipy.optimize.fmin_l_bfgs_b`` (lbfgs.py:64-70, SciPy defaults
m=10, factr=1e7, maxls=20).  Here the unbounded L-BFGS-B iteration is restated natively: ``fg`` is the
single-pass GEMV-pair kernel (K2), the direction is the one-launch two-loop kernel (K4), the line search moves
only scalars over the host (``_linesearch.py``) while its vector updates are device kernels.

fp32 note (SURVEY.md §7 "L-BFGS in fp32"): the iterate x is float64 on the device (like the FISTA state), the
gradient / direction / history vectors are float32, all dots and the loss accumulate in float64.  The
``factr`` test (2.2e-9 relative) is below float32 resolution, so the search can hit its noise floor a few
iterations before SciPy's float64 run stops; a line search that can make no further progress is treated as
convergence, exactly as L-BFGS-B treats its own ABNORMAL_TERMINATION_IN_LNSRCH.
"""
import numpy as np
import torch

from . import _core, _lib
from ._linesearch import LineSearch
from .iterative_solvers import _EventTimer, grad_call_times, reset_metrics
from .operators import vec_axpby

_M, _FACTR, _MAXLS = 10, 1e7, 20
_EPS = float(np.finfo(np.float64).eps)
_EPS32 = float(np.finfo(np.float32).eps)
_FLAT_TRIALS = 3     # consecutive trial points whose objective is indistinguishable from f(x_k) in float32


class LBFGSSolver:
    """L-BFGS for Ridge and smooth Elastic-Net, with tiny-α shortcut.   lbfgs.py:7-73"""

    def __init__(self, reg_type, alpha1, alpha2, max_iter=500, tol=1e-6, eps=1e-8):
        # tiny-α → 0 logic, lbfgs.py:10-35
        if reg_type == "lasso":
            self.reg_type, self.alpha1, self.alpha2 = "lasso", alpha1, 0.0
        elif reg_type == "ridge":
            self.reg_type, self.alpha1, self.alpha2 = "ridge", 0.0, alpha2
        elif reg_type == "elasticnet":
            if alpha1 < eps:
                self.reg_type, self.alpha1, self.alpha2 = "ridge", 0.0, alpha2
            elif alpha2 < eps:
                self.reg_type, self.alpha1, self.alpha2 = "lasso", alpha1, 0.0
            else:
                self.reg_type, self.alpha1, self.alpha2 = "elasticnet", alpha1, alpha2
        else:
            raise ValueError(f"Unsupported reg_type='{reg_type}'")
        self.max_iter = max_iter
        self.tol = tol
        self.history_ = []

    # ------------------------------------------------------------------------------------------------------
    def fit(self, A, b):
        reset_metrics()
        lib = _lib.load()
        prob = _core.as_problem(A, b)
        like = prob.like
        n, dev = prob.n_dev, prob.device          # device length (zero-padded columns stay exactly zero)
        a2 = float(self.alpha2) if self.reg_type in ("ridge", "elasticnet") else 0.0   # lbfgs.py:49-51
        gtimer = _EventTimer(grad_call_times)
        stats = torch.zeros(8, dtype=torch.float64, device=dev)
        rr_dev = stats[5:6]
        self.nfev_ = 0
        self.iterates_ = []
        device_iterates = []                      # fp64 device vectors; copied to the host once, after the run

        def fg(x, d):
            """loss, grad (device), and the scalars g.d, d.d, max|g| in one host read.   lbfgs.py:43-54
            x is the fp64 iterate; the pass over A sees it rounded once to fp32 (kept in fp64 on the two-pass path)."""
            ev = gtimer.start()
            g = torch.empty(n, dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                _lib.check(lib.fos_gemv_pair_f64(prob.h, _core.ptr(x), a2, _core.ptr(g), _core.ptr(rr_dev)),
                           "fos_gemv_pair_f64")
            gtimer.stop(ev)
            with torch.cuda.device(dev):
                _lib.check(lib.fos_vec_stats_f64(_core.ptr(x), _core.ptr(g), _core.ptr(d), n, _core.ptr(stats),
                                                 _core.stream_ptr()), "fos_vec_stats_f64")
            h = stats[:6].cpu().tolist()
            self.nfev_ += 1
            loss = 0.5 * h[5] + 0.5 * a2 * h[0]
            self._x1 = h[4]                       # ||x||_1 of the point just evaluated (for the history objective)
            return loss, g, h[1], h[2], h[3]

        def direction(g, S, Y, hist, head):
            d = torch.empty(n, dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                _lib.check(lib.fos_lbfgs_two_loop(_core.ptr(g), _core.ptr(S), _core.ptr(Y), hist, head, _M, n,
                                                  _core.ptr(d), _core.stream_ptr()), "fos_lbfgs_two_loop")
            return d

        def callback(xk, f_xk):                                                     # lbfgs.py:56-61
            # compute_objective(xk) = smooth loss + alpha1*||xk||_1 (objective_functions.py:13-26).  The line search
            # ends on the point it evaluated last, so fg's loss and ||x||_1 ARE those of xk: no extra pass over A
            # (the reference pays one per iteration here).
            device_iterates.append(xk)
            h = self.alpha1 * self._x1 if self.reg_type in ("lasso", "elasticnet") else 0.0
            self.history_.append(f_xk + h)

        def step_to(x_old, stp, d):
            out = torch.empty(n, dtype=torch.float64, device=dev)
            with torch.cuda.device(dev):
                _lib.check(lib.fos_vec_axpby_f64(1.0, _core.ptr(x_old), float(stp), _core.ptr(d), _core.ptr(out), n,
                                                 _core.stream_ptr()), "fos_vec_axpby_f64")
            return out

        x = torch.zeros(n, dtype=torch.float64, device=dev)                          # lbfgs.py:63 (fp64 iterate)
        S = torch.zeros(_M, n, dtype=torch.float32, device=dev)
        Y = torch.zeros(_M, n, dtype=torch.float32, device=dev)
        hist, head = 0, 0
        f, g, _, _, gmax = fg(x, None)
        nit, task = 0, None
        if gmax <= self.tol:
            task = "CONVERGENCE: NORM_OF_PROJECTED_GRADIENT_<=_PGTOL"
        while task is None:
            d = direction(g, S, Y, hist, head)
            with torch.cuda.device(dev):
                _lib.check(lib.fos_vec_stats(None, _core.ptr(g), _core.ptr(d), n, _core.ptr(stats),
                                             _core.stream_ptr()), "fos_vec_stats")
            _, gd0, dd, _ = stats[:4].cpu().tolist()
            if gd0 >= 0.0:                       # not a descent direction: drop the memory (L-BFGS-B info = -4)
                if hist == 0:
                    task = "ABNORMAL_TERMINATION_IN_LNSRCH"
                    break
                hist, head = 0, 0
                continue
            stp = min(1.0 / np.sqrt(dd), 1e10) if nit == 0 else 1.0
            x_old, g_old, f_old = x, g, f
            ls = LineSearch()
            stp = ls.begin(stp, f_old, gd0)
            evals, failed, gd1, flat = 0, False, gd0, 0
            while True:
                if evals >= _MAXLS:
                    failed = True
                    break
                x = step_to(x_old, stp, d)
                f, g, gd1, _, gmax = fg(x, d)
                evals += 1
                stp_used = stp
                stp = ls.step(stp, f, gd1)
                if ls.status != "FG":
                    break
                # A pass over A in float32 resolves f only to ~eps32*|f|: once several trial points in a row
                # are indistinguishable from f(x_k) the search cannot make progress (SURVEY §7 "L-BFGS in fp32").
                flat = flat + 1 if abs(f - f_old) <= 4.0 * _EPS32 * max(abs(f_old), 1.0) else 0
                if flat >= _FLAT_TRIALS:
                    break
            if flat >= _FLAT_TRIALS and ls.status == "FG":
                if f <= f_old:
                    nit += 1
                    callback(x, f)
                else:
                    x, g, f = x_old, g_old, f_old
                task = "CONVERGENCE: LINE SEARCH REACHED THE FLOAT32 RESOLUTION OF F"
                break
            if failed or ls.status.startswith("ERROR"):
                x, g, f = x_old, g_old, f_old
                if hist == 0:
                    task = "ABNORMAL_TERMINATION_IN_LNSRCH"
                    break
                hist, head = 0, 0
                continue
            stp = stp_used
            nit += 1
            callback(x, f)
            if nit >= self.max_iter:
                task = "STOP: TOTAL NO. OF ITERATIONS REACHED LIMIT"
                break
            if gmax <= self.tol:
                task = "CONVERGENCE: NORM_OF_PROJECTED_GRADIENT_<=_PGTOL"
                break
            if (f_old - f) <= _EPS * _FACTR * max(abs(f_old), abs(f), 1.0):
                task = "CONVERGENCE: REL_REDUCTION_OF_F_<=_FACTR*EPSMCH"
                break
            sy = (gd1 - gd0) * stp
            if sy > _EPS * (-gd0 * stp):
                slot = (head + hist) % _M
                if hist == _M:
                    head = (head + 1) % _M
                else:
                    hist += 1
                vec_axpby(stp, d, 0.0, None, out=S[slot])
                vec_axpby(1.0, g, -1.0, g_old, out=Y[slot])
        gtimer.flush()
        self.iterates_ = [_core.from_device_vec(prob.vec_out(xk), like) for xk in device_iterates]
        self.x_ = _core.from_device_vec(prob.vec_out(x), like)                        # lbfgs.py:71
        self.final_obj_ = f                                                           # lbfgs.py:72
        self.nit_, self.task_ = nit, task
        return self
