"""MI355X drop-in for the reference's ``lbfgs.py``.

The reference delegates the optimiser to ``scipy.optimize.fmin_l_bfgs_b`` (lbfgs.py:64-70, SciPy defaults
m=10, factr=1e7, maxls=20).  Here the unbounded L-BFGS-B iteration is restated natively: ``fg`` is the
single-pass GEMV-pair kernel (K2), the direction is the one-launch two-loop kernel (K4), the line search moves
only scalars over the host (``_linesearch.py``) while its vector updates are device kernels.

Precision: SciPy's optimiser is float64 end to end (lbfgs.py:64), and its stopping test is a 2.2e-9 RELATIVE change
of f - far below what a float32 pass over A resolves.  So everything the optimiser sees is float64 here too: the
iterate, gradient, direction and curvature pairs are fp64 device vectors, and ``fg`` is the fp64-accumulating
instantiation of the single-pass kernel (``fos_gemv_pair_dd``: every a_ij*x_j and a_ij*r_i formed and summed in fp64,
A and b as stored - fp32 / bf16).  The pass stays HBM-bound (one read of A).  What remains of float32 is the storage
rounding of A and b themselves (6e-8 relative per entry): the run is SciPy's algorithm on that slightly perturbed
problem, and reproduces SciPy's iterates to ~1e-7, its ``nit`` / ``nfev`` and its exit.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import _core, _lib
from ._linesearch import LineSearch
from .iterative_solvers import _EventTimer, grad_call_times, reset_metrics

_M, _FACTR, _MAXLS = 10, 1e7, 20
_EPS = float(np.finfo(np.float64).eps)


class _HipOps:
    """The device primitives of one fit: fp64 K2 pass, fp64 two-loop K4, fp64 n-vector kernels.  No CPU fallback."""

    def __init__(self, A, b, comm=None):
        self.lib = _lib.load()
        if comm is not None:                     # row shard: same device length on every rank, sums under the C ABI
            self.prob = A if isinstance(A, _core.Problem) else _core.Problem(A, b, pad=True)
            if getattr(self.prob, "comm", None) is not comm:
                self.prob.set_comm(comm)
        else:
            self.prob = _core.as_problem(A, b)
        self.n, self.dev = self.prob.n_dev, self.prob.device   # device length (zero-padded columns stay exactly zero)
        self._stats = torch.zeros(8, dtype=torch.float64, device=self.dev)
        self.rr = self._stats[5:6]

    def timer(self, sink):
        return _EventTimer(sink)

    def comm_buffer(self):
        return torch.zeros(self.n + 1, dtype=torch.float64, device=self.dev)

    def new_x(self):
        return torch.zeros(self.n, dtype=torch.float64, device=self.dev)

    def new_g(self):
        # n + 1 doubles: fos_gemv_pair_dd leaves ||r||^2 behind the gradient (the pair a sharded run all-reduces)
        return torch.empty(self.n + 1, dtype=torch.float64, device=self.dev)[: self.n]

    def new_history(self, cap):
        return (torch.zeros(cap, self.n, dtype=torch.float64, device=self.dev),
                torch.zeros(cap, self.n, dtype=torch.float64, device=self.dev))

    def grad(self, x, a2, g):
        with self.prob.ctx():
            _lib.check(self.lib.fos_gemv_pair_dd(self.prob.h, _core.ptr(x), float(a2), _core.ptr(g)), "fos_gemv_pair_dd")
        self.rr = g._base[self.n:]

    def stats(self, x, g, d):
        """Host list [x.x, g.d, d.d, max|g|, ||x||_1, ||r||^2 of the last grad()]: one launch, one read."""
        self._stats[5:6].copy_(self.rr)
        with torch.cuda.device(self.dev):
            _lib.check(self.lib.fos_vec_stats_dd(_core.ptr(x), _core.ptr(g), _core.ptr(d), self.n, _core.ptr(self._stats),
                                                 _core.stream_ptr()), "fos_vec_stats_dd")
        return self._stats[:6].cpu().tolist()

    def direction(self, g, S, Y, hist, head):
        d = torch.empty(self.n, dtype=torch.float64, device=self.dev)
        with torch.cuda.device(self.dev):
            _lib.check(self.lib.fos_lbfgs_two_loop_dd(_core.ptr(g), _core.ptr(S), _core.ptr(Y), hist, head, _M, self.n,
                                                      _core.ptr(d), _core.stream_ptr()), "fos_lbfgs_two_loop_dd")
        return d

    def _axpby(self, a, x, b, y, out):
        with torch.cuda.device(self.dev):
            _lib.check(self.lib.fos_vec_axpby_dd(float(a), _core.ptr(x), float(b), _core.ptr(y if b != 0.0 else None),
                                                 _core.ptr(out), self.n, _core.stream_ptr()), "fos_vec_axpby_dd")
        return out

    def step_to(self, x_old, stp, d):
        return self._axpby(1.0, x_old, stp, d, torch.empty(self.n, dtype=torch.float64, device=self.dev))

    def store_pair(self, S, Y, slot, stp, d, g, g_old):
        self._axpby(stp, d, 0.0, None, S[slot])
        self._axpby(1.0, g, -1.0, g_old, Y[slot])

    def to_caller(self, x):
        return _core.from_device_vec(self.prob.vec_out(x), self.prob.like)


class _HipColOps(_HipOps):
    """COLUMN-sharded fit (very wide A): this rank holds A[:, its columns] and its block of every n-vector.  `fg` exchanges
    the m-vector residual under the C ABI (fos_gemv_pair_dd on a column-sharded problem); every dot product of the
    iteration is a sum over the ranks: the five scalars of `stats` cross in one small all-reduce (max|g| in a slot per
    rank), the direction sums the ranks' partial Gram matrices (fos_lbfgs_direction_cols).  All ranks then take identical
    decisions on identical numbers."""

    def __init__(self, prob, comm):
        self.lib = _lib.load()
        self.prob, self.comm = prob, comm
        self.n, self.dev = prob.n_dev, prob.device
        self._stats = torch.zeros(8, dtype=torch.float64, device=self.dev)
        self._xchg = torch.zeros(4 + comm.world, dtype=torch.float64, device=self.dev)
        self._work = torch.zeros(64 * 256, dtype=torch.float64, device=self.dev)
        self._gd = torch.zeros(2, dtype=torch.float64, device=self.dev)
        self.rr = self._stats[5:6]

    def stats(self, x, g, d):
        with torch.cuda.device(self.dev):
            _lib.check(self.lib.fos_vec_stats_dd(_core.ptr(x), _core.ptr(g), _core.ptr(d), self.n, _core.ptr(self._stats),
                                                 _core.stream_ptr()), "fos_vec_stats_dd")
            s, xc = self._stats, self._xchg
            xc.zero_()
            xc[0], xc[1], xc[2], xc[3] = s[0], s[1], s[2], s[4]             # x.x, g.d, d.d, ||x||_1: sums over the blocks
            xc[4 + self.comm.rank] = s[3]                                   # max|g|: one slot per rank
            self.comm.allreduce(xc)
            h = xc.cpu().tolist()
        return [h[0], h[1], h[2], max(h[4:]), h[3], float(self.rr.cpu())]

    def direction(self, g, S, Y, hist, head):
        d = torch.empty(self.n, dtype=torch.float64, device=self.dev)
        with self.prob.ctx():
            _lib.check(self.lib.fos_lbfgs_direction_cols(self.prob.h, _core.ptr(g), _core.ptr(S), _core.ptr(Y), hist, head, _M,
                                                         _core.ptr(d), _core.ptr(self._gd), _core.ptr(self._work),
                                                         self._work.numel()), "fos_lbfgs_direction_cols")
        return d


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
    _TASKS = ("CONVERGENCE: NORM_OF_PROJECTED_GRADIENT_<=_PGTOL", "CONVERGENCE: REL_REDUCTION_OF_F_<=_FACTR*EPSMCH",
              "STOP: TOTAL NO. OF ITERATIONS REACHED LIMIT", "ABNORMAL_TERMINATION_IN_LNSRCH")

    def _fit_native(self, ops):
        """The whole iteration under the C ABI (fos_lbfgs_minimize: what SciPy's compiled optimiser is to lbfgs.py:64):
        line search and memory logic in C++ on the host, every vector on the device, six scalars read per fg."""
        import ctypes as C
        prob, n = ops.prob, ops.n
        a2 = float(self.alpha2) if self.reg_type in ("ridge", "elasticnet") else 0.0   # lbfgs.py:49-51
        max_iter = int(self.max_iter)
        x = ops.new_x()                                                              # lbfgs.py:63
        hist = (C.c_double * max(2 * max_iter, 1))()
        keep = max_iter * n * 8 <= (1 << 30)                  # iterates_ (an extension) only while they fit 1 GiB
        iterates = torch.empty(max(max_iter, 1), n, dtype=torch.float64, device=ops.dev) if keep else None
        cap = 21 * max_iter + 2
        fg_ms = (C.c_float * cap)()
        res = _lib.LbfgsResult()
        with prob.ctx():
            _lib.check(ops.lib.fos_lbfgs_minimize(prob.h, a2, max_iter, float(self.tol), _core.ptr(x), hist,
                                                  _core.ptr(iterates), fg_ms, cap, C.byref(res)), "fos_lbfgs_minimize")
        self.nit_, self.nfev_, self.task_ = int(res.nit), int(res.nfev), self._TASKS[res.task]
        grad_call_times.extend(float(fg_ms[i]) * 1e-3 for i in range(min(self.nfev_, cap)))
        l1 = self.reg_type in ("lasso", "elasticnet")
        self.history_.extend(hist[2 * k] + (self.alpha1 * hist[2 * k + 1] if l1 else 0.0) for k in range(self.nit_))
        # one conversion (and one copy to the host for ndarray callers) for all iterates; the list holds its rows
        self.iterates_ = list(ops.to_caller(iterates[: self.nit_])) if keep and self.nit_ else []
        self.x_ = ops.to_caller(x)                                                    # lbfgs.py:71
        self.final_obj_ = float(res.f)                                                # lbfgs.py:72
        return self

    def fit(self, A, b, *, group=None, comm=None, ops=None, cols=None):
        """``group`` / ``comm``: A, b are THIS RANK's rows of a row-sharded problem; every ``fg`` all-reduces
        [partial gradient ; partial ||r||^2] (n + 1 doubles) once and all ranks take identical decisions on
        identical numbers, so x stays replicated bit for bit.  ``comm`` (a `distributed.Comm`): the all-reduce runs
        under the C ABI on the kernels' stream (inside fos_gemv_pair_dd); ``group`` (a torch.distributed group, any
        backend): it runs here, between the kernels.  ``ops``: the vector/pass primitives (default: the HIP kernels;
        the gloo CPU test injects a stand-in)."""
        reset_metrics()
        if cols is not None:
            # ``cols=(lo, hi, n_total)`` with ``comm=``: COLUMN sharding - A is this rank's columns of all rows, b the whole
            # vector; x_, iterates_ are this rank's block; the host-side driver below runs replicated on global scalars
            from .iterative_solvers import _sharded_problem
            prob, _ = _sharded_problem(A, b, None, comm, None, cols)
            ops = _HipColOps(prob, comm)
        if ops is None and (group is None or dist.get_world_size(group) == 1):
            return self._fit_native(_HipOps(A, b, comm))
        ops = ops if ops is not None else _HipOps(A, b, comm)
        sharded = comm is None and group is not None and dist.get_world_size(group) > 1
        a2 = float(self.alpha2) if self.reg_type in ("ridge", "elasticnet") else 0.0   # lbfgs.py:49-51
        # alpha2*x enters the summed gradient exactly once: rank 0 adds it inside its pass, the others do not
        a2_pass = a2 if (not sharded or dist.get_rank(group) == 0) else 0.0
        gtimer = ops.timer(grad_call_times)
        comm = ops.comm_buffer() if sharded else None
        self.nfev_ = 0
        self.iterates_ = []
        device_iterates = []                      # fp64 device vectors; copied to the host once, after the run

        def fg(x, d):
            """loss, grad (device), and the scalars g.d, d.d, max|g| in one host read.   lbfgs.py:43-54
            x is the fp64 iterate and stays fp64 through the pass over A (fos_gemv_pair_dd)."""
            ev = gtimer.start()
            g = ops.new_g()
            ops.grad(x, a2_pass, g)                               # g = A_p^T (A_p x - b_p) [+ a2 x], ops.rr = ||r_p||^2
            if sharded:
                n_ = g.numel()
                comm[:n_].copy_(g)
                comm[n_:].copy_(ops.rr)
                dist.all_reduce(comm, op=dist.ReduceOp.SUM, group=group)
                g.copy_(comm[:n_])
                ops.rr.copy_(comm[n_:])
            gtimer.stop(ev)
            h = ops.stats(x, g, d)                                # [x.x, g.d, d.d, max|g|, ||x||_1, ||r||^2]
            self.nfev_ += 1
            loss = 0.5 * h[5] + 0.5 * a2 * h[0]
            self._x1 = h[4]                       # ||x||_1 of the point just evaluated (for the history objective)
            return loss, g, h[1], h[2], h[3]

        direction, step_to = ops.direction, ops.step_to

        def callback(xk, f_xk):                                                     # lbfgs.py:56-61
            # compute_objective(xk) = smooth loss + alpha1*||xk||_1 (objective_functions.py:13-26).  The line search
            # ends on the point it evaluated last, so fg's loss and ||x||_1 ARE those of xk: no extra pass over A
            # (the reference pays one per iteration here).
            device_iterates.append(xk)
            h = self.alpha1 * self._x1 if self.reg_type in ("lasso", "elasticnet") else 0.0
            self.history_.append(f_xk + h)

        x = ops.new_x()                                                              # lbfgs.py:63 (fp64 iterate)
        S, Y = ops.new_history(_M)
        hist, head = 0, 0
        f, g, _, _, gmax = fg(x, None)
        nit, task = 0, None
        if gmax <= self.tol:
            task = "CONVERGENCE: NORM_OF_PROJECTED_GRADIENT_<=_PGTOL"
        while task is None:
            d = direction(g, S, Y, hist, head)
            _, gd0, dd = ops.stats(None, g, d)[:3]
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
            evals, failed, gd1 = 0, False, gd0
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
                ops.store_pair(S, Y, slot, stp, d, g, g_old)
        gtimer.flush()
        self.iterates_ = [ops.to_caller(xk) for xk in device_iterates]
        self.x_ = ops.to_caller(x)                                                    # lbfgs.py:71
        self.final_obj_ = f                                                           # lbfgs.py:72
        self.nit_, self.task_ = nit, task
        return self
