"""MI355X drop-in for the reference's ``iterative_solvers.py``: same names, positional order, defaults,
return shapes, history keys, metric keys, exceptions and RNG consumption — the loop bodies are HIP kernels.

Reference lines cited as ``ref:LINE`` are ``iterative_solvers.py:LINE`` of ElBaldo1/FastOptSolver.

Extensions are keyword-only: ``L=`` (skip the power iteration), ``dtype="bf16"`` (store A in bf16,
accumulate in fp32), ``check_every=`` (how often the host polls the device stop flag).
``A`` may be an ndarray, a tensor or a ``prepare(A, b)`` handle; results come back as the kind that went in
(ndarray float64 / tensor float32).
"""
from __future__ import annotations

import math
import time

import numpy as np
import torch

from . import _core, _lib
from .operators import ElasticNetProx, L1Prox, LeastSquares

# ref:11 — Armijo sufficient-decrease constant, read at call time (callers may monkey-patch it)
C: float = 1e-2
_EPS64 = float(np.finfo(np.float64).eps)
_EPS32 = float(np.finfo(np.float32).eps)

# ref:16-18 — module-level metric lists (seconds; device time where a kernel is what was timed)
grad_call_times = []
ls_call_times = []
ls_call_iters = []


def reset_metrics() -> None:
    """ref:20-24"""
    grad_call_times.clear()
    ls_call_times.clear()
    ls_call_iters.clear()


def get_metrics():
    """ref:26-40 — the same seven keys.  With the fused step, "gradient time" is the device time of the
    gradient kernels (host-driven mode) or the per-iteration share of the fused run (device-driven mode)."""
    return {
        'grad_num_calls':   len(grad_call_times),
        'grad_time_total':  sum(grad_call_times),
        'grad_time_mean':   np.mean(grad_call_times) if grad_call_times else 0.0,
        'ls_num_calls':     len(ls_call_times),
        'ls_time_total':    sum(ls_call_times),
        'ls_time_mean':     np.mean(ls_call_times) if ls_call_times else 0.0,
        'ls_iters_total':   sum(ls_call_iters),
    }


class _EventTimer:
    """Pairs of device events on the current stream; resolved to seconds at flush()."""

    def __init__(self, sink):
        self.sink = sink
        self.pending = []

    def start(self):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def stop(self, ev0, count=1):
        ev1 = torch.cuda.Event(enable_timing=True)
        ev1.record()
        self.pending.append((ev0, ev1, count))

    def flush(self):
        if not self.pending:
            return
        self.pending[-1][1].synchronize()
        for ev0, ev1, count in self.pending:
            dt = ev0.elapsed_time(ev1) * 1e-3 / count
            self.sink.extend([dt] * count)
        self.pending.clear()


# ---------------------------------------------------------------------
# Estimate Lipschitz constant L = λ_max(AᵀA)                    ref:45-60
# ---------------------------------------------------------------------
def estimate_lipschitz(A, n_iter: int = 100, tol: float = 1e-6, *, group=None) -> float:
    """Power iteration on the device (w = Aᵀ(Av) is the single-pass GEMV-pair kernel with b = 0).
    Draws ``np.random.randn(n)`` from the global legacy stream exactly like ref:50.

    Row-sharded problems: with a ``Comm`` attached to the problem the kernels' all-reduce makes this the power
    iteration of the whole matrix as it stands; ``group=`` (a torch.distributed group, split-form sharding) sums
    w over the ranks here.  Every rank must draw the same v0 (seed the global stream identically)."""
    prob = _core.prepare(A)
    v0 = np.random.randn(prob.n)
    if group is not None:
        from .distributed import sharded_lipschitz
        bare = prob if prob.b is None else _core.Problem(prob.A, None, prob.dtype, pad=False)   # same A, b = 0 (ref:54)
        return sharded_lipschitz(lambda v: bare.gemv_pair(v, 0.0), prob.n, prob.vec_in(v0)[: prob.n], n_iter, tol,
                                 group=group)
    L, _, _ = prob.power_iter(v0, n_iter=n_iter, tol=tol)
    return L


class _GroupReducer:
    """Split-form sharding over a torch.distributed group (any backend): the sums over row blocks that the device
    would do itself with a Comm attached are done here, between the kernels, in the order every rank follows."""

    def __init__(self, prob, group):
        import torch.distributed as dist
        self.dist, self.group, self.prob = dist, group, prob
        self.on_device = dist.get_backend(group) == "nccl"
        self.gbuf64 = None          # precise mode (backtracking): the fp64 [gradient ; ||r||^2] of the state machine

    def grad(self):
        """[partial gradient ; partial ||r||^2] -> global, in gbuf (n + 1 floats: the one exchange per iteration; n + 1
        doubles in precise mode)."""
        buf = self.gbuf64 if self.gbuf64 is not None else self.prob.gbuf
        self.dist.all_reduce(buf[: self.prob.n_dev + 1], op=self.dist.ReduceOp.SUM, group=self.group)

    def rr_global(self):
        buf = self.gbuf64 if self.gbuf64 is not None else self.prob.gbuf
        return float(buf[self.prob.n_dev])

    def sum(self, vals):
        t = torch.tensor(list(vals), dtype=torch.float64, device=self.prob.device if self.on_device else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t.tolist()


# ---------------------------------------------------------------------
# Armijo acceptance (ref:191, :306, :101) in cancellation-free form
# ---------------------------------------------------------------------
_BATCH = 16      # candidate steps decided per pass over A (MFMA N dimension)
_HISTORY_CHUNK_BYTES = 256 << 20   # device-resident x history is read back in chunks of at most this size


def _armijo_accepts(tr, t_k, smooth_a2, grad_eps=8.0 * _EPS32):
    """g(x_tmp) <= g(y) + C*grad.dlt  <=>  (1-C)*grad.dlt + 0.5||A dlt||^2 + 0.5*a2*||dlt||^2 <= 0  (g quadratic).

    Resolution of the test: (a) the reference compares two float64 evaluations of g, so differences below
    eps64*g(y) read as "equal" there (and x_tmp == y ends its loop); (b) grad.dlt is only known to
    ~grad_eps*||grad||*||dlt||, where grad_eps is the relative resolution of the gradient pass; (c) a trial step
    shorter than t*grad_eps*||grad|| lies inside the noise ball of y_k: its direction is rounding noise, the
    decision meaningless and the step harmless - the counterpart of the reference's exact "x_tmp == y" exit.
    grad_eps: 8*eps32 for the fp32 pass; 64*eps64 in precise mode (fos_fista_set_precise: fp64-accumulating pass at the
    unrounded y_k, what fista(backtracking=True) runs), where (b) and (c) shrink to the reference's own rounding level."""
    excess = (1.0 - C) * tr["gd"] + 0.5 * tr["q"] + 0.5 * smooth_a2 * tr["dd"]
    g_y = 0.5 * tr["rr_y"] + 0.5 * smooth_a2 * tr["y2"]
    noise = max(_EPS64 * g_y, grad_eps * math.sqrt(tr["gnorm2"] * tr["dd"]))
    in_noise_ball = tr["dd"] <= (grad_eps * t_k) ** 2 * tr["gnorm2"]
    return tr["nnz"] == 0 or excess <= noise or in_noise_ball


# ---------------------------------------------------------------------
# shared FISTA / FISTA-Δ / fused-ISTA driver
# ---------------------------------------------------------------------
class _Run:
    """One solver run: the state machine, what the caller asked for, and the metric / history sinks.  Each execution
    strategy is a method that returns True when it ran the whole job and False when this problem / plan has no such form
    (the dispatcher `_drive` then tries the next one)."""

    def __init__(self, prob, like, st, *, tau, eta, max_iter, tol, tol_ratio, backtracking, grad_tol_check, history,
                 history_obj, log, check_every, reducer, smooth_a2, grad_eps, batch_trials):
        self.prob, self.like, self.st = prob, like, st
        self.tau, self.eta, self.max_iter, self.tol, self.tol_ratio = tau, eta, max_iter, tol, tol_ratio
        self.backtracking, self.grad_tol_check = backtracking, grad_tol_check
        self.history, self.history_obj, self.log = history, history_obj, log
        self.check_every, self.reducer = check_every, reducer
        self.smooth_a2, self.grad_eps, self.use_batch = smooth_a2, grad_eps, batch_trials
        self.gtimer = getattr(st, "make_timer", _EventTimer)(grad_call_times)     # stand-in states bring a host timer
        self.recording = history is not None or log is not None
        # ista passes x0 itself as `like`
        self.as_tensor = like.tensor if hasattr(like, "tensor") else _core.is_tensor(like)

    # ---- 1. nothing needs the host per iteration: enqueue everything, poll for stops -------------------------------
    def enqueue_only(self):
        st, gtimer = self.st, self.gtimer
        stops_possible = self.tol > 0.0 or self.tol_ratio > 0.0
        chunk = self.max_iter if not stops_possible else max(1, int(self.check_every or 8))
        done = 0
        while done < self.max_iter:
            todo = min(chunk, self.max_iter - done)
            ev = gtimer.start()
            st.run(todo)
            gtimer.stop(ev, todo)
            done += todo
            if stops_possible and st.status().stopped != _lib.STOP_NONE:
                break
        gtimer.flush()
        if stops_possible:
            # only the iterations that really ran count as gradient calls, plus the gradient whose norm ended the run
            s_end = st.status()
            del grad_call_times[int(s_end.k) + (1 if s_end.stopped == _lib.STOP_GRAD else 0):]
        return True

    # ---- 2. small problems (A fits one CU's LDS): every host-driven feature - backtracking, gradient-norm stop, history,
    #         ISTA log - runs inside ONE launch of the LDS-resident loop; the host only unpacks what the device recorded
    def resident(self):
        st, gtimer, like = self.st, self.gtimer, self.like
        ev = gtimer.start()
        ls_t0 = time.perf_counter()
        res = st.run_resident(self.max_iter, backtracking=self.backtracking, eta=self.eta, armijo_c=C,
                              grad_tol=self.tol if (self.grad_tol_check and self.tol > 0.0) else 0.0,
                              record=self.recording)
        if res is None:
            gtimer.pending.clear()
            return False
        k = res["done"]
        # one gradient per completed iteration, plus the one whose norm ended the run (ref:173-180)
        ngrad = k + (1 if st.status().stopped == _lib.STOP_GRAD else 0)
        gtimer.stop(ev, max(ngrad, 1))
        gtimer.flush()
        del grad_call_times[ngrad:]
        if self.backtracking:                                    # ref:183-197: one search per completed iteration
            share = (time.perf_counter() - ls_t0) / max(k, 1)
            ls_call_iters.extend(int(v) for v in res["ls"])
            ls_call_times.extend([share] * k)                    # the device does not time its phases: equal shares
        if self.recording:
            hs = res["hist"].cpu().numpy()
            xs = res["x"]
            rows = [_core.from_device_vec(xs[i], like) for i in range(k)] if self.as_tensor else list(xs.cpu().numpy())
            if self.history is not None:
                self.history["x"].extend(rows)
                self.history["obj"].extend(self.history_obj(float(r[0]), float(r[2]), float(r[1])) for r in hs)
            if self.log is not None:
                self.log["x"].extend(rows)
                self.log["t"].extend(res["taus"])
                self.log["delta"].extend(float(math.sqrt(r[3])) for r in hs)
        return True

    # ---- 3. history of a PLAIN run without any per-iteration host round trip: x and the objective ingredients are
    #         recorded on the device by the same two kernels of the plain run and read back once (ref:224-232, :319-322)
    def history_plain(self):
        st, gtimer, prob, like, history = self.st, self.gtimer, self.prob, self.like, self.history
        chunk = max(1, min(self.max_iter, _HISTORY_CHUNK_BYTES // (8 * prob.n_dev)))    # bound the device-side x history
        done = 0
        while done < self.max_iter:
            todo = min(chunk, self.max_iter - done)
            ev = gtimer.start()
            rec = st.run_history(todo)
            if rec is None:
                gtimer.pending.clear()
                return False
            gtimer.stop(ev, todo)
            xh, hs = rec
            hs = hs.cpu().numpy()
            if like.tensor:
                history["x"].extend(_core.from_device_vec(xh[i], like) for i in range(todo))
            else:
                history["x"].extend(list(xh.cpu().numpy()))
            history["obj"].extend(self.history_obj(float(r[0]), float(r[2]), float(r[1])) for r in hs)
            done += todo
        gtimer.flush()
        return True

    # ---- the Armijo search on the host (ref:183-197 / :298-312 / :92-108): used by the host-driven loop and to finish a
    #      search the device parked (all 16 candidates of a batch rejected: the reference's step-underflow regime)
    def search_on_host(self, t_k, bt_steps):
        st, reducer = self.st, self.reducer
        while True:
            # candidates t_k, t_k*eta, ... decided by ONE pass over A on the matrix cores (fos.h); ragged
            # problems (two-pass fallback) evaluate one candidate per pass.
            rows = st.trial_batch(t_k, self.eta, _BATCH) if self.use_batch else None
            if rows is None:
                self.use_batch = False
                rows = [st.trial(t_k, with_residual=True)]
            if reducer is not None:                        # ||A dlt||^2 = sum over the row blocks; ||r||^2 likewise
                for tr, q in zip(rows, reducer.sum([tr["q"] for tr in rows])):
                    tr["q"] = q
                    tr["rr_y"] = reducer.rr_global()
            for tr in rows:
                if _armijo_accepts(tr, t_k, self.smooth_a2, self.grad_eps):
                    return t_k, bt_steps
                t_k *= self.eta                            # ref:195
                bt_steps += 1

    # ---- 4. data-dependent control on the device - no host round trip per iteration:
    #   backtracking (fos_fista_run_backtracking): gradient, one matrix-core batch of 16 candidates, a decision kernel,
    #     the update with the accepted step, the bookkeeping - all enqueued;
    #   with history / ista's log (fos_fista_run_recorded; also adaptive restart and the stopping rules without
    #     backtracking): iterates and their norms recorded per iteration, ||A x - b||^2 of every iterate out of the NEXT
    #     iteration's gradient pass, the last objective closed by one residual pass.
    # The host polls every `check_every` iterations for stops and for a parked search, which it finishes itself before
    # handing the loop back to the device.
    def device_driven(self):
        st, gtimer, prob, like = self.st, self.gtimer, self.prob, self.like
        history, log, backtracking, recording = self.history, self.log, self.backtracking, self.recording
        cap = _HISTORY_CHUNK_BYTES // (8 * prob.n_dev) if recording else self.max_iter
        chunk = max(1, min(int(self.check_every or (16 if recording else 8)), cap))
        done, started_total, ls_t0 = 0, 0, time.perf_counter()
        rr_seen, norms = [], []                # rr_seen[t]: residual of the iterate iteration t started from
        while done < self.max_iter:
            todo = min(chunk, self.max_iter - done)
            ev = gtimer.start()
            if recording:
                rec = st.run_recorded(todo, backtracking, self.eta, C, self.grad_eps, want_rr=history is not None)
            else:
                pair = st.run_backtracking(todo, self.eta, C, self.grad_eps)
                rec = None if pair is None else dict(ls=pair[0], taus=pair[1])
            if rec is None:                                   # this plan has no candidate pass: the host drives
                gtimer.pending.clear()
                return False
            s = st.status()                                   # synchronises: k, stop / stall flag
            ran = int(s.k) - done
            stalled = s.stopped == _lib.STOP_LS_STALL
            started = ran + (1 if (stalled or s.stopped == _lib.STOP_GRAD) else 0)
            gtimer.stop(ev, max(started, 1))
            started_total += started
            if backtracking:
                ls_call_iters.extend(int(v) for v in rec["ls"][:ran].cpu().tolist())
            if recording:
                if history is not None:
                    rr_seen.extend(rec["rr_seen"][:started].cpu().tolist())
                hs = rec["hist"][:ran].cpu().numpy()
                xh = prob.vec_out(rec["x"][:ran])
                rows = [_core.from_device_vec(xh[i], like) for i in range(ran)] if self.as_tensor else list(xh.cpu().numpy())
                if history is not None:
                    history["x"].extend(rows)
                if log is not None:                            # ista's log (ref:117-120): x, the step used, ||dx||
                    log["x"].extend(rows)
                    log["t"].extend(rec["taus"][:ran].cpu().tolist() if backtracking else [self.tau] * ran)
                    log["delta"].extend(float(math.sqrt(r[3])) for r in hs)
                norms.extend((float(r[1]), float(r[2])) for r in hs)
            done += ran
            if stalled:                                       # finish this iteration's search on the host
                tau = st.resume_after_stall()
                tau, steps = self.search_on_host(tau, _BATCH)
                self.tau = tau
                ls_call_iters.append(steps)
                st.set_tau(tau)
                st.update()
                s = st.status()
                if recording:
                    row = _core.from_device_vec(st.x_tensor(), like)
                    if history is not None:
                        history["x"].append(row)
                    if log is not None:
                        log["x"].append(row)
                        log["t"].append(tau)
                        log["delta"].append(s.this_step)
                    norms.append((s.xnorm1, s.xnorm2))
                done += 1
            if s.stopped != _lib.STOP_NONE:
                break
        gtimer.flush()
        del grad_call_times[started_total:]
        if backtracking:
            share = (time.perf_counter() - ls_t0) / max(len(ls_call_iters), 1)
            ls_call_times.extend([share] * len(ls_call_iters))    # the device does not time its phases: equal shares
        # f(x after iteration t) needs ||A x - b||^2 of that iterate: seen by iteration t + 1, or by a closing pass
        if history is not None:
            rr_of = rr_seen[1:done + 1]
            if len(rr_of) < done:
                rr_of.append(prob.residual_objective(st.x_tensor())[0])
            history["obj"].extend(self.history_obj(rr, x2, x1) for rr, (x1, x2) in zip(rr_of, norms))
        return True

    # ---- 5. the host drives every iteration: split-form sharding over torch.distributed (the all-reduce sits between
    #         the gradient and the update), generic callables, plans without the candidate pass.
    # History objective f(x_k) without the reference's extra pass per iteration (ref:225-230, :321): the DUAL gradient
    # pass of iteration k also returns ||A x_k - b||^2, so f(x_k) is appended one iteration late and only the very last
    # iterate needs a residual pass of its own.
    def host_driven(self):
        st, gtimer, prob, like, reducer = self.st, self.gtimer, self.prob, self.like, self.reducer
        history, log, tol, tol_ratio = self.history, self.log, self.tol, self.tol_ratio

        def rr_x_of(status):         # ||A x_k - b||^2 over ALL rows (split-form sharding: summed here)
            return reducer.sum([status.rr_x])[0] if reducer is not None else status.rr_x

        owed = None                  # (||x||_1, ||x||_2^2) of the newest iterate whose objective is not recorded yet
        for _ in range(self.max_iter):
            ev = gtimer.start()
            st.grad(dual=owed is not None)                        # ref:173-175 (alpha2*y is added by the consumers)
            if reducer is not None:
                reducer.grad()
            gtimer.stop(ev)
            if self.grad_tol_check and tol > 0.0:                 # ref:179
                if math.sqrt(st.trial(self.tau, with_residual=False)["gnorm2"]) < tol:
                    if owed is not None:
                        history["obj"].append(self.history_obj(rr_x_of(st.status()), owed[1], owed[0]))
                        owed = None
                    break
            if self.backtracking:                                 # ref:183-197 / ref:298-312 / ref:92-108
                ls_t0 = time.perf_counter()
                t_k, bt_steps = self.search_on_host(self.tau, 0)
                ls_call_times.append(time.perf_counter() - ls_t0)
                ls_call_iters.append(bt_steps)
                self.tau = t_k
                st.set_tau(self.tau)
            st.update()                                           # ref:200-221
            if self.recording:
                xk = st.x_tensor()
                s = st.status()
                if history is not None:
                    if owed is not None:
                        history["obj"].append(self.history_obj(rr_x_of(s), owed[1], owed[0]))
                    history["x"].append(_core.from_device_vec(xk, like))
                    owed = (s.xnorm1, s.xnorm2)
                if log is not None:
                    log["x"].append(_core.from_device_vec(xk, like))
                    log["t"].append(self.tau)
                    log["delta"].append(s.this_step)
            else:
                s = st.status() if (tol > 0.0 or tol_ratio > 0.0) else None
            if s is not None and s.stopped != _lib.STOP_NONE:     # ref:238, :242
                break
        if owed is not None:
            rr, x2, x1 = prob.residual_objective(st.x_tensor())
            if reducer is not None:
                rr = reducer.sum([rr])[0]
            history["obj"].append(self.history_obj(rr, x2, x1))
        gtimer.flush()
        return True


def _drive(prob, like, *, mode, prox_kind, alpha1, alpha2, tau, delta=0.0, backtracking=False, eta=0.5,
           max_iter=500, tol=0.0, tol_ratio=0.0, adaptive_restart=False, restart_threshold=1.0,
           grad_tol_check=False, history=None, history_obj=None, x0=None, check_every=None, log=None,
           batch_trials=True, reducer=None, state=None):
    """Run the state machine with the first execution strategy of `_Run` that serves this configuration: device-driven
    wherever nothing needs a per-iteration host decision, host-driven otherwise (split-form sharding, ref:179 / :183-197 /
    :224-232 on plans without the device forms)."""
    st = state if state is not None else _core.Fista(prob)      # `state`: a stand-in with the same interface (CPU tests)
    x0_dev = None if x0 is None else _core.to_device_vec(x0, prob.device).double()   # padded by Fista.reset
    # reducer: split-form sharding - the all-reduce sits between the gradient and the update, so the host drives
    host_needed = backtracking or history is not None or log is not None or reducer is not None
    # gradient-norm stop (ref:179): on the device (fos_fista_params.tol_grad) when the run is enqueue-only, by the host
    # between grad() and update() when the host drives anyway
    device_loop = reducer is None                               # every such configuration has an enqueue-only form
    dev_grad_stop = grad_tol_check and tol > 0.0 and (not host_needed or device_loop)
    st.reset(tau, alpha1, alpha2, mode=mode, prox_kind=prox_kind, delta=delta, adaptive_restart=adaptive_restart,
             restart_threshold=restart_threshold, tol_step=tol if tol > 0.0 else 0.0,
             tol_ratio=tol_ratio if tol_ratio > 0.0 else 0.0, x0=x0_dev, **({"tol_grad": tol} if dev_grad_stop else {}))
    # Backtracking decides on a cancelling sum (grad.dlt): take the gradient from the fp64-accumulating pass then; in
    # split-form sharding the state machine keeps it in a tensor of ours and the reducer sums the n + 1 doubles
    grad_eps = 8.0 * _EPS32
    if backtracking and hasattr(st, "set_precise"):
        if reducer is None:
            st.set_precise(True)
            grad_eps = 64.0 * _EPS64
        elif not prob.plan()["resident"]:       # (LDS-resident plans have no split-form fp64 pass: they keep the fp32 gbuf)
            st.set_precise(True, own_buffer=True)
            reducer.gbuf64 = st.gbuf64
            grad_eps = 64.0 * _EPS64
    run = _Run(prob, like, st, tau=tau, eta=eta, max_iter=max_iter, tol=tol, tol_ratio=tol_ratio, backtracking=backtracking,
               grad_tol_check=grad_tol_check, history=history, history_obj=history_obj, log=log, check_every=check_every,
               reducer=reducer, smooth_a2=alpha2 if (prox_kind == _lib.PROX_L1 and alpha2 > 0) else 0.0,
               grad_eps=grad_eps, batch_trials=batch_trials)
    if not host_needed:
        run.enqueue_only()
        return st
    on_device = reducer is None and max_iter > 0
    if on_device and run.resident():
        return st
    plain = not (mode == _lib.MODE_FISTA and adaptive_restart) and tol == 0.0 and tol_ratio == 0.0
    if on_device and history is not None and log is None and not backtracking and plain and run.history_plain():
        return st
    if on_device and hasattr(st, "run_recorded") and (run.recording or backtracking) and run.device_driven():
        return st
    run.host_driven()
    return st


def _objective_by_alpha(alpha1, alpha2):
    """History objective of fista(): driven by alpha>0, not by reg_type.  ref:225-230"""
    def obj(rr, x2, x1):
        val = 0.5 * rr
        if alpha2 > 0:
            val += 0.5 * alpha2 * x2
        if alpha1 > 0:
            val += alpha1 * x1
        return val
    return obj


def _objective_by_reg(reg_type, alpha1, alpha2):
    """compute_objective semantics (objective_functions.py:3-30), used by fista_delta ref:321."""
    if reg_type not in ("lasso", "ridge", "elasticnet"):
        raise ValueError(f"Unsupported reg_type='{reg_type}'")

    def obj(rr, x2, x1):
        val = 0.5 * rr
        if reg_type in ("ridge", "elasticnet"):
            val += 0.5 * alpha2 * x2
        if reg_type in ("lasso", "elasticnet"):
            val += alpha1 * x1
        return val
    return obj


# ---------------------------------------------------------------------
# ISTA                                                          ref:65-125
# ---------------------------------------------------------------------
def ista(x0, g, grad_g, prox_h, L, backtracking: bool = False, eta: float = 0.5, t_init_factor: float = 1.0,
         max_iter: int = 500, tol: float = 0.0, return_history: bool = False):
    """Proximal gradient over callables.  When ``g``/``grad_g`` come from one ``LeastSquares`` object and
    ``prox_h`` is an ``L1Prox``/``ElasticNetProx`` the fused device state machine runs (single pass over A per
    gradient); any other callables are invoked as given on float32 device tensors, with the loop's own vector
    arithmetic still on the device."""
    reset_metrics()
    ls = getattr(g, "__self__", g)
    fused = (isinstance(ls, LeastSquares) and getattr(grad_g, "__self__", None) is ls
             and isinstance(prox_h, (L1Prox, ElasticNetProx)))
    t = t_init_factor / L                                                     # ref:81
    if fused:
        prob = ls.prob
        if isinstance(prox_h, ElasticNetProx):
            if ls.alpha2 != 0.0:
                fused = False          # l2 both in g and in the prox: not a state-machine configuration
            else:
                kind, a1, a2 = _lib.PROX_ENET, prox_h.alpha1, prox_h.alpha2
        else:
            kind, a1, a2 = _lib.PROX_L1, prox_h.alpha1, ls.alpha2
    if fused:
        x0_dev = _core.to_device_vec(x0, prob.device).double()
        log = {"x": [_core.from_device_vec(x0_dev, x0)], "t": [t], "delta": []} if return_history else None
        st = _drive(prob, x0, mode=_lib.MODE_ISTA, prox_kind=kind, alpha1=a1, alpha2=a2, tau=t,
                    backtracking=backtracking, eta=eta, max_iter=max_iter, tol=tol, x0=x0_dev, log=log)
        x = _core.from_device_vec(st.x_tensor(), x0)
        return (x, log) if return_history else x

    # ---- generic callables ----
    # The loop's own vector arithmetic runs on the device.  The callables are the caller's code: they get float32
    # device tensors; a callable written for ndarrays (what a user of the reference has: closures over a NumPy A)
    # is detected on its first call and from then on fed float64 ndarrays, its result moved back to the device.
    from .operators import vec_axpby, vec_stats
    x = _core.to_device_vec(x0).clone()                                        # ref:79
    wants_numpy = {}

    def call(fn, *args):
        key = id(fn)
        if not wants_numpy.get(key, False):
            try:
                return fn(*args)
            except (TypeError, RuntimeError, ValueError, AttributeError):
                if _core.is_tensor(x0):
                    raise
                wants_numpy[key] = True
        host = [a.detach().to("cpu", torch.float64).numpy() if _core.is_tensor(a) else a for a in args]
        out = fn(*host)
        return _core.to_device_vec(out) if isinstance(out, np.ndarray) and out.ndim == 1 else out

    log = {"x": [_core.from_device_vec(x, x0)], "t": [t], "delta": []} if return_history else None
    gtimer = _EventTimer(grad_call_times)
    for _ in range(max_iter):
        ev = gtimer.start()
        grad = _core.to_device_vec(call(grad_g, x))                             # ref:87-89
        gtimer.stop(ev)
        if backtracking:                                                        # ref:92-108
            bt_steps = 0
            ls_t0 = time.perf_counter()
            t_k = t
            gx = float(call(g, x))
            while True:
                x_new = _core.to_device_vec(call(prox_h, vec_axpby(1.0, x, -t_k, grad), t_k))
                diff = vec_axpby(1.0, x_new, -1.0, x)
                gd = vec_stats(None, grad, diff)[1]
                if float(call(g, x_new)) <= gx + C * gd:
                    break
                t_k *= eta
                bt_steps += 1
            ls_call_times.append(time.perf_counter() - ls_t0)
            ls_call_iters.append(bt_steps)
            t = t_k
        else:
            x_new = _core.to_device_vec(call(prox_h, vec_axpby(1.0, x, -t, grad), t))   # ref:110-111
            diff = vec_axpby(1.0, x_new, -1.0, x)
        delta = math.sqrt(vec_stats(None, None, diff)[2])                        # ref:114
        x = x_new
        if return_history:
            log["x"].append(_core.from_device_vec(x, x0))
            log["t"].append(t)
            log["delta"].append(delta)
        if tol > 0.0 and delta < tol:                                           # ref:122
            break
    gtimer.flush()
    out = _core.from_device_vec(x, x0)
    return (out, log) if return_history else out


def _lipschitz_cols(prob, comm, cols, n_iter=100, tol=1e-6):
    """estimate_lipschitz (ref:45-60) for a column-sharded matrix: v is partitioned like x; w_p = A_p^T (sum_q A_q v_q) is
    the two-phase pass with its one m-vector exchange, ||w|| sums the blocks' squares over the ranks.  Draws the n_total
    normals of ref:50 from the global stream on every rank (same seed everywhere) and keeps this rank's slice."""
    lo, hi, n_total = cols
    v = torch.from_numpy(np.random.randn(n_total)[lo:hi].copy()).to(prob.device, torch.float32)
    bare = prob if prob.b is None else None
    if bare is None:
        bare = _core.Problem(prob.A, None, prob.dtype, pad=False)               # same A, b = 0 (ref:54)
        bare.set_comm_cols(comm)
    nrm = lambda t: math.sqrt(float(comm.allreduce((t.double() @ t.double()).reshape(1))[0]))   # noqa: E731
    v = v / nrm(v)
    prev, L = 0.0, None
    for _ in range(n_iter):
        w = bare.gemv_pair(v, 0.0)
        L = nrm(w)
        v = w / L
        if abs(L - prev) < tol:
            break
        prev = L
    return L


def _sharded_problem(A, b, dtype, comm, group, cols=None):
    """(problem, reducer) for the solver front-ends: plain, Comm-attached (reductions under the C ABI, reducer None)
    or split-form over a torch.distributed group (reducer does the sums).  cols = (lo, hi, n_total): COLUMN sharding -
    A holds this rank's columns [lo, hi) of an m x n_total matrix, b is whole, the returned x is this rank's block."""
    if comm is None and group is None:
        return _core.as_problem(A, b, dtype), None
    if cols is not None:
        if comm is None:
            raise ValueError("column sharding needs a distributed.Comm (comm=)")
        prob = A if isinstance(A, _core.Problem) else _core.Problem(A, b, dtype, pad=False)
        if not getattr(prob, "col_sharded", False):
            prob.set_comm_cols(comm)
        return prob, None
    import torch.distributed as dist
    prob = A if isinstance(A, _core.Problem) else _core.Problem(A, b, dtype, pad=True)   # same n_dev on every rank
    if comm is not None:
        if getattr(prob, "comm", None) is not comm:
            prob.set_comm(comm)
        return prob, None
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return prob, None
    return prob, _GroupReducer(prob, group)


# ---------------------------------------------------------------------
# FISTA                                                        ref:132-245
# ---------------------------------------------------------------------
def fista(A, b, reg_type: str, alpha1: float, alpha2: float, backtracking: bool = False, eta: float = 0.5,
          t_init_factor: float = 1.0, max_iter: int = 500, tol: float = 0.0, tol_ratio: float = 0.0,
          adaptive_restart: bool = False, restart_threshold: float = 1.0, return_history: bool = False,
          *, L=None, dtype=None, check_every=None, comm=None, group=None, cols=None):
    """``comm=`` / ``group=``: A, b are THIS RANK's rows of a row-sharded problem (one process per GPU); every flag
    of the reference's loop works sharded - backtracking, history, restart, the stopping rules.  ``comm`` (a
    `distributed.Comm`) puts the one all-reduce per iteration on the kernels' stream under the C ABI; ``group`` (a
    torch.distributed group, any backend) does it between the kernels from Python.
    ``cols=(lo, hi, n_total)`` with ``comm=``: COLUMN sharding for very wide A - A is this rank's columns [lo, hi) of
    all rows, b the whole vector; x is partitioned (the result and the history are this rank's block), one all-reduce of
    an m-vector per iteration (backtracking: plus ONE all-reduce of the 16 candidates' m-vectors per search)."""
    reset_metrics()
    prob, reducer = _sharded_problem(A, b, dtype, comm, group, cols)
    like = prob.like
    if L is not None:
        L_val = float(L)
    elif cols is not None:
        L_val = _lipschitz_cols(prob, comm, cols)
    else:
        L_val = estimate_lipschitz(prob, group=group if reducer is not None else None)                        # ref:155
    if alpha2 > 0:                                                            # ref:156-157
        L_val += alpha2
    tau = t_init_factor / L_val                                               # ref:158
    history = None
    if return_history:
        zero = torch.zeros(prob.n, dtype=torch.float64, device=prob.device)
        history = {"x": [_core.from_device_vec(zero, like)], "obj": []}      # ref:160
    st = _drive(prob, like, mode=_lib.MODE_FISTA, prox_kind=_lib.PROX_L1, alpha1=alpha1, alpha2=alpha2, tau=tau,
                backtracking=backtracking, eta=eta, max_iter=max_iter, tol=tol, tol_ratio=tol_ratio,
                adaptive_restart=adaptive_restart, restart_threshold=restart_threshold, grad_tol_check=True,
                history=history, history_obj=_objective_by_alpha(alpha1, alpha2), check_every=check_every,
                reducer=reducer)
    x_k = _core.from_device_vec(st.x_tensor(), like)
    return (x_k, history) if return_history else x_k


# ---------------------------------------------------------------------
# FISTA-Δ                                                      ref:251-344
# ---------------------------------------------------------------------
def fista_delta(A, b, reg_type: str, alpha1: float, alpha2: float, delta: float, backtracking: bool = False,
                eta: float = 0.5, t_init_factor: float = 1.0, max_iter: int = 500, tol: float = 0.0,
                tol_ratio: float = 0.0, return_history: bool = False, *, L=None, dtype=None, check_every=None,
                comm=None, group=None, cols=None):
    reset_metrics()
    # Course requirement: delta > 2 for convergence guarantee                   ref:268
    assert delta > 2, "In FISTA-Δ, delta must be > 2 for convergence (course requirement)"
    prob, reducer = _sharded_problem(A, b, dtype, comm, group, cols)
    like = prob.like
    if L is not None:
        L_val = float(L)
    elif cols is not None:
        L_val = _lipschitz_cols(prob, comm, cols)
    else:
        L_val = estimate_lipschitz(prob, group=group if reducer is not None else None)                        # ref:273
    if alpha2 > 0:
        L_val += alpha2
    tau = t_init_factor / L_val
    history = {"x": [], "obj": []} if return_history else None               # ref:279 (no x0 entry)
    obj = _objective_by_reg(reg_type, alpha1, alpha2) if return_history else None
    st = _drive(prob, like, mode=_lib.MODE_DELTA, prox_kind=_lib.PROX_L1, alpha1=alpha1, alpha2=alpha2, tau=tau,
                delta=delta, backtracking=backtracking, eta=eta, max_iter=max_iter, tol=tol, tol_ratio=tol_ratio,
                grad_tol_check=False, history=history, history_obj=obj, check_every=check_every, reducer=reducer)
    x_k = _core.from_device_vec(st.x_tensor(), like)
    return (x_k, history) if return_history else x_k


# ---------------------------------------------------------------------
# Regularisation path (extension)
# ---------------------------------------------------------------------
def fista_path(A, b, alphas, t_init_factor: float = 1.0, max_iter: int = 500, *, delta=None, L=None, dtype=None,
               comm=None, cols=None, tol: float = 0.0, tol_ratio: float = 0.0, adaptive_restart: bool = False,
               restart_threshold: float = 1.0, return_info: bool = False):
    """Solve the same (A, b) for several regularisation weights at once.

    ``alphas`` is a sequence of ``(alpha1, alpha2)`` pairs.  The result is the list of solutions that
    ``fista(A, b, ..., alpha1, alpha2, t_init_factor=t_init_factor, max_iter=max_iter, L=L)`` (or ``fista_delta``
    when ``delta`` is given) would return one by one - same iterates - but the weights advance in lockstep: up to four
    share one read of A per iteration in the multi-vector form of the single-pass kernel (fp32, n <= 8192); five to
    sixteen run on the matrix cores as two GEMM-shaped products per iteration for all of them (fp32 and bf16 storage,
    any streaming shape: csrc/gram_batch.hpp).  Shapes without such a kernel simply run one by one.  L is estimated once (one power iteration, one draw from the global
    NumPy stream) unless given.
    ``adaptive_restart`` / ``restart_threshold`` / ``tol_ratio`` (fista's arguments of the same names) keep the lockstep:
    momentum restarts and the ratio stop are decided per weight on the device every iteration and a stopped weight
    becomes a masked column of the block (three or more weights, matrix-core pass).  ``tol`` adds the reference's
    gradient-norm rule, which sits before the update: those runs go one by one.  ``return_info=True`` also returns
    ``[(iterations, stop_code), ...]`` per weight.
    ``cols=(lo, hi, n_total)`` with ``comm=``: COLUMN sharding (A is this rank's columns, b whole, each returned x this
    rank's block) - the lockstep keeps ONE exchange per row panel, the panel's 16 residual columns between the two
    products, plus 64 doubles of step norms per iteration."""
    reset_metrics()
    if delta is not None:
        assert delta > 2, "In FISTA-Δ, delta must be > 2 for convergence (course requirement)"
    prob, _ = _sharded_problem(A, b, dtype, comm, None, cols)    # comm: A, b are this rank's rows (matrix-core pass, one
    like = prob.like                                             # all-reduce of the 16 gradients per iteration)
    if L is not None:
        L_val = float(L)
    else:
        L_val = _lipschitz_cols(prob, comm, cols) if cols is not None else estimate_lipschitz(prob)
    mode = _lib.MODE_FISTA if delta is None else _lib.MODE_DELTA
    handles = []
    for a1, a2 in alphas:
        st = _core.Fista(prob)
        st.reset(t_init_factor / (L_val + (a2 if a2 > 0 else 0.0)), a1, a2, mode=mode, delta=delta or 0.0,
                 adaptive_restart=bool(adaptive_restart) and delta is None, restart_threshold=restart_threshold,
                 tol_step=tol, tol_ratio=tol_ratio, tol_grad=tol if delta is None else 0.0)
        handles.append(st)
    gtimer = _EventTimer(grad_call_times)
    # up to 4 weights: the multi-vector VALU pass where the shape has one; up to 16: the matrix-core pass
    width = 4 if len(handles) <= 4 and cols is None else 16
    for i in range(0, len(handles), width):
        group = handles[i:i + width]
        ev = gtimer.start()
        if len(group) == 1 or not _core.run_multi(group, max_iter):
            for st in group:
                st.run(max_iter)
        gtimer.stop(ev, max_iter)
    gtimer.flush()
    xs = [_core.from_device_vec(st.x_tensor(), like) for st in handles]
    if return_info:
        stats = [st.status() for st in handles]
        return xs, [(int(s.k), int(s.stopped)) for s in stats]
    return xs
