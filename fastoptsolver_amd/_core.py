"""Device-side plumbing: tensors in, C-ABI handles out.  torch is used for device memory, streams and
events only; every arithmetic operation on the hot path is a HIP kernel behind include/fos.h."""
import ctypes as C

import numpy as np
import torch

from . import _lib


def require_gpu():
    if not torch.cuda.is_available():
        raise _lib.FosError("fastoptsolver_amd needs a ROCm GPU (torch.cuda.is_available() is False); "
                            "there is no CPU fallback by design")


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def is_tensor(x):
    return isinstance(x, torch.Tensor)


def to_device_vec(x, device=None):
    """1-D float32 contiguous CUDA tensor from ndarray / tensor."""
    if is_tensor(x):
        t = x.detach()
    else:
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(x)))
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    return t.to(device=dev, dtype=torch.float32).contiguous()


UPLOAD_CHUNK_BYTES = 256 << 20


def upload_matrix(src, out):
    """Host matrix -> the device view `out` (same shape; any float dtype, row- or column-major source).  The reference's
    callers hand float64 ndarrays: converting on the host first cost 0.44 s for 65536 x 8192 (the 500-iteration solve behind
    it takes 0.19 s).  Here the bytes cross in the SOURCE dtype and layout, 256 MiB at a time, and the conversion / transposition
    is the device copy into place - the rounding (to nearest even, via float32 for bf16) is the one the host cast performs."""
    m, n = src.shape
    if m == 0 or n == 0:
        return out
    esz = src.element_size()
    if src.dtype == out.dtype and src.is_contiguous() and out.is_contiguous():
        out.copy_(src)
    elif src.stride(1) == 1 and src.stride(0) == n:                      # row-major: row blocks
        step = max(1, UPLOAD_CHUNK_BYTES // (n * esz))
        for r0 in range(0, m, step):
            out[r0:r0 + step].copy_(src[r0:r0 + step].to(out.device))
    elif src.stride(0) == 1 and src.stride(1) == m:                      # column-major (Fortran order): column blocks, whose
        step = max(1, UPLOAD_CHUNK_BYTES // (m * esz))                   # transposes are contiguous on the host
        for c0 in range(0, n, step):
            out[:, c0:c0 + step].copy_(src[:, c0:c0 + step].t().to(out.device).t())
    else:
        out.copy_(src.to(out.dtype))
    return out


class Like:
    """What the caller handed in (so results come back as the same kind) without keeping the object alive."""

    def __init__(self, obj):
        self.tensor = is_tensor(obj)
        if self.tensor:
            self.dtype = obj.dtype if obj.dtype in (torch.float32, torch.float64) else torch.float32
            self.device = obj.device


def from_device_vec(t, like):
    """Return `t` as the same kind of object the caller handed in: ndarray float64, or a tensor on the caller's
    device in the caller's floating dtype (bf16 inputs get float32 back)."""
    if isinstance(like, Like):
        if like.tensor:
            return t.to(device=like.device, dtype=like.dtype, copy=True)
        return t.detach().to("cpu", torch.float64).numpy()
    if is_tensor(like):
        return from_device_vec(t, Like(like))
    return t.detach().to("cpu", torch.float64).numpy()


class Problem:
    """A (m x n, fp32 or bf16, row-major) and b bound to a fos_problem handle.

    Accepts ndarrays or tensors; a contiguous CUDA tensor of the right dtype is borrowed without a copy.
    Build it once with ``prepare(A, b)`` and pass it wherever the solvers take ``A`` to avoid re-uploading A.
    """

    def __init__(self, A, b=None, dtype=None, pad=None):
        """pad: zero-pad the columns of the device copy of A to the fused kernel's granularity (4 fp32 / 8 bf16
        elements, 16-byte aligned rows) so that a ragged n or a misaligned view still gets the single-pass kernel
        (4x faster than the two-pass path at 65536 x 8190).  Zero columns stay exactly zero through gradient and
        prox, and every vector is padded / trimmed here, so callers never see them.  None = only for problems large
        enough for it to matter (m*n >= 2^20: the padded single pass is 3-14x faster from there on); small ragged problems
        keep the fp64-accumulating two-pass path."""
        require_gpu()
        lib = _lib.load()
        self.like = Like(A)
        want_bf16 = (dtype in ("bf16", torch.bfloat16)) or (dtype is None and is_tensor(A) and A.dtype == torch.bfloat16)
        tdtype = torch.bfloat16 if want_bf16 else torch.float32
        gran = 8 if want_bf16 else 4
        dev = A.device if is_tensor(A) and A.is_cuda else torch.device("cuda", torch.cuda.current_device())
        if is_tensor(A):
            At = A.detach()
        else:
            At = torch.from_numpy(np.asarray(A))
        if At.dim() != 2:
            raise ValueError("A must be 2-D")
        m, n = int(At.shape[0]), int(At.shape[1])
        borrowable = At.is_cuda and At.dtype == tdtype and At.stride(1) == 1 and At.stride(0) >= n
        esz = 2 if want_bf16 else 4
        fused_ok = borrowable and n % gran == 0 and (At.stride(0) % gran == 0 or m == 1) and At.data_ptr() % 16 == 0
        if pad is None:
            # n <= 64 runs the row-per-thread kernel, which takes ragged / misaligned rows as they are
            pad = (not fused_ok) and m * n >= (1 << 20) and n > 64
        n_dev = n
        if pad and not fused_ok:
            n_dev = (n + gran - 1) // gran * gran
            Ap = torch.zeros(m, n_dev, dtype=tdtype, device=dev)
            if At.is_cuda:
                Ap[:, :n].copy_(At)                  # one strided device copy
            else:
                upload_matrix(At, Ap[:, :n])
            At = Ap
        elif not borrowable:
            if At.is_cuda:
                At = At.to(device=dev, dtype=tdtype).contiguous()
            else:
                At = upload_matrix(At, torch.empty(m, n, dtype=tdtype, device=dev))
        self.A = At
        self.m, self.n, self.n_dev = m, n, n_dev
        self.lda = int(At.stride(0)) if self.m > 1 else self.n_dev
        self.device = At.device
        self.dtype = "bf16" if want_bf16 else "f32"
        self.b = None if b is None else to_device_vec(b, self.device)
        if self.b is not None and self.b.numel() != self.m:
            raise ValueError("b must have m entries")
        self.gbuf = torch.zeros(self.n_dev + 4, dtype=torch.float32, device=self.device)
        self.scratch = torch.zeros(32, dtype=torch.float64, device=self.device)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(lib.fos_problem_create(C.byref(h), ptr(self.A), self.m, self.n_dev, self.lda,
                                              _lib.FOS_BF16 if want_bf16 else _lib.FOS_F32, ptr(self.b), stream_ptr()),
                       "fos_problem_create")
            self.h = h
            self._stream = torch.cuda.current_stream(self.device).cuda_stream
            _lib.check(lib.fos_problem_set_gbuf(self.h, ptr(self.gbuf)), "fos_problem_set_gbuf")
        self.lib = lib

    def __del__(self):
        h = getattr(self, "h", None)
        if h:
            try:
                self.lib.fos_problem_destroy(h)
            except Exception:
                pass
            self.h = None

    def ctx(self):
        """Device guard for a call into the library; the handle follows torch's CURRENT stream (fos_problem_set_stream
        orders the new stream behind work still enqueued on the old one), so temporaries allocated by the caching
        allocator and the kernels that read them always share a stream."""
        cur = torch.cuda.current_stream(self.device).cuda_stream
        if cur != self._stream:
            with torch.cuda.device(self.device):
                _lib.check(self.lib.fos_problem_set_stream(self.h, C.c_void_p(cur)), "fos_problem_set_stream")
            self._stream = cur
        return torch.cuda.device(self.device)

    # ---- user-length <-> device-length vectors ------------------------------------------------------------
    def vec_in(self, x, dtype=torch.float32):
        """1-D device tensor of the kernel's length n_dev (zero beyond n) from a user vector of length n or n_dev."""
        t = x.detach() if is_tensor(x) else torch.from_numpy(np.ascontiguousarray(np.asarray(x)))
        t = t.to(device=self.device, dtype=dtype).contiguous()
        if t.numel() == self.n_dev:
            return t
        if t.numel() != self.n:
            raise ValueError(f"expected a vector of length {self.n}")
        out = torch.zeros(self.n_dev, dtype=dtype, device=self.device)
        out[: self.n] = t
        return out

    def vec_out(self, t):
        """Trim a device-length vector (or [k, n_dev] history block) to the user's n."""
        return t if self.n_dev == self.n else t[..., : self.n]

    # ---- plan / tuning -------------------------------------------------------------------------------
    def plan(self):
        arr = (C.c_int32 * 8)()
        _lib.check(self.lib.fos_problem_plan(self.h, arr), "fos_problem_plan")
        keys = ("path", "threads", "chunks", "rows", "workgroups", "slabs", "nontemporal", "cus")
        plan = dict(zip(keys, list(arr)))
        flags = plan["nontemporal"]
        plan["nontemporal"] = flags & 1
        plan["resident"] = (flags >> 1) & 1       # small problem: whole runs execute in one LDS-resident launch
        plan["tall"] = (flags >> 2) & 1           # n <= 64: row-per-thread single pass (any alignment)
        plan["colblock"] = (flags >> 3) & 1       # rows wider than any single-pass kernel: column blocks, two phases
        plan["cluster"] = (flags >> 4) & 1        # multi-lambda pass: one-read cluster form planned (after its first run)
        plan["interleave"] = (flags >> 5) & 1     # streaming pass: rows dealt round-robin to the workgroups
        plan["fused_mfma"] = (flags >> 6) & 1     # opt-in: plain runs take the one-launch persistent step
        plan["chip_resident"] = (flags >> 7) & 1  # tall-skinny plain runs keep A in the LDS of up to all CUs: forced on
        return plan

    def replan(self, no_resident=False, no_tall=False, no_wide=False, no_colblock=False, cluster=None, interleave=None,
               fused_mfma=False, chip_resident=None):
        """Re-run the planner with kernel families switched off; ``cluster`` / ``interleave``: True / False force the
        one-read cluster form of the multi-weight pass / the round-robin row order on or off, None leaves the planner's
        choice; ``fused_mfma=True`` opts plain runs in to the one-launch persistent step (fos_fista_run_fused),
        ``chip_resident=True`` / ``False`` takes the chip-resident loop for tall-skinny plain runs (fos_fista_run_chip)
        wherever it is served / never, None leaves the planner's region (n <= 8, up to 131072 rows)
        (fos_problem_replan): tests and A/B measurements.
        Call before creating Fista handles on this problem."""
        flags = ((_lib.PLAN_NO_RESIDENT if no_resident else 0) | (_lib.PLAN_NO_TALL if no_tall else 0) |
                 (_lib.PLAN_NO_WIDE if no_wide else 0) | (_lib.PLAN_NO_COLBLOCK if no_colblock else 0) |
                 (0 if cluster is None else (_lib.PLAN_CLUSTER if cluster else _lib.PLAN_NO_CLUSTER)) |
                 (0 if interleave is None else (_lib.PLAN_INTERLEAVE if interleave else _lib.PLAN_NO_INTERLEAVE)) |
                 (_lib.PLAN_FUSED_MFMA if fused_mfma else 0) |
                 (0 if chip_resident is None else (_lib.PLAN_CHIP_RESIDENT if chip_resident else _lib.PLAN_NO_CHIP_RESIDENT)))
        with self.ctx():
            _lib.check(self.lib.fos_problem_replan(self.h, flags), "fos_problem_replan")

    def set_comm(self, comm):
        """Attach a `distributed.Comm` (or None): this problem's rows are one shard of a row-sharded problem and every
        row-sum is all-reduced on the stream (fos_problem_set_comm)."""
        _lib.check(self.lib.fos_problem_set_comm(self.h, comm.h if comm is not None else None), "fos_problem_set_comm")
        self.comm = comm               # keeps the communicator alive as long as the problem

    def set_comm_cols(self, comm):
        """COLUMN sharding: this problem holds A[:, this rank's columns] (all rows) and the whole b; the iterate is
        partitioned over the ranks (fos_problem_set_comm_cols)."""
        with self.ctx():
            _lib.check(self.lib.fos_problem_set_comm_cols(self.h, comm.h), "fos_problem_set_comm_cols")
        self.comm = comm
        self.col_sharded = True

    def tune(self, threads, chunks, rows, workgroups=0):
        _lib.check(self.lib.fos_problem_tune(self.h, threads, chunks, rows, workgroups), "fos_problem_tune")

    def profile(self, every=1):
        """Bracket every `every`-th A-pass launch with HIP events (0 / False = off)."""
        _lib.check(self.lib.fos_problem_profile(self.h, int(every)), "fos_problem_profile")

    def profile_read(self):
        """(device milliseconds, launches) of the A-pass kernel since the last read; synchronises."""
        ms, cnt = C.c_double(), C.c_int64()
        with self.ctx():
            _lib.check(self.lib.fos_problem_profile_read(self.h, C.byref(ms), C.byref(cnt)), "fos_problem_profile_read")
        return ms.value, cnt.value

    # ---- kernels -------------------------------------------------------------------------------------
    def gemv_pair(self, y, alpha2=0.0, out=None, rr_out=None):
        """grad = A^T (A y - b) + alpha2 y (device tensor); rr_out: optional 1-element float64 device tensor."""
        user_out = out
        if out is None or out.numel() != self.n_dev:
            out = torch.empty(self.n_dev, dtype=torch.float32, device=self.device)
        y = self.vec_in(y)
        with self.ctx():
            _lib.check(self.lib.fos_gemv_pair(self.h, ptr(y), float(alpha2), ptr(out), ptr(rr_out)), "fos_gemv_pair")
        if user_out is not None and user_out is not out:
            user_out.copy_(self.vec_out(out))
            return user_out
        return self.vec_out(out)

    def residual_objective(self, x):
        """Host tuple (||Ax-b||^2, ||x||^2, ||x||_1); synchronises.  x is rounded to fp32 for the pass over A."""
        x = self.vec_in(x)
        with self.ctx():
            _lib.check(self.lib.fos_residual_objective(self.h, ptr(x), ptr(self.scratch)), "fos_residual_objective")
        v = self.scratch[:3].cpu()
        return float(v[0]), float(v[1]), float(v[2])

    def residual_batch(self, X, use_b=True):
        """||A X_j - b||^2 for the <= 16 columns of X (n x nv) in one MFMA pass; host list.  Synchronises."""
        X = torch.as_tensor(X, device=self.device, dtype=torch.float32)
        nv = X.shape[1]
        Xf = torch.zeros(self.n_dev, 16, dtype=torch.float32, device=self.device)
        Xf[: X.shape[0], :nv] = X
        with self.ctx():
            _lib.check(self.lib.fos_residual_batch(self.h, ptr(Xf), nv, int(bool(use_b)), ptr(self.scratch)),
                       "fos_residual_batch")
        return self.scratch[:nv].cpu().tolist()

    def power_iter(self, v0, n_iter=100, tol=1e-6):
        v = self.vec_in(v0).clone()
        L = C.c_double()
        it = C.c_int()
        with self.ctx():
            _lib.check(self.lib.fos_power_iter(self.h, ptr(v), int(n_iter), float(tol), C.byref(L), C.byref(it)),
                       "fos_power_iter")
        return L.value, it.value, self.vec_out(v)


def prepare(A, b=None, dtype=None, pad=None):
    """Upload/bind A (and b) once; the result can be passed as ``A`` to every solver (``b`` may then be None)."""
    return A if isinstance(A, Problem) else Problem(A, b, dtype, pad)


def as_problem(A, b, dtype=None):
    if isinstance(A, Problem):
        if b is not None and A.b is None:
            raise ValueError("Problem was prepared without b")
        return A
    return Problem(A, b, dtype)


class Fista:
    """fos_fista handle: x_k, x_{k-1} and the momentum scalars live on the device."""

    def __init__(self, prob):
        self.prob = prob
        self.lib = prob.lib
        h = C.c_void_p()
        with prob.ctx():
            _lib.check(self.lib.fos_fista_create(prob.h, C.byref(h)), "fos_fista_create")
        self.h = h
        self.prm = _lib.FistaParams()

    def __del__(self):
        h = getattr(self, "h", None)
        if h:
            try:
                self.lib.fos_fista_destroy(h)
            except Exception:
                pass
            self.h = None

    def reset(self, tau, alpha1, alpha2, mode=_lib.MODE_FISTA, prox_kind=_lib.PROX_L1, delta=0.0,
              adaptive_restart=False, restart_threshold=1.0, tol_step=0.0, tol_ratio=0.0, tol_grad=0.0, x0=None):
        p = self.prm
        p.tau, p.alpha1, p.alpha2, p.delta = float(tau), float(alpha1), float(alpha2), float(delta)
        p.restart_threshold, p.tol_step, p.tol_ratio = float(restart_threshold), float(tol_step), float(tol_ratio)
        p.tol_grad = float(tol_grad)
        p.mode, p.prox_kind, p.adaptive_restart, p.reserved = int(mode), int(prox_kind), int(bool(adaptive_restart)), 0
        if x0 is not None:
            x0 = self.prob.vec_in(x0, torch.float64)
        with self.prob.ctx():
            _lib.check(self.lib.fos_fista_reset(self.h, C.byref(p), ptr(x0)), "fos_fista_reset")

    def set_precise(self, on=True, own_buffer=False):
        """Split-form gradient from the fp64-accumulating pass at the unrounded y_k (fos_fista_set_precise).
        own_buffer: keep [gradient ; ||r||^2] in a torch tensor of this object (`gbuf64`, n_dev + 4 doubles) so that a
        split-form reducer can sum it over the ranks between grad() and its consumers (fos_fista_set_gbuf64)."""
        with self.prob.ctx():
            if on and own_buffer and getattr(self, "gbuf64", None) is None:
                self.gbuf64 = torch.zeros(self.prob.n_dev + 4, dtype=torch.float64, device=self.prob.device)
                _lib.check(self.lib.fos_fista_set_gbuf64(self.h, ptr(self.gbuf64)), "fos_fista_set_gbuf64")
            _lib.check(self.lib.fos_fista_set_precise(self.h, int(bool(on))), "fos_fista_set_precise")
        self.precise = bool(on)

    def set_tau(self, tau):
        _lib.check(self.lib.fos_fista_set_tau(self.h, float(tau)), "fos_fista_set_tau")

    def run(self, iters):
        with self.prob.ctx():
            _lib.check(self.lib.fos_fista_run(self.h, int(iters)), "fos_fista_run")

    def run_fused(self, iters):
        """`iters` plain iterations in ONE persistent launch (fos_fista_run_fused: LDS-staged panels, row dots on the matrix
        cores, the owned slice of the iterate resident in LDS).  False when this problem / configuration is not served."""
        with self.prob.ctx():
            rc = self.lib.fos_fista_run_fused(self.h, int(iters))
        if rc == -4:
            return False
        _lib.check(rc, "fos_fista_run_fused")
        return True

    def run_chip(self, iters):
        """`iters` plain iterations in ONE launch with A resident in the LDS of up to all CUs and one grid-wide barrier per
        iteration (fos_fista_run_chip: tall-skinny fp32 problems, n <= 16).  False when not served."""
        with self.prob.ctx():
            rc = self.lib.fos_fista_run_chip(self.h, int(iters))
        if rc == -4:
            return False
        _lib.check(rc, "fos_fista_run_chip")
        return True

    def run_history(self, iters):
        """Device-resident history run: (x_hist [iters, n] float64, hist [iters, 4] float64 =
        {||Ax-b||^2, ||x||_1, ||x||_2^2, ||dx||^2}) as device tensors, or None when this solver configuration /
        problem shape needs the host-driven loop."""
        dev = self.prob.device
        xh = torch.empty(iters, self.prob.n_dev, dtype=torch.float64, device=dev)
        hist = torch.empty(iters, 4, dtype=torch.float64, device=dev)
        nbytes = self.lib.fos_fista_history_workspace(self.h, int(iters))
        work = torch.empty(max(1, (nbytes + 7) // 8), dtype=torch.float64, device=dev)
        with self.prob.ctx():
            rc = self.lib.fos_fista_run_history(self.h, int(iters), ptr(xh), ptr(hist), ptr(work))
        if rc == -4:
            return None
        _lib.check(rc, "fos_fista_run_history")
        self._keep = work            # stays alive until the stream has consumed it (next sync)
        return self.prob.vec_out(xh), hist

    def run_resident(self, iters, *, backtracking=False, eta=0.5, armijo_c=1e-2, grad_tol=0.0, record=False):
        """Small problems: the whole run - backtracking, restart, stops, history included - in one launch of the
        LDS-resident loop (fos_fista_run_resident).  Returns dict(done, tau, ls, taus[, x, hist]) with host lists for
        the per-iteration integers / steps and device tensors for the history, or None when the problem does not fit.
        Synchronises."""
        if not self.prob.plan().get("resident"):
            return None
        dev = self.prob.device
        iters = int(iters)
        xh = torch.empty(max(iters, 1), self.prob.n_dev, dtype=torch.float64, device=dev) if record else None
        hist = torch.empty(max(iters, 1), 4, dtype=torch.float64, device=dev) if record else None
        ls = torch.zeros(max(iters, 1), dtype=torch.int32, device=dev)
        taus = torch.zeros(max(iters, 1), dtype=torch.float64, device=dev)
        done, tau = C.c_int32(0), C.c_double(0.0)
        with self.prob.ctx():
            rc = self.lib.fos_fista_run_resident(self.h, iters, 1 if backtracking else 0, float(eta), float(armijo_c),
                                                 float(grad_tol), ptr(xh), ptr(hist), ptr(ls), ptr(taus),
                                                 C.byref(done), C.byref(tau))
        if rc == -4:
            return None
        _lib.check(rc, "fos_fista_run_resident")
        k = int(done.value)
        out = dict(done=k, tau=float(tau.value), ls=ls[:k].cpu().tolist(), taus=taus[:k].cpu().tolist())
        if record:
            out["x"] = self.prob.vec_out(xh[:k])
            out["hist"] = hist[:k]
        return out

    def run_backtracking(self, iters, eta, armijo_c, grad_eps):
        """Enqueue `iters` backtracking iterations decided on the device (fos_fista_run_backtracking).  Returns
        (ls_iters int32 tensor, tau_hist float64 tensor) - device tensors, valid for the completed iterations - or
        None when this plan has no candidate pass (callers then drive the search from the host)."""
        dev = self.prob.device
        ls = torch.zeros(max(int(iters), 1), dtype=torch.int32, device=dev)
        taus = torch.zeros(max(int(iters), 1), dtype=torch.float64, device=dev)
        with self.prob.ctx():
            rc = self.lib.fos_fista_run_backtracking(self.h, int(iters), float(eta), float(armijo_c), float(grad_eps),
                                                     ptr(ls), ptr(taus))
        if rc == -4:
            return None
        _lib.check(rc, "fos_fista_run_backtracking")
        return ls, taus

    def run_recorded(self, iters, backtracking, eta, armijo_c, grad_eps, want_rr=True):
        """Device-driven iterations with the history recorded on the device (fos_fista_run_recorded).  Returns dict of
        device tensors x [iters, n], hist [iters, 4], rr_seen [iters], ls [iters], taus [iters] - or None when
        unsupported for this plan."""
        dev, iters = self.prob.device, int(iters)
        rec = dict(x=torch.empty(max(iters, 1), self.prob.n_dev, dtype=torch.float64, device=dev),
                   hist=torch.zeros(max(iters, 1), 4, dtype=torch.float64, device=dev),
                   rr_seen=torch.full((max(iters, 1),), float("nan"), dtype=torch.float64, device=dev),
                   ls=torch.zeros(max(iters, 1), dtype=torch.int32, device=dev),
                   taus=torch.zeros(max(iters, 1), dtype=torch.float64, device=dev))
        with self.prob.ctx():
            rc = self.lib.fos_fista_run_recorded(self.h, iters, int(bool(backtracking)), float(eta), float(armijo_c),
                                                 float(grad_eps), ptr(rec["x"]), ptr(rec["hist"]),
                                                 ptr(rec["rr_seen"] if want_rr else None),
                                                 ptr(rec["ls"]), ptr(rec["taus"]))
        if rc == -4:
            return None
        _lib.check(rc, "fos_fista_run_recorded")
        return rec

    def resume_after_stall(self):
        """The device parked a search whose 16 candidates were all rejected: take the current step back to the host."""
        tau = C.c_double()
        with self.prob.ctx():
            _lib.check(self.lib.fos_fista_resume_after_stall(self.h, C.byref(tau)), "fos_fista_resume_after_stall")
        return float(tau.value)

    def grad(self, dual=False):
        """Gradient pass at y_k; dual=True also leaves ||A x_k - b||^2 in status().rr_x (same pass over A)."""
        with self.prob.ctx():
            if dual:
                _lib.check(self.lib.fos_fista_grad_dual(self.h), "fos_fista_grad_dual")
            else:
                _lib.check(self.lib.fos_fista_grad(self.h), "fos_fista_grad")

    def update(self):
        with self.prob.ctx():
            _lib.check(self.lib.fos_fista_update(self.h), "fos_fista_update")

    def trial(self, t, with_residual=True):
        out = (C.c_double * 8)()
        with self.prob.ctx():
            _lib.check(self.lib.fos_fista_trial(self.h, float(t), int(bool(with_residual)), out), "fos_fista_trial")
        keys = ("gd", "dd", "nnz", "gnorm2", "y2", "q", "rr_y")
        return dict(zip(keys, list(out)))

    def trial_batch(self, t, eta, nv=16):
        """Candidates t*eta^j, j < nv, decided by one MFMA pass over A; list of dicts like trial().  None when the
        problem runs the two-pass fallback."""
        out = (C.c_double * (8 * nv))()
        with self.prob.ctx():
            rc = self.lib.fos_fista_trial_batch(self.h, float(t), float(eta), int(nv), out)
        if rc == -4:
            return None
        _lib.check(rc, "fos_fista_trial_batch")
        keys = ("gd", "dd", "nnz", "gnorm2", "y2", "q", "rr_y")
        return [dict(zip(keys, out[8 * j: 8 * j + 7])) for j in range(nv)]

    def status(self):
        st = _lib.FistaStatus()
        with self.prob.ctx():
            _lib.check(self.lib.fos_fista_status_get(self.h, C.byref(st)), "fos_fista_status_get")
        return st

    def x_tensor(self):
        """Copy of x_k as a float64 device tensor."""
        out = torch.empty(self.prob.n_dev, dtype=torch.float64, device=self.prob.device)
        with self.prob.ctx():
            _lib.check(self.lib.fos_fista_get_x(self.h, ptr(out)), "fos_fista_get_x")
        return self.prob.vec_out(out)


def stream_read_probe(t, launches=20):
    """(GB/s, microseconds per pass) of a read-only pass over the contiguous CUDA tensor `t` (fos_stream_read_probe): the
    box's own streaming-read figure that bench.py reports beside the nominal peak.  Synchronises."""
    if not (t.is_cuda and t.is_contiguous()):
        raise ValueError("stream_read_probe: contiguous device tensor expected")
    nbytes = t.numel() * t.element_size() // 16 * 16
    gbps, us = C.c_double(), C.c_double()
    with torch.cuda.device(t.device):
        _lib.check(_lib.load().fos_stream_read_probe(ptr(t), nbytes, int(launches), torch.cuda.current_stream().cuda_stream,
                                                    C.byref(gbps), C.byref(us)), "fos_stream_read_probe")
    return gbps.value, us.value


def run_multi(handles, iters):
    """Advance up to 16 Fista handles of one Problem in lockstep (fos_fista_run_multi).
    Returns False when this shape / configuration has no multi-vector kernel (callers then run them one by one)."""
    lib = handles[0].lib
    arr = (C.c_void_p * len(handles))(*[h.h for h in handles])
    with handles[0].prob.ctx():
        rc = lib.fos_fista_run_multi(arr, len(handles), int(iters))
    if rc == -4:
        return False
    _lib.check(rc, "fos_fista_run_multi")
    return True
