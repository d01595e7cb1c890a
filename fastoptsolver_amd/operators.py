"""Device callables for ``ista`` (the reference passes closures; these are the recognised, fusable
equivalents) and thin wrappers over the vector kernels."""
import torch

from . import _core, _lib


def vec_axpby(a, x, b, y, out=None):
    """out = a*x + b*y on the device (fos_vec_axpby)."""
    lib = _lib.load()
    if out is None:
        out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(lib.fos_vec_axpby(float(a), _core.ptr(x), float(b), _core.ptr(y if b != 0.0 else None),
                                     _core.ptr(out), x.numel(), _core.stream_ptr()), "fos_vec_axpby")
    return out


def vec_stats(x, g, d):
    """Host list [x.x, g.d, d.d, max|g|, ||x||_1] in one launch (fos_vec_stats); any argument may be None.
    Synchronises."""
    lib = _lib.load()
    ref = next(t for t in (x, g, d) if t is not None)
    out = torch.empty(5, dtype=torch.float64, device=ref.device)
    with torch.cuda.device(ref.device):
        _lib.check(lib.fos_vec_stats(_core.ptr(x), _core.ptr(g), _core.ptr(d), ref.numel(), _core.ptr(out),
                                     _core.stream_ptr()), "fos_vec_stats")
    return out.cpu().tolist()


class LeastSquares:
    """g(x) = 0.5||Ax-b||^2 + 0.5*alpha2*||x||^2 bound to a device problem.

    ``ls = LeastSquares(A, b, alpha2); ista(x0, ls, ls.grad, L1Prox(alpha1), L)`` is recognised by ``ista`` and
    runs fused; called on their own, ``ls(x)`` is one residual pass (K5) and ``ls.grad(x)`` the single-pass
    GEMV pair (K2).  Mirrors the closures a reference user writes around ``A @ x - b``."""

    def __init__(self, A, b=None, alpha2=0.0, dtype=None):
        self.prob = _core.as_problem(A, b, dtype)
        self.alpha2 = float(alpha2)

    def __call__(self, x):
        xt = _core.to_device_vec(x, self.prob.device)
        rr, x2, _ = self.prob.residual_objective(xt)
        return 0.5 * rr + 0.5 * self.alpha2 * x2

    def grad(self, x):
        xt = _core.to_device_vec(x, self.prob.device)
        return self.prob.gemv_pair(xt, self.alpha2)


class L1Prox:
    """prox_h(v, t) = prox_l1(v, t*alpha1)   (the closure of prox_operators.py:3 a reference user writes)."""

    def __init__(self, alpha1):
        self.alpha1 = float(alpha1)

    def __call__(self, v, t):
        from .prox_operators import prox_l1
        return prox_l1(v, t * self.alpha1)


class ElasticNetProx:
    """prox_h(v, t) = prox_elastic_net(v, t, alpha1, alpha2)   (prox_operators.py:10)."""

    def __init__(self, alpha1, alpha2):
        self.alpha1, self.alpha2 = float(alpha1), float(alpha2)

    def __call__(self, v, t):
        from .prox_operators import prox_elastic_net
        return prox_elastic_net(v, t, self.alpha1, self.alpha2)
