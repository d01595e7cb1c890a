"""MI355X drop-in for the reference's ``prox_operators.py`` (stand-alone K3 kernels)."""
import numpy as np
import torch

from . import _core, _lib


def _is_array(tau):
    if isinstance(tau, np.ndarray):
        return tau.size > 1
    return isinstance(tau, torch.Tensor) and tau.numel() > 1


def _scalar(tau):
    return float(tau.reshape(-1)[0]) if isinstance(tau, (torch.Tensor, np.ndarray)) else float(tau)


def _apply(name, v, tau, *weights):
    """One launch of `fos_<name>` (scalar tau) or `fos_<name>_vec` (tau an array of v's length: the reference's NumPy
    expressions broadcast; an array of another shape is rejected here).  Returns the same kind of object as `v`."""
    lib = _lib.load()
    _core.require_gpu()
    vt = _core.to_device_vec(v)
    out = torch.empty_like(vt)
    if _is_array(tau):
        tt = _core.to_device_vec(tau, vt.device)
        if tt.numel() != vt.numel():
            raise ValueError(f"{name}: array-valued tau must have the length of v")
        fn, t_arg = getattr(lib, f"fos_{name}_vec"), _core.ptr(tt)
    else:
        fn, t_arg = getattr(lib, f"fos_{name}"), _scalar(tau)
    with torch.cuda.device(vt.device):
        _lib.check(fn(_core.ptr(vt), t_arg, *(float(w) for w in weights), _core.ptr(out), vt.numel(), _core.stream_ptr()),
                   fn.__name__)
    return out if isinstance(v, torch.Tensor) else _core.from_device_vec(out, v)


def prox_l1(v, tau):
    """Soft threshold  sign(v)·max(|v| − τ, 0).   prox_operators.py:3-8.  τ: scalar or an array of v's length."""
    return _apply("prox_l1", v, tau)


def prox_elastic_net(v, tau, alpha1, alpha2):
    """prox_{τ(α₁‖·‖₁ + ½α₂‖·‖²)}(v) = prox_l1(v, τα₁) / (1 + τα₂).   prox_operators.py:10-16.  τ: scalar or an array
    of v's length."""
    return _apply("prox_elastic_net", v, tau, alpha1, alpha2)
