"""MI355X drop-in for the reference's ``prox_operators.py`` (stand-alone K3 kernels)."""
import numpy as np
import torch

from . import _core, _lib


def _scalar(tau, name):
    if isinstance(tau, (torch.Tensor, np.ndarray)):
        count = tau.size if isinstance(tau, np.ndarray) else tau.numel()
        if count != 1:
            raise ValueError(f"{name}: expected a scalar")
        return float(tau.reshape(-1)[0])
    return float(tau)


def _is_array(tau):
    if isinstance(tau, np.ndarray):
        return tau.size > 1
    return isinstance(tau, torch.Tensor) and tau.numel() > 1


def prox_l1(v, tau):
    """Soft threshold  sign(v)·max(|v| − τ, 0).   prox_operators.py:3-8.  τ: scalar or an array of v's length
    (the reference's NumPy expression broadcasts; an array of another shape is rejected here)."""
    lib = _lib.load()
    _core.require_gpu()
    vt = _core.to_device_vec(v)
    out = torch.empty_like(vt)
    if _is_array(tau):
        tt = _core.to_device_vec(tau, vt.device)
        if tt.numel() != vt.numel():
            raise ValueError("prox_l1: array-valued tau must have the length of v")
        with torch.cuda.device(vt.device):
            _lib.check(lib.fos_prox_l1_vec(_core.ptr(vt), _core.ptr(tt), _core.ptr(out), vt.numel(), _core.stream_ptr()),
                       "fos_prox_l1_vec")
        return _core.from_device_vec(out, v) if not isinstance(v, torch.Tensor) else out
    with torch.cuda.device(vt.device):
        _lib.check(lib.fos_prox_l1(_core.ptr(vt), _scalar(tau, "prox_l1"), _core.ptr(out), vt.numel(),
                                   _core.stream_ptr()), "fos_prox_l1")
    return _core.from_device_vec(out, v) if not isinstance(v, torch.Tensor) else out


def prox_elastic_net(v, tau, alpha1, alpha2):
    """prox_{τ(α₁‖·‖₁ + ½α₂‖·‖²)}(v) = prox_l1(v, τα₁) / (1 + τα₂).   prox_operators.py:10-16.  τ: scalar or an array
    of v's length (the reference's expression broadcasts)."""
    lib = _lib.load()
    _core.require_gpu()
    vt = _core.to_device_vec(v)
    out = torch.empty_like(vt)
    if _is_array(tau):
        tt = _core.to_device_vec(tau, vt.device)
        if tt.numel() != vt.numel():
            raise ValueError("prox_elastic_net: array-valued tau must have the length of v")
        with torch.cuda.device(vt.device):
            _lib.check(lib.fos_prox_elastic_net_vec(_core.ptr(vt), _core.ptr(tt), float(alpha1), float(alpha2),
                                                    _core.ptr(out), vt.numel(), _core.stream_ptr()),
                       "fos_prox_elastic_net_vec")
        return _core.from_device_vec(out, v) if not isinstance(v, torch.Tensor) else out
    with torch.cuda.device(vt.device):
        _lib.check(lib.fos_prox_elastic_net(_core.ptr(vt), _scalar(tau, "prox_elastic_net"), float(alpha1),
                                            float(alpha2), _core.ptr(out), vt.numel(), _core.stream_ptr()),
                   "fos_prox_elastic_net")
    return _core.from_device_vec(out, v) if not isinstance(v, torch.Tensor) else out
