"""CPU: the C-ABI library builds/loads and exports exactly what include/fos.h declares (no compute calls)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from fastoptsolver_amd import build, _lib
    build.build()
    return _lib.load()


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "fos.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(fos_[a-z0-9_]+)\s*\(", txt)))


def test_header_matches_binding_table(lib):
    from fastoptsolver_amd import _lib
    assert header_symbols() == sorted(_lib.SIGNATURES), "include/fos.h and _lib.SIGNATURES disagree"


def test_integration_doc_names_every_entry_point():
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert [s for s in header_symbols() if s not in doc] == []


def test_every_symbol_exported(lib):
    from fastoptsolver_amd import _lib
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (fos_[a-z0-9_]+)", out))
    assert set(header_symbols()) <= exported
    for name in header_symbols():
        assert getattr(lib, name) is not None


def test_version_and_error_paths_without_gpu(lib):
    assert lib.fos_abi_version() == 3
    h = ctypes.c_void_p()
    # argument validation happens before any HIP call
    rc = lib.fos_problem_create(ctypes.byref(h), None, 4, 4, 4, 0, None, None)
    assert rc == -1 and b"bad shape" in lib.fos_last_error()
    rc = lib.fos_prox_l1(None, 0.5, None, 4, None)
    assert rc == -1
    rc = lib.fos_lbfgs_two_loop(None, None, None, 0, 0, 0, 8, None, None)
    assert rc == -1


def test_struct_layouts_match_header():
    from fastoptsolver_amd import _lib
    assert ctypes.sizeof(_lib.FistaParams) == 8 * 8 + 4 * 4          # 8 doubles (incl. tol_grad) + 4 int32
    assert ctypes.sizeof(_lib.FistaStatus) == 11 * 8 + 8 + 2 * 4        # 11 doubles (incl. tau), int64 k, 2 int32


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "fastoptsolver_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f"{f} imports the oracle"
                assert "fos_oracle" not in txt, f"{f} references the oracle module"


def test_missing_library_is_loud(monkeypatch, tmp_path):
    from fastoptsolver_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.FosError):
        _lib.load()


def test_solvers_refuse_to_run_without_a_gpu():
    """No CPU fallback anywhere on the product path: on a box without a ROCm device every entry point of the Python
    boundary raises FosError (it never computes on the host, never touches the oracle)."""
    import numpy as np
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    import fastoptsolver_amd as fos
    from fastoptsolver_amd import _lib
    A, b = np.ones((8, 4)), np.ones(8)
    calls = [lambda: fos.fista(A, b, "lasso", 0.1, 0.0, max_iter=2),
             lambda: fos.fista_delta(A, b, "lasso", 0.1, 0.0, 3.0, max_iter=2),
             lambda: fos.estimate_lipschitz(A),
             lambda: fos.LBFGSSolver("ridge", 0.0, 1.0).fit(A, b),
             lambda: fos.compute_objective(np.zeros(4), A, b, "lasso", 0.1, 0.0),
             lambda: fos.fista_path(A, b, [(0.1, 0.0), (0.2, 0.0)], max_iter=2),
             lambda: fos.prepare(A, b)]
    for call in calls:
        with pytest.raises(_lib.FosError):
            call()
