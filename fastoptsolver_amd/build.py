"""Build libfos_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m fastoptsolver_amd.build [--force]

Four translation units (csrc/fos_plan.hip, fos_comm.hip, fos_fista.hip, fos_lbfgs.hip) are compiled in parallel and linked
into ONE shared library; a unit is recompiled when it or any header is newer than its object.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
UNITS = ("fos_plan", "fos_comm", "fos_fista", "fos_lbfgs")
SOURCES = [os.path.join(CSRC, u + ".hip") for u in UNITS]
SRC = SOURCES[0]                       # (kept for callers that want "a" source of the library)
HEADERS = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp"))
HEADERS.append(os.path.join(os.path.dirname(HERE), "include", "fos.h"))
DEPS = SOURCES + HEADERS
OBJ_DIR = os.path.join(CSRC, "build")
OUT = os.path.join(HERE, "libfos_hip.so")
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-fvisibility=hidden"]


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC=/path/to/hipcc)")


def _obj(unit):
    return os.path.join(OBJ_DIR, unit + ".o")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def up_to_date():
    return not _stale(OUT, DEPS)


def compile_unit(unit, extra=(), out=None, verbose=True):
    """hipcc -c of one translation unit (also used by the compile-only tests, which add remark flags)."""
    out = out or _obj(unit)
    cmd = [hipcc_path(), *FLAGS, *extra, "-c", "-o", out, os.path.join(CSRC, unit + ".hip")]
    if verbose:
        print(" ".join(cmd), flush=True)
    return subprocess.run(cmd, check=True, capture_output=bool(extra), text=True)


def build(force=False, verbose=True):
    if not force and up_to_date():
        return OUT
    os.makedirs(OBJ_DIR, exist_ok=True)
    todo = [u for u in UNITS if force or _stale(_obj(u), [os.path.join(CSRC, u + ".hip")] + HEADERS)]
    with ThreadPoolExecutor(max_workers=len(UNITS)) as pool:
        list(pool.map(lambda u: compile_unit(u, verbose=verbose), todo))
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + [_obj(u) for u in UNITS]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
