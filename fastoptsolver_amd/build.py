"""Build libfos_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m fastoptsolver_amd.build [--force]
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "fos_api.hip")
# every source the one translation unit includes: any newer file triggers a rebuild
DEPS = sorted(os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc")) if f.endswith((".hip", ".hpp")))
DEPS.append(os.path.join(os.path.dirname(HERE), "include", "fos.h"))
OUT = os.path.join(HERE, "libfos_hip.so")


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC=/path/to/hipcc)")


def up_to_date():
    if not os.path.exists(OUT):
        return False
    t = os.path.getmtime(OUT)
    return all(os.path.getmtime(d) <= t for d in DEPS)


def build(force=False, verbose=True):
    if not force and up_to_date():
        return OUT
    cmd = [hipcc_path(), "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-o", OUT, SRC]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
