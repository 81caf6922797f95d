"""Build the gfx950 shared library ``pathpyg_amd/lib/libpathpyg_amd.so`` with hipcc.

The library is plain HIP behind a C ABI (include/pathpyg_amd.h); it does not link torch.
``hipcc`` cross-compiles for gfx950 without a GPU, so this runs in the build container and the
resulting ``.so`` travels in-tree to the GPU box.
"""
from __future__ import annotations

import concurrent.futures
import os
import pathlib
import shutil
import subprocess

PKG = pathlib.Path(__file__).resolve().parent
CSRC = PKG / "csrc"
OBJ = CSRC / "build"
LIB = PKG / "lib" / "libpathpyg_amd.so"
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the pathpyg_amd HIP library cannot be built")
    return exe


def sources() -> list[pathlib.Path]:
    return sorted(CSRC.glob("*.hip"))


def _stale(target: pathlib.Path, deps: list[pathlib.Path]) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(d.stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> pathlib.Path:
    """Compile every ``csrc/*.hip`` for gfx950 and link the shared library. Returns its path."""
    hipcc = _hipcc()
    OBJ.mkdir(parents=True, exist_ok=True)
    LIB.parent.mkdir(parents=True, exist_ok=True)
    headers = sorted(CSRC.glob("*.h")) + [PKG.parent / "include" / "pathpyg_amd.h"]
    jobs = []
    objs = []
    for src in sources():
        obj = OBJ / (src.stem + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            jobs.append([hipcc, *FLAGS, "-c", str(src), "-o", str(obj)])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        return r

    if jobs:
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(4, len(jobs))) as pool:
            list(pool.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs)])
    return LIB


if __name__ == "__main__":
    print(build(verbose=True))
