"""In-tree build of libacn_qp_hip.so (hipcc, gfx950 only).

``python -m adacharge_amd.build`` or ``__graft_entry__.build()``.  hipcc
cross-compiles without a GPU; the built .so is git-ignored but travels to the
GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "adacharge_amd", "csrc")
LIBDIR = os.path.join(ROOT, "adacharge_amd", "lib")
LIB = os.path.join(LIBDIR, "libacn_qp_hip.so")
SOURCES = [os.path.join(CSRC, "acn_qp_api.hip")]
DEPS = SOURCES + [os.path.join(CSRC, f) for f in ("acn_qp_tiled.hpp", "acn_qp_general.hpp", "acn_qp_stream.hpp", "acn_qp_long.hpp")] + [os.path.join(ROOT, "include", "acn_qp.h")]


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or add /opt/rocm/bin to PATH)")


def up_to_date() -> bool:
    if not os.path.exists(LIB):
        return False
    t = os.path.getmtime(LIB)
    return all(os.path.getmtime(d) <= t for d in DEPS)


def build_hip_library(force: bool = False, verbose: bool = True) -> str:
    if not force and up_to_date():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    cmd = [
        hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-honor-nans", "-mno-amdgpu-ieee", "-fPIC", "-shared",
        "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
        *SOURCES, "-o", LIB,
    ]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build_hip_library(force="--force" in sys.argv)
    print(LIB)
