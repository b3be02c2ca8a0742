"""In-tree build of libacn_qp_hip.so (hipcc, gfx950 only).

``python -m adacharge_amd.build`` or ``__graft_entry__.build()``.  hipcc
cross-compiles without a GPU; the built .so is git-ignored but travels to the
GPU box with the repo snapshot.  Every kernel family is its own translation unit
(objects under adacharge_amd/lib/obj/, rebuilt only when a file they include
changed), compiled in parallel and linked into one shared library.
"""
from __future__ import annotations

import concurrent.futures
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "adacharge_amd", "csrc")
INC = os.path.join(ROOT, "include")
LIBDIR = os.path.join(ROOT, "adacharge_amd", "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libacn_qp_hip.so")
UNITS = ("acn_qp_api", "acn_qp_wave", "acn_qp_tiled_ct1", "acn_qp_tiled_ct2", "acn_qp_stream", "acn_qp_long", "acn_qp_general", "acn_qp_polish")
SOURCES = [os.path.join(CSRC, u + ".hip") for u in UNITS]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-honor-nans", "-mno-amdgpu-ieee", "-fPIC"]

_INCLUDE = re.compile(r'^\s*#\s*include\s+"([^"]+)"', re.M)


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or add /opt/rocm/bin to PATH)")


def dependencies(path: str, seen=None) -> set:
    """The file and every project header it includes (transitively)."""
    seen = set() if seen is None else seen
    if path in seen or not os.path.exists(path):
        return seen
    seen.add(path)
    for name in _INCLUDE.findall(open(path).read()):
        for base in (CSRC, INC):
            dependencies(os.path.join(base, name), seen)
    return seen


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def up_to_date() -> bool:
    deps = set()
    for s in SOURCES:
        deps |= dependencies(s)
    return not _stale(LIB, deps | {os.path.abspath(__file__)})


def build_hip_library(force: bool = False, verbose: bool = True, extra_flags=(), out: str = LIB, check_hazards: bool = True) -> str:
    """Compile stale translation units (in parallel) and link.  ``extra_flags`` / ``out``: diagnostic builds
    (e.g. ``-DACNQP_STAMPS``) into another file, always from scratch.  ``check_hazards``: disassemble the linked
    library and refuse it if a >64-bit buffer store is followed by a VALU write of its data registers (the gfx950
    store-data hazard the compiler does not cover, adacharge_amd/store_hazard.py); a build whose scanner is missing
    FAILS unless ACNQP_SKIP_HAZARD_SCAN=1; the outcome is written next to the library (``*.hazard_scan.json``)."""
    diagnostic = bool(extra_flags) or out != LIB
    if not force and not diagnostic and up_to_date():
        return LIB
    objdir = OBJDIR if not diagnostic else OBJDIR + "_diag"
    os.makedirs(objdir, exist_ok=True)
    todo = []
    for unit in UNITS:
        obj = os.path.join(objdir, unit + ".o")
        if force or diagnostic or _stale(obj, dependencies(os.path.join(CSRC, unit + ".hip")) | {os.path.abspath(__file__)}):
            todo.append(unit)
    workers = max(1, min(len(todo), (os.cpu_count() or 2) - 1, 6))

    def one(unit):
        src = os.path.join(CSRC, unit + ".hip")
        obj = os.path.join(objdir, unit + ".o")
        cmd = [hipcc_path(), *FLAGS, *extra_flags, "-I" + INC, "-I" + CSRC, "-c", src, "-o", obj]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    if todo:
        with concurrent.futures.ThreadPoolExecutor(workers) as pool:
            list(pool.map(one, todo))
    objs = [os.path.join(objdir, u + ".o") for u in UNITS]
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-fPIC", "-shared", *objs, "-o", out]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    if check_hazards:
        try:
            from . import store_hazard
        except ImportError:   # run as a script
            sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
            import store_hazard

        if os.environ.get("ACNQP_SKIP_HAZARD_SCAN") == "1":
            print("[build] WARNING: store-data hazard scan skipped (ACNQP_SKIP_HAZARD_SCAN=1)", flush=True)
            store_hazard.write_stamp(out, "skipped")
            return out
        try:
            spills = []
            stores, found = store_hazard.scan_library(out, spills)
        except store_hazard.ScannerUnavailable as e:
            os.replace(out, out + ".rejected")
            raise RuntimeError(f"store-data hazard scan cannot run ({e}); the scan is the only protection against the gfx950 "
                               f"buffer-store hazard of DESIGN.md section 3.6 -- set ACNQP_SKIP_HAZARD_SCAN=1 to build "
                               f"without it; library kept as {out}.rejected") from e
        store_hazard.write_stamp(out, "scanned", stores, len(found))
        if verbose:
            print(f"[build] store-data hazard scan: {stores} buffer stores of more than 64 bits, {len(found)} unguarded", flush=True)
        if found:
            os.replace(out, out + ".rejected")
            raise RuntimeError(f"{len(found)} buffer stores of more than 64 bits are followed by a VALU write of their data "
                               f"registers (gfx950 store-data hazard, DESIGN.md section 3.6); library kept as {out}.rejected\n"
                               + store_hazard.describe(found))
        if verbose:
            print(f"[build] spill scan: {len(spills)} register spills / reloads ahead of an exec restore", flush=True)
        if spills:
            os.replace(out, out + ".rejected")
            raise RuntimeError(f"{len(spills)} register spills / reloads sit before the exec restore of a join block (stored or "
                               f"loaded with the lanes of the finished region only: store_hazard.scan_exec_spills); library "
                               f"kept as {out}.rejected\n" + "\n".join(f"  {k[:100]}\n    {l}" for k, l in spills[:8]))
    return out


if __name__ == "__main__":
    build_hip_library(force="--force" in sys.argv)
    print(LIB)
