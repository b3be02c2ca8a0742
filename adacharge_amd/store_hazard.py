"""Build-time check for the gfx950 store-data hazard found in round 3.

A MUBUF store of more than 64 bits per lane (``buffer_store_dwordx3 / x4``) reads its data registers over several
cycles; a VALU instruction that overwrites them in the next issue slots changes what the last lanes store.  LLVM's
hazard recognizer inserts the wait states unless the store has its soffset in a REGISTER
(``GCNHazardRecognizer::createsVALUHazard``) -- and on MI355X that exemption does not hold: with a scalar-resident
descriptor and an SGPR soffset the long-horizon kernel stored, for lanes 12..15 of each DPP row, the value computed for
the NEXT column tile, differently from run to run (``tools/gpu_long_race.py``).  The kernels pin ``s_nop 2`` behind every
such store (``acn_qp_long.hpp``, ``st2``); this module disassembles the gfx950 code objects of a built library and
reports every >64-bit buffer store that is followed within ``WAIT_STATES`` wait states by a VALU write of its data
registers.  ``adacharge_amd.build.build_hip_library`` refuses to keep a library that has one.
"""
import glob
import os
import re
import shutil
import subprocess
import tempfile

WAIT_STATES = 3
_LLVM_BIN = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin")


def _vregs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def scan_isa(text):
    """(number of >64-bit buffer stores, [(kernel symbol, store, wait states, overwriting instruction)]) of a disassembly
    or compiler ``-S`` listing."""
    ins = []
    for raw in text.splitlines():
        line = raw.split("//")[0].split(";")[0].strip()
        m = re.match(r"^[0-9a-f]+ <(\S+)>:$", line) or re.match(r"^(_Z\w+):$", line)
        if m:
            ins.append(("label", m.group(1)))
        elif line and not line.startswith(".") and not line.endswith(":"):
            ins.append(("ins", line))
    kern, stores, found = "?", 0, []
    for idx, (kind, l) in enumerate(ins):
        if kind == "label":
            kern = l
            continue
        if not re.match(r"buffer_store_(dwordx[34]|b(96|128))\b", l):
            continue
        stores += 1
        data = _vregs(l.split(None, 1)[1].split(",")[0].strip())
        n, j = 0, idx + 1
        while n < WAIT_STATES and j < len(ins):
            kind2, t = ins[j]
            j += 1
            if kind2 == "label":
                break
            if t.startswith("s_nop"):
                n += int(t.split()[1], 0) + 1
                continue
            n += 1
            if t.startswith("v_") and not t.startswith("v_cmp") and len(t.split(None, 1)) > 1:
                dst = t.split(None, 1)[1].split(",")[0].strip()
                if _vregs(dst) & data:
                    found.append((kern, l, n, t))
                    break
    return stores, found


class ScannerUnavailable(RuntimeError):
    """llvm-objdump is not where the ROCm image keeps it: the scan cannot run (the build warns and goes on)."""


def scan_library(path):
    """Disassemble every gfx950 code object bundled in the shared library at ``path``; returns (stores, hazards)."""
    objdump = os.path.join(_LLVM_BIN, "llvm-objdump")
    if not os.path.exists(objdump):
        raise ScannerUnavailable(f"{objdump} not found")
    stores, found = 0, []
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(path, local)   # the extractor writes next to its input
        r = subprocess.run([objdump, "--offloading", local], capture_output=True, text=True, cwd=tmp)
        if r.returncode != 0:
            raise RuntimeError("llvm-objdump --offloading failed: " + r.stderr[-400:])
        objs = sorted(glob.glob(local + ".*gfx950*"))
        if not objs:
            raise RuntimeError(f"no gfx950 code object found in {path}")
        for o in objs:
            d = subprocess.run([objdump, "-d", o], capture_output=True, text=True)
            if d.returncode != 0:
                raise RuntimeError("llvm-objdump -d failed: " + d.stderr[-400:])
            s, f = scan_isa(d.stdout)
            stores += s
            found += f
    return stores, found


def describe(found, limit=6):
    out = []
    for k, l, n, t in found[:limit]:
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip() or k
        out.append(f"  {name[:100]}\n    {l}\n    +{n}: {t}")
    return "\n".join(out)
