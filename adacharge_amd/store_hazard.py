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


_BRANCH = re.compile(r"^s_(branch|cbranch_\w+)\s+(\S+)")


def _parse(text):
    """[(kind, text, address or None, branch target or None)]: kind "label" (a kernel symbol) or "ins".  Targets are
    instruction addresses for an llvm-objdump disassembly (``<symbol+0xOFFSET>`` in the comment, resolved with the
    symbol's own address) and ``.L`` label names for a compiler ``-S`` listing."""
    out, sym_addr = [], {}
    for raw in text.splitlines():
        code, _, comment = raw.partition("//")
        line = code.split(";")[0].strip()
        m = re.match(r"^([0-9a-f]+) <(\S+)>:$", line)
        if m:
            sym_addr[m.group(2)] = int(m.group(1), 16)
            out.append(("label", m.group(2), int(m.group(1), 16), None))
            continue
        m = re.match(r"^(_Z\w+):$", line)
        if m:
            out.append(("label", m.group(1), None, None))
            continue
        m = re.match(r"^(\.L\w+):$", line)
        if m:
            out.append(("local", m.group(1), None, None))
            continue
        if not line or line.startswith(".") or line.endswith(":"):
            continue
        addr = None
        ma = re.match(r"\s*([0-9A-Fa-f]+):", comment)
        if ma:
            addr = int(ma.group(1), 16)
        target = None
        mb = _BRANCH.match(line)
        if mb:
            mt = re.search(r"<([^>+]+)(?:\+0x([0-9a-fA-F]+))?>", comment)
            if mt and mt.group(1) in sym_addr:
                target = sym_addr[mt.group(1)] + int(mt.group(2) or "0", 16)
            elif mb.group(2).startswith(".L"):
                target = mb.group(2)
        out.append(("ins", line, addr, target))
    return out


def scan_isa(text):
    """(number of >64-bit buffer stores, [(kernel symbol, store, wait states, overwriting instruction)]) of a disassembly
    or compiler ``-S`` listing.  Every path of ``WAIT_STATES`` wait states behind a store is walked: the fall-through
    one, and the target of every unconditional or conditional branch met on the way (a store that ends a basic block is
    followed into both successors)."""
    ins = _parse(text)
    where = {}
    for idx, (kind, l, addr, _) in enumerate(ins):
        if kind == "ins" and addr is not None:
            where[addr] = idx
        elif kind == "local":
            where[l] = idx
    kern, stores, found = "?", 0, []
    for idx, (kind, l, _, _) in enumerate(ins):
        if kind == "label":
            kern = l
            continue
        if kind != "ins" or not re.match(r"buffer_store_(dwordx[34]|b(96|128))\b", l):
            continue
        stores += 1
        data = _vregs(l.split(None, 1)[1].split(",")[0].strip())
        hit, todo, seen = None, [(idx + 1, 0)], set()
        while todo and hit is None:
            j, n = todo.pop()
            while n < WAIT_STATES and j < len(ins):
                if (j, n) in seen:
                    break
                seen.add((j, n))
                kind2, t, _, target = ins[j]
                j += 1
                if kind2 == "label":
                    break
                if kind2 == "local":
                    continue
                if t.startswith("s_nop"):
                    n += int(t.split()[1], 0) + 1
                    continue
                n += 1
                if t.startswith("s_endpgm"):
                    break
                mb = _BRANCH.match(t)
                if mb:
                    if target in where and n < WAIT_STATES:
                        todo.append((where[target], n))
                    if mb.group(1) == "branch":
                        break          # unconditional: no fall-through
                    continue
                if t.startswith("v_") and not t.startswith("v_cmp") and len(t.split(None, 1)) > 1:
                    dst = t.split(None, 1)[1].split(",")[0].strip()
                    if _vregs(dst) & data:
                        hit = (kern, l, n, t)
                        break
        if hit:
            found.append(hit)
    return stores, found


_EXEC_WRITE = re.compile(r"^s_\w+_saveexec_b64\b|^s_\w+\s+exec\b|^s_\w+\s+exec_(lo|hi)\b")
_EXEC_BLIND = re.compile(r"^(v_writelane_b32\b|v_readlane_b32\b|s_nop\b|s_waitcnt\b|scratch_store_|scratch_load_|v_accvgpr_(write|read)_b32\b)")
# (scratch accesses only: the allocator also spills to AGPRs, but a v_accvgpr copy cannot be told from the masked update of
#  a variable that lives there -- the tiled kernel has fourteen legitimate ones in this position)
_SPILL_LIKE = re.compile(r"^scratch_(store|load)_")


def scan_exec_spills(text):
    """[(kernel symbol, instruction)]: register spills / reloads (``scratch_store`` / ``scratch_load``) that sit at the
    top of a join block BEFORE the ``s_or_b64 exec, exec, s[..]`` that restores the lanes of the divergent region which
    ends there.  Round 4 found the compiler (ROCm 7.2 LLVM) doing this to values that live across the region: behind
    ``if (lane < 30) for (...) AaH[k] = 0`` the large-site kernel spilled two block-uniform doubles with exec = 0 --
    nothing was stored -- and the objective reported at the end was built from the slot's garbage.  The scan walks each
    scratch access forward over instructions that do not depend on exec (SGPR spills by v_writelane, scalar ALU, waits);
    reaching the exec restore before anything else, in a block that has not itself narrowed exec, is the pattern."""
    ins = _parse(text)
    leaders, targets = set(), set()
    where = {}
    for idx, (kind, l, addr, _) in enumerate(ins):
        if kind == "ins" and addr is not None:
            where[addr] = idx
        elif kind == "local":
            where[l] = idx
    for idx, (kind, l, _, target) in enumerate(ins):
        if kind != "ins":
            leaders.add(idx + 1)
            continue
        if _BRANCH.match(l):
            leaders.add(idx + 1)
            if target in where:
                t = where[target]
                while t < len(ins) and ins[t][0] != "ins":   # a .L label of a -S listing: its first instruction
                    t += 1
                leaders.add(t)
                targets.add(t)
    kern, found = "?", []
    for idx, (kind, l, _, _) in enumerate(ins):
        if kind == "label":
            kern = l
            continue
        if kind != "ins" or not _SPILL_LIKE.match(l):
            continue
        # back to the block leader: the block must be a JOIN (some branch lands on it -- the fall-through block behind
        # an s_cbranch_execz is the region's own body, whose masked accesses are its business) and must not have
        # narrowed exec itself before this access
        j, own, lead = idx, False, None
        while j >= 0 and ins[j][0] == "ins":
            if j != idx and _EXEC_WRITE.match(ins[j][1]):
                own = True
                break
            if j in leaders:
                lead = j
                break
            j -= 1
        if own or lead not in targets:
            continue
        j = idx + 1
        while j < len(ins) and ins[j][0] == "ins" and j not in leaders:
            t = ins[j][1]
            if re.match(r"^s_or_b64 exec, exec,", t):
                found.append((kern, l))
                break
            if _BRANCH.match(t) or t.startswith("s_endpgm"):
                break
            if _EXEC_BLIND.match(t) or (t.startswith("s_") and not _EXEC_WRITE.match(t)):
                j += 1
                continue
            break
    return found


class ScannerUnavailable(RuntimeError):
    """llvm-objdump is not where the ROCm image keeps it: the scan cannot run.  The build FAILS on this unless
    ACNQP_SKIP_HAZARD_SCAN=1 is set (the scan is the only protection against the hazard)."""


def scan_library(path, spills=None):
    """Disassemble every gfx950 code object bundled in the shared library at ``path``; returns (stores, hazards).
    ``spills``: a list that receives scan_exec_spills' findings for the same disassembly."""
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
            if spills is not None:
                spills += scan_exec_spills(d.stdout)
    return stores, found


def stamp_path(lib):
    return lib + ".hazard_scan.json"


def write_stamp(lib, state, stores=0, unguarded=0):
    """Record next to the library what the build's scan saw: {"state": "scanned" | "skipped", ...}; a test asserts the
    library it loads was scanned with no finding (tests/test_store_hazard.py)."""
    import json

    with open(stamp_path(lib), "w") as f:
        json.dump({"state": state, "wide_stores": stores, "unguarded": unguarded, "library_sha256": file_sha256(lib)}, f)


def file_sha256(path):
    import hashlib

    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def read_stamp(lib):
    import json

    try:
        with open(stamp_path(lib)) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def describe(found, limit=6):
    out = []
    for k, l, n, t in found[:limit]:
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip() or k
        out.append(f"  {name[:100]}\n    {l}\n    +{n}: {t}")
    return "\n".join(out)
