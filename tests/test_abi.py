"""The C-ABI library loads, exports every symbol include/acn_qp.h declares and
rejects misuse with return codes -- no compute calls (no GPU here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from adacharge_amd import backend

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "acn_qp.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(acnqp_[a-z_]+)\s*\(", text)))


def test_header_symbols_all_exported(hip_library):
    syms = declared_symbols()
    assert set(syms) == set(backend.EXPORTED_SYMBOLS)
    for s in syms:
        assert hasattr(hip_library, s), s


def test_abi_version_and_defaults(hip_library):
    assert hip_library.acnqp_abi_version() == 9
    o = backend.default_options()
    assert o.precision == 64 and 0 < o.alpha < 2 and o.max_iter > 0 and o.eps_abs > 0
    # ABI v7: stall rule and retry passes are options of the library (every entry point), not of the binding
    assert o.stall_iters == 3000 and o.retry_passes == 2 and o.retry_max_iter == 8000 and o.retry_rho == 0.5
    assert o.inaccurate_floor == 1e-5 and o.polish_iters == 800 and o.polish_stall == 0
    assert C.sizeof(backend.Options) == 120   # sizeof(acnqp_options) as gcc lays the header out
    o2 = backend.default_options(eps_abs=1e-9, max_iter=5)
    assert o2.eps_abs == 1e-9 and o2.max_iter == 5
    assert backend.default_options(polish_stall=100).polish_stall == 100      # ABI v9: the last field of the structure
    # the header's own layout (compiled here with gcc): the new field sits behind inaccurate_floor, where the binding has it
    import subprocess, tempfile
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, "o.c")
        with open(src, "w") as f:
            f.write('#include <stdio.h>\n#include <stddef.h>\n#include "acn_qp.h"\nint main(){printf("%zu %zu %zu", sizeof(acnqp_options), '
                    'offsetof(acnqp_options, polish_stall), offsetof(acnqp_options, inaccurate_floor));return 0;}')
        exe = os.path.join(tmp, "o")
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe], check=True)
        size, off_stall, off_floor = map(int, subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split())
    assert size == C.sizeof(backend.Options) and off_stall == backend.Options.polish_stall.offset and off_floor == backend.Options.inaccurate_floor.offset
    with pytest.raises(TypeError):
        backend.default_options(nonsense=1)


def test_create_rejects_bad_arguments(hip_library):
    h = C.c_void_p()
    assert hip_library.acnqp_create(None, 0, C.byref(h)) == -1
    assert b"null" in hip_library.acnqp_last_error()
    G = np.ones((1, 2000))
    lim = np.ones(1)
    desc = backend._Site(2000, 1, 1, 0, 0, 0, 0, G.ctypes.data_as(C.c_void_p), lim.ctypes.data_as(C.c_void_p))
    assert hip_library.acnqp_create(C.byref(desc), 0, C.byref(h)) == -1   # N > 1024
    desc = backend._Site(4, 1, 3, 0, 0, 0, 0, G.ctypes.data_as(C.c_void_p), lim.ctypes.data_as(C.c_void_p))
    assert hip_library.acnqp_create(C.byref(desc), 0, C.byref(h)) == -1   # n_rows inconsistent
    desc = backend._Site(4, 1, 1, 7, 0, 0, 0, G.ctypes.data_as(C.c_void_p), lim.ctypes.data_as(C.c_void_p))
    assert hip_library.acnqp_create(C.byref(desc), 0, C.byref(h)) == -1   # bad cone
    assert hip_library.acnqp_solve_batch(None, None, None, None) == -1
    assert hip_library.acnqp_last_kernel_ms(None) < 0
    assert hip_library.acnqp_accel_columns(None, 12, 1, 64, 10) == 0
    assert hip_library.acnqp_kernel_times(None, None, 0) == 0
    assert hip_library.acnqp_launch_count(None) == 0
    hip_library.acnqp_destroy(None)   # no-op


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setenv("ACNQP_LIBRARY", str(tmp_path / "nope.so"))
    monkeypatch.setattr(backend, "_lib", None)
    with pytest.raises(backend.BackendUnavailable, match="no CPU fallback"):
        backend.load_library()


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under adacharge_amd/ (nor
    bench.py's measured path) may import it."""
    pkg = os.path.join(ROOT, "adacharge_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "oracle/" not in src or f.endswith((".hpp", ".py")) and "import" not in src.split("oracle/")[0][-20:]
