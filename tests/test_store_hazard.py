"""The gfx950 store-data hazard: the scanner recognises the pattern, and the library the tests
run against has none (adacharge_amd/store_hazard.py; the build refuses a library that has one)."""
from adacharge_amd import backend, store_hazard

BAD = """
0000000000001000 <_ZN5acnqp16admm_long_kernelILi9ELi1ELi8ELb0ELb0ELb0EEEvNS_10StreamArgsE>:
	buffer_store_dwordx4 v[44:47], v1, s[56:59], s77 offen     // 000000001000: E07C1000 4D0E2C01
	v_add_f64 v[44:45], v[36:37], -v[68:69]                    // 000000001008: D2800000 4002892C
	s_endpgm
"""
GUARDED = BAD.replace("\tv_add_f64", "\ts_nop 2\n\tv_add_f64")
OTHER_REGS = BAD.replace("v_add_f64 v[44:45]", "v_add_f64 v[48:49]")
NARROW = BAD.replace("buffer_store_dwordx4 v[44:47]", "buffer_store_dwordx2 v[44:45]")


def test_scanner_finds_an_overwritten_wide_store_and_accepts_the_guarded_forms():
    stores, found = store_hazard.scan_isa(BAD)
    assert stores == 1 and len(found) == 1 and found[0][2] == 1 and "admm_long_kernel" in found[0][0]
    assert store_hazard.scan_isa(GUARDED) == (1, [])          # three wait states between store and overwrite
    assert store_hazard.scan_isa(OTHER_REGS) == (1, [])       # a VALU write to other registers
    assert store_hazard.scan_isa(NARROW) == (0, [])           # 64-bit stores do not have the hazard
    late = BAD.replace("\tv_add_f64", "\ts_mov_b32 s0, 0\n\ts_mov_b32 s1, 0\n\ts_mov_b32 s2, 0\n\tv_add_f64")
    assert store_hazard.scan_isa(late) == (1, [])


def test_built_library_has_wide_stores_and_none_is_unguarded(hip_library):
    try:
        stores, found = store_hazard.scan_library(backend.library_path())
    except store_hazard.ScannerUnavailable as e:   # no llvm-objdump in this image: nothing to check with
        import pytest
        pytest.skip(str(e))
    assert stores > 100, stores          # the long-horizon kernel's 16-byte accesses are there
    assert not found, store_hazard.describe(found)
