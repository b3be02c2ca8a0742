"""Scan a built library for the gfx950 store-data hazard (adacharge_amd/store_hazard.py).

    python tools/check_store_hazard.py [path/to/lib.so]          (default: the production library)

To see the pattern the guard removes, build a diagnostic library without it and scan that:
    python -c "from adacharge_amd.build import build_hip_library as b; b(extra_flags=['-DACNQP_LONG_STORE_NOP=0'], out='/tmp/nonop.so', check_hazards=False)"
    python tools/check_store_hazard.py /tmp/nonop.so
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adacharge_amd import store_hazard  # noqa: E402

path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "adacharge_amd", "lib", "libacn_qp_hip.so")
spills = []
stores, found = store_hazard.scan_library(path, spills)
print(f"{path}: {stores} buffer stores of more than 64 bits per lane, {len(found)} followed within "
      f"{store_hazard.WAIT_STATES} wait states by a VALU write of their data registers")
if found:
    print(store_hazard.describe(found))
print(f"{len(spills)} register spills / reloads ahead of an exec restore (store_hazard.scan_exec_spills)")
for k, l in spills[:8]:
    print(f"  {k[:100]}\n    {l}")
raise SystemExit(1 if found or spills else 0)
