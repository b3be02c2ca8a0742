"""Per-phase s_memtime shares of the wave-per-problem kernel (diagnostic build -DACNQP_STAMPS:
build_hip_library(extra_flags=["-DACNQP_STAMPS"], out="adacharge_amd/lib/libacn_qp_hip_stamps.so")).  Dev tool."""
import sys, os, ctypes as C
sys.path.insert(0, '.')
os.environ["ACNQP_LIBRARY"] = os.path.abspath("adacharge_amd/lib/libacn_qp_hip_stamps.so")
import numpy as np
from adacharge_amd import ObjectiveComponent, equal_share, quick_charge, sites
from adacharge_amd.backend import SiteHandle, default_options, load_library
from adacharge_amd.builder import build_batch
from tests import helpers as H
infra, iface = H.caltech_interface()
obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-12)]
names = ["r0 + w^ (4 mfma)", "EVSE->rows 16 mfma", "e^ h^ Qh^ (4 mfma)", "rows->EVSE 16 mfma", "anderson event", "site projection", "water-filling + y1", "check", "(passes x1000)"]
SITE = os.environ.get("STAMP_SITE")   # e.g. 3: eight_sites()[3] (two row tiles: the two-waves-per-problem variant)
T = int(os.environ.get("STAMP_T", "12"))
if SITE is not None:
    from adacharge_amd.acn import Interface
    infra = sites.eight_sites()[int(SITE)]
    iface = Interface({"infrastructure_info": infra, "period": 5})
for NB in [int(a) for a in sys.argv[1:]] or [256]:
    snaps = sites.snapshot_batch(infra, T, NB, seed=20240)
    batch = build_batch(snaps, infra, iface, obj, "SOC")
    h = SiteHandle(batch.site, 0)
    res = h.solve(batch, default_options(polish_iters=0))
    lib = load_library()
    buf = (C.c_ulonglong * (1024 * 16 * 12))()
    lib.acnqp_debug_read_wave_stamps(buf, 1024 * 16 * 12)
    nb = min(NB, 1024)
    st = np.array(buf, dtype=np.float64).reshape(1024, 16 * 12)[:nb, :24]
    per_iter = st / res.iters[:nb, None]
    print("B", NB, "kernel_ms %.2f" % res.kernel_ms, "iters mean %.1f max %d" % (res.iters.mean(), res.iters.max()))
    tot = per_iter[:, :8].sum(-1).mean()
    for k, n in enumerate(names):
        print("   %-22s %8.0f ticks/iter  %.1f%%" % (n, per_iter[:, k].mean(), 100 * per_iter[:, k].mean() / tot))
    for k, n in ((12, "aa: f, column, stores"), (13, "aa: dots, wave sums"), (14, "aa: bookkeeping, solve"), (15, "aa: correction")):
        print("   %-22s %8.0f ticks/iter" % (n, per_iter[:, k].mean()))
    print("   total %.0f s_memtime ticks/iter; s_memtime rate %.0f MHz; us/iter %.2f" % (tot, 100.0 * st[:, 10].sum() / st[:, 9].sum(), (st[:, 9] / 100.0 / res.iters[:nb]).mean()))
