"""Dev tool: soak the kernels on the randomised problem classes of tests/test_gpu_parity.py (two session slots,
ragged horizons, peak vectors, equality rows, min rates, two sites) at a larger batch, checking statuses,
finiteness and the structural invariants of every solved schedule."""
import sys
sys.path.insert(0, '.')
import numpy as np
from adacharge_amd import ObjectiveComponent, equal_share, quick_charge, sites, tou_energy_cost, total_energy
from adacharge_amd.acn import Interface
from adacharge_amd.backend import SiteHandle, default_options
from adacharge_amd.builder import build_batch
from tests.test_gpu_parity import _random_sessions_general

B0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 0     # second argument: seed offset (other instances of the same classes)
POLISH = int(sys.argv[3]) if len(sys.argv) > 3 else 800  # third: polish_iters (a low value soaks the polish: more hand-overs)
LAST = int(sys.argv[4]) if len(sys.argv) > 4 else 20     # fourth: cases 0 .. LAST - 1
bad_total = 0
# cases 8-13: the long-horizon kernel (horizons 40 ... 200, two session slots, peaks, equality rows) and its
# LDS-resident variant (jpl52 at horizons 24 / 30); cases 14-19: the large-site kernel (128 / 192 / 512 EVSE, horizons
# 12 ... 48, one / two / three column tiles, two session slots, equality rows, peaks)
WIDE = {14: ("wide128", 12), 15: ("wide128", 30), 16: ("wide192", 48), 17: ("synth512", 24), 18: ("synth512", 48), 19: ("wide192", 16)}
for case in range(LAST):
    rng = np.random.default_rng(5000 + case + 100 * SEED)
    B = B0
    T = [12, 16, 24, 30, 12, 20, 9, 32, 40, 72, 144, 200, 24, 30, 12, 30, 48, 24, 48, 16][case]
    ct = ["SOC", "LINEAR"][case % 2]
    eq = case in (2, 5, 9, 15, 18); two = case in (1, 3, 5, 7, 8, 10, 13, 16, 19); with_peak = case in (0, 3, 4, 7, 9, 11, 12, 17)
    infra = sites.caltech54() if case not in (6, 10, 12, 13) else sites.jpl52()
    if case in WIDE:
        infra = getattr(sites, WIDE[case][0])()
        B = max(32, B0 // (4 if infra.num_stations <= 192 else 16))
    if 8 <= case < 14 and T > 32:
        B = max(32, B0 // 16)
    iface = Interface({"infrastructure_info": infra, "period": 5, "prices": rng.uniform(0.05, 0.4, size=256)})
    obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 10.0 ** rng.uniform(-4, -2)),
           ObjectiveComponent(tou_energy_cost, float(rng.uniform(0, 5))), ObjectiveComponent(total_energy, float(rng.uniform(0, 2)))]
    snaps, peaks = [], []
    for _ in range(B):
        snaps.append(_random_sessions_general(infra, T, rng, two, min_rates=not eq, demand_scale=(0.5 if eq else 1.5) * (0.3 if case in WIDE else 1.0)))
        pscale = infra.num_stations / 54.0 * (3.0 if case in WIDE else 1.0)   # wide sites: sessions on most EVSEs, minimum rates
        peaks.append((float(rng.uniform(250, 600) * pscale) if rng.random() < 0.5 else rng.uniform(250, 600, size=T) * pscale) if with_peak else None)
    Ts = [max(s.arrival_offset + s.remaining_time for s in sl) for sl in snaps]
    peaks = [p if (p is None or np.isscalar(p)) else p[:t] for p, t in zip(peaks, Ts)]
    batch = build_batch(snaps, infra, iface, obj, ct, eq, peak_limits=peaks)
    h = SiteHandle(batch.site, 0)
    r = h.solve(batch, default_options(max_iter=30000, polish_iters=POLISH))
    ps = h.polish_stats()
    pdiff = -1.0
    if ps["attempted"]:   # the polish took problems: the same batch without it must give the same schedules (to tolerance)
        r0 = h.solve(batch, default_options(max_iter=30000, polish_iters=0))
        both = np.isin(r.status, (1,)) & np.isin(r0.status, (1,))
        pdiff = float(np.abs(r.x[both] - r0.x[both]).max()) if both.any() else 0.0
        bad_total += int(pdiff > 1e-4 * 32) + int((r.status != r0.status).sum() > 0 and not np.isin(r0.status[r.status != r0.status], (2, 5)).all())
    ok = np.isin(r.status, (1, 5))
    finite = np.isfinite(r.x).all()
    x = r.x[ok]
    box = (x <= batch.ub[ok] + 1e-9).all() and (x >= batch.lb[ok] - 1e-9).all()
    ph = np.deg2rad(infra.phases); cm = infra.constraint_matrix
    if ct == "SOC":
        mag = np.hypot(np.einsum("mn,bnt->bmt", cm * np.cos(ph), x), np.einsum("mn,bnt->bmt", cm * np.sin(ph), x))
    else:
        mag = np.einsum("mn,bnt->bmt", np.abs(cm), x)
    viol = float((mag - infra.constraint_limits[None, :, None]).max()) if len(x) else 0.0
    pk = float((x.sum(axis=1) - batch.peak[ok]).max()) if with_peak and len(x) else 0.0
    counts = np.bincount(r.status, minlength=6)[1:]
    print(f"case {case} {ct} T={T} eq={eq} K={batch.K} peak={with_peak}: status counts(1..5) {counts.tolist()} finite {finite} box {box} "
          f"max row violation {viol:.2e} max peak violation {pk:.2e} iters mean {r.iters.mean():.0f} max {r.iters.max()} kernel {r.kernel_ms:.1f} ms "
          f"polish {ps['attempted']}/{ps['solved']} gave up {ps['gave_up_rows']}/{ps['gave_up_pivot']}/{ps['gave_up_rounds']}/{ps['gave_up_kkt']} "
          f"max |x - x(no polish)| {pdiff:.2e}", flush=True)
    bad_total += int(not finite) + int(not box) + int(viol > 5e-3) + int(pk > 5e-3)
    h.close()
print("anomalies:", bad_total)
