"""Long-horizon kernel: r0 / zh in LDS (default where it fits) against the workspace placement (ACNQP_NO_RZL=1) on
LP-like 54 x T problems: statuses, iterations, max |dx|.   python tools/gpu_rzl_check.py [batch]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import numpy as np
    from adacharge_amd import ObjectiveComponent, equal_share, quick_charge, sites
    from adacharge_amd.acn import Interface
    from adacharge_amd.backend import SiteHandle, default_options
    from adacharge_amd.builder import build_batch
    B = int(sys.argv[3])
    infra = sites.caltech54(); iface = Interface({"infrastructure_info": infra, "period": 5})
    obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-12)]
    out = {}
    for T in (96, 144):
        batch = build_batch(sites.snapshot_batch(infra, T, B, seed=100 + T, demand_range=(5.0, 60.0)), infra, iface, obj, "SOC")
        h = SiteHandle(batch.site, 0)
        r = h.solve(batch, default_options()); r0 = h.solve(batch, default_options(retry_passes=0)); h.close()
        out["x%d" % T] = r.x; out["it%d" % T] = r.iters; out["st%d" % T] = r.status; out["it0_%d" % T] = r0.iters; out["st0_%d" % T] = r0.status
    np.savez(sys.argv[2], **out)
    raise SystemExit(0)
import numpy as np
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
res = {}
for tag, env in (("lds", {}), ("ws", {"ACNQP_NO_RZL": "1"})):
    f = "/tmp/rzl_%s.npz" % tag
    subprocess.run([sys.executable, os.path.abspath(__file__), "child", f, str(B)], check=True, env=dict(os.environ, **env))
    res[tag] = np.load(f)
for T in (96, 144):
    a, b = res["lds"], res["ws"]
    print("T=%d  max|dx| %.2e" % (T, np.abs(a["x%d" % T] - b["x%d" % T]).max()))
    for tag, r in (("lds", a), ("ws ", b)):
        print("  %s status %s  iters %s\n      single pass: status %s iters %s" % (tag, r["st%d" % T].tolist(), r["it%d" % T].tolist(), r["st0_%d" % T].tolist(), r["it0_%d" % T].tolist()))
