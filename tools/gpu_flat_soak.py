"""Soak of the round-4 byte savers of the streaming kernels (flat items / flat tiles, padded register pairs): the same
random batches -- bounds that compress, bounds that do not (per-period derates, minimum rates, several sessions per
EVSE), mixtures -- solved by the production library and by a diagnostic build without the savers
(-DACNQP_STREAM_FLAT_BOUNDS=0 -DACNQP_LONG_FLAT_BOUNDS=0 -DACNQP_LONG_PAD_SKIP=0) must agree BIT FOR BIT: schedules,
iteration counts, statuses, residuals, objectives.

    python tools/gpu_flat_soak.py child <out.npz> [seeds]     one library (ACNQP_LIBRARY), all cases
    python tools/gpu_flat_soak.py <plain.so> [seeds]          both, compared"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def cases(seed):
    from adacharge_amd import ObjectiveComponent, equal_share, load_flattening, quick_charge, sites
    from adacharge_amd.acn import Interface
    from adacharge_amd.builder import build_batch

    rng = np.random.default_rng(seed)
    out = []
    for name, site, T, B, general in (("long54x144", "caltech54", 144, 6, False), ("long54x96k", "caltech54", 96, 4, True),
                                      ("long52x64", "jpl52", 64, 4, False), ("stream192x48", None, 48, 4, False),
                                      ("stream128x40k", None, 40, 3, True)):
        if site is None:
            infra = sites.wide192() if "192" in name else sites.wide128()
        else:
            infra = getattr(sites, site)()
        iface = Interface({"infrastructure_info": infra, "period": 5})
        obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-3)]
        if general:
            snaps = [sites.random_sessions_general(infra, T, rng, True, min_rates=bool(k % 2), demand_scale=1.0) for k in range(B)]
        else:
            snaps = sites.snapshot_batch(infra, T, B, seed=int(rng.integers(1 << 30)), demand_range=(5.0, 40.0))
        batch = build_batch(snaps, infra, iface, obj, "SOC")
        tt = np.arange(batch.ub.shape[2])
        N = batch.ub.shape[1]
        mode = int(rng.integers(4))
        if mode == 1:      # per-period derates on a random subset of EVSEs: nothing of theirs compresses
            rows = rng.random(N) < 0.3
            batch.ub[:, rows, :] *= np.where(tt % int(rng.integers(2, 6)) == 0, 0.75, 1.0)
        elif mode == 2:    # minimum rates in some periods
            rows = rng.random(N) < 0.2
            batch.lb[:, rows, :] = np.where(batch.ub[:, rows, :] > 0, 1.0 * (tt % 4 == 1), 0.0)
        elif mode == 3:    # one lane of one row
            i, t0 = int(rng.integers(N)), int(rng.integers(16))
            batch.ub[:, i, t0::16] *= 0.5 + 0.5 * (np.arange(len(tt[t0::16])) % 2)
        out.append((f"{name}_s{seed}_m{mode}", batch))
    return out


def child(path, seeds):
    from adacharge_amd.backend import SiteHandle, default_options

    res = {}
    for seed in seeds:
        for name, batch in cases(seed):
            h = SiteHandle(batch.site, 0)
            r = h.solve(batch, default_options())
            res[name + "_x"] = r.x; res[name + "_it"] = r.iters; res[name + "_st"] = r.status
            res[name + "_pri"] = r.pri_res; res[name + "_dua"] = r.dua_res; res[name + "_obj"] = r.obj
            h.close()
    np.savez(path, **res)


if __name__ == "__main__":
    if sys.argv[1] == "child":
        child(sys.argv[2], [int(s) for s in sys.argv[3:]] or [0])
        raise SystemExit(0)
    plain = os.path.abspath(sys.argv[1])
    seeds = sys.argv[2:] or ["0", "1", "2"]
    outdir = os.path.join(ROOT, "gpurun_out", "flat_soak"); os.makedirs(outdir, exist_ok=True)
    files = []
    for tag, lib in (("prod", None), ("plain", plain)):
        env = dict(os.environ)
        if lib: env["ACNQP_LIBRARY"] = lib
        f = os.path.join(outdir, tag + ".npz"); files.append(f)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child", f, *seeds], check=True, env=env)
    a, b = np.load(files[0]), np.load(files[1])
    bad = 0
    for k in a.files:
        same = np.array_equal(a[k], b[k])
        if not same:
            bad += 1
            print("DIFFERS", k, float(np.abs(a[k].astype(float) - b[k].astype(float)).max()))
    names = sorted(set(k.rsplit("_", 1)[0] for k in a.files))
    print(len(names), "cases,", len(a.files), "arrays,", bad, "differ;", "statuses:", {n: a[n + "_st"].tolist() for n in names[:40]})
    raise SystemExit(1 if bad else 0)
