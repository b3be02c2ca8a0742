"""Turn the rocprofv3 output of tools/profile_r02.sh (gpurun_out/prof_<tag>/) into the committed
summaries under profiles/: <tag>_kernel_stats.csv, <tag>_pmc_summary.json, <traffic file>.

    python tools/summarise_profile.py r02a_bench admm_tiled_kernel hbm_traffic.json
    python tools/summarise_profile.py r02a_cfg5 admm_stream_kernel r02a_cfg5_traffic.json
    python tools/summarise_profile.py r04w_bench admm_wave_kernel hbm_traffic.json 4096 4   (four problems per workgroup)
"""
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
KERNEL = sys.argv[2] if len(sys.argv) > 2 else "admm_tiled_kernel"
TRAFFIC_FILE = sys.argv[3] if len(sys.argv) > 3 else "hbm_traffic.json"
MIN_PROBLEMS = int(sys.argv[4]) if len(sys.argv) > 4 else 0   # skip dispatches of fewer problems (round 4: the resume launch
                                                              # behind the polish is the same kernel on a handful of problems)
PER_WG = int(sys.argv[5]) if len(sys.argv) > 5 else 1         # problems per workgroup of the static schedule (4: acn_qp_wave.hpp)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")

stats = sorted(glob.glob(os.path.join(src, "trace", "**", "*_kernel_stats.csv"), recursive=True), key=os.path.getmtime)
if stats:   # newest run only (gpurun merges, it does not delete)
    shutil.copy(stats[-1], os.path.join(dst, f"{tag}_kernel_stats.csv"))
summary = {}
meta = None
newest = {}
problems = {}
for f in glob.glob(os.path.join(src, "pmc_*", "**", "*_counter_collection.csv"), recursive=True):
    d_ = os.path.dirname(f)
    if d_ not in newest or os.path.getmtime(f) > os.path.getmtime(newest[d_]):
        newest[d_] = f
for f in sorted(newest.values()):
    per = {}
    for row in csv.DictReader(open(f)):
        if KERNEL not in row["Kernel_Name"]:
            continue
        if MIN_PROBLEMS and PER_WG * int(row["Grid_Size"]) // max(int(row["Workgroup_Size"]), 1) < MIN_PROBLEMS:
            continue
        per.setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
        per[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
        try:   # one workgroup per problem: problems of the dispatch = grid / workgroup size
            problems.setdefault(row["Counter_Name"], {})[row["Dispatch_Id"]] = PER_WG * int(row["Grid_Size"]) // int(row["Workgroup_Size"])
        except (KeyError, ValueError, ZeroDivisionError):
            pass
        if meta is None:
            meta = {k: row[k] for k in ("Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count",
                                        "Accum_VGPR_Count", "SGPR_Count") if k in row}
    for name, d in per.items():
        v = list(d.values())
        summary[name] = {"dispatches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)}
        if name in problems and len(problems[name]) == len(v):
            summary[name]["problems"] = sum(problems[name].values())
            summary[name]["per_problem"] = sum(v) / max(summary[name]["problems"], 1)
summary["_kernel"] = meta
summary["_command"] = "rocprofv3 --pmc <counter> --output-format csv -- python3 <program> (one pass per counter group, tools/profile_r02.sh)"
json.dump(summary, open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w"), indent=1)
if "FETCH_SIZE" in summary and "WRITE_SIZE" in summary:
    fk, wk = summary["FETCH_SIZE"]["mean"], summary["WRITE_SIZE"]["mean"]
    json.dump({
        "round": tag,
        "FETCH_SIZE_KB_per_launch": fk, "WRITE_SIZE_KB_per_launch": wk,
        "correction": "MI355X_MICROARCH.md section HBM: on gfx950 FETCH_SIZE reports 1/2 of the bytes of a coalesced "
                      "streaming read -> doubled; WRITE_SIZE exact; KB = 1024 B",
        "hbm_bytes_per_launch": (2 * fk + wk) * 1024,
        "raw_uncorrected_bytes_per_launch": (fk + wk) * 1024,
        # launches differ in size (ramped chunks, the kernel-only launch): the per-problem figure is the one to scale
        "hbm_bytes_per_problem": ((2 * summary["FETCH_SIZE"]["per_problem"] + summary["WRITE_SIZE"]["per_problem"]) * 1024
                                  if "per_problem" in summary["FETCH_SIZE"] and "per_problem" in summary["WRITE_SIZE"] else None),
    }, open(os.path.join(dst, TRAFFIC_FILE), "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if not k.startswith("_")}, indent=1)[:3000])
