"""Register / scratch / LDS budget of every kernel instantiation, from the compiler's own remarks.

    python tools/kernel_resources.py [unit ...] [--md profiles/rNN_kernel_resources.md]

Compiles the given translation units of adacharge_amd/csrc (default: all that hold kernels) with
-Rpass-analysis=kernel-resource-usage (device pass only, objects thrown away) and prints one row per kernel.
"""
import concurrent.futures
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adacharge_amd.build import CSRC, FLAGS, INC, hipcc_path  # noqa: E402

KERNEL_UNITS = ("acn_qp_wave", "acn_qp_polish", "acn_qp_tiled_ct1", "acn_qp_tiled_ct2", "acn_qp_stream", "acn_qp_long", "acn_qp_general")
FIELDS = ("VGPRs", "AGPRs", "VGPRs Spill", "SGPRs Spill", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]")


def remarks(unit, extra=()):
    with tempfile.TemporaryDirectory() as tmp:
        cmd = [hipcc_path(), *FLAGS, *extra, "-I" + INC, "-I" + CSRC, "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage",
               "-c", os.path.join(CSRC, unit + ".hip"), "-o", os.path.join(tmp, "x.o")]
        out = subprocess.run(cmd, capture_output=True, text=True)
        if out.returncode != 0:
            raise SystemExit(out.stderr)
    rows, cur = [], None
    for line in out.stderr.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"^void acnqp::", "", name)
            name = re.sub(r"\(acnqp::\w+\)$", "", name)
            cur = {"kernel": name}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return rows


def main():
    args = [a for a in sys.argv[1:]]
    md = None
    if "--md" in args:
        md = args[args.index("--md") + 1]
        del args[args.index("--md"):args.index("--md") + 2]
    extra = [a for a in args if a.startswith("-")]
    units = [a for a in args if not a.startswith("-")] or list(KERNEL_UNITS)
    with concurrent.futures.ThreadPoolExecutor(min(len(units), 6)) as pool:
        rows = [r for rs in pool.map(lambda u: remarks(u, extra), units) for r in rs]
    head = "| kernel | VGPRs | AGPRs | VGPR spills | SGPR spills | scratch B/lane | waves/SIMD | LDS B |\n|---|---|---|---|---|---|---|---|\n"
    body = "".join("| `%s` | %s |\n" % (r["kernel"], " | ".join(str(r.get(f, "")) for f in FIELDS)) for r in sorted(rows, key=lambda r: r["kernel"]))
    print(head + body)
    if md:
        with open(md, "w") as f:
            f.write("# Kernel resource usage\n\n`python tools/kernel_resources.py` (hipcc -Rpass-analysis=kernel-resource-usage over the kernel "
                    "translation units of adacharge_amd/csrc, the flags of adacharge_amd/build.py).  Template arguments: tiled "
                    "`<real, NW, CT, MT, KS, OCC, AM>`, stream `<CT, MT, NWV>`, long `<CTL, MT, NWV[, LDS-resident, r0/zh in LDS, x in LDS]>`, "
                    "general `<real, threads>`.\n\n" + head + body)


if __name__ == "__main__":
    main()
