#!/bin/bash
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/prof_calib
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 tools/pmc_calibrate.py > $OUT/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 tools/pmc_calibrate.py > $OUT/w.log 2>&1
python3 - <<'PY'
import csv, glob
for kind in ("fetch", "write"):
    for f in glob.glob(f"gpurun_out/prof_calib/{kind}/**/*counter_collection.csv", recursive=True):
        agg = {}
        for row in csv.DictReader(open(f)):
            k = (row["Dispatch_Id"], row["Kernel_Name"][:70])
            agg[k] = agg.get(k, 0.0) + float(row["Counter_Value"])
        for (d, k), v in sorted(agg.items(), key=lambda kv: int(kv[0][0])):
            if v > 1e4: print(kind, d, k, "%.1f KB = %.3f GiB" % (v, v * 1024 / 2**30))
PY
