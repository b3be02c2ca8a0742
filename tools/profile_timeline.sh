#!/bin/bash
# Kernel + memory-copy trace of a short bench run (no counters): the timeline of one pipelined step.
#   tools/profile_timeline.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/timeline.txt (tools/timeline.py)
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline --no-other-configs --steps 3 --warmup 2 "$@" > $OUT/out.json 2> $OUT/trace.err
echo "trace rc=$?"
python3 tools/timeline.py $OUT/trace > $OUT/timeline.txt 2>&1
# keep the merged-back scratch small
find $OUT/trace -name "*.csv" -size +8M -delete
tail -n 80 $OUT/timeline.txt
