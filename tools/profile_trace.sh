#!/bin/bash
# Kernel trace + stats only:  tools/profile_trace.sh <tag> <program args...>
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 "$@" > $OUT/out_trace.json 2> $OUT/trace.err
echo "trace rc=$?"; tail -1 $OUT/out_trace.json | cut -c1-400
