#!/bin/bash
# Round-3 profiling recipe (run on the GPU box through gpurun):  tools/profile_r03.sh <tag> <program args...>
#   pass 1: kernel trace + stats; passes 2 and 3: FETCH_SIZE, WRITE_SIZE (PMC passes never combined with traces);
#   pass 4 (optional, PMC_SQ=1): instruction / wait counters.
set -o pipefail
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 "$@" > $OUT/out_trace.json 2> $OUT/trace.err
echo "trace rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 "$@" > $OUT/out_fetch.json 2> $OUT/fetch.err
echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 "$@" > $OUT/out_write.json 2> $OUT/write.err
echo "write rc=$?"
if [ "$PMC_SQ" = "1" ]; then
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq1 -- python3 "$@" > $OUT/out_sq1.json 2> $OUT/sq1.err
echo "sq1 rc=$?"
fi
cat $OUT/out_trace.json | tail -1 | cut -c1-600
