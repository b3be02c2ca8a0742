#!/bin/bash
# Round-4 profiling recipe (run on the GPU box through gpurun):  tools/profile_r04.sh <tag> <program args...>
#   pass 1: kernel trace + stats of the program AS IT RUNS (work queue, polish);
#   passes 2-5: PMC passes (never combined with traces): FETCH_SIZE, WRITE_SIZE, two SQ groups -- with ACNQP_NO_QUEUE=1, the
#   static schedule (one workgroup per problem), so that a dispatch's grid tells its problems and SQ_WAVES its
#   problem-waves: tools/summarise_profile.py and tools/sq_summary.py divide by them.  Traffic and instruction counts per
#   problem do not depend on which workgroup solves it.
set -o pipefail
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 "$@" > $OUT/out_trace.json 2> $OUT/trace.err
echo "trace rc=$?"
export ACNQP_NO_QUEUE=1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 "$@" > $OUT/out_fetch.json 2> $OUT/fetch.err
echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 "$@" > $OUT/out_write.json 2> $OUT/write.err
echo "write rc=$?"
if [ "$PMC_SQ" = "1" ]; then
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq1 -- python3 "$@" > $OUT/out_sq1.json 2> $OUT/sq1.err
echo "sq1 rc=$?"
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq2 -- python3 "$@" > $OUT/out_sq2.json 2> $OUT/sq2.err
echo "sq2 rc=$?"
fi
tail -1 $OUT/out_trace.json | cut -c1-300
