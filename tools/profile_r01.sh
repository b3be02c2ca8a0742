#!/bin/bash
# Round-1 profiling recipe (run on the GPU box through gpurun):
#   kernel trace + stats, then separate PMC passes (never combined with other traces).
set -o pipefail
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/prof_${1:-r01}
ARGS="--steps 16 --warmup 8 --no-cpu-baseline"
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
echo "trace rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err
echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py $ARGS > $OUT/bench_write.json 2> $OUT/write.err
echo "write rc=$?"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq1 -- python3 bench.py $ARGS > $OUT/bench_sq1.json 2> $OUT/sq1.err
echo "sq1 rc=$?"
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/pmc_sq2 -- python3 bench.py $ARGS > $OUT/bench_sq2.json 2> $OUT/sq2.err
echo "sq2 rc=$?"
find $OUT -name "*.csv" | head -40
