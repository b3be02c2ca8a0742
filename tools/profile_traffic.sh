#!/bin/bash
# HBM traffic only (two separate PMC passes), for quick before/after checks: gpurun_out/prof_<tag>/pmc_{fetch,write}
set -o pipefail
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/prof_${1:-traffic}
ARGS="--steps 16 --warmup 8 --no-cpu-baseline"
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err
echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py $ARGS > $OUT/bench_write.json 2> $OUT/write.err
echo "write rc=$?"
