#!/bin/bash
# Final headline profile of round 3 (final build: launch order, branch-free projection, scalar wave index, resident fragments): trace + HBM traffic + both SQ passes of the bench.
set -o pipefail
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
PMC_SQ=1 bash tools/profile_r03.sh r03j_bench bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-other-configs || exit 1
OUT=gpurun_out/prof_r03j_bench
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq2 -- python3 bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-other-configs > $OUT/out_sq2.json 2> $OUT/sq2.err || exit 1
echo "sq2 rc=$?"
