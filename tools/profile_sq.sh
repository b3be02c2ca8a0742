#!/bin/bash
# Instruction-mix / wait counters of one program (two PMC passes): tools/profile_sq.sh <tag> <program args...>
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq1 -- python3 "$@" > $OUT/out_sq1.json 2> $OUT/sq1.err
echo "sq1 rc=$?"
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq2 -- python3 "$@" > $OUT/out_sq2.json 2> $OUT/sq2.err
echo "sq2 rc=$?"
tail -2 $OUT/sq2.err
