#!/bin/bash
# A/B of the work queue on the legs that stream their state (tools/run_config.py): default (queue, workspace per
# workgroup slot) vs ACNQP_WS_PER_PROBLEM=1 (queue, workspace per problem) vs ACNQP_NO_QUEUE=1 (static
# one-workgroup-per-problem schedule), two rounds, interleaved.  Output: <dir>/queue_ab.log
out=${1:-gpurun_out/queue_ab}
legs=${2:-"stress144-2k cfg5"}
mkdir -p $out
for round in 1 2; do
for leg in $legs; do
  for mode in queue wsprob static; do
    unset ACNQP_NO_QUEUE ACNQP_WS_PER_PROBLEM
    [ $mode = static ] && export ACNQP_NO_QUEUE=1
    [ $mode = wsprob ] && export ACNQP_WS_PER_PROBLEM=1
    echo "== $leg $mode" >> $out/queue_ab.log
    timeout -k 10 300 python3 tools/run_config.py $leg >> $out/queue_ab.log 2>/dev/null || exit 1
  done
done
done
