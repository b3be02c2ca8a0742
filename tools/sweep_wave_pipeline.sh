#!/bin/bash
# Sweep of the pipelined host entry on the headline workload: chunk size x persistent grid per launch (x ramp).
#   tools/sweep_wave_pipeline.sh "<chunks>" "<grids>" [extra env assignments]
mkdir -p gpurun_out/sweep
for c in ${1:-"2048 4096"}; do for g in ${2:-"128 192 256"}; do
  env $3 ACNQP_CHUNK=$c ACNQP_WAVE_GRID=$g timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 3 > gpurun_out/sweep/b_${c}_$g.json 2> gpurun_out/sweep/b_${c}_$g.err || exit 1
  echo "chunk $c grid $g $3: $(python3 tools/show_bench.py gpurun_out/sweep/b_${c}_$g.json 2>&1 | head -1 | cut -c1-40)"
done; done
