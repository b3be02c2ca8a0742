mkdir -p gpurun_out/r4x14
for v in 0 1 100; do
  echo "== EARLY_HANDOVER=$v" >> gpurun_out/r4x14/eh.log
  for c in cfg3-site3 cfg3-site0 cfg2-caltech; do
    if [ $v = 0 ]; then python3 tools/run_config.py $c 2>&1 | cut -c1-420 >> gpurun_out/r4x14/eh.log; else ACNQP_EARLY_HANDOVER=$v python3 tools/run_config.py $c 2>&1 | cut -c1-420 >> gpurun_out/r4x14/eh.log; fi
  done
  if [ $v = 0 ]; then python3 tools/profile_single_step.py 300 2>&1 | grep median >> gpurun_out/r4x14/eh.log; else ACNQP_EARLY_HANDOVER=$v python3 tools/profile_single_step.py 300 2>&1 | grep median >> gpurun_out/r4x14/eh.log; fi
done
grep -v amdgpu.ids gpurun_out/r4x14/eh.log | sed 's/"note": "[^"]*", //'
