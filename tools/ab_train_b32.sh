#!/bin/bash
# gpurun job: A/B of environment settings on the batch-32 training iteration (ms), interleaved rounds:  tools/ab_train_b32.sh "A=1 B=2" "A=0" ... (each argument one setting)
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  for s in "$@"; do
    env $s python -c "
import bench
a = bench.training_leg(steps=200, baseline=False)
print('$s', a['ms_per_train_step'])
" 2>/dev/null | tail -1
  done
done
