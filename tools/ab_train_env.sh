#!/bin/bash
# gpurun job: A/B of VALUES of one environment switch of the training step, interleaved rounds on one box.
#   tools/ab_train_env.sh VAR "v1 v2 v3" [rounds]
# prints ms per iteration at batch 32 x D=4 | batch 128 x D=14 | its fraction of the fp32 peak | batch 512 x D=14
cd $GRAFT_REPO_ROOT; SW=$1; VALS=$2; R=${3:-3}
for r in $(seq 1 $R); do
  for v in $VALS; do
    env $SW=$v python -c "
import bench
a = bench.training_leg(steps=100, baseline=False); b = bench.training_leg(steps=100, B=128, D=14, baseline=False); c = bench.training_leg(steps=40, B=512, D=14, baseline=False)
print('$SW=$v', a['ms_per_train_step'], b['ms_per_train_step'], b['roofline']['frac'], c['ms_per_train_step'])
" 2>/dev/null | tail -1
  done
done
