#!/bin/bash
# gpurun job: A/B of libraries on the training iteration (batch 32 / 128 / 512), interleaved rounds on one box:  tools/ab_train_libs3.sh "lib1 lib2 ..." [rounds]
cd $GRAFT_REPO_ROOT
for r in $(seq 1 ${2:-3}); do
  for L in $1; do
    MPDX_LIB=$L python -c "
import bench
a = bench.training_leg(steps=100, baseline=False); b = bench.training_leg(steps=100, B=128, D=14, baseline=False); c = bench.training_leg(steps=40, B=512, D=14, baseline=False)
print('$L', a['ms_per_train_step'], b['ms_per_train_step'], c['ms_per_train_step'])
" 2>/dev/null | tail -1
  done
done
