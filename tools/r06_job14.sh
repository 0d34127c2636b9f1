#!/bin/bash
# round 6, job 14b: batch 16 / 32 / 48 with the backward programs: late weight gradients with divisor 2 / 4
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06l; mkdir -p $O
for r in 1 2 3; do
for v in "MPDX_TRAIN_WGRAD_LATE=0" "MPDX_TRAIN_WGRAD_LATE=1 MPDX_WGRAD_LATE_DIV=2" "MPDX_TRAIN_WGRAD_LATE=1 MPDX_WGRAD_LATE_DIV=4"; do
env $v python -c "
import bench
print('$v', bench.training_leg(steps=100, B=16, D=4, baseline=False)['ms_per_train_step'], bench.training_leg(steps=150, baseline=False)['ms_per_train_step'], bench.training_leg(steps=100, B=48, D=4, baseline=False)['ms_per_train_step'])
" 2>/dev/null | tail -1
done; done | tee $O/train_b32_late_ab2.txt
