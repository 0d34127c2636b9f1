#!/bin/bash
# gpurun job: a training-step dev switch (here MPDX_TRAIN_LOSS_RIDE: column sums as side blocks of the reduction launch) - training tests, then A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/s3; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_trained.py -m gpu -x -q 2>&1 | grep -v "amdgpu.ids" | tail -3 | tee $O/train_tests_lride.txt
for r in 1 2 3; do
  for v in 0 1; do
    MPDX_TRAIN_LOSS_RIDE=$v python -c "
import bench
a = bench.training_leg(steps=100, baseline=False); b = bench.training_leg(steps=60, B=128, D=14, baseline=False); c = bench.training_leg(steps=20, B=512, D=14, baseline=False)
print('lride=$v', a['ms_per_train_step'], b['ms_per_train_step'], b['roofline']['frac'], c['ms_per_train_step'])
" 2>/dev/null | tail -1
  done
done | tee $O/ab_loss_ride.txt
