#!/bin/bash
# gpurun job: the backward programs on the THREE-level network (dim_mults option 0): training tests, then A/B programs off / on at batch 32 / 128 / 512
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_train.py -x -q 2>&1 | tail -6
for r in 1 2 3; do
  for v in 0 1; do
    MPDX_TRAIN_BWD_PROG=$v timeout 600 python -c "
import bench
a = bench.training_leg(steps=100, opt=0, baseline=False); b = bench.training_leg(steps=100, B=128, D=14, opt=0, baseline=False); c = bench.training_leg(steps=40, B=512, D=14, opt=0, baseline=False)
print('opt0 PROG=$v', a['ms_per_train_step'], b['ms_per_train_step'], c['ms_per_train_step'])
" 2>/dev/null | tail -1
  done
done
