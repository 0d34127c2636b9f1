#!/bin/bash
# gpurun job: training tests + A/B of two libraries on the training iteration (batch 32 / 128 / 512), interleaved: build_ab/libmpdx_head.so (a build of the commit to compare with) vs the tree
cd $GRAFT_REPO_ROOT; O=gpurun_out/s3; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_parity.py -m gpu -x -q -k "train or loss or gradient or golden" 2>&1 | grep -v "amdgpu.ids" | tail -3 | tee $O/train_tests_ab.txt
for r in 1 2 3; do
  for L in build_ab/libmpdx_head.so mpd_public_amd/libmpdx.so; do
    MPDX_LIB=$L python -c "
import bench
a = bench.training_leg(steps=100, baseline=False); b = bench.training_leg(steps=100, B=128, D=14, baseline=False); c = bench.training_leg(steps=40, B=512, D=14, baseline=False)
print('$L', a['ms_per_train_step'], b['ms_per_train_step'], b['roofline']['frac'], c['ms_per_train_step'], c['roofline']['frac'])
" 2>/dev/null | tail -1
  done
done | tee $O/ab_train_libs.txt
