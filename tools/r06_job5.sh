#!/bin/bash
# round 6, job 5: late weight gradients - the split divisor (4 / 8 / 16) at batch 64 / 128 / 256 / 512, and late on / off at batch 64; the guide probe
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06e; mkdir -p $O
for r in 1 2; do
for v in 4 8 16; do
MPDX_WGRAD_LATE_DIV=$v python -c "
import bench
out = []
for (B, D, steps) in ((64, 4, 100), (128, 14, 100), (256, 14, 60), (512, 14, 40)):
    out.append(bench.training_leg(steps=steps, B=B, D=D, baseline=False)['ms_per_train_step'])
print('DIV=$v', *out)
" 2>/dev/null | tail -1
done
done | tee $O/train_late_div_ab2.txt
for r in 1 2; do
for v in 0 1; do
MPDX_TRAIN_WGRAD_LATE=$v MPDX_WGRAD_LATE_DIV=4 python -c "
import bench
print('LATE=$v (div 4) batch 64 / 96:', bench.training_leg(steps=100, B=64, D=4, baseline=False)['ms_per_train_step'], bench.training_leg(steps=100, B=96, D=14, baseline=False)['ms_per_train_step'])
" 2>/dev/null | tail -1
done
done | tee $O/train_late_b64_ab.txt
timeout 300 python tools/guide_inplan_probe.py 6400 2>/dev/null | tee $O/guide_inplan_probe.txt
timeout 300 python tools/guide_inplan_probe.py 100 2>/dev/null | tee -a $O/guide_inplan_probe.txt
