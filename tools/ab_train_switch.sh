#!/bin/bash
# gpurun job: A/B of ONE dev switch of the training step (read once per process: two processes per round), interleaved rounds on one box.
#   tools/ab_train_switch.sh MPDX_TIME_TAIL_SPLIT | MPDX_TRAIN_RESTREAM_RIDE | MPDX_TRAIN_REDUCE_JOIN | MPDX_TRAIN_GN_FUSE | ...   [rounds]
# prints ms per iteration at batch 32 x D=4 | batch 128 x D=14 | its fraction of the fp32 peak | batch 512 x D=14 with the switch at 0 and at 1
cd $GRAFT_REPO_ROOT; SW=${1:-MPDX_TIME_TAIL_SPLIT}; R=${2:-3}
for r in $(seq 1 $R); do
  for v in 0 1; do
    env $SW=$v python -c "
import bench
a = bench.training_leg(steps=100, baseline=False); b = bench.training_leg(steps=100, B=128, D=14, baseline=False); c = bench.training_leg(steps=40, B=512, D=14, baseline=False)
print('$SW=$v', a['ms_per_train_step'], b['ms_per_train_step'], b['roofline']['frac'], c['ms_per_train_step'])
" 2>/dev/null | tail -1
  done
done
