#!/bin/bash
# A/B of two builds of libmpdx.so on the training iteration, interleaved rounds: tools/ab_train.sh <libA> <libB> [rounds]
A=$1; B=$2; R=${3:-3}
for r in $(seq 1 $R); do
  for L in $A $B; do
    MPDX_LIB=$L python -c "
import bench
a = bench.training_leg(steps=100, baseline=False); b = bench.training_leg(steps=60, B=128, D=14, baseline=False)
print('$L', a['ms_per_train_step'], b['ms_per_train_step'])
" 2>/dev/null | tail -1
  done
done
