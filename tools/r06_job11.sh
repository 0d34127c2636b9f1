#!/bin/bash
# round 6, job 11: backward programs with the lean epilogue: gradient tests, A/B, small-batch split multiplier, kernel durations
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r06j; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_train.py -q -x > $O/pytest_train.log 2>&1; tail -5 $O/pytest_train.log
bash tools/ab_train_env.sh MPDX_TRAIN_BWD_PROG "0 1" 2 2>&1 | tee $O/train_bwd_prog_ab.txt
cd /tmp && export TMPDIR=/tmp
for spec in "32 4" "128 14"; do set -- $spec
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$1 -- python -c "
import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT')
import bench
print(bench.training_leg(steps=40, B=$1, D=$2, baseline=False))
" > /dev/null 2>&1
cp $(find $O/prof_$1 -name "*kernel_stats.csv" | head -1) $O/train$1_kernel_stats.csv; rm -rf $O/prof_$1
grep "fused_bwd\|wgrad_multi" $O/train$1_kernel_stats.csv | cut -c1-50,100-160
done
