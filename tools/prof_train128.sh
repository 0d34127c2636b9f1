#!/bin/bash
# gpurun job: rocprofv3 kernel stats of the training iteration at batch 128 x D = 14 (this tree)
cd $GRAFT_REPO_ROOT; O=gpurun_out/s3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_train -- python -c "
import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT')
import bench
print(bench.training_leg(steps=40, B=128, D=14, baseline=False))
" > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
cp $(find $O/prof_train -name "*kernel_stats.csv" | head -1) $O/train128_kernel_stats_new.csv; rm -rf $O/prof_train
