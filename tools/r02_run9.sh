#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02r; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_train -- python -c "
import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT')
import bench
print(bench.training_leg(steps=20))
" > $GRAFT_REPO_ROOT/$O/prof_train.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof_train -name "*kernel_stats.csv" | head -1); cp "$f" $O/train_kernel_stats.csv; head -45 $O/train_kernel_stats.csv | cut -c1-220
