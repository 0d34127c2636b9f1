#!/bin/bash
# round 6, job 6: guide probe variants; training kernel stats at batch 128 with the late weight gradients
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r06f; mkdir -p $O
timeout 300 python tools/guide_inplan_probe.py 6400 2>/dev/null | tee $O/guide_inplan_probe.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train128 -- python -c "
import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT')
import bench
print(bench.training_leg(steps=40, B=128, D=14, baseline=False))
" > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
cp $(find $O/prof_train128 -name "*kernel_stats.csv" | head -1) $O/train128_kernel_stats.csv; rm -rf $O/prof_train128
head -12 $O/train128_kernel_stats.csv | cut -c1-150
