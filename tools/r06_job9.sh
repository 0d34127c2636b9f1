#!/bin/bash
# round 6, job 9: duration of fused_bwd_kernel at batch 32 / 128 (rocprofv3 kernel stats of the training loop)
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r06h; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for spec in "32 4" "128 14"; do set -- $spec
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$1 -- python -c "
import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT')
import bench
print(bench.training_leg(steps=40, B=$1, D=$2, baseline=False))
" > /dev/null 2>&1
cp $(find $O/prof_$1 -name "*kernel_stats.csv" | head -1) $O/train$1_kernel_stats.csv; rm -rf $O/prof_$1
grep "fused_bwd\|fused_program\|wgrad_multi\|reduce_colsum" $O/train$1_kernel_stats.csv | cut -c1-60,150-260
done
