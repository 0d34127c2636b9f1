#!/bin/bash
# round 6, job 12: where do the backward programs' microseconds go?  ablations (MPDX_BWD_DBG: 1 no operand loads, 2 no global stores, 4 no GroupNorm backward)
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r06k; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for dbg in 0 1 2 4 7; do
MPDX_BWD_DBG=$dbg timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$dbg -- python -c "
import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT')
import bench
print(bench.training_leg(steps=30, B=32, D=4, baseline=False)['ms_per_train_step'])
" > /dev/null 2>&1
f=$(find $O/prof_$dbg -name "*kernel_stats.csv" | head -1)
echo "dbg=$dbg $(grep 'fused_bwd' $f | awk -F, '{print $(NF-4)}' | tr '\n' ' ')"
rm -rf $O/prof_$dbg
done | tee $O/bwd_prog_ablation.txt
