#!/bin/bash
# round 6, job 4: where do the 16 copyBuffer dispatches of a training iteration come from? (kernel trace of eager + graph iterations)
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r06d; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for mode in 0 1; do
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace$mode -- python $GRAFT_REPO_ROOT/tools/train_trace_probe.py run 32 4 $mode > $O/rocprof_mode$mode.log 2>&1
f=$(find $O/trace$mode -name "*kernel_trace.csv" | head -1)
(cd $GRAFT_REPO_ROOT && python tools/train_trace_probe.py show $f > $O/train_iteration_trace_mode$mode.txt); rm -rf $O/trace$mode
done
grep -c copyBuffer $O/train_iteration_trace_mode0.txt $O/train_iteration_trace_mode1.txt; tail -5 $O/rocprof_mode0.log
cd $GRAFT_REPO_ROOT
bash tools/ab_train_env.sh MPDX_WGRAD_LATE_DIV "1 2 4" 2 2>&1 | tee $O/train_late_div_ab.txt
