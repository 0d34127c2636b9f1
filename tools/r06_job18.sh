#!/bin/bash
# gpurun job: kernel trace of one training iteration (batch 32) on both networks:  option 1 (four levels), option 0 (three levels)
cd /tmp; export TMPDIR=/tmp; O=gpurun_out/j18; mkdir -p $GRAFT_REPO_ROOT/$O
for opt in 1 0; do
MPDX_TRAIN_GRAPH=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace$opt -- python $GRAFT_REPO_ROOT/tools/train_trace_probe.py run 32 4 1 $opt > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
for opt in 1 0; do python tools/train_trace_probe.py show $(find $O/trace$opt -name "*kernel_trace.csv" | head -1) > $O/train_iteration_trace_opt$opt.txt; rm -rf $O/trace$opt; done
wc -l $O/*.txt
