#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02n; mkdir -p $O
MPDX_DEBUG_FUSE=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | grep -v "^$" | tail -12 | tee $O/pytest_parity.txt
timeout 300 python tools/fused_trace.py 100 2>&1 | grep -v amdgpu > $O/fused_trace.txt; cat $O/fused_trace.txt
MPDX_BENCH_TABLE=1 timeout 900 python bench.py --no-cpu-baseline --no-extras > $O/bench_cfg2.json 2> $O/bench_cfg2.err; tail -1 $O/bench_cfg2.json | cut -c1-300; grep "^#" $O/bench_cfg2.err
