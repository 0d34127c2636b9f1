#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02o; mkdir -p $O
MPDX_DEBUG_FUSE=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | grep -v "^$" | tail -12 | tee $O/pytest_parity.txt
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -q -s 2>&1 | grep -v "^$" > $O/pytest_fullsize.txt; tail -40 $O/pytest_fullsize.txt | cut -c1-250
timeout 300 python tools/fused_trace.py 100 2>&1 | grep -v amdgpu > $O/fused_trace.txt; tail -30 $O/fused_trace.txt
MPDX_BENCH_TABLE=1 timeout 900 python bench.py --no-cpu-baseline --no-extras > $O/bench_cfg2.json 2> $O/bench_cfg2.err; tail -1 $O/bench_cfg2.json | cut -c1-300; grep "^#" $O/bench_cfg2.err
MPDX_NO_MERGE_UP=1 timeout 900 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | cut -c1-200
timeout 900 python bench.py --config cfg5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | cut -c1-200
MPDX_NO_MERGE_UP=1 timeout 900 python bench.py --config cfg5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | cut -c1-200
