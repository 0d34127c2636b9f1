#!/bin/bash
# GPU run 1 of round 2: full GPU test suite + default bench + launch-boundary microbenchmark + fresh cfg2 rocprof baseline
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02a; mkdir -p $O
timeout 300 tools/micro/launch_boundary > $O/launch_boundary.txt 2>&1; tail -12 $O/launch_boundary.txt
timeout 2400 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -v "^$" | tail -25 | tee $O/pytest_gpu.txt
MPDX_BENCH_TABLE=1 timeout 900 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err; tail -1 $O/bench_cfg2.json | cut -c1-600; grep "^#" $O/bench_cfg2.err
