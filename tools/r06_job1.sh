#!/bin/bash
# round 6, job 1: the GPU tests on the tree with the closed refusals + the new bench line
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06a; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; tail -15 $O/pytest_gpu.log
T0=$(date +%s); timeout 900 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "default bench.py wall: $(( $(date +%s) - T0 )) s" | tee $O/bench_default_wall.txt
tail -1 $O/bench_cfg2.json; wc -c $O/bench_cfg2.json
cp bench_full.json $O/bench_full.json 2>/dev/null
