#!/bin/bash
# gpurun job: the whole GPU suite on the current tree + the default bench line + guide timing (cooperative noise draw) + cfg5 kernel stats
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05; mkdir -p $O
timeout 300 python tools/guide_ab.py 2>&1 | grep -v "amdgpu.ids\|Warn" | tee $O/guide_ab4.txt
timeout 2700 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3 | tee $O/pytest_gpu_tail.txt; grep -E "^FAILED|^ERROR|Error" $O/pytest_gpu.log | head -20
timeout 900 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err; tail -1 $O/bench_cfg2.json | cut -c1-300
for c in cfg5; do timeout 400 python bench.py --config $c --no-cpu-baseline --no-extras > $O/bench_$c.json 2>/dev/null; tail -1 $O/bench_$c.json | cut -c1-200; done
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_cfg5 -- python $GRAFT_REPO_ROOT/bench.py --config cfg5 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extras > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
cp $(find $O/prof_cfg5 -name "*kernel_stats.csv" | head -1) $O/cfg5_kernel_stats_b.csv; rm -rf $O/prof_cfg5
head -8 $O/cfg5_kernel_stats_b.csv | cut -c1-160
