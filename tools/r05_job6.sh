#!/bin/bash
# gpurun job: conv_wsp under rocprofv3 (cfg5 kernel stats), interleaved cfg5 A/B with more steps, width tests
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_widths.py -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED" | tee $O/widths_tests.txt
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_cfg5 -- python $GRAFT_REPO_ROOT/bench.py --config cfg5 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extras > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
cp $(find $O/prof_cfg5 -name "*kernel_stats.csv" | head -1) $O/cfg5_kernel_stats_c.csv; rm -rf $O/prof_cfg5
head -12 $O/cfg5_kernel_stats_c.csv | cut -c1-170
for r in 1 2 3; do for w in 0 1; do
  MPDX_WSP=$w timeout 400 python bench.py --config cfg5 --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('MPDX_WSP=$w cfg5', d['ms_per_step'])"
done; done | tee $O/wsp_plan_ab2.txt
