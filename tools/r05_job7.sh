#!/bin/bash
# gpurun job: cooperative noise in the fused final op + coalesced guide prologue: parity / guide tests, guide timing, cfg5 kernel stats + plan time
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05; mkdir -p $O
run_tests() { timeout 1500 python -m pytest "$@" -m gpu -q > $O/_t7.log 2>&1; grep -E "passed|failed" $O/_t7.log | tail -1; grep -E "^FAILED|^ERROR" $O/_t7.log | head; grep -B30 "^FAILED" $O/_t7.log | grep -E "^E " | head -8; }
{ echo "== parity"; run_tests tests/test_gpu_parity.py
  echo "== guide"; run_tests tests/test_gpu_guide.py
  echo "== guided class + entry"; run_tests tests/test_gpu_guided_class.py tests/test_gpu_entry.py; } 2>&1 | tee $O/job7_tests.txt
timeout 300 python tools/guide_ab.py 2>&1 | grep -v "amdgpu.ids\|Warn" | tee $O/guide_ab7.txt
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_cfg5 -- python $GRAFT_REPO_ROOT/bench.py --config cfg5 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extras > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
cp $(find $O/prof_cfg5 -name "*kernel_stats.csv" | head -1) $O/cfg5_kernel_stats_d.csv; rm -rf $O/prof_cfg5
head -8 $O/cfg5_kernel_stats_d.csv | cut -c1-170
for r in 1 2 3; do timeout 400 python bench.py --config cfg5 --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg5', d['ms_per_step'])"; done | tee $O/cfg5_plan7.txt
