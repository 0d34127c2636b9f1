#!/bin/bash
# gpurun job: conv_wsp (128->256 + 1x1 pair) bit-identity + cfg5 A/B, width tests, new training tests, planner tests + GPMP2 timing after the linearisation rewrite
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05; mkdir -p $O
run_tests() { timeout 1500 python -m pytest "$@" -m gpu -q > $O/_t.log 2>&1; grep -E "passed|failed" $O/_t.log | tail -1; grep -E "^FAILED|^ERROR" $O/_t.log | head; }
{ echo "== parity (weight stationary)"; run_tests tests/test_gpu_parity.py -k "weight_stationary"
  echo "== widths"; run_tests tests/test_gpu_widths.py
  echo "== train (new)"; run_tests tests/test_gpu_train.py -k "launch_forms or pending or graph_replayed"
  echo "== planner"; run_tests tests/test_gpu_planner.py
  echo "== fullsize"; run_tests tests/test_gpu_fullsize.py; } 2>&1 | tee $O/job5_tests.txt
cp $O/_t.log $O/job5_last_test.log
for r in 1 2; do for w in 0 1; do
  MPDX_WSP=$w timeout 400 python bench.py --config cfg5 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('MPDX_WSP=$w cfg5', d['ms_per_step'])"
done; done | tee $O/wsp_plan_ab.txt
timeout 600 python -c "
import json, bench
print(json.dumps(bench.planner_baseline_leg(), indent=1))
" 2>/dev/null | tee $O/planner_baseline.json | head -30
timeout 300 python tools/gpmp_phase_probe.py 2>&1 | grep "solve=\|sigma_obs" | tee $O/gpmp_probe.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_planner -- python -c "
import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT')
import bench
print(bench.planner_baseline_leg())
" > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; cp $(find $O/prof_planner -name "*kernel_stats.csv" | head -1) $O/planner_kernel_stats.csv; rm -rf $O/prof_planner; head -6 $O/planner_kernel_stats.csv | cut -c1-160
