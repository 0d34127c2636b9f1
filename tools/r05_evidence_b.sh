#!/bin/bash
# Second evidence pass of round 5 (the planning kernels are unchanged since tools/r05_evidence.sh ran: same source fingerprint): the default bench line
# with the `trained` record and the two-form cpu_baseline, the training records and kernel statistics after the launch merges, the GPU tests, smoke.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05f; mkdir -p $O
T0=$(date +%s); timeout 900 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "default bench.py wall: $(( $(date +%s) - T0 )) s" | tee $O/bench_default_wall.txt; tail -1 $O/bench_cfg2.json | cut -c1-200
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_train -- python -c "
import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT')
import bench
print(bench.training_leg(steps=40, baseline=False))
" > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_train128 -- python -c "
import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT')
import bench
print(bench.training_leg(steps=40, B=128, D=14, baseline=False))
" > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
for n in train train128; do cp $(find $O/prof_$n -name "*kernel_stats.csv" | head -1) $O/${n}_kernel_stats.csv; rm -rf $O/prof_$n; done
timeout 900 python -c "
import json, bench
print(json.dumps({'batch32_D4': bench.training_leg(), 'batch128_D14': bench.training_leg(B=128, D=14), 'batch512_D14': bench.training_leg(steps=20, B=512, D=14, baseline=False)}, indent=1))
" 2>/dev/null > $O/training.json
(for v in 1 0; do MPDX_TORCH_DATALOADER=$v timeout 300 python tools/train_loop_probe.py 3000 2>/dev/null | tail -1; done) > $O/train_loop_probe.txt
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; grep -h "passed\|failed" $O/pytest_gpu.log | tail -3 > $O/pytest_gpu_tail.txt; cat $O/pytest_gpu_tail.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
