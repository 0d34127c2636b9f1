#!/bin/bash
# Second evidence pass of round 6 (the planning kernels are unchanged since tools/r06_evidence.sh ran: same source fingerprint): the training step with the
# whole-trajectory backward programs - kernel statistics at batch 32 / 128 / 512, the training records, an iteration's dispatch sequence - the default bench
# line, the GPU tests, smoke.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06evb; mkdir -p $O
T0=$(date +%s); timeout 900 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "default bench.py wall: $(( $(date +%s) - T0 )) s" | tee $O/bench_default_wall.txt; cp bench_full.json $O/bench_cfg2_full.json; tail -1 $O/bench_cfg2.json | cut -c1-200; wc -c $O/bench_cfg2.json
cd /tmp && export TMPDIR=/tmp
for spec in "train 32 4" "train128 128 14" "train512 512 14"; do set -- $spec
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$1 -- python -c "
import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT')
import bench
print(bench.training_leg(steps=40, B=$2, D=$3, baseline=False))
" > /dev/null 2>&1
done
MPDX_TRAIN_GRAPH=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace1 -- python $GRAFT_REPO_ROOT/tools/train_trace_probe.py run 32 4 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
for n in train train128 train512; do cp $(find $O/prof_$n -name "*kernel_stats.csv" | head -1) $O/${n}_kernel_stats.csv; rm -rf $O/prof_$n; done
python tools/train_trace_probe.py show $(find $O/trace1 -name "*kernel_trace.csv" | head -1) > $O/train_iteration_trace.txt; rm -rf $O/trace1; wc -l $O/train_iteration_trace.txt
timeout 900 python -c "
import json, bench
print(json.dumps({'batch32_D4': bench.training_leg(), 'batch128_D14': bench.training_leg(B=128, D=14), 'batch512_D14': bench.training_leg(steps=20, B=512, D=14, baseline=False)}, indent=1))
" 2>/dev/null > $O/training.json
bash tools/ab_train_env.sh MPDX_TRAIN_BWD_PROG "0 1" 3 2>&1 | tee $O/train_bwd_prog_ab.txt
timeout 300 python tools/guide_inplan_probe.py 6400 2>/dev/null > $O/guide_inplan_probe.txt
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; grep -h "passed\|failed" $O/pytest_gpu.log | tail -3 > $O/pytest_gpu_tail.txt; cat $O/pytest_gpu_tail.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
