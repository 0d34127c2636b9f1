#!/bin/bash
# Last evidence pass of round 6: the TRAINING records on the final tree (after the host fix of TrainStep.step and the split divisor), both networks - the
# planning sources are those tools/r06_evidence_c.sh measured (same fingerprint) - plus the default bench line, the GPU tests and smoke.  Outputs: gpurun_out/r06evd.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06evd; mkdir -p $O
T0=$(date +%s); timeout 900 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "default bench.py wall: $(( $(date +%s) - T0 )) s" | tee $O/bench_default_wall.txt; cp bench_full.json $O/bench_cfg2_full.json; tail -1 $O/bench_cfg2.json | cut -c1-200; wc -c $O/bench_cfg2.json
cd /tmp && export TMPDIR=/tmp
for spec in "train 32 4 1" "train128 128 14 1" "train512 512 14 1" "train128_three_level 128 14 0"; do set -- $spec
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$1 -- python -c "
import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT')
import bench
print(bench.training_leg(steps=40, B=$2, D=$3, opt=$4, baseline=False))
" > /dev/null 2>&1
done
MPDX_TRAIN_GRAPH=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace1 -- python $GRAFT_REPO_ROOT/tools/train_trace_probe.py run 32 4 1 1 > /dev/null 2>&1
MPDX_TRAIN_GRAPH=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace128 -- python $GRAFT_REPO_ROOT/tools/train_trace_probe.py run 128 14 1 1 > /dev/null 2>&1
MPDX_TRAIN_GRAPH=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace0 -- python $GRAFT_REPO_ROOT/tools/train_trace_probe.py run 32 4 1 0 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
for n in train train128 train512 train128_three_level; do cp $(find $O/prof_$n -name "*kernel_stats.csv" | head -1) $O/${n}_kernel_stats.csv; rm -rf $O/prof_$n; done
python tools/train_trace_probe.py show $(find $O/trace1 -name "*kernel_trace.csv" | head -1) > $O/train_iteration_trace.txt
python tools/train_trace_probe.py show $(find $O/trace128 -name "*kernel_trace.csv" | head -1) > $O/train_iteration_trace_batch128.txt
python tools/train_trace_probe.py show $(find $O/trace0 -name "*kernel_trace.csv" | head -1) > $O/train_iteration_trace_three_level.txt; rm -rf $O/trace1 $O/trace128 $O/trace0; wc -l $O/train_iteration_trace*.txt
timeout 900 python -c "
import json, bench
print(json.dumps({'batch32_D4': bench.training_leg(), 'batch128_D14': bench.training_leg(B=128, D=14), 'batch512_D14': bench.training_leg(steps=20, B=512, D=14, baseline=False),
 'three_level_batch32_D4': bench.training_leg(opt=0), 'three_level_batch128_D14': bench.training_leg(B=128, D=14, opt=0), 'three_level_batch512_D14': bench.training_leg(steps=20, B=512, D=14, opt=0, baseline=False)}, indent=1))
" 2>/dev/null > $O/training.json
bash tools/ab_train_env.sh MPDX_TRAIN_BWD_PROG "0 1" 3 2>&1 | tee $O/train_bwd_prog_ab.txt
{ echo "# training iteration (ms) of the THREE-level network (dim_mults option 0), batch 32 x D=4 | 128 x D=14 | 512 x D=14, interleaved on one MI355X"
  echo "# base: MPDX_TRAIN_BWD_PROG=0 MPDX_TRAIN_FUSED_FWD=0 (one launch per layer both ways) | fwd: forward programs only | all: the default (forward + backward programs)"
  for r in 1 2 3; do for v in "base 0 0" "fwd 0 1" "all 1 1"; do set -- $v
    MPDX_TRAIN_BWD_PROG=$2 MPDX_TRAIN_FUSED_FWD=$3 python -c "
import bench
a = bench.training_leg(steps=100, opt=0, baseline=False); b = bench.training_leg(steps=100, B=128, D=14, opt=0, baseline=False); c = bench.training_leg(steps=40, B=512, D=14, opt=0, baseline=False)
print('$1', a['ms_per_train_step'], b['ms_per_train_step'], c['ms_per_train_step'])
" 2>/dev/null | tail -1; done; done; } | tee $O/train_three_level_ab.txt
timeout 300 python tools/train_enqueue_probe2.py 32 2>/dev/null | head -7 > $O/train_enqueue_probe.txt
timeout 300 python tools/train_loop_probe.py 2>/dev/null > $O/train_loop_probe.txt
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; grep -ah "passed\|failed" $O/pytest_gpu.log | tail -3 > $O/pytest_gpu_tail.txt; cat $O/pytest_gpu_tail.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
