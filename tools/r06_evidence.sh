#!/bin/bash
# One gpurun call that regenerates the r06_* evidence under gpurun_out/r06e/ (copied into profiles/ afterwards).
# bench.py's stdout is the COMPACT line since round 6; the full record of a run is bench_full.json (copied per run below).
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06ev; mkdir -p $O
T0=$(date +%s); timeout 900 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "default bench.py wall: $(( $(date +%s) - T0 )) s" | tee $O/bench_default_wall.txt; cp bench_full.json $O/bench_cfg2_full.json; tail -1 $O/bench_cfg2.json | cut -c1-200; wc -c $O/bench_cfg2.json
for c in cfg3 cfg4 cfg5; do timeout 400 python bench.py --config $c --no-cpu-baseline --no-extras > $O/bench_$c.json 2>/dev/null; cp bench_full.json $O/bench_${c}_full.json; tail -1 $O/bench_$c.json | cut -c1-160; done
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extras"
BENCH1="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extras"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- $BENCH > /dev/null 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_fetch -- $BENCH1 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_write -- $BENCH1 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_sq -- $BENCH1 > /dev/null 2>&1
for c in cfg3 cfg4; do
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_${c}_fetch -- $BENCH1 --config $c > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_${c}_write -- $BENCH1 --config $c > /dev/null 2>&1
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_cfg5 -- $BENCH --config cfg5 --steps 2 > /dev/null 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc5_fetch -- $BENCH1 --config cfg5 > /dev/null 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc5_write -- $BENCH1 --config cfg5 > /dev/null 2>&1
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc5_sq -- $BENCH1 --config cfg5 > /dev/null 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_default -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/default_bench_under_rocprof.json 2>/dev/null
for spec in "train 32 4" "train128 128 14" "train512 512 14"; do set -- $spec
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$1 -- python -c "
import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT')
import bench
print(bench.training_leg(steps=40, B=$2, D=$3, baseline=False))
" > /dev/null 2>&1
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_planner -- python -c "
import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT')
import bench
print(bench.planner_baseline_leg())
" > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
for n in cfg5 train train128 train512 planner; do cp $(find $O/prof_$n -name "*kernel_stats.csv" | head -1) $O/${n}_kernel_stats.csv; rm -rf $O/prof_$n; done
timeout 900 python -c "
import json, bench
print(json.dumps({'batch32_D4': bench.training_leg(), 'batch128_D14': bench.training_leg(B=128, D=14), 'batch512_D14': bench.training_leg(steps=20, B=512, D=14, baseline=False)}, indent=1))
" 2>/dev/null > $O/training.json
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/cfg2_kernel_stats.csv
cp $(find $O/prof_default -name "*kernel_stats.csv" | head -1) $O/default_cmd_kernel_stats.csv
python tools/kernel_stats_json.py $O/cfg2_kernel_stats.csv 100 "python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extras" > $O/cfg2_kernel_stats.json
python tools/kernel_stats_json.py $O/cfg5_kernel_stats.csv 6400 "python bench.py --config cfg5 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extras" > $O/cfg5_kernel_stats.json
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write 100 > $O/pmc_traffic.json
for c in cfg3 cfg4; do python tools/pmc_traffic.py $O/pmc_${c}_fetch $O/pmc_${c}_write 100 > $O/guidepmc_$c.json; rm -rf $O/pmc_${c}_fetch $O/pmc_${c}_write; done
python tools/pmc_summary.py $O/pmc_sq > $O/pmc_sq_summary.json
python tools/pmc_traffic.py $O/pmc5_fetch $O/pmc5_write 6400 > $O/pmc_traffic_B6400.json
python tools/pmc_summary.py $O/pmc5_sq > $O/pmc_sq_summary_B6400.json
rm -rf $O/prof $O/prof_default $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc5_fetch $O/pmc5_write $O/pmc5_sq
MPDX_BENCH_TABLE=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>&1 >/dev/null | grep "^#" | grep -v bench_full > $O/launch_table.txt
MPDX_BENCH_TABLE=1 timeout 300 python bench.py --config cfg5 --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>&1 >/dev/null | grep "^#" | grep -v bench_full > $O/launch_table_cfg5.txt
timeout 600 python -c "
import json, bench
print(json.dumps(bench.planner_baseline_leg(), indent=1))
" 2>/dev/null > $O/planner_baseline.json
timeout 300 python tools/guide_inplan_probe.py 6400 2>/dev/null > $O/guide_inplan_probe.txt
# multi-GPU dry run on the single-GPU rig (8 ranks share the GPU over gloo): record shape + sharding / gather / checksum code, no fabric
timeout 1200 python bench.py --gpus 8 --steps 2 --warmup 1 > $O/bench_rig8_single_gpu.json 2> $O/bench_rig8.err; tail -1 $O/bench_rig8_single_gpu.json | cut -c1-300
head -4 $O/cfg2_kernel_stats.csv | cut -c1-160
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; grep -h "passed\|failed" $O/pytest_gpu.log | tail -3 > $O/pytest_gpu_tail.txt; cat $O/pytest_gpu_tail.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
