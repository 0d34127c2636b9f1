#!/bin/bash
# One gpurun call that regenerates the r01d_* evidence under gpurun_out/r01d/ (copied into profiles/ afterwards).
cd $GRAFT_REPO_ROOT
O=gpurun_out/r01d; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err; tail -1 $O/bench_cfg2.json | cut -c1-200
for c in cfg3 cfg4 cfg5; do timeout 400 python bench.py --config $c --no-cpu-baseline > $O/bench_$c.json 2>/dev/null; tail -1 $O/bench_$c.json | cut -c1-160; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_sq -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
python tools/pmc_summary.py $O/pmc_sq > $O/pmc_sq_summary.json
rm -rf $O/prof $O/pmc_sq
timeout 300 python tools/fused_trace.py 100 2>&1 | grep -v amdgpu > $O/fused_trace.txt
timeout 300 python tools/layer_trace.py 100 2>&1 | grep -v amdgpu > $O/layer_trace.txt
timeout 300 python tools/guide_trace.py 2>&1 | grep -v amdgpu > $O/guide_trace.txt
timeout 300 python tools/ablate_loop.py 100 2>&1 | grep -v amdgpu > $O/ablate_loop.txt
MPDX_BENCH_TABLE=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 >/dev/null | grep "^#" > $O/launch_table.txt
head -3 $O/kernel_stats.csv | cut -c1-160
