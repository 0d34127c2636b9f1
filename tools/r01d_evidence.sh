set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r01d
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python bench.py > gpurun_out/r01d/bench_cfg2.json 2> gpurun_out/r01d/bench_cfg2.err; tail -1 gpurun_out/r01d/bench_cfg2.json | cut -c1-400
for c in cfg3 cfg4 cfg5; do timeout 400 python bench.py --config $c --no-cpu-baseline > gpurun_out/r01d/bench_$c.json 2>/dev/null; tail -1 gpurun_out/r01d/bench_$c.json | cut -c1-200; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r01d/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/r01d/prof_bench.json 2>/dev/null
cd $GRAFT_REPO_ROOT
find gpurun_out/r01d/prof -name "*kernel_stats.csv" | head -2
timeout 300 python tools/fused_trace.py 100 2>&1 | grep -v amdgpu > gpurun_out/r01d/fused_trace.txt
timeout 300 python tools/layer_trace.py 100 2>&1 | grep -v amdgpu > gpurun_out/r01d/layer_trace.txt
timeout 300 python tools/guide_trace.py 2>&1 | grep -v amdgpu > gpurun_out/r01d/guide_trace.txt
MPDX_BENCH_TABLE=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2> gpurun_out/r01d/launch_table.txt >/dev/null
