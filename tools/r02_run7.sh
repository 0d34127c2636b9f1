#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02p; mkdir -p $O
(cd tools/micro && timeout 120 ./kp_plain > ../../$O/kp_plain.txt 2>&1; timeout 120 ./kp_preload > ../../$O/kp_preload.txt 2>&1)
echo "--- plain"; cat $O/kp_plain.txt; echo "--- preload"; cat $O/kp_preload.txt
HIP_FORCE_DEV_KERNARG=1 timeout 120 tools/micro/kp_plain | head -4
HIP_FORCE_DEV_KERNARG=0 timeout 120 tools/micro/kp_plain | head -4
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_guide.py -x -q 2>&1 | grep -v "^$" | tail -5 | tee $O/pytest_parity.txt
timeout 300 python tools/fused_trace.py 100 2>&1 | grep -v amdgpu > $O/fused_trace.txt; tail -8 $O/fused_trace.txt
MPDX_BENCH_TABLE=1 timeout 900 python bench.py --no-cpu-baseline --no-extras > $O/bench_cfg2.json 2> $O/bench_cfg2.err; tail -1 $O/bench_cfg2.json | cut -c1-200; grep "^#" $O/bench_cfg2.err | head -3
HIP_FORCE_DEV_KERNARG=1 timeout 900 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | cut -c1-200
HIP_FORCE_DEV_KERNARG=0 timeout 900 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | cut -c1-200
timeout 900 python bench.py --config cfg3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | cut -c1-200
