#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02l; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | grep -v "^$" | tail -2
for n in default NOPF; do
  L=mpd_public_amd/libmpdx_$n.so; [ $n = default ] && L=mpd_public_amd/libmpdx.so
  MPDX_LIB=$GRAFT_REPO_ROOT/$L MPDX_BENCH_TABLE=1 timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/bench_$n.json 2> $O/bench_$n.err
  python -c "import json;d=json.loads(open('$O/bench_$n.json').read().strip().splitlines()[-1]);print('$n cfg2 ms/plan', d['ms_per_step'])"; grep "^#" $O/bench_$n.err | sed -n 1,3p
done
python tools/fused_trace.py 100 2>&1 | grep -v amdgpu > $O/fused_trace.txt; grep "segment\|op0 \|op1 " $O/fused_trace.txt
