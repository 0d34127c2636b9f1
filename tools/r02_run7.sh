#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02j; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | grep -v "^$" | tail -2
for n in default RINGFIRST PF3 PF4 PF6; do
  L=mpd_public_amd/libmpdx_$n.so; [ $n = default ] && L=mpd_public_amd/libmpdx.so
  MPDX_LIB=$GRAFT_REPO_ROOT/$L MPDX_BENCH_TABLE=1 timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/bench_$n.json 2> $O/bench_$n.err
  python -c "import json;d=json.loads(open('$O/bench_$n.json').read().strip().splitlines()[-1]);print('$n cfg2 ms/plan', d['ms_per_step'])"; grep "^#" $O/bench_$n.err | sed -n 2,4p
done
for n in default PF4; do
  L=mpd_public_amd/libmpdx_$n.so; [ $n = default ] && L=mpd_public_amd/libmpdx.so
  MPDX_LIB=$GRAFT_REPO_ROOT/$L timeout 600 python bench.py --config cfg5 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-roofline > $O/bench_cfg5_$n.json 2>/dev/null
  python -c "import json;d=json.loads(open('$O/bench_cfg5_$n.json').read().strip().splitlines()[-1]);print('cfg5 $n ms/plan', d['ms_per_step'])"
done
