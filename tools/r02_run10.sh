#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02m; mkdir -p $O
for n in default R8 R12 DB3 R8DB1; do
  L=mpd_public_amd/libmpdx_$n.so; [ $n = default ] && L=mpd_public_amd/libmpdx.so
  MPDX_LIB=$GRAFT_REPO_ROOT/$L MPDX_BENCH_TABLE=1 timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/bench_$n.json 2> $O/bench_$n.err
  python -c "import json;d=json.loads(open('$O/bench_$n.json').read().strip().splitlines()[-1]);print('$n cfg2 ms/plan', d['ms_per_step'])"; grep "^#" $O/bench_$n.err | grep fused
done
