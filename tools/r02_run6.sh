#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02h; mkdir -p $O
for s in 0 1 2 3 4; do
  MPDX_STAGGER=$s timeout 600 python bench.py --config cfg5 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-roofline > $O/bench_cfg5_st$s.json 2>/dev/null
  python -c "import json;d=json.loads(open('$O/bench_cfg5_st$s.json').read().strip().splitlines()[-1]);print('cfg5 stagger $s ms/plan', d['ms_per_step'])"
done
