#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | grep -v "^$" | tail -4 | tee $O/pytest_parity.txt
for r in 4 8 16 32; do
  L=mpd_public_amd/libmpdx_ring$r.so; [ $r = 16 ] && L=mpd_public_amd/libmpdx.so
  echo "== ring blocks $r"
  MPDX_LIB=$GRAFT_REPO_ROOT/$L MPDX_BENCH_TABLE=1 timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/bench_ring$r.json 2> $O/bench_ring$r.err
  python -c "import json;d=json.loads(open('$O/bench_ring$r.json').read().strip().splitlines()[-1]);print('cfg2 ms/plan', d['ms_per_step'])"; grep "^#" $O/bench_ring$r.err | head -4
done
for r in 4 16 32; do
  L=mpd_public_amd/libmpdx_ring$r.so; [ $r = 16 ] && L=mpd_public_amd/libmpdx.so
  MPDX_LIB=$GRAFT_REPO_ROOT/$L timeout 600 python bench.py --config cfg5 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-roofline > $O/bench_cfg5_ring$r.json 2>/dev/null
  python -c "import json;d=json.loads(open('$O/bench_cfg5_ring$r.json').read().strip().splitlines()[-1]);print('cfg5 ring $r ms/plan', d['ms_per_step'])"
done
