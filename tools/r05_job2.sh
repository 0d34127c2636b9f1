#!/bin/bash
# gpurun job: Panda guide with the kinematic-chain sphere groups: A/B vs the round-4 library, stamps, guide + full-size tests, cfg4 / cfg5 plans
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05; mkdir -p $O
{ MPDX_LIB=build_ab/libmpdx_r04.so timeout 300 python tools/guide_ab.py save /tmp/g_r04.pt
  timeout 300 python tools/guide_ab.py cmp /tmp/g_r04.pt
  MPDX_LIB=build_ab/libmpdx_dev.so timeout 300 python tools/guide_trace.py 6400 | grep -A20 RobotPanda
} 2>&1 | grep -v "amdgpu.ids\|Warn" | tee $O/guide_ab2.txt
timeout 900 python -m pytest tests/test_gpu_guide.py -m gpu -x -q 2>&1 | tail -3 | tee $O/guide_tests2.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3 | tee -a $O/guide_tests2.txt
for L in build_ab/libmpdx_r04.so mpd_public_amd/libmpdx.so; do for c in cfg4 cfg5; do
  MPDX_LIB=$L timeout 400 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L $c', d['ms_per_step'])"
done; done | tee $O/guide_plan_ab.txt
