#!/bin/bash
# gpurun job: conv_wsn (128-channel inner layers without a K split) bit-identity + A/B, Panda guide after the gather fix, guided class test, cfg5 kernel stats
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05; mkdir -p $O
timeout 300 python tools/guide_ab.py 2>&1 | grep -v "amdgpu.ids\|Warn" | tee $O/guide_ab3.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "weight_stationary or unet_forward" 2>&1 | tail -3 | tee $O/wsn_tests.txt
timeout 900 python -m pytest tests/test_gpu_guided_class.py -m gpu -x -q -s 2>&1 | tail -12 | tee $O/guided_class.txt
for r in 1 2; do for w in 0 1; do
  MPDX_WSN=$w timeout 400 python bench.py --config cfg5 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('MPDX_WSN=$w cfg5', d['ms_per_step'])"
done; done | tee $O/wsn_plan_ab.txt
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_cfg5 -- python $GRAFT_REPO_ROOT/bench.py --config cfg5 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extras > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
cp $(find $O/prof_cfg5 -name "*kernel_stats.csv" | head -1) $O/cfg5_kernel_stats_a.csv; rm -rf $O/prof_cfg5
head -14 $O/cfg5_kernel_stats_a.csv | cut -c1-180
timeout 900 python -m pytest tests/test_gpu_guide.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3 | tee $O/guide_tests3.txt
