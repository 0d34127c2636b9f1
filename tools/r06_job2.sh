#!/bin/bash
# round 6, job 2: late weight gradients A/B (MPDX_TRAIN_WGRAD_LATE) + gradient tests with the switch on + the rest of the GPU suite + new bench line
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06b; mkdir -p $O
bash tools/ab_train_switch.sh MPDX_TRAIN_WGRAD_LATE 3 2>&1 | tee $O/train_wgrad_late_ab.txt
MPDX_TRAIN_WGRAD_LATE=1 timeout 900 python -m pytest tests/test_gpu_train.py -q -x > $O/pytest_train_late.log 2>&1; tail -4 $O/pytest_train_late.log
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
T0=$(date +%s); timeout 900 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "default bench.py wall: $(( $(date +%s) - T0 )) s" | tee $O/bench_default_wall.txt
tail -1 $O/bench_cfg2.json | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['roofline']); print(r['leg_seconds'])"
