#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02s; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_train.py -x -q 2>&1 | grep -v "^$" | tail -5
MPDX_TRAIN_DEFERRED=0 timeout 1200 python -m pytest tests/test_gpu_train.py -x -q -k "every_gradient" 2>&1 | grep -v "^$" | tail -2
timeout 1200 python - > $O/training_leg.txt 2>&1 <<'PY'
import json, bench
print(json.dumps(bench.training_leg()))
print(json.dumps(bench.training_leg(B=128, D=14)))
PY
grep -v amdgpu $O/training_leg.txt | cut -c1-400
