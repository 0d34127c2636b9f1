#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02s; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_train.py -x -q 2>&1 | grep -v "^$" | tail -5
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
timeout 1200 python - > $O/training_leg.txt 2>&1 <<'PY'
import json, bench
print(json.dumps(bench.training_leg()))
print(json.dumps(bench.training_leg(B=128, D=14)))
PY
grep -v amdgpu $O/training_leg.txt | cut -c1-400
timeout 900 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | cut -c1-200
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_train -- python -c "
import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT')
import bench
print(bench.training_leg(steps=20))
" > $GRAFT_REPO_ROOT/$O/prof_train.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof_train -name "*kernel_stats.csv" | head -1); cp "$f" $O/train_kernel_stats.csv
