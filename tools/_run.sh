#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_train.py tests/test_gpu_parity.py -x -q 2>&1 | grep -v "^$" | tail -3 | cut -c1-250
for r in 1 2; do timeout 600 python bench.py --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | tail -1 | cut -c60-140; done
timeout 600 python -c "
import json, bench
print('ms', bench.training_leg(baseline=False)['ms_per_train_step'], bench.training_leg(B=128, D=14, baseline=False)['ms_per_train_step'])
" 2>/dev/null | tail -1
