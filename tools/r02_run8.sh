#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02q; mkdir -p $O
MPDX_DEBUG=1 timeout 1200 python -m pytest tests/test_gpu_train.py -x -q -s 2>&1 | grep -v "^$" > $O/pytest_train.txt; tail -40 $O/pytest_train.txt | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
timeout 900 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | cut -c1-200
