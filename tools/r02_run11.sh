#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_train.py -x -q -k "end_to_end or mirrors" 2>&1 | grep -v "^$" | tail -25 | cut -c1-250
