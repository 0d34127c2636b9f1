#!/bin/bash
# round 6, job 3: the backward chain kernel - gradient tests, then A/B of MPDX_TRAIN_CHAIN (0 off / 32 / 16: smallest chained level)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06c; mkdir -p $O
MPDX_DEBUG_TRAIN=1 timeout 300 python -c "
import bench
print(bench.training_leg(steps=5, baseline=False)['ms_per_train_step'])
" 2>&1 | grep "mpdx\]\|^[0-9]" | sort | uniq -c | tail -5
timeout 900 python -m pytest tests/test_gpu_train.py -q -x > $O/pytest_train_chain.log 2>&1; tail -6 $O/pytest_train_chain.log
MPDX_TRAIN_CHAIN=16 timeout 900 python -m pytest tests/test_gpu_train.py -q -x > $O/pytest_train_chain16.log 2>&1; tail -3 $O/pytest_train_chain16.log
bash tools/ab_train_env.sh MPDX_TRAIN_CHAIN "0 32 16" 3 2>&1 | tee $O/train_chain_ab.txt
