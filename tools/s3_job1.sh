#!/bin/bash
# gpurun job: training at horizons 16 / 32 / 128 against the oracle; training-step A/B (HEAD library vs this tree)
cd $GRAFT_REPO_ROOT; O=gpurun_out/s3; mkdir -p $O
timeout 560 python -m pytest tests/test_gpu_train.py -m gpu -x -q -k "horizons" -s 2>&1 | grep -v amdgpu.ids | tail -30 | tee $O/train_h.txt
bash tools/ab_train.sh build_ab/libmpdx_head.so mpd_public_amd/libmpdx.so 3 2>&1 | tee $O/ab_train_reduce.txt
