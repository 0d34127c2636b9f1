#!/bin/bash
# round 6, job 8: the whole-trajectory backward program (downs[0..2]): gradient tests, then A/B against the per-layer path
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06g; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_train.py -q -x -k "every_gradient_vs_oracle or two_training_steps" > $O/pytest_prog.log 2>&1; tail -25 $O/pytest_prog.log
bash tools/ab_train_env.sh MPDX_TRAIN_BWD_PROG "0 1" 2 2>&1 | tee $O/train_bwd_prog_ab.txt
