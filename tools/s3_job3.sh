#!/bin/bash
# gpurun job: conv_ws merged epilogue A/B on cfg5 (HEAD library vs this tree) + the bit-identity tests of the weight-stationary kernels
cd $GRAFT_REPO_ROOT; O=gpurun_out/s3; mkdir -p $O
for r in 1 2; do
  for L in build_ab/libmpdx_head.so mpd_public_amd/libmpdx.so; do
    MPDX_LIB=$L python bench.py --config cfg5 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', d['ms_per_step'])"
  done
done 2>&1 | tee $O/ab_cfg5_merge.txt
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "stationary or weight or ws or fullsize or shard" 2>&1 | tail -4 | tee $O/ws_tests.txt
