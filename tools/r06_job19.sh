#!/bin/bash
# gpurun job: the three-level network with the Mid3 program + training variants of Down / Mid3: tests, then training A/B (MPDX_TRAIN_FUSED_FWD)
cd $GRAFT_REPO_ROOT
for opt in 0 1; do
MPDX_DEBUG_FUSE=1 python -c "
import mpd_public_amd as m
net = m.TemporalUnet(n_support_points=64, state_dim=4, unet_input_dim=32, dim_mults=m.UNET_DIM_MULTS[$opt]); net.cuda(); net._handle()
" 2>&1 | grep "fused segment" 
done
timeout 2400 python -m pytest tests/test_gpu_train.py tests/test_gpu_parity.py -q 2>&1 | tail -4
for r in 1 2; do
  for v in 1 0; do
    MPDX_TRAIN_FUSED_FWD=$v timeout 600 python -c "
import bench
a = bench.training_leg(steps=100, opt=0, baseline=False); b = bench.training_leg(steps=100, B=128, D=14, opt=0, baseline=False); c = bench.training_leg(steps=40, B=512, D=14, opt=0, baseline=False)
print('opt0 FUSED_FWD=$v', a['ms_per_train_step'], b['ms_per_train_step'], c['ms_per_train_step'])
" 2>/dev/null | tail -1
  done
done
