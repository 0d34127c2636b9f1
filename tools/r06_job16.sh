#!/bin/bash
# round 6, job 16: repack kernel with 4096-output chunks: training tests + timing, then the fuzz sweep
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06m; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_parity.py -q -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for r in 1 2 3; do python -c "
import bench
a = bench.training_leg(steps=150, baseline=False); b = bench.training_leg(steps=100, B=128, D=14, baseline=False)
print('pack4096', a['ms_per_train_step'], b['ms_per_train_step'])
" 2>/dev/null | tail -1; done | tee $O/train_pack4096.txt
bash tools/r06_fuzz_sweep.sh
