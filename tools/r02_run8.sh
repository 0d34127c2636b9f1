#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02k; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof5 -- python $GRAFT_REPO_ROOT/bench.py --config cfg5 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extras > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
cp $(find $O/prof5 -name "*kernel_stats.csv" | head -1) $O/cfg5_kernel_stats.csv; rm -rf $O/prof5
head -24 $O/cfg5_kernel_stats.csv | cut -c1-170
for b in 800 1600 3200; do for f in 0 1; do
MPDX_FUSED=$f python - <<PY
import sys,time,torch
sys.path[:0]=['.','tests']
from bench import build_model
dm,_=build_model(14,(1,2,4,8),100,'cuda')
B=$b
x=torch.randn(B,64,14,device='cuda'); tt=torch.full((B,),50,device='cuda',dtype=torch.long)
for _ in range(3): dm.model(x,tt)
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(10): dm.model(x,tt)
torch.cuda.synchronize(); print('B',B,'fused',$f,'unet pass ms',(time.perf_counter()-t0)/10*1e3)
PY
done; done
