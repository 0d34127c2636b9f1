#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | grep -v "^$" > gpurun_out/r02/pytest_gpu_full.txt; tail -5 gpurun_out/r02/pytest_gpu_full.txt | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/r02_evidence.sh
