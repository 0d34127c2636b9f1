#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02f; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^$" | tail -60 | tee $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
