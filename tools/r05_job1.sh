#!/bin/bash
# gpurun job: Panda guide A/B (round-4 library vs this tree), dense-variant stamps, guide parity tests
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05; mkdir -p $O
{ MPDX_LIB=build_ab/libmpdx_r04.so timeout 300 python tools/guide_ab.py save /tmp/g_r04.pt
  timeout 300 python tools/guide_ab.py cmp /tmp/g_r04.pt
  MPDX_LIB=build_ab/libmpdx_r04.so timeout 300 python tools/guide_ab.py
  timeout 300 python tools/guide_ab.py
  MPDX_LIB=build_ab/libmpdx_dev.so timeout 300 python tools/guide_trace.py 6400
  MPDX_LIB=build_ab/libmpdx_dev.so timeout 300 python tools/guide_trace.py 100
} 2>&1 | grep -v "amdgpu.ids\|Warn" | tee $O/guide_ab.txt
timeout 900 python -m pytest tests/test_gpu_guide.py -m gpu -x -q 2>&1 | tail -3 | tee $O/guide_tests.txt
