#!/bin/bash
# gpurun job: the end-to-end demo (generate -> train -> plan with the trained weights) on the three environments of BASELINE configs[1..3], longer runs
cd $GRAFT_REPO_ROOT; O=gpurun_out/s3; mkdir -p $O
{ timeout 500 python tools/trained_demo.py --env EnvDense2D --contexts 256 --per 16 --steps 20000 2>/dev/null | tail -1
  timeout 500 python tools/trained_demo.py --env EnvNarrowPassageDense2D --contexts 256 --per 16 --steps 20000 2>/dev/null | tail -1
  timeout 500 python tools/trained_demo.py --env EnvSpheres3D --robot RobotPanda --contexts 128 --per 16 --steps 10000 --T 25 2>/dev/null | tail -1
} | tee $O/trained_demo.jsonl
