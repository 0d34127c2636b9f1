#!/bin/bash
# Large seeded sweeps of the five fuzz tools on one MI355X (OMP_NUM_THREADS=16 for the CPU oracle), final tree of round 6
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06fz; mkdir -p $O; export OMP_NUM_THREADS=16; SEED=${1:-6}   # usage: tools/r06_fuzz_sweep.sh [seed]
{ echo "# Large seeded sweeps of the five fuzz tools on one MI355X (OMP_NUM_THREADS=16 for the CPU oracle), final tree of round 6, seed $SEED"
  echo "# python tools/<tool> <cases> <seed>; wall time per tool; the tests run a few cases of each (tests/test_gpu_fuzz.py)"
  for spec in "fuzz_plan.py 600 $SEED" "fuzz_contexts.py 250 $SEED" "fuzz_guide.py 600 $SEED" "fuzz_train.py 500 $SEED" "fuzz_planner.py 240 $SEED"; do set -- $spec
    echo "== $1"; ( time timeout 1500 python tools/$1 $2 $3 2>/dev/null | tail -4 ) 2>&1 | grep -v "^$\|user\|sys"
  done; } | tee $O/fuzz_sweep.txt
