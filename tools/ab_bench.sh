#!/bin/bash
# A/B of two builds of libmpdx.so on one box, interleaved rounds (cdna_hip_programming.md rule 24): tools/ab_bench.sh <libA> <libB> [config] [rounds]
A=$1; B=$2; CFG=${3:-cfg2}; R=${4:-3}
for r in $(seq 1 $R); do
  for L in $A $B; do
    MPDX_LIB=$L python bench.py --config $CFG --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', d['ms_per_step'])"
  done
done
