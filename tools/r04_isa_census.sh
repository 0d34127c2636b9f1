#!/bin/bash
# Instruction census (tools/isa_census.py) of the three hot kernel families, round-3 final (commit 9cb7e87) vs the working tree:
#   tools/r04_isa_census.sh > profiles/r04_isa_census.txt        (CPU only: hipcc cross-compiles gfx950)
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
T=$(mktemp -d)
HIPCC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on --cuda-device-only -S"
mkdir -p $T/old/mpd_public_amd/csrc $T/old/include
for f in conv_block.hpp conv_ws.hpp fused_level.hpp train.hpp guide.hpp mpdx.hip train_host.hpp planner.hpp planner_host.hpp; do git -C $ROOT show 9cb7e87:mpd_public_amd/csrc/$f > $T/old/mpd_public_amd/csrc/$f; done
git -C $ROOT show 9cb7e87:include/mpdx.h > $T/old/include/mpdx.h
$HIPCC -o $T/old.s $T/old/mpd_public_amd/csrc/mpdx.hip 2>/dev/null
for k in k_fused k_ws k_conv; do $HIPCC -o $T/$k.s $ROOT/mpd_public_amd/csrc/$k.hip 2>/dev/null; done
echo "# ISA census, per wave: static instruction counts of straight-line kernels (fused programs) / of kernel + loop body (conv kernels)."
echo "# issue-cycle estimate: MFMA 16x16x4 f32 = 32, VALU = 4, transcendental = 16 (one wave per SIMD; fp32 MFMA and VALU share the issue port)"
echo; echo "#### ROUND 3 (commit 9cb7e87) ####"
python $ROOT/tools/isa_census.py $T/old.s "fused_program_kernelINS_8FusedSeqIJLi0ELi1ELi2ELi2ELi3ELi4ELi5ELi6ELi6ELi7ELi16ELi17ELi18ELi18ELi19EEEELb0" --segments
python $ROOT/tools/isa_census.py $T/old.s "fused_program_kernelINS_8FusedSeqIJLi8ELi9ELi10ELi10ELi11ELi12ELi13ELi14ELi14ELi15ELi2ELi63EEEELb0"
python $ROOT/tools/isa_census.py $T/old.s "conv_ws_kernelILi16ELi32ELb0ELi2EE" --segments
python $ROOT/tools/isa_census.py $T/old.s "conv_block_kernelILi0ELi5ELi1ELi32ELi32ELi1ELi8EE" --segments
echo; echo "#### ROUND 4 (working tree) ####"
python $ROOT/tools/isa_census.py $T/k_fused.s "GeomDown3EJ" --segments | grep -v "Lb1EE"
python $ROOT/tools/isa_census.py $T/k_fused.s "GeomUpABEJ"
python $ROOT/tools/isa_census.py $T/k_ws.s "conv_ws_kernelILi16ELi32ELb0ELi2ELi1EE" --segments
python $ROOT/tools/isa_census.py $T/k_conv.s "conv_block_kernelILi0ELi5ELi1ELi32ELi32ELi1ELi8ENS_5GeoL8ILi16EEELi1EE" --segments
rm -rf $T
