#!/bin/bash
# gpurun job: SQ counters (MFMA busy, waits, LDS conflicts) and HBM-side traffic of the TRAINING iteration at batch 128 x D = 14 (the reference's launch batch),
# separate --pmc passes with --kernel-trace only (MI355X_MICROARCH.md, HBM / rocprofv3 section).  Outputs: gpurun_out/r06tp/
cd /tmp; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r06tp; mkdir -p $O
CMD="python -c \"import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT'); import bench; print(bench.training_leg(steps=12, B=128, D=14, baseline=False)['ms_per_train_step'])\""
MPDX_TRAIN_GRAPH=0 timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq -- bash -c "$CMD" > /dev/null 2>&1
MPDX_TRAIN_GRAPH=0 timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- bash -c "$CMD" > /dev/null 2>&1
MPDX_TRAIN_GRAPH=0 timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- bash -c "$CMD" > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $O/pmc_sq > $O/pmc_sq_summary_train128.json
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write 128 > $O/pmc_traffic_train128.json
rm -rf $O/pmc_sq $O/pmc_fetch $O/pmc_write; ls -la $O; head -c 1500 $O/pmc_sq_summary_train128.json
