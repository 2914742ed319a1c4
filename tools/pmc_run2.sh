#!/bin/bash
# Memory-pipeline PMC passes over tools/prof_frame.py (one rocprofv3 run per pass, counters only).
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/pmc2
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum GRBM_GUI_ACTIVE" \
           "TA_FLAT_READ_WAVEFRONTS_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum GRBM_GUI_ACTIVE" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC TD_TD_BUSY_sum TD_TC_STALL_sum"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $set -d $OUT/p$i -o p$i --output-format csv -- python3 tools/prof_frame.py "$@" > $OUT/p$i.log 2>&1
  tail -1 $OUT/p$i.log
done
python3 tools/pmc_summary.py $OUT/p* > gpurun_out/pmc2_summary.csv
find $OUT -name "*.csv" -size +2M -delete
