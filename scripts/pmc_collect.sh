#!/bin/bash
# PMC passes for one config (separate rocprofv3 runs: counters only, no tracing domains besides --kernel-trace).
# usage: scripts/pmc_collect.sh <config 2|3|4> <frames> <outdir under gpurun_out>
set -u
CFG=${1:-2}; FRAMES=${2:-4}; OUT=${3:-pmc}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p $ROOT/gpurun_out/$OUT
cd /tmp
i=0
for SET in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
  "SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
  "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
  "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum" \
  "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCC_READ_REQ_LATENCY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
  "FETCH_SIZE" \
  "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $ROOT/gpurun_out/$OUT/pass$i -- python3 $ROOT/scripts/pmc_run.py $CFG $FRAMES > $ROOT/gpurun_out/$OUT/pass$i.log 2>&1 || echo "pass $i failed"
  echo "pass $i done: $SET"
done
python3 $ROOT/scripts/pmc_summarize.py $ROOT/gpurun_out/$OUT > $ROOT/gpurun_out/$OUT/summary.txt 2>&1
cat $ROOT/gpurun_out/$OUT/summary.txt
