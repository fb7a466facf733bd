#!/bin/bash
# PMC counters of one fixture's launches: tools/pmc_case.sh <tag> <fixture.npz> [env assignments exported first, e.g. ARTN_WIDE=1]
set -u
TAG=$1; FIX=$2; shift 2
for kv in "$@"; do export "$kv"; done
export TMPDIR=/tmp
OUT=gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $OUT/a -- python3 tools/time_case.py $FIX > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS \
  --kernel-trace --output-format csv -d $OUT/b -- python3 tools/time_case.py $FIX > $OUT/b.log 2>&1
# (FETCH_SIZE and WRITE_SIZE in ONE pass hang the box until the limit kills the call -- twice in round 5, 25 GPU-minutes each:
#  the traffic figures come from tools/profile_traffic.sh, separate passes; nothing of that kind here any more)
python3 tools/pmc_case.py $OUT
