#!/bin/bash
# PMC counters of n30 fused pairs (real kernel) and of the stand-alone probe, same counter sets:
#   tools/pmc_pair.sh <tag> "<pairs for tools/run_pair.py>" "<tri_probe arguments>"
set -u
TAG=$1; PAIRS=$2; PROBE=$3
export TMPDIR=/tmp
OUT=gpurun_out/pmcp_$TAG
rm -rf $OUT; mkdir -p $OUT
SETA="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
SETB="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS"
SETC="SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA"
for S in a b c; do
  case $S in a) C="$SETA";; b) C="$SETB";; c) C="$SETC";; esac
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/real_$S -- python3 tools/run_pair.py $PAIRS 2 > $OUT/real_$S.log 2>&1
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/probe_$S -- tools/probes/tri_probe $PROBE > $OUT/probe_$S.log 2>&1
done
python3 tools/pmc_pair.py $OUT
