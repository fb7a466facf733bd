#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace + stats, then PMC counters in their
# own passes (never combined with sys/hip traces), all on the same bench command.
# usage: tools/profile_round.sh r01
set -u
R=${1:-r01}
OUT=gpurun_out/prof_$R
rm -rf $OUT
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-workloads --no-sliced"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $BENCH > $OUT/kt.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY \
  --kernel-trace --output-format csv -d $OUT/pmc_sq -- $BENCH > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1
grep -h '^{' $OUT/kt.log | tail -1 > $OUT/bench_under_rocprof.json
find $OUT -name '*.csv' | head -20
