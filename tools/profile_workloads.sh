#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats of the secondary workloads.
# usage: tools/profile_workloads.sh r01
set -u
R=${1:-r01}
OUT=gpurun_out/profw_$R
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
for w in n53 n53m20 rand2 rand4 rand3 rand6 n53m20bb; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$w -- python3 bench.py --workload $w --slices 4 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/$w.log 2>&1
  grep -h '^{' $OUT/$w.log | tail -1 > $OUT/$w.json
done
# BASELINE configs[4]: n53 m20 big-batch, fp32 and bf16 operands; plus the matrix-core counters of its GEMM kernel
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/n53m20b -- python3 bench.py --workload n53m20b --slices 2 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/n53m20b.log 2>&1
grep -h '^{' $OUT/n53m20b.log | tail -1 > $OUT/n53m20b.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/n53m20b_bf16 -- python3 bench.py --workload n53m20b --precision bf16 --slices 2 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/n53m20b_bf16.log 2>&1
grep -h '^{' $OUT/n53m20b_bf16.log | tail -1 > $OUT/n53m20b_bf16.json
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY \
  --kernel-trace --output-format csv -d $OUT/n53m20b_pmc -- python3 bench.py --workload n53m20b --slices 1 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/n53m20b_pmc.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/n53m20b_clk -- python3 bench.py --workload n53m20b --slices 1 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/n53m20b_clk.log 2>&1
# the packed-operand GEMM of the reduced-precision mode: matrix-core and clock counters
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY \
  --kernel-trace --output-format csv -d $OUT/n53m20b_bf16_pmc -- python3 bench.py --workload n53m20b --precision bf16 --slices 1 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/n53m20b_bf16_pmc.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/n53m20b_bf16_clk -- python3 bench.py --workload n53m20b --precision bf16 --slices 1 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/n53m20b_bf16_clk.log 2>&1
# round 6: BASELINE configs[4] at 65 536 bitstrings in the reduced-precision mode: kernel table + matrix-core counters
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/n53m20bb_bf16 -- python3 bench.py --workload n53m20bb --precision bf16 --slices 2 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/n53m20bb_bf16.log 2>&1
grep -h '^{' $OUT/n53m20bb_bf16.log | tail -1 > $OUT/n53m20bb_bf16.json
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY \
  --kernel-trace --output-format csv -d $OUT/n53m20bb_bf16_pmc -- python3 bench.py --workload n53m20bb --precision bf16 --slices 1 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/n53m20bb_bf16_pmc.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/n53m20bb_bf16_clk -- python3 bench.py --workload n53m20bb --precision bf16 --slices 1 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/n53m20bb_bf16_clk.log 2>&1
# the headline scheme in complex128 (artn_k_bits128 pairs + artn_k_gemm128): kernel statistics and matrix-core counters
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/n30_c128 -- python3 tools/trace_c128.py 3 > $OUT/n30_c128.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY \
  --kernel-trace --output-format csv -d $OUT/n30_c128_pmc -- python3 tools/trace_c128.py 1 > $OUT/n30_c128_pmc.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/n30_c128_clk -- python3 tools/trace_c128.py 1 > $OUT/n30_c128_clk.log 2>&1
for t in n30_sparse10000 n30_sparse100; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$t -- python3 tools/trace_sparse.py $t > $OUT/$t.log 2>&1
done
python3 tools/time_sparse.py 2>&1 | grep -v "^     " > $OUT/sparse_times.txt
find $OUT -name '*kernel_stats.csv' | head
