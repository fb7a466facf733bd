#!/bin/bash
# Run on the GPU box (via gpurun): complex128 truth of every big case on the engine's own complex128 path,
# the HIP complex64 results with 3M arithmetic on (default) and off, and the distance report.
set -u
mkdir -p gpurun_out/truth
timeout 2400 python3 tests/golden/make_c128_truth_gpu.py truth > gpurun_out/truth/truth.log 2>&1
echo "truth rc=$?"
timeout 900 python3 tests/golden/make_c128_truth_gpu.py c64 3m > gpurun_out/truth/c64_3m.log 2>&1
echo "c64 3m rc=$?"
ARTN_BITS_3M=0 ARTN_GEMM_3M=0 timeout 900 python3 tests/golden/make_c128_truth_gpu.py c64 4m > gpurun_out/truth/c64_4m.log 2>&1
echo "c64 4m rc=$?"
python3 tests/golden/make_c128_truth_gpu.py report > gpurun_out/truth/report.log 2>&1
tail -5 gpurun_out/truth/truth.log
cat gpurun_out/truth/report.md
