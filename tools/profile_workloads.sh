#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats of the secondary workloads.
# usage: tools/profile_workloads.sh r01
set -u
R=${1:-r01}
OUT=gpurun_out/profw_$R
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
for w in n53 n53m20 rand2 rand4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$w -- python3 bench.py --workload $w --slices 4 --steps 2 --warmup 1 > $OUT/$w.log 2>&1
  grep -h '^{' $OUT/$w.log | tail -1 > $OUT/$w.json
done
for t in n30_sparse10000 n30_sparse100; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$t -- python3 tools/trace_sparse.py $t > $OUT/$t.log 2>&1
done
python3 tools/time_sparse.py 2>&1 | grep -v "^     " > $OUT/sparse_times.txt
find $OUT -name '*kernel_stats.csv' | head
