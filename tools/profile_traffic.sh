#!/bin/bash
# Run on the GPU box (via gpurun): HBM traffic of every kernel family of every bench leg, from rocprofv3 PMC counters
# collected as MI355X_MICROARCH.md prescribes -- FETCH_SIZE and WRITE_SIZE in SEPARATE passes, never combined with
# sys/hip traces; FETCH_SIZE doubled on gfx950 by tools/summarize_traffic.py.
# usage: tools/profile_traffic.sh r04     ->  gpurun_out/traffic_r04/<leg>/{fetch,write}/..., then
#        python3 tools/summarize_traffic.py r04   (here, after the merge)  ->  profiles/r04_traffic.json + _traffic.md
set -u
R=${1:-r05}
OUT=gpurun_out/traffic_$R
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
for spec in ${LEGS:-n30:3 n30_sparse10000:2 n53:5 n53m20b:1 n53m20b_bf16:1 n53m20bb:1 n53m20bb_bf16:1 n53m20:3 rand2:5 rand4:3 rand3:3 rand6:3 n30_c128:2 n30_sliced3:8}; do
  leg=${spec%%:*}; units=${spec##*:}
  mkdir -p $OUT/$leg
  echo $units > $OUT/$leg/units
  timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/$leg/fetch -- python3 tools/trace_leg.py $leg $units > $OUT/$leg/fetch.log 2>&1
  timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/$leg/write -- python3 tools/trace_leg.py $leg $units > $OUT/$leg/write.log 2>&1
  # keep only the counter tables (the merge back is capped at 64 MiB)
  find $OUT/$leg -type f ! -name '*counter_collection.csv' ! -name '*.log' ! -name units -delete
done
du -sh $OUT
