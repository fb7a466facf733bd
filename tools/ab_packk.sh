#!/bin/bash
# GPU box: from how many contracted bits does the packed complex64 GEMM pay?  (A/B inside one session)
O=gpurun_out/ab_packk.txt
: > $O
for mk in 10 8 7; do
  echo "== ARTN_PACKED_MIN_K=$mk" >> $O
  ARTN_PACKED_MIN_K=$mk python3 bench.py --no-cpu-baseline --steps 2 --only-workloads n53,n53m20,rand2,rand4,n30_sparse10000 2>/dev/null | python3 -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,v in l['workloads'].items():
    if 'error' in v: print(k, v['error']); continue
    print(k, round(v['value'],1), 'TF', round(v['ms'],2), 'ms', v['check']['check'], {a:round(b,2) for a,b in v['roofline']['other_kernels_ms'].items()}, v['roofline']['kernel'], round(v['roofline']['kernel_ms'],2))
" >> $O
done
cat $O
