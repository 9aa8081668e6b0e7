#!/bin/bash
# usage: tools/ab_bench.sh /abs/path/libA.so /abs/path/libB.so ...   (alternating bench runs on the same box)
cd /root/repo
for rep in 1 2; do
for L in "$@"; do
NRMS_HIP_LIB=$L timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); k=d['kernels']
print('$L', round(d['value']), round(d['ms_per_step'],2), ' '.join('%s=%.2f'%(n[:12],v['ms_per_step']) for n,v in k.items() if v['ms_per_step']>0.25))"
done; done
