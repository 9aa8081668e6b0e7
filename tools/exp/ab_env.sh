#!/bin/bash
cd /root/repo
for rep in 1 2; do
for spec in "$@"; do
env $spec timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); k=d['kernels']
print('$spec', round(d['value']), round(d['ms_per_step'],2), ' '.join('%s=%.2f'%(n[:12],v['ms_per_step']) for n,v in k.items() if v['ms_per_step']>0.1))"
done; done
