#!/bin/bash
# round 3 exactness soak on the final kernels: every answer certified with the exact kernels (tools/soak.py)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03s
mkdir -p $O
cd $R
timeout 900 python tools/soak.py --rows 10000000 --dim 768 --queries 64 --out $O/soak_10Mx768.json > $O/soak_10Mx768.log 2>&1
timeout 900 python tools/soak.py --rows 10000000 --dim 768 --queries 128 --out $O/soak_10Mx768_nq128.json > $O/soak_10Mx768_nq128.log 2>&1
timeout 900 python tools/soak.py --rows 5000000 --dim 1536 --k 1000 --queries 32 --out $O/soak_5Mx1536_k1000.json > $O/soak_5Mx1536.log 2>&1
timeout 600 python tools/soak.py --rows 10000000 --dim 128 --queries 128 --out $O/soak_10Mx128_nq128.json > $O/soak_10Mx128.log 2>&1
timeout 600 python tools/soak.py --rows 50000 --dim 128 --k 10 --queries 64 --out $O/soak_50kx128_single_launch.json > $O/soak_50k.log 2>&1
timeout 600 python tools/soak.py --rows 2000000 --dim 3072 --queries 64 --out $O/soak_2Mx3072.json > $O/soak_2Mx3072.log 2>&1
timeout 600 python tools/soak.py --rows 8000000 --dim 384 --queries 64 --out $O/soak_8Mx384.json > $O/soak_8Mx384.log 2>&1
timeout 600 python tools/soak.py --rows 5000000 --dim 640 --queries 64 --out $O/soak_5Mx640.json > $O/soak_5Mx640.log 2>&1
for f in $O/*.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f'.split('/')[-1], 'cases', len(d['cases']), 'queries', sum(c['queries'] for c in d['cases']), 'not_certified_total', d['not_certified_total'], 'fallback_queries', sum(c['fallback_queries'] for c in d['cases']))"; done
tail -2 $O/*.log | tail -20
