#!/bin/bash
# round 4 exactness soak of the final tree: every answer certified with the exact kernels (tools/soak.py).  The round-3 shapes, the shapes
# whose sweeps now run on 768 waves (0.25 .. 2 GiB), and the opt-in queries-in-LDS sweep (NMN_I8B=1) on 10M x 768 batches.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04s_soak
mkdir -p $O
cd $R
timeout 900 python tools/soak.py --rows 10000000 --dim 768 --queries 64 --out $O/soak_10Mx768.json > $O/soak_10Mx768.log 2>&1
timeout 900 python tools/soak.py --rows 10000000 --dim 768 --queries 128 --out $O/soak_10Mx768_nq128.json > $O/soak_10Mx768_nq128.log 2>&1
timeout 900 python tools/soak.py --rows 5000000 --dim 1536 --k 1000 --queries 32 --out $O/soak_5Mx1536_k1000.json > $O/soak_5Mx1536.log 2>&1
timeout 600 python tools/soak.py --rows 1000000 --dim 768 --queries 64 --out $O/soak_1Mx768_768waves.json > $O/soak_1Mx768.log 2>&1
timeout 600 python tools/soak.py --rows 2000000 --dim 768 --queries 64 --out $O/soak_2Mx768_768waves.json > $O/soak_2Mx768.log 2>&1
timeout 600 python tools/soak.py --rows 600000 --dim 1536 --k 200 --queries 64 --out $O/soak_600kx1536_768waves.json > $O/soak_600kx1536.log 2>&1
timeout 600 python tools/soak.py --rows 3000000 --dim 256 --queries 64 --out $O/soak_3Mx256_768waves.json > $O/soak_3Mx256.log 2>&1
NMN_I8B=1 timeout 900 python tools/soak.py --rows 10000000 --dim 768 --queries 64 --out $O/soak_10Mx768_i8b_optin.json > $O/soak_10Mx768_i8b.log 2>&1
for f in $O/*.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f'.split('/')[-1], 'cases', len(d['cases']), 'queries', sum(c['queries'] for c in d['cases']), 'not_certified_total', d['not_certified_total'], 'fallback_queries', sum(c['fallback_queries'] for c in d['cases']))"; done | tee $O/summary.txt
tail -2 $O/*.log | tail -24
