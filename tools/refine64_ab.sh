#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$PWD}
for round in $(seq 1 ${ROUNDS:-4}); do
  for mode in off on; do
    if [ $mode = off ]; then unset NMN_REFINE_MIN_NQ; else export NMN_REFINE_MIN_NQ=1; fi
    python tools/mfma_loop.py --nq ${NQ:-64} --reps 16 --realloc 4 --tag refine_$mode ${SHAPES:-10000000:768} 2>/dev/null
  done
done | python -c "
import sys, re, collections
d = collections.defaultdict(list)
for ln in sys.stdin:
    m = re.match(r'\s*(\S+) wgs.*? (\d+x\d+) nq=(\d+).*med (\d+\.\d+)', ln)
    if m: d[(m.group(2), m.group(3), m.group(1))].append(float(m.group(4)))
for k, v in sorted(d.items()):
    v.sort()
    print('%-14s nq=%-4s %-11s n=%d  min %.3f  median %.3f  max %.3f' % (k[0], k[1], k[2], len(v), v[0], v[len(v)//2], v[-1]))
"
