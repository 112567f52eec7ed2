#!/bin/bash
# 128-query sweep with / without the mid-sweep bound refinement, over index rebuilds and processes (placement noise)
cd ${GRAFT_REPO_ROOT:-$PWD}
for round in $(seq 1 ${ROUNDS:-3}); do
  for mode in refine norefine; do
    if [ $mode = norefine ]; then export NMN_NO_REFINE=1; else unset NMN_NO_REFINE; fi
    python tools/mfma_loop.py --nq ${NQ:-128} --reps 16 --realloc ${REALLOC:-4} --tag $mode ${SHAPES:-10000000:768} 2>/dev/null
  done
done | python -c "
import sys, re, collections
d = collections.defaultdict(list)
for ln in sys.stdin:
    m = re.match(r'\s*(\S+) wgs.*? (\d+x\d+) .*med (\d+\.\d+)', ln)
    if m: d[(m.group(2), m.group(1))].append(float(m.group(3)))
for k, v in sorted(d.items()):
    v.sort()
    print('%-14s %-9s n=%d  min %.3f  median %.3f  max %.3f   all: %s' % (k[0], k[1], len(v), v[0], v[len(v)//2], v[-1], ' '.join('%.3f' % x for x in v)))
"
