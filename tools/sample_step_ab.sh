#!/bin/bash
# sampling pass every 32nd (default) / 64th / 128th tile: batched sweep over index rebuilds and processes (NQ, ROUNDS)
cd ${GRAFT_REPO_ROOT:-$PWD}
for round in $(seq 1 ${ROUNDS:-3}); do
  for st in 32 64 128; do
    if [ $st = 32 ]; then unset NMN_SAMPLE_STEP; else export NMN_SAMPLE_STEP=$st; fi
    python tools/mfma_loop.py --nq ${NQ:-64} --reps 16 --realloc 4 --tag step$st 2>/dev/null
  done
done | python -c "
import sys, re, collections
d = collections.defaultdict(list)
for ln in sys.stdin:
    m = re.match(r'\s*(\S+) wgs.*? (\d+x\d+) nq=(\d+).*med (\d+\.\d+)', ln)
    if m: d[(m.group(3), m.group(1))].append(float(m.group(4)))
for k, v in sorted(d.items()):
    v.sort()
    print('nq=%-4s %-8s n=%d  min %.3f  median %.3f  max %.3f' % (k[0], k[1], len(v), v[0], v[len(v)//2], v[-1]))
"
