#!/bin/bash
# A/B of an environment knob on the batched sweep over index rebuilds and processes: KNOB=NMN_X [KNOBVAL=1] tools/env_ab.sh  (NQ, SHAPES, ROUNDS, REALLOC, MIRROR: 1 default / 2 bf16 / 0 f32 rows)
cd ${GRAFT_REPO_ROOT:-$PWD}
for round in $(seq 1 ${ROUNDS:-3}); do
  for mode in default $KNOB; do
    if [ $mode = default ]; then unset $KNOB; else export $KNOB=${KNOBVAL:-1}; fi
    python tools/mfma_loop.py --nq ${NQ:-64} --reps 16 --realloc ${REALLOC:-4} --mirror ${MIRROR:-1} --tag $mode ${SHAPES:-10000000:768} 2>/dev/null
  done
done | python -c "
import sys, re, collections
d = collections.defaultdict(list)
for ln in sys.stdin:
    m = re.match(r'\s*(\S+) wgs.*? (\d+x\d+) nq=(\d+).*? med (\d+\.\d+)', ln)
    if m: d[(m.group(2), m.group(3), m.group(1))].append(float(m.group(4)))
for k, v in sorted(d.items()):
    v.sort()
    print('%-14s nq=%-4s %-18s n=%d  min %.3f  median %.3f  max %.3f   all: %s' % (k[0], k[1], k[2], len(v), v[0], v[len(v)//2], v[-1], ' '.join('%.3f' % x for x in v)))
"
