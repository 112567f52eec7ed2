#!/bin/bash
# variants x placements: every variant is timed over REALLOC index builds per process and ROUNDS processes (the sweep's time
# moves +-5 % with where the driver places the mirror and the workspace: compare distributions, not single runs)
cd ${GRAFT_REPO_ROOT:-$PWD}
for round in $(seq 1 ${ROUNDS:-2}); do
for v in "$@"; do
  if [ $v = default ]; then unset NEUMANN_GPU_LIB; else export NEUMANN_GPU_LIB=$PWD/neumann_amd/lib/variants/libneumann_gpu_$v.so; fi
  python tools/mfma_loop.py --nq ${NQ:-128} --reps 16 --realloc ${REALLOC:-4} --tag $v 2>/dev/null
done; done | python -c "
import sys, re, collections
d = collections.defaultdict(list)
for ln in sys.stdin:
    m = re.match(r'\s*(\S+) wgs.*med (\d+\.\d+)', ln)
    if m: d[m.group(1)].append(float(m.group(2)))
for k, v in d.items():
    v.sort()
    print('%-10s n=%d  min %.3f  median %.3f  max %.3f   all: %s' % (k, len(v), v[0], v[len(v)//2], v[-1], ' '.join('%.3f' % x for x in v)))
"
