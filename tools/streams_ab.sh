#!/bin/bash
# headline leg with 1 / 2 / 3 / 4 host streams on one box (same corpus build per process; two rounds)
mkdir -p gpurun_out/r02w
for round in 1 2; do
  for s in 2 3 4 1; do
    python bench.py --steps 100 --warmup 10 --streams $s --no-cpu-baseline --no-other-configs --no-f32-leg --callers 0 --no-live-pmc --no-parity --batched 0 2>/dev/null | tail -1 \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('streams', $s, 'round', $round, 'q/s %.1f' % d['value'], 'ms %.4f' % d['ms_per_step'], 'frac %.3f' % d['roofline']['frac'])"
  done
done | tee gpurun_out/r02w/streams_ab.txt
