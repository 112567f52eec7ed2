#!/bin/bash
# scratch: i8 ring sweep: tests + A/B
timeout 700 python -m pytest tests/test_gpu_ring.py tests/test_gpu_i8_mirror.py -x -q 2>&1 | tail -3
for r in 1 2; do for e in ring scan; do
  unset NMN_NO_RING_I8; if [ $e = scan ]; then export NMN_NO_RING_I8=1; fi
  python tools/mfma_loop.py --nq 1 --reps 16 --realloc 2 --mirror 1 --tag $e 10000000:768 1000000:768 5000000:1536 20000000:256 2>/dev/null
done; done
