#!/bin/bash
# A/B line of the matrix-core sweep on 10M x 768: 64 and 128 queries per step (certified) + per-kernel minimum from a trace
cd ${GRAFT_REPO_ROOT:-$PWD}
for nq in 64 128; do
  python bench.py --batched $nq --steps 12 --no-other-configs --no-cpu-baseline --callers 0 --no-f32-leg --no-live-pmc 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); b=d['batched']; print('nq=$nq q/s=%.0f step_ms=%.3f sweep_ms=%.3f frac=%.3f certified=%s' % (b['value'], b['ms_per_step'], b['sweep_ms_incl_sampling_pass'], b['roofline']['frac'], b['exact_topk_certified_3_of_batch']))"
done
for a in "0 64" "0 128"; do bash tools/mfma_trace.sh $a 2>&1 | grep scan_mfma; done
