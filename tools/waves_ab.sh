#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$PWD}
echo "== parity with 8 waves"
NMN_MFMA_WAVES=8 timeout 600 python -m pytest tests/test_gpu_batched.py tests/test_gpu_coalesce.py -x -q -k "768 or coalesce" 2>&1 | tail -2
for w in 4 8 4 8; do for nq in 64 128; do
  NMN_MFMA_WAVES=$w bash tools/mfma_trace.sh default $nq 2>&1 | grep variant | sed "s/variant=default/waves=$w/"
done; done
for w in 4 8; do for nq in 64 128; do
  NMN_MFMA_WAVES=$w python bench.py --batched $nq --steps 12 --no-other-configs --no-cpu-baseline --callers 0 --no-f32-leg --no-live-pmc 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); b=d['batched']; print('waves=$w nq=$nq q/s=%.0f step_ms=%.3f sweep_ms=%.3f frac=%.3f certified=%s' % (b['value'], b['ms_per_step'], b['sweep_ms_incl_sampling_pass'], b['roofline']['frac'], b['exact_topk_certified_3_of_batch']))"
done; done
