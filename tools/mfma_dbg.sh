#!/bin/bash
# bisect the matrix-core sweep: NMN_MFMA_DBG bits (1 no epilogue, 2 no score writes, 4 no tmax writes) at 64 / 128 queries
cd ${GRAFT_REPO_ROOT:-$PWD}
for nq in 64 128; do
for dbg in 0 1 2 4 6; do
  NMN_MFMA_DBG=$dbg python bench.py --batched $nq --steps 10 --no-other-configs --no-cpu-baseline --callers 0 --no-parity --no-f32-leg --no-live-pmc 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); b=d['batched']; print('nq=$nq dbg=$dbg q/s=%.0f step_ms=%.3f sweep_ms=%.3f' % (b['value'], b['ms_per_step'], b['sweep_ms_incl_sampling_pass']))"
done
done
NMN_NO_SAMPLE=1 python bench.py --batched 64 --steps 10 --no-other-configs --no-cpu-baseline --callers 0 --no-parity --no-f32-leg --no-live-pmc 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); b=d['batched']; print('nq=64 NO_SAMPLE q/s=%.0f step_ms=%.3f sweep_ms=%.3f' % (b['value'], b['ms_per_step'], b['sweep_ms_incl_sampling_pass']))"
