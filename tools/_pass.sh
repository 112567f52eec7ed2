O=gpurun_out/r06t; mkdir -p $O
B="--no-other-configs --no-cpu-baseline --callers 0 --no-live-pmc --mirror 1 --rebuilds 2 --steps 10 --no-parity"
for nq in 64 128; do for tp in 0 1; do
  if [ $tp = 1 ]; then export NMN_I8_TWO_PLANES=1; else unset NMN_I8_TWO_PLANES; fi
  python bench.py $B --batched $nq 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); b=d['batched']['mirror']
print('nq=$nq two_planes=$tp', 'q/s', [round(x) for x in b['rebuilds']], 'sweep_ms', [round(x,3) for x in b['sweep_ms_rebuilds']], 'ms_per_step', round(b['ms_per_step'],3))" | tee -a $O/pipelined_ab.txt
done; done
