#!/bin/bash
# round 4: the tail of a TOP-1000 search (config 5): select phases after the deeper tile gather, the launch chain on one stream, q/s
OUT=$PWD/gpurun_out/r04j; mkdir -p $OUT; R=$PWD
for v in seltrace; do
echo "== variant $v" >> $OUT/select_phases_k1000.txt
NEUMANN_GPU_LIB=$R/neumann_amd/lib/variants/libneumann_gpu_$v.so NMN_NO_SHORT_CHAIN=1 python - >> $OUT/select_phases_k1000.txt 2>&1 <<'P'
import numpy as np
from neumann_amd import GpuFlatIndex, synth_rows
rows = 10_000_000
idx = GpuFlatIndex(1536, rows, device=0); idx.fill_synthetic(3, rows)
Q = synth_rows(5, 0, 4, 1536)
keep = np.random.default_rng(1).random(rows) < 0.1
words = (rows + 63) // 64
pad = np.zeros(words * 64, bool); pad[:rows] = keep
mask = np.packbits(pad.reshape(words, 64), axis=1, bitorder="little").view(np.uint64).reshape(words)
for i in range(3): idx.search(Q[i], 1000, 1)
for i in range(3): idx.search(Q[i], 1000, 1, mask=mask)
idx.close()
P
done
grep -a "^==\|select" $OUT/select_phases_k1000.txt | cut -c1-240
B="--rebuilds 1 --no-cpu-baseline --no-other-configs --batched 0 --callers 0 --no-live-pmc --no-parity --no-mirror-legs --dim 1536 --metric euclidean --k 1000"
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d $OUT/trace -o t -- python $R/bench.py $B --streams 1 --steps 12 --warmup 3 --mask 0.1 > /dev/null 2>&1)
python tools/trace_gantt.py $(find $OUT/trace -name "*.db" | head -1) --kernel scan_i8_kernel --skip 6 --steps 2 > $OUT/gantt_config5_mask0.1_1stream.txt 2>&1; rm -rf $OUT/trace
cat $OUT/gantt_config5_mask0.1_1stream.txt
for m in 1.0 0.5 0.1 0.01; do
  python bench.py --rebuilds 2 --no-cpu-baseline --no-other-configs --batched 0 --callers 0 --no-live-pmc --no-mirror-legs --dim 1536 --metric euclidean --k 1000 --steps 30 --mask $m 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('mask $m  %9.1f q/s  %.4f ms/step  kernel %.4f ms  frac %.3f  certified %s' % (d['value'], d['ms_per_step'], r['avg_kernel_ms'], r['frac'], d['parity']['exact_topk_certified']))" | tee -a $OUT/config5_qps.txt
done
