#!/bin/bash
# power / clock while the batched sweeps loop (rocm-smi sampled every 0.2 s)
OUT=$PWD/gpurun_out/r04r; mkdir -p $OUT
for tag in i8b ring; do
  if [ $tag = ring ]; then export NMN_NO_I8B=1; else unset NMN_NO_I8B; fi
  python tools/mfma_loop.py --nq 64 --reps 6000 --tag $tag > $OUT/loop_$tag.txt 2>&1 &
  PID=$!
  sleep 12
  for i in $(seq 1 12); do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -i "power\|sclk\|mclk\|fclk\|junction\|socclk" | tr '\n' ' '; echo; sleep 0.3; done > $OUT/smi_$tag.txt
  wait $PID
  echo "== $tag"; grep -v amdgpu.ids $OUT/loop_$tag.txt; head -4 $OUT/smi_$tag.txt | cut -c1-600
done
