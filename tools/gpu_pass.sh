#!/bin/bash
# scratch: what sits between two sweeps of the 2-stream pipeline (1M x 768 f32)
R=$PWD; export TMPDIR=/tmp; cd /tmp
COMMON="--no-cpu-baseline --no-other-configs --batched 0 --callers 0 --no-mirror-legs --no-live-pmc --warmup 5 --rebuilds 1 --no-parity"
rm -rf /tmp/tr1
timeout 300 rocprofv3 --kernel-trace -d /tmp/tr1 -o tr -- python $R/bench.py $COMMON --rows 1000000 --steps 60 > /tmp/tr1.log 2>&1
DB=$(find /tmp/tr1 -name "*.db" | head -1)
python $R/tools/trace_gantt.py $DB --kernel scan_ring_kernel --skip 30 --steps 3
