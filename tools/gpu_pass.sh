#!/bin/bash
# scratch: exactness soak after the survivor walk
mkdir -p gpurun_out
timeout 600 python tools/soak.py --mirror 0 --rows 10000000 --dim 768 --out gpurun_out/soak_walk_f32_10Mx768.json 2>&1 | tail -1
timeout 500 python tools/soak.py --mirror 2 --rows 5000000 --dim 1536 --k 1000 --out gpurun_out/soak_walk_bf16_5Mx1536.json 2>&1 | tail -1
timeout 500 python tools/soak.py --mirror 0 --rows 10000000 --dim 128 --out gpurun_out/soak_walk_f32_10Mx128.json 2>&1 | tail -1
