#!/bin/bash
# scratch: exactness soak of the f32 ring sweep after its two late changes
mkdir -p gpurun_out
timeout 600 python tools/soak.py --mirror 0 --rows 10000000 --dim 768 --out gpurun_out/soak_f32_10Mx768.json 2>&1 | tail -2
timeout 400 python tools/soak.py --mirror 0 --rows 10000000 --dim 128 --out gpurun_out/soak_f32_10Mx128.json 2>&1 | tail -2
timeout 400 python tools/soak.py --mirror 0 --rows 5000000 --dim 1536 --k 1000 --out gpurun_out/soak_f32_5Mx1536.json 2>&1 | tail -2
