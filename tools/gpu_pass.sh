#!/bin/bash
# scratch: exactness soak of the mirror sweeps after the batched sweep's epilogue change
mkdir -p gpurun_out
timeout 700 python tools/soak.py --mirror 1 --rows 10000000 --dim 768 --out gpurun_out/soak_default_10Mx768.json 2>&1 | tail -1
timeout 500 python tools/soak.py --mirror 2 --rows 5000000 --dim 1536 --k 1000 --out gpurun_out/soak_bf16_5Mx1536.json 2>&1 | tail -1
