#!/bin/bash
# round 4, masked 8-bit sweep (config 5) under build variants of the survivor walk
OUT=gpurun_out/r04e; mkdir -p $OUT
for m in 0.1 0.01 0.5; do
  echo "== 10M x 1536 Euclidean TOP-1000, mask $m" | tee -a $OUT/masked_walk_variants_ab.txt
  ROUNDS=2 bash tools/variant_ab.sh "--dim 1536 --metric euclidean --k 1000 --steps 30 --mask $m" default pipe6 pipe6q0 walk1024 dense8 2>&1 | tee -a $OUT/masked_walk_variants_ab.txt
done
echo "== 10M x 768 cosine TOP-100, mask 0.1" | tee -a $OUT/masked_walk_variants_ab.txt
ROUNDS=2 bash tools/variant_ab.sh "--steps 30 --mask 0.1" default pipe6 walk1024 dense8 2>&1 | tee -a $OUT/masked_walk_variants_ab.txt
