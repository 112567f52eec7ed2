O=gpurun_out/r05s_soak2; mkdir -p $O
timeout 1500 python tools/soak.py --mirror 0 --rows 10000000 --dim 768 --queries 64 --out $O/soak_f32_10Mx768_b64.json > $O/soak_f32_10Mx768_b64.log 2>&1
timeout 900 python tools/soak.py --mirror 0 --rows 5000000 --dim 1536 --queries 64 --k 1000 --out $O/soak_f32_5Mx1536_k1000.json > $O/soak_f32_5Mx1536_k1000.log 2>&1
timeout 600 python tools/soak.py --mirror 0 --rows 2000000 --dim 3072 --queries 40 --corpora iid,clustered --out $O/soak_f32_2Mx3072.json > $O/soak_f32_2Mx3072.log 2>&1
timeout 600 python tools/soak.py --mirror 0 --rows 10000000 --dim 128 --queries 128 --corpora iid,duplicated --out $O/soak_f32_10Mx128_b128.json > $O/soak_f32_10Mx128_b128.log 2>&1
timeout 900 python tools/soak.py --mirror 1 --rows 10000000 --dim 768 --queries 64 --out $O/soak_default_10Mx768_b64.json > $O/soak_default_10Mx768_b64.log 2>&1
for f in $O/*.log; do tail -n 1 $f; done
