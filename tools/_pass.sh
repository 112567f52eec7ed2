O=gpurun_out/r06e; mkdir -p $O
python -m pytest tests/test_gpu_batched.py tests/test_gpu_sweep_kind.py -m gpu -x -q > $O/tests_a.log 2>&1; grep -n "passed\|failed" $O/tests_a.log | tail -2
for r in 1 2; do
for m in 0 2 1; do
  python tools/mfma_loop.py --mirror $m --tag run$m --realloc 2 --reps 20 >> $O/mfma_ab.txt 2>&1
  NMN_NO_RUN_BOUND=1 python tools/mfma_loop.py --mirror $m --tag old$m --realloc 2 --reps 20 >> $O/mfma_ab.txt 2>&1
done; done
python tools/mfma_loop.py --mirror 0 --tag run0_128 --nq 128 --reps 20 >> $O/mfma_ab.txt 2>&1
NMN_NO_RUN_BOUND=1 python tools/mfma_loop.py --mirror 0 --tag old0_128 --nq 128 --reps 20 >> $O/mfma_ab.txt 2>&1
python tools/mfma_loop.py --mirror 0 --tag run0_l2 --metric 1 --reps 20 5000000:1536 >> $O/mfma_ab.txt 2>&1
NMN_NO_RUN_BOUND=1 python tools/mfma_loop.py --mirror 0 --tag old0_l2 --metric 1 --reps 20 5000000:1536 >> $O/mfma_ab.txt 2>&1
python tools/mfma_loop.py --mirror 0 --tag run0_k256 --k 256 --reps 20 >> $O/mfma_ab.txt 2>&1
NMN_NO_RUN_BOUND=1 python tools/mfma_loop.py --mirror 0 --tag old0_k256 --k 256 --reps 20 >> $O/mfma_ab.txt 2>&1
grep -v amdgpu.ids $O/mfma_ab.txt
python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "config3" > $O/tests_b.log 2>&1; grep -n "passed\|failed" $O/tests_b.log | tail -2
