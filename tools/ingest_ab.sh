cd ${GRAFT_REPO_ROOT:-$PWD}
for r in 1 2 3; do
for v in default plainst; do
  if [ $v = default ]; then unset NEUMANN_GPU_LIB; else export NEUMANN_GPU_LIB=$PWD/neumann_amd/lib/variants/libneumann_gpu_$v.so; fi
  python tools/ingest_bench.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_total'], d['GBps_of_f32_rows_uploaded'])"
done; done
