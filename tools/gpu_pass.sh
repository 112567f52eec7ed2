O=gpurun_out/r05r; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_ivf.py tests/test_gpu_parity_basic.py tests/test_gpu_edge_cases.py tests/test_gpu_filter.py tests/test_gpu_i8_mirror.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -3 > $O/tests.txt
python bench.py --next-rows-child > $O/next_rows.json 2>/dev/null
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05r/next_rows.json') if l.startswith('{')][-1])
print({k:(v.get('value'),v.get('ms_per_query_wall'),v.get('ms_per_query_wall_32_per_call'),v.get('ms_per_query_wall_128_per_call')) for k,v in d.items() if isinstance(v,dict)})
PY
cat $O/tests.txt
