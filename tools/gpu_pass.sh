O=gpurun_out/r05z3; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ivf.py -x -q 2>&1 | grep "passed\|failed" > $O/tests.txt
python bench.py --next-rows-child > $O/next_rows.json 2>/dev/null
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05z3/next_rows.json') if l.startswith('{')][-1])
print({k:(v.get('value'),v.get('ms_per_query_wall'),v.get('ms_per_query_wall_32_per_call'),v.get('ms_per_query_wall_128_per_call')) for k,v in d.items() if isinstance(v,dict)})
PY
cat $O/tests.txt
