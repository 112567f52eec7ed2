O=gpurun_out/r06n; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/suite.log 2>&1; grep -n "passed\|failed" $O/suite.log | tail -2
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r06n/bench.json") if l.startswith("{")][-1])
r=d["roofline"]
print("value",d["value"],"ms",d["ms_per_step"])
for k in list(r)[:24]: print(k, r[k])
for k in ("c3_f32_sweep_launches","c3_i8_sweep_launches","c3_f32_sweep_ms","c3_i8_sweep_ms","c2_f32_step_frac","c5_mask0.1_f32_step_frac","frac_of_read_ceiling"): print(k, r.get(k))
print(d["config"]["filtered_similar_sel0.1_ms"]); print({k:v for k,v in d["cpu_baseline"].items() if k.startswith("pub_") and k!="pub_detail"})
PY
