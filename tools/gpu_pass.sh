O=gpurun_out/r05z5; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
for v in default pb16k pu2 pu8b; do
  if [ $v = default ]; then unset NEUMANN_GPU_LIB; else export NEUMANN_GPU_LIB=$R/neumann_amd/lib/variants/libneumann_gpu_$v.so; fi
  rm -rf /tmp/prof_p; ( cd /tmp && rocprofv3 --kernel-trace -d /tmp/prof_p -o p -- python $R/bench.py --next-rows-child > $R/$O/nr_$v.json 2>/dev/null )
  python - $(find /tmp/prof_p -name "*.db" | head -1) $v <<'PY'
import sqlite3, sys, json
db = sqlite3.connect(sys.argv[1]); v = sys.argv[2]
rows = [d for (d,) in db.execute("select end-start from kernels where name like '%pred_eval%'")]
d = json.loads([l for l in open(f'gpurun_out/r05z5/nr_{v}.json') if l.startswith('{')][-1])
print(f"{v:8s} pred_eval kernels {len(rows)}: median {sorted(rows)[len(rows)//2]/1e3:.1f} us, min {min(rows)/1e3:.1f}; filtered SIMILAR wall {d['filtered_similar_sel0.1']['ms_per_query_wall']:.4f} ms")
PY
done > $O/pred_unroll_ab.txt 2>&1
cat $O/pred_unroll_ab.txt
