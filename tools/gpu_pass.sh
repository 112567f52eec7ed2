O=gpurun_out/r05w; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 900 python -m pytest tests/test_gpu_parity_basic.py tests/test_gpu_i8_mirror.py tests/test_gpu_edge_cases.py tests/test_gpu_engine.py -x -q 2>&1 | tail -3 > $O/tests.txt
for v in default default; do
  if [ $v = default ]; then unset NEUMANN_GPU_LIB; else export NEUMANN_GPU_LIB=$R/neumann_amd/lib/variants/libneumann_gpu_$v.so; fi
  for shape in "10000000 768" "5000000 1536" "3000000 2048" "20000000 256" "30000000 128"; do set -- $shape
    rm -rf /tmp/prof_i; ( cd /tmp && rocprofv3 --kernel-trace -d /tmp/prof_i -o p -- python $R/tools/search_child.py --rows $1 --dim $2 --mirror 1 --api device --reps 1 > /dev/null 2>&1 )
    python - $(find /tmp/prof_i -name "*.db" | head -1) $v $1 $2 <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); v, rows, dim = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
for name, dur in db.execute("select name, end-start from kernels where name like '%ingest_q8%'"):
    ms = dur / 1e6; gb = rows * dim * 5 / 1e9
    print(f"{v:9s} {rows} x {dim}: ingest_q8_kernel {ms:.3f} ms  ({gb:.1f} GB read + written -> {gb / ms:.2f} TB/s = {gb / ms / 8:.3f} of 8 TB/s)")
PY
  done
done > $O/ingest_q8_ab.txt 2>&1
unset NEUMANN_GPU_LIB
cat $O/tests.txt $O/ingest_q8_ab.txt
