cd ${GRAFT_REPO_ROOT:-$PWD}
for rows in 1000000 10000000; do for st in 1 2 3 4; do
python bench.py --rows $rows --streams $st --steps 60 --warmup 6 --rebuilds 1 --no-cpu-baseline --no-other-configs --batched 0 --callers 0 --no-mirror-legs --no-live-pmc --no-parity 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('rows $rows streams $st  %8.1f q/s  ms/step %.4f  kernel %.4f ms  frac %.3f' % (d['value'], d['ms_per_step'], r['avg_kernel_ms'], r['frac']))"
done; done
