# usage (GPU box): bash tools/ab_bench.sh <tag>   -- the three bench workloads, kernel times from the bench line, into gpurun_out/<tag>/
TAG=${1:-ab}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
for w in s2 mixed nveto; do
  timeout -k 10 300 python3 bench.py --workload $w --steps 5 --warmup 2 --cpu-sample 0 > $OUT/$w.json 2> $OUT/$w.err || { echo "$w failed"; tail -5 $OUT/$w.err; exit 1; }
  python3 - $OUT/$w.json $w <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k=j['roofline'].get('kernels_ms',{})
print(sys.argv[2], round(j['ms_per_step'],3), {a:b for a,b in list(k.items())[:9]})
PY
done
