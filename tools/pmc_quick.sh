# usage (GPU box): [BENCH_ARGS='--workload mixed'] [TOP=8] bash tools/pmc_quick.sh <tag> <counters...>
#   one --pmc pass of the bench (1 step), per-kernel means printed
TAG=$1; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc -- python3 $R/bench.py --cpu-sample 0 --steps 1 --warmup 0 $BENCH_ARGS > $OUT/pmc.log 2>&1; echo "pmc rc=$?"
python3 - <<PY
import csv, glob, collections
f = sorted(glob.glob('$OUT/pmc/**/*counter_collection.csv', recursive=True))[-1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen=set()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].replace('void ','').split('(')[0].split('<')[0]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    if (k, r['Dispatch_Id']) not in seen: seen.add((k, r['Dispatch_Id'])); n[k]+=1
for k in sorted(agg, key=lambda k: -agg[k].get('SQ_WAVE_CYCLES', agg[k].get('SQ_INSTS_VALU',0)))[:${TOP:-4}]:
    print(k, n[k], {c: round(v/n[k]) for c, v in agg[k].items()})
PY
