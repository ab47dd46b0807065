# usage (GPU box): bash tools/ab_variants.sh "<label>|<env assignments>|<library or empty>" ...   [BENCH_ARGS=...] [PMC=1]
# every variant: step time and the first kernels of the bench line; with PMC=1 also VALU instructions / LDS cycles of the dominant kernel
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  IFS='|' read -r label envs lib <<< "$v"
  (
    for e in $envs; do export "$e"; done
    [ -n "$lib" ] && export WFSIM_AMD_LIB=$PWD/$lib
    python3 bench.py --steps 8 --warmup 2 --cpu-sample 0 --no-copy-ceiling ${BENCH_ARGS} 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels_ms']; print('$label', round(d['ms_per_step'],3), {x: k[x] for x in list(k)[:5]})"
    [ -n "$PMC" ] && TOP=1 bash tools/pmc_quick.sh ab_$label SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES | tail -1
  )
done
