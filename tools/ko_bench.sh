# usage (GPU box): bash tools/ko_bench.sh  -- every build_var_*.so (and the shipped library): step time, kernel time and VALU instructions per wave of k_s2_tile
cd $GRAFT_REPO_ROOT
for f in wfsim_amd/libwfsim_amd.so ${LIBS:-build_var_*.so}; do
  export WFSIM_AMD_LIB=$PWD/$f
  python3 bench.py --steps 8 --warmup 2 --cpu-sample 0 --no-copy-ceiling ${BENCH_ARGS} 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels_ms']; print('$f', round(d['ms_per_step'],3), {x: k[x] for x in list(k)[:4]})"
  TOP=1 bash tools/pmc_quick.sh ko_$(basename $f .so) SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES | tail -1
done
