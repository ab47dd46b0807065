# timing experiments: several builds of the library (build_var_*.so), kernel times of the bench
for f in ${LIBS:-build_var_*.so}; do
  WFSIM_AMD_LIB=$PWD/$f python bench.py --steps 5 --warmup 1 --cpu-sample 0 ${BENCH_ARGS} 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels_ms']; print('$f', round(d['ms_per_step'],2), {x: k[x] for x in list(k)[:10]})"
done
