# timing experiments: WFS_DBG variants of the bench, kernel times only
for v in ${VARIANTS:-0 1 2}; do
  WFS_DBG=$v python bench.py --steps 5 --warmup 1 --cpu-sample 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels_ms']; print('dbg', $v, d['ms_per_step'], {x: k[x] for x in ('k_photon_fill','k_photon_count','k_pulse_dense','k_block_ranges')})"
done
