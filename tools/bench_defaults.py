"""Side benchmark: the headline batch (1000 x 10^6-PE S2) through RawData.iter_windows -- the boundary that runs the electron
afterpulse pre-pass -- once as bench.py configures it and once with the reference's default switches (electron afterpulses on,
rawdata.py:194; garfield luminescence) on synthetic tables.  Prints one JSON line; kernel times show which kernels served it."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wfsim_amd
from wfsim_amd.workloads import bench_config, s2_batch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ins = s2_batch(n)
out = dict(instructions=n, reps=reps)
for tag, kw in (('bench_config', {}), ('reference_defaults', dict(reference_defaults=True)),
                ('electron_afterpulses_only', dict(enable_electron_afterpulses=True, uniform_to_ele_ap=wfsim_amd.workloads.synthetic_electron_afterpulses()))):
    rd = wfsim_amd.RawData(bench_config(seed=3, **kw))
    ms = []
    for rep in range(reps + 1):
        rd.engine.set_profiling(rep == reps)               # kernel timers on the last pass only
        t0 = time.perf_counter()
        n_rec = n_win = 0
        for w in rd.iter_windows(ins):
            n_rec += len(w['records']); n_win += 1
        ms.append((time.perf_counter() - t0) * 1e3)
    kt = {k: round(v[0], 3) for k, v in sorted(rd.engine.kernel_times().items(), key=lambda kv: -kv[1][0])[:8]}
    out[tag] = dict(ms_per_batch=round(min(ms[1:reps]) if reps > 1 else ms[-1], 2), ms_all=[round(x, 2) for x in ms], windows=n_win, records=n_rec,
                    kernel_ms_last_pass=kt)
print(json.dumps(out))
