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
from wfsim_amd.workloads import synthetic_electron_afterpulses
light = synthetic_electron_afterpulses(total=3e-5)          # 3 x 10^4 photo-ionisation electrons per 10^9 PE; the default table: 3 x 10^6
for tag, kw in (('bench_config', {}), ('reference_defaults_light', dict(reference_defaults=True, uniform_to_ele_ap=light)),
                ('reference_defaults', dict(reference_defaults=True)), ('garfield_only', dict(reference_defaults=True, enable_electron_afterpulses=False))):
    rd = wfsim_amd.RawData(bench_config(seed=3, **kw))
    ms, kt, runs = [], {}, [0]
    run = rd.engine.run

    def counted_run():
        c = run()
        runs[0] += 1
        for k, v in rd.engine.kernel_times().items():
            kt[k] = kt.get(k, 0.0) + v[0]
        return c
    for rep in range(reps + 1):
        last = rep == reps
        rd.engine.set_profiling(last)                      # kernel timers on the last pass only, summed over its engine runs
        rd.engine.run = counted_run if last else run
        t0 = time.perf_counter()
        n_rec = n_win = 0
        for w in rd.iter_windows(ins):
            n_rec += len(w['records']); n_win += 1
        ms.append((time.perf_counter() - t0) * 1e3)
    top = {k: round(v, 3) for k, v in sorted(kt.items(), key=lambda kv: -kv[1])[:10]}
    out[tag] = dict(ms_per_batch=round(min(ms[1:reps]) if reps > 1 else ms[-1], 2), ms_all=[round(x, 2) for x in ms], windows=n_win, records=n_rec,
                    engine_runs_last_pass=runs[0], kernel_ms_last_pass=top)
print(json.dumps(out))
