"""Side benchmark: BASELINE config[4] shape -- nVeto optical instructions at ~1 MHz, ~10 photons each, 120 channels.
Reports sustained instructions/s through RawDataOptical.iter_windows (host scheduling + GPU) -- not the headline metric."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wfsim_amd
from wfsim_amd.workloads import nveto_config, optical_instructions

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
cfg = nveto_config(seed=31)
ins, channels, timings = optical_instructions(n, 1000.0, 3)
rd = wfsim_amd.RawDataOptical(cfg, channels=channels, timings=timings)
list(rd.iter_windows(ins[:2000]))                         # warm-up
t0 = time.perf_counter()
n_rec = n_win = 0
for w in rd.iter_windows(ins):
    n_rec += len(w['records']); n_win += 1
dt = time.perf_counter() - t0
print(f'{n} optical instructions ({len(timings)} photons): {dt:.3f} s -> {n / dt:.3e} instructions/s, {n_win} windows, {n_rec} records')
try:
    print({k: round(v[0], 3) for k, v in sorted(rd.engine.kernel_times().items(), key=lambda kv: -kv[1][0])[:6]})
except Exception as e:
    print('no kernel times', e)
