"""cProfile of the nVeto host path (RawDataOptical.iter_windows): python tools/prof_nveto.py [n]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wfsim_amd
from wfsim_amd.workloads import nveto_config, optical_instructions

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
cfg = nveto_config(seed=31)
ins, channels, timings = optical_instructions(n, 1000.0, 3)
rd = wfsim_amd.RawDataOptical(cfg, channels=channels, timings=timings)
list(rd.iter_windows(ins[:2000]))
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
nw = sum(1 for w in rd.iter_windows(ins))
pr.disable()
dt = time.perf_counter() - t0
print(f'{n} instructions, {nw} windows, {dt:.3f} s -> {n / dt:.3e} instr/s (under the profiler)')
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
