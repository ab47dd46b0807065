"""cProfile of the nVeto plugin path (RawRecordsFromFaxnVeto through ministrax.run_plugin): python tools/prof_nveto_plugin.py [n]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wfsim_amd
from wfsim_amd import ministrax
from wfsim_amd.workloads import nveto_config, optical_instructions

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
ins, channels, timings = optical_instructions(n, 1000.0, 3)


def run():
    cfg = nveto_config(seed=31, chunk_size=0.05, instructions=ins, channels=channels, timings=timings)
    plugin = wfsim_amd.RawRecordsFromFaxnVeto(cfg)
    t0 = time.perf_counter()
    res = ministrax.run_plugin(plugin)
    return time.perf_counter() - t0, sum(len(c.data) for c in res['raw_records_nv'])


print('warm', run())
pr = cProfile.Profile(); pr.enable(); r = run(); pr.disable(); print('profiled', r)
pstats.Stats(pr).sort_stats('cumulative').print_stats(40)
