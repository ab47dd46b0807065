"""where the time of a ChunkRawRecords run goes (cProfile): python tools/prof_chunker.py [n_instructions]"""
import cProfile, pstats, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wfsim_amd
from wfsim_amd.config import xenonnt_test_config
from wfsim_amd.dtypes import instruction_dtype
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
cfg = xenonnt_test_config(s2_secondary_sc_gain=100.0, seed=3, chunk_size=1.0)
ins = np.zeros(n, dtype=instruction_dtype)
ins['type'], ins['z'], ins['amp'], ins['recoil'] = 2, -10.0, 10_000, 7
ins['time'] = 1_000_000 * (1 + np.arange(n)); ins['event_number'] = np.arange(n)
def run():
    sim = wfsim_amd.ChunkRawRecords(cfg)
    t0 = time.perf_counter(); nrec = 0
    for chunk in sim(ins):
        nrec += len(chunk['raw_records'])
    return time.perf_counter() - t0, nrec
print('warm', run())
pr = cProfile.Profile(); pr.enable(); r = run(); pr.disable(); print('profiled', r)
pstats.Stats(pr).sort_stats('tottime').print_stats(16)
