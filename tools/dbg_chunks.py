import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wfsim_amd
from wfsim_amd import strax_interface as si
from wfsim_amd.config import xenonnt_test_config
from wfsim_amd.dtypes import instruction_dtype
n = 4000
cfg = xenonnt_test_config(s2_secondary_sc_gain=100.0, seed=3, chunk_size=1.0)
ins = np.zeros(n, dtype=instruction_dtype)
ins['type'], ins['z'], ins['amp'], ins['recoil'] = 2, -10.0, 10_000, 7
ins['time'] = 1_000_000 * (1 + np.arange(n)); ins['event_number'] = np.arange(n)
orig = si.ChunkRawRecords._hand_out
def logged(self, n_out):
    t0 = time.perf_counter(); nb = self._next_buffer is not None; bl = self.blevel
    r = orig(self, n_out)
    print(f'hand_out n_out={n_out} blevel={bl} next_buffer={nb} moved={r[1]} {1e3 * (time.perf_counter() - t0):.1f} ms')
    return r
si.ChunkRawRecords._hand_out = logged
for rep in range(2):
    sim = wfsim_amd.ChunkRawRecords(cfg)
    t0 = time.perf_counter(); tl = t0
    for chunk in sim(ins):
        t = time.perf_counter(); print(f'  chunk {len(chunk["raw_records"])} records after {1e3 * (t - tl):.1f} ms'); tl = t
    print('total', 1e3 * (time.perf_counter() - t0))
