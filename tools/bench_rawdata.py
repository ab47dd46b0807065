"""Side benchmark: the production path above the engine -- RawData.iter_windows and ChunkRawRecords -- on the headline
batch (1000 x 10^6-PE S2) and on a mixed S1+S2 run with PMT afterpulses and noise (BASELINE configs[3] shape, one GPU)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wfsim_amd
from wfsim_amd.config import xenonnt_test_config
from wfsim_amd.dtypes import instruction_dtype

which = sys.argv[1] if len(sys.argv) > 1 else 'headline'
if which == 'headline':
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    cfg = xenonnt_test_config(s2_secondary_sc_gain=100.0, seed=3, chunk_size=1.0)
    ins = np.zeros(n, dtype=instruction_dtype)
    ins['type'], ins['z'], ins['amp'], ins['recoil'] = 2, -10.0, 10_000, 7
    ins['time'] = 1_000_000 * (1 + np.arange(n))
elif which == 's1':
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
    cfg = xenonnt_test_config(seed=2, chunk_size=5.0)
    rng = np.random.default_rng(2)
    ins = np.zeros(n, dtype=instruction_dtype)
    ins['type'], ins['amp'], ins['recoil'] = 1, 1667, 7
    ins['time'] = 1_000_000 * (1 + np.arange(n))
    r, phi = 50 * np.sqrt(rng.random(n)), rng.uniform(0, 2 * np.pi, n)
    ins['x'], ins['y'], ins['z'] = r * np.cos(phi), r * np.sin(phi), rng.uniform(-97, 0, n)
else:
    n_ev = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from wfsim_amd.workloads import mixed_config
    cfg = mixed_config(seed=4, chunk_size=5.0)
    rng = np.random.default_rng(4)
    ins = np.zeros(2 * n_ev, dtype=instruction_dtype)
    ins['type'] = np.tile([1, 2], n_ev)
    ins['time'] = np.repeat(1_000_000 * (1 + np.arange(n_ev)), 2)
    r, phi = 45 * np.sqrt(rng.random(n_ev)), rng.uniform(0, 2 * np.pi, n_ev)
    ins['x'], ins['y'], ins['z'] = np.repeat(r * np.cos(phi), 2), np.repeat(r * np.sin(phi), 2), np.repeat(-rng.uniform(1, 95, n_ev), 2)
    ins['amp'] = np.tile([3000, 1500], n_ev)
    ins['recoil'] = 7
ins['event_number'] = np.arange(len(ins))

for rep in range(2):
    rd = wfsim_amd.RawData(cfg)
    t0 = time.perf_counter()
    nrec = nwin = 0
    for w in rd.iter_windows(ins):
        nrec += len(w['records']); nwin += 1
    dt = time.perf_counter() - t0
    print(f'RawData.iter_windows: {len(ins)} instructions, {nwin} windows, {nrec} records, {dt * 1e3:.1f} ms')
for rep in range(2):
    sim = wfsim_amd.ChunkRawRecords(cfg)
    if os.environ.get('BATCH_QUANTA'):
        sim.rawdata.max_batch_quanta = int(float(os.environ['BATCH_QUANTA']))
    t0 = time.perf_counter()
    nrec = ntruth = 0
    for chunk in sim(ins):
        nrec += len(chunk['raw_records']); ntruth += len(chunk['truth'])
    dt = time.perf_counter() - t0
    print(f'ChunkRawRecords: {nrec} raw_records, {ntruth} truth rows, {dt * 1e3:.1f} ms -> {len(ins) / dt:.3e} instructions/s')
