"""Side benchmark: BASELINE config[4] shape -- nVeto optical instructions at ~1 MHz, ~10 photons each, 120 channels -- above the engine:
(a) RawDataOptical.iter_windows (host scheduling + GPU, records per digitise window) and (b) the plugin RawRecordsFromFaxnVeto end to end
(strax_interface.py:1009-1013: chunks of raw_records_nv + truth_nv through the chunker).  Prints one JSON line; not the headline metric."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import wfsim_amd
from wfsim_amd import ministrax
from wfsim_amd.workloads import nveto_config, optical_instructions

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
ins, channels, timings = optical_instructions(n, 1000.0, 3)
out = dict(instructions=n, photons=int(len(timings)), rate_hz=1e6)

rd = wfsim_amd.RawDataOptical(nveto_config(seed=31), channels=channels, timings=timings)
list(rd.iter_windows(ins[:2000]))                         # warm-up
best = None
for rep in range(3):
    t0 = time.perf_counter()
    n_rec = n_win = 0
    for w in rd.iter_windows(ins):
        n_rec += len(w['records']); n_win += 1
    dt = time.perf_counter() - t0
    best = dt if best is None else min(best, dt)
out['rawdata_iter_windows'] = dict(seconds=round(best, 4), instructions_per_s=round(n / best), windows=n_win, records=n_rec)

best = None
for rep in range(3):
    cfg = nveto_config(seed=31, chunk_size=0.05, instructions=ins, channels=channels, timings=timings)
    plugin = wfsim_amd.RawRecordsFromFaxnVeto(cfg)
    t0 = time.perf_counter()
    res = ministrax.run_plugin(plugin)
    dt = time.perf_counter() - t0
    best = dt if best is None else min(best, dt)
    n_chunks = len(res['raw_records_nv']); n_rr = sum(len(c.data) for c in res['raw_records_nv']); n_truth = sum(len(c.data) for c in res['truth_nv'])
out['plugin_RawRecordsFromFaxnVeto'] = dict(seconds=round(best, 4), instructions_per_s=round(n / best), chunks=n_chunks, raw_records_nv=n_rr, truth_nv=n_truth)
print(json.dumps(out))
