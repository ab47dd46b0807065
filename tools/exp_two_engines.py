"""Experiment: two engines (two handles, two streams) on one GPU working on two batches at once -- does the chip finish
two headline batches faster than one after the other?   python tools/exp_two_engines.py"""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from wfsim_amd.engine import Engine
from wfsim_amd.physics import instruction_params
from wfsim_amd.resource import Resource
from wfsim_amd.scheduler import schedule

cfg = bench.bench_config(seed=3)
res = Resource(cfg)
engines = []
for k in range(2):
    ins = bench.s2_batch(1000, first_gid=1000 * k)
    order, key, cluster = schedule(ins, cfg)
    s = ins[order]
    eng = Engine(cfg, res, device=0)
    eng.load_instructions(s, (1000 * k + order).astype(np.uint32), cluster, key, instruction_params(s, cfg, res))
    eng.run()
    engines.append(eng)

def loop(eng, n):
    for _ in range(n):
        eng.run()

N = 6
t0 = time.perf_counter(); loop(engines[0], N); loop(engines[1], N); seq = time.perf_counter() - t0
t0 = time.perf_counter()
th = [threading.Thread(target=loop, args=(e, N)) for e in engines]
[t.start() for t in th]; [t.join() for t in th]
par = time.perf_counter() - t0
print(f'sequential: {1e3 * seq / (2 * N):.2f} ms per batch; two in flight: {1e3 * par / (2 * N):.2f} ms per batch')
