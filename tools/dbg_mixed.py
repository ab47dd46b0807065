"""mixed workload kernel times with afterpulses / noise switched off in turn"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from wfsim_amd.engine import Engine
from wfsim_amd.physics import instruction_params
from wfsim_amd.resource import Resource
from wfsim_amd.scheduler import schedule
from wfsim_amd.config import xenonnt_test_config
for name, cfg in (('ap+noise', bench.mixed_config(3)), ('plain', xenonnt_test_config(seed=3))):
    for what in ('pairs', 's2only', 's1only'):
        ins = bench.mixed_batch(10000, 0)
        if what == 's2only': ins = ins[ins['type'] == 2]
        if what == 's1only': ins = ins[ins['type'] == 1]
        res = Resource(cfg)
        order, key, cluster = schedule(ins, cfg); s = ins[order]
        eng = Engine(cfg, res); eng.load_instructions(s, order.astype(np.uint32), cluster, key, instruction_params(s, cfg, res))
        eng.run(); eng.set_profiling(True); c = eng.run(); k = eng.kernel_times()
        top = sorted(k.items(), key=lambda kv: -kv[1][0])[:6]
        print(name, what, 'photons', c['n_photons'], 'tiles', c['n_tiles'], {a: round(b[0], 2) for a, b in top})
        eng.close()
