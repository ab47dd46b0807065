"""GPU box: distribution of (photons, samples) over the tiles of the mixed bench batch, per pulse-kernel class."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from wfsim_amd.engine import Engine
from wfsim_amd.resource import Resource
from wfsim_amd.scheduler import schedule
from wfsim_amd.physics import instruction_params

cfg = bench.mixed_config(seed=3)
res = Resource(cfg)
ins = bench.mixed_batch(1000, first_gid=0)
order, key, cluster = schedule(ins, cfg)
s_ins = ins[order]
ip = instruction_params(s_ins, cfg, res)
eng = Engine(cfg, res)
eng.load_instructions(s_ins, order.astype(np.uint32), cluster, key, ip)
counts = eng.run()
p = eng.pulses()
n, L = p['n_photons'], p['right'] - p['left'] + 1
live = n > 0
n, L = n[live], L[live]
nb = L - 21 - (cfg.get('samples_before_pulse_center', 2) + cfg.get('samples_after_pulse_center', 20)) + 22
print('tiles', len(n), 'photons', n.sum(), 'mean L', L.mean())
tiny = (n <= 4)
wave = (n > 4) & (n <= 64)
big = n > 64
for name, m in (('n<=4', tiny), ('5..64', wave), ('>64', big)):
    if m.sum() == 0: continue
    print(name, 'tiles', m.sum(), 'photon share', n[m].sum() / n.sum(), 'n pct', np.percentile(n[m], [10, 50, 90, 99]), 'L pct', np.percentile(L[m], [10, 50, 90, 99]),
          'sum n*ceil(L/64)', (n[m] * np.ceil(L[m] / 64)).sum() / m.sum())
