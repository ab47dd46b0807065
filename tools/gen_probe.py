"""Kernel times of the photon generator alone (wfs_set_debug bit 2) on the bench batch.
usage: [WFSIM_AMD_LIB=/path/to/variant.so] python tools/gen_probe.py [n_instructions]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import s2_batch, bench_config
from wfsim_amd.engine import Engine
from wfsim_amd.physics import instruction_params
from wfsim_amd.resource import Resource
from wfsim_amd.scheduler import schedule

M = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
cfg = bench_config(3); res = Resource(cfg)
ins = s2_batch(M, 0); order, key, cluster = schedule(ins, cfg)
eng = Engine(cfg, res, device=0)
eng.load_instructions(ins[order], order.astype(np.uint32), cluster, key, instruction_params(ins[order], cfg, res))
for it in range(3):
    eng.set_profiling(it == 2)
    eng.generate()
print(os.environ.get('WFSIM_AMD_LIB', 'default'), {k: round(v[0], 3) for k, v in sorted(eng.kernel_times().items(), key=lambda kv: -kv[1][0])[:4]})
