"""Where does tile-local generation (k_s2_tile, a workgroup per tile) beat the block generator + stand-alone pulse kernels?
Times 1000 S2 instructions of a given size with and without it (GPU box)."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import s2_batch, bench_config
from wfsim_amd.engine import Engine
from wfsim_amd.physics import instruction_params
from wfsim_amd.resource import Resource
from wfsim_amd.scheduler import schedule

for amp in [int(a) for a in (sys.argv[1:] or [250, 500, 1000, 2000, 4000, 8000])]:
    out = []
    for tg in (True, False):
        cfg = dict(bench_config(3), tile_local_generation=tg)
        res = Resource(cfg)
        ins = s2_batch(1000, 0); ins['amp'] = amp
        order, key, cluster = schedule(ins, cfg)
        eng = Engine(cfg, res)
        eng.load_instructions(ins[order], order.astype(np.uint32), cluster, key, instruction_params(ins[order], cfg, res))
        for _ in range(2):
            eng.run()
        t0 = time.perf_counter()
        for _ in range(5):
            c = eng.run()
        out.append((time.perf_counter() - t0) / 5 * 1e3)
        del eng
    print(f'amp {amp:6d}  photons/tile {c["n_photons"] / 494e3:8.1f}  tile-local {out[0]:7.3f} ms   block generator {out[1]:7.3f} ms', flush=True)
