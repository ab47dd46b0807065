import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from wfsim_amd.engine import Engine
from wfsim_amd.resource import Resource
from wfsim_amd.scheduler import schedule
from wfsim_amd.physics import instruction_params
torch.cuda.init()
for tl in (True, False):
    cfg = bench.bench_config(3, pmt_afterpulses=True); cfg['tile_local_generation'] = tl
    res = Resource(cfg); ins = bench.s2_batch(1000, first_gid=0)
    order, key, cluster = schedule(ins, cfg); s_ins = ins[order]; ip = instruction_params(s_ins, cfg, res)
    eng = Engine(cfg, res); eng.load_instructions(s_ins, order.astype(np.uint32), cluster, key, ip)
    for _ in range(2): eng.run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): c = eng.run()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    eng.set_profiling(True); eng.run(); kt = eng.kernel_times()
    top = sorted(kt.items(), key=lambda kv: -kv[1][0])[:9]
    print('tile_local', tl, round(dt * 1e3, 2), 'ms', c['n_photons'], [(k, round(v[0], 2)) for k, v in top])
    eng.close()
