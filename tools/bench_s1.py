"""Side benchmark: BASELINE config[1] (10^4 S1 instructions, ~200 PE each) -- not the headline metric."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wfsim_amd.config import xenonnt_test_config
from wfsim_amd.dtypes import instruction_dtype
from wfsim_amd.engine import Engine
from wfsim_amd.physics import instruction_params
from wfsim_amd.resource import Resource
from wfsim_amd.scheduler import schedule

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
cfg = xenonnt_test_config(seed=2)
res = Resource(cfg)
rng = np.random.default_rng(2)
ins = np.zeros(n, dtype=instruction_dtype)
ins['type'], ins['amp'], ins['recoil'] = 1, 1667, 7
ins['time'] = 1_000_000 * (1 + np.arange(n))
r, phi = 50 * np.sqrt(rng.random(n)), rng.uniform(0, 2 * np.pi, n)
ins['x'], ins['y'], ins['z'] = r * np.cos(phi), r * np.sin(phi), rng.uniform(-97, 0, n)
order, key, cluster = schedule(ins, cfg)
ip = instruction_params(ins[order], cfg, res)
eng = Engine(cfg, res)
eng.load_instructions(ins[order], order.astype(np.uint32), cluster, key, ip)
eng.run()
t0 = time.perf_counter()
for _ in range(3):
    c = eng.run()
dt = (time.perf_counter() - t0) / 3
eng.set_profiling(True); eng.run()
print(f'{n} S1: {dt*1e3:.2f} ms/step, {c["n_pe"]/dt:.3e} PE/s, {c["n_records"]} records, {c["n_tiles"]} tiles')
print({k: round(v[0], 3) for k, v in sorted(eng.kernel_times().items(), key=lambda kv: -kv[1][0])[:8]})
