import sys, os, numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from bench import s2_batch, bench_config
from wfsim_amd.engine import Engine
from wfsim_amd.physics import instruction_params
from wfsim_amd.resource import Resource
from wfsim_amd.scheduler import schedule
cfg = dict(bench_config(3), tile_local_generation=False)
res = Resource(cfg)
ins = s2_batch(1000, 0)
order, key, cluster = schedule(ins, cfg)
eng = Engine(cfg, res)
eng.load_instructions(ins[order], order.astype(np.uint32), cluster, key, instruction_params(ins[order], cfg, res))
eng.run(); eng.set_profiling(True); eng.run()
k = eng.kernel_times()
for n, v in sorted(k.items(), key=lambda kv: -kv[1][0])[:10]: print(n, round(v[0], 3), v[1])
