"""debug: one case of tests/test_gpu_delay_models.py, GPU photons vs oracle photons per pulse call"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.test_gpu_delay_models import _case
from tests.helpers import make_engine, make_oracle
from wfsim_amd.physics import instruction_params
from wfsim_amd.resource import Resource
from wfsim_amd.scheduler import schedule, run_sets
seed = int(sys.argv[1])
cfg, ins, ap = _case(seed)
print({k: cfg[k] for k in ('s1_model_type', 's2_luminescence_model', 's2_time_model', 'save_full_truth', 'enable_pmt_afterpulses') if k in cfg})
res = Resource(cfg)
order, key, cluster = schedule(ins, cfg)
s_ins, gid = ins[order], order.astype(np.uint32)
ip = instruction_params(s_ins, cfg, res)
orc = make_oracle(cfg, ap, resource=res); orc.simulate(s_ins, gid, ip)
eng = make_engine(cfg, resource=res)
rs = None if cfg.get('save_full_truth', True) else run_sets(s_ins, key, cluster, cfg)[0]
eng.load_instructions(s_ins, gid, cluster, key, ip, run_set=rs)
counts = eng.run(); o = orc.results(); ph = eng.photons()
print(counts['n_photons'], len(o['ph_t']), 'sets', counts['n_pulse_sets'], 'calls', len(o['call_kind']))
print('kinds', o['call_kind'][:40], 'runsets', o['call_runset'][:40])
PS = counts['n_pulse_sets'] // 2 if ap is not None else counts['n_pulse_sets']
for c in range(len(o['call_kind'])):
    a, b = o['call_ph_off'][c], o['call_ph_off'][c + 1]
    s = o['call_runset'][c] + (PS if o['call_kind'][c] == 3 else 0)
    g0, g1 = ph['set_off'][s], ph['set_off'][s + 1]
    if b - a != g1 - g0:
        print('call', c, 'kind', o['call_kind'][c], 'set', s, 'oracle', b - a, 'gpu', g1 - g0)
        continue
    ko = np.lexsort((o['ph_gain'][a:b], o['ph_t'][a:b], o['ph_ch'][a:b])); kg = np.lexsort((ph['gain'][g0:g1], ph['t'][g0:g1], ph['ch'][g0:g1]))
    bad = (o['ph_t'][a:b][ko] != ph['t'][g0:g1][kg]) | (o['ph_ch'][a:b][ko] != ph['ch'][g0:g1][kg]) | (o['ph_gain'][a:b][ko] != ph['gain'][g0:g1][kg])
    if bad.any():
        print('call', c, 'kind', o['call_kind'][c], 'set', s, 'n', b - a, 'differ', bad.sum())
        i = np.nonzero(bad)[0][:5]
        print('  oracle', o['ph_t'][a:b][ko][i], o['ph_ch'][a:b][ko][i], o['ph_gain'][a:b][ko][i], o['ph_dpe'][a:b][ko][i])
        print('  gpu   ', ph['t'][g0:g1][kg][i], ph['ch'][g0:g1][kg][i], ph['gain'][g0:g1][kg][i], ph['dpe'][g0:g1][kg][i])
print('amp', s_ins['amp'], 'type', s_ins['type'])
