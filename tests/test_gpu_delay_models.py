"""Model variants of the photon delays on the GPU against the CPU oracle: S1 custom recoil models / optical propagation,
S2 garfield luminescence / optical propagation (delay tables per instruction and PMT array, spline term per photon).
Random instruction mixes, records byte for byte; with PMT afterpulses and with electron afterpulses (RawData end to end)."""
import os

import numpy as np
import pytest

from tests.helpers import make_engine, make_oracle, ap_tables_from_golden, golden
from tests.test_delay_models_cpu import model_resources, gas_gap_resources
from wfsim_amd.config import xenonnt_test_config
from wfsim_amd.dtypes import instruction_dtype
from wfsim_amd.physics import instruction_params
from wfsim_amd.resource import Resource
from wfsim_amd.scheduler import schedule, run_sets

pytestmark = pytest.mark.gpu

S1_MODELS = ['custom', 'simple+custom', 'simple+optical_propagation', 'custom+optical_propagation', 'simple']
S2_MODELS = [('simple', 'optical_propagation'), ('garfield', 'zero_delay'), ('garfield', 'optical_propagation'),
             ('garfield', 's2_time_spread around zero'), ('simple', 's2_time_spread around zero'),
             ('garfield_gas_gap', 'zero_delay'), ('garfield_gas_gap', 'optical_propagation')]


def _case(seed):
    rng = np.random.default_rng(seed)
    d = golden('dists_models.npz')
    lum, tm = S2_MODELS[seed % len(S2_MODELS)]
    kw = dict(s1_model_type=S1_MODELS[(seed // len(S2_MODELS)) % len(S1_MODELS)], s2_luminescence_model=lum, s2_time_model=tm,
              led_pulse_length=float(rng.choice([12.5, 33.3])), s2_secondary_sc_gain=float(rng.choice([4.0, 21.3, 100.0])),
              seed=int(rng.integers(1, 10 ** 6)), **model_resources(d))
    if lum == 'garfield_gas_gap':
        kw.update(gas_gap_resources())
    if kw['s1_model_type'] == 'simple' and (lum, tm) == ('simple', 's2_time_spread around zero'):
        kw['s1_model_type'] = 'custom'          # never the all-default combination
    if lum == 'simple' and seed % 3 == 1:          # gas gap warping: a luminescence table per position (s2.py:360-378)
        kw.update(enable_gas_gap_warping=True, gas_gap_map=(lambda xy: 0.255 + 0.0003 * xy[:, 0] - 0.0002 * xy[:, 1]))
    if rng.random() < 0.4:
        kw['save_full_truth'] = False
    ap = ap_tables_from_golden() if rng.random() < 0.4 else None
    if ap is not None:
        kw.update(enable_pmt_afterpulses=True, uniform_to_pmt_ap=ap)
    cfg = xenonnt_test_config(**kw)
    n = int(rng.integers(3, 50))
    ins = np.zeros(n, dtype=instruction_dtype)
    ins['type'] = rng.choice([1, 2], n)
    ins['time'] = np.cumsum(rng.choice([200, 3_000, 40_000, 500_000, 3_000_000], n)).astype(np.int64) + 1_000_000
    ins['x'], ins['y'], ins['z'] = rng.uniform(-30, 30, n), rng.uniform(-30, 30, n), -rng.uniform(0.5, 95, n)
    s1 = ins['type'] == 1
    ins['amp'] = np.where(s1, rng.choice([0, 1, 40, 700, 5000, 30000], n), rng.choice([0, 1, 7, 60, 400, 2500, 9000], n, p=[.1, .15, .2, .2, .2, .1, .05]))
    ins['recoil'] = rng.choice([0, 6, 7, 8, 11, 12, 20], n)
    ins['event_number'] = np.arange(n)
    return cfg, ins, ap


@pytest.mark.parametrize('seed', list(range(int(os.environ.get('WFS_RANDOM_MODELS', 49)))))
def test_model_variants_match_oracle(seed):
    cfg, ins, ap = _case(seed)
    res = Resource(cfg)
    order, key, cluster = schedule(ins, cfg)
    s_ins, gid = ins[order], order.astype(np.uint32)
    ip = instruction_params(s_ins, cfg, res)
    orc = make_oracle(cfg, ap, resource=res)
    orc.simulate(s_ins, gid, ip)
    eng = make_engine(cfg, resource=res)
    assert eng.models.active
    rs = None if cfg.get('save_full_truth', True) else run_sets(s_ins, key, cluster, cfg)[0]
    eng.load_instructions(s_ins, gid, cluster, key, ip, run_set=rs)
    counts = eng.run()
    o = orc.results()
    assert counts['n_photons'] == len(o['ph_t'])
    assert eng.records().tobytes() == orc.pack_records().tobytes()
    assert counts['n_pe'] == orc.n_pe


def test_models_change_the_result():
    """the variants are not silently ignored: same seed, default models vs variants give different records"""
    cfg, ins, ap = _case(2)
    base = dict(cfg, s1_model_type='simple', s2_luminescence_model='simple', s2_time_model='s2_time_spread around zero')
    out = []
    for c in (cfg, base):
        res = Resource(c)
        order, key, cluster = schedule(ins, c)
        eng = make_engine(c, resource=res)
        rs = None if c.get('save_full_truth', True) else run_sets(ins[order], key, cluster, c)[0]
        eng.load_instructions(ins[order], order.astype(np.uint32), cluster, key, instruction_params(ins[order], c, res), run_set=rs)
        eng.run()
        out.append(eng.records().tobytes())
    assert out[0] != out[1]


@pytest.mark.parametrize('seed', list(range(int(os.environ.get('WFS_RANDOM_MODELS_EAP', 12)))))
def test_model_variants_with_electron_afterpulses(seed):
    """RawData end to end with electron afterpulses: the pre-pass recomputes photon times (k_photon_times) from the same
    tables, batches are cut at random"""
    import wfsim_amd
    from wfsim_amd.scheduler import feedback_schedule
    rng = np.random.default_rng(7000 + seed)
    d = golden('dists_models.npz')
    edges = np.linspace(0, 150e3, 141)
    hist = np.exp(-np.arange(140) / 30.0); hist *= 3e-3 / hist.sum()
    lum, tm = S2_MODELS[(0, 1, 5, 2, 6, 3)[seed % 6]]
    extra = gas_gap_resources() if lum == 'garfield_gas_gap' else {}
    cfg = xenonnt_test_config(enable_electron_afterpulses=True, uniform_to_ele_ap=(hist, edges), seed=int(rng.integers(1, 10 ** 6)),
                              s2_secondary_sc_gain=60.0, s1_model_type=S1_MODELS[seed % 4], s2_luminescence_model=lum, s2_time_model=tm,
                              led_pulse_length=20.0, **model_resources(d), **extra)
    n = int(rng.integers(3, 20))
    ins = np.zeros(n, dtype=instruction_dtype)
    ins['type'] = rng.choice([1, 2], n)
    ins['time'] = np.cumsum(rng.choice([300, 30_000, 250_000, 2_000_000], n)).astype(np.int64) + 1_000_000
    ins['x'], ins['y'], ins['z'] = rng.uniform(-30, 30, n), rng.uniform(-30, 30, n), -rng.uniform(0.5, 95, n)
    ins['amp'] = np.where(ins['type'] == 1, rng.choice([0, 50, 900, 6000], n), rng.choice([5, 80, 600, 2000], n))
    ins['recoil'], ins['event_number'] = rng.choice([0, 7, 20], n), np.arange(n)
    rd = wfsim_amd.RawData(cfg)
    rd.max_batch_quanta = int(rng.choice([30_000, 2_000_000_000]))
    windows = list(rd.iter_windows(ins))
    rec = np.concatenate([w['records'] for w in windows]) if windows else np.zeros(0)
    sec, sec_gid, sec_base, sec_parent = rd.electron_afterpulse_instructions(ins, np.arange(n), with_parent=True)
    allins = np.concatenate([ins, sec]); gids = np.concatenate([np.arange(n), sec_gid])
    base = np.concatenate([np.zeros(n, np.uint32), sec_base]); parent = np.concatenate([np.full(n, -1), sec_parent])
    order, key, cluster, rs = feedback_schedule(allins, parent, cfg)
    res = Resource(cfg)
    orc = make_oracle(cfg, resource=res)
    orc.simulate_scheduled(allins[order], gids[order].astype(np.uint32), instruction_params(allins[order], cfg, res), base[order], cluster, key, rs)
    assert rec.tobytes() == orc.pack_records().tobytes()
