"""BASELINE full sizes through size-independent properties (MI355X only): the headline batch (config[2]: 1000 S2
instructions of ~10^6 PE) and config[1] (10^4 S1 instructions)."""
import hashlib

import numpy as np
import pytest

from bench import s2_batch, bench_config
from tests.helpers import make_oracle
from wfsim_amd.dtypes import instruction_dtype
from wfsim_amd.engine import Engine
from wfsim_amd.physics import instruction_params
from wfsim_amd.resource import Resource
from wfsim_amd.scheduler import schedule

pytestmark = pytest.mark.gpu


def _run(eng, cfg, res, ins, gid0=0):
    order, key, cluster = schedule(ins, cfg)
    ip = instruction_params(ins[order], cfg, res)
    eng.load_instructions(ins[order], (gid0 + order).astype(np.uint32), cluster, key, ip)
    counts = eng.run()
    return counts, eng.records(), eng.groups()


def test_headline_batch_properties():
    cfg = bench_config(seed=3)
    res = Resource(cfg)
    eng = Engine(cfg, res)
    ins = s2_batch(1000, 0)
    counts, rec, groups = _run(eng, cfg, res, ins)
    # ~10^6 PE per instruction (1e4 e- * survival * 100/1.219 photons * 1.219 PE/photon)
    assert 0.85e9 < counts['n_pe'] < 0.93e9 and counts['n_groups'] == 1000
    assert abs(counts['n_pe'] / counts['n_photons'] - 1.219) < 2e-3
    # record structure (strax_interface.py:425-435)
    assert np.all(rec['dt'] == 10) and np.all(rec['baseline'] == 0) and np.all((rec['length'] > 0) & (rec['length'] <= 110))
    last = rec['record_i'] == (rec['pulse_length'] - 1) // 110
    assert np.all(rec['length'][~last] == 110)
    assert np.all(rec['length'][last] == rec['pulse_length'][last] - 110 * rec['record_i'][last])
    assert np.all(rec['pulse_length'] % 2 == 1)                 # even landing of both interval ends (rawdata.py:307-308)
    assert rec['data'].min() >= 0 and rec['data'].max() < 2 ** 14      # negative SPE charges push a little above the baseline
    # windows are time ordered and records stay inside their window
    first = np.append(groups['first_record'], len(rec))
    assert np.all(np.diff(groups['left']) > 0)
    for g in (0, 499, 999):
        r = rec[first[g]:first[g + 1]]
        assert r['time'].min() >= 10 * groups['left'][g] and (r['time'] + 10 * r['length']).max() <= 10 * (groups['right'][g] + 1)
        assert np.all(np.diff(r['channel']) >= 0)                # channel ascending inside a window (rawdata.py:282)
    # determinism: the same batch again gives the same bytes
    digest = hashlib.sha1(rec.tobytes()).hexdigest()
    _, rec2, _ = _run(eng, cfg, res, ins)
    assert hashlib.sha1(rec2.tobytes()).hexdigest() == digest
    # batch invariance + oracle: instructions 500..502 alone give the same records as inside the big batch
    sub = ins[500:503]
    c3, rec3, g3 = _run(eng, cfg, res, sub, gid0=500)
    assert rec3.tobytes() == rec[first[500]:first[503]].tobytes()
    orc = make_oracle(cfg)
    order, key, cluster = schedule(sub, cfg)
    orc.simulate(sub[order], (500 + order).astype(np.uint32), instruction_params(sub[order], cfg, res))
    assert orc.pack_records().tobytes() == rec3.tobytes()


def test_headline_batch_with_pmt_afterpulses():
    """The headline shape with PMT afterpulses on (300 instructions, ~3 x 10^8 PE): the afterpulses of tile-generated tiles are screened
    inside the pulse workgroup and placed tile by tile (k_s2_tile<.., AP>, k_ap_seg).  Afterpulse fraction, determinism, batch
    invariance, and three instructions against the oracle byte for byte."""
    from tests.helpers import ap_tables_from_golden
    cfg = bench_config(seed=3, pmt_afterpulses=True)
    res = Resource(cfg)
    eng = Engine(cfg, res)
    ins = s2_batch(300, 0)
    counts, rec, groups = _run(eng, cfg, res, ins)
    plain = Engine(bench_config(seed=3), Resource(bench_config(seed=3)))
    c0, rec0, _ = _run(plain, bench_config(seed=3), Resource(bench_config(seed=3)), ins)
    # same primaries (the afterpulse draws have their own sites); He + Xe + Uniform: ~4 % of the photons make an afterpulse
    n_ap = counts['n_photons'] - c0['n_photons']
    assert 0.030 < n_ap / c0['n_photons'] < 0.055 and counts['n_groups'] == 300
    first = np.append(groups['first_record'], len(rec))
    digest = hashlib.sha1(rec.tobytes()).hexdigest()
    _, rec2, _ = _run(eng, cfg, res, ins)
    assert hashlib.sha1(rec2.tobytes()).hexdigest() == digest
    sub = ins[150:153]
    c3, rec3, g3 = _run(eng, cfg, res, sub, gid0=150)
    assert rec3.tobytes() == rec[first[150]:first[153]].tobytes()
    orc = make_oracle(cfg, ap_tables_from_golden())
    order, key, cluster = schedule(sub, cfg)
    orc.simulate(sub[order], (150 + order).astype(np.uint32), instruction_params(sub[order], cfg, res))
    assert orc.pack_records().tobytes() == rec3.tobytes()


def test_config1_s1_batch_properties():
    from wfsim_amd.config import xenonnt_test_config
    n = 10_000
    cfg = xenonnt_test_config(seed=2)
    res = Resource(cfg)
    rng = np.random.default_rng(2)
    ins = np.zeros(n, dtype=instruction_dtype)
    ins['type'], ins['amp'], ins['recoil'] = 1, 1667, 7
    ins['time'] = 1_000_000 * (1 + np.arange(n))
    r, phi = 50 * np.sqrt(rng.random(n)), rng.uniform(0, 2 * np.pi, n)
    ins['x'], ins['y'], ins['z'] = r * np.cos(phi), r * np.sin(phi), rng.uniform(-97, 0, n)
    eng = Engine(cfg, res)
    counts, rec, groups = _run(eng, cfg, res, ins)
    assert counts['n_groups'] == n and abs(counts['n_photons'] / n - 1667 * 0.12 / 1.219) < 0.5
    assert abs(counts['n_pe'] / counts['n_photons'] - 1.219) < 5e-3
    assert np.all(rec['pulse_length'] >= 99) and np.all(rec['record_i'] <= rec['pulse_length'] // 110)
    first = np.append(groups['first_record'], len(rec))
    k = 1234
    c1, rec1, _ = _run(eng, cfg, res, ins[k:k + 1], gid0=k)
    assert rec1.tobytes() == rec[first[k]:first[k + 1]].tobytes()
