"""Optical-input path (RawDataOptical / nVeto, BASELINE config[4] shape) on the GPU against the CPU oracle."""
import numpy as np
import pytest

import wfsim_amd
from tests.helpers import make_oracle
from wfsim_amd.config import xenonnt_test_config
from wfsim_amd.dtypes import instruction_dtype, optical_extra_dtype

pytestmark = pytest.mark.gpu


def nveto_config(**kw):
    kw.setdefault('right_raw_extension', 2000)
    c = xenonnt_test_config(detector='XENONnT_neutron_veto', **kw)
    n = 120
    c['gains'] = np.full(n, 2e6)
    c['gains'][7] = 0.0                       # one dead PMT
    c['n_tpc_pmts'], c['n_top_pmts'] = n, 0
    c['channels_bottom'] = np.array([], dtype=np.int64)
    c['channel_map'] = dict(nveto=(2000, 2119), sum_signal=800, he=(500, 752))
    c['photon_area_distribution'] = dict(c['photon_area_distribution'], n_channels=n)
    return c


def optical_instructions(n, rate_ns, seed):
    rng = np.random.default_rng(seed)
    ins = np.zeros(n, dtype=instruction_dtype + optical_extra_dtype)
    ins['type'] = 1
    ins['time'] = 1_000_000 + np.cumsum(rng.exponential(rate_ns, n)).astype(np.int64)
    nph = rng.poisson(10, n)
    ins['_first'] = np.concatenate([[0], np.cumsum(nph)[:-1]])
    ins['_last'] = np.cumsum(nph)
    ins['amp'] = nph
    ins['event_number'] = np.arange(n)
    tot = int(nph.sum())
    channels = rng.integers(0, 120, tot)
    timings = rng.exponential(60, tot).astype(np.int64)
    timings[rng.random(tot) < 0.01] = -5          # a few photons outside the accepted window
    timings[rng.random(tot) < 0.01] = 2_000_000
    return ins, channels, timings


def test_nveto_high_rate_against_oracle():
    cfg = nveto_config(seed=31)
    ins, channels, timings = optical_instructions(3000, 1000.0, 3)       # ~1 MHz instruction rate
    keep = channels != 7                                                  # photons on the dead PMT make no pulse (pulse.py:89)
    rd = wfsim_amd.RawDataOptical(cfg, channels=channels, timings=timings)
    windows = list(rd.iter_windows(ins))
    rec = np.concatenate([w['records'] for w in windows])
    orc = make_oracle(cfg)
    orc.simulate_optical(ins, np.arange(len(ins), dtype=np.uint32), channels, timings, int(1e6))
    o = orc.results()
    assert len(windows) == len(o['dg_left'])
    assert np.array_equal([w['left'] for w in windows], o['dg_left'])
    assert np.array_equal([w['right'] for w in windows], o['dg_right'])
    assert rec.tobytes() == orc.pack_records().tobytes()
    assert len(rec) > 1000 and rec['channel'].max() < 120 and 7 not in rec['channel']


def test_nveto_chunker_output():
    cfg = nveto_config(seed=32, chunk_size=0.0005)
    ins, channels, timings = optical_instructions(2000, 1000.0, 4)
    sim = wfsim_amd.ChunkRawRecords(cfg, rawdata_generator=wfsim_amd.RawDataOptical, channels=channels, timings=timings)
    sim.truth_buffer = np.zeros(10000, dtype=instruction_dtype + optical_extra_dtype + sim.truth_dtype + [('fill', bool)])
    chunks = list(sim(ins))
    assert len(chunks) >= 3 and set(chunks[0].keys()) == {'raw_records', 'truth'}
    rr = np.concatenate([c['raw_records'] for c in chunks])
    truth = np.concatenate([c['truth'] for c in chunks])
    assert len(truth) == len(ins) and '_first' in truth.dtype.names
    assert np.all(np.diff(rr['time']) >= 0) and rr['data'].sum() > 0
    ok = (timings >= 0) & (timings < 1e6) & (channels != 7)
    assert truth['n_photon'].sum() == ok.sum()


def test_nveto_plugin():
    from wfsim_amd import ministrax
    ins, channels, timings = optical_instructions(1500, 1000.0, 5)
    cfg = nveto_config(seed=33, chunk_size=0.0005, instructions=ins, channels=channels, timings=timings)
    plugin = wfsim_amd.RawRecordsFromFaxnVeto(cfg)
    out = ministrax.run_plugin(plugin)
    rr = np.concatenate([c.data for c in out['raw_records_nv']])
    truth = np.concatenate([c.data for c in out['truth_nv']])
    assert len(out['raw_records_nv']) >= 2 and len(rr) > 500
    assert rr['channel'].min() >= 2000 and rr['channel'].max() <= 2119
    assert len(truth) == len(ins) and np.all(np.diff(rr['time']) >= 0)
