"""Optical-input path (RawDataOptical / nVeto, BASELINE config[4] shape) on the GPU against the CPU oracle."""
import numpy as np
import pytest

import wfsim_amd
from tests.helpers import make_oracle
from wfsim_amd.config import xenonnt_test_config
from wfsim_amd.dtypes import instruction_dtype, optical_extra_dtype

pytestmark = pytest.mark.gpu


from wfsim_amd.workloads import nveto_config, optical_instructions      # noqa: E402  (the builders of BASELINE configs[4])


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
    # the chunks are instruction_dtype + truth fields (strax_interface.py:478), without the buffer's _first / _last columns
    assert len(truth) == len(ins) and truth.dtype == np.dtype(instruction_dtype + sim.truth_dtype)
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


def test_mc_chain_plugin_tpc_and_nveto():
    """RawRecordsFromMcChain (strax_interface.py:753-1005) with supplied instructions: both detectors follow the same
    event times; chunks of all six data types share the plugin's chunk boundaries"""
    from wfsim_amd import ministrax
    from wfsim_amd.dtypes import instruction_dtype
    n_ev = 40
    rng = np.random.default_rng(8)
    tpc = np.zeros(2 * n_ev, dtype=instruction_dtype)
    tpc['g4id'] = np.repeat(np.arange(n_ev), 2)
    tpc['type'] = np.tile([1, 2], n_ev)
    tpc['amp'] = np.tile([800, 40], n_ev)
    tpc['z'], tpc['recoil'] = -rng.uniform(1, 90, 2 * n_ev), 7
    tpc['time'] = np.tile([0, 200], n_ev)
    tpc['event_number'] = tpc['g4id']
    nv, channels, timings = optical_instructions(n_ev, 1000.0, 6)
    nv['time'], nv['g4id'] = 0, np.arange(n_ev)
    ncfg = nveto_config()
    nveto_keys = {k: ncfg[k] for k in ('gains', 'n_tpc_pmts', 'n_top_pmts', 'channel_map', 'photon_area_distribution', 'right_raw_extension')}
    cfg = xenonnt_test_config(seed=4, chunk_size=0.01, event_rate=1000.0, targets=('tpc', 'nveto'), instructions_epix=tpc,
                              instructions_nveto=nv, nveto_channels=channels, nveto_timings=timings, fax_config_nveto=nveto_keys)
    cfg['channel_map'] = dict(cfg['channel_map'], nveto=(2000, 2119))          # straxen's map holds all detectors
    plugin = wfsim_amd.RawRecordsFromMcChain(cfg)
    out = ministrax.run_plugin(plugin)
    rr = np.concatenate([c.data for c in out['raw_records']])
    rr_nv = np.concatenate([c.data for c in out['raw_records_nv']])
    truth = np.concatenate([c.data for c in out['truth']])
    truth_nv = np.concatenate([c.data for c in out['truth_nv']])
    assert len(rr) > 100 and len(rr_nv) > 50 and len(truth) == 2 * n_ev and len(truth_nv) == n_ev
    assert rr['channel'].max() < 494 and rr_nv['channel'].min() >= 2000 and rr_nv['channel'].max() <= 2119
    assert np.all(np.diff(rr['time']) >= 0) and np.all(np.diff(rr_nv['time']) >= 0)
    # both detectors saw event g at the same time (the nVeto photons arrive within ~100 ns, S1 photons within ~200 ns)
    ev_t = plugin.event_times
    for g in (0, 7, 39):
        assert np.abs(rr['time'] - ev_t[g]).min() < 2000 and np.abs(rr_nv['time'] - ev_t[g]).min() < 2000
    # ---- the records of both detectors are the oracle's for the synchronised instructions (a chunk is sorted by time, ties by channel)
    from wfsim_amd.dtypes import raw_record_dtype
    from wfsim_amd.physics import instruction_params
    from wfsim_amd.resource import Resource
    from wfsim_amd.scheduler import schedule

    def by_time(x):
        return x[np.lexsort((x['channel'], x['time']))]
    epix = plugin.instructions_epix
    order, key, cluster = schedule(epix, plugin.config)
    orc = make_oracle(plugin.config)
    orc.simulate(epix[order], order.astype(np.uint32), instruction_params(epix[order], plugin.config, Resource(plugin.config)))
    ref = np.frombuffer(orc.pack_records(), dtype=raw_record_dtype())
    assert by_time(rr).tobytes() == by_time(ref).tobytes()
    nvi = plugin.instructions_nveto
    assert np.array_equal(nvi['time'], ev_t[nvi['g4id'] - plugin.config['entry_start']])       # the event time replaces the (zero) nVeto instruction time
    orc_nv = make_oracle(plugin.config_nveto)
    orc_nv.simulate_optical(nvi, np.arange(len(nvi), dtype=np.uint32), plugin.nveto_channels, plugin.nveto_timings, int(1e6))
    ref_nv = np.frombuffer(orc_nv.pack_records(), dtype=raw_record_dtype()).copy()
    ref_nv['channel'] += 2000
    assert by_time(rr_nv).tobytes() == by_time(ref_nv).tobytes()
    # TPC only
    plugin = wfsim_amd.RawRecordsFromMcChain(dict(cfg, targets=('tpc',), entry_stop=None))
    out2 = ministrax.run_plugin(plugin)
    assert sum(len(c.data) for c in out2['raw_records_nv']) == 0 and sum(len(c.data) for c in out2['raw_records']) > 100
