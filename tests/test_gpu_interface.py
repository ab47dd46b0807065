"""The strax-facing surface on the GPU (MI355X only): BASELINE config[0] (100 S1 instructions through
RawRecordsFromFaxNT), the RawData generator protocol, batching, truth rows."""
import numpy as np
import pytest

import wfsim_amd
from wfsim_amd import ministrax
from wfsim_amd.config import xenonnt_test_config
from wfsim_amd.dtypes import instruction_dtype, truth_extra_dtype
from tests.helpers import make_oracle
from wfsim_amd.physics import instruction_params
from wfsim_amd.resource import Resource
from wfsim_amd.scheduler import schedule, processing_order

pytestmark = pytest.mark.gpu
MS = 1_000_000


def _s1_instructions(n=100, amp=417):
    ins = np.zeros(n, dtype=instruction_dtype)
    ins['type'], ins['amp'], ins['z'], ins['recoil'] = 1, amp, -10.0, 7
    ins['time'] = MS * (1 + np.arange(n))
    ins['event_number'] = np.arange(n)
    return ins


def _mixed_instructions():
    rows = []
    for i in range(30):
        t = MS * (i + 1) + (i % 3) * 150_000
        rows.append((1, t, 500 + 40 * i, -5.0 - 2 * i))
        rows.append((2, t, 30 + 5 * i, -5.0 - 2 * i))
    ins = np.zeros(len(rows), dtype=instruction_dtype)
    for k, (ty, t, amp, z) in enumerate(rows):
        ins[k]['type'], ins[k]['time'], ins[k]['amp'], ins[k]['z'] = ty, t, amp, z
    ins['recoil'] = 7
    ins['event_number'] = np.arange(len(rows)) // 2
    return ins


def test_config0_plugin_sanity():
    """reference tests/test_wfsim.py:24-27 (_sanity_check) and :140-142 (truth consistency)"""
    cfg = xenonnt_test_config(seed=1, chunk_size=0.02, instructions=_s1_instructions())
    plugin = wfsim_amd.RawRecordsFromFaxNT(cfg)
    out = ministrax.run_plugin(plugin)
    rr = np.concatenate([c.data for c in out['raw_records']])
    truth = np.concatenate([c.data for c in out['truth']])
    assert len(out['raw_records']) >= 4                     # several chunks
    assert len(rr) > 0 and rr['data'].sum() > 0
    assert np.all(np.diff(rr['time']) >= 0)
    assert len(np.concatenate([c.data for c in out['raw_records_aqmon']])) == 0
    assert len(truth) == 100
    assert np.all(truth['n_photon'] > 10) and np.all(truth['n_pe'] >= truth['n_photon'])
    assert np.all(truth['n_photon_bottom'] <= truth['n_photon'])
    # every record lies inside its chunk
    for c in out['raw_records']:
        if len(c.data):
            assert c.data['time'].min() >= c.start and c.data['time'].max() <= c.end


def test_rawdata_protocol_matches_window_path():
    cfg = xenonnt_test_config(seed=11)
    ins = _mixed_instructions()
    rd = wfsim_amd.RawData(cfg)
    pulses = [(ch, l, r, d.copy(), rd.left, rd.right) for ch, l, r, d in rd(ins)]
    assert rd.source_finished
    rd2 = wfsim_amd.RawData(cfg)
    recs = np.concatenate([w['records'] for w in rd2.iter_windows(ins)])
    first = recs[recs['record_i'] == 0]
    assert len(first) == len(pulses)
    assert np.array_equal(first['channel'], [p[0] for p in pulses])
    assert np.array_equal(first['time'], [10 * p[1] for p in pulses])
    assert np.array_equal(first['pulse_length'], [p[2] - p[1] + 1 for p in pulses])
    for p in pulses[:50]:
        assert p[4] <= p[1] and p[2] <= p[5] and len(p[3]) == p[2] - p[1] + 1


def test_small_batches_give_identical_records():
    cfg = xenonnt_test_config(seed=12)
    ins = _mixed_instructions()
    rd = wfsim_amd.RawData(cfg)
    ref = [(w['left'], w['right'], w['records'].tobytes()) for w in rd.iter_windows(ins)]
    rd2 = wfsim_amd.RawData(cfg)
    rd2.max_batch_quanta = 3000           # a handful of instructions per batch: exercises the window carry / re-run
    got = [(w['left'], w['right'], w['records'].tobytes()) for w in rd2.iter_windows(ins)]
    assert len(ref) == len(got)
    assert ref == got


def test_truth_rows_against_oracle():
    cfg = xenonnt_test_config(seed=13)
    ins = _mixed_instructions()
    sim = wfsim_amd.ChunkRawRecords(cfg)
    chunks = list(sim(ins))
    truth = np.concatenate([c['truth'] for c in chunks])
    assert len(truth) == len(ins)
    # the oracle on the same Philox streams
    res = Resource(cfg)
    order, key, cluster = schedule(ins, cfg)
    s_ins = ins[order]
    ip = instruction_params(s_ins, cfg, res)
    orc = make_oracle(cfg)
    orc.simulate(s_ins, order.astype(np.uint32), ip)
    o = orc.results()
    proc = processing_order(s_ins, np.arange(len(s_ins)), cluster)
    acc = o['truth'].reshape(-1, 12)
    names = ['n_photon', 'n_pe', 'n_photon_trigger', 'n_pe_trigger', 'raw_area', 'raw_area_trigger']
    # match rows by (event_number, type)
    key_t = {(int(r['event_number']), int(r['type'])): r for r in truth}
    for k, i in enumerate(proc):
        r = key_t[(int(s_ins['event_number'][i]), int(s_ins['type'][i]))]
        for j, f in enumerate(names):
            # (n_pe_trigger too: pulse.py:255 counts the above-threshold photons among the FIRST n_dpe photons of the channel
            # slice -- the device keeps every tile in generation order, k_tile_order / the tile-local generator, as the oracle does)
            assert np.isclose(r[f], acc[k, j], rtol=1e-9), (f, r[f], acc[k, j])
            assert np.isclose(r[f + '_bottom'], acc[k, 6 + j], rtol=1e-9)
        a, b = o['call_ph_off'][k], o['call_ph_off'][k + 1]
        t = o['ph_t'][a:b]
        assert r['n_photon'] == b - a
        assert r['t_first_photon'] == t.min() and r['t_last_photon'] == t.max()
        assert abs(r['t_mean_photon'] - t.mean()) < 1e-3 and abs(r['t_sigma_photon'] - t.std()) < 1e-3
        assert r['endtime'] == t.max() + 230
        ea, eb = o['call_e_off'][k], o['call_e_off'][k + 1]
        if s_ins['type'][i] == 2:
            et = o['e_t'][ea:eb]
            assert r['n_electron'] == len(et)
            assert r['t_first_electron'] == et.min() and abs(r['t_mean_electron'] - et.mean()) < 1e-3
        else:
            assert r['n_electron'] == 0 and np.isnan(r['t_mean_electron'])


def test_per_pmt_truth_against_oracle():
    """config per_pmt_truth (strax_interface.py:77-116, pulse.py:61-66, 268-269): per-channel truth arrays instead of the
    total / bottom split; every channel's numbers against the oracle's photon list on the same Philox streams."""
    cfg = dict(xenonnt_test_config(seed=17), per_pmt_truth=True)
    ins = _mixed_instructions()
    sim = wfsim_amd.ChunkRawRecords(cfg)
    truth = np.concatenate([c['truth'] for c in sim(ins)])
    assert len(truth) == len(ins)
    assert 'n_photon_per_pmt' in truth.dtype.names and 'n_photon_bottom' not in truth.dtype.names
    n_ch = truth['n_photon_per_pmt'].shape[1]
    assert n_ch == cfg['n_tpc_pmts']
    res = Resource(cfg)
    order, key, cluster = schedule(ins, cfg)
    s_ins = ins[order]
    orc = make_oracle(cfg)
    orc.simulate(s_ins, order.astype(np.uint32), instruction_params(s_ins, cfg, res))
    o = orc.results()
    proc = processing_order(s_ins, np.arange(len(s_ins)), cluster)
    gains = np.asarray(cfg['gains'], dtype=np.float64)
    key_t = {(int(r['event_number']), int(r['type'])): r for r in truth}
    for k, i in enumerate(proc):
        r = key_t[(int(s_ins['event_number'][i]), int(s_ins['type'][i]))]
        a, b = o['call_ph_off'][k], o['call_ph_off'][k + 1]
        ch, dpe, g = o['ph_ch'][a:b].astype(np.int64), o['ph_dpe'][a:b].astype(np.int64), o['ph_gain'][a:b]
        n_ph = np.bincount(ch, minlength=n_ch)
        assert np.array_equal(r['n_photon_per_pmt'], n_ph)
        assert np.array_equal(r['n_pe_per_pmt'], n_ph + np.bincount(ch, weights=dpe, minlength=n_ch).astype(np.int64))
        area = np.bincount(ch, weights=g, minlength=n_ch) / np.where(gains[:n_ch] > 0, gains[:n_ch], 1.0)
        assert np.allclose(r['raw_area_per_pmt'], area, rtol=1e-9, atol=1e-12)
        # totals are the sums over the PMTs (pulse.py:267)
        for f in ['n_photon', 'n_pe', 'n_photon_trigger']:
            assert r[f] == r[f + '_per_pmt'].sum(), f
        assert np.isclose(r['raw_area'], r['raw_area_per_pmt'].sum(), rtol=1e-9)
        assert np.isclose(r['raw_area_trigger'], r['raw_area_trigger_per_pmt'].sum(), rtol=1e-9)
        assert np.all(r['n_photon_trigger_per_pmt'] <= r['n_photon_per_pmt'])


def test_truth_rows_of_run_sets_match_reference_summary():
    """save_full_truth=False through ChunkRawRecords: one truth row per run set with the reference's summary of the
    grouped instructions (rawdata.py:364-372: mean x/y/z, summed amp, the rest from the first instruction) --
    against the truth rows the reference wrote for the same instructions (golden chain G)"""
    from tests.helpers import golden
    d = golden('chain_runsets.npz')
    cfg = xenonnt_test_config(seed=3, save_full_truth=False)
    sim = wfsim_amd.ChunkRawRecords(cfg)
    truth = np.concatenate([c['truth'] for c in sim(d['instructions'])])
    ref = d['truth']
    assert len(truth) == len(ref) == 6
    for f in ['type', 'amp', 'event_number', 'recoil', 'g4id' if 'g4id' in ref.dtype.names else 'type']:
        assert np.array_equal(np.sort(truth[f]), np.sort(ref[f])), f
    key = lambda r: (int(r['type']), int(r['amp']), int(r['event_number']))
    got = {key(r): r for r in truth}
    for r in ref:
        g = got[key(r)]
        for f in ['x', 'y', 'z']:
            assert np.isclose(g[f], r[f]), f
        # photon / electron counts are random variates: same distribution (Poisson-ish), compared loosely
        assert abs(g['n_photon'] - r['n_photon']) < 6 * np.sqrt(r['n_photon'] + 25)
        if r['type'] == 2:
            assert abs(g['n_electron'] - r['n_electron']) < 6 * np.sqrt(r['n_electron'] + 9)


def _ele_ap_config(**kw):
    # a synthetic delay-time histogram (the real x1t_se_afterpulse_delaytime.pkl.gz is a private resource):
    # ~2e-3 electrons per detected photon, delays up to 700 us
    edges = np.linspace(0, 700e3, 141)
    hist = 2e-3 * np.exp(-np.arange(140) / 30.0); hist *= 2e-3 / hist.sum()
    return xenonnt_test_config(enable_electron_afterpulses=True, uniform_to_ele_ap=(hist, edges), **kw)


def test_electron_afterpulses_against_oracle():
    """enable_electron_afterpulses (+ gate afterpulses): every S2 queues type-4 / type-6 secondary instructions
    (afterpulse.py:14-139) that are simulated like S2s; all type-4 instructions of a cluster share one Pulse call.
    The secondaries come from the host pre-pass; primaries + secondaries: GPU against the oracle, record for record."""
    from wfsim_amd import electron_afterpulse as ea
    from wfsim_amd.scheduler import run_sets
    cfg = _ele_ap_config(seed=23, enable_gate_afterpulses=True, photoelectric_p=2e-3)
    MS = 1_000_000
    ins = np.zeros(5, dtype=instruction_dtype)
    ins['type'] = [1, 2, 2, 1, 2]
    ins['time'] = [MS, MS, 3 * MS, 3 * MS + 50_000, 6 * MS]
    ins['x'], ins['y'], ins['z'] = [0, 0, 5, 5, -9], [0, 0, -3, -3, 4], [-20, -20, -60, -61, -5]
    ins['amp'] = [2000, 800, 1500, 900, 300]
    ins['recoil'], ins['event_number'] = 7, np.arange(5)
    rd = wfsim_amd.RawData(cfg)
    truth = np.zeros(400, dtype=instruction_dtype + truth_extra_dtype + [('fill', bool)])
    windows = list(rd.iter_windows(ins, truth_buffer=truth))
    rec = np.concatenate([w['records'] for w in windows])
    # the secondaries, regenerated: deterministic in (seed, parent gid), independent of batching
    sec, sec_gid, sec_base = rd.electron_afterpulse_instructions(ins, np.arange(5))
    assert len(sec) > 20 and set(np.unique(sec['type'])) == {4, 6}
    assert np.all(np.isin(sec_gid, [1, 2, 4])) and np.all(sec['z'] <= 0)          # parents are the three S2s
    assert np.all(sec['amp'][sec['type'] == 6] == 1) and np.all(sec['amp'] >= 1)
    for g in (1, 2, 4):                                                            # later than the parent, by at most the histogram's range
        tz = sec['time'][(sec_gid == g) & (sec['type'] == 4)] + 1700
        assert np.all(tz >= ins['time'][g]) and np.all(tz < ins['time'][g] + 800_000)
    # oracle on the union with the same stream ids, in the order / clusters / pulse sets of the reference's feedback loop
    from wfsim_amd.scheduler import feedback_schedule
    sec2, sec_gid2, sec_base2, sec_parent = rd.electron_afterpulse_instructions(ins, np.arange(5), with_parent=True)
    assert np.array_equal(sec2, sec)                                               # reproducible
    allins = np.concatenate([ins, sec]); gids = np.concatenate([np.arange(5), sec_gid]); base = np.concatenate([np.zeros(5, np.uint32), sec_base])
    parent = np.concatenate([np.full(5, -1), sec_parent])
    order, key, cluster, rs = feedback_schedule(allins, parent, cfg)
    s_ins = allins[order]
    orc = make_oracle(cfg)
    orc.simulate_scheduled(s_ins, gids[order].astype(np.uint32), instruction_params(s_ins, cfg, Resource(cfg)), base[order], cluster, key, rs)
    o = orc.results()
    assert rec.tobytes() == orc.pack_records().tobytes()
    assert len(windows) == len(o['dg_left']) and np.array_equal([w['left'] for w in windows], o['dg_left'])
    # call structure: per dynamic cluster S1s, S2s, then ONE type-4 call and ONE type-6 call (rawdata.py:102-127)
    n_sets = int(rs.max()) + 1
    assert n_sets == len(o['call_kind']) and set(o['call_kind']) == {1, 2, 4, 5}
    # small batches (pre-pass and main pass): same windows, same bytes
    rd3 = wfsim_amd.RawData(cfg)
    rd3.max_batch_quanta = 20000
    got = [(w['left'], w['right'], w['records'].tobytes()) for w in rd3.iter_windows(ins)]
    assert got == [(w['left'], w['right'], w['records'].tobytes()) for w in windows]
    t = truth[truth['fill']]
    # one row per Pulse call, except electron-afterpulse calls that made no photon (rawdata.py:336-338)
    nph = np.diff(o['call_ph_off'])
    lost = (nph == 0) & np.isin(o['call_kind'], (4, 5))
    assert len(t) == n_sets - lost.sum() and (t['type'] == 4).sum() == ((o['call_kind'] == 4) & ~lost).sum() >= 3
    assert np.all(t['n_photon'][np.isin(t['type'], (4, 6))] > 0)
    assert np.all(t['n_electron'][t['type'] == 2] > 0) and t['n_electron'][t['type'] == 4].sum() > 10      # a lone secondary electron may be lost on the way


def test_xenon1t_detector_through_the_1t_plugin():
    """detector XENON1T (248 PMTs, no HE / aqmon streams, RawRecordsFromFax1T, strax_interface.py:716-719): records equal the
    oracle's, the plugin provides raw_records + truth only"""
    n_pmt, n_top = 248, 127
    cfg = xenonnt_test_config(detector='XENON1T', seed=12, chunk_size=0.02, n_tpc_pmts=n_pmt, n_top_pmts=n_top)
    cfg['gains'] = np.full(n_pmt, 2e6)
    cfg['gains'][[1, 130]] = 0
    cfg['channels_bottom'] = np.arange(n_top, n_pmt)
    cfg['channel_map'] = dict(tpc=(0, n_pmt - 1), sum_signal=254)
    cfg['photon_area_distribution'] = dict(cfg['photon_area_distribution'], n_channels=n_pmt)
    cfg['s1_pattern_map'] = ['constant dummy', 1, [n_pmt]]
    cfg['s2_pattern_map'] = ['constant dummy', 1, [n_pmt]]
    ins = _s1_instructions()[:40]
    ins['type'][1::2] = 2
    ins['amp'][1::2] = 60
    res = Resource(cfg)
    order, key, cluster = schedule(ins, cfg)
    ip = instruction_params(ins[order], cfg, res)
    orc = make_oracle(cfg)
    orc.simulate(ins[order], order.astype(np.uint32), ip)
    rd = wfsim_amd.RawData(cfg)
    rec = np.concatenate([w['records'] for w in rd.iter_windows(ins)])
    assert rec.tobytes() == orc.pack_records().tobytes()
    assert rec['channel'].max() < n_pmt and not np.isin(rec['channel'], [1, 130]).any()
    plugin = wfsim_amd.RawRecordsFromFax1T(dict(cfg, instructions=ins))
    out = ministrax.run_plugin(plugin)
    assert set(out) == {'raw_records', 'truth'}
    rr = np.concatenate([c.data for c in out['raw_records']])
    assert len(rr) == len(rec) and np.all(np.diff(rr['time']) >= 0)
    assert len(np.concatenate([c.data for c in out['truth']])) == len(ins)


def test_config3_shape_csv_instructions_afterpulses_noise(tmp_path):
    """BASELINE configs[3] on one GPU: mixed S1+S2 instructions from a CSV file through RawRecordsFromFaxNT with PMT
    afterpulses and noise on; the same records as RawData on the in-memory instructions"""
    import pandas as pd
    from tests.helpers import ap_tables_from_golden, golden
    ins = _s1_instructions()[:60]
    ins['type'][1::2] = 2
    ins['amp'][1::2] = 80
    path = str(tmp_path / 'instructions.csv')
    pd.DataFrame({k: ins[k] for k in ins.dtype.names}).to_csv(path, index=False)
    cfg = xenonnt_test_config(seed=21, chunk_size=0.02, enable_pmt_afterpulses=True, uniform_to_pmt_ap=ap_tables_from_golden(),
                              enable_noise=True, noise_data=golden('noise.npz')['noise'], fax_file=path)
    out = ministrax.run_plugin(wfsim_amd.RawRecordsFromFaxNT(cfg))
    rr = np.concatenate([c.data for c in out['raw_records']] + [c.data for c in out['raw_records_he']])
    ref = np.concatenate([w['records'] for w in wfsim_amd.RawData(cfg).iter_windows(ins)])
    assert len(rr) == len(ref) > 500
    a = rr[np.lexsort((rr['channel'], rr['time']))]
    b = ref[np.lexsort((ref['channel'], ref['time']))]
    assert a.tobytes() == b.tobytes()
    assert len(np.concatenate([c.data for c in out['truth']])) == len(ins)


def test_reference_invariant_totals_equal_per_pmt_sums():
    """the one numerical invariant the reference's own integration test holds for this path (tests/test_wfsim.py:140-142):
    with per_pmt_truth, truth[f] == sum(truth[f + '_per_pmt']) for n_pe, n_photon, raw_area; and records are non-empty with
    a positive data sum (tests/test_wfsim.py:24-27)"""
    cfg = dict(xenonnt_test_config(seed=23, enable_pmt_afterpulses=False), per_pmt_truth=True)
    ins = _mixed_instructions()
    sim = wfsim_amd.ChunkRawRecords(cfg)
    chunks = list(sim(ins))
    truth = np.concatenate([c['truth'] for c in chunks])
    rr = np.concatenate([c['raw_records'] for c in chunks])
    assert len(rr) > 0 and rr['data'].sum() > 0
    for field in 'n_pe n_photon raw_area'.split():
        assert np.all(np.isclose(truth[field], np.sum(truth[field + '_per_pmt'], axis=1))), field
    assert truth['n_photon'].sum() > 0
