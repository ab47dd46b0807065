"""Resident rows (k_row_pulse: one wave makes, finishes and zero-suppresses a whole (window, channel) row in LDS) against the
accumulator path (config row_resident=False: integer accumulators in HBM, k_zle, k_pack) and against the oracle: records, truth
and counts must not depend on which path made a row.  rawdata.py:231-239, 302-311, 398-458."""
import numpy as np
import pytest

from tests.helpers import make_engine, make_oracle
from wfsim_amd import workloads as W
from wfsim_amd.config import xenonnt_test_config
from wfsim_amd.dtypes import instruction_dtype
from wfsim_amd.physics import instruction_params
from wfsim_amd.resource import Resource
from wfsim_amd.scheduler import schedule

pytestmark = pytest.mark.gpu


def _run(cfg, ins, **kw):
    res = Resource(cfg)
    order, key, cluster = schedule(ins, cfg)
    s_ins, gid = ins[order], order.astype(np.uint32)
    eng = make_engine(cfg)
    eng.load_instructions(s_ins, gid, cluster, key, instruction_params(s_ins, cfg, res, device_maps=eng.device_maps))
    counts = eng.run()
    return eng, counts, (s_ins, gid, res)


def _truth(eng):
    """truth accumulators and time statistics of every pulse set.  The area sums and time moments are f64 sums over a tile's photons: their
    order is a property of the tile's class (at most 4 photons in at most 32 start bins: in photon order; up to 64 photons: the DPP tree of
    wave_sum), not of the kernel that happens to make the tile -- so the bits do not depend on how a row was digitised"""
    acc, ts = eng.truth()
    return np.asarray(acc), np.asarray(ts)


def _assert_same_truth(a, b):
    for x, y in zip(a, b):
        assert x.shape == y.shape and np.array_equal(x, y, equal_nan=True)


def _both(make_cfg, ins):
    out = []
    for on in (True, False):
        eng, counts, _ = _run(make_cfg(row_resident=on), ins)
        out.append((eng.records().tobytes(), _truth(eng), {k: counts[k] for k in ('n_records', 'n_intervals', 'n_pe', 'n_photons', 'n_rows', 'n_raw_samples', 'n_tiles')},
                    eng.kernel_times() if hasattr(eng, 'kernel_times') else {}))
    return out


@pytest.mark.parametrize('noise', [False, True])
def test_mixed_batch_same_records_either_way(noise):
    ins = W.mixed_batch(300)
    (r_on, t_on, c_on, k_on), (r_off, t_off, c_off, k_off) = _both(lambda **kw: W.mixed_config(seed=11, enable_noise=noise, **kw), ins)
    assert c_on == c_off
    assert r_on == r_off
    _assert_same_truth(t_on, t_off)
    assert c_on['n_records'] > 1000


def test_resident_kernel_is_the_one_that_runs():
    ins = W.mixed_batch(100)
    eng, counts, _ = _run(W.mixed_config(seed=5), ins)
    eng.set_profiling(True)
    eng.run()
    kt = eng.kernel_times()
    assert 'k_row_pulse' in kt and 'k_tile_assign' in kt
    eng2, _, _ = _run(W.mixed_config(seed=5, row_resident=False), ins)
    eng2.set_profiling(True)
    eng2.run()
    assert 'k_row_pulse' not in eng2.kernel_times()
    assert eng.records().tobytes() == eng2.records().tobytes()


def test_rows_on_both_sides_of_the_length_and_photon_limits():
    """S1s and small S2s (resident rows) next to S2s whose tiles hold more than a wave's photons and to long afterpulse rows
    (accumulator rows), in shared and in separate windows; against the oracle"""
    rng = np.random.default_rng(77)
    n = 60
    ins = np.zeros(n, dtype=instruction_dtype)
    ins['type'] = rng.choice([1, 2], n)
    ins['time'] = 1_000_000 + np.cumsum(rng.choice([300, 5_000, 30_000, 400_000], n)).astype(np.int64)
    ins['x'], ins['y'], ins['z'] = rng.uniform(-30, 30, n), rng.uniform(-30, 30, n), -rng.uniform(0.5, 95, n)
    ins['amp'] = np.where(ins['type'] == 1, rng.choice([3, 80, 900, 20000, 200000], n), rng.choice([1, 5, 40, 300, 2500], n))
    ins['recoil'], ins['event_number'] = 7, np.arange(n)
    ap = W.synthetic_afterpulse_tables()
    for noise in (False, True):
        kw = dict(seed=9, enable_pmt_afterpulses=True, uniform_to_pmt_ap=ap, s2_secondary_sc_gain=21.3)
        if noise:
            kw.update(enable_noise=True, noise_data=W.synthetic_noise())
        recs = []
        for on in (True, False):
            cfg = xenonnt_test_config(row_resident=on, **kw)
            eng, counts, (s_ins, gid, res) = _run(cfg, ins)
            recs.append(eng.records().tobytes())
            if on:
                orc = make_oracle(cfg, ap)
                orc.simulate(s_ins, gid, instruction_params(s_ins, cfg, res))
                assert recs[0] == orc.pack_records().tobytes()
                assert counts['n_pe'] == orc.n_pe
        assert recs[0] == recs[1]


def test_optical_instructions_same_records_either_way():
    from wfsim_amd.scheduler import schedule as sched
    out = []
    for on in (True, False):
        cfg = W.nveto_config(seed=31, row_resident=on)
        ins, channels, timings = W.optical_instructions(3000, 1000.0, 3)
        order, key, cluster = sched(ins, cfg)
        eng = make_engine(cfg)
        eng.load_optical(ins[order], order.astype(np.uint32), cluster, key, channels, timings, int(1e6))
        c = eng.run()
        out.append((eng.records().tobytes(), c['n_records'], c['n_pe']))
    assert out[0] == out[1] and out[0][1] > 100


@pytest.mark.parametrize('seg', [256, 512, 1024])
def test_rows_longer_than_the_lds_of_a_wave_are_made_in_segments(monkeypatch, seg):
    """WFS_RES_MAX_LEN shrinks the segment: most rows of a mixed batch then take several, with pulses across the seams"""
    ins = W.mixed_batch(120)
    ref, _, _ = _run(W.mixed_config(seed=21, row_resident=False), ins)
    monkeypatch.setenv('WFS_RES_MAX_LEN', str(seg))
    eng, counts, _ = _run(W.mixed_config(seed=21, row_resident=True), ins)
    assert eng.records().tobytes() == ref.records().tobytes()
    _assert_same_truth(_truth(eng), _truth(ref))


def test_chunker_with_device_sorted_records_same_chunks_either_way():
    """ChunkRawRecords asks the device for records in (time, channel) order (wfs_set_record_order: k_rec_keys, radix sort, k_pack /
    k_pack_res write to the sorted slots): chunks of raw_records and truth rows must not depend on how the rows were digitised"""
    import wfsim_amd
    ins = W.mixed_batch(150)
    ins['event_number'] = np.arange(len(ins)) // 2
    out = []
    for on in (True, False):
        sim = wfsim_amd.ChunkRawRecords(W.mixed_config(seed=13, chunk_size=0.02, row_resident=on))
        chunks = list(sim(ins))
        out.append(chunks)
        rr = np.concatenate([c['raw_records'] for c in chunks])
        assert len(rr) > 1000 and np.all(np.diff(rr['time']) >= 0)
    assert len(out[0]) == len(out[1]) >= 2
    for a, b in zip(*out):
        assert a['raw_records'].tobytes() == b['raw_records'].tobytes()
        assert len(a['truth']) == len(b['truth'])
        assert a['truth'].tobytes() == b['truth'].tobytes()
