"""Edge cases of the HIP path (MI355X only): empty and degenerate batches, very large tiles, epoch-scale times,
the 10^6-sample window limit."""
import numpy as np
import pytest

import wfsim_amd
from wfsim_amd.config import xenonnt_test_config
from wfsim_amd.dtypes import instruction_dtype
from wfsim_amd.engine import WfsError
from tests.test_gpu_generation import _instructions, _run_both, _compare, MS

pytestmark = pytest.mark.gpu


def test_batch_without_photons():
    cfg = xenonnt_test_config(seed=41, s1_detection_efficiency=0.0)
    ins = _instructions([dict(type=1, time=MS * (i + 1), x=0, y=0, z=-10, amp=100) for i in range(5)])
    rd = wfsim_amd.RawData(cfg)
    assert list(rd(ins)) == [] and rd.source_finished
    sim = wfsim_amd.ChunkRawRecords(cfg)
    chunks = list(sim(ins))
    assert sum(len(c['raw_records']) for c in chunks) == 0
    # like the reference, the last chunk ends at chunk_time_pre + dt when nothing was digitised (strax_interface.py:438-439),
    # so truth rows of photon-less instructions later than that are never flushed
    assert sum(len(c['truth']) for c in chunks) == 0
    assert sim.truth_buffer['fill'].sum() == 5 and np.all(sim.truth_buffer['n_photon'][sim.truth_buffer['fill']] == 0)


def test_single_instruction_and_empty_input():
    cfg = xenonnt_test_config(seed=42)
    orc, o, eng, counts, s_ins = _run_both(cfg, _instructions([dict(type=2, time=5 * MS, x=1, y=1, z=-3, amp=7)]))
    _compare(orc, o, eng, counts, s_ins)
    rd = wfsim_amd.RawData(cfg)
    assert list(rd(_instructions([]))) == [] and rd.source_finished


def test_very_large_tiles_dense_kernel_multi_window():
    # 3e4 electrons x ~82 photons: ~5000 photons per PMT (register batches loop), deep z: ~2000 start bins (8 windows)
    cfg = xenonnt_test_config(seed=43, s2_secondary_sc_gain=100.0)
    rows = [dict(type=2, time=MS, x=0, y=0, z=-95.0, amp=30000), dict(type=2, time=3 * MS, x=5, y=5, z=-2.0, amp=20000)]
    orc, o, eng, counts, s_ins = _run_both(cfg, _instructions(rows))
    assert counts['n_photons'] > 2_000_000            # deep electrons are attenuated by the 650 us lifetime
    _compare(orc, o, eng, counts, s_ins)
    # tiles of ~5000 photons are generated tile by tile in passes and reach the dense kernel in generation order: the truth rows equal
    # the oracle's in every column, n_pe_trigger (pulse.py:255: the first n_dpe photons of the channel slice) included
    acc, ts = eng.truth()
    tr = o['truth'].reshape(-1, 12)
    for k in range(len(tr)):
        kk = int(np.argmin(np.abs(acc[:, 0] - tr[k, 0])))
        assert np.allclose(acc[kk], tr[k], rtol=1e-9), (k, acc[kk], tr[k])


def test_epoch_scale_times():
    t0 = 1_700_000_000_000_000_000          # ns since the epoch: beyond float64's integer range
    cfg = xenonnt_test_config(seed=44)
    rows = [dict(type=1, time=t0 + MS * i, x=0, y=0, z=-20, amp=2000) for i in range(4)]
    rows += [dict(type=2, time=t0 + MS * i, x=0, y=0, z=-20, amp=80) for i in range(4)]
    orc, o, eng, counts, s_ins = _run_both(cfg, _instructions(rows))
    _compare(orc, o, eng, counts, s_ins)
    rec = eng.records()
    assert rec['time'].min() > t0 - MS and rec['time'].max() < t0 + 5 * MS


def test_window_longer_than_1e6_samples_is_an_error():
    # instructions 90 us apart never leave right_raw_extension (100 us): one window of 13 ms > 10^6 samples
    cfg = xenonnt_test_config(seed=45)
    ins = _instructions([dict(type=1, time=MS + 90_000 * i, x=0, y=0, z=-10, amp=500) for i in range(150)])
    rd = wfsim_amd.RawData(cfg)
    with pytest.raises(WfsError, match='Pulse cache too long'):     # the reference asserts (rawdata.py:219)
        list(rd(ins))
    # the engine stays usable afterwards
    assert len(list(rd(ins[:20]))) > 0


def test_rows_with_more_than_64_intervals():
    # 100 S1s 90 us apart share one window (right_raw_extension is 100 us): every channel's row is ~900k samples with an
    # interval per S1 that reached it -- k_zle keeps closed intervals in lanes and flushes them 64 at a time
    cfg = xenonnt_test_config(seed=47)
    ins = _instructions([dict(type=1, time=MS + 90_000 * i, x=0, y=0, z=-10, amp=20000) for i in range(100)])
    orc, o, eng, counts, s_ins = _run_both(cfg, ins)
    z = eng.intervals()
    per_row = np.unique(z['channel'], return_counts=True)[1]
    assert per_row.max() > 64 and (per_row > 64).sum() > 10
    _compare(orc, o, eng, counts, s_ins)


def test_float_noise_array():
    """a noise array of floats with non-integral values (rawdata.py:436 stores the truncated sum into the int64 row):
    device == oracle, and different from the same array truncated on upload"""
    import wfsim_amd
    from tests.helpers import golden, make_engine, make_oracle
    from wfsim_amd.physics import instruction_params
    from wfsim_amd.resource import Resource
    from wfsim_amd.scheduler import schedule
    nz = golden('noise.npz')['noise']
    rng = np.random.default_rng(3)
    noise_f = nz.astype(np.float64) + rng.uniform(-0.9, 0.9, nz.shape)
    ins = np.zeros(12, dtype=instruction_dtype)
    ins['type'] = np.tile([1, 2], 6)
    ins['time'] = 1_000_000 + 400_000 * np.arange(12)
    ins['z'], ins['amp'], ins['recoil'], ins['event_number'] = -20.0, np.tile([3000, 200], 6), 7, np.arange(12)
    out = {}
    for tag, arr in (('float', noise_f), ('truncated', np.trunc(noise_f))):
        cfg = xenonnt_test_config(seed=12, enable_noise=True, noise_data=arr)
        res = Resource(cfg)
        order, key, cluster = schedule(ins, cfg)
        s_ins, gid = ins[order], order.astype(np.uint32)
        ip = instruction_params(s_ins, cfg, res)
        eng = make_engine(cfg, resource=res)
        eng.load_instructions(s_ins, gid, cluster, key, ip)
        eng.run()
        orc = make_oracle(cfg, resource=res)
        orc.simulate(s_ins, gid, ip)
        out[tag] = eng.records().tobytes()
        assert out[tag] == orc.pack_records().tobytes() and len(out[tag]) > 0
    assert out['float'] != out['truncated']


@pytest.mark.parametrize('resident', [True, False])
@pytest.mark.parametrize('noise_len,as_float', [(100, False), (511, True), (512, False), (700, True), (3000, False)])
def test_rows_longer_than_the_noise_table(noise_len, as_float, resident):
    """the noise index of a row wraps modulo the table length (rawdata.py:433-434) as often as the row is long: tables shorter
    than a block of samples take the general path of k_zle / k_pack, longer ones the scalar-start path; S1 + S2 pairs with PMT
    afterpulses give rows of several thousand samples -- through the accumulators and as resident rows in segments (k_row_pulse: tables of
    at least 512 samples; shorter ones keep the accumulators whatever the switch says)"""
    from tests.helpers import golden, ap_tables_from_golden
    nz = golden('noise.npz')['noise'][:noise_len]
    if as_float: nz = nz.astype(np.float64) + np.random.default_rng(5).uniform(-0.9, 0.9, nz.shape)
    cfg = xenonnt_test_config(seed=70 + noise_len, enable_noise=True, noise_data=nz, row_resident=resident)
    rows = []
    for i in range(6):
        rows += [dict(type=1, time=MS * (i + 1), x=2, y=1, z=-40, amp=4000), dict(type=2, time=MS * (i + 1), x=2, y=1, z=-40, amp=300)]
    orc, o, eng, counts, s_ins = _run_both(cfg, _instructions(rows), ap=ap_tables_from_golden())
    assert (eng.intervals()['right'] - eng.intervals()['left']).max() > 0 and counts['n_records'] > 1000
    _compare(orc, o, eng, counts, s_ins)


@pytest.mark.parametrize('resident', [True, False])
@pytest.mark.parametrize('tw', [0, 10, 31, 50, 200])
def test_trigger_windows_on_both_sides_of_the_chunk_rule(tw, resident):
    """k_zle keeps its interval state on the scalar unit when the hold-off (2 * trigger_window + 1, rawdata.py:297-308) spans a chunk
    of 64 samples, and in vector form otherwise: windows of 0, 10 (hold-off 21), 31 (63: the first scalar case), 50 (the default) and
    200 samples, with noise, against the oracle -- with and without the debug copy of the rows (which takes the vector form too)."""
    from tests.helpers import golden, make_engine, make_oracle
    from wfsim_amd.physics import instruction_params
    from wfsim_amd.resource import Resource
    from wfsim_amd.scheduler import schedule
    cfg = xenonnt_test_config(seed=90 + tw, trigger_window=tw, enable_noise=True, noise_data=golden('noise.npz')['noise'], row_resident=resident)
    rows = [dict(type=1, time=MS * (i + 1), x=i, y=-i, z=-30, amp=3000 + 700 * i) for i in range(5)]
    rows += [dict(type=2, time=MS * (i + 1), x=i, y=-i, z=-30, amp=200 + 60 * i) for i in range(5)]
    ins = _instructions(rows)
    res = Resource(cfg)
    order, key, cluster = schedule(ins, cfg)
    s_ins, gid = ins[order], order.astype(np.uint32)
    ip = instruction_params(s_ins, cfg, res)
    orc = make_oracle(cfg, resource=res)
    orc.simulate(s_ins, gid, ip)
    ref = orc.pack_records().tobytes()
    assert len(ref) > 244 * 200
    for debug in (False, True):
        eng = make_engine(cfg, resource=res)
        eng.set_debug(debug)
        eng.load_instructions(s_ins, gid, cluster, key, ip)
        eng.run()
        assert eng.records().tobytes() == ref, (tw, debug)


def small_array_config(n, **kw):
    """a detector of n PMTs (upper half = top array): fewer channels than a wave has lanes"""
    c = xenonnt_test_config(**kw)
    c['gains'] = np.full(n, 2e6)
    c['n_tpc_pmts'], c['n_top_pmts'] = n, n // 2
    c['channels_bottom'] = np.arange(n // 2, n, dtype=np.int64)
    c['channel_map'] = dict(tpc=(0, n - 1), he=(500, 500 + n // 2 - 1), sum_signal=800)
    c['photon_area_distribution'] = dict(c['photon_area_distribution'], n_channels=n)
    c['s1_pattern_map'] = ['constant dummy', 0.00014 * 494 / n, [n]]
    c['s2_pattern_map'] = ['constant dummy', 0.0003 * 494 / n, [n]]
    return c


@pytest.mark.parametrize('n_pmts', [8, 20, 33])
def test_fewer_channels_than_lanes(n_pmts):
    """the alias cells of a channel row with 2^lg < 64 cells (k_chan_alias): photons must stay inside the n_pmts channels and
    agree with the oracle photon by photon"""
    cfg = small_array_config(n_pmts, seed=50 + n_pmts)
    rows = [dict(type=1, time=MS * (i + 1), x=3, y=-2, z=-30, amp=3000) for i in range(4)]
    rows += [dict(type=2, time=MS * (i + 1), x=3, y=-2, z=-30, amp=150) for i in range(4)]
    orc, o, eng, counts, s_ins = _run_both(cfg, _instructions(rows))
    _compare(orc, o, eng, counts, s_ins)
    ph = eng.photons()
    assert counts['n_photons'] > 5000 and ph['ch'].min() >= 0 and ph['ch'].max() < n_pmts
    assert len(np.unique(ph['ch'])) == n_pmts


def test_generation_order_of_tiles_beyond_the_workgroup_sort():
    """Tiles of the per-electron generator with more than 4096 photons (an S2 of 3 x 10^4 electrons with a gain spread: ~6000 photons per
    PMT, and PMT afterpulses riding along): the bucketing leaves them in slot order; their order keys go through the segmented radix sort
    (k_tile_order_huge).  n_pe_trigger (pulse.py:255: the triggered photons among the FIRST n_dpe of the channel slice) then equals the
    oracle's, like every other truth column."""
    from tests.helpers import ap_tables_from_golden
    cfg = xenonnt_test_config(seed=47, s2_secondary_sc_gain=100.0, s2_gain_spread=2.0)
    rows = [dict(type=2, time=MS, x=0, y=0, z=-5.0, amp=30000), dict(type=1, time=3 * MS, x=1, y=1, z=-20, amp=4000)]
    orc, o, eng, counts, s_ins = _run_both(cfg, _instructions(rows), ap=ap_tables_from_golden())
    ph = eng.photons()
    per_tile = np.bincount(ph['ch'][ph['set_off'][0]:ph['set_off'][1]], minlength=494)
    assert per_tile.max() > 4096
    _compare(orc, o, eng, counts, s_ins)
    acc, ts = eng.truth()
    tr = o['truth'].reshape(-1, 12)
    for k in range(len(tr)):
        kk = int(np.argmin(np.abs(acc[:, 0] - tr[k, 0])))
        assert np.allclose(acc[kk], tr[k], rtol=1e-9), (k, acc[kk], tr[k])
