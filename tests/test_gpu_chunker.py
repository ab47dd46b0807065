"""The batch-level chunker (records sorted on the device, committed window stretches at a time) against the per-window
replay of the reference's loop: identical chunks -- boundaries, records, truth -- on TPC and nVeto runs, with chunk
sizes that cut inside batches, small batches, and a record buffer that overflows."""
import os

import numpy as np
import pytest

import wfsim_amd
from tests.helpers import ap_tables_from_golden, golden
from tests.test_gpu_optical import nveto_config, optical_instructions
from wfsim_amd.config import xenonnt_test_config
from wfsim_amd.dtypes import instruction_dtype

pytestmark = pytest.mark.gpu


def _mixed(n_ev, seed):
    rng = np.random.default_rng(seed)
    ins = np.zeros(2 * n_ev, dtype=instruction_dtype)
    ins['type'] = np.tile([1, 2], n_ev)
    ins['time'] = np.repeat(np.cumsum(rng.choice([150_000, 1_000_000, 6_000_000], n_ev)), 2).astype(np.int64) + 1_000_000
    ins['x'], ins['y'] = np.repeat(rng.uniform(-30, 30, n_ev), 2), np.repeat(rng.uniform(-30, 30, n_ev), 2)
    ins['z'] = np.repeat(-rng.uniform(1, 95, n_ev), 2)
    ins['amp'] = np.tile([2000, 300], n_ev)
    ins['recoil'], ins['event_number'] = 7, np.arange(2 * n_ev)
    return ins


def _chunks(sim_factory, ins, scalar, time_zero=None):
    if scalar:
        os.environ['WFSIM_AMD_SCALAR_CHUNKER'] = '1'
    try:
        sim = sim_factory()
        out = []
        for c in sim(ins, time_zero=time_zero):
            out.append((sim.chunk_time_pre, sim.chunk_time, {k: v.copy() for k, v in c.items()}))
        return out
    finally:
        os.environ.pop('WFSIM_AMD_SCALAR_CHUNKER', None)


def _assert_same(a, b):
    assert len(a) == len(b) and len(a) > 0
    for (p0, t0, c0), (p1, t1, c1) in zip(a, b):
        assert (p0, t0) == (p1, t1)
        assert c0.keys() == c1.keys()
        for k in c0:
            if k != 'truth':
                assert c0[k].tobytes() == c1[k].tobytes(), k
                continue
            assert len(c0[k]) == len(c1[k])
            for f in c0[k].dtype.names:         # raw_area sums f64 contributions with atomics: last-bit differences between two runs
                # (n_pe_trigger depends on the photon order inside a channel, pulse.py:255: generation order on the device, so equal)
                if c0[k][f].dtype.kind == 'f' and 'n_pe_trigger' not in f:
                    assert np.allclose(c0[k][f], c1[k][f], rtol=1e-12, atol=0, equal_nan=True), f
                else:
                    assert np.array_equal(c0[k][f], c1[k][f]), f


@pytest.mark.parametrize('chunk_size,quanta,kw', [
    (0.02, 2_000_000_000, {}),
    (0.003, 40_000, {}),
    (0.05, 2_000_000_000, dict(enable_pmt_afterpulses=True, enable_noise=True)),
    (0.004, 150_000, dict(save_full_truth=False, per_pmt_truth=True)),
    (0.006, 100_000, dict(enable_electron_afterpulses=True, s2_secondary_sc_gain=60.0)),
])
def test_tpc_chunks_identical_to_the_per_window_replay(chunk_size, quanta, kw):
    if kw.get('enable_pmt_afterpulses'):
        kw = dict(kw, uniform_to_pmt_ap=ap_tables_from_golden(), noise_data=golden('noise.npz')['noise'])
    if kw.get('enable_electron_afterpulses'):
        hist = np.exp(-np.arange(140) / 30.0)
        kw = dict(kw, uniform_to_ele_ap=(hist * 3e-3 / hist.sum(), np.linspace(0, 150e3, 141)))
    cfg = xenonnt_test_config(seed=17, chunk_size=chunk_size, **kw)
    ins = _mixed(120, 3)

    def factory():
        sim = wfsim_amd.ChunkRawRecords(cfg)
        sim.rawdata.max_batch_quanta = quanta
        return sim
    fast, slow = _chunks(factory, ins, False), _chunks(factory, ins, True)
    _assert_same(fast, slow)
    assert sum(len(c[2]['raw_records']) for c in fast) > 1000 and sum(len(c[2]['truth']) for c in fast) > 0
    if chunk_size < 0.01:
        assert len(fast) > 5


class _PulseProtocolOnly:
    """the HIP generator behind nothing but the reference's RawData protocol (rawdata.py:38-157: iterate (channel, left,
    right, data), read .left / .right / .source_finished): ChunkRawRecords then runs its per-pulse loop, the path that
    tests/test_chunker_reference.py pins on the reference's own chunker"""

    def __init__(self, config, **kwargs):
        self._rd = wfsim_amd.RawData(config, **kwargs)

    def __call__(self, instructions, truth_buffer=None, **kwargs):
        return self._rd(instructions, truth_buffer=truth_buffer, **kwargs)

    left = property(lambda self: self._rd.left)
    right = property(lambda self: self._rd.right)
    source_finished = property(lambda self: self._rd.source_finished, lambda self, v: setattr(self._rd, 'source_finished', v))


@pytest.mark.parametrize('chunk_size,quanta,kw', [
    (0.003, 40_000, {}),
    (0.02, 2_000_000_000, dict(enable_pmt_afterpulses=True)),
    (0.004, 150_000, dict(save_full_truth=False)),
])
def test_batch_chunker_identical_to_the_reference_pinned_per_pulse_loop(chunk_size, quanta, kw):
    """closes the chain reference chunker == per-pulse loop (CPU, golden replay) == batch chunker (here, on the GPU's pulses)"""
    if kw.get('enable_pmt_afterpulses'):
        kw = dict(kw, uniform_to_pmt_ap=ap_tables_from_golden())
    cfg = xenonnt_test_config(seed=23, chunk_size=chunk_size, **kw)
    ins = _mixed(90, 4)

    def batch():
        sim = wfsim_amd.ChunkRawRecords(cfg)
        sim.rawdata.max_batch_quanta = quanta
        return sim
    fast = _chunks(batch, ins, False)
    slow = _chunks(lambda: wfsim_amd.ChunkRawRecords(cfg, rawdata_generator=_PulseProtocolOnly), ins, False)
    _assert_same(fast, slow)
    assert len(fast) > (5 if chunk_size < 0.01 else 1)


def test_sorted_fast_path_equals_sort_by_time(monkeypatch):
    monkeypatch.setenv('WFSIM_AMD_CHECK_SORTED', '1')            # final_results asserts prefix == sort_by_time(mask)
    cfg = xenonnt_test_config(seed=5, chunk_size=0.01, s2_secondary_sc_gain=60.0)
    sim = wfsim_amd.ChunkRawRecords(cfg)
    n = sum(len(c['raw_records']) + len(c['raw_records_he']) for c in sim(_mixed(60, 9)))
    assert n > 1000


def test_record_buffer_overflow_flushes_like_the_reference():
    cfg = xenonnt_test_config(seed=6, chunk_size=10.0)
    ins = _mixed(80, 11)

    def factory():
        sim = wfsim_amd.ChunkRawRecords(cfg)
        sim.record_buffer = sim.record_buffer[:12000].copy()          # far too small for one chunk, large enough for any window
        sim.rawdata.max_batch_quanta = 60_000
        return sim
    fast, slow = _chunks(factory, ins, False), _chunks(factory, ins, True)
    total = sum(len(c[2]['raw_records']) + len(c[2]['raw_records_he']) for c in fast)
    assert len(fast) > 3 and total == sum(len(c[2]['raw_records']) + len(c[2]['raw_records_he']) for c in slow)
    _assert_same(fast, slow)


def test_nveto_chunks_identical():
    ins, channels, timings = optical_instructions(2500, 1000.0, 5)
    cfg = nveto_config(seed=33, chunk_size=0.0004)

    def factory():
        sim = wfsim_amd.ChunkRawRecords(cfg, rawdata_generator=wfsim_amd.RawDataOptical, channels=channels, timings=timings)
        sim.rawdata.max_batch_quanta = 3000
        return sim
    fast, slow = _chunks(factory, ins, False), _chunks(factory, ins, True)
    _assert_same(fast, slow)
    assert len(fast) > 3


@pytest.mark.parametrize('seed', list(range(int(os.environ.get('WFS_RANDOM_CHUNKER', 12)))))
def test_random_chunker_runs_identical_to_the_per_window_replay(seed):
    rng = np.random.default_rng(9000 + seed)
    kw = dict(seed=int(rng.integers(1, 10 ** 6)), chunk_size=float(rng.choice([0.0007, 0.003, 0.02, 0.3])),
              right_raw_extension=int(rng.choice([5_000, 100_000])))
    if rng.random() < 0.3:
        kw.update(enable_pmt_afterpulses=True, uniform_to_pmt_ap=ap_tables_from_golden())
    if rng.random() < 0.3:
        kw.update(enable_noise=True, noise_data=golden('noise.npz')['noise'])
    if rng.random() < 0.3:
        kw['save_full_truth'] = False
    cfg = xenonnt_test_config(**kw)
    n_ev = int(rng.integers(5, 150))
    ins = _mixed(n_ev, int(rng.integers(1, 1000)))
    ins['time'] = np.repeat(np.cumsum(rng.choice([800, 30_000, 400_000, 5_000_000], n_ev)), 2).astype(np.int64) + 1_000_000
    quanta = int(rng.choice([15_000, 200_000, 2_000_000_000]))
    # (a buffer smaller than a single window is the one place where the two forms differ: which records of that window are
    # dropped -- the reference skips pulses one at a time in channel order, the batch form keeps the earliest records)
    buf = int(rng.choice([150_000, 5_000_000]))

    def factory():
        sim = wfsim_amd.ChunkRawRecords(cfg)
        sim.rawdata.max_batch_quanta = quanta
        if buf < 5_000_000:
            sim.record_buffer = sim.record_buffer[:buf].copy()
        return sim
    _assert_same(_chunks(factory, ins, False), _chunks(factory, ins, True))


@pytest.mark.parametrize('hold', [True, False])
def test_chunks_handed_out_without_a_copy_stay_intact(monkeypatch, hold):
    """Large chunks leave as the page-locked buffer their records arrived in (ChunkRawRecords._hand_out); the chunker carries
    on in a spare buffer.  Held by the consumer (hold) they are never overwritten -- when no spare buffer is left the chunker
    copies as the reference does; dropped, their buffers go back to the pool.  Either way: the bytes of the copying path."""
    import gc
    from wfsim_amd import engine as eng
    monkeypatch.setattr(wfsim_amd.ChunkRawRecords, 'zero_copy_min_records', 50)
    gc.collect()                                # chunkers of earlier tests give their buffers back
    cfg = xenonnt_test_config(seed=8, chunk_size=0.004, s2_secondary_sc_gain=60.0, high_energy_deamplification_factor=0)
    ins = _mixed(120, 4)
    ref_sim = wfsim_amd.ChunkRawRecords(dict(cfg, zero_copy_chunks=False))
    ref_sim.rawdata.max_batch_quanta = 80_000
    ref = [c['raw_records'].copy() for c in ref_sim(ins)]
    assert len(ref) > 8
    sim = wfsim_amd.ChunkRawRecords(cfg)
    sim.rawdata.max_batch_quanta = 80_000
    assert eng.is_pooled_record_buffer(sim.record_buffer)
    held, leased, k = [], 0, 0
    for c in sim(ins):
        r = c['raw_records']
        leased += not r.flags.owndata and len(r) > 0 and isinstance(r.base, memoryview)
        if hold:
            held.append(r)
        else:
            assert r.tobytes() == ref[k].tobytes()
        k += 1
        del r, c
    assert k == len(ref) and leased >= (2 if hold else len([x for x in ref if len(x) >= 50]) // 2)
    for a, b in zip(held, ref):               # every chunk still holds what it held when it was handed out
        assert a.tobytes() == b.tobytes()
    held.clear(); del sim, ref_sim
    gc.collect()
    assert not any(slot[1] for slot in eng._RECORD_BUFFERS)       # every pooled buffer is free again
