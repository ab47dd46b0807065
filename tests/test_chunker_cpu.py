"""ChunkRawRecords host logic with a stand-in generator (the reference's own injection point
``rawdata_generator=``, strax_interface.py:355-359).  CPU only: no GPU, no oracle."""
import numpy as np
import pytest

from wfsim_amd.config import xenonnt_test_config
from wfsim_amd.dtypes import instruction_dtype, raw_record_dtype
from wfsim_amd.strax_interface import ChunkRawRecords


class FakeRawData:
    """Yields deterministic pulses: one digitise window per instruction, a few channels each."""

    def __init__(self, config, with_windows=False):
        self.config = config
        self.source_finished = False
        self.left = self.right = 0
        if with_windows:
            self.iter_windows = self._iter_windows

    def _windows(self, instructions):
        rng = np.random.default_rng(0)
        for k, t in enumerate(np.sort(instructions['time'])):
            left = int(t // 10) - 50
            left -= left % 2
            pulses = []
            for ch in sorted(rng.choice(494, size=rng.integers(1, 6), replace=False)):
                a = left + 2 * int(rng.integers(0, 20))
                n = 2 * int(rng.integers(30, 200)) + 1
                pulses.append((int(ch), a, a + n - 1, rng.integers(15000, 16000, n)))
            right = max(p[2] for p in pulses) + 50
            yield k, left, right, pulses

    def __call__(self, instructions, truth_buffer=None, **kw):
        ws = list(self._windows(instructions))
        for k, left, right, pulses in ws:
            self.left, self.right = left, right
            if k == len(ws) - 1:
                self.source_finished = True
            for p in pulses:
                yield p
        self.source_finished = True

    def _iter_windows(self, instructions, truth_buffer=None, **kw):
        dtype = np.dtype(raw_record_dtype())
        ws = list(self._windows(instructions))
        for k, left, right, pulses in ws:
            recs = []
            for ch, a, b, data in pulses:
                n = b - a + 1
                need = -(-n // 110)
                r = np.zeros(need, dtype=dtype)
                r['channel'], r['dt'], r['pulse_length'] = ch, 10, n
                r['record_i'] = np.arange(need)
                r['time'] = 10 * (a + 110 * np.arange(need))
                r['length'] = [min(n, 110 * (i + 1)) - 110 * i for i in range(need)]
                r['data'] = np.pad(data, (0, need * 110 - n)).reshape(-1, 110)
                recs.append(r)
            self.left, self.right = left, right
            if k == len(ws) - 1:
                self.source_finished = True
            yield dict(left=left, right=right, records=np.concatenate(recs))
        self.source_finished = True


def _instructions(times):
    ins = np.zeros(len(times), dtype=instruction_dtype)
    ins['time'] = times
    ins['type'] = 1
    ins['amp'] = 100
    return ins


def _run(times, chunk_size, with_windows):
    cfg = xenonnt_test_config(chunk_size=chunk_size)
    ChunkRawRecords.record_buffer_length = 20000
    sim = ChunkRawRecords(cfg, rawdata_generator=FakeRawData, with_windows=with_windows)
    chunks = []
    for res in sim(_instructions(times)):
        chunks.append((sim.chunk_time_pre, sim.chunk_time, res['raw_records'].copy()))
    assert sim.source_finished()
    return chunks


TIMES = [np.array([1_000_000 * (i + 1) for i in range(40)], dtype=np.int64),
         np.array([5_000_000, 5_300_000, 2_000_000_000, 2_000_400_000, 9_000_000_000], dtype=np.int64)]


@pytest.mark.parametrize('times', TIMES)
@pytest.mark.parametrize('chunk_size', [0.005, 1])
def test_window_path_equals_pulse_path(times, chunk_size):
    a = _run(times, chunk_size, False)
    b = _run(times, chunk_size, True)
    assert len(a) == len(b)
    for (p0, t0, r0), (p1, t1, r1) in zip(a, b):
        assert (p0, t0) == (p1, t1)
        assert r0.tobytes() == r1.tobytes()


@pytest.mark.parametrize('times', TIMES)
def test_chunks_are_sorted_complete_and_disjoint(times):
    chunks = _run(times, 0.005, True)
    allrec = np.concatenate([c[2] for c in chunks])
    one = _run(times, 1000, True)
    ref = np.concatenate([c[2] for c in one])
    assert len(allrec) == len(ref)
    assert np.array_equal(np.sort(allrec, order=['time', 'channel']), np.sort(ref, order=['time', 'channel']))
    last_end = None
    for pre, end, rec in chunks:
        if len(rec):
            assert np.all(np.diff(rec['time']) >= 0)
            assert rec['time'].min() >= pre and rec['time'].max() <= end
        if last_end is not None:
            assert pre == last_end
        last_end = end


def test_empty_instructions():
    cfg = xenonnt_test_config()
    sim = ChunkRawRecords(cfg, rawdata_generator=FakeRawData)
    assert list(sim(_instructions([]))) == []
    assert sim.source_finished()


def test_optical_adjustment_matches_reference():
    """host preparation of optical input (utils.py:122-165) against the reference's own output on the same arrays:
    entry times moved to their first photon, long entries split into an early and a late instruction"""
    from tests.helpers import golden
    from wfsim_amd.optical import optical_adjustment, PULSE_MAX_DURATION
    d = golden('optical_adjustment.npz')
    t, c = d['timings_in'].copy(), d['channels_in'].copy()
    out = optical_adjustment(d['ins_in'], t, c)
    assert len(out) == len(d['ins_out']) > len(d['ins_in'])
    for f in out.dtype.names:
        assert np.array_equal(out[f], d['ins_out'][f]), f
    assert np.array_equal(t, d['timings_out']) and np.array_equal(c, d['channels_out'])
    # the entries kept by the original rows now fit the pulse length; the appended rows hold the late photons
    n0 = len(d['ins_in'])
    for r in out[:n0]:
        if r['_last'] > r['_first']:
            assert t[r['_first']:r['_last']].max() <= PULSE_MAX_DURATION and t[r['_first']:r['_last']].min() >= 0
    for r in out[n0:]:
        assert t[r['_first']:r['_last']].min() > PULSE_MAX_DURATION


def test_synchronise_timing_of_tpc_and_nveto():
    """RawRecordsFromMcChain.set_timing (strax_interface.py:824-863): one time per g4 event, shared by both detectors;
    TPC instructions keep their physical delays; what lands behind the last event slot is dropped"""
    from wfsim_amd import synchronise_timing
    from wfsim_amd.dtypes import instruction_dtype, optical_extra_dtype
    tpc = np.zeros(9, dtype=instruction_dtype)
    tpc['g4id'] = [3, 3, 4, 4, 6, 6, 9, 9, 9]
    tpc['time'] = [0, 50, 0, 120, 0, 10, 0, 5, 10 ** 12]            # the last one is far beyond the last slot
    nv = np.zeros(4, dtype=instruction_dtype + optical_extra_dtype)
    nv['g4id'] = [3, 5, 9, 6]
    cfg = dict(event_rate=1000.0, entry_start=0, entry_stop=None, seed=12)
    a, b, t = synchronise_timing(cfg, tpc, nv)
    assert (cfg['entry_start'], cfg['entry_stop']) == (3, 10) and len(t) == 7 and np.all(np.diff(t) >= 0)
    assert t.min() >= int(3.5e6) and t.max() <= int(10.5e6)
    assert len(a) == 8 and len(b) == 4
    slot = {g: t[g - 3] for g in range(3, 10)}
    assert np.array_equal(a['time'], [slot[g] + d for g, d in zip(tpc['g4id'][:8], tpc['time'][:8])])
    assert np.array_equal(b['time'], [slot[g] for g in nv['g4id']])
    a2, b2, t2 = synchronise_timing(dict(cfg, entry_stop=None), tpc, nv)
    assert np.array_equal(t, t2)                                     # seeded by the config


def test_read_optical_events_thins_by_quantum_efficiency():
    """read_optical behind the ROOT reader (strax_interface.py:235-333): entry selection, nVeto QE thinning per photon,
    0-based channels, _first / _last ranges, times moved to the first photon"""
    from wfsim_amd.optical import read_optical_events
    rng = np.random.default_rng(5)
    n_ev = 300
    nph = rng.poisson(40, n_ev)
    events = dict(eventid=np.arange(100, 100 + n_ev),
                  pmthitID=[rng.integers(1995, 2125, k) for k in nph],             # a few outside the nVeto range
                  pmthitTime=[np.sort(rng.exponential(80e-9, k)) + 1e-6 for k in nph],
                  pmthitEnergy=[np.full(k, 1239.841984 / 400.0) for k in nph],      # 400 nm
                  xp_pri=rng.uniform(-500, 500, n_ev), yp_pri=rng.uniform(-500, 500, n_ev), zp_pri=rng.uniform(-900, 0, n_ev))
    qe = dict(nv_pmt_qe_wavelength=[300.0, 400.0, 500.0], nv_pmt_qe={str(c): [10.0, 30.0, 10.0] for c in range(2000, 2120)})
    cfg = dict(detector='XENONnT_neutron_veto', channel_map=dict(nveto=(2000, 2119)), entry_start=150, entry_stop=None, seed=3, nv_pmt_ce_factor=0.5)
    ins, channels, timings = read_optical_events(cfg, events, qe_data=qe)
    assert cfg['entry_stop'] == 400 and len(ins) >= 250 and np.all(ins['g4id'] >= 150)
    assert channels.min() >= 0 and channels.max() <= 119 and len(channels) == len(timings)
    kept = len(channels) / sum(len(h[(h >= 2000) & (h <= 2119)]) for h in events['pmthitID'][50:])
    assert abs(kept - 0.15) < 0.01                                     # QE 30 % x CE 0.5
    first = ins[ins['_last'] > ins['_first']]
    assert np.all(timings[first['_first'][:250]] >= 0) and np.all(first['time'][:250] >= 1000)      # time = first photon (>= 1 us)
    assert np.all(np.diff(ins['_first'][:250]) >= 0) and ins['_last'][249] <= len(channels)
    # TPC flavour: no thinning
    tcfg = dict(detector='XENONnT', entry_start=0, entry_stop=None)
    ev2 = dict(events, pmthitID=[rng.integers(0, 494, k) for k in nph])
    ins2, ch2, t2 = read_optical_events(tcfg, ev2)
    assert len(ch2) == nph.sum() and np.array_equal((ins2['_last'] - ins2['_first'])[:n_ev], nph)


def test_instruction_from_csv_round_trip(tmp_path):
    """strax_interface.py:336-350: columns named after instruction_dtype fields; missing columns stay zero"""
    import pandas as pd
    from wfsim_amd import instruction_from_csv
    from wfsim_amd.dtypes import instruction_dtype
    ins = np.zeros(6, dtype=instruction_dtype)
    ins['type'], ins['time'], ins['amp'] = [1, 2] * 3, 1_000_000 * np.repeat([1, 2, 3], 2), [500, 40] * 3
    ins['x'], ins['z'], ins['recoil'], ins['event_number'] = 1.5, -20.25, 7, np.repeat([0, 1, 2], 2)
    path = str(tmp_path / 'ins.csv')
    pd.DataFrame({k: ins[k] for k in ('event_number', 'type', 'time', 'x', 'z', 'amp', 'recoil')}).to_csv(path, index=False)
    got = instruction_from_csv(path)
    assert got.dtype == np.dtype(instruction_dtype) and np.array_equal(got, ins)
