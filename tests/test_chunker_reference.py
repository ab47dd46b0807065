"""ChunkRawRecords pinned on the REFERENCE's chunker (strax_interface.py:354-504).

tests/golden/chunker_*.npz (make_golden.py: fixture_chunker) hold runs of the reference's ChunkRawRecords around the
reference's RawData: the stream the generator handed to the chunker -- every (channel, left, right, data) tuple with
rawdata.left / .right at that moment, every truth row with the place in the stream where it was written -- and every
chunk the reference yielded (bounds, records of each data type field by field, truth rows).  Here the recorded stream
is replayed through wfsim_amd.ChunkRawRecords (the per-pulse path any foreign generator takes) and the chunks must be
identical: chunk_time_pre / chunk_time, extension of a chunk that would end inside an event, flushes of a full record
buffer and skipped pulses, truth rows per chunk.  CPU only; the GPU batch path is compared with this per-pulse path in
tests/test_gpu_chunker.py."""
import numpy as np
import pytest

import wfsim_amd
from tests.helpers import golden
from wfsim_amd.config import xenonnt_test_config

CASES = {'multi': dict(chunk_size=0.004), 'midevent': dict(chunk_size=0.001, right_raw_extension=500_000),
         'tinybuffer': dict(chunk_size=0.005), 'ele_ap': dict(chunk_size=0.002)}
RECORD_FIELDS = ('time', 'length', 'dt', 'channel', 'pulse_length', 'record_i', 'baseline')


class ReplayRawData:
    """a generator with the reference's RawData protocol (rawdata.py:38-157) that replays a recorded stream"""

    def __init__(self, config, fixture=None, **kwargs):
        self.config, self.d = config, fixture          # (a dict of arrays: an NpzFile decompresses on every access)
        self.left = self.right = 0
        self.source_finished = False

    def _write_truth(self, truth_buffer, k):
        ix = np.argmin(truth_buffer['fill'])          # rawdata.py:320: the first empty row
        for name in truth_buffer.dtype.names:
            if name != 'fill':
                truth_buffer[ix][name] = self.d['t_' + name][k]
        truth_buffer[ix]['fill'] = True

    def __call__(self, instructions, truth_buffer=None, **kwargs):
        d = self.d
        off, k = d['p_data_off'], 0
        for i in range(len(d['p_ch'])):
            while k < len(d['t_at']) and d['t_at'][k] == i:
                self._write_truth(truth_buffer, k); k += 1
            self.left, self.right = int(d['p_gen_left'][i]), int(d['p_gen_right'][i])
            yield int(d['p_ch'][i]), int(d['p_left'][i]), int(d['p_right'][i]), d['p_data'][off[i]:off[i + 1]].astype(np.int64)
        while k < len(d['t_at']):
            self._write_truth(truth_buffer, k); k += 1
        self.source_finished = True


def _same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.dtype.kind == 'f' or b.dtype.kind == 'f':
        return a.shape == b.shape and np.array_equal(a, b, equal_nan=True)
    return np.array_equal(a, b)


@pytest.mark.parametrize('case', list(CASES))
def test_chunks_identical_to_the_reference_chunker(case):
    z = golden(f'chunker_{case}.npz')
    d = {k: z[k] for k in z.files}
    cfg = xenonnt_test_config(**CASES[case])
    sim = wfsim_amd.ChunkRawRecords(cfg, rawdata_generator=ReplayRawData, fixture=d)
    if int(d['record_buffer_length']) != len(sim.record_buffer):
        sim.record_buffer = np.zeros(int(d['record_buffer_length']), dtype=sim.record_buffer.dtype)
    chunks = []
    for res in sim(d['instructions']):
        chunks.append((int(sim.chunk_time_pre), int(sim.chunk_time), {k: np.array(v) for k, v in res.items()}))
    assert sim.source_finished()
    # chunk bounds
    assert [c[0] for c in chunks] == d['c_pre'].tolist()
    assert [c[1] for c in chunks] == d['c_end'].tolist()
    # records of every data type, chunk by chunk and field by field
    for kind in ('raw_records', 'raw_records_he', 'raw_records_aqmon'):
        off = d[f'c_{kind}_off']
        assert [len(c[2][kind]) for c in chunks] == np.diff(off).tolist(), kind
        got = np.concatenate([c[2][kind] for c in chunks])
        for f in RECORD_FIELDS:
            assert np.array_equal(got[f], d[f'c_{kind}_{f}']), (kind, f)
        assert np.array_equal(got['data'].astype(np.int64).sum(axis=1) if len(got) else np.zeros(0, np.int64), d[f'c_{kind}_data_sum'])
        assert np.array_equal(got['data'][:, :8].astype(np.int32).reshape(-1, 8), d[f'c_{kind}_data_head'].reshape(-1, 8))
    # truth rows: which chunk, which order, every field; and the field set itself (instruction_dtype + truth dtype)
    toff = d['c_truth_off']
    assert [len(c[2]['truth']) for c in chunks] == np.diff(toff).tolist()
    truth = np.concatenate([c[2]['truth'] for c in chunks])
    ref_names = {k[len('c_truth_'):] for k in d if k.startswith('c_truth_') and k != 'c_truth_off'}
    assert set(truth.dtype.names) == ref_names
    for name in truth.dtype.names:
        assert _same(truth[name], d['c_truth_' + name]), name


def test_the_fixtures_cover_the_corners():
    """what the four runs exercise (so that a regenerated fixture that lost a corner is noticed)"""
    d = golden('chunker_midevent.npz')
    assert np.any(np.diff(np.stack([d['c_pre'], d['c_end']]), axis=0) > 1_000_000)         # a 1 ms chunk extended over an event
    d = golden('chunker_tinybuffer.npz')
    n_in = len(np.unique(np.stack([d['p_left'], d['p_ch']]), axis=1).T)
    assert len(d['c_pre']) > 100 and d['c_raw_records_off'][-1] < int(np.ceil((d['p_right'] - d['p_left'] + 1) / 110).sum())   # flushes, skipped pulses
    assert n_in > 0
    d = golden('chunker_ele_ap.npz')
    assert np.any(d['t_type'] == 4)                                                           # secondaries got truth rows
    assert len(golden('chunker_multi.npz')['c_pre']) >= 4


def test_truth_chunks_never_carry_the_optical_columns():
    """the optical plugins give the truth buffer _first / _last columns (strax_interface.py:730); the chunks are still
    instruction_dtype + truth fields (:478), which is what the plugins declare (:699, :911)"""
    from wfsim_amd.dtypes import instruction_dtype, optical_extra_dtype
    z = golden('chunker_multi.npz')
    d = {k: z[k] for k in z.files}
    d['t__first'] = np.arange(len(d['t_at']), dtype=np.int32); d['t__last'] = d['t__first'] + 7
    cfg = xenonnt_test_config(chunk_size=0.004)
    sim = wfsim_amd.ChunkRawRecords(cfg, rawdata_generator=ReplayRawData, fixture=d)
    sim.truth_buffer = np.zeros(10000, dtype=instruction_dtype + optical_extra_dtype + sim.truth_dtype + [('fill', bool)])
    truth = np.concatenate([c['truth'] for c in sim(d['instructions'])])
    assert truth.dtype == np.dtype(instruction_dtype + sim.truth_dtype) and len(truth) == len(d['t_at'])
    assert np.array_equal(np.sort(truth['n_photon']), np.sort(d['c_truth_n_photon']))
