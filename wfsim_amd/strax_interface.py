"""The strax-facing surface of the hot path: ``ChunkRawRecords`` and the ``RawRecordsFromFax*`` plugins.

Drop-in for /root/reference/wfsim/strax_interface.py:353-504 (ChunkRawRecords) and :536-733, :1008-1011 (plugins):
same constructor, same generator protocol, same ``chunk_time_pre`` / ``chunk_time`` / ``source_finished()``
attributes, same record layout and chunk boundaries.  The instruction readers that need nestpy / epix / uproot
(rand_instructions, epix, the ROOT reader of read_optical) are out of scope; CSV input is kept and
read_optical is restated behind its reader (optical.read_optical_events).
"""
import logging
import os

import numpy as np

from .dtypes import (instruction_dtype, optical_extra_dtype, extra_truth_dtype_per_pmt,
                     raw_record_dtype, DEFAULT_RECORD_LENGTH)
from .rawdata import RawData, RawDataOptical

try:                                    # the real strax when it is installed, the in-repo stand-ins otherwise
    import strax as _strax
    _Plugin, sort_by_time = _strax.Plugin, _strax.sort_by_time
    HAVE_STRAX = True
except ImportError:                     # pragma: no cover - depends on the environment
    from . import ministrax as _strax
    _Plugin, sort_by_time = _strax.Plugin, _strax.sort_by_time
    HAVE_STRAX = False

log = logging.getLogger('wfsim_amd.interface')


def _copy_records(src, threads=8, min_records=200_000):
    """a copy of a large record array made by a few threads (numpy releases the GIL while copying): the chunks the plugin
    hands to strax are copies of the record buffer, 0.6 GB per headline batch"""
    n = len(src)
    if n < min_records:
        return src.copy()
    from concurrent.futures import ThreadPoolExecutor
    threads = max(1, min(threads, os.cpu_count() or 1))
    out = np.empty_like(src)            # (fresh pages: most of the time goes to the first touch, which the threads share)
    cuts = np.linspace(0, n, threads + 1).astype(np.int64)

    def part(k):
        out[cuts[k]:cuts[k + 1]] = src[cuts[k]:cuts[k + 1]]
    with ThreadPoolExecutor(threads) as pool:
        list(pool.map(part, range(threads)))
    return out


def instruction_from_csv(filename):
    """Instructions from a csv whose columns are named after instruction_dtype fields (strax_interface.py:336-350); fields
    without a column stay zero, a column that is no field is an error (numpy's, as in the reference)."""
    import pandas
    table = pandas.read_csv(filename)
    out = np.zeros(table.shape[0], dtype=instruction_dtype)
    for name in table.columns:
        out[name] = table[name].to_numpy()
    return out


class ChunkRawRecords(object):
    record_buffer_length = 5000000          # strax_interface.py:360
    zero_copy_min_records = 200_000         # chunks of at least this many records are handed out without a copy (_hand_out)

    def __init__(self, config, rawdata_generator=RawData, **kwargs):
        self.config, self.rawdata = config, rawdata_generator(config, **kwargs)
        # strax_interface.py:360-361; page-locked when the HIP generator fills it (the records of a batch then arrive at PCIe
        # speed while the next batch's kernels run) and recycled between instances: pinning 1.2 GB takes a few hundred ms
        rdt = raw_record_dtype(samples_per_record=DEFAULT_RECORD_LENGTH)
        if hasattr(self.rawdata, 'iter_batches') and config.get('pin_record_buffer', True):
            from .engine import acquire_record_buffer, is_pooled_record_buffer
            self.record_buffer = acquire_record_buffer(self.record_buffer_length, rdt)
            self._own_buffers = [self.record_buffer] if is_pooled_record_buffer(self.record_buffer) else []
        else:
            self.record_buffer = np.zeros(self.record_buffer_length, dtype=rdt)
        # strax_interface.py:363-366: truth rows wait in a 10^4-row buffer ('fill' marks the rows in use)
        self.truth_dtype = extra_truth_dtype_per_pmt(len(config.get('gains', [])) if config.get('per_pmt_truth') else False)
        self.truth_buffer = np.zeros(10_000, dtype=instruction_dtype + self.truth_dtype + [('fill', bool)])
        self.blevel = 0
        self._next_buffer = None
        self._batch_leases = []

    _own_buffers = ()           # pooled buffers this instance holds (the current one, a spare taken ahead)

    def __del__(self):
        try:
            from .engine import release_record_buffer
            for b in self._own_buffers:
                release_record_buffer(b)
        except Exception:
            pass

    # ------------------------------------------------------------------------------------------
    def __call__(self, instructions, time_zero=None, chunk_start=None, chunk_end=None, **kwargs):
        """Generator of chunks (strax_interface.py:368-440): dicts of record arrays + truth rows; ``chunk_time_pre`` / ``chunk_time``
        are the bounds of the chunk just yielded.  ``chunk_start`` / ``chunk_end`` (sharded delivery, distributed.shard_plan): the
        first chunk opens exactly at ``chunk_start`` and the last one closes at ``chunk_end`` instead of before the first instruction /
        behind the last window, so that the chunk streams of consecutive time ranges join without gap or overlap."""
        if not len(instructions):               # nothing to simulate: no chunk at all (:370-373)
            setattr(self.rawdata, 'source_finished', True)
            return
        c = self.config
        self._dt, self._rext, self._cksz = c['sample_duration'], int(c['right_raw_extension']), int(c['chunk_size'] * 1e9)
        dt = self._dt
        self.blevel = 0
        # the first chunk opens right_raw_extension before time_zero (a falsy time_zero: before the first instruction), :378-380
        origin = time_zero if time_zero else np.min(instructions['time'])
        self.chunk_time_pre = origin - self._rext if chunk_start is None else int(chunk_start)
        self.chunk_time = self.chunk_time_pre + self._cksz
        self.last_digitized_right = self.current_digitized_right = 0
        self._sorted_stream = False
        if hasattr(self.rawdata, 'iter_batches') and not os.environ.get('WFSIM_AMD_SCALAR_CHUNKER'):
            yield from self._run_batches(instructions, **kwargs)
        elif hasattr(self.rawdata, 'iter_windows'):
            yield from self._run_windows(instructions, **kwargs)
        else:
            yield from self._run_pulses(instructions, **kwargs)
        # the last chunk ends behind the last digitised window, and is at least one sample long (:438-440)
        self.last_digitized_right = int(self.current_digitized_right)
        end_of_data = (self.last_digitized_right + 1) * dt
        if chunk_end is not None:
            if end_of_data > chunk_end:
                raise RuntimeError(f'a digitise window ends at {end_of_data}, behind the end of this time range ({chunk_end}): the '
                                   f'next range would depend on it (shard cut too tight)')
            end_of_data = int(chunk_end)
        self.chunk_time = end_of_data if end_of_data > self.chunk_time_pre + dt else self.chunk_time_pre + dt
        yield from self.final_results()

    def _maybe_close_chunk(self):
        """strax_interface.py:398-407: the window being emitted starts beyond the current chunk."""
        if not (self.rawdata.left * self._dt > self.chunk_time + self._rext):
            return
        # a chunk never ends inside the window digitised before this one: its end moves behind that window
        self.chunk_time = max(self.chunk_time, (self.last_digitized_right + 1) * self._dt)
        yield from self.final_results()
        self._open_next_chunk()
        if hasattr(self.rawdata, 'cut_origin'):
            self.rawdata.cut_origin = self.chunk_time + self._rext

    def _open_next_chunk(self):
        self.chunk_time_pre, self.chunk_time = self.chunk_time, self.chunk_time + self._cksz

    def _buffer_full_flush(self):
        """strax_interface.py:409-418"""
        log.warning('record buffer full before the chunk is: the chunk ends early (a TPC and an nVeto simulation no longer share '
                    'chunk boundaries); reduce chunk_size')
        self.chunk_time = (self.last_digitized_right + 1) * self._dt
        yield from self.final_results()
        self._open_next_chunk()

    def _track_window(self):
        now = self.rawdata.right                 # a new window: remember where the one before it ended (strax_interface.py:394-396)
        if now != self.current_digitized_right:
            self.last_digitized_right, self.current_digitized_right = self.current_digitized_right, now

    # ---- batch granularity: the GPU hands over whole batches of windows, records already ordered by (time, channel) ----
    def _run_batches(self, instructions, **kwargs):
        """Same decisions as the per-pulse loop of the reference (strax_interface.py:388-436), taken once per stretch of
        windows that cannot close a chunk: windows start in time order, so the first one beyond ``chunk_time + rext`` is
        found by bisection and everything in front of it is committed in one go.  The records of a batch are copied
        from the device straight into the record buffer (behind what is committed), ordered as strax.sort_by_time
        would order them -- final_results then only cuts the buffer at ``chunk_time``."""
        rd, dt = self.rawdata, self._dt
        L = len(self.record_buffer)
        rd.engine.set_record_order(True)
        rd.record_budget = max(L // 2, 1)       # a batch's records go from the device straight into the buffer
        rd.cut_period, rd.cut_origin = self._cksz, self.chunk_time + self._rext     # batches end where chunks do, if that is cheap
        self._sorted_stream = True

        def sink(n, first_left=None):
            # (self.record_buffer: final_results may have handed the previous buffer to the consumer)
            # A batch whose first window lies beyond the open chunk closes it before any of its records is committed, and every
            # record in the buffer then belongs to that chunk: such a batch goes to the front of a spare buffer, the full one is
            # handed to the consumer as it is (_hand_out) -- no record is copied on the host at all.
            from .engine import acquire_record_buffer, is_pooled_record_buffer
            if (first_left is not None and n <= L and self.blevel >= self.zero_copy_min_records and self._next_buffer is None
                    and first_left * dt > self.chunk_time + self._rext and self.config.get('zero_copy_chunks', True)
                    and is_pooled_record_buffer(self.record_buffer)):
                spare = acquire_record_buffer(L, self.record_buffer.dtype, spare_only=True)
                if spare is not None:
                    self._next_buffer = spare
                    self._own_buffers.append(spare)
                    return spare[:n]
            return self.record_buffer[self.blevel:self.blevel + n] if self.blevel + n <= L else None

        for batch in rd.iter_batches(instructions, want_truth=True, record_sink=sink, **kwargs):
            self._batch_leases = []             # buffers handed out while this batch is worked on stay taken until it is done
            rec, first, left, right = batch['records'], batch['first'], batch['left'], batch['right']
            table, rows, before = batch['truth_table'], batch['truth_rows'], batch['truth_before']
            left_ns = left * dt
            n_win, w, k = len(left), 0, 0
            while w < n_win:
                stop = w + int(np.searchsorted(left_ns[w:], self.chunk_time + self._rext, side='right'))
                n_new = int(first[min(stop, n_win)] - first[w])
                if stop > w and self.blevel + n_new <= L:
                    # none of these windows starts beyond the open chunk: commit them together
                    k1 = int(np.searchsorted(before, stop, side='left'))
                    if k1 > k:
                        rd._write_truth(table, rows[k:k1], self.truth_buffer)
                        k = k1
                    for r in right[w:stop].tolist():
                        if r != self.current_digitized_right:
                            self.last_digitized_right, self.current_digitized_right = self.current_digitized_right, r
                    rd.left, rd.right = int(left[stop - 1]), int(right[stop - 1])
                    buf = self.record_buffer
                    src = rec[first[w]:first[stop]]
                    dst = buf[self.blevel:self.blevel + n_new]
                    # cutting the buffer at chunk_time by bisection needs it time sorted: windows of a run do not overlap in time
                    # and a batch is sorted on the device -- should a stretch ever start before the last committed record
                    # (tiny right_raw_extension, forced cluster breaks), the chunks fall back to the reference's mask + sort
                    if n_new and self.blevel and int(src['time'][0]) < int(buf['time'][self.blevel - 1]):
                        self._sorted_stream = False
                    if n_new and src.ctypes.data != dst.ctypes.data:
                        dst[:] = src
                    self.blevel += n_new
                    w = stop
                    continue
                # one window the reference's way: each of its pulses re-tests the chunk condition and can close one chunk
                k1 = int(np.searchsorted(before, w, side='right'))
                if k1 > k:
                    rd._write_truth(table, rows[k:k1], self.truth_buffer)
                    k = k1
                rd.left, rd.right = int(left[w]), int(right[w])
                self._track_window()
                wrec = rec[first[w]:first[w + 1]]
                n_pulses = int(np.count_nonzero(wrec['record_i'] == 0))
                closed = 0
                while closed < n_pulses and rd.left * dt > self.chunk_time + self._rext:
                    yield from self._maybe_close_chunk()
                    closed += 1
                if self.blevel + len(wrec) > L:
                    yield from self._buffer_full_flush()
                if self.blevel + len(wrec) > L:
                    log.warning('pulse longer than the record buffer has room for: skipped')
                    # the reference skips pulse by pulse in the order it yields them; here whole records in time order
                    wrec = wrec[:max(L - self.blevel, 0)]
                buf = self.record_buffer
                dst = buf[self.blevel:self.blevel + len(wrec)]
                if len(wrec) and self.blevel and int(wrec['time'][0]) < int(buf['time'][self.blevel - 1]):
                    self._sorted_stream = False
                if len(wrec) and wrec.ctypes.data != dst.ctypes.data:
                    dst[:] = wrec
                self.blevel += len(wrec)
                w += 1
            if k < len(rows):
                rd._write_truth(table, rows[k:], self.truth_buffer)
            if batch['finished']:
                rd.source_finished = True
            self._drop_next_buffer()            # (a spare buffer this batch did not move into after all: its records were copied over)
        self._batch_leases = []

    def _drop_next_buffer(self):
        if self._next_buffer is not None:
            from .engine import release_record_buffer
            release_record_buffer(self._next_buffer)
            self._own_buffers = [b for b in self._own_buffers if b is not self._next_buffer]
            self._next_buffer = None

    def _hand_out(self, n_out):
        """The first n_out records of the (time-sorted) record buffer as the chunk's array.  Large chunks are not copied: the
        buffer itself goes to the consumer (it returns to the pool when the consumer drops the chunk) and the chunker carries
        on in a spare page-locked buffer, into which only the records behind the cut are moved.  Without a spare buffer
        (a consumer that keeps every chunk) or for small chunks: a copy, as the reference makes.  Returns (array, moved)."""
        from .engine import acquire_record_buffer, is_pooled_record_buffer, lease_record_buffer
        old = self.record_buffer
        if (n_out < self.zero_copy_min_records or not self.config.get('zero_copy_chunks', True) or not is_pooled_record_buffer(old)):
            return _copy_records(old[:n_out]), False
        if self._next_buffer is not None and n_out == self.blevel:
            new, self._next_buffer = self._next_buffer, None       # the batch in hand already lies at its front (sink)
        elif self._next_buffer is not None:
            return _copy_records(old[:n_out]), False               # (records stay behind the cut: not the case the spare was taken for)
        else:
            new = acquire_record_buffer(len(old), old.dtype, spare_only=True)
            if new is None:
                return _copy_records(old[:n_out]), False
            self._own_buffers.append(new)
        self._own_buffers = [b for b in self._own_buffers if b is not old]      # the lease's finaliser returns it to the pool
        n_left = self.blevel - n_out
        new[:n_left] = old[n_out:self.blevel]
        out = lease_record_buffer(old, n_out)
        if getattr(self, '_batch_leases', None) is not None:
            self._batch_leases.append(out)      # records of the batch in hand may still be read from the old buffer
        self.record_buffer, self.blevel = new, n_left
        return out, True

    # ---- window granularity: records arrive packed from the GPU ------------------------------------
    def _run_windows(self, instructions, **kwargs):
        capacity = len(self.record_buffer)
        for w in self.rawdata.iter_windows(instructions, truth_buffer=self.truth_buffer, **kwargs):
            rec = w['records']
            n_pulses = int(np.count_nonzero(rec['record_i'] == 0))
            self._track_window()
            # every pulse of the window re-tests the chunk condition and can close one chunk (strax_interface.py:398)
            closed = 0
            while closed < n_pulses and self.rawdata.left * self._dt > self.chunk_time + self._rext:
                yield from self._maybe_close_chunk()
                closed += 1
            if self.blevel + len(rec) > capacity:
                yield from self._buffer_full_flush()
            if self.blevel + len(rec) > capacity:
                # the reference skips pulses one at a time here; keep whole pulses that fit
                keep = 0
                starts = np.where(rec['record_i'] == 0)[0]
                for s, e in zip(starts, np.append(starts[1:], len(rec))):
                    if self.blevel + e > capacity:
                        break
                    keep = e
                log.warning('pulse longer than the record buffer has room for: skipped')
                rec = rec[:keep]
            self.record_buffer[self.blevel:self.blevel + len(rec)] = rec
            self.blevel += len(rec)

    # ---- pulse granularity: any generator with the RawData protocol (strax_interface.py:388-436) ---
    def _pulse_records(self, channel, left, data):
        """one ZLE pulse as raw_records: fragments of DEFAULT_RECORD_LENGTH samples, the last one zero padded (:425-435)"""
        spr, n = DEFAULT_RECORD_LENGTH, len(data)
        n_frag = (n + spr - 1) // spr
        frag = np.arange(n_frag)
        rec = np.zeros(n_frag, dtype=self.record_buffer.dtype)
        rec['time'] = self._dt * (left + spr * frag)
        rec['length'] = np.minimum(n - spr * frag, spr)
        rec['dt'], rec['channel'], rec['pulse_length'], rec['record_i'] = self._dt, channel, n, frag
        samples = np.zeros(n_frag * spr, dtype=rec['data'].dtype)
        samples[:n] = data
        rec['data'] = samples.reshape(n_frag, spr)
        return rec

    def _run_pulses(self, instructions, **kwargs):
        """a generator that yields (channel, left, right, data) pulse by pulse -- the reference's own RawData or a replay of
        it (tests/test_chunker_reference.py); the product's generators deliver batches or windows (above)"""
        room = len(self.record_buffer)
        for channel, left, right, data in self.rawdata(instructions=instructions, truth_buffer=self.truth_buffer, **kwargs):
            rec = self._pulse_records(channel, left, data[:right - left + 1])
            self._track_window()
            yield from self._maybe_close_chunk()
            if self.blevel + len(rec) > room:
                yield from self._buffer_full_flush()
                if self.blevel + len(rec) > room:           # longer than the whole buffer: dropped (:419-422)
                    log.warning('pulse longer than the record buffer has room for: skipped')
                    continue
            self.record_buffer[self.blevel:self.blevel + len(rec)] = rec
            self.blevel += len(rec)

    # ------------------------------------------------------------------------------------------
    def final_results(self):
        """strax_interface.py:442-497"""
        records = self.record_buffer[:self.blevel]
        if getattr(self, '_sorted_stream', False):
            # the buffer is already in sort_by_time order (device sort inside a batch, windows in time order): cut it
            t = records['time']
            lo, hi = 0, self.blevel
            while lo < hi:                      # first record later than chunk_time (a bisection on the strided field)
                mid = (lo + hi) // 2
                if t[mid] <= self.chunk_time:
                    lo = mid + 1
                else:
                    hi = mid
            n_out = lo
            if os.environ.get('WFSIM_AMD_CHECK_SORTED'):
                assert np.array_equal(records[:n_out], sort_by_time(records[records['time'] <= self.chunk_time]))
            in_chunk = None
            records = records[:n_out]
        else:
            in_chunk = records['time'] <= self.chunk_time
            records = sort_by_time(records[in_chunk])

        # truth rows whose first photon (or, without photons, whose instruction time) lies in the chunk leave the buffer
        # (strax_interface.py:458-483): ordered by instruction time, stamped with the first photon's time, ordered again
        tb = self.truth_buffer
        tfp = tb['t_first_photon']
        no_photon = np.isnan(tfp)
        due = tb['fill'] & np.where(no_photon, tb['time'] <= self.chunk_time, tfp <= self.chunk_time)
        leaving = tb[due]
        tb['fill'][due] = False
        leaving.sort(order='time')
        _truth = np.zeros(len(leaving), dtype=self._truth_out_dtype())
        for column in _truth.dtype.names:
            _truth[column] = leaving[column]
        seen = ~np.isnan(_truth['t_first_photon'])
        _truth['time'][seen] = _truth['t_first_photon'][seen].astype(int)
        _truth.sort(order=('time',))

        det = self.config['detector']
        moved = False
        if det == 'XENON1T' or det == 'XENONnT_neutron_veto':
            if in_chunk is None:
                records, moved = self._hand_out(n_out)
            yield dict(raw_records=records, truth=_truth)
        elif det == 'XENONnT':
            he = self.config['channel_map']['he']
            engine = getattr(self.rawdata, 'engine', None)
            if in_chunk is None and engine is not None and not engine.emits_he_records:
                # every record is a TPC record (no HE rows are digitised, row 800 is never emitted): the buffer's prefix as it is
                empty = records[:0].copy()
                records, moved = self._hand_out(n_out)
                yield dict(raw_records=records, raw_records_he=empty, raw_records_aqmon=empty.copy(), truth=_truth)
            else:
                ch = records['channel']
                yield dict(raw_records=records[ch < he[0]], raw_records_he=records[(ch >= he[0]) & (ch <= he[-1])],
                           raw_records_aqmon=records[ch == 800], truth=_truth)
        if moved:
            return                  # _hand_out switched buffers and moved the records behind the cut
        if in_chunk is None:
            n_left = self.blevel - n_out
            self.record_buffer[:n_left] = self.record_buffer[n_out:self.blevel]
        else:
            n_left = int(np.sum(~in_chunk))
            self.record_buffer[:n_left] = self.record_buffer[:self.blevel][~in_chunk]
        self.blevel = n_left

    def _truth_out_dtype(self):
        """strax_interface.py:478: always instruction_dtype + the truth fields -- also when a plugin gave the truth buffer more
        columns (the optical plugins add _first / _last, :730): those never leave the chunker"""
        return instruction_dtype + self.truth_dtype

    def source_finished(self):
        return self.rawdata.source_finished


# ---------------------------------------------------------------------------------------------------------------
class SimulatorPlugin(_Plugin):
    """strax_interface.py:536-663: stateful source plugin around a ChunkRawRecords iterator."""
    compressor = 'zstd'
    depends_on = tuple()
    rechunk_on_save = False
    parallel = False
    last_chunk_time = -999999999999999
    input_timeout = 3600

    #: config defaults of the strax Options of the reference (strax_interface.py:506-535)
    option_defaults = dict(detector='XENONnT', event_rate=1000, chunk_size=100, n_chunk=10, per_pmt_truth=False,
                           fax_file=None, fax_config_override=None, right_raw_extension=100000, seed=False)

    def __init__(self, config=None, run_id=None, device=0):
        if HAVE_STRAX:          # pragma: no cover
            super().__init__()
            self.config = dict(config or {})
        else:
            super().__init__(config, run_id)
        for k, v in self.option_defaults.items():
            self.config.setdefault(k, v)
        self.device = device

    def setup(self):
        """strax calls this once per run: config, instructions, sanity checks, then the chunk iterator (strax_interface.py:553-563)"""
        for stage in (self.set_config, self.get_instructions, self.check_instructions, self._setup):
            stage()

    def set_config(self):
        """strax_interface.py:566-608 without the CMT / straxen look-ups: gains must be in the config (or to_pe)."""
        c = self.config
        overrides = c.get('fax_config_override')
        if overrides:
            c.update(overrides)
        if 'field_distortion_on' in c and 'field_distortion_model' not in c:
            c['field_distortion_model'] = 'inverse_fdc' if c['field_distortion_on'] else 'none'
        if 'gains' not in c:
            to_pe = np.asarray(c['to_pe'], dtype=np.float64)
            adc_2_current = c['digitizer_voltage_range'] / 2 ** (c['digitizer_bits']) / c['pmt_circuit_load_resistor']
            c['gains'] = np.divide(adc_2_current, to_pe, out=np.zeros_like(to_pe), where=to_pe != 0)
        c['channel_map'] = dict(c['channel_map'])
        c['channel_map']['sum_signal'] = 800
        c['channels_bottom'] = np.arange(c['n_top_pmts'], c['n_tpc_pmts'])

    def _setup(self):
        pass

    def get_instructions(self):
        pass

    def check_instructions(self):
        pass

    def _sort_check(self, results):
        """What strax relies on (strax_interface.py:622-640): every record array of a chunk is time sorted and starts at least
        1 us behind the last record time noted so far.  As in the reference a one-record array is only checked for the spacing,
        and the time noted is that of the last array with two or more records."""
        noted = self.last_chunk_time
        for data in (results if isinstance(results, list) else [results]):
            if not len(data):
                continue
            t = data['time']
            if t[0] < self.last_chunk_time + 1000:
                raise RuntimeError(f'Simulator returned chunks with insufficient spacing: the last chunk reached {self.last_chunk_time}, '
                                   f'this one starts at {t[0]}')
            if len(t) >= 2:
                if np.any(t[1:] < t[:-1]):
                    raise RuntimeError('Simulator returned non-sorted records!')
                noted = max(t.max(), self.last_chunk_time)
        self.last_chunk_time = noted

    def is_ready(self, chunk_i):
        """strax polls a source plugin before every chunk; the answer alternates, starting with True (strax_interface.py:642-649)"""
        self.ready = not getattr(self, 'ready', False)
        return self.ready

    def source_finished(self):
        sim = self.sim
        return sim.source_finished()

    @property
    def _n_channels(self):
        return len(self.config.get('gains', []))

    @property
    def _truth_dtype(self):
        return extra_truth_dtype_per_pmt(self._n_channels if self.config.get('per_pmt_truth') else False)


class RawRecordsFromFaxNT(SimulatorPlugin):
    """strax_interface.py:665-714"""
    provides = ('raw_records', 'raw_records_he', 'raw_records_aqmon', 'truth')

    def _setup(self):
        self.sim = ChunkRawRecords(self.config, device=self.device)
        shard = self._shard()
        if shard is None:
            self.sim_iter = iter(self.sim(self.instructions))
            return
        # Sharded delivery (one process per GPU): this process simulates ITS contiguous time range of whole clusters and
        # emits it as its own strax chunks -- no record leaves the rank.  The ranges are disjoint and meet exactly (shard_plan), the
        # RNG streams are keyed by run-wide instruction ids, so the chunk streams of ranks 0 .. world - 1, one after the other, hold
        # byte for byte the records and truth rows of a single-process run.  The reference has one feeding process
        # (parallel = False, strax_interface.py:546); `shard` = None keeps that contract.
        from .distributed import shard_plan
        rank, world = shard
        plan = shard_plan(self.config, self.instructions, world)
        b = plan['bounds']
        mine = plan['sorted_instructions'][b[rank]:b[rank + 1]]
        self.sim.rawdata.global_ids = plan['order'][b[rank]:b[rank + 1]]
        self.shard_range = (plan['starts'][rank], plan['ends'][rank])
        self.sim_iter = iter(self.sim(mine, chunk_start=plan['starts'][rank], chunk_end=plan['ends'][rank]))

    def _shard(self):
        """(rank, world_size) of the sharded delivery, or None: config['shard'] = (rank, world) | 'auto' (the torch.distributed
        process group, when one is initialised) | None (default)"""
        shard = self.config.get('shard')
        if shard is None:
            return None
        if shard == 'auto':
            import torch.distributed as dist
            if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
                return None
            return dist.get_rank(), dist.get_world_size()
        rank, world = (int(x) for x in shard)
        if not 0 <= rank < world:
            raise ValueError(f'shard = {shard!r}: rank must lie in [0, world)')
        return (rank, world) if world > 1 else None

    def get_instructions(self):
        if self.config.get('instructions') is not None:           # in-memory instructions (tests, benchmarks)
            self.instructions = np.asarray(self.config['instructions'])
        elif self.config['fax_file']:
            path = self.config['fax_file']
            assert path.endswith('csv'), 'Only csv input is supported'
            self.instructions = instruction_from_csv(path)
        else:
            raise NotImplementedError('rand_instructions needs nestpy (SURVEY.md 2.1 row 7): pass instructions or a csv')

    def check_instructions(self):
        """strax_interface.py:687-700: S2s below the cathode are dropped (S1s there pass); everything else must lie inside the TPC"""
        c, ins = self.config, self.instructions
        ins = ins[~((ins['type'] == 2) & (ins['z'] < -c['tpc_length']))]
        radius = np.hypot(ins['x'], ins['y'])
        assert np.all((radius < c['tpc_radius']) | np.isclose(radius, c['tpc_radius'])), 'Interaction is outside the TPC (radius)'
        assert np.all(ins['z'] < 0.25), 'Interaction is outside the TPC (in Z)'
        assert np.all(ins['amp'] > 0), 'Interaction has zero size'
        self.instructions = ins

    def infer_dtype(self):
        records = raw_record_dtype(samples_per_record=DEFAULT_RECORD_LENGTH)
        return {t: (instruction_dtype + self._truth_dtype) if t == 'truth' else records for t in self.provides}

    def compute(self):
        result = next(self.sim_iter, None)
        if result is None:
            raise RuntimeError('Bug in chunk count computation')         # strax asked for a chunk behind the last one
        self._sort_check(result[self.provides[0]])
        start, end = self.sim.chunk_time_pre, self.sim.chunk_time
        return {t: self.chunk(start=start, end=end, data=result[t], data_type=t) for t in self.provides}


class RawRecordsFromFax1T(RawRecordsFromFaxNT):
    provides = ('raw_records', 'truth')


def _optical_truth_buffer(truth_dtype):
    """the truth buffer of an optical chunker carries the instructions' _first / _last columns (strax_interface.py:730, 899)"""
    return np.zeros(10_000, dtype=instruction_dtype + optical_extra_dtype + truth_dtype + [('fill', bool)])


class RawRecordsFromFaxOpticalNT(RawRecordsFromFaxNT):
    """strax_interface.py:722-737: photon channels / timings supplied with the instructions (optical Geant4 input).
    Pass ``instructions``, ``channels``, ``timings`` (or a Geant4 file as ``fax_file`` where uproot is installed:
    optical.read_optical)."""

    def _setup(self):
        self.sim = ChunkRawRecords(self.config, rawdata_generator=RawDataOptical, channels=self.channels,
                                   timings=self.timings, device=self.device)
        self.sim.truth_buffer = _optical_truth_buffer(self._truth_dtype)
        self.sim_iter = iter(self.sim(self.instructions))

    def get_instructions(self):
        c = self.config
        if c.get('instructions') is None or c.get('channels') is None or c.get('timings') is None:
            from .optical import read_optical            # strax_interface.py:728: needs uproot for the Geant4 file
            self.instructions, self.channels, self.timings = read_optical(c)
            return
        self.instructions = np.asarray(c['instructions'])
        self.channels, self.timings = np.asarray(c['channels']), np.asarray(c['timings'])

    def check_instructions(self):
        assert '_first' in self.instructions.dtype.names, 'Require indexing info in optical instruction see optical extra dtype'
        assert np.all(self.instructions['type'] == 1), 'Only s1 type is supported for generating rawdata from optical input'


class RawRecordsFromFaxnVeto(RawRecordsFromFaxOpticalNT):
    """strax_interface.py:1008-1011 with only the nVeto target of RawRecordsFromMcChain (:753-1005): optical instructions
    for the neutron veto, detector ``XENONnT_neutron_veto``, channels shifted by channel_map['nveto'][0] on output."""
    provides = ('raw_records_nv', 'truth_nv')

    def set_config(self):
        self.config.setdefault('detector', 'XENONnT_neutron_veto')
        self.config['detector'] = 'XENONnT_neutron_veto'
        c = self.config
        overrides = c.get('fax_config_override_nveto') or c.get('fax_config_override')
        if overrides:
            c.update(overrides)
        if 'gains' not in c:
            to_pe = np.asarray(c['to_pe_nveto'], dtype=np.float64)
            # strax_interface.py:776: nveto digitiser 2 V range, 14 bit, 50 Ohm, 2 ns sampling constant
            c['gains'] = np.divide((2e-9 * 2 / 2 ** 14) / (1.6e-19 * 1 * 50), to_pe, out=np.zeros_like(to_pe), where=to_pe != 0)
        c['channel_map'] = dict(c['channel_map'])
        c['channels_bottom'] = np.array([], np.int64)

    def infer_dtype(self):
        # (the reference's class derives from RawRecordsFromMcChain: truth chunks are instruction_dtype + truth fields,
        # strax_interface.py:910-914, without the optical _first / _last columns of the truth buffer)
        return {'raw_records_nv': raw_record_dtype(samples_per_record=DEFAULT_RECORD_LENGTH),
                'truth_nv': instruction_dtype + self._truth_dtype}

    def compute(self):
        result = next(self.sim_iter, None)
        if result is None:
            raise RuntimeError('Bug in chunk count computation')
        rr = result['raw_records'].copy()
        rr['channel'] += self.config['channel_map']['nveto'][0]           # strax_interface.py:937
        self._sort_check(rr)
        return {'raw_records_nv': self.chunk(start=self.sim.chunk_time_pre, end=self.sim.chunk_time, data=rr, data_type='raw_records_nv'),
                'truth_nv': self.chunk(start=self.sim.chunk_time_pre, end=self.sim.chunk_time, data=result['truth'], data_type='truth_nv')}


def synchronise_timing(config, instructions_tpc=None, instructions_nveto=None, rng=None):
    """RawRecordsFromMcChain.set_timing, strax_interface.py:824-863: one random time per Geant4 event (``g4id``) at the
    configured ``event_rate``, added to the TPC instructions (which keep their physical delays) and to the nVeto
    instructions, so that both detectors see the same event at the same time; instructions that end up later than the
    last event slot are dropped.  Returns (tpc, nveto, timings).  ``rng``: numpy Generator (default: seeded by the config)."""
    c = config
    g4 = [np.asarray(i['g4id']) for i in (instructions_tpc, instructions_nveto) if i is not None and len(i)]
    g4id = np.unique(np.concatenate(g4)) if g4 else np.zeros(0, np.int64)
    if c.get('entry_stop') is None:
        c['entry_start'] = int(np.min(g4id))
        c['entry_stop'] = int(np.max(g4id)) + 1
    rate = c['event_rate'] / 1e9                      # Hz -> 1 / ns
    rng = rng or np.random.default_rng(int(c.get('seed', 0) or 0))
    n = c['entry_stop'] - c['entry_start']
    timings = np.sort(rng.uniform((c['entry_start'] + 0.5) / rate, (c['entry_stop'] + 0.5) / rate, n)).astype(np.int64)
    max_time = int((c['entry_stop'] + 0.5) / rate)
    out = []
    for ins in (instructions_tpc, instructions_nveto):
        if ins is None:
            out.append(None)
            continue
        ins = ins.copy()
        ins['time'] += timings[np.searchsorted(np.arange(c['entry_start'], c['entry_stop']), ins['g4id'])]
        out.append(ins[~(ins['time'] > max_time)])
    return out[0], out[1], timings


class RawRecordsFromMcChain(SimulatorPlugin):
    """strax_interface.py:753-1005: TPC and neutron veto of the same Geant4 events in one plugin.  The readers in front
    of it (epix for the TPC, read_optical for the nVeto: uproot) are out of scope; the instructions are supplied:
    ``instructions_epix`` (instruction_dtype with g4id), ``instructions_nveto`` + ``nveto_channels`` + ``nveto_timings``
    (optical), ``fax_config_nveto`` a dict of nVeto config values, ``to_pe_nveto`` or ``gains`` inside it."""
    provides = ('raw_records', 'raw_records_he', 'raw_records_aqmon', 'raw_records_nv', 'truth', 'truth_nv')

    def set_config(self):
        self.config.setdefault('targets', ('tpc',))
        self.config.setdefault('entry_start', 0)
        self.config.setdefault('entry_stop', None)
        super().set_config()
        if 'nveto' in self.config['targets']:
            import copy
            skip = ('instructions_epix', 'instructions_nveto', 'nveto_channels', 'nveto_timings')
            cn = {k: (v if k in skip else copy.deepcopy(v)) for k, v in self.config.items()}
            cn.update(self.config.get('fax_config_nveto') or {})
            cn['detector'] = 'XENONnT_neutron_veto'
            cn['channel_map'] = dict(cn['channel_map'])
            if self.config.get('fax_config_override_nveto') is not None:
                cn.update(self.config['fax_config_override_nveto'])
            if 'to_pe_nveto' in cn:
                to_pe = np.asarray(cn['to_pe_nveto'], dtype=np.float64)
                cn['gains'] = np.divide((2e-9 * 2 / 2 ** 14) / (1.6e-19 * 1 * 50), to_pe, out=np.zeros_like(to_pe), where=to_pe != 0)
            cn['channels_bottom'] = np.array([], np.int64)
            self.config_nveto = cn

    def get_instructions(self):
        c = self.config
        self.instructions_epix = self.instructions_nveto = None
        if 'tpc' in c['targets']:
            if c.get('instructions_epix') is None:
                raise NotImplementedError('epix is not available (SURVEY.md 2.1 row 7): pass instructions_epix')
            self.instructions_epix = np.asarray(c['instructions_epix'])
        if 'nveto' in c['targets']:
            if c.get('instructions_nveto') is None:
                raise NotImplementedError('read_optical needs uproot (SURVEY.md 2.1 row 7): pass instructions_nveto, nveto_channels, nveto_timings')
            ins = np.asarray(c['instructions_nveto'])
            self.nveto_channels, self.nveto_timings = np.asarray(c['nveto_channels']), np.asarray(c['nveto_timings'])
            if self.instructions_epix is not None:
                ins = ins[(ins['_last'] - ins['_first']) >= 0]
            self.instructions_nveto = ins
        self.instructions_epix, self.instructions_nveto, self.event_times = synchronise_timing(c, self.instructions_epix, self.instructions_nveto)

    def check_instructions(self):
        c = self.config
        if 'tpc' in c['targets']:
            ins = self.instructions_epix
            ins = ins[~((ins['z'] < - c['tpc_length']) & (ins['type'] == 2))]       # S1s below the cathode pass, S2s do not
            self.instructions_epix = ins
            r = np.sqrt(ins['x'] ** 2 + ins['y'] ** 2)
            assert np.all((r < c['tpc_radius']) | np.isclose(r, c['tpc_radius'])), 'Interaction is outside the TPC (radius)'
            assert np.all(ins['z'] < 0.25), 'Interaction is outside the TPC (in Z)'
            assert np.all(ins['amp'] > 0), 'Interaction has zero size'
            assert all(ins['g4id'] >= c['entry_start']) and all(ins['g4id'] < c['entry_stop'])
        if 'nveto' in c['targets']:
            ins = self.instructions_nveto
            assert all(ins['g4id'] >= c['entry_start']) and all(ins['g4id'] < c['entry_stop'])
            assert '_first' in ins.dtype.names, 'Require indexing info in optical instruction see optical extra dtype'
            assert np.all(ins['type'] == 1), 'Only s1 type is supported for generating rawdata from optical input'

    def _setup(self):
        c = self.config
        time_zero = int((c['entry_start'] + 0.5) / c['event_rate'] * 1e9)
        if 'tpc' in c['targets']:
            self.sim = ChunkRawRecords(c, device=self.device)
            self.sim_iter = self.sim(self.instructions_epix, time_zero=time_zero)
        if 'nveto' in c['targets']:
            self.sim_nv = ChunkRawRecords(self.config_nveto, rawdata_generator=RawDataOptical, channels=self.nveto_channels,
                                          timings=self.nveto_timings, device=self.device)
            self.sim_nv.truth_buffer = _optical_truth_buffer(self._truth_dtype)
            self.sim_nv_iter = self.sim_nv(self.instructions_nveto, time_zero=time_zero)

    def infer_dtype(self):
        return {t: (instruction_dtype + self._truth_dtype) if 'truth' in t else raw_record_dtype(samples_per_record=DEFAULT_RECORD_LENGTH)
                for t in self.provides}

    def compute(self):
        """strax_interface.py:916-996: one chunk of each detector; a depleted detector follows the other's chunk times"""
        targets = self.config['targets']
        dt = self.infer_dtype()
        result, result_nv = None, None
        def depleted(sim):
            if not sim.source_finished():
                raise RuntimeError('Bug in getting source finished')            # the iterator ended before its source did
        if 'tpc' in targets:
            result = next(self.sim_iter, None)
            if result is None:
                depleted(self.sim)
                result = {t: np.zeros(0, dt[t]) for t in self.provides if 'nv' not in t}
                if 'nveto' in targets:
                    self.sim.chunk_time, self.sim.chunk_time_pre = self.sim_nv.chunk_time, self.sim_nv.chunk_time_pre
        if 'nveto' in targets:
            result_nv = next(self.sim_nv_iter, None)
            if result_nv is not None:
                shifted = result_nv['raw_records'].copy()
                shifted['channel'] += self.config['channel_map']['nveto'][0]
                result_nv = dict(result_nv, raw_records=shifted)
            else:
                depleted(self.sim_nv)
                result_nv = {t[:-3]: np.zeros(0, dt[t]) for t in self.provides if 'nv' in t}
                if 'tpc' in targets:
                    self.sim_nv.chunk_time, self.sim_nv.chunk_time_pre = self.sim.chunk_time, self.sim.chunk_time_pre
        exist_tpc = result is not None and any(len(result[t]) > 0 for t in self.provides if 'nv' not in t)
        exist_nv = result_nv is not None and any(len(result_nv[t[:-3]]) > 0 for t in self.provides if 'nv' in t)
        out = {}
        for t in self.provides:
            mine, other = (exist_nv, exist_tpc) if 'nv' in t else (exist_tpc, exist_nv)
            sim_mine, sim_other = (getattr(self, 'sim_nv', None), getattr(self, 'sim', None)) if 'nv' in t else (getattr(self, 'sim', None), getattr(self, 'sim_nv', None))
            if mine:
                data = result_nv[t[:-3]] if 'nv' in t else result[t]
                out[t] = self.chunk(start=sim_mine.chunk_time_pre, end=sim_mine.chunk_time, data=data, data_type=t)
            elif other:
                out[t] = self.chunk(start=sim_other.chunk_time_pre, end=sim_other.chunk_time, data=np.zeros(0, dt[t]), data_type=t)
            else:
                out[t] = self.chunk(start=0, end=0, data=np.zeros(0, dt[t]), data_type=t)
        self._sort_check([out[t].data for t in self.provides])
        return out

    def source_finished(self):
        sims = [sim for target, sim in (('tpc', getattr(self, 'sim', None)), ('nveto', getattr(self, 'sim_nv', None)))
                if target in self.config['targets']]
        return all(sim.source_finished() for sim in sims)
