"""GPU-backed ``RawData``: same call protocol as the reference's event scheduler + digitiser.

Reference: /root/reference/wfsim/core/rawdata.py:24-157 (``RawData.__call__``), :313-375 (``get_truth``).

    rd = RawData(config)
    for channel, left, right, data in rd(instructions, truth_buffer=tb):
        rd.left, rd.right            # sample bounds of the digitise window being emitted
    rd.source_finished

The reference simulates one time-cluster at a time in Python; here whole batches of clusters go through the HIP
pipeline (photon generation -> Pulse -> digitise -> ZLE -> record packing, wfsim_amd/csrc) and the generator
replays the results window by window, so a consumer written against the reference (``ChunkRawRecords``) cannot tell
the difference.  ``iter_windows`` is the same stream at window granularity with the records already packed on the
GPU (what the in-repo ``ChunkRawRecords`` uses).
"""
import logging

import os

import numpy as np

from .config import afterpulse_switches
from .dtypes import raw_record_dtype
from .engine import Engine
from .physics import instruction_params
from .resource import Resource
from . import electron_afterpulse as ea
from .scheduler import schedule, processing_order, run_sets, feedback_schedule

log = logging.getLogger('wfsim_amd.core')

PULSE_TYPE_NAMES = ('RESERVED', 's1', 's2', 'unknown', 'pi_el', 'pmt_ap', 'pe_el')


class RawData:
    #: upper bound on the expected photons of one GPU batch (sizes the HBM arenas; 288 GB HBM3E leaves room for ~10^10)
    max_batch_quanta = 2_000_000_000
    #: upper bound on the expected records of one batch (None: unbounded).  ChunkRawRecords sets it to half its record buffer
    #: so that a batch's records can be copied from the device straight into the buffer (strax_interface.py:360-364).
    record_budget = None

    def __init__(self, config, device=0, resource=None, seed=None):
        self.config = config
        self.resource = resource if resource is not None else Resource(config)
        self.engine = Engine(config, self.resource, device=device, seed=seed)
        # electron afterpulses (afterpulse.py:14-139): secondaries are made in a pre-pass over the primaries
        self._pi_hist = self._pi_grid = None
        if afterpulse_switches(config)['pmt'] and not hasattr(self.resource, 'uniform_to_pmt_ap'):
            raise ValueError('enable_pmt_afterpulses (on unless the config says otherwise, rawdata.py:176) needs the afterpulse '
                             'tables: config uniform_to_pmt_ap with enable_pmt_afterpulses set (load_resource.py:228)')
        if afterpulse_switches(config)['electron']:
            if not hasattr(self.resource, 'uniform_to_ele_ap'):
                raise ValueError('enable_electron_afterpulses (on unless the config says otherwise, rawdata.py:194) needs '
                                 'resource.uniform_to_ele_ap, the delay-time histogram (load_resource.py:233)')
            self._pi_hist = ea.DelayHistogram.wrap(self.resource.uniform_to_ele_ap)
            self._pi_grid = ea.coarse_delay_grid(self._pi_hist, config)
        self.source_finished = False
        self.left = self.right = 0
        #: run the kernels of the next batch while the consumer of iter_batches works on the current one
        self.prefetch = os.environ.get('WFSIM_AMD_PREFETCH', '1') != '0'
        #: run-wide index of every instruction passed to __call__ (RNG stream ids); None: position in the input
        self.global_ids = None
        #: chunk boundaries of the consumer (ns): batches end just behind them where that costs little (_batch_end)
        self.cut_period, self.cut_origin = None, 0

    @staticmethod
    def symtype(ptype):
        return PULSE_TYPE_NAMES[ptype]

    # ------------------------------------------------------------------------------------------
    def _batch_end(self, start, est_csum, cluster, scale=1, rec_csum=None, key=None):
        """End of the batch starting at sorted index ``start``: whole clusters, bounded by expected quanta.
        ``cut_period`` / ``cut_origin`` (set by the chunker): a batch that would run across the next chunk boundary ends just
        behind it instead, when that does not make it much smaller -- the chunk then closes at the end of a batch and its
        records leave as the buffer they arrived in (ChunkRawRecords._hand_out).  Results do not depend on the cuts."""
        n = len(cluster)
        limit = (est_csum[start - 1] if start else 0) + self.max_batch_quanta * scale
        stop = int(np.searchsorted(est_csum, limit, side='right'))
        if self.record_budget is not None and rec_csum is not None:
            rlimit = (rec_csum[start - 1] if start else 0) + self.record_budget * scale / self._rec_scale
            stop = min(stop, int(np.searchsorted(rec_csum, rlimit, side='right')))
        if self.cut_period and key is not None and scale == 1 and stop < n:
            boundary = self.cut_origin + (np.floor((int(key[start]) - self.cut_origin) / self.cut_period) + 1) * self.cut_period
            s = int(np.searchsorted(key, boundary, side='right')) + 1        # one instruction beyond: its window closes the chunk
            if start + (stop - start) // 2 <= s < stop:
                stop = s
        stop = max(stop, start + 1)
        while stop < n and cluster[stop] == cluster[stop - 1]:      # never cut a cluster
            stop += 1
        return stop

    def iter_windows(self, instructions, truth_buffer=None, **kwargs):
        """Yields dict(left, right, records) per digitise window, in time order.  ``records`` are packed strax
        raw_records in the order the reference yields pulses (channel ascending, interval ascending)."""
        if truth_buffer is None:
            truth_buffer = []
        self.engine.set_record_order(False)
        for batch in self.iter_batches(instructions, want_truth=len(truth_buffer) > 0, **kwargs):
            table, rows, before = batch['truth_table'], batch['truth_rows'], batch['truth_before']
            k = 0
            n_win = len(batch['left'])
            for w in range(n_win):
                k1 = int(np.searchsorted(before, w, side='right'))      # the rows of this window's clusters (and of empty ones before it)
                if k1 > k:
                    self._write_truth(table, rows[k:k1], truth_buffer)
                    k = k1
                self.left, self.right = int(batch['left'][w]), int(batch['right'][w])
                if batch['finished'] and w == n_win - 1:
                    self.source_finished = True
                yield dict(left=self.left, right=self.right, records=batch['records'][batch['first'][w]:batch['first'][w + 1]])
            if k < len(rows):
                self._write_truth(table, rows[k:], truth_buffer)
        self.source_finished = True

    def iter_batches(self, instructions, want_truth=False, record_sink=None, device_records=False, **kwargs):
        """The same stream batch by batch (what the chunker consumes): dict(left[], right[] of the batch's digitise windows
        in time order, first[] record offsets (one more than windows), records, truth_table + truth_rows + truth_before (row truth_rows[k] of the table belongs in
        front of window truth_before[k]; == number of windows: after the last), finished).  ``record_sink(n, first_left)`` may hand out
        the array the records are copied into (device -> host without a staging copy); ``device_records``: the records stay
        on the GPU (``records`` is a torch uint8 tensor of n * 244 bytes)."""
        self.source_finished = False
        if len(instructions) == 0:
            self.source_finished = True
            return
        cfg = self.config
        dt, tw, rext = cfg['sample_duration'], cfg['trigger_window'], cfg['right_raw_extension']
        gids = np.arange(len(instructions)) if self.global_ids is None else np.asarray(self.global_ids)
        em_base = np.zeros(len(instructions), dtype=np.uint32)
        self._all_run_sets = None
        if self._has_electron_afterpulses():
            # pre-pass: the secondaries of every S2 (type 4 / 6 instructions); then primaries and secondaries go through the
            # normal path in the order, clusters and pulse sets of the reference's feedback loop (rawdata.py:70-151)
            sec, sec_gid, sec_base, sec_parent = self.electron_afterpulse_instructions(instructions, gids, with_parent=True)
            parent = np.concatenate([np.full(len(instructions), -1, dtype=np.int64), sec_parent])
            instructions = np.concatenate([instructions, sec])
            gids, em_base = np.concatenate([gids, sec_gid]), np.concatenate([em_base, sec_base])
            order, key, cluster, self._all_run_sets = feedback_schedule(instructions, parent, cfg)
        else:
            order, key, cluster = schedule(instructions, cfg)
        s_ins = instructions[order]
        n = len(s_ins)
        self.instruction_event_number = np.min(instructions['event_number'])
        # crude photon estimate per instruction: only used to bound a batch
        est_csum = np.cumsum(self._expected_quanta(s_ins))
        rec_csum = np.cumsum(self._expected_records(s_ins))
        self._rec_scale = 1.0
        # smallest key of every cluster (in feedback order the first instruction of a cluster need not carry it)
        cl_start = np.concatenate([[0], np.where(np.diff(cluster) != 0)[0] + 1])
        cl_min_key = np.minimum.reduceat(key, cl_start)
        st = dict(s_ins=s_ins, gids=gids, order=order, key=key, cluster=cluster, em_base=em_base, est_csum=est_csum, rec_csum=rec_csum,
                  cl_min_key=cl_min_key, n=n)
        # the kernels of the next batch run (from a worker thread: the engine calls drop the GIL) while the consumer works on
        # this one.  The worker owns the engine meanwhile; the one call the consumer's thread makes during that time is
        # engine.wait_records(), which only waits for the copy stream and touches no state of the handle (wfs_wait_records)
        pool = None
        if self.prefetch and n > 1:
            from concurrent.futures import ThreadPoolExecutor
            pool = ThreadPoolExecutor(1)
        launched, future = self._launch(st, 0, False, 0), None
        try:
            while True:
                L = launched
                ins, cl, n_emit, first, nonempty, ins_group, groups = L['ins'], L['cl'], L['n_emit'], L['first'], L['nonempty'], L['ins_group'], L['groups']
                n_rec = int(first[n_emit])
                keep = np.where(nonempty[:n_emit])[0]
                out = None
                if record_sink is not None and not device_records:
                    # (the sink is told where the batch's first window starts: the chunker can tell a chunk that is about to close)
                    out = record_sink(n_rec, int(groups['left'][keep[0]]) if len(keep) else None)
                if out is None and not device_records:
                    out = np.empty(n_rec, dtype=raw_record_dtype())
                # window position of every emitted group: rows of a group go in front of its window, rows of a group without
                # pulses in front of the next window that has some
                pos_of_group = np.searchsorted(keep, np.arange(n_emit), side='left')
                truth_table, truth_rows, truth_before = None, np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
                if want_truth:
                    truth_table = self._truth_rows(ins, cl, L['run_set'])           # one row per run set, in processing order
                    grp = ins_group[truth_table['first']].astype(np.int64)
                    truth_rows = np.flatnonzero(grp < n_emit)
                    truth_before = pos_of_group[grp[truth_rows]] if n_emit else np.zeros(0, dtype=np.int64)
                # the records travel on the engine's copy stream: the next batch's kernels are started first, then the copy is
                # awaited (the small copies above come first: behind 0.6 GB of records they would wait for them)
                if device_records:
                    # the records stay on the GPU (a torch uint8 tensor of n_rec * 244 bytes): the multi-GPU gather sends them
                    # from there (distributed.simulate_sharded)
                    import torch
                    if not torch.cuda.is_initialized():
                        try:
                            torch.cuda.init()
                        except RuntimeError as e:
                            raise RuntimeError('device_records needs torch.cuda initialised BEFORE the first wfsim_amd Engine of the process is '
                                               'created (import torch; torch.cuda.init()): torch brings its own HIP runtime') from e
                    records = torch.empty(n_rec * np.dtype(raw_record_dtype()).itemsize, dtype=torch.uint8,
                                          device=torch.device('cuda', self.engine.device))
                    if n_rec:
                        self.engine.copy_records_to_device(records.data_ptr(), n_rec)
                else:
                    records = self.engine.records_into_async(out, n_rec)
                batch = dict(left=groups['left'][keep], right=groups['right'][keep],
                             first=np.append(first[keep], first[n_emit]) if len(keep) else np.array([first[n_emit]]),
                             records=records, truth_table=truth_table, truth_rows=truth_rows, truth_before=truth_before, finished=L['b'] >= n)
                # everything else of this batch is on the host now: the engine is free for the next one (it keeps the records of
                # two batches: this batch's copy overlaps the next batch's kernels)
                if L['b'] < n:
                    if pool is not None:
                        future = pool.submit(self._launch, st, L['b'], L['has_pulse'], L['runmax'])
                self.engine.wait_records()
                yield batch
                if L['b'] >= n:
                    break
                launched = future.result() if future is not None else self._launch(st, L['b'], L['has_pulse'], L['runmax'])
                future = None
        finally:
            if future is not None:          # the consumer stopped early: let the running batch finish before the engine is reused
                try:
                    future.result()
                except Exception:
                    pass
            if pool is not None:
                pool.shutdown(wait=True)
        self.source_finished = True

    def _launch(self, st, a, has_pulse, runmax):
        """Loads and runs the batch that starts at sorted instruction ``a``; decides how much of it can be emitted."""
        cfg = self.config
        dt, tw, rext = cfg['sample_duration'], cfg['trigger_window'], cfg['right_raw_extension']
        s_ins, gids, order, key, cluster, n = st['s_ins'], st['gids'], st['order'], st['key'], st['cluster'], st['n']
        scale = 1
        while True:
            b = self._batch_end(a, st['est_csum'], cluster, scale, st.get('rec_csum'), key if self._all_run_sets is None else None)
            ins = s_ins[a:b]
            gid = gids[order[a:b]].astype(np.uint32)
            cl = (cluster[a:b] - cluster[a]).astype(np.int32)
            self.engine.set_window_carry(has_pulse, runmax)
            self._batch_em_base = st['em_base'][order[a:b]]
            self._batch_run_set = None if self._all_run_sets is None else self._all_run_sets[a:b] - self._all_run_sets[a]
            self._load_batch(ins, gid, cl, key[a:b])
            counts = self.engine.run()
            groups = self.engine.groups()
            first = np.append(groups['first_record'], counts['n_records'])
            if st.get('rec_csum') is not None:           # records per expected record of this batch, with a margin: sizes the next one
                est = st['rec_csum'][b - 1] - (st['rec_csum'][a - 1] if a else 0)
                if est > 0:
                    self._rec_scale = float(min(8.0, max(0.125, 1.3 * counts['n_records'] / est)))
            cl_group = self.engine.cluster_groups(int(cl[-1]) + 1)
            ins_group = cl_group[cl]
            n_groups = len(groups['left'])
            nonempty = groups['right'] >= groups['left']
            ends = np.where(nonempty, (groups['right'] - tw) * dt, np.iinfo(np.int64).min)   # max(pulse right) * dt per window
            n_emit = n_groups
            if b < n and nonempty.any():
                # would the next cluster have been simulated before this batch's last window was digitised
                # (rawdata.py:96-98)?  then that window is not complete: simulate it again with the next batch
                run_all = max(int(ends.max()), runmax) if has_pulse else int(ends.max())
                if not (int(st['cl_min_key'][cluster[b] - cluster[0]]) - run_all > rext):
                    g_last = int(np.where(nonempty)[0][-1])
                    restart = a + int(np.argmax(ins_group >= g_last))
                    if restart == a:            # the whole batch is one open window: take a bigger batch
                        scale *= 2
                        continue
                    n_emit, b = g_last, restart
            keep = np.where(nonempty[:n_emit])[0]
            if len(keep):
                e = ends[keep]
                runmax = max(int(e.max()), runmax) if has_pulse else int(e.max())
                has_pulse = True
            return dict(b=b, ins=ins, cl=cl, n_emit=n_emit, first=first, nonempty=nonempty, ins_group=ins_group, groups=groups,
                        run_set=getattr(self, '_run_set', None), has_pulse=has_pulse, runmax=runmax)

    def _expected_quanta(self, s_ins):
        return np.where(s_ins['type'] == 1, s_ins['amp'] * 0.15,
                        s_ins['amp'] * float(self.config.get('s2_secondary_sc_gain', 30)))     # types 2, 4, 6: electrons

    def _expected_records(self, s_ins):
        """crude: a record per quantum for small signals, up to ~5 records on every PMT plus one per 400 quanta for big ones"""
        q = self._expected_quanta(s_ins)
        return np.minimum(q, 5 * len(self.config.get('gains', np.zeros(494))) + q / 400)

    def _load_batch(self, ins, gid, cl, key):
        ip = instruction_params(ins, self.config, self.engine.resource, gids=gid, device_maps=self.engine.device_maps, device_aft=self.engine._aft_on_device)
        # one pulse set per instruction, or -- save_full_truth off -- per group of nearby S1s / S2s (rawdata.py:106-127)
        # ...; electron-afterpulse instructions (types 4 / 6) of a cluster always share one call
        plain = self.config.get('save_full_truth', True) and bool(np.all(ins['type'] <= 2))
        given = getattr(self, '_batch_run_set', None)                   # electron afterpulses: from the feedback schedule
        self._run_set = given if given is not None else (None if plain else run_sets(ins, key, cl, self.config)[0])
        self.engine.load_instructions(ins, gid, cl, key, ip, run_set=self._run_set, em_base=getattr(self, '_batch_em_base', None))

    # ---- electron afterpulses (afterpulse.py:14-139, rawdata.py:192-202) ------------------------------
    def _has_electron_afterpulses(self):
        sw = afterpulse_switches(self.config)
        return sw['electron'] or sw['gate']

    def electron_afterpulse_instructions(self, instructions, gids, with_parent=False):
        """Pre-pass: photons of the primaries only (no pulses), then for every S2 pulse set its secondary instructions
        (type 4 photo-ionisation / type 6 gate electrons).  Returns (secondaries, their gid, their emitter offsets)."""
        cfg = self.config
        order, key, cluster = schedule(instructions, cfg)
        s_ins, s_gid = instructions[order], np.asarray(gids)[order]
        n = len(s_ins)
        est_csum = np.cumsum(self._expected_quanta(s_ins))
        out, out_gid, out_base, out_parent = [], [], [], []
        a = 0
        while a < n:
            b = self._batch_end(a, est_csum, cluster)
            ins, gid = s_ins[a:b], s_gid[a:b]
            cl = (cluster[a:b] - cluster[a]).astype(np.int32)
            self._batch_em_base = self._batch_run_set = None
            self._load_batch(ins, gid.astype(np.uint32), cl, key[a:b])
            self.engine.generate()
            ph_off = self.engine.instruction_photon_offsets()           # generation order: instruction by instruction
            rs = np.arange(len(ins)) if self._run_set is None else self._run_set
            members = {}
            for i, q in enumerate(rs):
                members.setdefault(int(q), []).append(i)
            plans, req_idx = [], []
            for q, m in sorted(members.items()):
                i = m[0]                                                 # signal_pulse_instruction[0] (afterpulse.py:49)
                if ins['type'][i] != 2:                                  # only S2s make electron afterpulses (rawdata.py:194-200)
                    continue
                n_ph = np.array([ph_off[k + 1] - ph_off[k] for k in m])  # the call's photons: its instructions one after the other
                p = ea.plan_secondaries(ins[i:i + 1], int(gid[i]), int(n_ph.sum()), cfg, self._pi_hist, self._pi_grid)
                starts = np.concatenate([[0], np.cumsum(n_ph)])
                for pl in p:
                    which = np.searchsorted(starts, pl[1], side='right') - 1          # picked photon -> instruction of the set
                    req_idx.append(ph_off[np.asarray(m)[which]] + (pl[1] - starts[which]))
                if p:
                    plans.append((i, p))
            t_all = self.engine.gather_photon_times(np.concatenate(req_idx)) if req_idx else np.zeros(0, np.int64)
            pos = 0
            for i, p in plans:
                tz = []
                for pl in p:
                    tz.append(t_all[pos:pos + len(pl[1])]); pos += len(pl[1])
                sec = ea.build_instructions(ins[i:i + 1], p, tz, cfg)
                if len(sec) > ea.MAX_SECONDARIES_PER_PARENT:
                    raise ValueError(f'{len(sec)} electron-afterpulse instructions from one S2: more than the stream ids allow')
                out.append(sec); out_gid.append(np.full(len(sec), gid[i])); out_base.append(((np.arange(len(sec)) + 1) << 20).astype(np.uint32))
                out_parent.append(np.full(len(sec), order[a + i]))       # index of the parent (first instruction of its set) in the input
            a = b
        if not out:
            res = (np.zeros(0, dtype=instructions.dtype), np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.uint32), np.zeros(0, dtype=np.int64))
        else:
            res = (np.concatenate(out), np.concatenate(out_gid).astype(np.int64), np.concatenate(out_base), np.concatenate(out_parent).astype(np.int64))
        return res if with_parent else res[:3]

    # ---- truth (rawdata.py:313-375) ----------------------------------------------------------------
    def _truth_rows(self, ins, cl, run_set=None):
        """Truth of the batch, one row per pulse set in processing order, as columns (RawData.get_truth, rawdata.py:313-375):
        dict(first = index of the set's first instruction in the batch, n = rows, cols = {truth field: array})."""
        acc, ts = self.engine.truth()
        es = self.engine.electron_stats()
        per_pmt = self.engine.truth_per_pmt() if self.config.get('per_pmt_truth', False) else None      # pulse.py:62-66
        names = ['n_photon', 'n_pe', 'n_photon_trigger', 'n_pe_trigger', 'raw_area', 'raw_area_trigger']
        if run_set is None:
            # every instruction is its own pulse set (set index = position in the sorted batch); rows in processing order
            first = np.asarray(processing_order(ins, np.arange(len(ins)), cl), dtype=np.int64)
            set_ids, row = first, ins[first].copy()
        else:
            order = np.argsort(run_set, kind='stable')
            starts = np.concatenate([[0], np.where(np.diff(run_set[order]) != 0)[0] + 1]) if len(order) else np.zeros(0, np.int64)
            first = order[starts]
            set_ids = run_set[first].astype(np.int64)
            row = ins[first].copy()
            counts = np.diff(np.append(starts, len(order)))
            multi = np.where(counts > 1)[0]
            for k in multi:         # rawdata.py:364-372: mean position, summed amp, everything else from the set's first instruction
                m = order[starts[k]:starts[k] + counts[k]]
                for f in ('x', 'y', 'z'):
                    row[f][k] = np.mean(ins[f][m])
                row['amp'][k] = np.sum(ins['amp'][m])
        q = set_ids
        cols = {}
        n_ph = ts[q, 0]
        has = n_ph > 0
        cols['n_photon_t'] = np.where(has, n_ph, 0)
        for j, f in enumerate(['t_mean_photon', 't_first_photon', 't_last_photon', 't_sigma_photon'], start=1):
            cols[f] = np.where(has, ts[q, j], np.nan)
        is_s2 = row['type'] % 2 == 0
        n_el = np.where(is_s2, es[q, 0], 0)
        he = n_el > 0
        cols['n_electron'] = np.where(he, n_el, 0)
        for j, f in enumerate(['t_mean_electron', 't_first_electron', 't_last_electron', 't_sigma_electron'], start=1):
            cols[f] = np.where(he, es[q, j], np.nan)
        cfg = self.config
        tail = (cfg['samples_before_pulse_center'] + cfg['samples_after_pulse_center'] + 1) * cfg['sample_duration']
        cols['endtime'] = np.where(has, cols['t_last_photon'] + tail, row['time']).astype(np.float64)
        for j, f in enumerate(names):
            cols[f] = acc[q, j]
            cols[f + '_bottom'] = acc[q, 6 + j]
            if per_pmt is not None:
                cols[f + '_per_pmt'] = per_pmt[q, :, j]
        # mean observed position of the electrons under a field distortion model (get_mean_xy_electron, rawdata.py:377-390)
        cols['x_mean_electron'] = np.full(len(first), np.nan)
        cols['y_mean_electron'] = np.full(len(first), np.nan)
        if cfg.get('field_distortion_model', 'none') in ('comsol', 'inverse_fdc'):
            from .physics import s2_observed_positions
            s2rows = np.where(row['type'] == 2)[0]
            if len(s2rows):
                if run_set is None:
                    _, xy = s2_observed_positions(ins[first[s2rows]], cfg, self.resource)
                    cols['x_mean_electron'][s2rows], cols['y_mean_electron'][s2rows] = xy[:, 0], xy[:, 1]
                else:
                    for k in s2rows:
                        _, xy = s2_observed_positions(ins[order[starts[k]:starts[k] + counts[k]]], cfg, self.resource)
                        cols['x_mean_electron'][k], cols['y_mean_electron'][k] = np.mean(xy[:, 0]), np.mean(xy[:, 1])
        # zero-photon electron afterpulses leave no truth row (rawdata.py:336-338)
        keep = ~((~has) & ~np.isin(row['type'], (1, 2)))
        return dict(first=first, n=len(first), cols=cols, instruction=row, keep=keep)

    def _write_truth(self, table, rows, truth_buffer):
        """rows of ``table`` (indices, in order) into the first free rows of the truth buffer (rawdata.py:320: np.argmin(fill)
        per row = the free slots in ascending order)"""
        rows = np.asarray(rows, dtype=np.int64)
        rows = rows[table['keep'][rows]]
        if len(rows) == 0:
            return
        free = np.flatnonzero(~truth_buffer['fill'])
        if len(free) < len(rows):
            # the reference keeps writing into row 0 once the buffer is full (argmin of an all-True array); so do we
            free = np.concatenate([free, np.zeros(len(rows) - len(free), dtype=np.int64)])
        slots = free[:len(rows)]
        cols, ins = table['cols'], table['instruction'][rows]
        names = truth_buffer.dtype.names
        tb = truth_buffer
        tb['n_photon'][slots] = cols['n_photon_t'][rows]
        for f in ['t_mean_photon', 't_first_photon', 't_last_photon', 't_sigma_photon', 'n_electron', 't_mean_electron',
                  't_first_electron', 't_last_electron', 't_sigma_electron']:
            tb[f][slots] = cols[f][rows]
        tb['x_mean_electron'][slots] = cols['x_mean_electron'][rows]
        tb['y_mean_electron'][slots] = cols['y_mean_electron'][rows]
        tb['endtime'][slots] = cols['endtime'][rows]
        for f in ['n_pe', 'n_pe_trigger', 'n_photon', 'n_photon_trigger', 'raw_area', 'raw_area_trigger']:
            for suffix in ['', '_bottom', '_per_pmt']:          # rawdata.py:355-362: total + (bottom | per PMT)
                if f + suffix in names and f + suffix in cols:
                    tb[f + suffix][slots] = cols[f + suffix][rows]
        for f in ins.dtype.names:
            if f in names:
                tb[f][slots] = ins[f]
        tb['fill'][slots] = True

    # ---- reference protocol ------------------------------------------------------------------------
    def __call__(self, instructions, truth_buffer=None, progress_bar=True, **kwargs):
        rec_dtype = np.dtype(raw_record_dtype())
        spr = rec_dtype['data'].shape[0]
        for w in self.iter_windows(instructions, truth_buffer, **kwargs):
            rec = w['records']
            k = 0
            while k < len(rec):
                # the fragments of one ZLE interval are consecutive: record_i = 0 .. ceil(pulse_length / spr) - 1
                plen = int(rec['pulse_length'][k])
                nfrag = -(-plen // spr)
                data = rec['data'][k:k + nfrag].reshape(-1)[:plen].astype(np.int64)
                left = int(rec['time'][k]) // int(rec['dt'][k])
                yield int(rec['channel'][k]), left, left + plen - 1, data
                k += nfrag


class RawDataOptical(RawData):
    """Photon channels / timings supplied up front (nVeto, optical Geant4 input): rawdata.py:461-495.

    ``instructions`` carry ``_first`` / ``_last`` into the flat ``channels`` / ``timings`` arrays; only S1-type
    instructions exist in this mode (strax_interface.py:903-904)."""

    def __init__(self, config, channels=tuple(), timings=tuple(), device=0, resource=None, seed=None):
        super().__init__(config, device=device, resource=resource, seed=seed)
        self.channels = np.asarray(channels)
        self.timings = np.asarray(timings)

    def _expected_quanta(self, s_ins):
        return (s_ins['_last'] - s_ins['_first']).astype(np.float64)

    def _load_batch(self, ins, gid, cl, key):
        assert np.all(ins['type'] == 1), 'Only s1 type is supported for generating rawdata from optical input'
        cutoff = self.config.get('nveto_time_max_cutoff', int(1e6))
        self.engine.load_optical(ins, gid, cl, key, self.channels, self.timings, cutoff)
