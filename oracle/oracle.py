"""ctypes front end of the CPU oracle (oracle/wfsim_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Nothing under wfsim_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, '_build', 'liboracle.so')

_I32_FIELDS = ['dt', 'samples_before', 'samples_after', 'store_before', 'store_after', 'tlen',
               'trigger_window', 'baseline', 'n_rows',
               'n_tpc', 'n_top', 'he_first', 'he_factor', 'sum_channel', 'last_bottom', 'detector_nt',
               'n_spe_channels', 'noise_len', 'noise_channels', 'enable_noise',
               's1_simple', 's2_time_model', 'n_lum', 'enable_pmt_ap', 'n_ap_elements', 'tile_gen', 'tile_gen_min', 'fma']
_F64_FIELDS = ['c2a', 'tts_mean', 'tts_sigma', 'p_dpe', 's1_decay_time', 's1_decay_spread',
               'sf_gas', 't1_gas', 't3_gas', 's2_time_spread', 'trap_time', 'gain_spread',
               'pmt_ap_modifier', 'pmt_ap_t_modifier', 'rext', 'drift_velocity']


class OrcConfig(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in _I32_FIELDS] + [(n, C.c_double) for n in _F64_FIELDS]
                + [('seed', C.c_uint64)])


def build(force=False):
    src = os.path.join(HERE, 'wfsim_oracle.c')
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', HERE, '-s', '-B'])
    return LIB_PATH


_lib = None

_GETTERS = dict(
    pl_ch=np.int32, pl_runset=np.int32, pl_left=np.int64, pl_right=np.int64, pl_cur_off=np.int64, pl_nph=np.int64,
    cur=np.float64, ph_t=np.int64, ph_ch=np.int16, ph_dpe=np.uint8, ph_gain=np.float64, call_ph_off=np.int64,
    call_kind=np.int32, call_runset=np.int32, e_t=np.int64, call_e_off=np.int64,
    dg_left=np.int64, dg_right=np.int64, dg_first_pulse=np.int64, dg_n_pulses=np.int64, dg_ix_rand=np.int64,
    dg_row_off=np.int64, row_ch=np.int32, row_left=np.int64, row_right=np.int64, row_data_off=np.int64,
    row_data=np.int32, zl_digit=np.int64, zl_ch=np.int32, zl_left=np.int64, zl_right=np.int64,
    zl_data_off=np.int64, zl_data=np.int32, truth=np.float64)


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
        _lib.orc_new.restype = C.c_void_p
        _lib.orc_new.argtypes = [C.POINTER(OrcConfig)] + [C.c_void_p] * 8
        _lib.orc_free.argtypes = [C.c_void_p]
        _lib.orc_n_pe.restype = C.c_int64
        _lib.orc_n_pe.argtypes = [C.c_void_p]
        _lib.orc_pack_records.restype = C.c_int64
        _lib.orc_pack_records.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]
        _lib.orc_find_intervals_below_threshold.restype = C.c_int64
        for g in _GETTERS:
            f = getattr(_lib, 'orc_' + g)
            f.restype = C.c_void_p
            f.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    return _lib


def _p(a):
    return C.c_void_p(a.ctypes.data) if a is not None else C.c_void_p(0)


def _arr(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


class Oracle:
    """One simulation session.  ``params``: scalars named like OrcConfig fields; ``tables``: dict with
    templates[10,tlen], spe[n,2001], gains[n_tpc], thr_truth[n_rows], thr_zle[n_rows], lum_x, lum_t, noise."""

    def __init__(self, params, tables, ap_tables=None):
        L = lib()
        t = self.tables = dict(
            templates=_arr(tables['templates'], np.float64), spe=_arr(tables['spe'], np.float64),
            gains=_arr(tables['gains'], np.float64), thr_truth=_arr(tables['thr_truth'], np.float64),
            thr_zle=_arr(tables['thr_zle'], np.int64),
            lum_x=_arr(tables.get('lum_x', np.array([0.0, 1.0])), np.float64),
            lum_t=_arr(tables.get('lum_t', np.array([0.0, 0.0])), np.float64),
            noise=_arr(tables['noise'], np.int16) if tables.get('noise') is not None else None)
        cfg = OrcConfig()
        p = dict(params)
        p['n_spe_channels'] = t['spe'].shape[0]
        p['n_lum'] = len(t['lum_x'])
        p['noise_len'], p['noise_channels'] = (t['noise'].shape if t['noise'] is not None else (0, 0))
        if t['noise'] is None:
            p['enable_noise'] = 0
        p['n_ap_elements'] = len(ap_tables) if ap_tables else 0
        if not ap_tables:
            p['enable_pmt_ap'] = 0
        for n, _ in OrcConfig._fields_:
            if n in p:
                setattr(cfg, n, p[n])
        self.cfg = cfg
        self._s = C.c_void_p(L.orc_new(C.byref(cfg), _p(t['templates']), _p(t['spe']), _p(t['gains']), _p(t['thr_truth']),
                                       _p(t['thr_zle']), _p(t['lum_x']), _p(t['lum_t']), _p(t['noise'])))
        self._keep = []
        nz = tables.get('noise')
        if nz is not None and np.asarray(nz).dtype.kind == 'f' and not np.array_equal(np.asarray(nz), np.trunc(nz)):
            self._noise_f = _arr(nz, np.float64)                # a float noise array: the truncated sum is stored (rawdata.py:436)
            L.orc_set_noise_float(self._s, _p(self._noise_f))
        if ap_tables:
            for e, (name, d) in enumerate(ap_tables.items()):
                dc = _arr(d['delaytime_cdf'], np.float64)
                ac = _arr(d['amplitude_cdf'], np.float64)
                self._keep += [dc, ac]
                L.orc_set_ap_element(self._s, C.c_int(e), C.c_int(dc.shape[1]), C.c_int(ac.shape[-1]), C.c_int(ac.ndim == 2),
                                     C.c_int('Uniform' in name), C.c_double(d['delaytime_bin_size']),
                                     C.c_double(d['amplitude_bin_size']), _p(dc), _p(ac))

    def close(self):
        if self._s:
            lib().orc_free(self._s)
            self._s = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- stages -------------------------------------------------------------------------------
    @staticmethod
    def add_current(t, g, pulse_left, dt, templates, length, fma=False):
        t = _arr(t, np.int64)
        g = _arr(g, np.float64)
        T = _arr(templates, np.float64)
        cur = np.zeros(length, dtype=np.float64)
        (lib().orc_add_current_fma if fma else lib().orc_add_current)(_p(t), _p(g), C.c_int64(len(t)), C.c_int64(pulse_left), C.c_int64(dt), _p(T),
                              C.c_int64(T.shape[1]), _p(cur))
        return cur

    @staticmethod
    def find_intervals_below_threshold(w, threshold, holdoff, size=50000):
        w = _arr(w, np.int64)
        res = np.zeros((size, 2), dtype=np.int64)
        n = lib().orc_find_intervals_below_threshold(_p(w), C.c_int64(len(w)), C.c_int64(threshold), C.c_int64(holdoff),
                                                     _p(res), C.c_int64(size))
        return res[:n]

    def pulse_call(self, kind, runset, t, ch, dpe, gain, gains_preassigned=False):
        t, ch, dpe, gain = _arr(t, np.int64), _arr(ch, np.int16), _arr(dpe, np.uint8), _arr(gain, np.float64)
        lib().orc_pulse_call(self._s, C.c_int(kind), C.c_int(runset), C.c_int64(len(t)), _p(t), _p(ch), _p(dpe), _p(gain),
                             C.c_int(int(gains_preassigned)))

    def set_noise_override(self, ix_rand):
        self._noise_override = _arr(ix_rand, np.int64)
        lib().orc_set_noise_override(self._s, _p(self._noise_override), C.c_int64(len(self._noise_override)))

    def set_save_full_truth(self, on):
        """rawdata.py:42: False groups S1s within 100 ns / S2s within 2 mm into one Pulse call"""
        lib().orc_set_save_full_truth(self._s, C.c_int(int(bool(on))))

    def digitize_and_zle(self, noise_gid=0):
        lib().orc_digitize_and_zle(self._s, C.c_uint32(noise_gid))

    def set_delay_models(self, models):
        """models: wfsim_amd.delay_models.DelayModels (the extra delay terms as probability mass functions + spline nodes)"""
        self._models = models if models is not None and models.active else None
        if self._models is None:
            return
        base, off, pmf, vmin = models.table_arrays()
        lib().orc_set_delay_models(self._s, C.c_int32(len(base)), _p(base), _p(off), _p(pmf), _p(vmin))
        sp = models.s1_prop
        if sp is not None:
            lib().orc_set_s1_propagation(self._s, C.c_int32(len(sp['z'])), C.c_int32(sp['nu']), C.c_double(sp['u0']), C.c_double(sp['du']),
                                         _p(sp['top']), _p(sp['bottom']))
        if models.gas_gap is not None:
            inv = models.gas_gap['inv']
            lib().orc_set_gas_gap_model(self._s, C.c_int32(inv.shape[0]), C.c_int32(inv.shape[1]), _p(inv))

    def _instruction_models(self, instructions, gid):
        if getattr(self, '_models', None) is None:
            lib().orc_set_instruction_models(self._s, C.c_int64(0), C.c_void_p(0), C.c_void_p(0), C.c_void_p(0), C.c_void_p(0))
            return
        self._im = self._models.instruction_tables(instructions, gid)       # kept alive: the C side keeps the pointers
        if getattr(self._models, 'per_batch', False):
            base, off, pmf, vmin = self._models.table_arrays()
            lib().orc_set_delay_models(self._s, C.c_int32(len(base)), _p(base), _p(off), _p(pmf), _p(vmin))
        lib().orc_set_instruction_models(self._s, C.c_int64(len(instructions)), *[_p(x) for x in self._im])
        if self._models.gas_gap is not None:
            self._igg = self._models.instruction_gas_gap(instructions)       # kept alive, as above
            lib().orc_set_instruction_gas_gap(self._s, C.c_int64(len(instructions)), _p(self._igg[0]), _p(self._igg[1]))

    def simulate(self, instructions, gid, ip, em_base=None):
        n = len(instructions)
        self._instruction_models(instructions, gid)
        a = dict(type=_arr(instructions['type'], np.int8), time=_arr(instructions['time'], np.int64),
                 z=_arr(instructions['z'], np.float32), amp=_arr(instructions['amp'], np.int32),
                 gid=_arr(gid, np.uint32), p_hit=_arr(ip['p_hit'], np.float64),
                 drift_mean=_arr(ip['drift_mean'], np.float64), drift_spread=_arr(ip['drift_spread'], np.float64),
                 sc_gain=_arr(ip['sc_gain'], np.float64), cdf_row=_arr(ip['cdf_row'], np.int32),
                 cdf_table=_arr(ip['cdf_table'], np.float64))
        lib().orc_simulate(self._s, C.c_int64(n), *[_p(a[k]) for k in
                           ['type', 'time', 'z', 'amp', 'gid', 'p_hit', 'drift_mean', 'drift_spread', 'sc_gain',
                            'cdf_row', 'cdf_table']], _p(_arr(em_base, np.uint32) if em_base is not None else None))

    def simulate_scheduled(self, instructions, gid, ip, em_base, cluster, tmin, run_set):
        """instructions in processing order with their clusters, window-rule keys and pulse sets (feedback_schedule)"""
        n = len(instructions)
        self._instruction_models(instructions, gid)
        a = [_arr(instructions['type'], np.int8), _arr(instructions['time'], np.int64), _arr(instructions['amp'], np.int32),
             _arr(gid, np.uint32), _arr(ip['p_hit'], np.float64), _arr(ip['drift_mean'], np.float64), _arr(ip['drift_spread'], np.float64),
             _arr(ip['sc_gain'], np.float64), _arr(ip['cdf_row'], np.int32), _arr(ip['cdf_table'], np.float64), _arr(em_base, np.uint32),
             _arr(cluster, np.int32), _arr(tmin, np.int64), _arr(run_set, np.int32)]
        lib().orc_simulate_scheduled(self._s, C.c_int64(n), *[_p(x) for x in a])

    def simulate_optical(self, instructions, gid, channels, timings, cutoff):
        a = dict(time=_arr(instructions['time'], np.int64), gid=_arr(gid, np.uint32), first=_arr(instructions['_first'], np.int32),
                 last=_arr(instructions['_last'], np.int32), channels=_arr(channels, np.int32), timings=_arr(timings, np.int64))
        lib().orc_simulate_optical(self._s, C.c_int64(len(instructions)), _p(a['time']), _p(a['gid']), _p(a['first']),
                                   _p(a['last']), _p(a['channels']), _p(a['timings']), C.c_int64(cutoff))

    def pack_records(self, samples_per_record=110):
        n = lib().orc_pack_records(self._s, C.c_int64(samples_per_record), C.c_void_p(0), C.c_int64(0))
        out = np.zeros(n * (24 + 2 * samples_per_record), dtype=np.uint8)
        lib().orc_pack_records(self._s, C.c_int64(samples_per_record), _p(out), C.c_int64(n))
        return out

    def sample_term(self, kind, n, p0=0.0, p1=0.0):
        out = np.zeros(n, dtype=np.int64)
        lib().orc_sample_term(self._s, C.c_int(kind), C.c_int64(n), C.c_double(p0), C.c_double(p1), _p(out))
        return out

    def sample_delay(self, n, is_s2, tab=-1, bottom=False, pzi=-1, pzf=0.0):
        out = np.zeros(n, dtype=np.int64)
        lib().orc_sample_delay(self._s, C.c_int64(n), C.c_int(int(is_s2)), C.c_int32(tab), C.c_int(int(bottom)), C.c_int32(pzi),
                               C.c_double(pzf), _p(out))
        return out

    def alias_pmf(self, slot=8, xtab=-1):
        """(pmf the alias table of a delay table samples, pmf of its cumulative table, vmin); slot: 0 transit time, 7 / 8 the
        summed delay of an S1 / S2 photon; xtab >= 0: a model-variant table"""
        cap = 1 << 17
        pa, pc = np.zeros(cap), np.zeros(cap)
        n, vmin = C.c_int64(0), C.c_int64(0)
        lib().orc_alias_pmf(self._s, C.c_int(slot), C.c_int32(xtab), _p(pa), _p(pc), C.c_int64(cap), C.byref(n), C.byref(vmin))
        k = 2
        while k < n.value:
            k *= 2
        return pa[:k], pc[:k], int(vmin.value)

    def chan_alias_pmf(self, cdf_row):
        """the channel probabilities the alias cells of a cumulative row encode"""
        row = _arr(cdf_row, np.float64)
        out = np.zeros(len(row))
        lib().orc_chan_alias_pmf(self._s, _p(row), _p(out))
        return out

    def sample_channels(self, cdf_row, n):
        row = _arr(cdf_row, np.float64)
        out = np.zeros(n, dtype=np.int32)
        lib().orc_sample_channels(self._s, _p(row), C.c_int64(n), _p(out))
        return out

    def sample_diffusion(self, gid, em_base, amp, p_survive):
        """(survive[amp], z_radial[amp], z_azimuthal[amp]) of the candidate electrons of one S2 instruction"""
        sv, z0, z1 = np.zeros(amp, dtype=np.uint8), np.zeros(amp), np.zeros(amp)
        lib().orc_sample_diffusion(self._s, C.c_uint32(gid), C.c_uint32(em_base), C.c_int64(amp), C.c_double(p_survive), _p(sv), _p(z0), _p(z1))
        return sv.astype(bool), z0, z1

    def sample_gas_gap(self, n, table, weight):
        """the 'garfield_gas_gap' luminescence term of the n photons of one instruction (mean subtracted, truncated)"""
        out = np.zeros(n, dtype=np.int64)
        lib().orc_sample_gas_gap(self._s, C.c_int64(n), C.c_int32(table), C.c_double(weight), _p(out))
        return out

    def sample_poisson(self, lam, n):
        out = np.zeros(n, dtype=np.int64)
        lib().orc_sample_poisson(self._s, C.c_double(lam), C.c_int64(n), _p(out))
        return out

    @property
    def n_pe(self):
        return lib().orc_n_pe(self._s)

    def get(self, name):
        n = C.c_int64(0)
        ptr = getattr(lib(), 'orc_' + name)(self._s, C.byref(n))
        dt = np.dtype(_GETTERS[name])
        if n.value == 0 or not ptr:
            return np.zeros(0, dtype=dt)
        buf = (C.c_char * (n.value * dt.itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dt).copy()

    def results(self):
        return {k: self.get(k) for k in _GETTERS}


def philox(ctr, key):
    c = _arr(ctr, np.uint32)
    k = _arr(key, np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    lib().orc_philox(_p(c), _p(k), _p(out))
    return out


def add_noise(data, mask, left, right, noise, ix_rand):
    """rawdata.py:398-437 after the ix_rand draw: in place on int64 data[n_ch, L]; returns nothing"""
    data_c = np.ascontiguousarray(data, dtype=np.int64)
    assert data_c is data or data.size == 0 or np.shares_memory(data_c, data)
    m, l, r = _arr(mask, np.uint8), _arr(left, np.int64), _arr(right, np.int64)
    if np.asarray(noise).dtype.kind == 'f':          # float noise: numba stores the truncated sum into the int64 row
        nz = _arr(noise, np.float64)
        lib().orc_add_noise_float(_p(data_c), C.c_int64(data.shape[0]), C.c_int64(data.shape[1]), _p(m), _p(l), _p(r), _p(nz),
                                  C.c_int64(nz.shape[0]), C.c_int64(nz.shape[1]), C.c_int64(int(ix_rand)))
        return
    nz = _arr(noise, np.int16)
    lib().orc_add_noise(_p(data_c), C.c_int64(data.shape[0]), C.c_int64(data.shape[1]), _p(m), _p(l), _p(r), _p(nz),
                        C.c_int64(nz.shape[0]), C.c_int64(nz.shape[1]), C.c_int64(int(ix_rand)))


def noise_high(mask, left, right, noise_len):
    """upper bound of np.random.randint(0, high) for the noise start index (rawdata.py:407-417); -1: no masked channel"""
    m, l, r = _arr(mask, np.uint8), _arr(left, np.int64), _arr(right, np.int64)
    f = lib().orc_noise_high
    f.restype = C.c_int64
    return int(f(C.c_int64(len(m)), _p(m), _p(l), _p(r), C.c_int64(int(noise_len))))


def make_oracle(config, ap_tables=None, resource=None):
    """An oracle session for a fax config: the scalars and host tables the HIP engine gets (wfsim_amd.config / tables / resource build
    them for both sides), the run-set switch and the delay-model tables.  Used by tests/, __graft_entry__.smoke() and bench.py's
    cpu_baseline leg -- never by the product path."""
    from wfsim_amd import tables as T
    from wfsim_amd.config import kernel_params, N_ROWS
    from wfsim_amd.delay_models import DelayModels
    from wfsim_amd.resource import Resource
    resource = resource or Resource(config)
    thr_truth, thr_zle = T.thresholds(config, N_ROWS)
    lum_x, lum_t = T.luminescence_table(config)
    tables = dict(templates=T.pmt_current_templates(config), spe=T.spe_scaling_table(resource.spe_charge, resource.spe_pdfs),
                  gains=np.asarray(config['gains'], dtype=np.float64), thr_truth=thr_truth, thr_zle=thr_zle,
                  lum_x=lum_x, lum_t=lum_t, noise=getattr(resource, 'noise_data', None))
    orc = Oracle(kernel_params(config), tables, ap_tables)
    orc.set_save_full_truth(config.get('save_full_truth', True))
    orc.set_delay_models(DelayModels(config, resource))
    return orc
