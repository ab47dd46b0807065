"""ctypes binding of libwfsim_amd.so (include/wfsim_amd.h) -- the only way the Python host side reaches the GPU.

There is no CPU fallback: if the HIP library is missing, fails to load, or no MI355X is visible, constructing an
``Engine`` raises.  Nothing in this module (or anywhere under wfsim_amd/) imports the test oracle.
"""
import ctypes as C
import os

import numpy as np

from .config import kernel_params, N_ROWS
from .dtypes import raw_record_dtype
from . import tables as T

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('WFSIM_AMD_LIB') or os.path.join(HERE, 'libwfsim_amd.so')

_I32 = ['dt', 'samples_before', 'samples_after', 'store_before', 'store_after', 'tlen', 'trigger_window', 'baseline',
        'n_rows', 'n_tpc', 'n_top', 'he_first', 'he_factor', 'sum_channel', 'last_bottom', 'detector_nt', 'enable_noise',
        's1_simple', 's2_time_model', 'enable_pmt_ap', 'tile_gen', 'tile_gen_min', 'fma', 'row_resident']
_F64 = ['c2a', 'tts_mean', 'tts_sigma', 'p_dpe', 's1_decay_time', 's1_decay_spread', 'sf_gas', 't1_gas', 't3_gas',
        's2_time_spread', 'trap_time', 'gain_spread', 'pmt_ap_modifier', 'pmt_ap_t_modifier', 'rext', 'drift_velocity']


def pin_host(array):
    """page-locks a host array (hipHostRegister): device -> host copies into it then run at PCIe speed and asynchronously.
    Returns True on success; a failure is not an error (copies still work, staged by the driver)."""
    a = np.asarray(array)
    if a.nbytes == 0 or not a.flags['C_CONTIGUOUS']:
        return False
    return load_library().wfs_host_register(C.c_void_p(a.ctypes.data), C.c_int64(a.nbytes)) == 0


def unpin_host(array):
    load_library().wfs_host_unregister(C.c_void_p(np.asarray(array).ctypes.data))


_RECORD_BUFFERS = []        # [array, in use]: page-locked record buffers, recycled between chunkers
MAX_RECORD_BUFFERS = 4      # ceiling of the page-locked pool (1.2 GB each by default): beyond it buffers are pageable / the chunker copies


def acquire_record_buffer(length, dtype, pin=True, spare_only=False):
    """A record buffer for ChunkRawRecords (strax_interface.py:360-361: np.zeros(5000000, raw_record_dtype), 1.2 GB).
    Page-locking that much memory takes a few hundred ms, so pinned buffers are recycled between instances.  The pool never
    holds more than MAX_RECORD_BUFFERS page-locked buffers (a free one of another size is unpinned to make room); past the
    ceiling the caller gets an ordinary pageable array (``spare_only``: None) -- copies into it still work, staged by the driver."""
    dtype = np.dtype(dtype)
    for slot in _RECORD_BUFFERS:
        if not slot[1] and len(slot[0]) == length and slot[0].dtype == dtype:
            slot[1] = True
            return slot[0]
    if len(_RECORD_BUFFERS) >= MAX_RECORD_BUFFERS:
        for k, slot in enumerate(_RECORD_BUFFERS):
            if not slot[1]:
                unpin_host(slot[0])
                del _RECORD_BUFFERS[k]
                break
    if len(_RECORD_BUFFERS) >= MAX_RECORD_BUFFERS:
        return None if spare_only else np.zeros(length, dtype=dtype)
    buf = np.zeros(length, dtype=dtype)
    if pin and pin_host(buf):
        _RECORD_BUFFERS.append([buf, True])
    elif spare_only:
        return None
    return buf


def release_record_buffer(buf):
    for slot in _RECORD_BUFFERS:
        if slot[0] is buf:
            slot[1] = False


def is_pooled_record_buffer(buf):
    return any(slot[0] is buf for slot in _RECORD_BUFFERS)


def lease_record_buffer(buf, n):
    """The first n records of a pooled buffer as an array the caller may hand out: the buffer stays taken until that array
    and every view derived from it are gone, then it returns to the pool.  (The array is built on a memoryview, so numpy's
    base chain of every derived view ends at it and not at the buffer: its finaliser runs when the last view dies.)"""
    import weakref
    arr = np.frombuffer(memoryview(buf.view(np.uint8).reshape(-1)), dtype=buf.dtype, count=n)
    weakref.finalize(arr, release_record_buffer, buf)
    return arr


class WfsConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in _I32] + [(n, C.c_double) for n in _F64] + [('seed', C.c_uint64)]


class WfsCounts(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ['n_instructions', 'n_pulse_sets', 'n_emitters', 'n_photons', 'n_pe', 'n_tiles',
                                         'n_groups', 'n_rows', 'n_raw_samples', 'n_intervals', 'n_records']]


class WfsError(RuntimeError):
    pass


_lib = None
EXPORTS = ['wfs_create', 'wfs_destroy', 'wfs_last_error', 'wfs_device_count', 'wfs_set_tables', 'wfs_set_ap_element',
           'wfs_load_instructions', 'wfs_load_photons', 'wfs_load_optical', 'wfs_run', 'wfs_get_counts', 'wfs_copy_records',
           'wfs_copy_records_dev', 'wfs_records_dev_ptr', 'wfs_copy_groups', 'wfs_copy_intervals',
           'wfs_copy_interval_data', 'wfs_copy_pulses', 'wfs_copy_currents', 'wfs_copy_rows', 'wfs_copy_row_data',
           'wfs_copy_photons', 'wfs_copy_truth', 'wfs_copy_truth_per_pmt', 'wfs_copy_instruction_photon_offsets', 'wfs_gather_photon_times', 'wfs_copy_electron_stats', 'wfs_set_window_carry', 'wfs_copy_cluster_groups', 'wfs_set_noise_offsets', 'wfs_set_debug', 'wfs_set_stream', 'wfs_synchronize',
           'wfs_kernel_times', 'wfs_set_profiling', 'wfs_set_delay_models', 'wfs_set_s1_propagation', 'wfs_set_instruction_models',
           'wfs_set_pattern_map', 'wfs_eval_pattern_rows', 'wfs_copy_cdf_rows', 'wfs_set_record_order', 'wfs_copy_records_range',
           'wfs_copy_records_range_async', 'wfs_wait_records', 'wfs_host_register', 'wfs_host_unregister',
           'wfs_set_gas_gap_model', 'wfs_set_instruction_gas_gap', 'wfs_set_pattern_map_points', 'wfs_set_instruction_aft',
           'wfs_scalar_map_grid', 'wfs_scalar_map_points', 'wfs_scalar_map_spline', 'wfs_scalar_map_eval', 'wfs_set_noise_float', 'wfs_set_instruction_diffusion',
           'wfs_scalar_map_grid_array', 'wfs_scalar_map_points_array', 'wfs_scalar_map_linear', 'wfs_scalar_map_eval_array']


def load_library():
    """Loads libwfsim_amd.so; raises if it is missing (build it with ``python -c "import __graft_entry__ as g; g.build()"``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise WfsError(f'{LIB_PATH} not found: the HIP extension is not built; there is no CPU fallback')
        # torch ships its own HIP runtime: in a process that uses both (the multi-GPU gather keeps records in torch tensors),
        # torch.cuda must be initialised BEFORE this library pulls in the system's libamdhip64 -- afterwards torch finds no GPU
        import sys
        if 'torch' in sys.modules:
            try:
                sys.modules['torch'].cuda.init()
            except Exception:
                pass
        lib = C.CDLL(LIB_PATH)
        for name in EXPORTS:
            getattr(lib, name)          # AttributeError if a declared symbol is missing
        lib.wfs_last_error.restype = C.c_char_p
        lib.wfs_last_error.argtypes = [C.c_void_p]
        lib.wfs_records_dev_ptr.restype = C.c_void_p
        lib.wfs_records_dev_ptr.argtypes = [C.c_void_p]
        _lib = lib
    return _lib


def device_count():
    n = C.c_int(0)
    rc = load_library().wfs_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def _p(a):
    return C.c_void_p(a.ctypes.data) if a is not None else C.c_void_p(0)


def _arr(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def first_instruction_of_sets(run_set, n_sets):
    """index of the first instruction of every run set -- the number the library knows the set by (include/wfsim_amd.h,
    wfs_load_instructions) -- or None when a set number is unused.  (The sets need not be numbered in order of first appearance: the
    scheduler numbers the S1 calls of a cluster before its S2 calls, rawdata.py:102.)"""
    run_set = np.asarray(run_set)
    n = len(run_set)
    first = np.full(n_sets, n, dtype=np.int64)
    np.minimum.at(first, run_set, np.arange(n))
    return first if np.all(first < n) else None


class Engine:
    """One GPU, one stream, one fax configuration."""

    def __init__(self, config, resource, device=0, seed=None, keep_photons=False):
        """keep_photons: the photons of tile-generated S2 instructions (wfs_config.tile_gen) are also written to the photon
        array, for ``photons()`` -- by default they only exist in the registers of the pulse workgroup (tests: True)"""
        self.lib = load_library()
        self.keep_photons = bool(keep_photons)
        self.config = config
        params = kernel_params(config)
        if seed is not None:
            params['seed'] = int(seed)
        self.params = params
        cfg = WfsConfig()
        for n, _ in WfsConfig._fields_:
            setattr(cfg, n, params[n])
        self._h = C.c_void_p(0)
        self.device = int(device)
        self._pinned = []
        rc = self.lib.wfs_create(C.byref(cfg), C.c_int(device), C.byref(self._h))
        if rc != 0:
            raise WfsError(f'wfs_create failed with code {rc}: no usable MI355X / HIP runtime (device {device})')
        thr_truth, thr_zle = T.thresholds(config, N_ROWS)
        lum_x, lum_t = T.luminescence_table(config)
        self.tables = dict(
            templates=T.pmt_current_templates(config), spe=T.spe_scaling_table(resource.spe_charge, resource.spe_pdfs),
            gains=_arr(config['gains'], np.float64), thr_truth=_arr(thr_truth, np.float64), thr_zle=_arr(thr_zle, np.int64),
            lum_x=_arr(lum_x, np.float64), lum_t=_arr(lum_t, np.float64),
            noise=_arr(resource.noise_data, np.int16) if (params['enable_noise'] and hasattr(resource, 'noise_data')) else None)
        t = self.tables
        assert t['templates'].shape == (params['dt'], params['tlen'])
        nl, nc = t['noise'].shape if t['noise'] is not None else (0, 0)
        self._check(self.lib.wfs_set_tables(
            self._h, _p(t['templates']), _p(t['spe']), C.c_int32(t['spe'].shape[0]), _p(t['gains']), _p(t['thr_truth']),
            _p(t['thr_zle']), _p(t['lum_x']), _p(t['lum_t']), C.c_int32(len(t['lum_x'])), _p(t['noise']), C.c_int32(nl), C.c_int32(nc)))
        if params['enable_pmt_ap'] and hasattr(resource, 'uniform_to_pmt_ap'):
            for e, (name, d) in enumerate(resource.uniform_to_pmt_ap.items()):
                dc, ac = _arr(d['delaytime_cdf'], np.float64), _arr(d['amplitude_cdf'], np.float64)
                self._check(self.lib.wfs_set_ap_element(
                    self._h, C.c_int32(e), C.c_int32(dc.shape[1]), C.c_int32(ac.shape[-1]), C.c_int32(ac.ndim == 2),
                    C.c_int32('Uniform' in name), C.c_double(d['delaytime_bin_size']), C.c_double(d['amplitude_bin_size']), _p(dc), _p(ac)))

        if self.keep_photons or os.environ.get('WFS_CHECK_LAUNCHES', '0') not in ('', '0'):
            self.set_debug(False)          # (every launch checked from the first run on)
        # HE records exist only when the HE rows can differ from a flat baseline (wfs_engine.hip refresh_dev): a non-zero
        if t['noise'] is not None and np.asarray(resource.noise_data).dtype.kind == 'f' \
                and not np.array_equal(np.asarray(resource.noise_data), np.trunc(resource.noise_data)):
            # a float noise array with non-integral values: numba's `int64 row += float64 noise` (rawdata.py:436) stores the
            # truncated SUM, which the int16 table cannot express -- the device then adds in f64 and truncates as well
            nf = _arr(resource.noise_data, np.float64)
            self._check(self.lib.wfs_set_noise_float(self._h, _p(nf), C.c_int32(nl), C.c_int32(nc)))
        # int(high_energy_deamplification_factor) (rawdata.py:242) or noise columns for the HE channels
        he_noise = bool(params['enable_noise']) and nc > params['he_first']
        self.emits_he_records = bool(params['detector_nt'] and params['n_top'] > 0 and (params['he_factor'] != 0 or he_noise))
        # pattern maps on regular grids are evaluated on the device (one channel CDF row per instruction)
        self.device_maps = set()
        if config.get('device_pattern_maps', True):
            from .itp_map import InterpolatingMap
            for which, (kind, dims) in enumerate([('s1', 3), ('s2', 2)], start=1):
                pm = getattr(resource, kind + '_pattern_map', None)
                if not isinstance(pm, InterpolatingMap):
                    # a straxen.InterpolatingMap handed over from the reference's own Resource: same data dict
                    data = getattr(pm, 'data', None)
                    csys = data.get('coordinate_system') if isinstance(data, dict) else None
                    if not (csys and isinstance(csys[0], (list, tuple)) and isinstance(csys[0][0], str) and 'map' in data):
                        continue
                    pm = InterpolatingMap({k: data[k] for k in ('coordinate_system', 'map')}, method=getattr(pm, 'method', 'WeightedNearestNeighbors'))
                if not (pm.dimensions == dims and pm.method == 'WeightedNearestNeighbors' and pm.map_names == ['map']):
                    continue
                if pm.grid is None:           # a point list (irregular coordinate system): brute-force neighbour search on the device
                    vals = np.asarray(pm.data['map'])
                    pts = np.ascontiguousarray(pm.coordinate_system, dtype=np.float64)
                    if vals.ndim != 2 or len(vals) != len(pts) or vals.shape[1] > params['n_tpc'] or len(pts) < 2 * dims:
                        continue
                    if kind == 's2' and vals.shape[1] != params['n_tpc'] and (vals.shape[1] - 1) in np.asarray(config['channels_bottom']):
                        continue
                    v = np.ascontiguousarray(vals, dtype=np.float32)
                    self._check(self.lib.wfs_set_pattern_map_points(self._h, C.c_int32(which), C.c_int32(dims), C.c_int64(len(pts)), _p(pts), _p(v), C.c_int32(v.shape[1])))
                    self.device_maps.add(kind)
                    continue
                grid, vals = pm.regular_grid()
                if vals.ndim != dims + 1 or vals.shape[-1] > params['n_tpc']:
                    continue
                if kind == 's2' and vals.shape[-1] != params['n_tpc'] and (vals.shape[-1] - 1) in np.asarray(config['channels_bottom']):
                    continue          # s2.py:648: such a map is not padded; leave the odd case to the host
                v = np.ascontiguousarray(vals.reshape(-1, vals.shape[-1]), dtype=np.float32)
                nn = np.asarray([len(g) for g in grid], dtype=np.int32)
                lo, hi = np.asarray([g[0] for g in grid], dtype=np.float64), np.asarray([g[-1] for g in grid], dtype=np.float64)
                self._check(self.lib.wfs_set_pattern_map(self._h, C.c_int32(which), C.c_int32(dims), _p(nn), _p(lo), _p(hi), _p(v), C.c_int32(v.shape[1])))
                self.device_maps.add(kind)
        # s2_aft_sigma on device rows needs top = the first n_top channels and bottom = the rest (s2.py:660-665 scales exactly those)
        cb = np.asarray(config['channels_bottom'])
        self._aft_on_device = bool(len(cb)) and np.array_equal(cb, np.arange(params['n_top'], params['n_tpc']))
        # scalar maps of the resource on the device: physics.instruction_params evaluates through this view
        self._scalar_maps = []
        if config.get('device_scalar_maps', True):
            from .device_maps import DeviceResource
            self.resource = DeviceResource(resource, self)
        else:
            self.resource = resource
        # model variants of the photon delays (S1 custom / optical propagation, S2 garfield / optical propagation)
        from .delay_models import DelayModels
        self.models = DelayModels(config, resource)
        if self.models.active:
            base, off, pmf, vmin = self.models.table_arrays()
            self._check(self.lib.wfs_set_delay_models(self._h, C.c_int32(len(base)), _p(base), _p(off), _p(pmf), _p(vmin)))
            sp = self.models.s1_prop
            if sp is not None:
                self._check(self.lib.wfs_set_s1_propagation(self._h, C.c_int32(len(sp['z'])), C.c_int32(sp['nu']), C.c_double(sp['u0']),
                                                            C.c_double(sp['du']), _p(sp['top']), _p(sp['bottom'])))
            gg = self.models.gas_gap
            if gg is not None:
                self._check(self.lib.wfs_set_gas_gap_model(self._h, C.c_int32(gg['inv'].shape[0]), C.c_int32(gg['inv'].shape[1]), _p(gg['inv'])))

    # ------------------------------------------------------------------------------------------
    def _check(self, rc):
        if rc != 0:
            msg = self.lib.wfs_last_error(self._h)
            raise WfsError(f'libwfsim_amd error {rc}: {msg.decode() if msg else ""}')

    def close(self):
        if self._h:
            self.lib.wfs_wait_records(self._h)
            self.unpin_all()
            self.lib.wfs_destroy(self._h)
            self._h = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------------------------------
    def load_instructions(self, ins, gid, cluster, tmin, ip, run_set=None, em_base=None):
        """ins: instruction array sorted by the scheduler key; ip: dict from physics.instruction_params;
        run_set: pulse set of every instruction (scheduler.run_sets), None = one set per instruction."""
        n = len(ins)
        a = [_arr(ins['type'], np.int8), _arr(ins['time'], np.int64), _arr(ins['amp'], np.int32), _arr(gid, np.uint32),
             _arr(cluster, np.int32), _arr(tmin, np.int64), _arr(ip['p_hit'], np.float64), _arr(ip['drift_mean'], np.float64),
             _arr(ip['drift_spread'], np.float64), _arr(ip['sc_gain'], np.float64), _arr(ip['cdf_row'], np.int32),
             _arr(ip['cdf_table'], np.float64)]
        rs = _arr(run_set, np.int32) if run_set is not None else None
        self._n_loaded = n
        self._n_run_sets = (int(rs.max()) + 1 if len(rs) else 0) if rs is not None else n
        # Run sets go to the library numbered by their FIRST instruction (the numbers of the other members stay unused): a set of one
        # instruction then carries that instruction's index, which is what lets it take the tile-local generator (wfs_tilegen.h) next to
        # the shared Pulse calls of electron afterpulses.  _set_rows: the library's row of every set of the caller, for the outputs below.
        self._set_rows = None
        if rs is not None and len(rs):
            first = first_instruction_of_sets(rs, self._n_run_sets)
            if first is not None:
                self._set_rows = first
                rs = _arr(first[rs], np.int32)
        eb = _arr(em_base, np.uint32) if em_base is not None else None       # (a local: the converted copy must outlive the call)
        self._check(self.lib.wfs_load_instructions(self._h, C.c_int64(n), *[_p(x) for x in a], C.c_int32(a[-1].shape[0]),
                                                   _p(rs), C.c_int64((n if self._set_rows is not None else int(rs.max()) + 1) if rs is not None and len(rs) else 0), _p(eb)))
        if np.any(a[10] < 0):           # rows from the device pattern maps
            aft = ip.get('aft_factor')
            if aft is not None:
                aft = _arr(aft, np.float64)
                self._check(self.lib.wfs_set_instruction_aft(self._h, C.c_int64(n), _p(aft)))
            ds = ip.get('diff_sigma')
            if ds is not None:                  # diffusion_transverse_map: those rows are averaged over the electrons inside wfs_run
                sr, sa = _arr(ds[0], np.float64), _arr(ds[1], np.float64)
                self._check(self.lib.wfs_set_instruction_diffusion(self._h, C.c_int64(n), _p(sr), _p(sa), C.c_double(float(self.config['tpc_radius']))))
            pxy = ip.get('pattern_xy')          # S2: the observed position under a field distortion model
            xyz = [_arr(ins['x'] if pxy is None else pxy[:, 0], np.float32), _arr(ins['y'] if pxy is None else pxy[:, 1], np.float32), _arr(ins['z'], np.float32)]
            self._check(self.lib.wfs_eval_pattern_rows(self._h, C.c_int64(n), *[_p(q) for q in xyz]))
        self._n_cdf_rows = a[11].shape[0] + int(np.sum(a[10] < 0))
        if self.models.active:
            tab, tabb, zi, zf = self.models.instruction_tables(ins, gid)
            if self.models.per_batch:           # tables that depend on the batch (gas gap warping): upload them first
                base, off, pmf, vmin = self.models.table_arrays()
                self._check(self.lib.wfs_set_delay_models(self._h, C.c_int32(len(base)), _p(base), _p(off), _p(pmf), _p(vmin)))
            self._check(self.lib.wfs_set_instruction_models(self._h, C.c_int64(n), _p(tab), _p(tabb), _p(zi), _p(zf)))
            if self.models.gas_gap is not None:
                gi, gw = self.models.instruction_gas_gap(ins)
                self._check(self.lib.wfs_set_instruction_gas_gap(self._h, C.c_int64(n), _p(gi), _p(gw)))

    def register_scalar_map(self, m, name='map'):
        """one map of an itp_map.InterpolatingMap on the device: (map id, host result has a trailing axis) or None if it is not of
        a kind the device evaluates (see device_maps.py)"""
        mid = C.c_int32(-1)
        if m.method == 'RectBivariateSpline' and name in m.splines:
            spl = m.splines[name]
            tx, ty, c = (np.ascontiguousarray(q, dtype=np.float64) for q in spl.tck)
            kx, ky = spl.degrees
            self._check(self.lib.wfs_scalar_map_spline(self._h, C.c_int32(len(tx)), _p(tx), C.c_int32(len(ty)), _p(ty), C.c_int32(kx), C.c_int32(ky), _p(c), C.byref(mid)))
            return mid.value, False
        if m.method not in ('WeightedNearestNeighbors', 'RegularGridInterpolator') or not 1 <= m.dimensions <= 3:
            return None
        v = np.asarray(m.data[name], dtype=np.float64)
        n_points = len(m.coordinate_system)
        if v.size % n_points or v.size // n_points > 4096:
            return None
        nv = v.size // n_points
        # values per node on a trailing axis: flat layout [points, nv] or nested along the grid axes [..., nv]
        array_valued = (v.ndim == 2 and v.shape[0] == n_points) or v.ndim == m.dimensions + 1
        if nv > 1 and not array_valued:
            return None
        trailing = array_valued and nv == 1
        v = np.ascontiguousarray(v.reshape(n_points, nv))
        if m.method == 'RegularGridInterpolator' and m.grid is None:
            return None                 # (straxen falls back to nearest neighbours on a point list: the WeightedNearestNeighbors path below)
        if m.grid is not None:
            nn = np.asarray([len(g) for g in m.grid], dtype=np.int32)
            lo, hi = np.asarray([g[0] for g in m.grid], dtype=np.float64), np.asarray([g[-1] for g in m.grid], dtype=np.float64)
            if m.method == 'RegularGridInterpolator':
                self._check(self.lib.wfs_scalar_map_linear(self._h, C.c_int32(m.dimensions), _p(nn), _p(lo), _p(hi), _p(v), C.c_int32(nv), C.byref(mid)))
            elif nv == 1:
                self._check(self.lib.wfs_scalar_map_grid(self._h, C.c_int32(m.dimensions), _p(nn), _p(lo), _p(hi), _p(v), C.byref(mid)))
            else:
                self._check(self.lib.wfs_scalar_map_grid_array(self._h, C.c_int32(m.dimensions), _p(nn), _p(lo), _p(hi), _p(v), C.c_int32(nv), C.byref(mid)))
        else:
            if n_points < 2 * m.dimensions:
                return None
            pts = np.ascontiguousarray(m.coordinate_system, dtype=np.float64)
            if nv == 1:
                self._check(self.lib.wfs_scalar_map_points(self._h, C.c_int32(m.dimensions), C.c_int64(n_points), _p(pts), _p(v), C.byref(mid)))
            else:
                self._check(self.lib.wfs_scalar_map_points_array(self._h, C.c_int32(m.dimensions), C.c_int64(n_points), _p(pts), _p(v), C.c_int32(nv), C.byref(mid)))
        self._smap_nv = getattr(self, '_smap_nv', {})
        self._smap_nv[mid.value] = nv
        return mid.value, bool(trailing)

    def eval_scalar_map(self, map_id, positions):
        pos = np.ascontiguousarray(positions, dtype=np.float64)
        pos = pos.reshape(len(pos), -1)
        nv = getattr(self, '_smap_nv', {}).get(map_id, 1)
        if nv == 1:
            out = np.empty(len(pos), dtype=np.float64)
            self._check(self.lib.wfs_scalar_map_eval(self._h, C.c_int32(map_id), C.c_int64(len(pos)), _p(pos), _p(out)))
            return out
        out = np.empty((len(pos), nv), dtype=np.float64)          # array-valued map: one row of values per position
        self._check(self.lib.wfs_scalar_map_eval_array(self._h, C.c_int32(map_id), C.c_int64(len(pos)), _p(pos), _p(out), C.c_int32(nv)))
        return out

    def cdf_rows(self):
        """(cdf_row, cdf_table) of the loaded batch as the generator uses them, device-evaluated rows included"""
        n, rows = self._n_loaded, self._n_cdf_rows
        cdf_row, table = np.zeros(n, dtype=np.int32), np.zeros((rows, self.params['n_tpc']), dtype=np.float64)
        self._check(self.lib.wfs_copy_cdf_rows(self._h, _p(cdf_row), _p(table), C.c_int64(rows)))
        return cdf_row, table

    def load_optical(self, ins, gid, cluster, tmin, channels, timings, time_cutoff):
        """ins: optical instructions (with _first/_last) sorted by time; channels/timings: the flat photon arrays"""
        a = [_arr(ins['time'], np.int64), _arr(gid, np.uint32), _arr(cluster, np.int32), _arr(tmin, np.int64),
             _arr(ins['_first'], np.int32), _arr(ins['_last'], np.int32), _arr(channels, np.int32), _arr(timings, np.int64)]
        self._check(self.lib.wfs_load_optical(self._h, C.c_int64(len(ins)), *[_p(x) for x in a], C.c_int64(len(a[-1])),
                                              C.c_int64(int(time_cutoff))))

    def load_photons(self, set_cluster, set_tmin, set_off, t, ch, gain, dpe=None):
        a = [_arr(set_cluster, np.int32), _arr(set_tmin, np.int64), _arr(set_off, np.int64), _arr(t, np.int64),
             _arr(ch, np.int16), _arr(gain, np.float64), _arr(dpe, np.uint8) if dpe is not None else None]
        self._check(self.lib.wfs_load_photons(self._h, C.c_int64(len(a[0])), *[_p(x) for x in a]))

    def run(self):
        self._check(self.lib.wfs_run(self._h))
        c = WfsCounts()
        self._check(self.lib.wfs_get_counts(self._h, C.byref(c)))
        self.counts = {n: getattr(c, n) for n, _ in WfsCounts._fields_}
        self._lib_sets = self.counts['n_pulse_sets']            # the library's pulse-set rows (run sets numbered by their first instruction leave gaps)
        rows = self._caller_sets(self._lib_sets)
        if rows is not None:
            self.counts['n_pulse_sets'] = len(rows)
        return self.counts

    def set_debug(self, on=True, force_dense=False, generate_only=False, check_launches=None):
        """check_launches (default: the environment variable WFS_CHECK_LAUNCHES): every kernel launch of wfs_run is checked and
        synchronised on the spot, so that a failing kernel is reported under its own name (slow; debugging only)"""
        if check_launches is None:
            check_launches = os.environ.get('WFS_CHECK_LAUNCHES', '0') not in ('', '0')
        self._check(self.lib.wfs_set_debug(self._h, C.c_int32(int(bool(on)) | (2 if force_dense else 0) | (4 if generate_only else 0)
                                                              | (8 if check_launches else 0) | (16 if self.keep_photons else 0))))

    def generate(self):
        """photon generation only (no pulses / records): the pre-pass of the electron afterpulses"""
        self.set_debug(False, generate_only=True)
        try:
            self._check(self.lib.wfs_run(self._h))
        finally:
            self.set_debug(False)

    def instruction_photon_offsets(self):
        """first generated photon of every instruction of the batch (+ the total): generation order"""
        n = getattr(self, '_n_loaded', 0)
        out = np.zeros(n + 1, dtype=np.int64)
        self._check(self.lib.wfs_copy_instruction_photon_offsets(self._h, _p(out), C.c_int64(n + 1)))
        return out

    def gather_photon_times(self, index):
        """arrival times [ns] of photons given by their index in generation order"""
        i = _arr(index, np.int64)
        out = np.zeros(len(i), dtype=np.int64)
        if len(i):
            self._check(self.lib.wfs_gather_photon_times(self._h, C.c_int64(len(i)), _p(i), _p(out)))
        return out

    def set_profiling(self, on=True):
        self._check(self.lib.wfs_set_profiling(self._h, C.c_int32(int(on))))

    def set_stream(self, stream_ptr):
        self._check(self.lib.wfs_set_stream(self._h, C.c_void_p(stream_ptr)))

    # ---- results ---------------------------------------------------------------------------------
    def records(self):
        n = self.counts['n_records']
        out = np.zeros(n, dtype=raw_record_dtype())
        self._check(self.lib.wfs_copy_records(self._h, _p(out), C.c_int64(n)))
        return out

    def records_into(self, out, count=None):
        """device -> host copy of the first ``count`` packed records (default: all) straight into ``out`` (a contiguous
        raw_record array, e.g. a slice of the chunker's record buffer)"""
        n = self.counts['n_records'] if count is None else int(count)
        assert out.flags['C_CONTIGUOUS'] and out.dtype.itemsize == np.dtype(raw_record_dtype()).itemsize and len(out) >= n
        self._check(self.lib.wfs_copy_records_range(self._h, C.c_void_p(out.ctypes.data), C.c_int64(0), C.c_int64(n)))
        return out[:n]

    def records_into_async(self, out, count=None):
        """the same on the copy stream, without waiting: ``out`` must stay alive (and should be pinned, ``pin``) until
        ``wait_records`` returns; the copy overlaps the next ``run``"""
        n = self.counts['n_records'] if count is None else int(count)
        assert out.flags['C_CONTIGUOUS'] and out.dtype.itemsize == np.dtype(raw_record_dtype()).itemsize and len(out) >= n
        self._check(self.lib.wfs_copy_records_range_async(self._h, C.c_void_p(out.ctypes.data), C.c_int64(0), C.c_int64(n)))
        return out[:n]

    def wait_records(self):
        self._check(self.lib.wfs_wait_records(self._h))

    def pin(self, array):
        """page-locks a host array for this engine's lifetime (see ``pin_host``)"""
        if not pin_host(array):
            return False
        self._pinned.append(np.asarray(array))
        return True

    def unpin_all(self):
        for a in self._pinned:
            unpin_host(a)
        self._pinned = []

    def set_record_order(self, by_time):
        self._check(self.lib.wfs_set_record_order(self._h, C.c_int32(int(bool(by_time)))))

    def records_dev_ptr(self):
        return self.lib.wfs_records_dev_ptr(self._h)

    def copy_records_to_device(self, dev_ptr, capacity):
        self._check(self.lib.wfs_copy_records_dev(self._h, C.c_void_p(dev_ptr), C.c_int64(capacity)))

    def groups(self):
        g = self.counts['n_groups']
        left, right, first, ix = (np.zeros(g, dtype=np.int64) for _ in range(4))
        self._check(self.lib.wfs_copy_groups(self._h, _p(left), _p(right), _p(first), _p(ix)))
        return dict(left=left, right=right, first_record=first, ix_rand=ix)

    def intervals(self):
        n = self.counts['n_intervals']
        group, ch = np.zeros(n, np.int32), np.zeros(n, np.int32)
        left, right, off = np.zeros(n, np.int64), np.zeros(n, np.int64), np.zeros(n + 1, np.int64)
        self._check(self.lib.wfs_copy_intervals(self._h, _p(group), _p(ch), _p(left), _p(right), _p(off), C.c_int64(n + 1)))
        data = np.zeros(int(off[n]), dtype=np.int16)
        self._check(self.lib.wfs_copy_interval_data(self._h, _p(data), C.c_int64(len(data))))
        return dict(group=group, channel=ch, left=left, right=right, data_off=off, data=data)

    def pulses(self, currents=False):
        n = self.counts['n_tiles']
        s, ch = np.zeros(n, np.int32), np.zeros(n, np.int32)
        left, right, nph, off = (np.zeros(n, np.int64) for _ in range(4))
        self._check(self.lib.wfs_copy_pulses(self._h, _p(s), _p(ch), _p(left), _p(right), _p(nph), _p(off), C.c_int64(n)))
        rows = self._caller_sets(self._lib_sets)
        if rows is not None:                        # the library's set numbers -> the caller's
            inv = np.full(self._lib_sets, -1, dtype=np.int32)
            inv[rows] = np.arange(len(rows), dtype=np.int32)
            s = inv[s]
        out = dict(set=s, channel=ch, left=left, right=right, n_photons=nph, cur_off=off)
        if currents:
            total = int((right - left + 1).sum())
            cur = np.zeros(total, dtype=np.float64)
            self._check(self.lib.wfs_copy_currents(self._h, _p(cur), C.c_int64(total)))
            out['current'] = cur
        return out

    def rows(self):
        n = self.counts['n_rows']
        g, ch = np.zeros(n, np.int32), np.zeros(n, np.int32)
        left, right, off = (np.zeros(n, np.int64) for _ in range(3))
        self._check(self.lib.wfs_copy_rows(self._h, _p(g), _p(ch), _p(left), _p(right), _p(off), C.c_int64(n)))
        total = int((right - left + 1).sum())
        data = np.zeros(total, dtype=np.int32)
        self._check(self.lib.wfs_copy_row_data(self._h, _p(data), C.c_int64(total)))
        return dict(group=g, channel=ch, left=left, right=right, data_off=off, data=data)

    def _caller_sets(self, s):
        """rows of the library's pulse sets that belong to the caller's run sets (None: all of them, in order): primaries, then -- when the
        library made PMT-afterpulse sets, a second block of the same size -- theirs"""
        rows = getattr(self, '_set_rows', None)
        if rows is None or not self.counts['n_instructions']:
            return None
        n = self.counts['n_instructions']
        return rows if s == n else np.concatenate([rows, n + rows])

    def photons(self):
        n, s = self.counts['n_photons'], self._lib_sets
        off = np.zeros(s + 1, np.int64)
        t, ch, gain, dpe = np.zeros(n, np.int64), np.zeros(n, np.int16), np.zeros(n, np.float64), np.zeros(n, np.uint8)
        self._check(self.lib.wfs_copy_photons(self._h, _p(off), _p(t), _p(ch), _p(gain), _p(dpe), C.c_int64(n)))
        rows = self._caller_sets(s)
        if rows is not None:                        # the caller's sets, in the caller's order (the unused set numbers hold no photons)
            lens = off[rows + 1] - off[rows]
            if np.all(np.diff(rows) > 0):
                off = np.append(off[rows], off[-1])
            else:
                new_off = np.concatenate([[0], np.cumsum(lens)])
                idx = np.repeat(off[rows] - new_off[:-1], lens) + np.arange(int(new_off[-1]))
                t, ch, gain, dpe, off = t[idx], ch[idx], gain[idx], dpe[idx], new_off
        return dict(set_off=off, t=t, ch=ch, gain=gain, dpe=dpe)

    def truth(self):
        s = self._lib_sets
        acc, ts = np.zeros((s, 12)), np.zeros((s, 5))
        self._check(self.lib.wfs_copy_truth(self._h, _p(acc), _p(ts), C.c_int64(s)))
        rows = self._caller_sets(s)
        return (acc, ts) if rows is None else (acc[rows], ts[rows])

    def truth_per_pmt(self):
        """[pulse set][channel][n_photon, n_pe, n_photon_trigger, n_pe_trigger, raw_area, raw_area_trigger] (pulse.py:259-271)"""
        s = self._lib_sets
        acc = np.zeros((s, int(self.params['n_tpc']), 6))
        if s:
            self._check(self.lib.wfs_copy_truth_per_pmt(self._h, _p(acc), C.c_int64(s)))
        rows = self._caller_sets(s)
        return acc if rows is None else acc[rows]

    def set_noise_offsets(self, ix_rand):
        a = _arr(ix_rand, np.int64)
        self._check(self.lib.wfs_set_noise_offsets(self._h, _p(a), C.c_int64(len(a))))

    def set_window_carry(self, has_pulse, last_pulse_end_time):
        self._check(self.lib.wfs_set_window_carry(self._h, C.c_int32(int(has_pulse)), C.c_int64(int(last_pulse_end_time))))

    def cluster_groups(self, n_clusters):
        g = np.zeros(n_clusters, dtype=np.int32)
        if n_clusters:
            self._check(self.lib.wfs_copy_cluster_groups(self._h, _p(g), C.c_int64(n_clusters)))
        return g

    def electron_stats(self):
        """per run set (= per instruction unless run sets were given): n, mean, min, max, std of the electron times"""
        n = getattr(self, '_n_run_sets', None)
        rows = getattr(self, '_set_rows', None)
        if n is None or self.counts['n_instructions'] == 0:
            n, rows = self.counts['n_instructions'], None
        if rows is not None:
            n = self.counts['n_instructions']
        es = np.zeros((n, 5))
        if n:
            self._check(self.lib.wfs_copy_electron_stats(self._h, _p(es), C.c_int64(n)))
        return es if rows is None else es[rows]

    def kernel_times(self):
        names = C.create_string_buffer(4096)
        ms = (C.c_float * 64)()
        nl = (C.c_int32 * 64)()
        nk = C.c_int32(0)
        self._check(self.lib.wfs_kernel_times(self._h, names, C.c_int64(4096), ms, nl, C.byref(nk)))
        parts = names.raw.split(b'\0')[:nk.value]
        return {parts[k].decode(): (float(ms[k]), int(nl[k])) for k in range(nk.value)}
