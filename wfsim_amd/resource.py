"""Minimal resource layer of the hot path: maps are host callables, tables are numpy arrays.

The reference resolves file names through straxen downloaders and builds ``straxen.InterpolatingMap`` objects
(/root/reference/wfsim/load_resource.py:49-380).  Downloaders and InterpolatingMap are third party and out of
scope; what the hot path needs from a resource is kept:

* ``DummyMap`` / ``make_map(["constant dummy", const, shape])`` -- load_resource.py:383-457
* any callable ``positions[n, d] -> values[n, ...]`` may be assigned to the map attributes
* ``photon_area_distribution`` (SPE CSV or arrays), ``noise_data``, ``uniform_to_pmt_ap`` tables
"""
import numpy as np


class DummyMap:
    """A map that is the same everywhere: called with n positions it returns an array of shape (n, *shape) filled with
    ``const`` (the behaviour of load_resource.py:438-457, which the bundled test config relies on)."""

    def __init__(self, const, shape=()):
        self.const, self.shape = const, shape

    def __call__(self, x, **kwargs):
        return np.full((len(x), *self.shape), self.const, dtype=np.float64)

    def reduce_last_dim(self):
        """the map summed over its last axis (kept with length 1): ``shape[-1]`` equal entries"""
        assert len(self.shape) >= 1, 'Need at least 1 dim to reduce further'
        *lead, last = self.shape
        return DummyMap(self.const * last, [*lead, 1])


def make_map(map_file, fmt=None, method='WeightedNearestNeighbors'):
    """``["constant dummy", const, shape]`` -> DummyMap; a callable is returned as is; a dict (the map data) or the
    path of a json / json.gz / pkl file of it -> ``itp_map.InterpolatingMap`` (load_resource.py:383-401; the reference
    resolves file NAMES through straxen's downloader, here the path is opened as given)."""
    if isinstance(map_file, (list, tuple)):
        assert map_file[0] == 'constant dummy', ('Alternative file input can only be '
                                                 '("constant dummy", constant: int, shape: list')
        return DummyMap(map_file[1], map_file[2])
    if callable(map_file):
        return map_file
    if isinstance(map_file, dict):
        from .itp_map import InterpolatingMap
        return InterpolatingMap(map_file, method=method)
    if isinstance(map_file, str):
        from .itp_map import InterpolatingMap, load_map_data
        return InterpolatingMap(load_map_data(map_file, fmt), method=method)
    raise TypeError("Can't handle map_file except a string or a list")


def rz_wrapper(rz_map):
    """load_resource.py:335-338: a map in (r, z) called with (z, xy)"""
    def wrapped(z, xy, **kwargs):
        r = np.sqrt(xy[:, 0] ** 2 + xy[:, 1] ** 2)
        return rz_map(np.array([r, z]).T, **kwargs)
    return wrapped


def make_patternmap(map_file, fmt=None, method='WeightedNearestNeighbors', pmt_mask=None):
    """Pattern maps with the PMT mask applied and the compressed / quantised storage undone (load_resource.py:403-435).
    Compression codecs: the ones python ships (bz2, zlib, lzma); strax's blosc / zstd / lz4 are not installed here."""
    if isinstance(map_file, (list, tuple)) or callable(map_file):
        return make_map(map_file)
    from copy import deepcopy
    from .itp_map import InterpolatingMap, load_map_data
    map_data = deepcopy(map_file) if isinstance(map_file, dict) else load_map_data(map_file, fmt)
    if 'compressed' in map_data:
        import bz2
        import lzma
        import zlib
        codecs = dict(bz2=bz2.decompress, zlib=zlib.decompress, lzma=lzma.decompress)
        compressor, dtype, shape = map_data['compressed']
        if compressor not in codecs:
            raise NotImplementedError(f'map compressor {compressor!r}: only {sorted(codecs)} are available offline')
        map_data['map'] = np.frombuffer(codecs[compressor](map_data['map']), dtype=dtype).reshape(*shape)
        del map_data['compressed']
    if 'quantized' in map_data:
        map_data['map'] = map_data['quantized'] * np.asarray(map_data['map']).astype(np.float32)
        del map_data['quantized']
    if pmt_mask is not None:
        m = np.array(map_data['map'])
        assert m.shape[-1] == pmt_mask.shape[0], 'Error! Pattern map and PMT gains must have same dimensions!'
        m[..., ~pmt_mask] = 0.0
        map_data['map'] = m
    return InterpolatingMap(map_data, method=method)


class Resource:
    """The attributes of the reference's ``Resource`` that the hot path reads."""

    def __init__(self, config):
        c = config
        if c.get('detector', 'XENONnT') not in ('XENON1T', 'XENONnT', 'XENONnT_neutron_veto'):
            raise ValueError(f"Unsupported detector {c['detector']}")          # load_resource.py:115
        pmt_mask = np.asarray(c['gains']) > 0
        if c.get('detector', 'XENONnT') == 'XENON1T':
            # load_resource.py:216-221: plain maps for 1T (its S2 map holds the top array only; S2.photon_channels pads the rest)
            self.s1_pattern_map = make_map(c['s1_pattern_map'])
            self.s2_pattern_map = make_map(c['s2_pattern_map'])
        else:
            self.s1_pattern_map = make_patternmap(c['s1_pattern_map'], pmt_mask=pmt_mask)
            self.s2_pattern_map = make_patternmap(c['s2_pattern_map'], pmt_mask=pmt_mask)
        # target mean area fraction top of the S2 pattern (load_resource.py:255-272): top and bottom arrays are rescaled
        # separately so that the total efficiency is preserved; a dummy map is left alone
        aft = c.get('s2_mean_area_fraction_top', -1)
        if aft is not None and aft >= 0.0 and hasattr(self.s2_pattern_map, 'data'):
            from .itp_map import InterpolatingMap
            data = dict(self.s2_pattern_map.data)
            m = np.array(data['map'], dtype=np.float64)
            n_top = c['n_top_pmts']
            top, tot = m[..., 0:n_top].sum(axis=-1), m.sum(axis=-1)
            orig_aft = np.mean((top / np.where(tot > 0, tot, 1))[tot > 0.0])
            m[..., 0:n_top] *= aft / orig_aft
            m[..., n_top:c['n_tpc_pmts']] *= (1 - aft) / (1 - orig_aft)
            data['map'] = m
            self.s2_pattern_map = InterpolatingMap(data, method=getattr(self.s2_pattern_map, 'method', 'WeightedNearestNeighbors'))
        # light-yield / S2 correction maps: the given ones, else derived from the pattern maps (load_resource.py:242-284)
        from .itp_map import InterpolatingMap

        def summed(pm, normalise):
            data = dict(pm.data)
            m = np.sum(np.array(data['map'], dtype=np.float64), axis=-1, keepdims=True, where=pmt_mask)
            if normalise:
                m = m / np.median(m[m > 0])
            data['map'] = m
            return InterpolatingMap(data, method=getattr(pm, 'method', 'WeightedNearestNeighbors'))
        if c.get('s1_lce_correction_map'):
            self.s1_lce_correction_map = make_map(c['s1_lce_correction_map'])
        elif hasattr(self.s1_pattern_map, 'data'):
            self.s1_lce_correction_map = summed(self.s1_pattern_map, False)
        else:
            self.s1_lce_correction_map = make_map(['constant dummy', 1, []])
        if c.get('s2_correction_map'):
            self.s2_correction_map = make_map(c['s2_correction_map'])
        elif hasattr(self.s2_pattern_map, 'data'):
            self.s2_correction_map = summed(self.s2_pattern_map, True)
        else:
            self.s2_correction_map = make_map(['constant dummy', 1, []])
        self.se_gain_map = make_map(c.get('se_gain_map', ['constant dummy', 1, []]))
        efd = c.get('enable_field_dependencies', {})
        if any(efd.values()):
            fmap = make_map(c.get('field_dependencies_map', ['constant dummy', 1, []]), method='RectBivariateSpline')
            self.drift_velocity_scaling = 1.0
            # scale the drift speed map so that the drift time from the cathode at r = 0 matches the configured velocity
            # (load_resource.py:325-333)
            if efd.get('norm_drift_velocity', False):
                norm_dvel = fmap(np.array([[0], [- c['tpc_length']]]).T, map_name='drift_speed_map')[0] * 1e-4
                self.drift_velocity_scaling = c['drift_velocity_liquid'] / norm_dvel

            self.field_dependencies_rz = fmap      # the maps in (r, z); the reference keeps only the (z, xy) wrappers below
            self.field_dependencies_map = rz_wrapper(fmap)
            if efd.get('diffusion_longitudinal_map', False):       # data-driven longitudinal diffusion (load_resource.py:340-347)
                self.diffusion_longitudinal_rz = make_map(c['diffusion_longitudinal_map'])
                self.diffusion_longitudinal_map = rz_wrapper(self.diffusion_longitudinal_rz)
        # field distortion models of S2.__call__ (load_resource.py:310-315)
        if c.get('field_distortion_model', 'none') == 'inverse_fdc':
            self.fdc_3d = make_map(c['fdc_3d'])
            if hasattr(self.fdc_3d, 'scale_coordinates'):
                self.fdc_3d.scale_coordinates([1., 1., - c['drift_velocity_liquid']])
        if c.get('field_distortion_model', 'none') == 'comsol':
            self.fd_comsol = make_map(c['field_distortion_comsol_map'], method='RectBivariateSpline')
        # photon propagation splines (load_resource.py:354-365) and the garfield luminescence table (load_resource.py:293-309)
        if c.get('s1_time_spline', False):
            self.s1_optical_propagation_spline = make_map(c['s1_time_spline'], method='RegularGridInterpolator')
        if c.get('s2_time_spline', False):
            self.s2_optical_propagation_spline = make_map(c['s2_time_spline'])
        if c.get('s2_luminescence_model', 'simple') == 'garfield':
            lum = c['s2_luminescence']
            if isinstance(lum, str):
                lum = np.load(lum, allow_pickle=False)
                lum = lum['arr_0'] if hasattr(lum, 'files') and 'arr_0' in lum.files else lum
            if getattr(lum, 'dtype', None) is not None and lum.dtype.names and 'll' in lum.dtype.names:
                # several liquid levels in one file: take the simulated one (load_resource.py:303-307)
                levels = np.unique(lum['ll'])
                level = min(levels, key=lambda x: abs(x - (c['gate_to_anode_distance'] - c['elr_gas_gap_length'])))
                lum = lum[lum['ll'] == level]
            self.s2_luminescence = dict(t=np.asarray(lum['t']), x=np.asarray(lum['x']))
        if 'garfield_gas_gap' in c.get('s2_luminescence_model', ''):
            # load_resource.py:284-291: excitation-time inverse CDFs per tabulated gas gap + the map (x, y) -> gas gap
            gg = c['s2_luminescence_gg']
            if isinstance(gg, str):
                gg = np.load(gg, allow_pickle=True)
                gg = gg['arr_0'] if hasattr(gg, 'files') and 'arr_0' in gg.files else gg
            self.s2_luminescence_gg = dict(gas_gap=np.asarray(gg['gas_gap'], dtype=np.float64).reshape(-1),
                                           timing_inv_cdf=np.ascontiguousarray(np.asarray(gg['timing_inv_cdf'], dtype=np.float64)))
            self.garfield_gas_gap_map = make_map(c['garfield_gas_gap_map'])
        if c.get('enable_gas_gap_warping', False):
            self.gas_gap_length = make_map(c['gas_gap_map'])
        # SPE area distributions: dict(charge, pdf[, n_channels]) or dict(charge, pdfs[n_ch, n_bins]) or a CSV path
        pad = c['photon_area_distribution']
        if isinstance(pad, str):
            import pandas as pd
            df = pd.read_csv(pad)
            cols = list(df.columns[1:])             # pulse.py:201: first column is the charge axis
            self.spe_charge = df[df.columns[0]].values.astype(np.float64)
            self.spe_pdfs = np.stack([df[k].values.astype(np.float64) for k in cols])
        else:
            self.spe_charge = np.asarray(pad['charge'], dtype=np.float64)
            if 'pdfs' in pad:
                self.spe_pdfs = np.asarray(pad['pdfs'], dtype=np.float64)
            else:
                self.spe_pdfs = np.asarray(pad['pdf'], dtype=np.float64)[None, :]   # one shared distribution
        if c.get('enable_noise', False):
            self.noise_data = np.ascontiguousarray(c['noise_data'])
        if c.get('enable_pmt_afterpulses', False):
            self.uniform_to_pmt_ap = c['uniform_to_pmt_ap']
        if c.get('enable_electron_afterpulses', False):
            # delay-time histogram of the photo-ionisation electrons (load_resource.py:233, a multihist.Hist1d in the
            # reference): any object with ``histogram`` and ``bin_edges``, or a (histogram, bin_edges) pair
            h = c['uniform_to_ele_ap']
            if isinstance(h, (tuple, list)):
                from .electron_afterpulse import DelayHistogram
                h = DelayHistogram(h[0], h[1])
            self.uniform_to_ele_ap = h
