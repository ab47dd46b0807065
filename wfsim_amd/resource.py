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
    """Constant map; the output's first dimension matches the input, the rest is ``shape``.

    Same behaviour as /root/reference/wfsim/load_resource.py:438-457.
    """

    def __init__(self, const, shape=()):
        self.const = const
        self.shape = shape

    def __call__(self, x, **kwargs):
        shape = [len(x)] + list(self.shape)
        return np.ones(shape) * self.const

    def reduce_last_dim(self):
        assert len(self.shape) >= 1, 'Need at least 1 dim to reduce further'
        const = self.const * self.shape[-1]
        shape = list(self.shape)
        shape[-1] = 1
        return DummyMap(const, shape)


def make_map(map_file, fmt=None, method='WeightedNearestNeighbors'):
    """``["constant dummy", const, shape]`` -> DummyMap; a callable is returned as is.

    File-backed interpolating maps need straxen and are not available here (load_resource.py:393-399).
    """
    if isinstance(map_file, (list, tuple)):
        assert map_file[0] == 'constant dummy', ('Alternative file input can only be '
                                                 '("constant dummy", constant: int, shape: list')
        return DummyMap(map_file[1], map_file[2])
    if callable(map_file):
        return map_file
    if isinstance(map_file, str):
        raise NotImplementedError(
            f'map file {map_file!r}: file-backed InterpolatingMaps need straxen; pass a callable or a dummy map')
    raise TypeError("Can't handle map_file except a string or a list")


class Resource:
    """The attributes of the reference's ``Resource`` that the hot path reads."""

    def __init__(self, config):
        c = config
        self.s1_pattern_map = make_map(c['s1_pattern_map'])
        self.s2_pattern_map = make_map(c['s2_pattern_map'])
        self.s1_lce_correction_map = make_map(c.get('s1_lce_correction_map', ['constant dummy', 1, []]))
        self.s2_correction_map = make_map(c.get('s2_correction_map', ['constant dummy', 1, []]))
        self.se_gain_map = make_map(c.get('se_gain_map', ['constant dummy', 1, []]))
        efd = c.get('enable_field_dependencies', {})
        if any(efd.values()):
            fmap = make_map(c.get('field_dependencies_map', ['constant dummy', 1, []]))
            self.drift_velocity_scaling = 1.0

            def rz_map(z, xy, **kwargs):           # load_resource.py:335-338
                r = np.sqrt(xy[:, 0] ** 2 + xy[:, 1] ** 2)
                return fmap(np.array([r, z]).T, **kwargs)
            self.field_dependencies_map = rz_map
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
