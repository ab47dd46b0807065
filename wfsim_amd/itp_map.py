"""Interpolating maps of the resource files (pattern / LCE / gain maps, propagation splines) on the host.

The reference builds ``straxen.InterpolatingMap`` objects (/root/reference/wfsim/load_resource.py:383-433; straxen >= 2.2.0,
requirements.txt:3).  straxen is third party and not under /root/reference, so this is a restatement of its published
behaviour for the map formats the reference loads -- parity with straxen itself is unpinned:

* ``data`` is a dict (or json / gzipped json of it) with ``coordinate_system`` and one or more maps; every key that is not
  metadata is a map (``map_name``), default ``'map'``.
* ``coordinate_system`` is either a list of points or a regular grid ``[[name, [min, max, n]], ...]``.
* ``method='WeightedNearestNeighbors'`` (the default of load_resource.make_map): inverse-distance weighted average of the
  ``2 * dimensions`` nearest points found with a KD-tree, distances clipped at 1e-6 -- this also extrapolates.
* ``method='RegularGridInterpolator'``: scipy's multilinear interpolation on the regular grid, extrapolating
  (``bounds_error=False, fill_value=None``); ``'RectBivariateSpline'``: scipy's spline, 2-D regular grids only.
* maps may be array valued (one more trailing dimension: the PMT patterns).
"""
import gzip
import json
import pickle

import numpy as np

METADATA_FIELDS = ('timestamp', 'description', 'coordinate_system', 'name', 'irregular', 'compressed', 'quantized')


class InterpolateAndExtrapolate:
    """Inverse-distance weighted average over the nearest points (KD-tree), scalar or array valued"""

    def __init__(self, points, values, neighbours_to_use=None, array_valued=False):
        from scipy.spatial import cKDTree
        self.kdtree = cKDTree(points)
        self.values = values
        self.neighbours_to_use = points.shape[1] * 2 if neighbours_to_use is None else neighbours_to_use
        self.array_valued = array_valued
        if array_valued:
            self.n_dim = values.shape[-1]

    def __call__(self, points):
        points = np.asarray(points)
        distances, indices = self.kdtree.query(points, self.neighbours_to_use)
        if self.neighbours_to_use == 1:
            distances, indices = distances[:, None], indices[:, None]
        result = np.full((len(points), self.n_dim) if self.array_valued else len(points), np.nan)
        valid = (distances < np.inf).max(axis=-1)            # a NaN coordinate gives infinite distances
        values = self.values[indices[valid]]
        weights = 1 / np.clip(distances[valid], 1e-6, np.inf)
        if self.array_valued:
            weights = np.repeat(weights, self.n_dim).reshape(values.shape)
            result[valid] = np.average(values, weights=weights, axis=-2)
        else:
            result[valid] = np.average(values, weights=weights, axis=-1)
        return result


class InterpolatingMap:
    def __init__(self, data, method='WeightedNearestNeighbors', **kwargs):
        if isinstance(data, bytes):
            data = gzip.decompress(data).decode()
        if isinstance(data, str):
            data = json.loads(data)
        assert isinstance(data, dict), f'Expected map data to be a dict, got {type(data)}'
        self.data = data
        self.method = method
        csys = data['coordinate_system']
        self.grid = None
        if not len(csys):
            self.dimensions = 0
        elif isinstance(csys[0], (list, tuple)) and isinstance(csys[0][0], str):
            self.dimensions = len(csys)
            self.grid = [np.linspace(left, right, int(points)) for _, (left, right, points) in csys]
            mesh = np.array(np.meshgrid(*self.grid, indexing='ij'))
            csys = np.transpose(mesh, np.roll(np.arange(self.dimensions + 1), -1)).reshape(-1, self.dimensions)
        else:
            csys = np.array(csys)
            self.dimensions = len(csys[0])
        self.coordinate_system = csys
        self.interpolators = {}
        self.splines = {}                    # map name -> scipy RectBivariateSpline (method 'RectBivariateSpline')
        self.map_names = sorted(k for k in data.keys() if k not in METADATA_FIELDS)
        for name in self.map_names:
            m = np.array(data[name])
            if self.dimensions == 0:
                self.interpolators[name] = (lambda positions, m=m: m * np.ones_like(positions))
                continue
            shape = tuple(len(g) for g in self.grid) if self.grid is not None else None
            if m.shape[0] == len(csys):                      # one entry per point: flat layout
                array_valued = m.ndim == 2
            else:                                            # nested along the grid axes
                array_valued = m.ndim == self.dimensions + 1
            if method == 'RegularGridInterpolator' and self.grid is not None:
                from scipy.interpolate import RegularGridInterpolator
                vals = m.reshape(shape + (m.shape[-1],)) if array_valued else m.reshape(shape)
                self.interpolators[name] = RegularGridInterpolator(tuple(self.grid), vals, bounds_error=False, fill_value=None)
            elif method == 'RectBivariateSpline' and self.grid is not None:
                from scipy.interpolate import RectBivariateSpline
                assert self.dimensions == 2 and not array_valued, 'RectBivariateSpline: scalar maps on 2-D grids'
                spl = RectBivariateSpline(self.grid[0], self.grid[1], m.reshape(shape), s=0)
                self.splines[name] = spl
                self.interpolators[name] = (lambda positions, spl=spl: spl.ev(np.asarray(positions)[:, 0], np.asarray(positions)[:, 1]))
            elif method in ('WeightedNearestNeighbors', 'RegularGridInterpolator', 'RectBivariateSpline'):
                vals = m.reshape((len(csys), m.shape[-1])) if array_valued else m.reshape(-1)
                self.interpolators[name] = InterpolateAndExtrapolate(csys, vals, array_valued=array_valued, **kwargs)
            else:
                raise ValueError(f'Interpolation method {method} is not supported')

    def __call__(self, *args, map_name='map'):
        return self.interpolators[map_name](*args)

    def scale_coordinates(self, scaling_factor, map_name='map'):
        """multiply every coordinate axis by a factor and rebuild the interpolators (load_resource.py:311)"""
        f = np.asarray(scaling_factor, dtype=np.float64)
        data = dict(self.data)
        if self.grid is not None:
            data['coordinate_system'] = [[n, [lo * f[i], hi * f[i], k]] for i, (n, (lo, hi, k)) in enumerate(self.data['coordinate_system'])]
        else:
            data['coordinate_system'] = (np.asarray(self.data['coordinate_system']) * f).tolist()
        self.__init__(data, method=self.method)

    def regular_grid(self, map_name='map'):
        """(grid axes, node values) of a regular-grid map: what the device-side evaluation is given"""
        assert self.grid is not None, 'not a regular grid'
        m = np.array(self.data[map_name], dtype=np.float64)
        shape = tuple(len(g) for g in self.grid)
        return self.grid, (m.reshape(shape) if m.size == int(np.prod(shape)) else m.reshape(shape + (-1,)))


def load_map_data(path, fmt=None):
    """json, json.gz, pkl, pkl.gz files of the map dict (straxen.get_resource's formats for maps)"""
    if fmt is None:
        fmt = 'json.gz' if path.endswith('json.gz') else 'pkl.gz' if path.endswith('pkl.gz') else path.rsplit('.', 1)[-1]
    if fmt == 'json':
        with open(path) as f:
            return json.load(f)
    if fmt == 'json.gz':
        with gzip.open(path, 'rt') as f:
            return json.load(f)
    if fmt == 'pkl':
        with open(path, 'rb') as f:
            return pickle.load(f)
    if fmt == 'pkl.gz':
        with gzip.open(path, 'rb') as f:
            return pickle.load(f)
    raise ValueError(f'unknown map file format {fmt!r} ({path})')
