"""Scalar maps of the resource evaluated on the device (wfs_scalar_map_*): the light-yield, S2-correction, SE-gain,
longitudinal-diffusion, field-distortion and field-dependence maps the reference evaluates per instruction on the host
(/root/reference/wfsim/core/s1.py:125, s2.py:41, 66, 150, 170, 193-196, 229-234, 248).

``DeviceResource(resource, engine)`` is a view of a Resource whose eligible maps are replaced by ``DeviceMap`` callables
with the calling convention of ``straxen.InterpolatingMap`` (``m(positions, map_name='map')``); everything else is handed
through.  Eligible: InterpolatingMap objects (this package's, or a straxen one carrying the same ``data`` dict) with

* method WeightedNearestNeighbors, scalar or array valued (up to 4096 values per node), on a regular grid or a point list;
* method RegularGridInterpolator on a regular grid (load_resource.py:357 builds the S1 time spline with it): scipy's multilinear
  interpolation, the edge cell continued outside the grid; scalar or array valued;
* method RectBivariateSpline on a 2-D regular grid (load_resource.py:316, 326) -- the knots and coefficients of scipy's own
  spline object are uploaded, so the device evaluates the very spline the host would.

A map that is not eligible (callables, DummyMap constants, other methods) stays a host callable: the maps are inputs at the
drop-in boundary, not part of the generator.  Tolerance against the host evaluation: rtol 1e-6 (tests/test_gpu_scalar_maps.py).
"""
import numpy as np

from .itp_map import InterpolatingMap

MAP_ATTRIBUTES = ('s1_lce_correction_map', 's2_correction_map', 'se_gain_map', 'fdc_3d', 'fd_comsol',
                  'field_dependencies_rz', 'diffusion_longitudinal_rz')


def as_interpolating_map(m):
    """this package's InterpolatingMap for ``m`` (itself, or rebuilt from a straxen map's data dict), else None"""
    if isinstance(m, InterpolatingMap):
        return m
    data = getattr(m, 'data', None)
    csys = data.get('coordinate_system') if isinstance(data, dict) else None
    if csys is None or not len(csys):
        return None
    try:
        return InterpolatingMap(data, method=getattr(m, 'method', 'WeightedNearestNeighbors'))
    except Exception:
        return None


class DeviceMap:
    def __init__(self, engine, host_map):
        self.host = host_map
        self.engine = engine
        self.ids = {}                       # map name -> (device map id, trailing axis of the host result or None)
        for name in host_map.map_names:
            reg = engine.register_scalar_map(host_map, name)
            if reg is not None:
                self.ids[name] = reg

    def __call__(self, positions, map_name='map'):
        reg = self.ids.get(map_name)
        if reg is None:
            return self.host(positions, map_name=map_name)
        out = self.engine.eval_scalar_map(reg[0], positions)
        return out[:, None] if (reg[1] and out.ndim == 1) else out

    def __getattr__(self, k):               # data, method, map_names, scale_coordinates ...
        if k in ('host', 'engine', 'ids'):      # (not yet set: copy / pickle probing an empty instance)
            raise AttributeError(k)
        return getattr(self.host, k)


class DeviceResource:
    def __init__(self, resource, engine):
        from .resource import rz_wrapper
        self._resource = resource
        self.on_device = []
        for attr in MAP_ATTRIBUTES:
            m = as_interpolating_map(getattr(resource, attr, None))
            if m is None:
                continue
            dm = DeviceMap(engine, m)
            if not dm.ids:
                continue
            self.on_device.append(attr)
            setattr(self, attr, dm)
            if attr == 'field_dependencies_rz':
                self.field_dependencies_map = rz_wrapper(dm)
            if attr == 'diffusion_longitudinal_rz':
                self.diffusion_longitudinal_map = rz_wrapper(dm)

    def __getattr__(self, k):
        if k == '_resource':
            raise AttributeError(k)
        return getattr(self._resource, k)
