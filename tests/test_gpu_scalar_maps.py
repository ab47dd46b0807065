"""Scalar maps, point-list pattern maps and the s2_aft_sigma rescaling on the device (wfs_scalar_map_*,
wfs_set_pattern_map_points, wfs_set_instruction_aft) against the host evaluation (itp_map.InterpolatingMap -- the restatement
of straxen's WeightedNearestNeighbors -- and scipy's RectBivariateSpline), rtol 1e-6; records end to end against the oracle."""
import numpy as np
import pytest

from tests.helpers import make_engine, make_oracle
from tests.test_gpu_pattern_maps import map_config, instructions
from wfsim_amd.config import xenonnt_test_config
from wfsim_amd.itp_map import InterpolatingMap
from wfsim_amd.physics import instruction_params, s2_aft_factors
from wfsim_amd.resource import Resource
from wfsim_amd.scheduler import schedule

pytestmark = pytest.mark.gpu


def scalar_config(map_seed=0, points=False, **kw):
    """a configuration where every scalar map of the S1 / S2 chain is a real map: LCE (3-D), S2 correction and SE gain (2-D),
    field-dependence splines in (r, z), data-driven longitudinal diffusion in (r, z)"""
    rng = np.random.default_rng(map_seed)
    gx = np.linspace(-70, 70, 29)
    gz = np.linspace(-150, 2, 20)
    X, Y, Z = np.meshgrid(gx, gx, gz, indexing='ij')
    lce = 0.12 * (1.0 + 0.3 * np.cos(X / 50) * np.cos(Y / 45)) * (1.0 - 0.002 * Z)
    X2, Y2 = np.meshgrid(gx, gx, indexing='ij')
    s2c = 1.0 + 0.2 * np.exp(-(X2 ** 2 + Y2 ** 2) / 3000.0) + 0.01 * rng.standard_normal(X2.shape)
    seg = 28.0 * (1.0 + 0.1 * np.sin(X2 / 30.0) * np.cos(Y2 / 25.0))
    gr, gzz = np.linspace(0, 67, 23), np.linspace(-150, 0, 31)
    R, ZZ = np.meshgrid(gr, gzz, indexing='ij')
    fmap = dict(coordinate_system=[['r', [0, 67, 23]], ['z', [-150, 0, 31]]],
                drift_speed_map=0.6 + 0.05 * np.cos(R / 40) + 0.0004 * ZZ,          # mm / us
                survival_probability_map=np.clip(1.02 - 0.3 * (R / 67) ** 6 * (1 + ZZ / 300), None, None))
    dl = dict(coordinate_system=[['r', [0, 67, 12]], ['z', [-150, 0, 16]]], map=(28e-8 * (1 + 0.2 * np.cos(np.linspace(0, 3, 12))[:, None] * np.ones(16)[None, :])))
    maps = dict(s1_lce_correction_map=dict(coordinate_system=[['x', [-70, 70, 29]], ['y', [-70, 70, 29]], ['z', [-150, 2, 20]]], map=lce),
                s2_correction_map=dict(coordinate_system=[['x', [-70, 70, 29]], ['y', [-70, 70, 29]]], map=s2c),
                se_gain_map=dict(coordinate_system=[['x', [-70, 70, 29]], ['y', [-70, 70, 29]]], map=seg))
    if points:        # the same maps on an irregular coordinate system (a list of points)
        p3 = np.array([X.ravel(), Y.ravel(), Z.ravel()]).T + rng.uniform(-1, 1, (X.size, 3))
        p2 = np.array([X2.ravel(), Y2.ravel()]).T + rng.uniform(-1, 1, (X2.size, 2))
        maps = dict(s1_lce_correction_map=dict(coordinate_system=p3.tolist(), map=lce.ravel()),
                    s2_correction_map=dict(coordinate_system=p2.tolist(), map=s2c.ravel()),
                    se_gain_map=dict(coordinate_system=p2.tolist(), map=seg.ravel()))
    return xenonnt_test_config(ext_eff_from_map=True, se_gain_from_map=True, g2_mean=11.0, field_dependencies_map=fmap,
                               diffusion_longitudinal_map=dl,
                               enable_field_dependencies=dict(drift_speed_map=True, survival_probability_map=True, diffusion_longitudinal_map=True,
                                                              norm_drift_velocity=True),
                               **maps, **kw)


@pytest.mark.parametrize('points', [False, True])
def test_scalar_maps_match_the_host_maps(points):
    cfg = scalar_config(3, points, seed=5)
    res = Resource(cfg)
    eng = make_engine(cfg, resource=res)
    dev = eng.resource
    assert set(dev.on_device) == {'s1_lce_correction_map', 's2_correction_map', 'se_gain_map', 'field_dependencies_rz', 'diffusion_longitudinal_rz'}
    rng = np.random.default_rng(1)
    n = 5000
    xyz = np.array([rng.uniform(-75, 75, n), rng.uniform(-75, 75, n), rng.uniform(-160, 5, n)]).T       # incl. positions outside the maps
    for name, pos in (('s1_lce_correction_map', xyz), ('s2_correction_map', xyz[:, :2]), ('se_gain_map', xyz[:, :2])):
        h, d = getattr(res, name)(pos), getattr(dev, name)(pos)
        assert h.shape == d.shape and np.allclose(d, h, rtol=1e-6, atol=0), name
    z, xy = xyz[:, 2], xyz[:, :2]
    for name in ('drift_speed_map', 'survival_probability_map'):
        h, d = res.field_dependencies_map(z, xy, map_name=name), dev.field_dependencies_map(z, xy, map_name=name)
        assert np.allclose(d, h, rtol=1e-9, atol=1e-12), name
    assert np.allclose(dev.diffusion_longitudinal_map(z, xy), res.diffusion_longitudinal_map(z, xy), rtol=1e-6)
    # and the per-instruction inputs of the generator computed through the view
    ins = instructions(400, 8)
    ip_h, ip_d = instruction_params(ins, cfg, res), instruction_params(ins, cfg, dev)
    for k in ('p_hit', 'drift_mean', 'drift_spread', 'sc_gain'):
        assert np.allclose(ip_d[k], ip_h[k], rtol=1e-6, atol=1e-15), k
        assert np.any(ip_h[k] != ip_h[k][0])


def test_one_dimensional_and_named_maps():
    cfg = xenonnt_test_config(seed=3)
    eng = make_engine(cfg)
    m = InterpolatingMap(dict(coordinate_system=[['z', [-10, 0, 11]]], map=np.arange(11.0) ** 2, other=np.cos(np.arange(11.0))))
    from wfsim_amd.device_maps import DeviceMap
    dm = DeviceMap(eng, m)
    assert set(dm.ids) == {'map', 'other'}
    pos = np.random.default_rng(0).uniform(-12, 2, (300, 1))
    for name in ('map', 'other'):
        assert np.allclose(dm(pos, map_name=name), m(pos, map_name=name), rtol=1e-6, atol=1e-12)
    # a NaN coordinate gives NaN (the host's KD-tree refuses it outright)
    assert np.isnan(dm(np.array([[np.nan]]))[0])
    # array-valued maps and method RegularGridInterpolator are evaluated on the device too (round 4: wfs_scalar_map_*_array / _linear)
    pm = InterpolatingMap(dict(coordinate_system=[['x', [0, 1, 3]], ['y', [0, 1, 3]]], map=np.ones((3, 3, 4))))
    assert set(DeviceMap(eng, pm).ids) == {'map'} and DeviceMap(eng, pm)(np.array([[0.3, 0.4]])).shape == (1, 4)
    rg = InterpolatingMap(dict(coordinate_system=[['x', [0, 1, 3]], ['y', [0, 1, 3]]], map=np.ones((3, 3))), method='RegularGridInterpolator')
    assert set(DeviceMap(eng, rg).ids) == {'map'}
    # what stays on the host: a RegularGridInterpolator map given as a point list (straxen falls back to nearest neighbours there)
    pl = InterpolatingMap(dict(coordinate_system=[[0.0, 0.0], [1.0, 0.0], [0.0, 1.0], [1.0, 1.0], [0.5, 0.5]], map=np.arange(5.0)), method='RegularGridInterpolator')
    assert DeviceMap(eng, pl).ids == {}


def point_map_config(map_seed, **kw):
    """the regular-grid pattern maps of test_gpu_pattern_maps as jittered point lists"""
    cfg = map_config(map_seed, **kw)
    rng = np.random.default_rng(map_seed + 100)
    for kind in ('s1', 's2'):
        m = InterpolatingMap(cfg[kind + '_pattern_map'])
        pts = m.coordinate_system + rng.uniform(-0.8, 0.8, m.coordinate_system.shape)
        vals = np.asarray(cfg[kind + '_pattern_map']['map']).reshape(len(pts), -1)
        cfg[kind + '_pattern_map'] = dict(coordinate_system=pts.tolist(), map=vals)
    return cfg


@pytest.mark.parametrize('seed', [0, 1])
def test_point_list_pattern_maps(seed):
    cfg = point_map_config(seed, seed=11 + seed, s2_secondary_sc_gain=30.0)
    res = Resource(cfg)
    assert res.s2_pattern_map.grid is None and res.s1_pattern_map.grid is None
    eng = make_engine(cfg, resource=res)
    assert eng.device_maps == {'s1', 's2'}
    ins = instructions(120, 40 + seed)
    order, key, cluster = schedule(ins, cfg)
    s_ins, gid = ins[order], order.astype(np.uint32)
    ip = instruction_params(s_ins, cfg, res, device_maps=eng.device_maps)
    assert np.all(ip['cdf_row'] == -1)
    eng.load_instructions(s_ins, gid, cluster, key, ip)
    counts = eng.run()
    row, table = eng.cdf_rows()
    ip_host = instruction_params(s_ins, cfg, res)
    p_host = np.diff(ip_host['cdf_table'][ip_host['cdf_row']], axis=1, prepend=0.0)
    p_dev = np.diff(table[row], axis=1, prepend=0.0)
    assert np.allclose(p_dev, p_host, rtol=1e-6, atol=1e-12)
    orc = make_oracle(cfg, resource=res)
    orc.simulate(s_ins, gid, dict(ip, cdf_row=row, cdf_table=table))
    assert counts['n_photons'] == len(orc.results()['ph_t']) > 0
    assert eng.records().tobytes() == orc.pack_records().tobytes()


def test_aft_smearing_on_device_rows():
    """s2.py:660-665 on the rows the device makes: the same rows as the host path with the same per-instruction factors"""
    cfg = map_config(2, seed=21, s2_aft_sigma=0.09, s2_aft_skewness=1.2, s2_secondary_sc_gain=30.0)
    res = Resource(cfg)
    eng = make_engine(cfg, resource=res)
    assert eng._aft_on_device
    ins = instructions(200, 3)
    ins['type'] = 2
    order, key, cluster = schedule(ins, cfg)
    s_ins, gid = ins[order], order.astype(np.uint32)
    ip = instruction_params(s_ins, cfg, res, gids=gid, device_maps=eng.device_maps)
    assert np.all(ip['cdf_row'] == -1) and np.array_equal(ip['aft_factor'], s2_aft_factors(len(gid), cfg, gid))
    eng.load_instructions(s_ins, gid, cluster, key, ip)
    counts = eng.run()
    row, table = eng.cdf_rows()
    ip_host = instruction_params(s_ins, cfg, res, gids=gid)               # host rows, same factors
    p_host = np.diff(ip_host['cdf_table'][ip_host['cdf_row']], axis=1, prepend=0.0)
    p_dev = np.diff(table[row], axis=1, prepend=0.0)
    assert np.allclose(p_dev, p_host, rtol=1e-6, atol=1e-12)
    n_top = cfg['n_top_pmts']
    plain = instruction_params(s_ins, dict(cfg, s2_aft_sigma=0.0), res, gids=gid)
    aft0 = np.diff(plain['cdf_table'][plain['cdf_row']], axis=1, prepend=0.0)[:, :n_top].sum(axis=1)
    assert np.allclose(p_dev[:, :n_top].sum(axis=1), np.clip(aft0 * ip['aft_factor'], 0, 1), rtol=1e-9)
    orc = make_oracle(cfg, resource=res)
    orc.simulate(s_ins, gid, dict(ip, cdf_row=row, cdf_table=table))
    assert eng.records().tobytes() == orc.pack_records().tobytes() and counts['n_photons'] > 0


def test_rawdata_with_all_maps_on_the_device_is_batching_invariant():
    import wfsim_amd
    cfg = dict(scalar_config(4, seed=33, s2_aft_sigma=0.05), **{k: map_config(4)[k] for k in ('s1_pattern_map', 's2_pattern_map')})
    ins = instructions(80, 17)
    out = []
    for quanta in (2_000_000_000, 30_000):
        rd = wfsim_amd.RawData(cfg)
        assert rd.engine.device_maps == {'s1', 's2'} and len(rd.engine.resource.on_device) >= 5
        rd.max_batch_quanta = quanta
        out.append(b''.join(w['records'].tobytes() for w in rd.iter_windows(ins)))
    assert out[0] == out[1] and len(out[0]) > 0


@pytest.mark.parametrize('dims', [1, 2, 3])
@pytest.mark.parametrize('nv', [1, 5])
def test_regular_grid_interpolator_maps(dims, nv):
    """method 'RegularGridInterpolator' (load_resource.py:357, 383-401: scipy's multilinear interpolation, bounds_error=False,
    fill_value=None -- the edge cell continued outside the grid), scalar and array valued: wfs_scalar_map_linear ==
    itp_map.InterpolatingMap (which calls scipy) to rtol 1e-10, positions outside the grid included"""
    from wfsim_amd.device_maps import DeviceMap
    rng = np.random.default_rng(10 * dims + nv)
    axes = [('x', [-50.0, 50.0, 11]), ('y', [-40.0, 60.0, 7]), ('z', [-150.0, 0.0, 9])][:dims]
    shape = tuple(a[1][2] for a in axes)
    values = rng.normal(1.0, 0.3, shape + ((nv,) if nv > 1 else ()))
    host = InterpolatingMap(dict(coordinate_system=[[n, list(r)] for n, r in axes], map=values), method='RegularGridInterpolator')
    eng = make_engine(xenonnt_test_config())
    dev = DeviceMap(eng, host)
    assert 'map' in dev.ids
    pos = np.array([rng.uniform(r[0] - 20, r[1] + 20, 4000) for _, r in axes]).T
    got, ref = dev(pos), host(pos)
    assert got.shape == ref.shape and np.allclose(got, ref, rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize('points', [False, True])
def test_array_valued_nearest_neighbour_maps(points):
    """array-valued WeightedNearestNeighbors maps (make_map, load_resource.py:383-401) on a grid and on a point list: every value of a
    node is averaged with the node's weight -- wfs_scalar_map_grid_array / _points_array == the host map to rtol 1e-6"""
    from wfsim_amd.device_maps import DeviceMap
    rng = np.random.default_rng(3 + points)
    gx = np.linspace(-60, 60, 13)
    X, Y = np.meshgrid(gx, gx, indexing='ij')
    vals = np.stack([1.0 + 0.1 * np.cos(X / 20.0), 2.0 + 0.2 * np.sin(Y / 15.0), 0.5 + 0.001 * X * Y / 10], axis=-1)
    if points:
        p = np.array([X.ravel(), Y.ravel()]).T + rng.uniform(-1, 1, (X.size, 2))
        data = dict(coordinate_system=p.tolist(), map=vals.reshape(-1, 3))
    else:
        data = dict(coordinate_system=[['x', [-60, 60, 13]], ['y', [-60, 60, 13]]], map=vals)
    host = InterpolatingMap(data)
    eng = make_engine(xenonnt_test_config())
    dev = DeviceMap(eng, host)
    assert 'map' in dev.ids
    pos = rng.uniform(-70, 70, (3000, 2))
    got, ref = dev(pos), host(pos)
    assert got.shape == ref.shape == (3000, 3) and np.allclose(got, ref, rtol=1e-6)
