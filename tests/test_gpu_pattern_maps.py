"""Pattern maps evaluated on the device (wfs_set_pattern_map / wfs_eval_pattern_rows) against the host InterpolatingMap
(the restatement of straxen's WeightedNearestNeighbors) and, end to end, against the oracle fed with the device's rows.

Tolerance of the map evaluation: rtol 1e-6 on the channel probabilities (float32 maps, float64 accumulation in a different
order than numpy's); everything downstream of the rows is exact."""
import time

import numpy as np
import pytest

from tests.helpers import make_engine, make_oracle
from wfsim_amd.config import xenonnt_test_config
from wfsim_amd.dtypes import instruction_dtype
from wfsim_amd.physics import instruction_params
from wfsim_amd.resource import Resource
from wfsim_amd.scheduler import schedule

pytestmark = pytest.mark.gpu


def map_config(map_seed=0, anisotropic=False, **kw):
    rng = np.random.default_rng(map_seed)
    nx, ny = (41, 17) if anisotropic else (31, 31)
    gx, gy = np.linspace(-66, 66, nx), np.linspace(-66, 66, ny)
    pmt = rng.uniform(-60, 60, (494, 2))
    d2 = (gx[:, None, None] - pmt[None, None, :, 0]) ** 2 + (gy[None, :, None] - pmt[None, None, :, 1]) ** 2
    s2 = (1.0 / (1.0 + d2 / 40.0)).astype(np.float32)                          # [nx][ny][494]
    gz = np.linspace(-150, 0, 9)
    s1 = ((1.0 / (1.0 + d2[::3, ::2] / 900.0))[:, :, None, :] * (1.0 + 0.3 * np.cos(gz / 40.0))[None, None, :, None]).astype(np.float32)
    cfg = xenonnt_test_config(
        s2_pattern_map=dict(coordinate_system=[['x', [-66, 66, nx]], ['y', [-66, 66, ny]]], map=s2),
        s1_pattern_map=dict(coordinate_system=[['x', [-66, 66, s1.shape[0]]], ['y', [-66, 66, s1.shape[1]]], ['z', [-150, 0, 9]]], map=s1), **kw)
    cfg['gains'] = np.array(cfg['gains'], dtype=np.float64)
    cfg['gains'][[3, 260, 493]] = 0
    return cfg


def instructions(n, seed):
    rng = np.random.default_rng(seed)
    ins = np.zeros(n, dtype=instruction_dtype)
    ins['type'] = rng.choice([1, 2], n)
    ins['time'] = np.cumsum(rng.choice([300, 40_000, 3_000_000], n)).astype(np.int64) + 1_000_000
    r, phi = 48 * np.sqrt(rng.random(n)), rng.uniform(0, 2 * np.pi, n)
    ins['x'], ins['y'], ins['z'] = r * np.cos(phi), r * np.sin(phi), -rng.uniform(0.5, 95, n)
    ins['amp'] = np.where(ins['type'] == 1, rng.choice([0, 40, 700, 5000], n), rng.choice([0, 7, 60, 400, 2500], n))
    ins['recoil'], ins['event_number'] = 7, np.arange(n)
    return ins


@pytest.mark.parametrize('seed,anisotropic', [(0, False), (1, True), (2, False)])
def test_device_rows_match_the_host_map(seed, anisotropic):
    cfg = map_config(seed, anisotropic, seed=5)
    res = Resource(cfg)
    eng = make_engine(cfg, resource=res)
    assert eng.device_maps == {'s1', 's2'}
    ins = instructions(300, seed)
    order, key, cluster = schedule(ins, cfg)
    s_ins = ins[order]
    ip_dev = instruction_params(s_ins, cfg, res, device_maps=eng.device_maps)
    assert np.all(ip_dev['cdf_row'] == -1)
    eng.load_instructions(s_ins, order.astype(np.uint32), cluster, key, ip_dev)
    row, table = eng.cdf_rows()
    ip_host = instruction_params(s_ins, cfg, res)
    host = ip_host['cdf_table'][ip_host['cdf_row']]
    dev = table[row]
    p_host, p_dev = np.diff(host, axis=1, prepend=0.0), np.diff(dev, axis=1, prepend=0.0)
    assert np.all(p_dev[:, [3, 260, 493]] == 0)
    assert np.allclose(p_dev, p_host, rtol=1e-6, atol=1e-12)
    assert np.allclose(dev[:, -1], 1.0) and np.all(np.diff(dev, axis=1) >= 0)


@pytest.mark.parametrize('seed', [0, 1, 2, 3])
def test_records_with_device_maps_match_the_oracle(seed):
    """the generator on the device's own rows == the oracle fed with those rows, byte for byte"""
    cfg = map_config(seed, seed % 2 == 1, seed=100 + seed, s2_secondary_sc_gain=30.0)
    res = Resource(cfg)
    eng = make_engine(cfg, resource=res)
    ins = instructions(40, 50 + seed)
    order, key, cluster = schedule(ins, cfg)
    s_ins, gid = ins[order], order.astype(np.uint32)
    ip = instruction_params(s_ins, cfg, res, device_maps=eng.device_maps)
    eng.load_instructions(s_ins, gid, cluster, key, ip)
    counts = eng.run()
    row, table = eng.cdf_rows()
    orc = make_oracle(cfg, resource=res)
    orc.simulate(s_ins, gid, dict(ip, cdf_row=row, cdf_table=table))
    assert counts['n_photons'] == len(orc.results()['ph_t']) > 0
    assert eng.records().tobytes() == orc.pack_records().tobytes()


def test_mixed_host_and_device_rows():
    """S1 pattern on the device, S2 pattern a host callable: rows of both kinds in one batch"""
    cfg = map_config(4, seed=9)
    res = Resource(cfg)
    host_s2 = res.s2_pattern_map
    res.s2_pattern_map = (lambda pos, **kw: host_s2(pos))           # a plain callable: stays on the host
    eng = make_engine(cfg, resource=res)
    assert eng.device_maps == {'s1'}
    ins = instructions(30, 77)
    order, key, cluster = schedule(ins, cfg)
    s_ins, gid = ins[order], order.astype(np.uint32)
    ip = instruction_params(s_ins, cfg, res, device_maps=eng.device_maps)
    assert set(np.unique(ip['cdf_row'][s_ins['type'] == 1])) == {-1} and np.all(ip['cdf_row'][s_ins['type'] == 2] >= 0)
    eng.load_instructions(s_ins, gid, cluster, key, ip)
    eng.run()
    row, table = eng.cdf_rows()
    assert np.array_equal(table[:len(ip['cdf_table'])], ip['cdf_table'])
    orc = make_oracle(cfg, resource=res)
    orc.simulate(s_ins, gid, dict(ip, cdf_row=row, cdf_table=table))
    assert eng.records().tobytes() == orc.pack_records().tobytes()


def test_rawdata_uses_the_device_maps_and_is_batching_invariant():
    import wfsim_amd
    cfg = map_config(5, seed=21, s2_secondary_sc_gain=30.0)
    ins = instructions(60, 91)
    out = []
    for quanta in (2_000_000_000, 20_000):
        rd = wfsim_amd.RawData(cfg)
        assert rd.engine.device_maps == {'s1', 's2'}
        rd.max_batch_quanta = quanta
        out.append(b''.join(w['records'].tobytes() for w in rd.iter_windows(ins)))
    assert out[0] == out[1] and len(out[0]) > 0


def test_map_evaluation_speed():
    """10^4 S1 instructions: host evaluation takes seconds, the device a few milliseconds (printed, loosely asserted)"""
    cfg = map_config(6, seed=3)
    res = Resource(cfg)
    eng = make_engine(cfg, resource=res)
    ins = instructions(10_000, 5)
    ins['type'], ins['amp'] = 1, 100
    ins['time'] = 1_000_000 * (1 + np.arange(len(ins)))
    order, key, cluster = schedule(ins, cfg)
    ip = instruction_params(ins[order], cfg, res, device_maps=eng.device_maps)
    eng.load_instructions(ins[order], order.astype(np.uint32), cluster, key, ip)
    t0 = time.perf_counter()
    eng.load_instructions(ins[order], order.astype(np.uint32), cluster, key, ip)
    t_dev = time.perf_counter() - t0
    t0 = time.perf_counter()
    instruction_params(ins[order][:1000], cfg, res)
    t_host = 10 * (time.perf_counter() - t0)
    print(f'load + device map evaluation of 10^4 S1: {1e3 * t_dev:.1f} ms; host evaluation: {1e3 * t_host:.0f} ms')
    assert t_dev < t_host


def test_field_distortion_positions_reach_the_device_map():
    """comsol field distortion (s2.py:51-71): the S2 pattern is looked up at the OBSERVED position, also on the device;
    truth carries the mean observed position (rawdata.py:377-390)"""
    import wfsim_amd
    from wfsim_amd.dtypes import truth_extra_dtype
    rg = np.linspace(0, 70, 36)
    comsol = dict(coordinate_system=[['r', [0, 70, 36]], ['z', [-160, 10, 18]]], r_distortion_map=(0.8 * rg[:, None] * np.ones(18)[None, :]).tolist())
    cfg = map_config(7, seed=13, field_distortion_model='comsol', field_distortion_comsol_map=comsol)
    res = Resource(cfg)
    eng = make_engine(cfg, resource=res)
    ins = instructions(80, 21)
    order, key, cluster = schedule(ins, cfg)
    s_ins = ins[order]
    ip_dev = instruction_params(s_ins, cfg, res, device_maps=eng.device_maps)
    eng.load_instructions(s_ins, order.astype(np.uint32), cluster, key, ip_dev)
    row, table = eng.cdf_rows()
    ip_host = instruction_params(s_ins, cfg, res)
    p_host = np.diff(ip_host['cdf_table'][ip_host['cdf_row']], axis=1, prepend=0.0)
    p_dev = np.diff(table[row], axis=1, prepend=0.0)
    assert np.allclose(p_dev, p_host, rtol=1e-6, atol=1e-12)
    undistorted = instruction_params(s_ins, dict(cfg, field_distortion_model='none'), res)
    s2 = s_ins['type'] == 2
    assert not np.allclose(np.diff(undistorted['cdf_table'][undistorted['cdf_row']], axis=1, prepend=0.0)[s2], p_host[s2], rtol=1e-3)
    truth = np.zeros(400, dtype=instruction_dtype + truth_extra_dtype + [('fill', bool)])
    list(wfsim_amd.RawData(cfg).iter_windows(ins, truth_buffer=truth))
    t = truth[truth['fill']]
    t2 = t[t['type'] == 2]
    assert len(t2) > 5 and np.allclose(np.hypot(t2['x_mean_electron'], t2['y_mean_electron']), 0.8 * np.hypot(t2['x'], t2['y']), rtol=1e-4)
    assert np.all(np.isnan(t[t['type'] == 1]['x_mean_electron']))


def test_foreign_interpolating_map_objects_are_recognised():
    """an object that merely looks like straxen.InterpolatingMap (callable with a .data dict on a regular grid) also goes to
    the device"""
    cfg = map_config(8, seed=3)
    res = Resource(cfg)

    class Foreign:                       # what wfsim.load_config(...).s2_pattern_map is: data + __call__
        def __init__(self, m):
            self.data, self._m = m.data, m

        def __call__(self, *a, **k):
            return self._m(*a, **k)
    res.s1_pattern_map, res.s2_pattern_map = Foreign(res.s1_pattern_map), Foreign(res.s2_pattern_map)
    eng = make_engine(cfg, resource=res)
    assert eng.device_maps == {'s1', 's2'}
    ins = instructions(50, 3)
    order, key, cluster = schedule(ins, cfg)
    eng.load_instructions(ins[order], order.astype(np.uint32), cluster, key, instruction_params(ins[order], cfg, res, device_maps=eng.device_maps))
    row, table = eng.cdf_rows()
    host = instruction_params(ins[order], cfg, res)
    assert np.allclose(np.diff(table[row], axis=1, prepend=0.0), np.diff(host['cdf_table'][host['cdf_row']], axis=1, prepend=0.0), rtol=1e-6, atol=1e-12)


def test_top_only_s2_map_is_padded_with_ones_for_the_bottom_array():
    """s2.py:648-650: a pattern map that stops before the bottom channels gets weight 1 for every bottom PMT -- on the device too"""
    from wfsim_amd.itp_map import InterpolatingMap
    cfg = map_config(9, seed=4)
    cfg['s2_mean_area_fraction_top'] = -1
    res = Resource(cfg)
    # a map as the XENON1T loader builds it (load_resource.py:220: no PMT mask, top array only)
    res.s2_pattern_map = InterpolatingMap(dict(coordinate_system=cfg['s2_pattern_map']['coordinate_system'],
                                               map=np.asarray(cfg['s2_pattern_map']['map'])[..., :253] * 40))
    eng = make_engine(cfg, resource=res)
    assert 's2' in eng.device_maps
    ins = instructions(60, 33)
    ins['type'] = 2
    order, key, cluster = schedule(ins, cfg)
    s_ins = ins[order]
    eng.load_instructions(s_ins, order.astype(np.uint32), cluster, key, instruction_params(s_ins, cfg, res, device_maps=eng.device_maps))
    row, table = eng.cdf_rows()
    host = instruction_params(s_ins, cfg, res)
    p_host, p_dev = np.diff(host['cdf_table'][host['cdf_row']], axis=1, prepend=0.0), np.diff(table[row], axis=1, prepend=0.0)
    assert np.allclose(p_dev, p_host, rtol=1e-6, atol=1e-12)
    live_bottom = np.setdiff1d(np.arange(253, 494), [260, 493])
    assert np.allclose(p_dev[:, live_bottom], p_dev[:, live_bottom[:1]])          # equal weights on the bottom array
