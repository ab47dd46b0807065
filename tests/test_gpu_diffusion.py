"""diffusion_constant_transverse with the diffusion_transverse_map field maps on the GPU (k_diffuse_patterns,
wfs_set_instruction_diffusion): the channel rows of the device against the host restatement fed with the same electron draws
(rtol 1e-6; that restatement is pinned on the reference in tests/test_diffusion_cpu.py), records against the oracle."""
import numpy as np
import pytest

from tests.helpers import make_engine, make_oracle, host_diffuse_patterns
from tests.test_gpu_pattern_maps import map_config, instructions
from wfsim_amd.physics import instruction_params
from wfsim_amd.resource import Resource
from wfsim_amd.scheduler import schedule

pytestmark = pytest.mark.gpu


def diffusion_config(**kw):
    gr, gz = np.linspace(0, 67, 12), np.linspace(-150, 0, 16)
    R, Z = np.meshgrid(gr, gz, indexing='ij')
    fmap = dict(coordinate_system=[['r', [0, 67, 12]], ['z', [-150, 0, 16]]],
                diffusion_radial_map=700.0 + 4.0 * R - 1.5 * Z, diffusion_azimuthal_map=250.0 + 1.0 * R - 0.5 * Z)      # cm^2 / s
    return map_config(1, seed=44, s2_secondary_sc_gain=30.0, tpc_radius=50.0, diffusion_constant_transverse=1.0, field_dependencies_map=fmap,
                      enable_field_dependencies=dict(diffusion_transverse_map=True), **kw)


def _as_point_list(cfg, seed=7):
    """the S2 pattern map of the configuration as a jittered point list (an irregular coordinate system): the device finds every
    electron's neighbours through its cell index (points_nearest_2d)"""
    from wfsim_amd.itp_map import InterpolatingMap
    m = InterpolatingMap(cfg['s2_pattern_map'])
    pts = m.coordinate_system + np.random.default_rng(seed).uniform(-0.8, 0.8, m.coordinate_system.shape)
    cfg['s2_pattern_map'] = dict(coordinate_system=pts.tolist(), map=np.asarray(cfg['s2_pattern_map']['map']).reshape(len(pts), -1))
    return cfg


@pytest.mark.parametrize('aft,points', [(False, False), (True, False), (False, True), (True, True)])
def test_rows_averaged_over_the_electrons(aft, points):
    cfg = diffusion_config(**(dict(s2_aft_sigma=0.06) if aft else {}))
    if points:
        cfg = _as_point_list(cfg)
    res = Resource(cfg)
    assert (res.s2_pattern_map.grid is None) == points
    eng = make_engine(cfg, resource=res)
    ins = instructions(60, 5)
    ins['amp'][ins['type'] == 2] = np.random.default_rng(2).choice([3, 40, 900, 5000], int(np.sum(ins['type'] == 2)))
    order, key, cluster = schedule(ins, cfg)
    s_ins, gid = ins[order], order.astype(np.uint32)
    ip = instruction_params(s_ins, cfg, eng.resource, gids=gid, device_maps=eng.device_maps)
    s2 = s_ins['type'] == 2
    assert ip['diff_sigma'] is not None and np.all(np.isfinite(ip['diff_sigma'][0][s2])) and np.all(np.isnan(ip['diff_sigma'][0][~s2]))
    assert 0.3 < ip['diff_sigma'][0][s2].max() < 6
    eng.load_instructions(s_ins, gid, cluster, key, ip)
    counts = eng.run()
    row, table = eng.cdf_rows()
    p_dev = np.diff(table[row], axis=1, prepend=0.0)
    orc = make_oracle(cfg, resource=res)
    xy = np.array([s_ins['x'], s_ins['y']], dtype=np.float64).T
    mine, n_in, _ = host_diffuse_patterns(orc, res.s2_pattern_map, xy[s2], gid[s2], s_ins['amp'][s2], ip['p_hit'][s2],
                                          ip['diff_sigma'][0][s2], ip['diff_sigma'][1][s2], cfg['tpc_radius'])
    off = np.asarray(cfg['gains']) == 0
    checked = 0
    for k, i in enumerate(np.where(s2)[0]):
        if n_in[k] == 0:
            continue
        p = mine[k].copy()
        p[off] = 0
        p /= p.sum()
        if aft:
            n_top = cfg['n_top_pmts']
            cur = p[:n_top].sum()
            new = np.clip(cur * ip['aft_factor'][i], 0, 1)
            p[:n_top] *= new / cur
            p[n_top:] *= (1 - new) / (1 - cur)
        assert np.allclose(p_dev[i], p, rtol=1e-6, atol=1e-12), i
        checked += 1
    assert checked > 10
    # the records: the oracle fed with the device's rows
    orc.simulate(s_ins, gid, dict(ip, cdf_row=row, cdf_table=table))
    assert counts['n_photons'] == len(orc.results()['ph_t']) > 0
    assert eng.records().tobytes() == orc.pack_records().tobytes()
    # the rows are reproducible (fixed summation order)
    eng.load_instructions(s_ins, gid, cluster, key, ip)
    eng.run()
    assert np.array_equal(eng.cdf_rows()[1], table)


def test_an_instruction_whose_electrons_all_leave_the_tpc_makes_no_photons():
    cfg = diffusion_config()
    res = Resource(cfg)
    eng = make_engine(cfg, resource=res)
    ins = instructions(4, 9)
    ins['type'], ins['amp'] = 2, 500
    ins['x'], ins['y'] = [0.0, 58.0, 10.0, -70.0], [0.0, 30.0, -5.0, 2.0]          # the second and fourth lie far outside r = 50
    order, key, cluster = schedule(ins, cfg)
    s_ins, gid = ins[order], order.astype(np.uint32)
    ip = instruction_params(s_ins, cfg, eng.resource, gids=gid, device_maps=eng.device_maps)
    eng.load_instructions(s_ins, gid, cluster, key, ip)
    eng.run()
    n_ph = np.diff(eng.instruction_photon_offsets())
    outside = np.hypot(s_ins['x'], s_ins['y']) > 55
    assert np.all(n_ph[outside] == 0) and np.all(n_ph[~outside] > 0)


def test_rawdata_with_transverse_diffusion_is_batching_invariant():
    import wfsim_amd
    cfg = diffusion_config()
    ins = instructions(50, 12)
    out = []
    for quanta in (2_000_000_000, 15_000):
        rd = wfsim_amd.RawData(cfg)
        rd.max_batch_quanta = quanta
        out.append(b''.join(w['records'].tobytes() for w in rd.iter_windows(ins)))
    assert out[0] == out[1] and len(out[0]) > 0


def test_transverse_diffusion_needs_the_map_on_the_device():
    cfg = diffusion_config(device_pattern_maps=False)
    res = Resource(cfg)
    with pytest.raises(NotImplementedError):
        instruction_params(instructions(5, 1), cfg, res)
