"""Host-side per-instruction physics (wfsim_amd/physics.py): the S2 pattern variants of S2.photon_channels
(/root/reference/wfsim/core/s2.py:616-682) and the map resource formats.  CPU only."""
import bz2

import numpy as np
import pytest

from wfsim_amd.config import xenonnt_test_config
from wfsim_amd.dtypes import instruction_dtype
from wfsim_amd.physics import instruction_params, s2_channel_probabilities
from wfsim_amd.resource import Resource, make_patternmap


def _s2(n, seed=0):
    rng = np.random.default_rng(seed)
    ins = np.zeros(n, dtype=instruction_dtype)
    ins['type'], ins['z'], ins['amp'] = 2, -10.0, 100
    ins['x'], ins['y'] = rng.uniform(-30, 30, n), rng.uniform(-30, 30, n)
    return ins


def test_transverse_diffusion_constant_is_the_reference_noop():
    """s2.py:581 reads the constant with getattr() on a dict -> 0: the 'diffused' pattern is the pattern at xy; only
    instructions outside tpc_radius lose their pattern (s2.py:598)"""
    ins = _s2(5)
    ins['x'][2] = 60.0
    a = instruction_params(ins, xenonnt_test_config(diffusion_constant_transverse=0), Resource(xenonnt_test_config()))
    cfg = xenonnt_test_config(diffusion_constant_transverse=2e-8)
    b = instruction_params(ins, cfg, Resource(cfg))
    keep = np.arange(5) != 2
    assert np.array_equal(a['cdf_table'][a['cdf_row'][keep]], b['cdf_table'][b['cdf_row'][keep]])
    assert np.array_equal(a['p_hit'][keep], b['p_hit'][keep]) and b['p_hit'][2] == 0.0
    cfg = xenonnt_test_config(diffusion_constant_transverse=2e-8, enable_field_dependencies=dict(diffusion_transverse_map=True))
    with pytest.raises(NotImplementedError):
        s2_channel_probabilities(np.zeros((1, 2)), cfg, Resource(xenonnt_test_config()))


def test_aft_smearing_follows_the_skew_normal_and_the_instruction_id():
    from scipy.stats import skewnorm
    n = 4000
    cfg = xenonnt_test_config(s2_aft_sigma=0.08, s2_aft_skewness=1.5, seed=9)
    res = Resource(cfg)
    pos = np.zeros((n, 2))
    p0 = s2_channel_probabilities(pos[:1], dict(cfg, s2_aft_sigma=0.0), res)[0]
    top = np.arange(cfg['n_top_pmts'])
    aft0 = p0[top].sum()
    p = s2_channel_probabilities(pos, cfg, res, gids=np.arange(n))
    assert np.allclose(p.sum(axis=1), 1.0)
    ratio = p[:, top].sum(axis=1) / aft0
    m, v = skewnorm.stats(a=1.5, loc=1.0, scale=0.08, moments='mv')
    assert abs(ratio.mean() - m) < 5 * np.sqrt(v / n) and abs(ratio.std() / np.sqrt(v) - 1) < 0.05
    # inside one array the pattern keeps its shape
    assert np.allclose(p[7, top] / p[7, top].sum(), p0[top] / p0[top].sum())
    # keyed by the run-wide instruction id: independent of how the run is cut into batches
    q = s2_channel_probabilities(pos[:10], cfg, res, gids=np.arange(100, 110))
    assert np.array_equal(q, p[100:110])
    ip = instruction_params(_s2(6), cfg, res, gids=np.arange(6))
    assert len(np.unique(ip['cdf_row'])) == 6


def test_pattern_map_formats_mask_quantised_compressed():
    """make_patternmap (load_resource.py:403-435): PMT mask, quantised storage, compressed storage"""
    rng = np.random.default_rng(4)
    m = rng.random((5, 6, 8)).astype(np.float32)
    csys = [['x', [-1, 1, 5]], ['y', [-1, 1, 6]]]
    mask = np.ones(8, dtype=bool); mask[3] = False
    plain = make_patternmap(dict(coordinate_system=csys, map=m), pmt_mask=mask)
    pts = np.array([[0.13, -0.4], [0.9, 0.9]])
    out = plain(pts)
    assert out.shape == (2, 8) and np.all(out[:, 3] == 0) and np.all(out[:, 0] > 0)
    q = np.round(m / 0.001).astype(np.uint16)
    quant = make_patternmap(dict(coordinate_system=csys, map=q, quantized=0.001), pmt_mask=mask)
    assert np.allclose(quant(pts), out, atol=1e-3)
    comp = make_patternmap(dict(coordinate_system=csys, map=bz2.compress(q.tobytes()), compressed=('bz2', 'uint16', q.shape), quantized=0.001), pmt_mask=mask)
    assert np.array_equal(comp(pts), quant(pts))
    with pytest.raises(NotImplementedError):
        make_patternmap(dict(coordinate_system=csys, map=b'', compressed=('blosc', 'uint16', q.shape)))


def test_interpolated_pattern_maps_feed_the_channel_tables():
    """a position dependent (InterpolatingMap) S1 and S2 pattern through instruction_params: one CDF row per instruction"""
    rng = np.random.default_rng(6)
    s2map = dict(coordinate_system=[['x', [-50, 50, 11]], ['y', [-50, 50, 11]]], map=rng.random((11, 11, 494)))
    s1map = dict(coordinate_system=[['x', [-50, 50, 5]], ['y', [-50, 50, 5]], ['z', [-100, 0, 6]]], map=rng.random((5, 5, 6, 494)))
    cfg = xenonnt_test_config(s1_pattern_map=s1map, s2_pattern_map=s2map)
    cfg['gains'] = np.array(cfg['gains'], dtype=np.float64); cfg['gains'][[5, 300]] = 0
    res = Resource(cfg)
    ins = _s2(7)
    ins['type'][:3] = 1
    ip = instruction_params(ins, cfg, res)
    cdf = ip['cdf_table'][ip['cdf_row']]
    assert cdf.shape == (7, 494) and np.allclose(cdf[:, -1], 1.0) and np.all(np.diff(cdf, axis=1) >= 0)
    p = np.diff(cdf, axis=1, prepend=0.0)
    assert np.all(p[:, [5, 300]] == 0)                       # turned-off PMTs never fire
    assert len(np.unique(ip['cdf_row'])) == 7


def test_error_behaviour_of_bad_configs():
    """the reference's exceptions for this path: ValueError for an unknown detector (load_resource.py:115),
    KeyError for unknown S2 model names (s2.py:536, 552), AssertionError for unknown S1 model names (s1.py:52-58)"""
    from wfsim_amd.config import kernel_params
    with pytest.raises(ValueError):
        Resource(xenonnt_test_config(detector='LZ'))
    with pytest.raises(KeyError):
        kernel_params(xenonnt_test_config(s2_luminescence_model='fancy'))
    with pytest.raises(KeyError):
        kernel_params(xenonnt_test_config(s2_time_model='whenever'))
    with pytest.raises(AssertionError):
        kernel_params(xenonnt_test_config(s1_model_type='simple+magic'))
    with pytest.raises(NotImplementedError):
        kernel_params(xenonnt_test_config(s1_model_type='nest'))


def test_field_distortion_models_move_the_observed_position():
    """S2.__call__ (s2.py:81-103): survival / drift at the true position, S2 maps at the observed one; comsol: radial
    map; inverse_fdc: fixed point of r_obs = r - dr(r_obs...) after six damped iterations"""
    from wfsim_amd.physics import s2_observed_positions
    ins = _s2(6, seed=3)
    # comsol: r_obs = 0.9 * r on a regular (r, z) grid
    rg, zg = np.linspace(0, 70, 36), np.linspace(-160, 10, 18)
    comsol = dict(coordinate_system=[['r', [0, 70, 36]], ['z', [-160, 10, 18]]], r_distortion_map=(0.9 * rg[:, None] * np.ones(18)[None, :]).tolist())
    cfg = xenonnt_test_config(field_distortion_model='comsol', field_distortion_comsol_map=comsol)
    res = Resource(cfg)
    z_obs, xy = s2_observed_positions(ins, cfg, res)
    r = np.hypot(ins['x'], ins['y']).astype(np.float64)
    assert np.allclose(np.hypot(xy[:, 0], xy[:, 1]), 0.9 * r, rtol=1e-6) and np.allclose(np.arctan2(xy[:, 1], xy[:, 0]), np.arctan2(ins['y'], ins['x']))
    ip = instruction_params(ins, cfg, res)
    assert np.allclose(ip['pattern_xy'], xy)
    base = instruction_params(ins, xenonnt_test_config(), Resource(xenonnt_test_config()))
    assert np.array_equal(ip['drift_mean'], base['drift_mean'])               # drift from the true position
    # inverse_fdc with a constant correction dr = 1.5 cm: r_obs = r - 1.5, z_obs = -sqrt(z^2 + dr^2)
    fdc = dict(coordinate_system=[['x', [-70, 70, 8]], ['y', [-70, 70, 8]], ['z', [-160 / 1.335e-4 * -1, 0, 5]]], map=np.full((8, 8, 5), 1.5).tolist())
    cfg = xenonnt_test_config(field_distortion_model='inverse_fdc', fdc_3d=fdc)
    z_obs, xy = s2_observed_positions(ins, cfg, Resource(cfg))
    assert np.allclose(np.hypot(xy[:, 0], xy[:, 1]), r - 1.5, rtol=1e-6)
    assert np.allclose(z_obs, -np.sqrt(ins['z'].astype(np.float64) ** 2 + 1.5 ** 2))


def test_field_dependency_maps():
    """enable_field_dependencies (load_resource.py:318-347, s2.py:158-179, 241-252): drift speed map with normalisation,
    survival probability map, data-driven longitudinal diffusion -- all host maps over (r, z)"""
    from wfsim_amd.physics import s2_drift_time_params, s2_electron_survival
    rg, zg = np.linspace(0, 70, 15), np.linspace(-150, 0, 16)
    speed = (1.2 + 0.002 * rg[:, None] + 0.0 * zg[None, :])                         # mm / us
    surv = np.clip(0.9 - 0.002 * rg[:, None] + 0.0 * zg[None, :], 0, 1)
    fmap = dict(coordinate_system=[['r', [0, 70, 15]], ['z', [-150, 0, 16]]], drift_speed_map=speed.tolist(), survival_probability_map=surv.tolist())
    dmap = dict(coordinate_system=[['r', [0, 70, 15]], ['z', [-150, 0, 16]]], map=np.full((15, 16), 3.1e-8).tolist())
    efd = dict(drift_speed_map=True, survival_probability_map=True, diffusion_longitudinal_map=True, norm_drift_velocity=True)
    cfg = xenonnt_test_config(enable_field_dependencies=efd, field_dependencies_map=fmap, diffusion_longitudinal_map=dmap)
    res = Resource(cfg)
    assert np.isclose(res.drift_velocity_scaling, cfg['drift_velocity_liquid'] / 1.2e-4)
    ins = _s2(5, seed=8)
    xy = np.array([ins['x'], ins['y']], dtype=np.float64).T
    z = ins['z'].astype(np.float64)
    r = np.hypot(xy[:, 0], xy[:, 1])
    mean, spread = s2_drift_time_params(z, xy, cfg, res)
    v = (1.2 + 0.002 * r) * 1e-4 * res.drift_velocity_scaling
    assert np.allclose(mean, -z / v + cfg['drift_time_gate'], rtol=1e-6)
    assert np.allclose(spread, np.sqrt(2 * 3.1e-8 * mean) / v, rtol=1e-6)
    cy = s2_electron_survival(z, xy, xy, cfg, res)
    assert np.allclose(cy, cfg['electron_extraction_yield'] * np.exp(-mean / cfg['electron_lifetime_liquid']) * (0.9 - 0.002 * r), rtol=1e-5)


def test_s2_mean_area_fraction_top_rescales_the_pattern_map():
    """load_resource.py:255-272: the S2 pattern map is rescaled to the configured mean AFT, total efficiency preserved"""
    rng = np.random.default_rng(12)
    m = rng.random((9, 9, 494)) + 0.1
    s2map = dict(coordinate_system=[['x', [-60, 60, 9]], ['y', [-60, 60, 9]]], map=m)
    plain = Resource(xenonnt_test_config(s2_pattern_map=s2map, s2_mean_area_fraction_top=-1))
    scaled = Resource(xenonnt_test_config(s2_pattern_map=s2map, s2_mean_area_fraction_top=0.7))
    a, b = np.array(plain.s2_pattern_map.data['map']), np.array(scaled.s2_pattern_map.data['map'])
    aft = (b[..., :253].sum(-1) / b.sum(-1)).mean()
    assert abs(aft - 0.7) < 0.01 and not np.allclose(a, b)
    orig = (a[..., :253].sum(-1) / a.sum(-1)).mean()
    assert np.allclose(b[..., :253], a[..., :253] * 0.7 / orig) and np.allclose(b[..., 253:], a[..., 253:] * 0.3 / (1 - orig))
    pos = np.array([[3.0, -7.0]])
    assert scaled.s2_pattern_map(pos).shape == (1, 494)


def test_correction_maps_derived_from_the_pattern_maps():
    """load_resource.py:242-284: without explicit maps the S1 light-yield map is the pattern summed over the live PMTs, the
    S2 correction map the summed pattern divided by its median"""
    rng = np.random.default_rng(13)
    s2map = dict(coordinate_system=[['x', [-60, 60, 7]], ['y', [-60, 60, 7]]], map=rng.random((7, 7, 494)) + 0.1)
    s1map = dict(coordinate_system=[['x', [-60, 60, 4]], ['y', [-60, 60, 4]], ['z', [-100, 0, 5]]], map=0.001 * (rng.random((4, 4, 5, 494)) + 0.1))
    cfg = xenonnt_test_config(s1_pattern_map=s1map, s2_pattern_map=s2map, s1_lce_correction_map=None, s2_correction_map=None,
                              s2_mean_area_fraction_top=-1)
    cfg['gains'] = np.array(cfg['gains'], dtype=np.float64); cfg['gains'][10] = 0
    res = Resource(cfg)
    live = np.arange(494) != 10
    node = np.array([[-60.0, -60.0, -100.0]])
    assert np.allclose(res.s1_lce_correction_map(node), s1map['map'][0, 0, 0, live].sum(), rtol=1e-4)
    tot = s2map['map'][..., live].sum(-1)
    assert np.allclose(res.s2_correction_map(np.array([[-60.0, -60.0]])), tot[0, 0] / np.median(tot), rtol=1e-4)
    ins = _s2(4, seed=2)
    ins['type'][:2] = 1
    ip = instruction_params(ins, cfg, res)
    assert np.all(ip['p_hit'] > 0) and np.all(ip['sc_gain'][2:] > 0)


def test_aft_smearing_against_the_reference_draws():
    """tests/golden/aft_sigma.npz: S2.photon_channels of the reference with s2_aft_sigma = 0.15, skewness 2 -- photons on the
    top array for 3000 instructions of 2000 photons each.  Ours: the per-instruction top fraction of the smeared rows, with
    the binomial counting noise of 2000 photons on top; two-sample KS and moments."""
    from tests.helpers import golden
    from tests.test_oracle_distributions import _ks, _ks_limit
    d = golden('aft_sigma.npz')
    p0, n_ph = d['pattern'], int(d['n_photons'])
    cfg = xenonnt_test_config(s2_aft_sigma=float(d['sigma']), s2_aft_skewness=float(d['skewness']), seed=4)
    n_top = cfg['n_top_pmts']

    class Res:
        @staticmethod
        def s2_pattern_map(pos):
            return np.tile(p0, (len(pos), 1))
    n = 20000
    p = s2_channel_probabilities(np.zeros((n, 2)), cfg, Res, gids=np.arange(n))
    aft = p[:, :n_top].sum(axis=1)
    assert np.allclose(p.sum(axis=1), 1.0) and aft.max() <= 1.0 + 1e-12
    ours = np.random.default_rng(0).binomial(n_ph, np.clip(aft, 0, 1))
    v, c = np.unique(d['top_counts'], return_counts=True)
    assert _ks(v, c, ours) < _ks_limit(c.sum(), n)
    ref = d['top_counts'] / n_ph
    assert abs(ours.mean() / n_ph - ref.mean()) < 5 * ref.std() * np.sqrt(1 / n + 1 / len(ref))
    assert abs(ours.std() / n_ph / ref.std() - 1) < 0.05
    assert abs((aft > 1 - 1e-9).mean() - (ref == 1).mean()) < 0.004              # instructions clipped to the top array only
    # inside the top array the pattern keeps its shape (pooled reference histogram against p0)
    h = d['top_hist'] / d['top_hist'].sum()
    q = p0[:n_top] / p0[:n_top].sum()
    assert np.abs(h - q).max() < 5 * np.sqrt(q.max() / d['top_hist'].sum())
