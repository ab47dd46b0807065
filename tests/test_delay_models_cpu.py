"""Model variants of the photon delays (wfsim_amd/delay_models.py, the oracle's tables) against histograms of the
REFERENCE's own draws for the same models (tests/golden/dists_models.npz: S1 custom recoil models and optical
propagation, S2 garfield luminescence and optical propagation).  CPU only; two-sample KS tests + moments."""
import numpy as np
import pytest

from tests.helpers import golden, make_oracle
from tests.test_oracle_distributions import _ks, _ks_limit, _moments
from wfsim_amd.config import xenonnt_test_config
from wfsim_amd.delay_models import DelayModels, pmf_piecewise_linear, pmf_recombination, pmf_uniform, pmf_exp
from wfsim_amd.dtypes import instruction_dtype
from wfsim_amd.itp_map import InterpolatingMap
from wfsim_amd.resource import Resource

N = 1_000_000


def model_resources(d):
    """the synthetic spline / garfield resources the fixture was generated with"""
    s1 = dict(coordinate_system=[['z', [-100.0, 0.0, 11]], ['u', [0.0, 1.0, 21]]], top=d['s1_top'].tolist(), bottom=d['s1_bottom'].tolist())
    s2 = dict(coordinate_system=[['u', [0.0, 1.0, 33]]], top=d['s2_top'].tolist(), bottom=d['s2_bottom'].tolist())
    return dict(s1_time_spline=s1, s2_time_spline=s2, s2_luminescence=dict(t=d['garfield_t'], x=d['garfield_x']))


def one_instruction(typ, recoil=7, x=0.0, z=-33.3):
    ins = np.zeros(1, dtype=instruction_dtype)
    ins['type'], ins['recoil'], ins['x'], ins['z'], ins['amp'] = typ, recoil, x, z, 100
    return ins


def _check(d, key, x):
    v, c = d[key + '_v'], d[key + '_c']
    assert _ks(v, c, x) < _ks_limit(c.sum(), len(x)), key
    m, s = _moments(v, c)
    assert abs(x.mean() - m) < 5 * s / np.sqrt(len(x)) + 5 * s / np.sqrt(c.sum()), key
    # the spread of a heavy-tailed delay (ER recombination: kurtosis ~ 280) fluctuates by sqrt((kurt - 1) / 4n): 5 sigma of that
    kurt = ((v - m) ** 4 * c).sum() / c.sum() / s ** 4
    assert abs(x.std() / s - 1) < max(0.01, 5 * np.sqrt((kurt - 1) / 4 * (1 / len(x) + 1 / c.sum()))), key


@pytest.mark.parametrize('tag,model,recoil', [('er', 'custom', 7), ('nr', 'custom', 0), ('alpha', 'custom', 6), ('led', 'custom', 20),
                                              ('er_simple', 'simple+custom', 7)])
def test_s1_custom_models(tag, model, recoil):
    d = golden('dists_models.npz')
    cfg = dict(xenonnt_test_config(s1_model_type=model, led_pulse_length=33.3), seed=77)
    orc = make_oracle(cfg)
    tab, tabb, zi, zf = DelayModels(cfg, Resource(cfg)).instruction_tables(one_instruction(1, recoil))
    assert tab[0] >= 0 and tab[0] == tabb[0] and zi[0] == -1
    _check(d, 's1_' + tag, orc.sample_delay(N, False, tab=int(tab[0])))


def test_s1_custom_unknown_recoil_raises():
    cfg = xenonnt_test_config(s1_model_type='custom')
    with pytest.raises(AttributeError):
        DelayModels(cfg, Resource(cfg)).instruction_tables(one_instruction(1, recoil=3))


@pytest.mark.parametrize('side', ['top', 'bottom'])
def test_s1_optical_propagation(side):
    d = golden('dists_models.npz')
    cfg = dict(xenonnt_test_config(s1_model_type='simple+optical_propagation', **model_resources(d)), seed=78)
    res = Resource(cfg)
    orc = make_oracle(cfg, resource=res)
    tab, tabb, zi, zf = DelayModels(cfg, res).instruction_tables(one_instruction(1, z=-33.3))
    assert tab[0] == -1 and zi[0] == 6 and abs(zf[0] - 0.67) < 1e-5            # float32 z of the instruction
    _check(d, 's1_prop_' + side, orc.sample_delay(N, False, tab=-1, bottom=side == 'bottom', pzi=int(zi[0]), pzf=float(zf[0])))


def test_s1_propagation_is_the_regular_grid_interpolator():
    """the multilinear evaluation behind the oracle / device == scipy's RegularGridInterpolator on the same nodes"""
    d = golden('dists_models.npz')
    sp = InterpolatingMap(model_resources(d)['s1_time_spline'], method='RegularGridInterpolator')
    rng = np.random.default_rng(5)
    z, u = rng.uniform(-110, 5, 2000), rng.random(2000)                      # also outside the grid: extrapolation
    zg, ug = sp.grid
    i = np.clip(np.searchsorted(zg, z) - 1, 0, len(zg) - 2); zf = (z - zg[i]) / (zg[i + 1] - zg[i])
    j = np.clip(np.floor((u - ug[0]) / (ug[1] - ug[0])).astype(int), 0, len(ug) - 2); uf = (u - (ug[0] + j * (ug[1] - ug[0]))) / (ug[1] - ug[0])
    T = d['s1_top']
    v = T[i, j] * ((1 - zf) * (1 - uf)) + T[i, j + 1] * ((1 - zf) * uf) + T[i + 1, j] * (zf * (1 - uf)) + T[i + 1, j + 1] * (zf * uf)
    assert np.allclose(v, sp(np.array([z, u]).T, map_name='top'), rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize('tag,lum,tm', [('prop', 'simple', 'optical_propagation'), ('garfield', 'garfield', 'zero_delay'),
                                        ('garfield_prop', 'garfield', 'optical_propagation')])
@pytest.mark.parametrize('side', ['top', 'bottom'])
def test_s2_models(tag, lum, tm, side):
    d = golden('dists_models.npz')
    cfg = dict(xenonnt_test_config(s2_luminescence_model=lum, s2_time_model=tm, **model_resources(d)), seed=79)
    res = Resource(cfg)
    orc = make_oracle(cfg, resource=res)
    tab, tabb, zi, zf = DelayModels(cfg, res).instruction_tables(one_instruction(2, x=0.21))
    if tm == 'optical_propagation':
        assert tab[0] != tabb[0]
    _check(d, f's2_{tag}_{side}', orc.sample_delay(N, True, tab=int(tabb[0] if side == 'bottom' else tab[0])))


def test_garfield_rows_follow_the_wire_distance():
    d = golden('dists_models.npz')
    cfg = xenonnt_test_config(s2_luminescence_model='garfield', s2_time_model='zero_delay', **model_resources(d))
    m = DelayModels(cfg, Resource(cfg))
    ins = np.zeros(50, dtype=instruction_dtype)
    rng = np.random.default_rng(3)
    ins['type'], ins['x'], ins['y'] = 2, rng.uniform(-40, 40, 50), rng.uniform(-40, 40, 50)
    rows = m.garfield_rows(ins)
    tilt, pitch = np.pi / 4, 0.5
    y = ins['x'].astype(np.float64) * -np.sin(tilt) + ins['y'].astype(np.float64) * np.cos(tilt)     # second column of xy @ rot
    dist = (y + pitch / 2) % pitch - pitch / 2
    assert np.array_equal(rows, np.abs(dist[:, None] - d['garfield_x'][None, :]).argmin(axis=1))
    assert len(np.unique(rows)) > 3


def test_pmf_builders_match_direct_sampling():
    rng = np.random.default_rng(1)
    n = 2_000_000
    u, t = np.linspace(0, 1, 7), np.array([0, 0, 3.5, 2.2, -2.7, 9.1, 9.1])       # flat pieces, non-monotone, negative values
    p = pmf_piecewise_linear(u, t)
    x = np.trunc(np.interp(rng.random(n), u, t)).astype(int)
    assert np.abs(np.bincount(x - p.vmin, minlength=len(p.p)) / n - p.p).max() < 2e-3 and abs(p.p.sum() - 1) < 1e-12
    r = pmf_recombination(4.3)
    y = np.trunc(np.clip(4.3 / (-1 + 1 / rng.random(n)), 0, 1000)).astype(int)
    assert np.abs(np.bincount(y, minlength=len(r.p)) / n - r.p).max() < 2e-3
    e = pmf_exp(24.0)
    y = (rng.exponential(1, n) * 24.0).astype(int)
    assert np.abs(np.bincount(y, minlength=len(e.p))[:len(e.p)] / n - e.p).max() < 2e-3
    q = pmf_uniform(33.3)
    assert np.abs(np.bincount(rng.uniform(0, 33.3, n).astype(int)) / n - q.p).max() < 2e-3


def test_interpolating_map_methods():
    """1-D nearest-neighbour weighting is linear interpolation between nodes; array-valued 2-D maps; the file formats"""
    import gzip, json, os, tempfile
    m = InterpolatingMap(dict(coordinate_system=[['u', [0, 1, 5]]], map=[0, 1, 4, 9, 16.]))
    x = np.array([0.1, 0.3, 0.77])
    assert np.allclose(m(x[:, None]), np.interp(x, np.linspace(0, 1, 5), [0, 1, 4, 9, 16.]))
    rng = np.random.default_rng(2)
    data = dict(coordinate_system=[['x', [-1, 1, 5]], ['y', [-1, 1, 6]]], map=rng.random((5, 6, 7)).tolist())
    pm = InterpolatingMap(data)
    out = pm(np.array([[0.1, 0.2], [-0.9, 0.95]]))
    assert out.shape == (2, 7)
    # the 4 nearest nodes, weighted by 1 / distance
    gx, gy = np.linspace(-1, 1, 5), np.linspace(-1, 1, 6)
    P = np.array([[a, b] for a in gx for b in gy]); V = np.array(data['map']).reshape(-1, 7)
    dist = np.linalg.norm(P - np.array([0.1, 0.2]), axis=1); k = np.argsort(dist)[:4]
    assert np.allclose(out[0], (V[k] / dist[k, None]).sum(0) / (1 / dist[k]).sum())
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, 'map.json.gz')
        with gzip.open(path, 'wt') as f:
            json.dump(data, f)
        from wfsim_amd.resource import make_map
        assert np.allclose(make_map(path)(np.array([[0.1, 0.2]])), out[0])


@pytest.mark.parametrize('tag,x', [('narrow', -20.0), ('wide', 30.0)])
def test_gas_gap_warping_tables(tag, x):
    """enable_gas_gap_warping (s2.py:360-378): one luminescence table per gas gap, i.e. per position"""
    d = golden('dists_models.npz')
    cfg = dict(xenonnt_test_config(enable_gas_gap_warping=True, gas_gap_map=(lambda xy: 0.25 + 0.0004 * xy[:, 0]), s2_time_model='zero_delay'), seed=80)
    res = Resource(cfg)
    orc = make_oracle(cfg, resource=res)
    models = DelayModels(cfg, res)
    assert models.per_batch
    ins = np.concatenate([one_instruction(2, x=x), one_instruction(2, x=x), one_instruction(2, x=0.0), one_instruction(1)])
    tab, tabb, zi, zf = models.instruction_tables(ins)
    assert tab[0] == tab[1] != tab[2] and tab[3] == -1 and np.array_equal(tab, tabb)
    orc.set_delay_models(models)                     # the tables of this batch
    _check(d, 's2_warp_' + tag, orc.sample_delay(N, True, tab=int(tab[0])))


def gas_gap_resources(g=None):
    """config entries of the 'garfield_gas_gap' luminescence: the synthetic tables tests/golden/gas_gap.npz was made with"""
    g = g if g is not None else golden('gas_gap.npz')
    return dict(s2_luminescence_gg=dict(gas_gap=g['gas_gap'], timing_inv_cdf=g['timing_inv_cdf']),
                garfield_gas_gap_map=(lambda xy: 0.2 + 0.0009 * (np.asarray(xy)[:, 0] + 50.0)))


@pytest.mark.parametrize('k', [0, 1, 2])
def test_garfield_gas_gap_luminescence(k):
    """s2.py:413-483 on the reference's own draws: three instructions under different gas gaps, 250k photons each
    (inverse CDF interpolated between the two neighbouring tabulated gaps, the instruction's mean subtracted, truncated)"""
    g = golden('gas_gap.npz')
    cfg = dict(xenonnt_test_config(s2_luminescence_model='garfield_gas_gap', **gas_gap_resources(g)), seed=91 + k)
    res = Resource(cfg)
    models = DelayModels(cfg, res)
    ins = one_instruction(2, x=float(g['xy'][k, 0]))
    ins['y'] = g['xy'][k, 1]
    idx, w = models.instruction_gas_gap(ins)
    gaps = g['gas_gap']
    assert idx[0] == np.digitize(g['cont_gap'][k], gaps) - 1 and np.isclose(gaps[idx[0]] + w[0] * (gaps[1] - gaps[0]), g['cont_gap'][k])
    orc = make_oracle(cfg, resource=res)
    x = orc.sample_gas_gap(250_000, int(idx[0]), float(w[0]))
    _check(g, f'lum{k}', x)


def test_garfield_gas_gap_needs_its_resources():
    with pytest.raises(KeyError):
        Resource(xenonnt_test_config(s2_luminescence_model='garfield_gas_gap'))
    ins = one_instruction(1)
    cfg = xenonnt_test_config(s2_luminescence_model='garfield_gas_gap', **gas_gap_resources())
    idx, w = DelayModels(cfg, Resource(cfg)).instruction_gas_gap(ins)
    assert idx[0] == -1                     # S1s carry no luminescence term


def test_gas_gap_below_the_first_table_warns_and_uses_the_first_table():
    """the reference's np.digitize(...) - 1 = -1 wraps to the LAST table (s2.py:476-477, numpy's negative index); here the first
    table is used, with a warning -- the documented deviation"""
    res = gas_gap_resources()
    res['garfield_gas_gap_map'] = lambda xy: np.full(len(xy), float(res['s2_luminescence_gg']['gas_gap'][0]) - 0.01)
    cfg = xenonnt_test_config(s2_luminescence_model='garfield_gas_gap', **res)
    models = DelayModels(cfg, Resource(cfg))
    with pytest.warns(UserWarning, match='below the first tabulated'):
        idx, w = models.instruction_gas_gap(one_instruction(2))
    assert idx[0] == 0 and w[0] < 0
