"""Random stages of the CPU oracle (Philox streams) against histograms of the REFERENCE's own draws
(tests/golden/dists.npz, made by running the reference under numpy's seeded generator).  CPU only.

The streams cannot be identical (numpy's MT19937 vs counter-based Philox), so these are two-sample tests on
integer-valued variates: a Kolmogorov-Smirnov distance and the first two moments.
"""
import numpy as np
import pytest

from tests.helpers import golden, make_oracle, host_tables
from wfsim_amd.config import xenonnt_test_config
from wfsim_amd.dtypes import instruction_dtype
from wfsim_amd.physics import instruction_params
from wfsim_amd.resource import Resource

N = 1_000_000


def _ks(values, counts, sample):
    """KS distance between a reference histogram (values, counts) and an integer sample"""
    sv, sc = np.unique(sample, return_counts=True)
    grid = np.union1d(values, sv)
    c1 = np.cumsum(np.bincount(np.searchsorted(grid, values), weights=counts, minlength=len(grid))) / counts.sum()
    c2 = np.cumsum(np.bincount(np.searchsorted(grid, sv), weights=sc, minlength=len(grid))) / sc.sum()
    return np.abs(c1 - c2).max()


def _ks_limit(n1, n2, alpha_c=1.95):      # c(alpha = 0.001) = 1.95
    return alpha_c * np.sqrt(1.0 / n1 + 1.0 / n2)


def _moments(values, counts):
    m = (values * counts).sum() / counts.sum()
    return m, np.sqrt(((values - m) ** 2 * counts).sum() / counts.sum())


@pytest.fixture(scope='module')
def orc():
    return make_oracle(dict(xenonnt_test_config(), seed=1234))


@pytest.mark.parametrize('kind,key', [(0, 'lum'), (1, 'st_gas'), (2, 'tts'), (3, 's1_simple')])
def test_timing_terms(orc, kind, key):
    d = golden('dists.npz')
    v, c = d[key + '_v'], d[key + '_c']
    x = orc.sample_term(kind, N)
    assert _ks(v, c, x) < _ks_limit(c.sum(), N), key
    m, s = _moments(v, c)
    assert abs(x.mean() - m) < 5 * s / np.sqrt(N) + 5 * s / np.sqrt(c.sum())
    assert abs(x.std() / s - 1) < 0.01


@pytest.mark.parametrize('slot', [0, 7, 8])
def test_alias_tables_sample_the_pmf_of_their_table(orc, slot):
    """photon delays are drawn with Walker's alias method: the distribution the cells encode equals the pmf of the
    cumulative table (which the tests below pin on the reference's draws) to the 2^-32 resolution of the thresholds"""
    pa, pc, _ = orc.alias_pmf(slot)
    assert abs(pa.sum() - 1) < 1e-12 and abs(pc.sum() - 1) < 1e-12
    assert np.abs(pa - pc).max() < 2.0 ** -31


def test_alias_draw_of_the_transit_time_matches_the_inverse_cdf_draw(orc):
    a, b = orc.sample_term(2, N), orc.sample_term(10, N)      # alias vs inverse CDF of the same table
    va, ca = np.unique(a, return_counts=True)
    assert _ks(va, ca, b) < _ks_limit(N, N)


def test_tables_reproduce_the_reference_expressions(orc):
    """trunc(Exp * tau) and trunc(np.interp(u, ...)) sampled from their tables == the reference's expressions on the
    same uniforms (pulse.py:341, s2.py:338)"""
    assert np.array_equal(orc.sample_term(7, N), orc.sample_term(9, N))
    assert np.array_equal(orc.sample_term(8, N), orc.sample_term(0, N))


@pytest.mark.parametrize('total,parts', [(5, (3, 2)), (6, (0, 1, 2))])
def test_summed_delay_table_is_the_sum_of_its_terms(orc, total, parts):
    """one draw from the convolution table vs independently drawn terms added up (s1.py:193-194 / s2.py:504-557 + pulse.py:54-56)"""
    x = orc.sample_term(total, N)
    rng = np.random.default_rng(7)
    y = sum(orc.sample_term(k, N)[rng.permutation(N)] for k in parts)      # permuted: the term samplers share draw sites
    sv, sc = np.unique(y, return_counts=True)
    assert _ks(sv, sc, x) < _ks_limit(N, N)
    assert abs(x.mean() - y.mean()) < 5 * x.std() * np.sqrt(2 / N)
    assert abs(x.std() / y.std() - 1) < 0.01


@pytest.mark.parametrize('tag', ['z10', 'z90'])
def test_electron_arrival_and_photons_per_electron(orc, tag):
    d = golden('dists.npz')
    dm, ds = d[f'drift_{tag}']
    # drift parameters come from the host physics; check them against the reference's numbers first
    cfg = xenonnt_test_config()
    ins = np.zeros(1, dtype=instruction_dtype)
    ins['type'], ins['z'], ins['amp'] = 2, (-10.0 if tag == 'z10' else -90.0), 100
    ip = instruction_params(ins, cfg, Resource(cfg))
    # (the fixture evaluated them with a float64 z; S2.__call__ hands over the instruction's float32 z, as we do)
    assert abs(ip['drift_mean'][0] / dm - 1) < 1e-6 and abs(ip['drift_spread'][0] / ds - 1) < 1e-6
    assert ip['sc_gain'][0] == d[f'sc_gain_{tag}'][0]
    x = orc.sample_term(4, 400_000, float(dm), float(ds))
    v, c = d[f'etime_{tag}_v'], d[f'etime_{tag}_c']
    assert _ks(v, c, x) < _ks_limit(c.sum(), len(x))
    k = orc.sample_poisson(float(ip['sc_gain'][0]), 400_000)
    v, c = d[f'nph_e_{tag}_v'], d[f'nph_e_{tag}_c']
    assert _ks(v, c, k) < _ks_limit(c.sum(), len(k))
    m, s = _moments(v, c)
    assert abs(k.mean() - m) < 6 * s / np.sqrt(len(k))


def test_poisson_small_mean(orc):
    k = orc.sample_poisson(3.7, 500_000)
    assert abs(k.mean() - 3.7) < 0.02 and abs(k.var() - 3.7) < 0.05


def test_binomial_counts_and_channels():
    d = golden('dists.npz')
    cfg = dict(xenonnt_test_config(), seed=99)
    res = Resource(cfg)
    n = 3_000
    ins = np.zeros(n, dtype=instruction_dtype)
    ins['type'], ins['amp'], ins['z'] = 1, 1667, -50.0
    ins['time'] = 1_000_000 * (1 + np.arange(n))
    ip = instruction_params(ins, cfg, res)
    o = make_oracle(cfg)
    o.simulate(ins, np.arange(n, dtype=np.uint32), ip)
    r = o.results()
    nh = np.diff(r['call_ph_off'])
    v, c = d['s1_nhits_v'], d['s1_nhits_c']
    assert _ks(v, c, nh) < _ks_limit(c.sum(), n)
    m, s = _moments(v, c)
    assert abs(nh.mean() - m) < 6 * s / np.sqrt(n)
    # uniform dummy pattern: channels are uniform over the 494 PMTs (chi-square)
    cnt = np.bincount(r['ph_ch'], minlength=494)
    chi2 = ((cnt - cnt.mean()) ** 2 / cnt.mean()).sum()
    assert chi2 < 494 + 6 * np.sqrt(2 * 494)
    # DPE fraction and SPE table index uniformity (pulse.py:76-79, 226)
    assert abs(r['ph_dpe'].mean() - d['dpe_frac'][0]) < 5 * np.sqrt(0.219 * 0.781 / len(r['ph_dpe']))


def test_s2_photon_times_and_survival():
    d = golden('dists.npz')
    cfg = dict(xenonnt_test_config(), seed=5)
    res = Resource(cfg)
    ins = np.zeros(40, dtype=instruction_dtype)
    ins['type'], ins['amp'], ins['z'] = 2, 1000, -10.0
    ins['time'] = 1_000_000 * (1 + np.arange(40))
    ip = instruction_params(ins, cfg, res)
    o = make_oracle(cfg)
    o.simulate(ins, np.arange(40, dtype=np.uint32), ip)
    r = o.results()
    # surviving electrons ~ Binomial(1000, cy)
    ne = np.diff(r['call_e_off'])
    v, c = d['nel_z10_v'], d['nel_z10_c']
    m, s = _moments(v, c)
    assert abs(ne.mean() - m) < 5 * s / np.sqrt(len(ne))
    # photon time relative to the instruction: all terms together (photons of one electron are correlated, so the
    # effective sample size is the number of electrons)
    t = r['ph_t'] - np.repeat(ins['time'], np.diff(r['call_ph_off']))
    v, c = d['s2_full_v'], d['s2_full_c']
    assert _ks(v, c, t) < _ks_limit(20000, ne.sum(), 2.5)
    m, s = _moments(v, c)
    assert abs(t.mean() - m) < 6 * s / np.sqrt(ne.sum()) and abs(t.std() / s - 1) < 0.03


@pytest.mark.parametrize('lam', [0.3, 9.9, 17.5, 82.03, 216.9, 400.0])
def test_photons_per_electron_are_poisson(orc, lam):
    """the per-instruction Poisson table (gains up to 217) and PTRS (above) against scipy's exact pmf: chi-square over the
    populated values, mean and variance"""
    from scipy.stats import poisson, chisquare
    n = 400_000
    k = orc.sample_poisson(lam, n)
    assert k.min() >= 0
    assert abs(k.mean() - lam) < 5 * np.sqrt(lam / n) and abs(k.var() / lam - 1) < 0.02
    vals, cnt = np.unique(k, return_counts=True)
    exp = poisson.pmf(vals, lam) * n
    keep = exp > 20
    obs, ex = cnt[keep].astype(float), exp[keep]
    ex *= obs.sum() / ex.sum()
    assert chisquare(obs, ex).pvalue > 1e-4


def test_channel_alias_cells_sample_the_row(orc):
    """photon channels are drawn with Walker's alias method over the instruction's cumulative row (np.random.choice's
    distribution, s1.py:154-158 / s2.py:673-677): the cells encode the row's probabilities to 2^-32 per channel -- also with
    channels of probability zero (turned-off PMTs) and a very uneven row -- and draws follow them (chi-square)"""
    from scipy.stats import chisquare
    rng = np.random.default_rng(6)
    nch = 494
    for kind in ('uniform', 'uneven'):
        p = np.ones(nch) if kind == 'uniform' else rng.gamma(0.3, 1.0, nch)
        p[[3, 260, 493]] = 0.0
        p /= p.sum()
        cdf = np.cumsum(p); cdf /= cdf[-1]
        pa = orc.chan_alias_pmf(cdf)
        pc = np.diff(cdf, prepend=0.0)
        assert abs(pa.sum() - 1) < 1e-12 and np.abs(pa - pc).max() < 2.0 ** -31
        assert np.all(pa[[3, 260, 493]] < 2.0 ** -31)
        n = 2_000_000
        ch = orc.sample_channels(cdf, n)
        cnt = np.bincount(ch, minlength=nch)
        assert cnt[[3, 260, 493]].sum() == 0
        keep = pc * n > 20
        ex = pc[keep] * n
        assert chisquare(cnt[keep], ex * cnt[keep].sum() / ex.sum()).pvalue > 1e-4
