"""The synthetic workload builders (wfsim_amd/workloads.py): the tables they make are the golden fixtures' (same seeds, same
constructions), and two calls build identical workloads."""
import numpy as np

from tests.helpers import ap_tables_from_golden, golden
from wfsim_amd import workloads as W


def test_synthetic_tables_are_the_golden_ones():
    a, b = W.synthetic_afterpulse_tables(), ap_tables_from_golden()
    assert sorted(a) == sorted(b)
    for name in b:
        for q in b[name]:
            assert np.array_equal(np.asarray(a[name][q]), np.asarray(b[name][q])), (name, q)
    assert np.array_equal(W.synthetic_noise(), golden('noise.npz')['noise'])


def test_builders_are_deterministic():
    assert np.array_equal(W.mixed_batch(200, 7), W.mixed_batch(200, 7))
    assert np.array_equal(W.s2_batch(50, 3, spread_xy=True), W.s2_batch(50, 3, spread_xy=True))
    a, b = W.optical_instructions(500, 1000.0, 3), W.optical_instructions(500, 1000.0, 3)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert W.s1_batch(10)['amp'][0] == 1667 and np.all(W.s2_batch(10)['amp'] == 10_000)


def test_synthetic_s2_pattern_map_is_a_probability_pattern():
    m = W.synthetic_s2_pattern_map()
    p = np.asarray(m['map'], dtype=np.float64)
    assert p.shape == (61, 61, 494) and np.allclose(p.sum(axis=-1), 1.0, atol=1e-5)
    top = p[..., :W.N_TOP].sum(axis=-1)
    assert np.allclose(top, 0.75, atol=1e-5)
    assert 0.02 < p[30, 30].max() < 0.15          # the PMT above the event: a few percent of the light


def test_reference_default_switches_of_the_side_bench():
    """bench_config(reference_defaults=True): electron afterpulses on (rawdata.py:194) with a delay histogram whose integral is the
    probability per photon, garfield luminescence with the fixture's table layout; the Resource and the kernel parameters accept both"""
    from wfsim_amd.config import kernel_params
    from wfsim_amd.resource import Resource
    hist, edges = W.synthetic_electron_afterpulses()
    assert len(edges) == len(hist) + 1 and np.isclose(hist.sum(), 3e-3) and np.all(np.diff(hist) < 0) and edges[-1] == 150e3
    assert np.isclose(W.synthetic_electron_afterpulses(total=3e-5)[0].sum(), 3e-5)
    g = W.synthetic_garfield_table()
    d = golden('dists_models.npz')
    assert np.array_equal(g['t'], d['garfield_t']) and np.array_equal(g['x'], d['garfield_x'])
    cfg = W.bench_config(3, reference_defaults=True)
    assert cfg['enable_electron_afterpulses'] and cfg['s2_luminescence_model'] == 'garfield'
    res = Resource(cfg)
    assert res.s2_luminescence['t'].shape == (11, 4000) and np.isclose(np.sum(res.uniform_to_ele_ap.histogram), 3e-3)
    assert kernel_params(cfg)['tile_gen'] == 1 and kernel_params(cfg)['row_resident'] == 2
    assert kernel_params(dict(cfg, row_resident=False))['row_resident'] == 0 and kernel_params(dict(cfg, row_resident=True))['row_resident'] == 1
