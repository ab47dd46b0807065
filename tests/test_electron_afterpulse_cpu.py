"""The electron-afterpulse generators (wfsim_amd/electron_afterpulse.py: plan_secondaries) against draws of the REFERENCE's
generators (afterpulse.py:29-92 photo-ionisation electrons, :107-131 gate electrons): tests/golden/ele_ap_draws.npz holds
600 calls of each for a parent S2 with 4000 detected photons (make_golden.py: fixture_ele_ap_generators).  Different
random streams (numpy's global MT19937 there, a Philox generator keyed by (seed, parent, kind) here), same distributions:
instructions and electrons per call, the diffusion-wide drift-time grid, electrons per instruction, positions, which
photon gives the time zero.  CPU only."""
import numpy as np
from scipy.stats import ks_2samp

from tests.helpers import golden
from wfsim_amd import electron_afterpulse as ea
from wfsim_amd.config import xenonnt_test_config
from wfsim_amd.dtypes import instruction_dtype


def _draws(kind, n_calls=1500):
    d = golden('ele_ap_draws.npz')
    cfg = xenonnt_test_config(enable_electron_afterpulses=True, enable_gate_afterpulses=True, photoelectric_p=float(d['photoelectric_p']),
                              uniform_to_ele_ap=(d['histogram'], d['bin_edges']), seed=99)
    hist = ea.DelayHistogram(d['histogram'], d['bin_edges'])
    grid = ea.coarse_delay_grid(hist, cfg)
    parent = np.zeros(1, dtype=instruction_dtype)
    parent['type'], parent['time'], parent['z'], parent['amp'] = 2, 5_000_000, -30.0, 200
    out = dict(n_ins=[], n_el=[], delay=[], amp=[], r2=[], pick=[])
    for gid in range(n_calls):
        for k, pick, delay, cnt, x, y in ea.plan_secondaries(parent, gid, int(d['n_photons']), cfg, hist, grid):
            if k != kind:
                continue
            out['n_ins'].append(len(pick)); out['n_el'].append(int(np.sum(cnt)))
            out['delay'].append(delay); out['amp'].append(cnt); out['r2'].append(np.asarray(x, float) ** 2 + np.asarray(y, float) ** 2)
            out['pick'].append(pick)
    return d, {k: (np.concatenate(v) if k not in ('n_ins', 'n_el') else np.array(v)) for k, v in out.items()}, cfg


def _means_agree(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return abs(a.mean() - b.mean()) <= 5 * np.sqrt(a.var() / len(a) + b.var() / len(b)) + 1e-12


def test_photoionisation_electrons_like_the_reference():
    d, o, cfg = _draws(4)
    assert len(o['n_ins']) == 1500
    for f in ('n_ins', 'n_el'):
        assert _means_agree(o[f], d['pi_' + f]) and ks_2samp(o[f], d['pi_' + f]).pvalue > 1e-3, f
    # the delays sit on the diffusion-wide grid of afterpulse.py:61-71 and follow the histogram
    assert ks_2samp(o['delay'], d['pi_delay']).pvalue > 1e-3
    assert _means_agree(o['amp'], d['pi_amp'])
    assert ks_2samp(o['r2'], d['pi_r2']).pvalue > 1e-3 and o['r2'].max() <= cfg['tpc_radius'] ** 2 * (1 + 1e-6)
    assert ks_2samp(o['pick'], d['pi_pick']).pvalue > 1e-3 and o['pick'].min() >= 0 and o['pick'].max() < int(d['n_photons'])


def test_gate_electrons_like_the_reference():
    d, o, cfg = _draws(6)
    for f in ('n_ins', 'n_el'):
        assert _means_agree(o[f], d['pe_' + f]) and ks_2samp(o[f], d['pe_' + f]).pvalue > 1e-3, f
    assert ks_2samp(o['delay'], d['pe_delay']).pvalue > 1e-3 and o['delay'].min() >= 0          # clipped normal, afterpulse.py:113-116
    assert np.all(o['amp'] == 1) and np.all(d['pe_amp'] == 1)
    assert ks_2samp(o['r2'], d['pe_r2']).pvalue > 1e-3
    assert ks_2samp(o['pick'], d['pe_pick']).pvalue > 1e-3


def test_the_delay_grid_is_the_references():
    """the reference's instructions carry z = -delay * v on its coarse grid: every delay in the fixture is one of our grid points"""
    d = golden('ele_ap_draws.npz')
    cfg = xenonnt_test_config()
    grid = ea.coarse_delay_grid(ea.DelayHistogram(d['histogram'], d['bin_edges']), cfg)
    # (z is a float32 field: the round trip through it costs ~1e-7 relative)
    nearest = grid[np.clip(np.searchsorted(grid, d['pi_delay']), 1, len(grid) - 1) - 0]
    lower = grid[np.clip(np.searchsorted(grid, d['pi_delay']), 1, len(grid) - 1) - 1]
    err = np.minimum(np.abs(nearest - d['pi_delay']), np.abs(lower - d['pi_delay']))
    assert np.all(err <= 2e-7 * np.abs(d['pi_delay']) + 1e-3)
