"""End-to-end statistical pin of the HIP path on the REFERENCE: tests/golden/chain_stats.npz holds, for 300 S1 and 60 S2
instructions run through the reference's RawData (make_golden.py: fixture_chain_stats), every truth row and per digitise
window the number of ZLE pulses, their samples, their area and the window width.  The same instructions through
wfsim_amd.ChunkRawRecords (more of them: the reference sample is the noisy one) must give the same distributions:
means within 5 standard errors, spreads within 25 %, a two-sample KS test on the photon counts.  The random streams
differ by construction (numpy MT19937 vs Philox), everything deterministic is pinned bit for bit elsewhere."""
import numpy as np
import pytest
from scipy.stats import ks_2samp

import wfsim_amd
from tests.helpers import golden
from wfsim_amd.config import xenonnt_test_config
from wfsim_amd.dtypes import instruction_dtype

pytestmark = pytest.mark.gpu
MS = 1_000_000
TRUTH = ['n_photon', 'n_pe', 'n_photon_trigger', 'n_pe_trigger', 'raw_area', 'raw_area_trigger', 'n_photon_bottom', 'n_electron']


def _run(kind, n):
    ins = np.zeros(n, dtype=instruction_dtype)
    ins['type'], ins['amp'], ins['z'] = (1, 1667, -50.0) if kind == 's1' else (2, 300, -10.0)
    ins['time'], ins['recoil'], ins['event_number'] = MS * (1 + np.arange(n)), 7, np.arange(n)
    cfg = xenonnt_test_config(seed=4242 if kind == 's1' else 4343, chunk_size=100.0)
    chunks = list(wfsim_amd.ChunkRawRecords(cfg)(ins))
    rec = np.concatenate([c['raw_records'] for c in chunks])
    truth = np.concatenate([c['truth'] for c in chunks])
    truth = truth[np.argsort(truth['event_number'])]
    assert len(truth) == n
    # per digitise window (one per instruction, 1 ms apart): ZLE pulses, their samples, their area, the span they cover
    ev = ((rec['time'] + MS // 2) // MS - 1).astype(np.int64)        # (a window starts a little before its instruction)
    head = rec['record_i'] == 0
    n_zle = np.bincount(ev[head], minlength=n)
    samples = np.bincount(ev[head], weights=rec['pulse_length'][head], minlength=n)
    valid = np.arange(rec['data'].shape[1])[None, :] < rec['length'][:, None]
    area = np.bincount(ev, weights=((16000 - rec['data'].astype(np.int64)) * valid).sum(axis=1), minlength=n)
    return truth, dict(n_zle=n_zle, zle_samples=samples, zle_area=area)


def _close(name, a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    se = np.sqrt(a.var() / len(a) + b.var() / len(b))
    assert abs(a.mean() - b.mean()) <= 5 * se + 1e-9 * abs(b.mean()), (name, a.mean(), b.mean(), se)
    if b.std() > 0 and len(b) >= 50:
        assert 0.75 < (a.std() + 1e-12) / b.std() < 1.33, (name, a.std(), b.std())


@pytest.mark.parametrize('kind,n', [('s1', 3000), ('s2', 400)])
def test_full_chain_statistics_match_the_reference_runs(kind, n):
    d = golden('chain_stats.npz')
    truth, win = _run(kind, n)
    t_ins = MS * (1 + truth['event_number'].astype(np.int64))           # (the chunker overwrites 'time' with the first photon, :482)
    for f in TRUTH:
        _close(f, truth[f], d[f'{kind}_{f}'])
    # photon times relative to the instruction, their spread per signal, electron times
    for f in ['t_mean_photon', 't_first_photon', 't_last_photon'] + (['t_mean_electron'] if kind == 's2' else []):
        _close(f, truth[f] - t_ins, d[f'{kind}_{f}'] - d[f'{kind}_time'])
    _close('t_sigma_photon', truth['t_sigma_photon'], d[f'{kind}_t_sigma_photon'])
    _close('endtime', truth['endtime'] - t_ins, d[f'{kind}_endtime'] - d[f'{kind}_time'])
    assert ks_2samp(truth['n_photon'], d[f'{kind}_n_photon']).pvalue > 1e-3
    assert ks_2samp(truth['n_pe'], d[f'{kind}_n_pe']).pvalue > 1e-3
    for f in ('n_zle', 'zle_samples', 'zle_area'):
        _close(f, win[f], d[f'{kind}_{f}'])
