"""The CPU oracle against vectors produced by running the reference (tests/golden/make_golden.py).

Bit-exact for everything downstream of the photon arrays: currents (f64), pulse bounds, digitised rows, masks,
ZLE tuples, truth accumulators.  CPU only.
"""
import numpy as np
import pytest

from oracle import oracle as O
from tests.helpers import (golden, make_oracle, replay_chain_on_oracle, host_tables, ap_tables_from_golden, with_fma,
                           assert_currents_close)
from wfsim_amd.config import xenonnt_test_config


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    assert [hex(x) for x in O.philox([0, 0, 0, 0], [0, 0])] == ['0x6627e8d5', '0xe169c58d', '0xbc57ac4c', '0x9b00dbd8']
    f = 0xffffffff
    assert [hex(x) for x in O.philox([f, f, f, f], [f, f])] == ['0x408f276d', '0x41c83b0e', '0xa20bc7c6', '0x6d5451fd']
    assert [hex(x) for x in O.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0])] == \
        ['0xd16cfe09', '0x94fdcceb', '0x5001e420', '0x24126ea1']


def test_tables_match_reference():
    g = golden('tables.npz')
    t = host_tables(xenonnt_test_config())
    assert np.array_equal(t['templates'], g['templates'])
    assert np.array_equal(t['spe'][0], g['spe_table_row_full'])
    assert t['spe'].shape == (1, 2001)


def test_add_current_bit_exact():
    g = golden('add_current.npz')
    T = golden('tables.npz')['templates']
    for i in range(int(g['n'])):
        cur = O.Oracle.add_current(g[f't{i}'], g[f'g{i}'], int(g[f'left{i}']), 10, T, len(g[f'cur{i}']))
        ref = g[f'cur{i}']
        _, counts = np.unique(g[f't{i}'], return_counts=True)
        if counts.max() <= 2 or i == 1:
            assert np.array_equal(cur, ref), f'case {i}: max diff {np.abs(cur - ref).max()}'
        else:
            # >= 3 photons in one ns: the reference sums their gains in the order of numpy's unstable argsort
            # (pulse.py:297), which is unspecified; only the last bits of the merged gain can differ.
            assert np.all(np.abs(cur - ref) <= 4 * np.spacing(np.abs(ref).max())), f'case {i}'
            assert np.array_equal(np.around(cur * 5.8e-4), np.around(ref * 5.8e-4))


def test_add_current_fused():
    """the same vectors with fused multiply-adds (wfs_config.fma): within FMA_CURRENT_TOL_ULP of the reference's currents, the
    per-pulse rounded ADC values equal"""
    g = golden('add_current.npz')
    T = golden('tables.npz')['templates']
    worst = 0.0
    for i in range(int(g['n'])):
        cur = O.Oracle.add_current(g[f't{i}'], g[f'g{i}'], int(g[f'left{i}']), 10, T, len(g[f'cur{i}']), fma=True)
        ref = g[f'cur{i}']
        assert_currents_close(cur, ref, f'case {i}')
        worst = max(worst, np.abs(cur - ref).max() / np.spacing(np.abs(ref).max()))
        assert np.array_equal(np.around(cur * 5.8e-4), np.around(ref * 5.8e-4))
    assert worst > 0          # (the two forms do differ: the switch reaches the arithmetic)


def _check_chain(name, config, ap=None, noise_override=False):
    """The golden chain `name` on the oracle in both arithmetic modes: the reference's separately rounded product and sum (currents
    bit-exact; the session of this mode is returned) and fused multiply-adds (the default of the HIP path: currents within
    FMA_CURRENT_TOL_ULP of the reference, digitised rows and ZLE tuples EQUAL)."""
    d = golden(name)
    for fma in (True, False):
        orc = make_oracle(with_fma(config, fma), ap)
        if noise_override:
            orc.set_noise_override(d['dg_ix_rand'])
        r = replay_chain_on_oracle(orc, d)
        assert np.array_equal(r['pl_ch'], d['pl_ch'])
        assert np.array_equal(r['pl_left'], d['pl_left'])
        assert np.array_equal(r['pl_right'], d['pl_right'])
        assert np.array_equal(r['pl_nph'], d['pl_photons'])
        if 'pl_current' in d:
            if fma:
                assert_currents_close(r['cur'], d['pl_current'], name)
            else:
                assert np.array_equal(r['cur'], d['pl_current'])
        if fma:
            assert np.array_equal(r['row_data'], d['row_data'])
            for k in ['ch', 'left', 'right', 'data_off', 'data']:
                assert np.array_equal(r['zl_' + k], d['zle_' + k]), k
    assert np.array_equal(r['dg_left'], d['dg_left'])
    assert np.array_equal(r['dg_right'], d['dg_right'])
    assert np.array_equal(r['row_ch'], d['row_ch'])
    assert np.array_equal(r['row_left'], d['row_left'])
    assert np.array_equal(r['row_right'], d['row_right'])
    return d, r, orc


@pytest.mark.parametrize('name', ['chain_s1.npz', 'chain_s2.npz'])
def test_chain_replay_bit_exact(name):
    d, r, orc = _check_chain(name, xenonnt_test_config())
    assert np.array_equal(r['row_data'], d['row_data'])
    for k in ['ch', 'left', 'right', 'data_off', 'data', 'digit']:
        assert np.array_equal(r['zl_' + k], d['zle_' + k]), k
    # truth accumulators (pulse.py:229-271)
    tr = r['truth'].reshape(-1, 12)
    names = ['n_photon', 'n_pe', 'n_photon_trigger', 'n_pe_trigger', 'raw_area', 'raw_area_trigger']
    for j, f in enumerate(names):
        assert np.array_equal(tr[:, j], d['call_truth_' + f].astype(np.float64)), f
        assert np.array_equal(tr[:, 6 + j], d['call_truth_' + f + '_bottom'].astype(np.float64)), f + '_bottom'


def test_chain_he_channels():
    d, r, orc = _check_chain('chain_he.npz', xenonnt_test_config(high_energy_deamplification_factor=20))
    assert r['zl_ch'].max() > 500
    assert np.array_equal(r['row_data'], d['row_data'])
    for k in ['ch', 'left', 'right', 'data']:
        assert np.array_equal(r['zl_' + k], d['zle_' + k]), k


def test_chain_nondefault_parameters():
    """chain F: trigger window 30, store 20/70 samples, zle_threshold 25 + special thresholds, baseline 15000,
    rext 30000, non-uniform gains with three turned-off PMTs"""
    from tests.helpers import params_chain_config
    cfg = params_chain_config()
    d, r, orc = _check_chain('chain_params.npz', cfg)
    assert not np.isin(d['ph_ch'], cfg['turned_off_pmts']).any()          # s1.py / s2.py zero the pattern of turned-off PMTs
    assert len(d['dg_left']) == 5                                         # rext 30000: the S1 50 us later gets a window of its own
    assert np.array_equal(r['row_data'], d['row_data'])
    for k in ['ch', 'left', 'right', 'data']:
        assert np.array_equal(r['zl_' + k], d['zle_' + k]), k


def test_chain_other_digitiser_geometry():
    """chain I: sample_duration 5 ns, templates of 3 + 37 samples (pulse.py:146-187), other stored / trigger windows: currents
    bit-exact, rows and ZLE exact"""
    from tests.helpers import geometry_chain_config
    cfg = geometry_chain_config()
    from wfsim_amd.config import kernel_params
    assert (kernel_params(cfg)['dt'], kernel_params(cfg)['tlen']) == (5, 40)
    d, r, orc = _check_chain('chain_geometry.npz', cfg)
    assert np.array_equal(r['row_data'], d['row_data'])
    for k in ['ch', 'left', 'right', 'data']:
        assert np.array_equal(r['zl_' + k], d['zle_' + k]), k


def test_chain_pmt_afterpulse_pulses():
    # afterpulse photons are injected with their pre-assigned gains (Pulse.__call__ branch pulse.py:105-107)
    d, r, orc = _check_chain('chain_pmt_ap.npz', xenonnt_test_config())
    assert np.array_equal(r['row_data'], d['row_data'])
    assert np.array_equal(r['zl_data'], d['zle_data'])


def test_record_packing_layout():
    from wfsim_amd.dtypes import raw_record_dtype
    d, r, orc = _check_chain('chain_s2.npz', xenonnt_test_config())
    rec = orc.pack_records().view(np.dtype(raw_record_dtype()))
    n_expected = np.ceil((d['zle_right'] - d['zle_left'] + 1) / 110).astype(int)
    assert len(rec) == n_expected.sum()
    first = np.concatenate([[0], np.cumsum(n_expected)[:-1]])
    assert np.array_equal(rec['time'][first], 10 * d['zle_left'])
    assert np.array_equal(rec['channel'][first], d['zle_ch'])
    assert np.array_equal(rec['pulse_length'][first], d['zle_right'] - d['zle_left'] + 1)
    assert np.all(rec['dt'] == 10) and np.all(rec['baseline'] == 0)
    # data of the first interval
    k = 0
    n = int(d['zle_data_off'][1])
    got = np.concatenate([rec['data'][i][:rec['length'][i]] for i in range(n_expected[k])])
    assert np.array_equal(got, d['zle_data'][:n])
    assert rec['length'][n_expected[0] - 1] == n - 110 * (n_expected[0] - 1)


def test_chain_noise():
    cfg = xenonnt_test_config(enable_noise=True, noise_data=golden('noise.npz')['noise'])
    d, r, orc = _check_chain('chain_noise.npz', cfg, noise_override=True)
    assert np.array_equal(r['row_data'], d['row_data'])
    for k in ['ch', 'left', 'right', 'data']:
        assert np.array_equal(r['zl_' + k], d['zle_' + k]), k


def test_run_sets_grouping_matches_reference():
    """save_full_truth=False (rawdata.py:106-127): golden chain G.  The host scheduler's run sets and the oracle's own
    scheduler group the instructions as the reference did (one Pulse call and one truth row per group); the merged
    calls replay bit-exact."""
    from wfsim_amd.scheduler import schedule, run_sets
    from wfsim_amd.physics import instruction_params
    from wfsim_amd.resource import Resource
    cfg = xenonnt_test_config(save_full_truth=False)
    d, r, orc = _check_chain('chain_runsets.npz', cfg)
    assert np.array_equal(r['row_data'], d['row_data'])
    ins = d['instructions']
    order, key, cluster = schedule(ins, cfg)
    s_ins = ins[order]
    rs, n_sets = run_sets(s_ins, key, cluster, cfg)
    assert n_sets == len(d['call_kind']) == 6
    ref_truth = d['truth']
    for q in range(n_sets):
        m = np.where(rs == q)[0]
        row = ref_truth[q]                                   # the reference fills truth rows in processing order
        assert s_ins['type'][m[0]] == d['call_kind'][q] == row['type']
        assert row['amp'] == s_ins['amp'][m].sum()
        assert np.isclose(row['x'], s_ins['x'][m].mean()) and np.isclose(row['z'], s_ins['z'][m].mean())
        assert row['time'] == s_ins['time'][m[0]] and row['event_number'] == s_ins['event_number'][m[0]]
    assert [int((rs == q).sum()) for q in range(n_sets)] == [3, 1, 2, 1, 1, 1]
    # with save_full_truth every instruction is a set of its own, numbered in processing order
    rs1, n1 = run_sets(s_ins, key, cluster, xenonnt_test_config())
    assert n1 == len(ins) and sorted(rs1.tolist()) == list(range(len(ins)))
    # the oracle's own scheduler makes the same calls
    o2 = make_oracle(cfg)
    o2.simulate(s_ins, order.astype(np.uint32), instruction_params(s_ins, cfg, Resource(cfg)))
    res = o2.results()
    assert np.array_equal(res['call_kind'], d['call_kind'])
    cfg1 = xenonnt_test_config(tile_local_generation=False)       # (the per-electron generator, as grouped calls always use it)
    o3 = make_oracle(cfg1)
    o3.simulate(s_ins, order.astype(np.uint32), instruction_params(s_ins, cfg1, Resource(cfg1)))
    assert len(o3.results()['call_kind']) == len(ins)
    # grouping does not change the photons (same Philox coordinates), only which call they belong to
    assert len(res['ph_t']) == len(o3.results()['ph_t'])


def test_electron_afterpulse_feedback_schedule_matches_reference():
    """golden chain H (enable_electron_afterpulses, secondaries recorded from the reference): the host's replay of the
    scheduler feedback loop (rawdata.py:70-151) makes the same Pulse calls in the same order -- type-4 instructions are
    seen one pass after their parent, grouped per residual cluster -- and the recorded calls replay bit-exact"""
    from tests.helpers import chain_union
    from wfsim_amd.scheduler import feedback_schedule
    cfg = xenonnt_test_config()
    d, r, orc = _check_chain('chain_ele_ap.npz', cfg)
    assert np.array_equal(r['row_data'], d['row_data'])
    for k in ['ch', 'left', 'right', 'data']:
        assert np.array_equal(r['zl_' + k], d['zle_' + k]), k
    ins, parent = chain_union(d)
    assert (parent >= 0).sum() == len(d['secondaries']) > 10 and np.all(d['secondaries']['type'] == 4)
    order, key, cluster, rs = feedback_schedule(ins, parent, cfg)
    s_ins = ins[order]
    n_sets = int(rs.max()) + 1
    kinds = np.array([s_ins['type'][np.where(rs == q)[0][0]] for q in range(n_sets)])
    assert np.array_equal(kinds, d['call_kind'])
    # the reference's truth rows of the calls that made photons carry the summed amp of their instructions
    amps = np.array([s_ins['amp'][rs == q].sum() for q in range(n_sets)])
    with_photons = np.diff(d['call_ph_off']) > 0
    assert np.array_equal(amps[with_photons], d['truth']['amp'])
    assert np.all(np.diff(cluster) >= 0) and np.all(np.diff(rs) >= 0)
