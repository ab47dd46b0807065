"""HIP path against the golden vectors of the reference and against the CPU oracle (MI355X only).

Everything here goes through the C ABI (wfsim_amd.engine -> libwfsim_amd.so).  Photons are injected at the
stage boundary Pulse.__call__ so that all results are deterministic:
  * pulse bounds, photon counts: exact
  * tile currents (f64): bit-exact with fused_multiply_add off (tolerance 4 ulp of the tile maximum only for tiles with >= 3
    photons in one ns, whose merge order the reference leaves to numpy's unstable argsort); with the switch on (the default: one
    rounding per template * gain term instead of two) within FMA_CURRENT_TOL_ULP of the tile maximum (tests/helpers.py)
  * digitised rows, ZLE intervals, records: exact in BOTH modes (the north-star bar is 1 ADC count; we get 0)
"""
import numpy as np
import pytest

from tests.helpers import (golden, make_engine, make_oracle, replay_chain_on_engine, replay_chain_on_oracle,
                           canonical_intervals, ap_tables_from_golden, with_fma, assert_currents_close)
from wfsim_amd.config import xenonnt_test_config

pytestmark = pytest.mark.gpu


def _nonempty_groups(eng):
    g = eng.groups()
    keep = np.where(g['right'] >= g['left'])[0]
    return g, keep


def _check_chain(name, config, force_dense=False, noise_offsets=False):
    """both arithmetic modes of add_current; returns the engine of the default one (fused multiply-adds)"""
    _check_chain_mode(name, with_fma(config, False), force_dense, noise_offsets)
    return _check_chain_mode(name, with_fma(config, True), force_dense, noise_offsets)


def _check_chain_mode(name, config, force_dense=False, noise_offsets=False):
    d = golden(name)
    eng = make_engine(config)
    counts = replay_chain_on_engine(eng, d, config, force_dense=force_dense)
    if noise_offsets:
        # the reference drew its noise start index from numpy's stream: inject it per digitise window
        g = eng.groups()
        ix = np.full(len(g['left']), -1, dtype=np.int64)
        ix[g['right'] >= g['left']] = d['dg_ix_rand']
        eng.set_noise_offsets(ix)
        counts = replay_chain_on_engine(eng, d, config, force_dense=force_dense)
    # ---- pulses
    p = eng.pulses(currents=True)
    order = np.lexsort((p['channel'], p['set']))
    assert np.array_equal(p['channel'][order], d['pl_ch'])
    assert np.array_equal(p['left'][order], d['pl_left'])
    assert np.array_equal(p['right'][order], d['pl_right'])
    assert np.array_equal(p['n_photons'][order], d['pl_photons'])
    if 'pl_current' in d:
        for j, k in enumerate(order):
            cur = p['current'][p['cur_off'][k]:p['cur_off'][k] + p['right'][k] - p['left'][k] + 1]
            ref = d['pl_current'][d['pl_cur_off'][j]:d['pl_cur_off'][j + 1]]
            if config['fused_multiply_add']:
                assert_currents_close(cur, ref, f'tile {j}')
                continue
            assert np.array_equal(cur, ref), f'tile {j}: max diff {np.abs(cur - ref).max()}'
    # ---- digitise windows
    g, keep = _nonempty_groups(eng)
    assert np.array_equal(g['left'][keep], d['dg_left'])
    assert np.array_equal(g['right'][keep], d['dg_right'])
    gmap = {int(gi): j for j, gi in enumerate(keep)}
    # ---- rows
    r = eng.rows()
    rows = sorted((gmap[int(r['group'][k])], int(r['channel'][k]), int(r['left'][k]), int(r['right'][k]),
                   r['data'][r['data_off'][k]:r['data_off'][k] + r['right'][k] - r['left'][k] + 1].astype(np.int64).tobytes())
                  for k in range(len(r['group'])))
    dig_of_row = np.repeat(np.arange(len(d['dg_left'])), np.diff(d['dg_row_off']))
    ref_rows = sorted((int(dig_of_row[k]), int(d['row_ch'][k]), int(d['row_left'][k]), int(d['row_right'][k]),
                       d['row_data'][d['row_data_off'][k]:d['row_data_off'][k + 1]].astype(np.int64).tobytes())
                      for k in range(len(d['row_ch'])))
    if max(x[1] for x in rows) < 500:
        # int(high_energy_deamplification_factor) == 0 and no HE noise: the reference keeps flat baseline rows for the
        # high-energy channels (rawdata.py:242-249) that can never produce a ZLE interval; the HIP path skips them
        flat = np.full(1, int(config['digitizer_reference_baseline']), dtype=np.int64)
        assert all(np.all(np.frombuffer(x[4], dtype=np.int64) == flat) for x in ref_rows if x[1] >= 500)
        ref_rows = [x for x in ref_rows if x[1] < 500]
    assert len(rows) == len(ref_rows)
    for a, b in zip(rows, ref_rows):
        assert a[:4] == b[:4]
        assert a[4] == b[4], f'row {a[:4]} differs'
    # ---- ZLE
    z = eng.intervals()
    got = canonical_intervals([gmap[int(x)] for x in z['group']], z['channel'], z['left'], z['right'], z['data_off'], z['data'])
    ref = canonical_intervals(d['zle_digit'], d['zle_ch'], d['zle_left'], d['zle_right'], d['zle_data_off'], d['zle_data'])
    assert len(got) == len(ref)
    assert got == ref
    # ---- records: yield order of the reference = (window, channel, interval)
    rec = eng.records()
    n_expected = np.ceil((d['zle_right'] - d['zle_left'] + 1) / 110).astype(int)
    assert len(rec) == n_expected.sum()
    first = np.concatenate([[0], np.cumsum(n_expected)[:-1]])
    dt = int(config.get('sample_duration', 10))
    assert np.array_equal(rec['time'][first], dt * d['zle_left'])
    assert np.array_equal(rec['channel'][first], d['zle_ch'])
    assert np.array_equal(rec['pulse_length'][first], d['zle_right'] - d['zle_left'] + 1)
    assert np.all(rec['dt'] == dt) and np.all(rec['baseline'] == 0)
    return d, eng


@pytest.mark.parametrize('force_dense', [False, True])
@pytest.mark.parametrize('name', ['chain_s1.npz', 'chain_s2.npz'])
def test_chain_vs_reference_golden(name, force_dense):
    d, eng = _check_chain(name, xenonnt_test_config(), force_dense)
    # truth accumulators
    acc, ts = eng.truth()
    for j, f in enumerate(['n_photon', 'n_pe', 'n_photon_trigger', 'n_pe_trigger']):
        assert np.array_equal(acc[:, j], d['call_truth_' + f]), f
        assert np.array_equal(acc[:, 6 + j], d['call_truth_' + f + '_bottom']), f
    for j, f in [(4, 'raw_area'), (5, 'raw_area_trigger')]:
        assert np.allclose(acc[:, j], d['call_truth_' + f], rtol=1e-12, atol=0)
        assert np.allclose(acc[:, 6 + j], d['call_truth_' + f + '_bottom'], rtol=1e-12, atol=0)


def test_chain_he_channels():
    _check_chain('chain_he.npz', xenonnt_test_config(high_energy_deamplification_factor=20))


def test_chain_pmt_afterpulse_pulses():
    _check_chain('chain_pmt_ap.npz', xenonnt_test_config())


@pytest.mark.parametrize('force_dense', [False, True])
def test_chain_nondefault_parameters(force_dense):
    """chain F: non-default trigger window / stored samples / thresholds / baseline / rext, non-uniform gains, dead PMTs"""
    from tests.helpers import params_chain_config
    _check_chain('chain_params.npz', params_chain_config(), force_dense=force_dense)


def test_chain_other_digitiser_geometry():
    """chain I (made by the reference with sample_duration 5 ns and 3 + 37 template samples): every tile goes through
    k_pulse_generic; currents bit-exact, windows, rows, ZLE, records exact; the records equal the oracle's bytes; and the
    photon generators (block generator, generic geometry) agree with the oracle photon by photon"""
    from tests.helpers import geometry_chain_config
    cfg = geometry_chain_config()
    d, eng = _check_chain('chain_geometry.npz', cfg)
    orc = make_oracle(cfg)
    replay_chain_on_oracle(orc, d)
    assert eng.records().tobytes() == orc.pack_records().tobytes()
    from tests.test_gpu_generation import _instructions, _run_both, _compare, MS
    rows = [dict(type=1, time=MS * (i + 1), x=3, y=-2, z=-30, amp=2500) for i in range(3)]
    rows += [dict(type=2, time=MS * (i + 1), x=3, y=-2, z=-30, amp=120) for i in range(3)]
    _compare(*_run_both(cfg, _instructions(rows)))


def test_chain_run_sets_and_electron_afterpulses():
    """golden chains G (save_full_truth=False: several instructions per Pulse call) and H (electron afterpulses: type-4
    calls, windows closed right after a cluster of secondaries, rawdata.py:148-149): clusters, window-rule keys and pulse
    sets from the host scheduler, photons of the reference's calls injected -> windows, rows, ZLE, records bit-exact"""
    _check_chain('chain_runsets.npz', xenonnt_test_config(save_full_truth=False))
    _check_chain('chain_ele_ap.npz', xenonnt_test_config())


def test_chain_noise():
    cfg = xenonnt_test_config(enable_noise=True, noise_data=golden('noise.npz')['noise'])
    _check_chain('chain_noise.npz', cfg, noise_offsets=True)


def test_records_match_oracle_bytes():
    cfg = xenonnt_test_config()
    d = golden('chain_s2.npz')
    eng = make_engine(cfg)
    replay_chain_on_engine(eng, d, cfg, debug=False)
    orc = make_oracle(cfg)
    replay_chain_on_oracle(orc, d)
    assert eng.records().tobytes() == orc.pack_records().tobytes()
