"""RNG spec v9, tile-local generation (device: wfs_tilegen.h, oracle: gen_s2 / fuse_eligible): a primary S2
with one secondary gain for all electrons draws, per (instruction, channel) tile, Poisson(n_surviving * g * p_ch) photons and a
uniform surviving electron for each.  By Poisson splitting that is the joint distribution of the reference's draws
(np.random.poisson per electron, s2.py:308, then np.random.choice of the channel, s2.py:673).  Checked here on the CPU oracle,
against the reference's own draws where a fixture holds them (tests/golden/chain_stats.npz, dists.npz) and against the exact laws:
instruction totals, channel counts under an uneven pattern, photons per electron and their independence, and that the
per-electron generator (tile_local_generation=False) gives the same distributions."""
import numpy as np
import pytest
from scipy.stats import chisquare, ks_2samp, poisson

from tests.helpers import golden, make_oracle
from wfsim_amd.config import kernel_params, xenonnt_test_config
from wfsim_amd.dtypes import instruction_dtype
from wfsim_amd.physics import instruction_params
from wfsim_amd.resource import Resource

MS = 1_000_000


def _s2(n, amp, z=-10.0, **cfg_kw):
    cfg_kw.setdefault('tile_local_min_photons', 0)       # (every S2 size through the tile path: by default only tiles worth a workgroup take it)
    cfg = xenonnt_test_config(**cfg_kw)
    ins = np.zeros(n, dtype=instruction_dtype)
    ins['type'], ins['amp'], ins['z'], ins['recoil'] = 2, amp, z, 7
    ins['time'], ins['event_number'] = MS * (1 + np.arange(n)), np.arange(n)
    return cfg, ins


def _simulate(cfg, ins):
    orc = make_oracle(cfg)
    orc.simulate(ins, np.arange(len(ins), dtype=np.uint32), instruction_params(ins, cfg, Resource(cfg)))
    return orc.results()


def test_switch_and_eligibility_rule():
    assert kernel_params(xenonnt_test_config())['tile_gen'] == 1
    assert kernel_params(xenonnt_test_config(tile_local_generation=False))['tile_gen'] == 0
    assert kernel_params(xenonnt_test_config(save_full_truth=False))['tile_gen'] == 1        # (per instruction: alone in its Pulse call or not)
    assert kernel_params(xenonnt_test_config(enable_electron_afterpulses=True))['tile_gen'] == 1       # (the pre-pass is served from the tiles' counters)
    # with and without the switch the photons differ (other streams) but not their number on average; since RNG spec v11 a tile of any
    # size is generated tile by tile (the device makes tiles above the 2048 photon registers of a workgroup in passes): a bright S2 too
    cfg, ins = _s2(2, 11500, s2_secondary_sc_gain=100.0, seed=3)        # 11500 e- x 82 photons / 494 PMTs = ~1900 per tile
    a = _simulate(cfg, ins)
    b = _simulate(dict(cfg, tile_local_generation=False), ins)
    assert len(a['ph_t']) != len(b['ph_t']) and abs(len(a['ph_t']) - len(b['ph_t'])) < 6 * np.sqrt(len(b['ph_t']) * 100)      # (Poisson(100) per electron)
    cfg, ins = _s2(3, 300, seed=3)
    a, b = _simulate(cfg, ins), _simulate(dict(cfg, tile_local_generation=False), ins)
    assert len(a['ph_t']) != len(b['ph_t']) or not np.array_equal(a['ph_t'], b['ph_t'])
    # small tiles keep the per-electron generator by default (wfs_config.tile_gen_min: a workgroup per tile does not pay below ~60 photons)
    small = dict(cfg); small.pop('tile_local_min_photons')
    assert kernel_params(small)['tile_gen_min'] == 64
    e, f = _simulate(small, ins), _simulate(dict(small, tile_local_generation=False), ins)
    assert np.array_equal(e['ph_t'], f['ph_t'])
    # a gain spread needs the per-electron generator
    c = _simulate(dict(cfg, s2_gain_spread=2.0), ins)
    d = _simulate(dict(cfg, s2_gain_spread=2.0, tile_local_generation=False), ins)
    assert np.array_equal(c['ph_t'], d['ph_t'])


def test_instruction_totals_match_the_reference_runs():
    """photons per S2 of the reference's own runs (chain_stats.npz: 300 e- at z = -10 cm): the sum of the tiles' Poisson draws"""
    d = golden('chain_stats.npz')
    cfg, ins = _s2(700, 300, seed=71)
    r = _simulate(cfg, ins)
    n_ph = np.diff(r['call_ph_off'])
    ref = d['s2_n_photon']
    assert ks_2samp(n_ph, ref).pvalue > 1e-3
    se = np.sqrt(n_ph.var() / len(n_ph) + ref.var() / len(ref))
    assert abs(n_ph.mean() - ref.mean()) < 5 * se
    # given the surviving electrons the total is Poisson(n_e g): index of dispersion of (n_photon - g n_e) / sqrt(g n_e)
    n_e = np.diff(r['call_e_off'])
    g = float(instruction_params(ins[:1], cfg, Resource(cfg))['sc_gain'][0])
    z = (n_ph - g * n_e) / np.sqrt(g * n_e)
    assert abs(z.mean()) < 5 / np.sqrt(len(z)) and abs(z.var() - 1) < 5 * np.sqrt(2 / len(z))
    # the per-electron generator: same law
    r2 = _simulate(dict(cfg, tile_local_generation=False), ins[:300])
    assert ks_2samp(n_ph, np.diff(r2['call_ph_off'])).pvalue > 1e-3


def test_channel_counts_follow_an_uneven_pattern():
    """np.random.choice over the pattern row (s2.py:673) <-> independent Poisson(n_e g p_ch) tiles: chi-square of the channel
    totals, and the tiles of one channel over many instructions are Poisson distributed"""
    rng = np.random.default_rng(8)
    p = rng.gamma(0.5, 1.0, 494); p[[7, 300]] = 0.0; p /= p.sum()

    class Pattern:
        def __call__(self, xy, **kw):
            return np.tile(p, (len(xy), 1))
    cfg, ins = _s2(600, 200, seed=72)
    cfg = dict(cfg, s2_pattern_map=Pattern())
    r = _simulate(cfg, ins)
    cnt = np.bincount(r['ph_ch'], minlength=494)
    assert cnt[[7, 300]].sum() == 0
    keep = p * cnt.sum() > 30
    ex = p[keep] * cnt.sum()
    assert chisquare(cnt[keep], ex * cnt[keep].sum() / ex.sum()).pvalue > 1e-4
    # one bright channel, instruction by instruction: Poisson(n_e g p_ch)
    ch = int(np.argmax(p))
    n_e = np.diff(r['call_e_off'])
    g = float(instruction_params(ins[:1], cfg, Resource(cfg))['sc_gain'][0])
    per = np.array([np.count_nonzero(r['ph_ch'][a:b] == ch) for a, b in zip(r['call_ph_off'][:-1], r['call_ph_off'][1:])])
    z = (per - n_e * g * p[ch]) / np.sqrt(n_e * g * p[ch])
    assert abs(z.mean()) < 5 / np.sqrt(len(z)) and abs(z.var() - 1) < 6 * np.sqrt(2 / len(z))


def test_photons_per_electron_are_independent_poisson():
    """Two-electron S2s whose electrons arrive far apart: every photon can be told to its electron.  Per electron the number of
    photons is Poisson(g) (s2.py:308) -- mean, variance, chi-square against the exact pmf -- and the two electrons' counts are
    uncorrelated; the electron of a photon is uniform (a fair split)."""
    cfg, ins = _s2(12000, 2, z=-50.0, seed=73, s2_secondary_sc_gain=30.0)
    r = _simulate(cfg, ins)
    g = float(instruction_params(ins[:1], cfg, Resource(cfg))['sc_gain'][0])
    first, second = [], []
    e_off, p_off = r['call_e_off'], r['call_ph_off']
    for k in range(len(ins)):
        et = r['e_t'][e_off[k]:e_off[k + 1]]
        if len(et) != 2 or abs(int(et[1]) - int(et[0])) < 2000:      # delays beyond 1 us are rarer than 1e-3: the nearer electron is the photon's
            continue
        t = r['ph_t'][p_off[k]:p_off[k + 1]]
        near0 = np.abs(t - et[0]) < np.abs(t - et[1])
        first.append(int(near0.sum())); second.append(int((~near0).sum()))
    first, second = np.array(first), np.array(second)
    assert len(first) > 400
    both = np.concatenate([first, second])
    assert abs(both.mean() - g) < 5 * np.sqrt(g / len(both)) and abs(both.var() / g - 1) < 6 * np.sqrt(2 / len(both))
    vals, cnt = np.unique(both, return_counts=True)
    ex = poisson.pmf(vals, g) * len(both)
    keep = ex > 15
    assert chisquare(cnt[keep], ex[keep] * cnt[keep].sum() / ex[keep].sum()).pvalue > 1e-4
    assert abs(np.corrcoef(first, second)[0, 1]) < 5 / np.sqrt(len(first))
    assert abs(first.sum() / both.sum() - 0.5) < 5 * 0.5 / np.sqrt(both.sum())


def test_photon_times_match_the_reference_draws():
    """all delay terms + electron arrival against the reference's S2 photons (dists.npz: s2_full, 2e4 electrons at z = -10 cm)"""
    from tests.test_oracle_distributions import _ks, _ks_limit, _moments
    d = golden('dists.npz')
    cfg, ins = _s2(30, 1500, seed=74)
    r = _simulate(cfg, ins)
    t = r['ph_t'] - np.repeat(ins['time'], np.diff(r['call_ph_off']))
    v, c = d['s2_full_v'], d['s2_full_c']
    n_e = len(r['e_t'])
    assert _ks(v, c, t) < _ks_limit(20000, n_e, 2.5)
    m, s = _moments(v, c)
    assert abs(t.mean() - m) < 6 * s / np.sqrt(n_e) and abs(t.std() / s - 1) < 0.03


def test_scheduled_runs_apply_the_rule_to_instructions_alone_in_their_call():
    """RNG spec v12: with run sets given (electron afterpulses: orc_simulate_scheduled) an S2 that is alone in its Pulse call is
    generated tile by tile like an S2 of a plain run; one that shares its call keeps the per-electron generator."""
    from wfsim_amd.scheduler import schedule
    cfg, ins = _s2(4, 400, s2_secondary_sc_gain=60.0, seed=11)
    res = Resource(cfg)
    order, key, cluster = schedule(ins, cfg)
    s_ins = ins[order]
    gid = order.astype(np.uint32)
    ip = instruction_params(s_ins, cfg, res)
    plain = make_oracle(cfg); plain.simulate(s_ins, gid, ip)
    sched = make_oracle(cfg); sched.simulate_scheduled(s_ins, gid, ip, np.zeros(len(ins), np.uint32), cluster, key, np.arange(len(ins)))
    a, b = plain.results(), sched.results()
    assert np.array_equal(a['ph_t'], b['ph_t']) and plain.pack_records().tobytes() == sched.pack_records().tobytes()
    off = make_oracle(dict(cfg, tile_local_generation=False)); off.simulate(s_ins, gid, ip)
    assert not np.array_equal(a['ph_t'], off.results()['ph_t'])            # (the tile path was really taken)
    # two S2s 200 ns apart in ONE call: the per-electron generator for both, whatever the switch
    cfg2, ins2 = _s2(2, 400, s2_secondary_sc_gain=60.0, seed=11)
    ins2['time'] = [MS, MS + 200]
    o2, k2, c2 = schedule(ins2, cfg2)
    ip2 = instruction_params(ins2[o2], cfg2, Resource(cfg2))
    out = []
    for c in (cfg2, dict(cfg2, tile_local_generation=False)):
        orc = make_oracle(c)
        orc.simulate_scheduled(ins2[o2], o2.astype(np.uint32), ip2, np.zeros(2, np.uint32), c2, k2, np.zeros(2, np.int32))
        out.append(orc.results()['ph_t'])
    assert np.array_equal(out[0], out[1])


def test_run_sets_are_numbered_by_their_first_instruction():
    from wfsim_amd.engine import first_instruction_of_sets
    assert np.array_equal(first_instruction_of_sets([0, 1, 1, 2, 1, 3], 4), [0, 1, 3, 5])
    assert np.array_equal(first_instruction_of_sets(np.arange(5), 5), np.arange(5))
    assert np.array_equal(first_instruction_of_sets([1, 0, 2], 3), [1, 0, 2])  # any numbering of used sets (S1 calls before S2 calls)
    assert first_instruction_of_sets([0, 2, 2], 3) is None                  # a set number without instructions


def test_grouped_pulse_calls_keep_the_per_electron_generator_their_neighbours_do_not():
    """save_full_truth=False (rawdata.py:106-127): S2s whose keys are at most int(0.2 / v) ns apart share a Pulse call and are generated
    electron by electron; an S2 alone in its call takes the tile path -- the same photons as with save_full_truth=True"""
    cfg, ins = _s2(5, 400, s2_secondary_sc_gain=60.0, seed=21)
    ins['time'] = [MS, MS + 300, 3 * MS, 5 * MS, 5 * MS + 200]          # two pairs and one S2 on its own (the third)
    grouped = _simulate(dict(cfg, save_full_truth=False), ins)
    grouped_off = _simulate(dict(cfg, save_full_truth=False, tile_local_generation=False), ins)
    single = _simulate(cfg, ins)
    single_off = _simulate(dict(cfg, tile_local_generation=False), ins)
    def of(res, t0, t1):
        t = res['ph_t']; return np.sort(t[(t >= t0) & (t < t1)])
    lo, hi = 3 * MS - 100_000, 3 * MS + 900_000
    assert np.array_equal(of(grouped, lo, hi), of(single, lo, hi)) and not np.array_equal(of(single, lo, hi), of(single_off, lo, hi))
    for t0 in (MS, 5 * MS):                                              # the pairs: the per-electron generator, whatever the switch
        assert np.array_equal(of(grouped, t0 - 100_000, t0 + 900_000), of(grouped_off, t0 - 100_000, t0 + 900_000))
