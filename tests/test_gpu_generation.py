"""Full HIP path (photon generation -> records) against the CPU oracle on the same Philox streams (MI355X only).

Oracle and GPU implement the stream layout of DESIGN.md independently; results must agree photon by photon:
times, channels, DPE flags and gains exact, and therefore pulses, rows, ZLE intervals and record bytes exact.
(Transcendentals come from different libm's -- ocml on the GPU, glibc on the host -- so a photon time could
in principle differ by 1 ns when a value lands within ~1e-13 of an integer; none does at these sizes.)
"""
import numpy as np
import pytest

from tests.helpers import make_engine, make_oracle
from wfsim_amd.config import xenonnt_test_config
from wfsim_amd.dtypes import instruction_dtype
from wfsim_amd.physics import instruction_params
from wfsim_amd.resource import Resource
from wfsim_amd.scheduler import schedule

pytestmark = pytest.mark.gpu
MS = 1_000_000


def _instructions(rows):
    ins = np.zeros(len(rows), dtype=instruction_dtype)
    for i, r in enumerate(rows):
        for k, v in r.items():
            ins[i][k] = v
        ins[i]['event_number'] = i
        ins[i]['recoil'] = 7
    return ins


def _run_both(cfg, ins, seed=7, ap=None):
    cfg = dict(cfg)
    cfg.setdefault('tile_local_min_photons', 0)          # every eligible S2 through the tile-local generator, also the small ones
    if ap is not None:
        cfg = dict(cfg, enable_pmt_afterpulses=True, uniform_to_pmt_ap=ap)
    res = Resource(cfg)
    order, key, cluster = schedule(ins, cfg)
    s_ins = ins[order]
    gid = order.astype(np.uint32)
    ip = instruction_params(s_ins, cfg, res)
    cfg = dict(cfg, seed=seed)
    orc = make_oracle(cfg, ap)
    orc.simulate(s_ins, gid, ip)
    o = orc.results()
    eng = make_engine(cfg)
    from wfsim_amd.scheduler import run_sets
    rs = None if cfg.get('save_full_truth', True) else run_sets(s_ins, key, cluster, cfg)[0]
    eng.load_instructions(s_ins, gid, cluster, key, ip, run_set=rs)
    counts = eng.run()
    return orc, o, eng, counts, s_ins


def _compare(orc, o, eng, counts, s_ins):
    # ---- photons per pulse set (oracle runs S1s before S2s inside a cluster; sets are matched by instruction)
    ph = eng.photons()
    assert counts['n_photons'] == len(o['ph_t'])
    # oracle call -> sorted instruction index: calls are in processing order; run-set id increases with it
    from wfsim_amd.scheduler import processing_order
    # the oracle received sorted instructions, so 'order' is the identity
    cluster = schedule(s_ins, eng.config)[2]
    proc = processing_order(s_ins, np.arange(len(s_ins)), cluster)
    n_ins = len(s_ins)
    # oracle calls: per instruction its primary call, then (PMT afterpulses on and photons present) a kind-3 call;
    # device sets: instruction i and, for its afterpulses, n_ins + i
    sets, it = [], iter(proc)
    for kind in o['call_kind']:
        if kind != 3:
            cur = next(it)
            sets.append(cur)
        else:
            sets.append(n_ins + cur)
    assert len(sets) == len(o['call_kind'])
    if 3 in o['call_kind']:
        assert counts['n_pulse_sets'] == 2 * n_ins and (o['call_kind'] == 3).sum() > 0
    for k, i in enumerate(sets):
        a, b = o['call_ph_off'][k], o['call_ph_off'][k + 1]
        c, e = ph['set_off'][i], ph['set_off'][i + 1]
        assert b - a == e - c, f'instruction {i}: {b - a} vs {e - c} photons'
        ko = np.lexsort((o['ph_gain'][a:b], o['ph_t'][a:b], o['ph_ch'][a:b]))
        kg = np.lexsort((ph['gain'][c:e], ph['t'][c:e], ph['ch'][c:e]))
        assert np.array_equal(o['ph_ch'][a:b][ko], ph['ch'][c:e][kg])
        assert np.array_equal(o['ph_t'][a:b][ko], ph['t'][c:e][kg])
        assert np.array_equal(o['ph_gain'][a:b][ko], ph['gain'][c:e][kg])
        assert np.array_equal(o['ph_dpe'][a:b][ko], ph['dpe'][c:e][kg])
    # ---- digitise windows, intervals, records
    g = eng.groups()
    keep = g['right'] >= g['left']
    assert np.array_equal(g['left'][keep], o['dg_left'])
    assert np.array_equal(g['right'][keep], o['dg_right'])
    assert eng.records().tobytes() == orc.pack_records().tobytes()
    assert counts['n_pe'] == orc.n_pe


def test_s2_sizes_around_the_tile_thresholds():
    """default thresholds (wfs_config.tile_gen_min = 64 photons on the brightest channel, 2048 photon registers): S2s below, inside
    and above the band in one batch -- block generator, tile-local generator, block generator -- equal the oracle photon by photon"""
    cfg = xenonnt_test_config(s2_secondary_sc_gain=100.0, tile_local_min_photons=64)
    rows = [dict(type=2, time=MS * (i + 1), x=1.0 * i, y=-2.0, z=-8.0 - 3 * i, amp=a) for i, a in enumerate([150, 380, 420, 2000, 9000, 11200, 13000])]
    rows += [dict(type=1, time=MS * 9, x=0, y=0, z=-40, amp=3000)]
    orc, o, eng, counts, s_ins = _run_both(cfg, _instructions(rows), seed=23)
    _compare(orc, o, eng, counts, s_ins)
    assert counts['n_photons'] > 2_000_000


def test_s1_batch():
    rng = np.random.default_rng(2)
    rows = [dict(type=1, time=MS * (i + 1), x=rng.uniform(-30, 30), y=rng.uniform(-30, 30), z=rng.uniform(-97, 0),
                 amp=int(a)) for i, a in enumerate(rng.integers(1, 4000, 200))]
    _compare(*_run_both(xenonnt_test_config(), _instructions(rows)))


def test_s2_batch():
    rng = np.random.default_rng(3)
    rows = [dict(type=2, time=MS * (i + 1), x=rng.uniform(-30, 30), y=rng.uniform(-30, 30), z=rng.uniform(-97, -0.5),
                 amp=int(a)) for i, a in enumerate(rng.integers(1, 600, 40))]
    _compare(*_run_both(xenonnt_test_config(), _instructions(rows)))


def test_mixed_clusters_and_large_s2():
    rows = []
    for i in range(12):
        t = MS * (i + 1)
        rows += [dict(type=1, time=t, x=3 * i, y=-2 * i, z=-5 - 5 * i, amp=800 + 100 * i),
                 dict(type=2, time=t, x=3 * i, y=-2 * i, z=-5 - 5 * i, amp=40 + 10 * i)]
    rows += [dict(type=1, time=20 * MS, x=0, y=0, z=-30, amp=500), dict(type=1, time=20 * MS + 300, x=1, y=1, z=-31, amp=600),
             dict(type=2, time=25 * MS, x=0, y=0, z=-10, amp=4000)]          # ~3e5 PE
    _compare(*_run_both(xenonnt_test_config(s2_secondary_sc_gain=100.0), _instructions(rows)))


def test_tiles_above_one_register_batch():
    """an S2 of ~4x10^6 PE: ~6000 photons per PMT, more than the 2048 a pulse workgroup keeps in registers -> its tiles are generated
    in passes (k_s2_tile<!FULL>, photons to the photon array in generation order) and pulsed by the dense kernel, one workgroup per
    (tile, window); a second, small S2 rides along in the same launch"""
    cfg = xenonnt_test_config(s2_secondary_sc_gain=100.0)
    rows = [dict(type=2, time=MS, x=2, y=-1, z=-12, amp=40000), dict(type=2, time=3 * MS, x=0, y=0, z=-30, amp=200),
            dict(type=1, time=5 * MS, x=0, y=0, z=-30, amp=900)]
    orc, o, eng, counts, s_ins = _run_both(cfg, _instructions(rows), seed=21)
    assert np.diff(eng.photons()['set_off']).max() > 2048 * 494
    _compare(orc, o, eng, counts, s_ins)
    # truth of the big S2 through the windowed kernel's own reduction path
    acc, ts = eng.truth()
    tr = o['truth'].reshape(-1, 12)
    big = int(np.argmax(acc[:, 0]))
    kbig = int(np.argmax(tr[:, 0]))
    for j in range(12):          # n_pe_trigger(_bottom) too: it depends on the photon order (pulse.py:255), the tiles are in generation order
        assert np.isclose(acc[big, j], tr[kbig, j], rtol=1e-9), j


def test_run_sets_several_instructions_per_pulse_call():
    """save_full_truth=False: S1s within 100 ns / S2s within 2 mm share one Pulse call (rawdata.py:106-127): tiles are
    per (run set, channel), photons of the grouped instructions merge into the same pulses; PMT afterpulses follow the set"""
    from tests.helpers import golden, ap_tables_from_golden
    from wfsim_amd.scheduler import run_sets
    cfg = xenonnt_test_config(save_full_truth=False)
    rows = [dict(type=1, time=MS, x=1, y=2, z=-10, amp=900), dict(type=1, time=MS + 50, x=3, y=-2, z=-12, amp=700),
            dict(type=1, time=MS + 120, x=-5, y=0, z=-11, amp=500), dict(type=1, time=MS + 400, x=9, y=9, z=-10, amp=1100),
            dict(type=2, time=MS, x=1, y=2, z=-10.0, amp=160), dict(type=2, time=MS + 50, x=3, y=-2, z=-10.1, amp=140),
            dict(type=2, time=MS + 120, x=-5, y=0, z=-10.5, amp=130),
            dict(type=1, time=3 * MS, x=0, y=0, z=-30, amp=2000), dict(type=2, time=3 * MS, x=0, y=0, z=-30, amp=100),
            dict(type=2, time=3 * MS + 700, x=4, y=0, z=-30, amp=5000)]        # a big S2 sharing its set with a small one
    for ap in (None, ap_tables_from_golden()):
        orc, o, eng, counts, s_ins = _run_both(cfg, _instructions(rows), seed=9, ap=ap)
        order, key, cluster = schedule(s_ins, cfg)
        rs, n_sets = run_sets(s_ins, key, cluster, cfg)
        assert n_sets == 6 and counts['n_pulse_sets'] == (12 if ap else 6)
        prim = o['call_kind'] != 3
        assert prim.sum() == n_sets
        # photons per pulse set: oracle calls are in processing order = set numbering
        ph = eng.photons()
        k_set = 0
        for k, kind in enumerate(o['call_kind']):
            i = (n_sets + k_set - 1) if kind == 3 else k_set
            if kind != 3:
                k_set += 1
            a, b = o['call_ph_off'][k], o['call_ph_off'][k + 1]
            c, e = ph['set_off'][i], ph['set_off'][i + 1]
            assert b - a == e - c
            ko = np.lexsort((o['ph_gain'][a:b], o['ph_t'][a:b], o['ph_ch'][a:b]))
            kg = np.lexsort((ph['gain'][c:e], ph['t'][c:e], ph['ch'][c:e]))
            assert np.array_equal(o['ph_t'][a:b][ko], ph['t'][c:e][kg]) and np.array_equal(o['ph_ch'][a:b][ko], ph['ch'][c:e][kg])
            assert np.array_equal(o['ph_gain'][a:b][ko], ph['gain'][c:e][kg])
        assert eng.records().tobytes() == orc.pack_records().tobytes()
        assert counts['n_pe'] == orc.n_pe
        # electron statistics per run set
        es = eng.electron_stats()
        assert len(es) == n_sets
        ks = [k for k, kind in enumerate(o['call_kind']) if kind != 3]
        for q, k in enumerate(ks):
            et = o['e_t'][o['call_e_off'][k]:o['call_e_off'][k + 1]]
            if o['call_kind'][k] == 2:
                assert es[q, 0] == len(et) and es[q, 2] == et.min() and es[q, 3] == et.max()
                assert abs(es[q, 1] - et.mean()) < 1e-3 and abs(es[q, 4] - et.std()) < 1e-3


def test_small_poisson_mean_and_gain_spread():
    rows = [dict(type=2, time=MS * (i + 1), x=0, y=0, z=-20, amp=300) for i in range(5)]
    _compare(*_run_both(xenonnt_test_config(s2_secondary_sc_gain=7.0, s2_gain_spread=2.5, s2_time_spread=30.0),
                        _instructions(rows)))


def test_pmt_afterpulses_generated_on_device():
    from tests.helpers import ap_tables_from_golden
    rows = [dict(type=1, time=MS * (i + 1), x=2 * i, y=-i, z=-10 - 3 * i, amp=3000 + 500 * i) for i in range(8)]
    rows += [dict(type=2, time=MS * (i + 20), x=i, y=2 * i, z=-8 - 4 * i, amp=150 + 30 * i) for i in range(8)]
    rows += [dict(type=1, time=40 * MS, x=0, y=0, z=-30, amp=1)]
    orc, o, eng, counts, s_ins = _run_both(xenonnt_test_config(), _instructions(rows), seed=21, ap=ap_tables_from_golden())
    n_ap = sum(o['call_ph_off'][k + 1] - o['call_ph_off'][k] for k in range(len(o['call_kind'])) if o['call_kind'][k] == 3)
    assert n_ap > 200
    _compare(orc, o, eng, counts, s_ins)


@pytest.mark.parametrize('scale,scale_uniform,modifier', [(1.0, 1.0, 1.0), (12.0, 12.0, 0.37), (4.0, 4.0, 2.5), (2.0, 60.0, 2.5), (0.0, 0.0, 1.0), (3.0, 3.0, 0.0)])
def test_pmt_afterpulse_screen_never_loses_an_afterpulse(scale, scale_uniform, modifier):
    """The generator screens afterpulse candidates with an integer threshold per (element, channel, single / double PE parent)
    (RNG spec v10, ap_threshold in wfs_engine.hip) and only candidates get the reference's floating-point comparison
    (afterpulse.py:196-204); the oracle makes that comparison for every photon and element.  Probabilities from 0 to beyond 1
    after the modifier (the 'Uniform' element of the fourth case fires for every photon), ~3 x 10^4 parent photons: every
    afterpulse the oracle makes must be there, photon by photon."""
    from tests.helpers import ap_tables_from_golden
    ap = ap_tables_from_golden()
    for name in ap:                                      # the last column of the delay CDF is the channel's afterpulse probability
        k = scale_uniform if name == 'Uniform' else scale
        ap[name] = dict(ap[name], delaytime_cdf=ap[name]['delaytime_cdf'] * k)
    rows = [dict(type=1, time=MS * (i + 1), x=i, y=-i, z=-20, amp=10000) for i in range(6)]
    rows += [dict(type=2, time=MS * (i + 10), x=i, y=i, z=-10, amp=150) for i in range(6)]
    orc, o, eng, counts, s_ins = _run_both(xenonnt_test_config(pmt_ap_modifier=modifier), _instructions(rows), seed=23, ap=ap)
    n_ap = sum(o['call_ph_off'][k + 1] - o['call_ph_off'][k] for k in range(len(o['call_kind'])) if o['call_kind'][k] == 3)
    n_par = counts['n_photons'] - n_ap
    assert n_par > 15000
    if scale == 0.0 or modifier == 0.0: assert n_ap == 0          # (modifier 0: rU0 / 0 = inf accepts nothing, afterpulse.py:198)
    elif scale_uniform == 60.0: assert n_ap >= n_par     # P(Uniform) * modifier = 0.48 * 2.5 > 1: one for every parent, and the others
    else: assert n_ap > 500
    _compare(orc, o, eng, counts, s_ins)


def test_pmt_afterpulses_of_large_tile_generated_tiles():
    """PMT afterpulses of tile-generated S2s (k_s2_tile<.., AP>): ~800 photons per tile, afterpulse probabilities x 4 so that about half
    of the tiles have more candidates than the workgroup's LDS stage holds (AP_STAGE = 128: the excess goes to the list entry by
    entry and is placed by the generic kernels, the rest as one key-ordered stretch per tile by k_ap_seg); one small S2 and an S1
    through the block generator in the same batch.  Device == oracle photon by photon, afterpulse tiles in generation order."""
    from tests.helpers import ap_tables_from_golden
    ap = ap_tables_from_golden()
    for name in ap: ap[name] = dict(ap[name], delaytime_cdf=ap[name]['delaytime_cdf'] * 4.0)
    rows = [dict(type=2, time=MS * (i + 1), x=3 * i, y=-2 * i, z=-10 - i, amp=4000) for i in range(2)]
    rows += [dict(type=2, time=MS * 5, x=1, y=1, z=-30, amp=40), dict(type=1, time=MS * 7, x=0, y=0, z=-50, amp=5000)]
    orc, o, eng, counts, s_ins = _run_both(xenonnt_test_config(s2_secondary_sc_gain=100.0), _instructions(rows), seed=29, ap=ap)
    n_ap = sum(o['call_ph_off'][k + 1] - o['call_ph_off'][k] for k in range(len(o['call_kind'])) if o['call_kind'][k] == 3)
    assert n_ap > 60000 and counts['n_photons'] - n_ap > 500000
    _compare(orc, o, eng, counts, s_ins)


def test_noise_on_generated_path():
    from tests.helpers import golden
    noise = golden('noise.npz')['noise']
    rows = [dict(type=1, time=MS * (i + 1), x=2 * i, y=-i, z=-10 - 3 * i, amp=300 + 200 * i) for i in range(10)]
    rows += [dict(type=2, time=MS * (i + 20), x=i, y=2 * i, z=-8 - 4 * i, amp=50 + 30 * i) for i in range(6)]
    cfg = xenonnt_test_config(enable_noise=True, noise_data=noise)
    orc, o, eng, counts, s_ins = _run_both(cfg, _instructions(rows), seed=22)
    g = eng.groups()
    keep = g['right'] >= g['left']
    assert np.array_equal(g['ix_rand'][keep], o['dg_ix_rand']) and np.all(o['dg_ix_rand'] >= 0)
    _compare(orc, o, eng, counts, s_ins)


@pytest.mark.parametrize('n_target', [1, 63, 65, 257])
def test_tile_reductions_with_inactive_lanes(n_target, monkeypatch):
    """Regression test of the round-3 fault: the tile's tmin / tmax / n_dpe reductions in k_s2_tile run on DPP moves, and a
    DPP read of a switched-off lane inside a divergent region returns the old value -- a tile whose last wave has inactive lanes then
    got a garbage time range.  Tile-generated S2s whose brightest tile holds about 1, 63, 65 and 257 photons on ONE channel (a pattern
    row with all light on channel 17), so that the first / second / fifth wave of the workgroup is partly empty; every launch checked
    (WFS_CHECK_LAUNCHES=1: a faulting kernel is named on the spot); device == oracle photon by photon, records byte for byte."""
    monkeypatch.setenv('WFS_CHECK_LAUNCHES', '1')
    from wfsim_amd.resource import Resource
    cfg = dict(xenonnt_test_config(s2_secondary_sc_gain=n_target / 4.0), tile_local_min_photons=0, seed=40 + n_target)
    ins = _instructions([dict(type=2, time=MS * (i + 1), x=0, y=0, z=-5 - i, amp=4) for i in range(12)])
    res = Resource(cfg)
    one_channel = np.zeros(494); one_channel[17] = 1.0
    res.s2_pattern_map = (lambda pos, **kw: np.repeat(one_channel[None, :], len(pos), axis=0))      # a plain callable: host rows
    order, key, cluster = schedule(ins, cfg)
    s_ins, gid = ins[order], order.astype(np.uint32)
    ip = instruction_params(s_ins, cfg, res)
    orc = make_oracle(cfg, resource=res)
    orc.simulate(s_ins, gid, ip)
    o = orc.results()
    eng = make_engine(cfg, resource=res)
    eng.load_instructions(s_ins, gid, cluster, key, ip)
    counts = eng.run()
    assert set(np.unique(o['ph_ch'])) == {17}
    sizes = np.diff(o['call_ph_off'])
    assert sizes.min() >= 0 and abs(np.median(sizes) - n_target) < max(8, 0.5 * n_target)
    _compare(orc, o, eng, counts, s_ins)


@pytest.mark.parametrize('afterpulses', [False, True])
def test_bright_tiles_next_to_ordinary_ones(afterpulses):
    """A position dependent hit pattern: channel 17 takes 5 % of the light, channel 300 1 %, the others share the rest -- one S2 then
    holds tiles of ~30000, ~6000 and ~1200 photons.  Which kernel makes a tile is decided per tile (k_tile_counts): up to 2048 photons
    photons and pulse in one workgroup, above that generation in passes + the dense pulse kernel; the Philox coordinates of photon q
    of a tile are the same either way.  Device == oracle photon by photon (PMT afterpulses included), records byte for byte, truth rows
    equal in every column (n_pe_trigger depends on the order inside the channel: generation order on both sides)."""
    from tests.helpers import ap_tables_from_golden
    from wfsim_amd.resource import Resource
    cfg = dict(xenonnt_test_config(s2_secondary_sc_gain=100.0), tile_local_min_photons=0, seed=77)
    ap = ap_tables_from_golden() if afterpulses else None
    if ap is not None:
        cfg = dict(cfg, enable_pmt_afterpulses=True, uniform_to_pmt_ap=ap)
    ins = _instructions([dict(type=2, time=MS, x=0, y=0, z=-8, amp=6000), dict(type=2, time=3 * MS, x=1, y=1, z=-20, amp=900),
                         dict(type=1, time=5 * MS, x=0, y=0, z=-30, amp=2000)])
    res = Resource(cfg)
    p = np.full(494, (1.0 - 0.06) / 492); p[17], p[300] = 0.05, 0.01
    res.s2_pattern_map = (lambda pos, **kw: np.repeat(p[None, :], len(pos), axis=0))
    order, key, cluster = schedule(ins, cfg)
    s_ins, gid = ins[order], order.astype(np.uint32)
    ip = instruction_params(s_ins, cfg, res)
    orc = make_oracle(cfg, ap, resource=res)
    orc.simulate(s_ins, gid, ip)
    o = orc.results()
    eng = make_engine(cfg, resource=res)
    eng.load_instructions(s_ins, gid, cluster, key, ip)
    counts = eng.run()
    per_channel = np.bincount(o['ph_ch'][o['call_ph_off'][0]:o['call_ph_off'][1]], minlength=494)
    assert per_channel[17] > 20000 and 2048 < per_channel[300] < 10000 and np.median(per_channel) < 2048
    _compare(orc, o, eng, counts, s_ins)
    acc, ts = eng.truth()
    tr = o['truth'].reshape(-1, 12)
    assert len(acc) == len(tr)
    for k in range(len(tr)):
        kk = int(np.argmin(np.abs(acc[:, 0] - tr[k, 0])))          # (sets are matched by their photon number)
        assert np.allclose(acc[kk], tr[k], rtol=1e-9), (k, acc[kk], tr[k])


def test_prepass_counts_and_photon_times_of_tile_generated_instructions():
    """The electron-afterpulse pre-pass (rawdata.py:133-145 -> afterpulse.py:49, 70-87) needs, per parent S2, its photon number and the
    arrival times of a few picked photons.  For tile-generated instructions both come from the tiles' counters and the photons' own
    Philox coordinates -- no photon is generated in the pre-pass (Engine.generate()): the photon offsets equal the photons per
    instruction of the full run, and the times recomputed for EVERY index of an instruction are, as a multiset, the times of its photons
    in the full run (tile-generated S2s with ordinary and bright tiles, a small S2 and an S1 of the per-electron generator)."""
    from wfsim_amd.resource import Resource
    cfg = dict(xenonnt_test_config(s2_secondary_sc_gain=80.0), seed=61)          # default tile_local_min_photons: the small S2 keeps the per-electron generator
    ins = _instructions([dict(type=2, time=MS, x=0, y=0, z=-8, amp=3000), dict(type=2, time=3 * MS, x=1, y=1, z=-20, amp=30),
                         dict(type=1, time=5 * MS, x=0, y=0, z=-30, amp=2000), dict(type=2, time=7 * MS, x=2, y=-1, z=-40, amp=1500)])
    res = Resource(cfg)
    p = np.full(494, (1.0 - 0.05) / 493); p[40] = 0.05                            # channel 40 holds bright tiles (> 2048 photons)
    res.s2_pattern_map = (lambda pos, **kw: np.repeat(p[None, :], len(pos), axis=0))
    order, key, cluster = schedule(ins, cfg)
    s_ins, gid = ins[order], order.astype(np.uint32)
    ip = instruction_params(s_ins, cfg, res)
    eng = make_engine(cfg, resource=res)
    eng.load_instructions(s_ins, gid, cluster, key, ip)
    eng.run()
    ph = eng.photons()
    n_per_ins = np.diff(ph['set_off'])[:len(s_ins)]
    eng.keep_photons = False                                                      # the pre-pass proper: counts only
    eng.generate()
    off = eng.instruction_photon_offsets()
    assert np.array_equal(np.diff(off), n_per_ins) and n_per_ins.max() > 100000
    for i in range(len(s_ins)):
        t = eng.gather_photon_times(np.arange(off[i], off[i + 1]))
        ref = ph['t'][ph['set_off'][i]:ph['set_off'][i + 1]]
        assert np.array_equal(np.sort(t), np.sort(ref)), i
    # channel-major order inside a tile-generated instruction: the first photons are those of channel 0
    c0 = int(np.sum(ph['ch'][ph['set_off'][0]:ph['set_off'][1]] == 0))
    t0 = eng.gather_photon_times(np.arange(off[0], off[0] + c0))
    sel = ph['ch'][ph['set_off'][0]:ph['set_off'][1]] == 0
    assert np.array_equal(np.sort(t0), np.sort(ph['t'][ph['set_off'][0]:ph['set_off'][1]][sel]))


def test_tile_generation_next_to_the_shared_calls_of_electron_afterpulses():
    """enable_electron_afterpulses (the reference's default, rawdata.py:194): the secondaries of a cluster share one Pulse call, so the
    batch carries run sets -- numbered by their first instruction (engine.load_instructions), which keeps every primary S2 that is alone
    in its call on the tile-local generator.  The tile kernel runs, and records equal the oracle's (whose rule is the same)."""
    import wfsim_amd
    from wfsim_amd.scheduler import feedback_schedule
    edges = np.linspace(0, 150e3, 141)
    hist = np.exp(-np.arange(140) / 30.0); hist *= 3e-4 / hist.sum()
    cfg = xenonnt_test_config(enable_electron_afterpulses=True, uniform_to_ele_ap=(hist, edges), seed=77, s2_secondary_sc_gain=100.0)
    n = 6
    ins = np.zeros(n, dtype=instruction_dtype)
    ins['type'] = [2, 1, 2, 2, 1, 2]
    ins['time'] = 1_000_000 * (1 + np.arange(n))
    ins['x'], ins['y'], ins['z'] = [0, 3, -8, 12, 0, 5], [1, 0, 4, -6, 9, -2], [-10, -30, -55, -70, -20, -90]
    ins['amp'] = [3000, 800, 9000, 500, 3000, 2000]
    ins['recoil'], ins['event_number'] = 7, np.arange(n)
    rd = wfsim_amd.RawData(cfg)
    rd.engine.set_profiling(True)
    windows = list(rd.iter_windows(ins))
    assert 'k_s2_tile' in rd.engine.kernel_times()
    rec = np.concatenate([w['records'] for w in windows])
    sec, sec_gid, sec_base, sec_parent = rd.electron_afterpulse_instructions(ins, np.arange(n), with_parent=True)
    assert len(sec) > 10
    allins = np.concatenate([ins, sec]); gids = np.concatenate([np.arange(n), sec_gid])
    base = np.concatenate([np.zeros(n, np.uint32), sec_base]); parent = np.concatenate([np.full(n, -1), sec_parent])
    order, key, cluster, rs = feedback_schedule(allins, parent, cfg)
    s_ins = allins[order]
    assert len(np.unique(rs)) < len(rs)                      # at least one shared call
    orc = make_oracle(cfg)
    orc.simulate_scheduled(s_ins, gids[order].astype(np.uint32), instruction_params(s_ins, cfg, Resource(cfg)), base[order], cluster, key, rs)
    assert rec.tobytes() == orc.pack_records().tobytes()


def test_tile_generation_with_grouped_pulse_calls():
    """save_full_truth=False: two S2s 300 ns apart share a Pulse call (per-electron generator), the S2s on their own take the tile kernel;
    photons, records and truth rows equal the oracle's"""
    from wfsim_amd.scheduler import run_sets
    cfg = xenonnt_test_config(save_full_truth=False, s2_secondary_sc_gain=100.0, seed=31)
    ins = np.zeros(5, dtype=instruction_dtype)
    ins['type'], ins['amp'], ins['z'], ins['recoil'] = 2, [4000, 3000, 6000, 2500, 2500], -10.0, 7
    ins['time'] = [1_000_000, 1_000_300, 3_000_000, 5_000_000, 5_000_150]
    ins['event_number'] = np.arange(5)
    res = Resource(cfg)
    order, key, cluster = schedule(ins, cfg)
    s_ins, gid = ins[order], order.astype(np.uint32)
    ip = instruction_params(s_ins, cfg, res)
    rs, n_sets = run_sets(s_ins, key, cluster, cfg)
    assert n_sets == 3
    eng = make_engine(cfg)
    eng.set_profiling(True)
    eng.load_instructions(s_ins, gid, cluster, key, ip, run_set=rs)
    counts = eng.run()
    assert 'k_s2_tile' in eng.kernel_times() and 'k_photon_fill' in eng.kernel_times() and counts['n_pulse_sets'] == 3
    orc = make_oracle(cfg)
    orc.simulate(s_ins, gid, ip)
    o = orc.results()
    assert counts['n_photons'] == len(o['ph_t'])
    assert eng.records().tobytes() == orc.pack_records().tobytes()
    acc, ts = eng.truth()
    assert acc.shape[0] == 3 and acc[:, 0].sum() == counts['n_photons']          # one truth row per Pulse call, every photon in one of them
