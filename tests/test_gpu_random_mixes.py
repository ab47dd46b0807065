"""Randomised end-to-end runs on the GPU against the CPU oracle: instruction mixes, secondary gains, run sets, PMT
afterpulses and noise drawn at random (fixed seeds); records must agree byte for byte every time.  Aimed at the seams:
photon blocks that straddle instructions, blocks with more emitters than the LDS window, empty instructions, tiles on
either side of the tiny / sparse / dense classes."""
import numpy as np
import pytest

from tests.helpers import make_engine, make_oracle, ap_tables_from_golden, golden
from wfsim_amd.config import xenonnt_test_config
from wfsim_amd.dtypes import instruction_dtype
from wfsim_amd.physics import instruction_params
from wfsim_amd.resource import Resource
from wfsim_amd.scheduler import schedule, run_sets

pytestmark = pytest.mark.gpu


def _random_case(seed):
    rng = np.random.default_rng(seed)
    kw = dict(s2_secondary_sc_gain=float(rng.choice([1.5, 4.0, 21.3, 100.0])), seed=int(rng.integers(1, 10 ** 6)))
    kw['tile_local_min_photons'] = int(rng.choice([0, 0, 64]))       # (the tile-local generator also for S2s too small to pay for it)
    if rng.random() < 0.4:
        kw['save_full_truth'] = False
    if rng.random() < 0.3:
        kw['s2_time_spread'] = float(rng.choice([0.0, 30.0]))
    ap = ap_tables_from_golden() if rng.random() < 0.35 else None
    if ap is not None:
        kw.update(enable_pmt_afterpulses=True, uniform_to_pmt_ap=ap)
    if rng.random() < 0.3:
        kw.update(enable_noise=True, noise_data=golden('noise.npz')['noise'])
    u = np.random.default_rng(seed + 77).random()             # (its own stream: the cases of the earlier rounds keep their draws)
    if u < 0.5:                                                # resident rows forced on / off for a quarter of the cases each, else the batch decides
        kw['row_resident'] = u < 0.25
    cfg = xenonnt_test_config(**kw)
    n = int(rng.integers(3, 60))
    ins = np.zeros(n, dtype=instruction_dtype)
    ins['type'] = rng.choice([1, 2], n, p=[0.5, 0.5])
    # clusters of instructions: a few within ns..us of each other, the rest far apart
    base = np.cumsum(rng.choice([200, 3_000, 40_000, 500_000, 3_000_000], n)).astype(np.int64) + 1_000_000
    ins['time'] = base
    ins['x'], ins['y'] = rng.uniform(-30, 30, n), rng.uniform(-30, 30, n)
    ins['z'] = -rng.uniform(0.5, 95, n)
    s1 = ins['type'] == 1
    ins['amp'] = np.where(s1, rng.choice([0, 1, 40, 700, 5000, 30000], n), rng.choice([0, 1, 7, 60, 400, 2500, 9000], n, p=[.1, .15, .2, .2, .2, .1, .05]))
    ins['recoil'], ins['event_number'] = 7, np.arange(n)
    return cfg, ins, ap


import os


@pytest.mark.parametrize('seed', list(range(int(os.environ.get('WFS_RANDOM_MIXES', 30)))))
def test_random_mix_matches_oracle(seed):
    cfg, ins, ap = _random_case(1000 + seed)
    res = Resource(cfg)
    order, key, cluster = schedule(ins, cfg)
    s_ins = ins[order]
    gid = order.astype(np.uint32)
    ip = instruction_params(s_ins, cfg, res)
    orc = make_oracle(cfg, ap)
    orc.simulate(s_ins, gid, ip)
    eng = make_engine(cfg)
    rs = None if cfg.get('save_full_truth', True) else run_sets(s_ins, key, cluster, cfg)[0]
    eng.load_instructions(s_ins, gid, cluster, key, ip, run_set=rs)
    counts = eng.run()
    o = orc.results()
    assert counts['n_photons'] == len(o['ph_t'])
    g = eng.groups()
    keep = g['right'] >= g['left']
    assert np.array_equal(g['left'][keep], o['dg_left']) and np.array_equal(g['right'][keep], o['dg_right'])
    assert eng.records().tobytes() == orc.pack_records().tobytes()
    assert counts['n_pe'] == orc.n_pe


@pytest.mark.parametrize('seed', list(range(int(os.environ.get('WFS_RANDOM_MIXES_EAP', 15)))))
def test_random_mix_with_electron_afterpulses(seed):
    """the same with electron afterpulses (and, half of the time, gate afterpulses / run sets): RawData end to end
    against the oracle fed with the same secondaries in the order of the feedback schedule"""
    import wfsim_amd
    from wfsim_amd.scheduler import feedback_schedule
    rng = np.random.default_rng(5000 + seed)
    edges = np.linspace(0, float(rng.choice([150e3, 700e3])), 141)
    hist = np.exp(-np.arange(140) / 30.0); hist *= float(rng.choice([5e-4, 3e-3])) / hist.sum()
    kw = dict(enable_electron_afterpulses=True, uniform_to_ele_ap=(hist, edges), seed=int(rng.integers(1, 10 ** 6)),
              s2_secondary_sc_gain=float(rng.choice([21.3, 60.0])))
    if rng.random() < 0.5:
        kw.update(enable_gate_afterpulses=True, photoelectric_p=float(rng.choice([1e-4, 2e-3])))
    if rng.random() < 0.5:
        kw['save_full_truth'] = False
    cfg = xenonnt_test_config(**kw)
    n = int(rng.integers(3, 25))
    ins = np.zeros(n, dtype=instruction_dtype)
    ins['type'] = rng.choice([1, 2], n)
    ins['time'] = np.cumsum(rng.choice([300, 30_000, 250_000, 2_000_000], n)).astype(np.int64) + 1_000_000
    ins['x'], ins['y'], ins['z'] = rng.uniform(-30, 30, n), rng.uniform(-30, 30, n), -rng.uniform(0.5, 95, n)
    ins['amp'] = np.where(ins['type'] == 1, rng.choice([0, 50, 900, 6000], n), rng.choice([0, 5, 80, 600, 2000], n))
    ins['recoil'], ins['event_number'] = 7, np.arange(n)
    rd = wfsim_amd.RawData(cfg)
    rd.max_batch_quanta = int(rng.choice([30_000, 2_000_000_000]))
    windows = list(rd.iter_windows(ins))
    rec = np.concatenate([w['records'] for w in windows]) if windows else np.zeros(0)
    sec, sec_gid, sec_base, sec_parent = rd.electron_afterpulse_instructions(ins, np.arange(n), with_parent=True)
    allins = np.concatenate([ins, sec]); gids = np.concatenate([np.arange(n), sec_gid])
    base = np.concatenate([np.zeros(n, np.uint32), sec_base]); parent = np.concatenate([np.full(n, -1), sec_parent])
    order, key, cluster, rs = feedback_schedule(allins, parent, cfg)
    s_ins = allins[order]
    orc = make_oracle(cfg)
    orc.simulate_scheduled(s_ins, gids[order].astype(np.uint32), instruction_params(s_ins, cfg, Resource(cfg)), base[order], cluster, key, rs)
    o = orc.results()
    assert len(windows) == len(o['dg_left']) and np.array_equal([w['left'] for w in windows], o['dg_left'])
    assert (rec.tobytes() if len(rec) else b'') == orc.pack_records().tobytes()


@pytest.mark.parametrize('seed', list(range(int(os.environ.get('WFS_RANDOM_MIXES_BATCH', 15)))))
def test_random_mix_small_batches_equal_one_batch(seed):
    """RawData.iter_windows: windows and record bytes do not depend on how the run is cut into GPU batches
    (window carry across batches, re-run of an open window, truth rows) -- random mixes incl. afterpulses, noise, run sets"""
    import wfsim_amd
    from wfsim_amd.dtypes import truth_extra_dtype
    cfg, ins, ap = _random_case(9000 + seed)
    rng = np.random.default_rng(seed)

    def run(mbq):
        rd = wfsim_amd.RawData(cfg)
        rd.max_batch_quanta = mbq
        truth = np.zeros(4 * len(ins) + 10, dtype=instruction_dtype + truth_extra_dtype + [('fill', bool)])
        w = [(x['left'], x['right'], x['records'].tobytes()) for x in rd.iter_windows(ins, truth_buffer=truth)]
        t = truth[truth['fill']]
        return w, t
    w1, t1 = run(2_000_000_000)
    w2, t2 = run(int(rng.choice([500, 5_000, 60_000])))
    assert w1 == w2
    assert len(t1) == len(t2)
    for f in ('n_photon', 'n_pe', 't_first_photon', 't_last_photon', 'n_electron', 'amp', 'time', 'event_number'):
        assert np.array_equal(t1[f], t2[f], equal_nan=True), f
    assert np.allclose(t1['raw_area'], t2['raw_area'], rtol=1e-12)        # float sum over the photons of a tile: order dependent


@pytest.mark.parametrize('seed', list(range(int(os.environ.get('WFS_RANDOM_MIXES_SHARD', 12)))))
def test_random_mix_sharded_equals_single(seed):
    """what every rank of a multi-GPU run would compute (contiguous cluster ranges from shard_clusters, run-wide
    instruction ids), one after the other on this GPU: the concatenation equals the single-GPU run byte for byte"""
    import wfsim_amd
    from wfsim_amd.distributed import shard_clusters, safe_cut_gap
    cfg, ins, ap = _random_case(12000 + seed)
    world = int(np.random.default_rng(seed).choice([2, 3, 8]))
    single = b''.join(w['records'].tobytes() for w in wfsim_amd.RawData(cfg).iter_windows(ins))
    order, key, cluster = schedule(ins, cfg)
    s_ins = ins[order]
    b = shard_clusters(cluster, np.maximum(s_ins['amp'], 1), world, key=key, min_gap=safe_cut_gap(cfg))
    parts = []
    for r in range(world):
        mine = s_ins[b[r]:b[r + 1]]
        if len(mine) == 0:
            continue
        rd = wfsim_amd.RawData(cfg)
        rd.global_ids = order[b[r]:b[r + 1]]
        parts.append(b''.join(w['records'].tobytes() for w in rd.iter_windows(mine)))
    assert b''.join(parts) == single


@pytest.mark.parametrize('seed', list(range(int(os.environ.get('WFS_RANDOM_MIXES_OPT', 12)))))
def test_random_optical_matches_oracle(seed):
    """RawDataOptical at random rates / batch sizes against the oracle's optical scheduler"""
    import wfsim_amd
    from tests.test_gpu_optical import nveto_config, optical_instructions
    rng = np.random.default_rng(15000 + seed)
    cfg = nveto_config(seed=int(rng.integers(1, 10 ** 6)), right_raw_extension=int(rng.choice([500, 2000, 20000])))
    ins, channels, timings = optical_instructions(int(rng.integers(50, 1500)), float(rng.choice([300.0, 1000.0, 30000.0])), int(rng.integers(1, 1000)))
    rd = wfsim_amd.RawDataOptical(cfg, channels=channels, timings=timings)
    rd.max_batch_quanta = int(rng.choice([200, 5000, 2_000_000_000]))
    windows = list(rd.iter_windows(ins))
    rec = b''.join(w['records'].tobytes() for w in windows)
    orc = make_oracle(cfg)
    orc.simulate_optical(ins, np.arange(len(ins), dtype=np.uint32), channels, timings, int(1e6))
    o = orc.results()
    assert len(windows) == len(o['dg_left']) and np.array_equal([w['left'] for w in windows], o['dg_left'])
    assert rec == orc.pack_records().tobytes()


def test_simulate_sharded_single_rank_equals_rawdata():
    """the multi-GPU entry point end to end on one rank (gloo): scheduling, the shard-cut check, the record gather"""
    import torch.distributed as dist
    import wfsim_amd
    from wfsim_amd.distributed import simulate_sharded
    import socket
    cfg, ins, ap = _random_case(1234)
    sock = socket.socket()
    sock.bind(('127.0.0.1', 0))
    port = sock.getsockname()[1]
    sock.close()
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1)
    try:
        rec = simulate_sharded(cfg, ins, device=0)
    finally:
        dist.destroy_process_group()
    rd = wfsim_amd.RawData(cfg)
    ref = np.concatenate([w['records'] for w in rd.iter_windows(ins)])
    assert rec.tobytes() == ref.tobytes() and len(rec) > 0
