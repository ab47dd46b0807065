"""Event sharding + record gather, world_size 2 over gloo on the CPU (no GPU, no oracle)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from wfsim_amd.distributed import shard_clusters, gather_records, wait_gather, gather_batches
from wfsim_amd.dtypes import raw_record_dtype


def test_shard_clusters_whole_clusters_balanced():
    rng = np.random.default_rng(0)
    cluster = np.repeat(np.arange(200), rng.integers(1, 6, 200))
    weight = rng.uniform(1, 100, len(cluster))
    for world in (1, 2, 4, 8):
        b = shard_clusters(cluster, weight, world)
        assert b[0] == 0 and b[-1] == len(cluster) and np.all(np.diff(b) >= 0)
        for x in b[1:-1]:
            assert cluster[x] != cluster[x - 1]                    # cuts only between clusters
        loads = np.array([weight[b[r]:b[r + 1]].sum() for r in range(world)])
        assert loads.max() < 1.25 * loads.mean() + weight.max() * 6
    assert np.array_equal(shard_clusters(np.zeros(0, int), np.zeros(0), 4), np.zeros(5))
    assert np.array_equal(shard_clusters(np.zeros(3, int), np.ones(3), 2), [0, 3, 3])   # one cluster cannot be split
    # electron afterpulses: cuts only where the key jumps by more than rext + the longest delay
    cluster = np.arange(10); key = np.array([0, 1, 2, 3, 50, 51, 52, 53, 54, 55]) * 1_000_000
    b = shard_clusters(cluster, np.ones(10), 4, key=key, min_gap=5_000_000)
    assert set(b[1:-1].tolist()) <= {4, 10} and b[0] == 0 and b[-1] == 10 and np.all(np.diff(b) >= 0)
    assert np.array_equal(shard_clusters(cluster, np.ones(10), 2, key=key, min_gap=5_000_000), [0, 4, 10])


def _fake_records(rank, n):
    rec = np.zeros(n, dtype=raw_record_dtype())
    rec['time'] = 10_000_000 * rank + 10 * np.arange(n)
    rec['channel'] = np.arange(n) % 494
    rec['length'] = 110
    rec['data'][:] = (rank + 1)
    return rec


def _worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    rec = _fake_records(rank, [5, 0, 7][rank] if world == 3 else [5, 7][rank])
    t = torch.from_numpy(rec.view(np.uint8).copy())
    if world == 3:
        bufs = gather_records(t, dst=0)
    else:                                   # posted, overlapped with other work, then waited for
        bufs, handles = gather_records(t, dst=0, async_op=True)
        _ = torch.ones(1000).sum()
        wait_gather(handles)
    if rank == 0:
        allrec = np.concatenate([b.numpy().view(raw_record_dtype()) for b in bufs])
        np.save(out, allrec)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_gather_records_gloo(tmp_path, world):
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / 'gathered.npy')
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    got = np.load(out)
    sizes = [5, 0, 7] if world == 3 else [5, 7]
    ref = np.concatenate([_fake_records(r, n) for r, n in enumerate(sizes)])
    assert got.tobytes() == ref.tobytes()
    assert np.all(np.diff(got['time']) >= 0)


def _batches_of(rank):
    """rank r delivers r + 1 batches (rank 2: none) of a few fake records each, later batches later in time"""
    n_batches = {0: 1, 1: 3, 2: 0}[rank]
    out = []
    for k in range(n_batches):
        rec = _fake_records(rank, 3 + k)
        rec['time'] += 1_000_000 * k
        out.append(rec)
    return out


def _rounds_worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    parts = gather_batches((torch.from_numpy(r.view(np.uint8).copy()) for r in _batches_of(rank)), dst=0, device=torch.device('cpu'))
    if rank == 0:
        np.save(out, np.concatenate([p.numpy().view(raw_record_dtype()) for p in parts]))
    else:
        assert parts is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_gather_batches_with_unequal_numbers_of_rounds(tmp_path, world):
    """the multi-GPU product path (simulate_sharded): per-batch gathers, ranks with fewer batches keep taking part"""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / 'rounds.npy')
    mp.spawn(_rounds_worker, args=(world, port, out), nprocs=world, join=True)
    got = np.load(out)
    ref = np.concatenate([r for rank in range(world) for r in _batches_of(rank)])
    assert got.tobytes() == ref.tobytes() and np.all(np.diff(got['time']) >= 0)


def test_shard_plan_ranges_meet_and_cover_the_run():
    """shard_plan (no communication: every rank computes it from the same instructions): slices partition the sorted instructions at
    cluster boundaries, run-wide ids follow the slices, chunk ranges of consecutive non-empty shards meet at first key - rext"""
    from wfsim_amd.config import xenonnt_test_config
    from wfsim_amd.distributed import shard_plan
    from wfsim_amd.dtypes import instruction_dtype
    rng = np.random.default_rng(5)
    n = 400
    ins = np.zeros(n, dtype=instruction_dtype)
    ins['type'] = rng.choice([1, 2], n)
    ins['time'] = np.cumsum(rng.choice([300, 40_000, 30_000_000], n)).astype(np.int64) + 1_000_000
    ins['z'], ins['amp'] = -rng.uniform(1, 90, n), rng.integers(1, 500, n)
    cfg = xenonnt_test_config(enable_electron_afterpulses=False)
    for world in (1, 2, 3, 8):
        p = shard_plan(cfg, ins, world)
        b = p['bounds']
        assert b[0] == 0 and b[-1] == n and np.all(np.diff(b) >= 0) and len(b) == world + 1
        assert np.array_equal(np.sort(p['order']), np.arange(n))
        nonempty = [r for r in range(world) if b[r + 1] > b[r]]
        assert p['starts'][nonempty[0]] is None and p['ends'][nonempty[-1]] is None
        for a, nxt in zip(nonempty[:-1], nonempty[1:]):
            assert p['ends'][a] == p['starts'][nxt] == int(p['key'][b[nxt]]) - int(cfg['right_raw_extension'])
            assert p['cluster'][b[nxt]] != p['cluster'][b[nxt] - 1]                  # a cut never splits a cluster
            assert p['key'][b[nxt]] - p['key'][b[nxt] - 1] > cfg['right_raw_extension']
    # two calls agree (what lets the ranks plan without talking to each other)
    q = shard_plan(cfg, ins, 3)
    assert np.array_equal(q['bounds'], shard_plan(cfg, ins.copy(), 3)['bounds'])
