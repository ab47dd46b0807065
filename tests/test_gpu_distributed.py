"""simulate_sharded with several ranks sharing the one GPU of the test box (gloo for the exchange; the driver's multi-GPU
runs use RCCL, one rank per GPU): the gathered raw_records equal the single-process run byte for byte."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _case():
    from tests.test_gpu_random_mixes import _random_case
    cfg, ins, ap = _random_case(1007)
    rng = np.random.default_rng(8)
    # many well separated events so that every rank gets work
    big = np.concatenate([ins] * 6)
    big['time'] = np.cumsum(rng.choice([300, 40_000, 30_000_000], len(big))).astype(np.int64) + 1_000_000
    big['event_number'] = np.arange(len(big))
    return cfg, big


def _worker(rank, world, port, out, quanta=None, gather=True):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from wfsim_amd.distributed import simulate_sharded
    cfg, ins = _case()
    rec = simulate_sharded(cfg, ins, device=0, max_batch_quanta=quanta, gather=gather)
    if not gather:
        np.save(out + f'.{rank}.npy', rec)          # every rank keeps its own time range
    elif rank == 0:
        np.save(out, rec)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world,quanta', [(2, None), (3, None), (2, 20_000)])
def test_sharded_ranks_equal_single_process(tmp_path, world, quanta):
    """(quanta: small batches -- the ranks then run different numbers of gather rounds)"""
    import wfsim_amd
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / 'records.npy')
    mp.spawn(_worker, args=(world, port, out, quanta), nprocs=world, join=True)
    got = np.load(out)
    cfg, ins = _case()
    ref = np.concatenate([w['records'] for w in wfsim_amd.RawData(cfg).iter_windows(ins)])
    assert len(ref) > 1000 and got.tobytes() == ref.tobytes()


def test_sharded_without_the_gather_every_rank_keeps_its_time_range(tmp_path):
    """gather=False (delivery stays sharded, nothing funnels through one PCIe link): the per-rank records,
    concatenated in rank order, are the run"""
    import wfsim_amd
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / 'records')
    mp.spawn(_worker, args=(3, port, out, None, False), nprocs=3, join=True)
    parts = [np.load(out + f'.{r}.npy') for r in range(3)]
    cfg, ins = _case()
    ref = np.concatenate([w['records'] for w in wfsim_amd.RawData(cfg).iter_windows(ins)])
    assert all(len(p) for p in parts) and np.concatenate(parts).tobytes() == ref.tobytes()
    assert all(parts[r]['time'].max() < parts[r + 1]['time'].min() for r in range(2))


def test_device_resident_records_equal_host_records():
    """RawData.iter_batches(device_records=True): what simulate_sharded sends over RCCL (a torch uint8 tensor filled from the
    engine's record arena) == the host copy.  In a child process: torch.cuda has to be initialised before the HIP library."""
    import subprocess
    import sys
    code = """
import torch
torch.cuda.init()
import numpy as np, wfsim_amd
from tests.test_gpu_random_mixes import _random_case
cfg, ins, ap = _random_case(1008)
rd = wfsim_amd.RawData(cfg); rd.engine.set_record_order(False); rd.max_batch_quanta = 30_000
dev = [b['records'] for b in rd.iter_batches(ins, device_records=True)]
assert len(dev) > 1 and all(t.is_cuda for t in dev)
got = torch.cat(dev).cpu().numpy().tobytes()
ref = np.concatenate([w['records'] for w in wfsim_amd.RawData(cfg).iter_windows(ins)]).tobytes()
assert got == ref and len(ref) > 244 * 100
print('device records ok', len(ref) // 244)
"""
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), timeout=300)
    assert r.returncode == 0 and 'device records ok' in r.stdout, r.stderr[-2000:]


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher (what a driver that starts every bench line the same way does): bench.py
    starts the ranks itself and relays rank 0's JSON line.  Rehearsed over gloo, both ranks on the one GPU of the test box."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')}
    env['WFS_BENCH_BACKEND'] = 'gloo'
    for extra, gather in (([], 'none'), (['--gather'], 'gloo')):       # default: every rank keeps its records (no exchange step on this path)
        r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--instructions', '12', '--steps', '2', '--warmup', '1',
                            '--cpu-sample', '0'] + extra, capture_output=True, text=True, cwd=root, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert line['n_gpus'] == 2 and line['scaling'] == 'weak' and line['value'] > 0
        assert line['config']['instructions_per_gpu'] == 12 and gather in line['config']['gather']
