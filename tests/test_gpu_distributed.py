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
    # default: the step ends with the gather of every rank's records on rank 0 (the exchange step north_star names), overlapped with the
    # next batch; the same run also reports the rate without it.  --no-gather: only that one.
    for extra, gather in (([], 'send/recv'), (['--no-gather'], 'none')):
        r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--instructions', '12', '--steps', '2', '--warmup', '1',
                            '--cpu-sample', '0'] + extra, capture_output=True, text=True, cwd=root, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert line['n_gpus'] == 2 and line['scaling'] == 'weak' and line['value'] > 0
        assert line['config']['instructions_per_gpu'] == 12 and gather in line['config']['gather']
        if not extra:
            assert 'gloo send/recv to rank 0, overlapped' in line['config']['gather'] and 'gathered on rank 0' in line['metric']
            assert line['value_no_gather'] > 0 and line['config']['gather_ms_per_step'] > 0
            assert line['config']['gather_bytes_per_step'] == 244 * line['config']['records_per_step']
        else:
            assert 'value_no_gather' not in line


def _plugin_chunks(cfg):
    import wfsim_amd
    from wfsim_amd import ministrax
    return ministrax.run_plugin(wfsim_amd.RawRecordsFromFaxNT(cfg))


def _plugin_case():
    cfg, ins = _case()
    ok = (np.hypot(ins['x'], ins['y']) < cfg['tpc_radius']) & (ins['z'] < 0.25) & (ins['amp'] > 0) & ~((ins['type'] == 2) & (ins['z'] < -cfg['tpc_length']))
    return dict(cfg, chunk_size=0.05), ins[ok]        # (what check_instructions accepts, strax_interface.py:687-700)


def _assert_sharded_chunks_equal_single(parts, single, world):
    for kind in ('raw_records', 'raw_records_he', 'raw_records_aqmon'):
        ref = np.concatenate([c.data for c in single[kind]])
        got = np.concatenate([c.data for p in parts for c in p[kind]])
        assert got.tobytes() == ref.tobytes(), kind
    assert len(np.concatenate([c.data for c in single['raw_records']])) > 1000
    # truth rows: the same rows (a row without photons leaves with the chunk its instruction time falls into, which a shard cut can move)
    ref = np.concatenate([c.data for c in single['truth']])
    got = np.concatenate([c.data for p in parts for c in p['truth']])
    assert len(got) == len(ref) and np.sort(got, order=['event_number', 'type', 'time']).tobytes() == np.sort(ref, order=['event_number', 'type', 'time']).tobytes()
    # the chunk streams join: every rank's chunks are contiguous, rank r + 1 starts where rank r ended, every record inside its chunk
    chunks = [c for p in parts for c in p['raw_records']]
    assert len(chunks) >= world and all(a.end == b.start for a, b in zip(chunks[:-1], chunks[1:]))
    assert chunks[0].start == single['raw_records'][0].start and chunks[-1].end == single['raw_records'][-1].end
    for c in chunks:
        if len(c.data):
            assert c.data['time'].min() >= c.start and c.data['time'].max() <= c.end


@pytest.mark.parametrize('world', [2, 3])
def test_plugin_sharded_delivery_equals_single_process(world):
    """RawRecordsFromFaxNT with config['shard'] = (rank, world): every rank emits ITS time range as its own strax chunks (no gather, no
    collective); the chunk streams of the ranks, one after the other, hold the single-process run byte for byte.  (The ranks need no
    communication, so they can run one after the other in this process.)"""
    cfg, ins = _plugin_case()
    single = _plugin_chunks(dict(cfg, instructions=ins))
    parts = [_plugin_chunks(dict(cfg, instructions=ins, shard=(r, world))) for r in range(world)]
    assert all(len(p['raw_records']) for p in parts)
    _assert_sharded_chunks_equal_single(parts, single, world)


def _plugin_worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import pickle
    cfg, ins = _plugin_case()
    res = _plugin_chunks(dict(cfg, instructions=ins, shard='auto'))            # rank and world size from the process group
    with open(out + f'.{rank}.pkl', 'wb') as f:
        pickle.dump({k: [(c.start, c.end, c.data) for c in v] for k, v in res.items()}, f)
    dist.barrier()
    dist.destroy_process_group()


def test_plugin_sharded_delivery_two_processes(tmp_path):
    """the same with two processes of a torch.distributed group (gloo; shard = 'auto'), sharing the test box's GPU"""
    import pickle
    from types import SimpleNamespace
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / 'chunks')
    mp.spawn(_plugin_worker, args=(2, port, out), nprocs=2, join=True)
    parts = []
    for r in range(2):
        with open(out + f'.{r}.pkl', 'rb') as f:
            parts.append({k: [SimpleNamespace(start=a, end=b, data=d) for a, b, d in v] for k, v in pickle.load(f).items()})
    cfg, ins = _plugin_case()
    _assert_sharded_chunks_equal_single(parts, _plugin_chunks(dict(cfg, instructions=ins)), 2)
