"""Multi-GPU: event-sharded data parallelism + a gather of raw_records (one process per GPU, torch.distributed).

The reference is single process (``parallel = False``, /root/reference/wfsim/strax_interface.py:546).  Time clusters
of instructions are independent, so every rank simulates a contiguous time range of whole clusters
with the same code path as a single GPU; RNG streams are keyed by the run-wide instruction index, so the records
do not depend on the sharding.  The only exchange is the final variable-length gather of the packed 244-byte
records on the rank that feeds strax: sizes by ``all_gather``, payload by direct peer -> root ``send`` / ``recv``
(RCCL over xGMI with the ``nccl`` backend: the root's seven links fill in parallel; ``gloo`` in the CPU tests).
Concatenating the shards in rank order gives the globally time-ordered stream.
"""
import numpy as np

from .config import afterpulse_switches


def shard_clusters(cluster, weight, world_size, key=None, min_gap=None):
    """Contiguous ranges of whole clusters, balanced by ``weight`` (e.g. expected quanta per instruction).

    ``cluster``: non-decreasing cluster index of every (sorted) instruction.  Returns ``bounds`` with
    ``world_size + 1`` instruction indices: rank r owns ``[bounds[r], bounds[r + 1])``.
    With ``key`` and ``min_gap`` a range may only start where the scheduler key jumps by more than ``min_gap`` ns:
    electron afterpulses land up to the longest delay after their S2 and merge into the clusters that follow
    (rawdata.py:133-140), so shards are cut only at gaps larger than rext + that delay."""
    n = len(cluster)
    if n == 0:
        return np.zeros(world_size + 1, dtype=np.int64)
    w = np.cumsum(np.asarray(weight, dtype=np.float64))
    total = w[-1]
    starts = np.concatenate([[0], np.where(np.diff(cluster) != 0)[0] + 1])      # first instruction of every cluster
    if key is not None and min_gap is not None:
        key = np.asarray(key)
        ok = np.concatenate([[True], (key[starts[1:]] - key[starts[1:] - 1]) > min_gap])
        starts = starts[ok]
    bounds = [0]
    for r in range(1, world_size):
        target = total * r / world_size
        i = int(np.searchsorted(w, target))                     # instruction where the cumulative weight crosses
        j = int(starts[np.searchsorted(starts, i, side='left')]) if np.searchsorted(starts, i, side='left') < len(starts) else n
        bounds.append(max(j, bounds[-1]))
    bounds.append(n)
    return np.asarray(bounds, dtype=np.int64)


def safe_cut_gap(config):
    """Gap of the scheduler key behind which a new shard may start.  Consecutive clusters (gap > rext) still share a
    digitise window when a pulse of the earlier one is not over ``rext`` before the next one starts
    (``min key - last_pulse_end_time > rext``, rawdata.py:96-98), so a cut needs rext plus the longest a signal lasts past
    its key: the S1 / S2 photon delays (tables end below ~40 lifetimes: 10 us), PMT afterpulse delays, and the reach of
    the electron afterpulses.  ``simulate_sharded`` verifies afterwards that no window crossed a cut."""
    reach = 20_000.0
    sw = afterpulse_switches(config)
    if sw['pmt'] and config.get('uniform_to_pmt_ap'):
        for el in config['uniform_to_pmt_ap'].values():
            reach = max(reach, 20_000.0 + float(el['delaytime_bin_size']) * np.asarray(el['delaytime_cdf']).shape[-1])
    if sw['electron']:
        if config.get('uniform_to_ele_ap') is None:
            raise ValueError('enable_electron_afterpulses (default: on, rawdata.py:194) needs uniform_to_ele_ap (the delay-time histogram)')
        h = config['uniform_to_ele_ap']
        reach += (float(h[1][-1]) if isinstance(h, (tuple, list)) else float(h.bin_edges[-1])) + 2 * config['drift_time_gate']
    if sw['gate']:
        reach += config['photoelectric_t_center'] + 2 * config['drift_time_gate'] + 6 * config['photoelectric_t_spread']
    return float(config['right_raw_extension']) + reach


def shard_plan(config, instructions, world_size):
    """Who simulates what, computed by every rank on its own from the same instructions (no communication): the scheduler's
    order, and per rank its slice ``[bounds[r], bounds[r + 1])`` of the sorted instructions, the run-wide ids of that slice and the
    time range its strax chunks cover.  Ranges meet at ``first key of the next shard - right_raw_extension``: every window of a
    shard ends before it (``safe_cut_gap``; checked by the consumers), so per-rank chunk streams concatenate into one contiguous,
    time-ordered stream (sharded delivery)."""
    from .scheduler import schedule
    order, key, cluster = schedule(instructions, config)
    s_ins = instructions[order]
    weight = np.where(s_ins['type'] == 1, s_ins['amp'] * 0.15, s_ins['amp'] * float(config.get('s2_secondary_sc_gain', 30)))
    b = shard_clusters(cluster, weight, world_size, key=key, min_gap=safe_cut_gap(config))
    rext = int(config['right_raw_extension'])
    # rank r's chunks start at starts[r] (None: before its first instruction, as a single process does) and end at ends[r] (None: behind
    # its last window); an empty shard passes its range on to the next non-empty one
    starts, ends = [None] * world_size, [None] * world_size
    nonempty = [r for r in range(world_size) if b[r + 1] > b[r]]
    for a, nxt in zip(nonempty[:-1], nonempty[1:]):
        cut = int(key[b[nxt]]) - rext
        ends[a], starts[nxt] = cut, cut
    return dict(order=order, key=key, cluster=cluster, sorted_instructions=s_ins, bounds=b, starts=starts, ends=ends)


def gather_records(records, dst=0, group=None, async_op=False):
    """Variable-length gather of packed records (uint8 tensor of n * 244 bytes per rank, on the backend's device).

    Returns the list of per-rank tensors on ``dst`` (rank order = time order), ``None`` elsewhere.  With
    ``async_op=True`` returns ``(buffers_or_None, work_handles)``: the payload transfers are only posted, call
    ``wait_gather`` before touching the buffers (lets the transfer overlap the next batch's kernels)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = torch.tensor([records.numel()], dtype=torch.int64, device=records.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    bufs, ops = None, []
    if rank == dst:
        bufs = [records if r == dst else torch.empty(sizes[r], dtype=records.dtype, device=records.device) for r in range(world)]
        ops = [dist.P2POp(dist.irecv, bufs[r], r, group) for r in range(world) if r != dst and sizes[r] > 0]
    elif sizes[rank] > 0:
        ops = [dist.P2POp(dist.isend, records, dst, group)]
    # ONE grouped call (ncclGroupStart/End under RCCL): the root's receives run concurrently over all its xGMI links;
    # separate irecv calls would queue behind each other on the communicator's stream
    reqs = dist.batch_isend_irecv(ops) if ops else []
    if async_op:
        return bufs, (reqs, records)          # keep the send buffer alive until the transfer is done
    wait_gather(reqs)
    return bufs


def wait_gather(handles):
    reqs = handles[0] if isinstance(handles, tuple) else handles
    for q in reqs:
        q.wait()


def gather_batches(tensors, dst=0, device=None, group=None):
    """Gathers a stream of per-batch record tensors from every rank on ``dst``: rank ``dst`` gets the list of non-empty
    tensors in (rank, batch) order -- shard order is time order, and so is the batch order inside a shard --, the others
    ``None``.  The ranks may have different numbers of batches: all run the same number of rounds (one small all-reduce
    per round decides whether anybody has a batch left; a rank that is out sends nothing), and the transfer of round k is
    only waited for after the producer has delivered batch k + 1, so that it overlaps that batch's kernels."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    it = iter(tensors)
    rounds, pending = [], None
    while True:
        t = next(it, None)
        more = torch.tensor([0 if t is None else 1], dtype=torch.int32, device=device)
        dist.all_reduce(more, op=dist.ReduceOp.MAX, group=group)
        if pending is not None:
            wait_gather(pending)
            pending = None
        if int(more.item()) == 0:
            break
        if t is None:
            t = torch.empty(0, dtype=torch.uint8, device=device)
        bufs, pending = gather_records(t, dst=dst, group=group, async_op=True)
        if rank == dst:
            rounds.append(bufs)
    if rank != dst:
        return None
    return [rounds[k][r] for r in range(world) for k in range(len(rounds)) if rounds[k][r].numel()]


def simulate_sharded(config, instructions, device=None, dst=0, max_batch_quanta=None, gather=True):
    """All ranks call this with the same instructions; rank ``dst`` gets the time-ordered raw_records of the whole
    run (numpy structured array), the others ``None``.  One process per GPU (``LOCAL_RANK`` picks the device).

    Every rank simulates its shard batch by batch; the packed records of a batch go from the engine's arena into a device
    tensor and from there over RCCL to ``dst`` (no host hop), the transfer of batch k overlapping the kernels of batch
    k + 1 -- the path ``bench.py --gpus N`` times.  Ranks run a common number of rounds (a rank that is out of batches
    sends nothing).  Only ``dst`` moves records to the host, once, at the end: on an 8-GPU node that is 8 shards through ONE
    PCIe link.  ``gather=False``: no record leaves its rank -- every rank gets the records of ITS time range (ranges are
    contiguous in time and disjoint: concatenated in rank order they are the run); what a consumer that can write per-rank
    output should use.  The check that no digitise window reaches into the next shard is made either way."""
    import os
    import torch
    import torch.distributed as dist
    from .dtypes import raw_record_dtype
    from .rawdata import RawData
    rank, world = dist.get_rank(), dist.get_world_size()
    device = int(os.environ.get('LOCAL_RANK', rank)) if device is None else device
    on_gpu = dist.get_backend() == 'nccl'
    plan = shard_plan(config, instructions, world)
    order, key, s_ins, b = plan['order'], plan['key'], plan['sorted_instructions'], plan['bounds']
    mine = s_ins[b[rank]:b[rank + 1]]
    rd = RawData(config, device=device)
    # run-wide instruction ids keep the RNG streams independent of the sharding
    rd.global_ids = order[b[rank]:b[rank + 1]]
    rd.engine.set_record_order(False)
    if max_batch_quanta is not None:
        rd.max_batch_quanta = int(max_batch_quanta)
    coll = torch.device('cuda', device) if on_gpu else torch.device('cpu')
    # RCCL: the records stay on the device; gloo (CPU collectives: tests, rehearsals on one card): host arrays
    batches = rd.iter_batches(mine, device_records=on_gpu) if len(mine) else iter(())
    last_end = np.iinfo(np.int64).min

    def tensors():
        nonlocal last_end
        for batch in batches:
            if len(batch['right']):
                last_end = max(last_end, (int(np.max(batch['right'])) - config['trigger_window']) * config['sample_duration'])
            yield batch['records'] if on_gpu else torch.from_numpy(np.ascontiguousarray(batch['records']).view(np.uint8).copy())
    if gather:
        parts = gather_batches(tensors(), dst=dst, device=coll)
    else:
        parts = [t for t in tensors() if t.numel()]
    # no digitise window may reach into the next shard (rawdata.py:96-98 applied across the cut)
    ends = [None] * world
    dist.all_gather_object(ends, int(last_end))
    run_end = np.iinfo(np.int64).min
    for r in range(world - 1):
        run_end = max(run_end, ends[r])
        if b[r + 1] < len(key) and b[r + 1] > b[r] and not (key[b[r + 1]] - run_end > config['right_raw_extension']):
            raise RuntimeError(f'a digitise window of shard {r} reaches into shard {r + 1}: the result would depend on the sharding')
    if gather and rank != dst:
        return None
    if not parts:
        return np.zeros(0, dtype=raw_record_dtype())
    return torch.cat(parts).cpu().numpy().view(raw_record_dtype())
