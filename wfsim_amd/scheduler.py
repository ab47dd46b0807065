"""Host-side event scheduler: the part of RawData.__call__ that orders and clusters instructions.

/root/reference/wfsim/core/rawdata.py:56-63: primary instructions are ordered by ``time - z/v`` for S2-like
types (the drift delay) and split into time clusters wherever the gap exceeds ``right_raw_extension``.  Clusters
are the independent units of the simulation: they are what a batch is made of and what is sharded across GPUs
(SURVEY.md 8e).  Which clusters end up in one digitise window additionally depends on the simulated pulse end
times (rawdata.py:96-98) and is decided on the GPU (kernel k_groups).
"""
import numpy as np

from .physics import instruction_time


def schedule(instructions, config):
    """Returns (order, key, cluster): ``instructions[order]`` is sorted by the scheduler key ``key`` (already
    ordered), and ``cluster`` is the non-decreasing cluster index of every sorted instruction."""
    key = instruction_time(instructions, config)
    order = np.argsort(key, kind='stable')
    key = key[order]
    rext = config['right_raw_extension']
    new_cluster = np.diff(key) > rext
    cluster = np.concatenate([[0], np.cumsum(new_cluster)]).astype(np.int32) if len(key) else np.zeros(0, np.int32)
    return order, key.astype(np.int64), cluster


def processing_order(instructions, order, cluster):
    """Order in which the reference simulates the sorted instructions: cluster by cluster, S1s first, then S2s
    (rawdata.py:102-105); returns indices into the *sorted* array."""
    typ = instructions['type'][order]
    rank = np.searchsorted(np.array([1, 2, 4, 6]), typ)
    return np.lexsort((np.arange(len(typ)), rank, cluster)).astype(np.int64)


def run_sets(s_ins, key, cluster, config):
    """Pulse set ("run set") of every sorted instruction: the instructions the reference hands to one Pulse call
    (rawdata.py:106-127).  With ``save_full_truth`` (the default) every instruction is its own set; without it S1s
    whose keys are at most 100 ns apart and S2s at most ``int(0.2 / v)`` ns apart (2 mm of drift) share a set.
    Sets are numbered in processing order (cluster, S1 sets, then S2 sets).  Returns (run_set int32[n], n_sets)."""
    n = len(s_ins)
    if n == 0:
        return np.zeros(0, dtype=np.int32), 0
    full = config.get('save_full_truth', True)
    gap = np.array([100, int(0.2 / config['drift_velocity_liquid']), 0, 0], dtype=np.int64)      # S1, S2; types 4 / 6: one set per cluster
    po = processing_order(s_ins, np.arange(n), cluster)               # cluster by cluster, S1s, S2s, type 4, type 6, key order inside
    rank = np.searchsorted(np.array([1, 2, 4, 6]), s_ins['type'][po])
    cl, k = np.asarray(cluster)[po], np.asarray(key)[po]
    new_group = np.concatenate([[True], (np.diff(cl) != 0) | (np.diff(rank) != 0)])
    dk = np.concatenate([[0], np.diff(k)])
    new_set = new_group | ((rank < 2) & (full | (dk > gap[rank])))
    out = np.zeros(n, dtype=np.int32)
    out[po] = (np.cumsum(new_set) - 1).astype(np.int32)
    return out, int(new_set.sum())


FORCED_BREAK_KEY = np.int64(2 ** 62)


def feedback_schedule(instructions, parent, config):
    """The reference's scheduler loop with electron-afterpulse feedback (rawdata.py:70-151), replayed on the host for a
    run whose secondaries are already known.

    ``instructions``: primaries followed by secondaries; ``parent[i]``: for a secondary the index of the instruction
    whose pulse set produced it (the first instruction of that set), -1 for primaries.  The loop: one cluster of
    primaries is added to the buffer per pass, the buffer is re-clustered (gap > rext), clusters are simulated in time
    order up to and including the first one that holds a primary; a secondary enters the buffer when its parent's set
    has been simulated, i.e. it is seen from the NEXT pass on.  A cluster without primaries is digitised right after
    it was simulated (rawdata.py:148-149), the others when the next pass starts and finds a gap (rawdata.py:96-98).

    Returns (order, key, cluster, run_set): ``instructions[order]`` in processing order (cluster by cluster, key
    order inside), ``cluster`` the non-decreasing dynamic cluster index, ``key`` the scheduler key except that the
    members of a cluster that is always preceded by a digitisation carry FORCED_BREAK_KEY (the device's window rule
    ``min key - last pulse end > rext`` then fires whenever a pulse exists), ``run_set`` the pulse set of every
    instruction (numbered in processing order)."""
    from .physics import instruction_time
    n = len(instructions)
    key = instruction_time(instructions, config).astype(np.int64)
    rext = config['right_raw_extension']
    typ = instructions['type']
    prim = np.where(parent < 0)[0]
    prim = prim[np.argsort(key[prim], kind='stable')]
    cuts = np.where(np.diff(key[prim]) > rext)[0] + 1
    queue = [q for q in np.split(prim, cuts)] if len(prim) else []
    children = {}
    for i in np.where(parent >= 0)[0]:
        children.setdefault(int(parent[i]), []).append(int(i))
    full = config.get('save_full_truth', True)
    gaps = {1: 100, 2: int(0.2 / config['drift_velocity_liquid'])}
    buf = []
    order, cluster, eff_key, run_set = [], [], [], []
    n_cl = n_set = 0
    while queue or buf:
        if queue:
            buf.extend(queue.pop(0).tolist())
        b = np.asarray(buf, dtype=np.int64)
        b = b[np.argsort(key[b], kind='stable')]
        cls = np.split(b, np.where(np.diff(key[b]) > rext)[0] + 1)
        first = True
        for cl in cls:
            has_prim = bool(np.any(typ[cl] <= 2))
            released = []
            for ptype in (1, 2, 4, 6):
                idx = cl[typ[cl] == ptype]
                if len(idx) == 0:
                    continue
                if ptype in gaps:
                    new = np.ones(len(idx), dtype=bool) if full else np.concatenate([[True], np.diff(key[idx]) > gaps[ptype]])
                else:
                    new = np.concatenate([[True], np.zeros(len(idx) - 1, dtype=bool)])
                sid = n_set + np.cumsum(new) - 1
                n_set += int(new.sum())
                order.extend(idx.tolist()); run_set.extend(sid.tolist()); cluster.extend([n_cl] * len(idx))
                eff_key.extend((key[idx] if first else np.full(len(idx), FORCED_BREAK_KEY)).tolist())
                if ptype == 2:                           # secondaries hang on the first instruction of their parent set
                    for i0 in idx[new]:
                        released.extend(children.get(int(i0), []))
            members = set(cl.tolist())
            buf = [i for i in buf if i not in members] + released
            n_cl += 1
            first = False
            if has_prim:
                break
    assert len(order) == n, 'every instruction is scheduled exactly once'
    return (np.asarray(order, dtype=np.int64), np.asarray(eff_key, dtype=np.int64), np.asarray(cluster, dtype=np.int32),
            np.asarray(run_set, dtype=np.int32))
