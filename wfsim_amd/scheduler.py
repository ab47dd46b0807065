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
    out = []
    for c in np.unique(cluster):
        idx = np.where(cluster == c)[0]
        for ptype in (1, 2, 4, 6):
            out.extend(idx[typ[idx] == ptype].tolist())
    return np.asarray(out, dtype=np.int64)


def run_sets(s_ins, key, cluster, config):
    """Pulse set ("run set") of every sorted instruction: the instructions the reference hands to one Pulse call
    (rawdata.py:106-127).  With ``save_full_truth`` (the default) every instruction is its own set; without it S1s
    whose keys are at most 100 ns apart and S2s at most ``int(0.2 / v)`` ns apart (2 mm of drift) share a set.
    Sets are numbered in processing order (cluster, S1 sets, then S2 sets).  Returns (run_set int32[n], n_sets)."""
    n = len(s_ins)
    out = np.zeros(n, dtype=np.int32)
    full = config.get('save_full_truth', True)
    gaps = {1: 100, 2: int(0.2 / config['drift_velocity_liquid'])}
    typ = s_ins['type']
    k = 0
    bounds = np.concatenate([[0], np.where(np.diff(cluster) != 0)[0] + 1, [n]]) if n else np.zeros(1, dtype=np.int64)
    for a, b in zip(bounds[:-1], bounds[1:]):
        for ptype in (1, 2, 4, 6):
            idx = np.arange(a, b)[typ[a:b] == ptype]
            if len(idx) == 0:
                continue
            if full or ptype not in gaps:
                new = np.ones(len(idx), dtype=bool) if ptype in gaps else np.concatenate([[True], np.zeros(len(idx) - 1, dtype=bool)])
            else:
                new = np.concatenate([[True], np.diff(key[idx]) > gaps[ptype]])
            out[idx] = k + np.cumsum(new) - 1
            k += int(new.sum())
    return out, k
