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
