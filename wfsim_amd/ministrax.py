"""The few pieces of strax the simulator plugins touch, for environments without strax (this container, the GPU box).

strax (>= 1.6.0) is a third-party dependency of the reference that is not vendored in /root/reference and is not
installed here.  When it is importable the plugin module uses the real thing; otherwise these stand-ins, written to
strax's published behaviour, carry the plugin life cycle (setup / is_ready / compute / source_finished) so that the
reference's config 1 ("100 S1 instructions through RawRecordsFromFaxNT") can run end to end.
"""
import numpy as np

from .dtypes import raw_record_dtype, DEFAULT_RECORD_LENGTH  # noqa: F401


def sort_by_time(x):
    """strax.sort_by_time: sort by time, ties by channel."""
    if len(x) == 0:
        return x
    if 'channel' in x.dtype.names:
        return x[np.lexsort((x['channel'], x['time']))]
    return x[np.argsort(x['time'], kind='stable')]


class Chunk:
    def __init__(self, start, end, data, data_type, run_id=None):
        if len(data):
            assert data['time'].min() >= start, 'chunk data starts before the chunk'
        self.start, self.end, self.data, self.data_type, self.run_id = start, end, data, data_type, run_id

    def __len__(self):
        return len(self.data)


class Plugin:
    """Life cycle of a strax source plugin (depends_on = ()): setup(), then alternately is_ready / compute until
    source_finished."""
    provides = tuple()
    depends_on = tuple()
    run_id = '000000'

    def __init__(self, config=None, run_id=None):
        self.config = dict(config or {})
        if run_id is not None:
            self.run_id = run_id

    def chunk(self, *, start, end, data, data_type=None, run_id=None):
        # strax refuses a chunk whose data is not of the dtype the plugin declared (strax.Plugin.chunk -> Chunk.__init__)
        if data_type is not None and hasattr(self, 'infer_dtype'):
            want = self.infer_dtype()
            want = want[data_type] if isinstance(want, dict) else want
            assert np.dtype(data.dtype) == np.dtype(want), f'{data_type}: data of dtype {data.dtype}, declared {np.dtype(want)}'
        return Chunk(start=start, end=end, data=data, data_type=data_type, run_id=run_id or self.run_id)

    def setup(self):
        pass


def run_plugin(plugin, max_chunks=10 ** 6):
    """What strax's mailbox loop does for a source plugin: returns {data_type: [Chunk, ...]}."""
    plugin.setup()
    out = {k: [] for k in plugin.provides}
    chunk_i = 0
    for _ in range(max_chunks):
        if not plugin.is_ready(chunk_i):
            if plugin.source_finished():
                break
            continue
        result = plugin.compute()
        for k in plugin.provides:
            out[k].append(result[k])
        chunk_i += 1
    return out


def get_array(plugin, target):
    chunks = run_plugin(plugin)[target]
    if not chunks:
        return np.zeros(0, dtype=raw_record_dtype())
    return np.concatenate([c.data for c in chunks])
