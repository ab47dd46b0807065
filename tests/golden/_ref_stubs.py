"""Stub modules that let the reference's hot-path modules be imported in the build container.

Only used by ``make_golden.py`` (fixture generation, run once in the build container where
``/root/reference`` exists).  Nothing here is product code and nothing here is needed on the GPU box.

The reference needs ``strax``, ``straxen``, ``numba`` (none installed, no network).  The stubs give
identity ``njit`` -- the mode the reference's own coverage CI exercises (``NUMBA_DISABLE_JIT=1``,
/root/reference/.github/workflows/pytest.yml:82) -- so that every random draw comes from numpy's
global generator and a seed makes a run reproducible.
"""
import hashlib
import importlib
import sys
import types

import pandas as pd

REFERENCE_ROOT = '/root/reference'


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _exporter(export_self=False):
    names = []

    def export(obj):
        names.append(obj.__name__)
        return obj
    return export, names


def _njit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]
    return lambda f: f


class _Signature:
    def __call__(self, *a, **k):
        return self

    def __getitem__(self, item):
        return self


class _NoProgressBar:
    def __init__(self, *a, **k):
        pass

    def update(self, n):
        pass

    def close(self):
        pass


def _deterministic_hash(obj):
    if isinstance(obj, dict):
        obj = sorted((k, repr(v)) for k, v in obj.items())
    return hashlib.sha1(repr(obj).encode()).hexdigest()


def _get_resource(path, fmt='csv'):
    if fmt == 'csv':
        return pd.read_csv(path)
    if fmt == 'npy':
        import numpy as np
        return np.load(path)
    if fmt == 'json.gz':
        import gzip
        import json
        with gzip.open(path, 'rt') as f:
            return json.load(f)
    raise NotImplementedError(fmt)


def import_reference():
    """Returns a namespace with the reference's hot-path modules."""
    _module('numba', njit=_njit, jit=_njit, int32=_Signature(), int64=_Signature())
    strax = _module('strax', exporter=_exporter, deterministic_hash=_deterministic_hash)
    strax.utils = _module('strax.utils', tqdm=_NoProgressBar)
    _module('straxen', get_resource=_get_resource)
    pkg = types.ModuleType('wfsim')
    pkg.__path__ = [REFERENCE_ROOT + '/wfsim']      # package shell: wfsim/__init__.py is not executed
    sys.modules['wfsim'] = pkg
    ns = types.SimpleNamespace()
    for name in ['units', 'load_resource', 'utils', 'core.pulse', 'core.s1', 'core.s2',
                 'core.afterpulse', 'core.rawdata']:
        setattr(ns, name.split('.')[-1], importlib.import_module('wfsim.' + name))
    return ns


def import_reference_interface():
    """The reference's strax_interface module (ChunkRawRecords, strax_interface.py:354-504) on top of ``import_reference``.

    strax / straxen are not installed: ``strax.raw_record_dtype`` / ``sort_by_time`` / ``DEFAULT_RECORD_LENGTH`` come from the
    repo's RECALLED restatement (wfsim_amd/ministrax.py), the plugin machinery (``Plugin``, ``Option``, ``takes_config``,
    ``straxen.URLConfig``) from empty shells.  The chunk DECISIONS this is used to pin (chunk bounds, which records and truth
    rows go into which chunk) do not depend on the record layout; the fixtures store record fields, not record bytes."""
    import numpy as np
    ns = import_reference()
    import importlib.util
    import os
    # (by file: importing the wfsim_amd package would pick up the strax shell registered above)
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location('_dtypes_for_stubs', os.path.join(here, '..', '..', 'wfsim_amd', 'dtypes.py'))
    dtypes = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(dtypes)

    def sort_by_time(x):            # strax.sort_by_time: by time, ties by channel (as wfsim_amd/ministrax.py)
        return x if len(x) == 0 else x[np.lexsort((x['channel'], x['time']))]

    class immutabledict(dict):
        pass
    _module('immutabledict', immutabledict=immutabledict)
    _module('uproot')
    strax = sys.modules['strax']

    class Plugin:
        def __init__(self, *a, **k):
            pass

    class Option:
        def __init__(self, *a, **k):
            pass
    strax.Plugin, strax.Option = Plugin, Option
    strax.takes_config = lambda *opts: (lambda cls: cls)
    strax.raw_record_dtype = dtypes.raw_record_dtype
    strax.DEFAULT_RECORD_LENGTH = dtypes.DEFAULT_RECORD_LENGTH
    strax.sort_by_time = sort_by_time
    straxen = sys.modules['straxen']
    straxen.tpc_r, straxen.tpc_z, straxen.n_tpc_pmts = 66.4, 148.6, 494
    straxen.URLConfig = lambda *a, **k: None
    pkg = sys.modules['wfsim']
    pkg.RawData, pkg.RawDataOptical = ns.rawdata.RawData, ns.rawdata.RawDataOptical
    pkg.load_config = ns.load_resource.load_config
    pkg.optical_adjustment = ns.utils.optical_adjustment
    ns.strax_interface = importlib.import_module('wfsim.strax_interface')
    return ns
