"""The C-ABI library loads on a CPU-only machine and exports every symbol include/wfsim_amd.h declares; compute entry
points are not called here (no GPU).  Also: the product refuses to run without a GPU instead of falling back."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    txt = open(os.path.join(ROOT, 'include', 'wfsim_amd.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(wfs_[a-z_0-9]+)\s*\(', txt)))


def test_library_exports_every_declared_symbol():
    from wfsim_amd import engine
    lib = engine.load_library()
    names = _declared_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/wfsim_amd.h but not exported by libwfsim_amd.so'
    assert set(engine.EXPORTS) == set(names)


def test_struct_layouts_match_header():
    from wfsim_amd.engine import WfsConfig, WfsCounts
    txt = open(os.path.join(ROOT, 'include', 'wfsim_amd.h')).read()
    body = re.search(r'typedef struct wfs_config \{(.*?)\} wfs_config;', txt, flags=re.S).group(1)
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    fields = []
    for decl in body.split(';'):
        decl = decl.strip()
        if not decl:
            continue
        typ, names = decl.split(None, 1)
        fields += [(n.strip(), typ) for n in names.split(',')]
    assert [f[0] for f in fields] == [f[0] for f in WfsConfig._fields_]
    ctype = {'int32_t': ctypes.c_int32, 'double': ctypes.c_double, 'uint64_t': ctypes.c_uint64}
    assert [ctype[f[1]] for f in fields] == [f[1] for f in WfsConfig._fields_]
    assert ctypes.sizeof(WfsCounts) == 11 * 8


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    import wfsim_amd
    from wfsim_amd.engine import WfsError
    with pytest.raises(WfsError):
        wfsim_amd.RawData(wfsim_amd.xenonnt_test_config())


def test_product_does_not_import_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'wfsim_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', src, flags=re.M), f
                assert 'liboracle' not in src and 'wfsim_oracle' not in src.replace('oracle/wfsim_oracle.c', ''), f
