"""CPU suite: the C-ABI library loads and exports every symbol include/lhvi.h declares; the product has no route
into the oracle and fails loudly without a GPU."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'lifted-hybrid-variational-inference_amd')


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'lhvi.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(lhvi_[a-z0-9_]+)\s*\(', text)))


@pytest.fixture(scope='module')
def built():
    lib = os.path.join(PKG, 'csrc', 'liblhvi.so')
    if not os.path.exists(lib):
        import __graft_entry__
        __graft_entry__.build()
    return lib


def test_header_symbols_are_exported(built):
    import ctypes
    handle = ctypes.CDLL(built)
    syms = declared_symbols()
    assert len(syms) >= 25
    for name in syms:
        assert hasattr(handle, name), 'liblhvi.so does not export %s' % name
    # one version number in three places: the header, the library built from it, the Python binding's struct layouts
    header = open(os.path.join(ROOT, 'include', 'lhvi.h')).read()
    declared = int(re.search(r'#define\s+LHVI_ABI_VERSION\s+(\d+)', header).group(1))
    from lhvi import _abi
    assert handle.lhvi_version() == declared == _abi.ABI_VERSION


def test_python_binding_covers_header(built):
    from lhvi import _abi
    assert sorted(_abi.SIGNATURES) == declared_symbols()
    _abi.lib()
    assert _abi.MISSING == []
    assert _abi.lib().lhvi_strerror(-1) == b'invalid argument'


def test_structs_match_header_layout(built):
    """field order of the ctypes mirrors == field order in lhvi.h (the structs cross the ABI by pointer)"""
    from lhvi import _abi
    text = open(os.path.join(ROOT, 'include', 'lhvi.h')).read()
    for cname, struct in (('lhvi_graph', _abi.GraphStruct), ('lhvi_pots', _abi.PotsStruct), ('lhvi_pbp', _abi.PbpStruct),
                          ('lhvi_vi', _abi.ViStruct)):
        body = re.search(r'typedef struct %s \{(.*?)\} %s_t;' % (cname, cname), text, flags=re.S).group(1)
        body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
        names = []
        for decl in body.split(';'):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(','):
                names.append(re.findall(r'([A-Za-z_][A-Za-z0-9_]*)\s*$', part.strip())[0])
        assert names == [f[0] for f in struct._fields_], cname


def test_no_cpu_fallback_without_gpu():
    from conftest import has_gpu
    if has_gpu():
        pytest.skip('GPU present')
    from lhvi import _abi, synth
    from lhvi.gabp import GaBP
    g, _ = synth.gaussian_chain(6)
    with pytest.raises(_abi.LhviError):
        GaBP(g).run(3)


def test_product_never_imports_oracle():
    hits = subprocess.run(['grep', '-rIl', '-E', r'^\s*(from|import)\s+oracle', PKG], capture_output=True, text=True).stdout
    assert hits.strip() == '', hits
    hits = subprocess.run(['grep', '-rIl', 'liboracle', PKG], capture_output=True, text=True).stdout
    assert hits.strip() == '', hits
    # outside tests/ the oracle is named by __graft_entry__.py (build, smoke), bench.py (cpu_baseline) and the labelled CPU
    # baselines of scripts/bench_configs.py only; the random-instance soaks, which check against it, live under tests/soak/
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hits = subprocess.run(['grep', '-rIl', '-E', r'^\s*(from|import)\s+oracle|from oracle import', os.path.join(root, 'scripts')],
                          capture_output=True, text=True).stdout.split()
    assert [os.path.relpath(h, root) for h in hits] == ['scripts/bench_configs.py'], hits


def test_graft_entry_build_runs():
    """the driver's build check: __graft_entry__.build() compiles (or finds up to date) the library and the oracle and
    verifies the ABI version and the exported symbols"""
    import importlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    try:
        entry = importlib.import_module('__graft_entry__')
        entry.build()
    finally:
        sys.path.remove(root)
