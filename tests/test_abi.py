"""The C-ABI library loads and exports every symbol include/pnl_hip.h declares; host-side error paths that do
not need a GPU behave as documented.  No compute calls here (CPU-only container)."""
import ctypes
import os
import re
import numpy as np
import pytest
from pynucleus_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, 'include', 'pnl_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(pnl_[a-z_0-9]+)\s*\(', txt)))


def test_header_and_binding_agree():
    assert sorted(_lib.EXPORTS) == declared_symbols()


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), 'build the HIP library first (__graft_entry__.build())'
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(L, name), name
    assert b'gfx950' in _lib.load().pnl_version()


def test_create_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is visible')
    with pytest.raises(_lib.PnlError):
        _lib.Context(0)


def test_builder_has_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is visible')
    from pynucleus_amd import disc, P1_DoFMap, PHYSICAL, getFractionalKernel
    from pynucleus_amd.builder import nonlocalBuilder
    mesh = disc(1)
    b = nonlocalBuilder(P1_DoFMap(mesh, PHYSICAL), getFractionalKernel(2, 0.5))
    with pytest.raises(_lib.PnlError):
        b.getDense()


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, 'pynucleus_amd')
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(dirpath, fn)).read()
                assert 'import oracle' not in src and 'from oracle' not in src and 'nl_oracle' not in src, fn


@pytest.mark.gpu
def test_work_list_overflow_fails_the_call():
    """a work list that is too small (the option PNL_WL_FRAC shrinks it to 64 entries per pass) must not pass silently: the status
    of pnl_synchronize / pnl_get_counters says the operator is incomplete (the check INTEGRATION.md's stub performs)"""
    from pynucleus_amd import disc, P1_DoFMap, P2_DoFMap, PHYSICAL, getFractionalKernel
    from pynucleus_amd.builder import nonlocalBuilder
    from pynucleus_amd._lib import PnlError
    mesh = disc(4)
    try:
        for DoFMap in (P1_DoFMap, P2_DoFMap):
            dm = DoFMap(mesh, PHYSICAL)
            _lib.set_option('PNL_WL_FRAC', None)
            A = nonlocalBuilder(dm, getFractionalKernel(2, 0.75), {}).getDense()
            assert np.isfinite(A.toarray()).all()
            _lib.set_option('PNL_WL_FRAC', '1e-12')
            with pytest.raises(PnlError, match='work list overflow'):
                nonlocalBuilder(dm, getFractionalKernel(2, 0.75), {}).getDense()
    finally:
        _lib.set_option('PNL_WL_FRAC', None)


def test_options_are_explicit_not_environment(monkeypatch):
    """the product library reads no environment variable on the assembly path: options exist only through pnl_set_option, and a
    product build refuses the A/B switches of the tuning builds"""
    L = _lib.load()
    if b'tuning' in L.pnl_version():
        pytest.skip('tuning build')
    assert L.pnl_set_option(b'PNL_NO_POWTAB', b'1') == 0
    assert L.pnl_set_option(b'PNL_NO_POWTAB', None) == 0
    assert L.pnl_set_option(b'PNL_NO_SLOT', b'1') == _lib.PNL_ERR_UNSUPPORTED
    with pytest.raises(_lib.PnlError):
        _lib.set_option('PNL_UNI_PER_CU', 1)
    # the planner's thread count is an option like the others
    assert L.pnl_set_option(b'PNL_PLAN_THREADS', b'2') == 0
    assert L.pnl_set_option(b'PNL_PLAN_THREADS', None) == 0
    # no getenv with a literal name anywhere in the product sources (pnl_tune's fallback of the tuning builds takes a variable)
    import glob
    for fn in glob.glob(os.path.join(ROOT, 'pynucleus_amd', 'csrc', '*.h*')):
        src = open(fn).read()
        calls = re.findall(r'\bgetenv\("([A-Z_0-9]+)"\)', src)
        assert not calls, (fn, calls)
