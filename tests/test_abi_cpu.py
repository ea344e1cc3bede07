"""The C-ABI library loads and exports every symbol include/lcmi.h declares, and the Python prototype
table covers exactly that set.  No compute calls: there is no GPU here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'lcmi.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(lc_[a-z0-9_]+)\s*\(', text)))


def test_header_symbols_are_exported_and_prototyped():
    from lightcurver_amd import _lib
    syms = _declared_symbols()
    assert len(syms) >= 45
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(handle, s), f'{s} declared in lcmi.h but not exported by liblcmi.so'
    assert sorted(_lib.SIGNATURES) == syms
    assert _lib.lib().lc_version() >= 100


def test_no_cpu_fallback_without_gpu():
    from lightcurver_amd import _lib
    lib = _lib.lib()
    h = ctypes.c_void_p()
    rc = lib.lc_ctx_create(0, ctypes.byref(h))
    if rc == 0:
        lib.lc_ctx_destroy(h)
        pytest.skip('a GPU is visible here')
    assert rc == -2 and b'device' in lib.lc_last_error(None).lower()
    with pytest.raises(_lib.LcError):
        _lib.Context(0)
    # stamp-size queries need no device
    assert lib.lc_psf_supported(32, 2) == 1 and lib.lc_psf_supported(33, 2) == 0
    assert lib.lc_joint_supported(64, 2) == 1 and lib.lc_joint_supported(128, 2) == 1 and lib.lc_joint_supported(100, 2) == 0
    assert all(lib.lc_joint_supported(n, 2) == 1 for n in (16, 24, 32, 40, 48, 56, 64, 128))


def test_missing_library_fails_loudly(monkeypatch):
    from lightcurver_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/liblcmi.so')
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        _lib.lib()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'lightcurver_amd')
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.h', '.hip')):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f


def test_embedded_fit_sizes_are_chosen_from_the_kernel_table():
    """build_psf fits a stamp size without a kernel of its own inside the next instantiated one (psf_routines._fit_size);
    lc_psf_supported is a host-side table look-up, so the choice can be checked without a GPU."""
    import pytest
    from lightcurver_amd import _lib
    from lightcurver_amd.starred.procedures.psf_routines import _fit_size
    l = _lib.lib()
    native = [n for n in range(2, 130, 2) if l.lc_psf_supported(n, 2)]
    assert native == [16, 24, 32, 64]
    for n in range(8, 66, 2):
        m = _fit_size(n, 2)
        assert m in native and m >= n and (m - n) % 2 == 0 and (m == n) == (n in native)
        assert all(k < n for k in native if k < m)          # the smallest one that holds the stamp
    with pytest.raises(_lib.LcError):
        _fit_size(66, 2)
    with pytest.raises(_lib.LcError):
        _fit_size(20, 3)


def test_embedded_joint_fit_sizes_are_chosen_from_the_kernel_table():
    """The same for the joint fit (lightcurver_amd.joint.joint_fit_size): lc_joint_supported is a host-side table look-up."""
    import pytest
    from lightcurver_amd import _lib
    from lightcurver_amd.joint import joint_fit_size
    l = _lib.lib()
    native = [n for n in range(2, 130, 2) if l.lc_joint_supported(n, 2)]
    assert native == [16, 24, 32, 40, 48, 56, 64, 128]
    for n in range(10, 130, 2):
        m = joint_fit_size(n, 2)
        assert m in native and m >= n and (m - n) % 2 == 0 and (m == n) == (n in native)
        assert all(k < n for k in native if k < m)
    with pytest.raises(_lib.LcError):
        joint_fit_size(130, 2)
    with pytest.raises(_lib.LcError):
        joint_fit_size(21, 2)           # odd sizes have no embedding: the stamp centre would move by half a pixel
