"""The C-ABI shared library: builds for gfx950 (hipcc cross-compiles without a GPU), loads, and
exports every symbol that include/refvsr_hip.h declares and refvsr_amd/hip.py binds.  No compute
calls here (no GPU in the build container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def libpath():
    from refvsr_amd import hip
    if not os.path.exists(hip.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return hip.LIB_PATH


def _declared():
    src = open(os.path.join(ROOT, 'include', 'refvsr_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(refvsr_[a-z0-9_]+)\s*\(', src)))


def test_header_binding_and_library_agree(libpath):
    from refvsr_amd import hip
    declared = _declared()
    assert declared == sorted(hip.EXPORTS), set(declared) ^ set(hip.EXPORTS)
    h = ctypes.CDLL(libpath)
    for name in declared:
        assert hasattr(h, name), 'library does not export ' + name


def test_abi_version_and_error_channel(libpath):
    from refvsr_amd import hip
    h = hip.lib()
    assert h.refvsr_abi_version() == hip.ABI_VERSION
    # argument validation runs before any device work: a null descriptor is rejected with a message
    rc = h.refvsr_conv_mfma(None, None)
    assert rc != 0 and b'null descriptor' in h.refvsr_last_error()
    with pytest.raises(RuntimeError, match='null descriptor'):
        hip.check(rc, 'conv_mfma')


def test_conv_descriptor_layout_matches_header():
    """Field order / types of the ctypes mirror follow `struct RefvsrConv` in the header."""
    from refvsr_amd import hip
    src = open(os.path.join(ROOT, 'include', 'refvsr_hip.h')).read()
    body = src[src.index('typedef struct RefvsrConv {'):src.index('} RefvsrConv;')]
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S).split('{', 1)[1]
    names = []
    for decl in body.split(';'):
        decl = decl.strip()
        if not decl:
            continue
        for part in decl.split(','):
            names.append(part.replace('*', ' ').split()[-1])
    assert names == [f[0] for f in hip.RefvsrConv._fields_]


def test_packing_contract_matches_the_library(libpath):
    """The K-block order is computed twice -- packing.py on the host side of the boundary, rv_kslot inside the kernels.
    The library exports its closed form as host functions so that the two are pinned against each other without a GPU."""
    from refvsr_amd import hip
    from refvsr_amd.packing import kslot, ksteps
    h = hip.lib()
    for ks in (1, 3, 5, 7):
        for ncg in range(1, 18):
            assert h.refvsr_ksteps(ks, ncg) == ksteps(ks, ncg)
            slots = set()
            for ty in range(ks):
                for tx in range(ks):
                    for cg in range(ncg):
                        j = h.refvsr_kslot(ty, tx, cg, ks, ncg)
                        assert j == kslot(ty, tx, cg, ks, ncg)
                        slots.add(j)
            assert len(slots) == ks * ks * ncg and max(slots) < 4 * ksteps(ks, ncg)
