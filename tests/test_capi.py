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


def test_argument_validation_precedes_device_work(libpath):
    """Error behaviour of the boundary: every entry point validates its arguments before touching the device and
    reports through refvsr_last_error() -- so the checks run (and are pinned) in the GPU-less container too."""
    from refvsr_amd import hip
    h = hip.lib()
    P = ctypes.c_void_p
    buf = (ctypes.c_char * 4096)()                    # host memory standing in for device pointers: never dereferenced
    p = ctypes.cast(buf, P)

    def expect(rc, text):
        assert rc != 0 and text in h.refvsr_last_error().decode(), h.refvsr_last_error()

    d = hip.RefvsrConv()
    d.src0, d.c0, d.h_in, d.w_in, d.h_out, d.w_out = p, 12, 8, 8, 8, 8       # 12 channels: not a 16-byte multiple
    expect(h.refvsr_conv_mfma(ctypes.byref(d), None), 'src0/c0 invalid (c0=12)')
    d.c0, d.ksize, d.stride, d.pad = 24, 9, 1, 4
    expect(h.refvsr_conv_mfma(ctypes.byref(d), None), 'bad geometry')
    d.ksize, d.pad, d.wpack, d.bias, d.out, d.mt_per_block, d.cout, d.out_c = 3, 1, p, p, p, 4, 24, 24
    expect(h.refvsr_conv_mfma(ctypes.byref(d), None), 'mt_per_block must be 1..3')
    d.mt_per_block, d.cout = 2, 22
    expect(h.refvsr_conv_mfma(ctypes.byref(d), None), 'nhwc16 output needs cout % 4 == 0')
    d.cout, d.out_mode, d.res = 24, 1, p                                      # pixel shuffle with a residual
    expect(h.refvsr_conv_mfma(ctypes.byref(d), None), 'pixel-shuffle output constraints')

    assert h.refvsr_resblock_lean_fits(24) == 1 and h.refvsr_resblock_lean_fits(16) == 1
    assert h.refvsr_resblock_lean_fits(48) == 0 and h.refvsr_resblock_lean_fits(20) == 0
    expect(h.refvsr_resblock_lean(p, 24, 8, 8, p, p, p, p, 7, 0.0, 1.0, p, None), 'in-place')
    q = ctypes.cast(ctypes.addressof(buf) + 2048, P)
    expect(h.refvsr_resblock_lean(p, 48, 8, 8, p, p, p, p, 14, 0.0, 1.0, q, None), '48')

    expect(h.refvsr_pack_nhwc16(p, 3, 8, 8, q, 12, None), 'pack_nhwc16: bad args')
    expect(h.refvsr_resize(p, 3, 8, 8, q, 16, 16, 7, 0.0, 1.0, None, None, None, 0, 0, 0, None), 'resize: bad mode 7')
    expect(h.refvsr_warp_nhwc16(p, 1, 8, 24, q, 8, 8, q, None), 'warp_nhwc16: bad args')
    expect(h.refvsr_match_top2(p, 1, q, 512, 1, q, q, None), 'match_top2')
    expect(h.refvsr_aligned_sample(p, 1, 1, 1, 24, q, q, None), 'map too small for reflection padding')
    expect(h.refvsr_block_gather_nhwc16(p, 8, 8, 20, q, 4, 4, 2, q, None), 'block_gather')
    flags = (ctypes.c_void_p * 1)(p)
    expect(h.refvsr_buffers_equal(flags, flags, 1, 24, q, None), 'multiple of 16 bytes')


def test_multimap_entry_points_validate_before_device_work(libpath):
    """The ABI 11 entry points (host arrays of per-map device pointers): batch bounds, null table entries, shape support and aliasing are
    rejected with a message before anything touches the device."""
    from refvsr_amd import hip
    h = hip.lib()
    P = ctypes.c_void_p
    buf = (ctypes.c_char * 65536)()
    at = lambda off: ctypes.cast(ctypes.addressof(buf) + off, P)
    arr = lambda *ps: (P * len(ps))(*[p.value for p in ps])

    def expect(rc, text):
        assert rc != 0 and text in h.refvsr_last_error().decode(), h.refvsr_last_error()

    assert hip.MAX_MAPS == 4
    s2, o2 = arr(at(0), at(4096)), arr(at(8192), at(12288))
    five = arr(at(0), at(64), at(128), at(192), at(256))
    expect(h.refvsr_resblock24_chain_batch(five, 5, 8, 8, 1, at(1024), 43264, 0.0, None, None, five, None), 'resblock24_chain: bad args')
    expect(h.refvsr_resblock24_chain_batch(s2, 2, 8, 8, 2, at(1024), 43264, 0.0, None, None, o2, None), 'n >= 2 needs scratch0')
    expect(h.refvsr_resblock24_chain_batch(s2, 2, 8, 8, 1, at(1024), 43264, 0.0, None, None, s2, None), 'buffers must be distinct')
    expect(h.refvsr_resblock24_chain_batch(arr(at(0), P(0)), 2, 8, 8, 1, at(1024), 43264, 0.0, None, None, o2, None), 'null map pointer (map 1)')
    expect(h.refvsr_resblock48_chain_batch(five, 5, 8, 8, 1, at(1024), 172544, 0.0, None, None, five, None), 'resblock48_chain: bad args')
    expect(h.refvsr_resblock48_chain_batch(s2, 2, 8, 8, 3, at(1024), 172544, 0.0, at(16384), None, o2, None), 'n >= 3 needs scratch1')
    expect(h.refvsr_resblock48_chain_batch(s2, 2, 8, 8, 1, at(1024), 172544, 0.0, None, None, arr(at(8192), at(8192)), None), 'buffers must be distinct')
    expect(h.refvsr_resblock48_chain_batch(s2, 2, 8, 8, 1, at(1024), 1024, 0.0, None, None, o2, None), 'stride >= 172544')
    expect(h.refvsr_conv24_batch(s2, 24, None, 0, 5, 8, 8, at(1024), 1.0, None, None, 1.0, o2, None), '1..4 maps per launch')
    expect(h.refvsr_conv24_batch(s2, 20, None, 0, 2, 8, 8, at(1024), 1.0, None, None, 1.0, o2, None), 'input channels not supported')
    expect(h.refvsr_conv24_batch(s2, 24, None, 24, 2, 8, 8, at(1024), 1.0, None, None, 1.0, o2, None), 'src1 / c1 mismatch')
    expect(h.refvsr_conv24_batch(s2, 24, None, 0, 2, 8, 8, at(1024), 1.0, None, None, 1.0, s2, None), 'in-place operation is not supported')
    expect(h.refvsr_conv_shuffle2_batch(s2, 2, 48, 8, 8, at(1024), 1.0, o2, None), '48 channels not supported (24)')
    expect(h.refvsr_conf_alpha_batch(s2, s2, 2, 8, 8, 3, at(1024), at(2048), 0.2, at(3072), 24, 0.2, o2, None, None), 'up must be 1 or 2')
    expect(h.refvsr_conf_alpha_batch(s2, s2, 2, 8, 8, 2, at(1024), at(2048), 0.2, at(3072), 24, 0.2, o2, o2, None), 'max by-product exists at up = 1 only')
    expect(h.refvsr_conf_alpha_batch(s2, s2, 2, 8, 8, 1, at(1024), at(2048), 0.2, at(3072), 48, 0.2, o2, None, None), '48 output channels not supported (24)')
    expect(h.refvsr_warp_nhwc16_batch(s2, 2, 8, 8, 20, s2, 8, 8, o2, None), 'warp_nhwc16: bad args')
    expect(h.refvsr_warp_nhwc16_up2_batch(five, 5, 8, 8, 24, five, 8, 8, five, None), '(map, flow) pairs per launch')
    expect(h.refvsr_warp_planar_batch(arr(at(0), P(0)), 2, 1, 8, 8, s2, 8, 8, o2, None), 'null pointer (pair 1)')



def test_cu_budget_argument_checks_and_constants(libpath):
    """ABI 13 / 14 host-side contracts that need no GPU: REFVSR_MAX_MAPS of the library equals the binding's copy (checked at load
    time too), the CU-budget setter validates before any device work, the result-format enum of the header equals the binding's."""
    from refvsr_amd import hip
    h = hip.lib()
    assert h.refvsr_max_maps() == hip.MAX_MAPS == 4
    src = open(os.path.join(ROOT, 'include', 'refvsr_hip.h')).read()
    assert re.search(r'REFVSR_RESULT_F32 = 0, REFVSR_RESULT_F16 = 1, REFVSR_RESULT_U8 = 2', src)
    assert (hip.RESULT_F32, hip.RESULT_F16, hip.RESULT_U8) == (0, 1, 2)
    assert h.refvsr_stream_set_cu_budget(ctypes.c_void_p(0x1000), 12) != 0 and b'multiple of 8' in h.refvsr_last_error()
    assert h.refvsr_stream_set_cu_budget(ctypes.c_void_p(0x1000), -8) != 0
    assert h.refvsr_stream_set_cu_budget(ctypes.c_void_p(0x1000), 0) == 0        # forgetting an unknown stream is not an error
    assert h.refvsr_convert_result(None, 16, hip.RESULT_U8, None, None) != 0 and b'bad args' in h.refvsr_last_error()
    assert h.refvsr_stream_destroy(None) != 0
