"""ctypes binding of librefvsr_hip.so (the C-ABI declared in include/refvsr_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a call fails, a
RuntimeError is raised.  Build it with `make -C refvsr_amd/csrc` (or `__graft_entry__.build()`).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'librefvsr_hip.so')

OUT_NHWC16, OUT_NHWC16_SHUFFLE2, OUT_PLANAR32 = 0, 1, 2
RS_BICUBIC, RS_BILINEAR, RS_BILINEAR_AC, RS_NEAREST = 0, 1, 2, 3
RESULT_F32, RESULT_F16, RESULT_U8 = 0, 1, 2
MATCH_KP, MATCH_ROWCHUNK, MATCH_COLBLOCK = 152, 256, 512
ABI_VERSION = 14
MAX_MAPS = 4
RESBLOCK24_BLOB_BYTES = 43264
RESBLOCK48_BLOB_BYTES = 172544


class RefvsrConv(C.Structure):
    """Mirror of `struct RefvsrConv` (include/refvsr_hip.h)."""
    _fields_ = [
        ('src0', C.c_void_p), ('c0', C.c_int),
        ('src1', C.c_void_p), ('c1', C.c_int),
        ('h_in', C.c_int), ('w_in', C.c_int),
        ('h_out', C.c_int), ('w_out', C.c_int),
        ('ksize', C.c_int), ('stride', C.c_int), ('pad', C.c_int),
        ('wpack', C.c_void_p), ('bias', C.c_void_p),
        ('cout', C.c_int), ('mt_per_block', C.c_int), ('ksteps', C.c_int),
        ('act_slope', C.c_float),
        ('mul', C.c_void_p), ('mul_c', C.c_int),
        ('res', C.c_void_p), ('res_c', C.c_int),
        ('post_slope', C.c_float),
        ('out_mode', C.c_int),
        ('out', C.c_void_p), ('out_c', C.c_int),
        ('res_planar', C.c_void_p),
        ('f32', C.c_int),
        ('add_const', C.c_float), ('clamp_lo', C.c_float), ('clamp_hi', C.c_float),
        ('batch', C.c_int), ('bs_src0', C.c_size_t), ('bs_src1', C.c_size_t), ('bs_out', C.c_size_t), ('bs_res_planar', C.c_size_t),
    ]


_P, _I, _F, _Z = C.c_void_p, C.c_int, C.c_float, C.c_size_t

# name -> argtypes; every function returns int (0 = ok) except the two listed in _SPECIAL
SIGNATURES = {
    'refvsr_init': [],
    'refvsr_max_maps': [],                       # returns REFVSR_MAX_MAPS
    'refvsr_num_cus': [],                        # returns the CU count
    'refvsr_stream_create_cu_range': [_I, _I, C.POINTER(C.c_void_p)],
    'refvsr_stream_set_cu_budget': [_P, _I],
    'refvsr_stream_destroy': [_P],
    'refvsr_conv_mfma': [C.POINTER(RefvsrConv), _P],
    'refvsr_set_conv_workgroup_cap': [_I],
    'refvsr_kslot': [_I, _I, _I, _I, _I],        # returns the slot, not a status
    'refvsr_ksteps': [_I, _I],                   # returns the K-step count
    'refvsr_set_probe': [_P, _I],
    'refvsr_resblock_lean_fits': [_I],
    'refvsr_set_resblock_waves': [_I],
    'refvsr_resblock_lean': [_P, _I, _I, _I, _P, _P, _P, _P, _I, _F, _F, _P, _P],
    'refvsr_resblock_chain': [_P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _F, _F, _P, _P, _P, _P],
    'refvsr_resblock24_chain': [_P, _I, _I, _I, _P, _Z, _F, _P, _P, _P, _P],
    'refvsr_resblock48_chain': [_P, _I, _I, _I, _P, _Z, _F, _P, _P, _P, _P],
    'refvsr_resblock24_kblock': [_I, _I],        # returns the packed K-block, not a status
    'refvsr_set_resblock24_waves': [_I],
    'refvsr_set_resblock24_store': [_I],
    'refvsr_conv24_supported': [_I, _I],         # returns 0 / 1
    'refvsr_conv24_blob_bytes': [_I, _I],        # returns the size
    'refvsr_conv24_kblock': [_I, _I, _I],        # returns the packed K-block
    'refvsr_conv24': [_P, _I, _P, _I, _I, _I, _P, _F, _P, _P, _F, _P, _P],
    'refvsr_conv32_supported': [_I, _I],         # returns 0 / 1
    'refvsr_conv32_blob_bytes': [_I, _I],        # returns the size
    'refvsr_conv32': [_P, _I, _P, _I, _I, _I, _P, _F, _P, _P, _F, _P, _P],
    'refvsr_conv48_supported': [_I, _I],         # returns 0 / 1
    'refvsr_conv48_blob_bytes': [_I, _I],        # returns the size
    'refvsr_conv48': [_P, _I, _P, _I, _I, _I, _P, _F, _P, _P, _F, _P, _P],
    'refvsr_conv_shuffle2_supported': [_I],      # returns 0 / 1
    'refvsr_conv_shuffle2_blob_bytes': [_I],     # returns the size
    'refvsr_conv_shuffle2': [_P, _I, _I, _I, _P, _F, _P, _P],
    'refvsr_conv_direct_f32': [_P, _I, _I, _I, _P, _P, _I, _I, _I, _I, _F, _P, _I, _I, _P],
    'refvsr_conv1x1_f32': [_P, _I, _I, _I, _P, _P, _F, _P, _P],
    'refvsr_pack_nhwc16': [_P, _I, _I, _I, _P, _I, _P],
    'refvsr_pack_nhwc32': [_P, _I, _I, _I, _P, _I, _P],
    'refvsr_unpack_nhwc16': [_P, _I, _I, _I, _I, _P, _P],
    'refvsr_resize': [_P, _I, _I, _I, _P, _I, _I, _I, _F, _F, _P, _P, _P, _I, _I, _I, _P],
    'refvsr_avgpool2': [_P, _I, _I, _I, _P, _P],
    'refvsr_maxpool2': [_P, _I, _I, _I, _P, _P],
    'refvsr_max2': [_P, _P, _P, _Z, _P],
    'refvsr_buffers_equal': [_P, _P, _I, _Z, _P, _P],
    'refvsr_warp_nhwc16': [_P, _I, _I, _I, _P, _I, _I, _P, _P],
    'refvsr_warp_nhwc16_up2': [_P, _I, _I, _I, _P, _I, _I, _P, _P],
    'refvsr_warp_planar': [_P, _I, _I, _I, _P, _I, _I, _P, _P],
    'refvsr_spynet_level_input': [_P, _P, _P, _I, _I, _P, _P, _P],
    'refvsr_conv_hr_last': [_P, _I, _I, _P, _F, _P, _I, _I, _P, _P],
    'refvsr_conv_last_supported': [_I],          # returns 0 / 1
    'refvsr_conv_last_blob_bytes': [_I],         # returns the size
    'refvsr_conv_last': [_P, _I, _I, _I, _P, _P, _I, _I, _P, _P],
    'refvsr_conv_last_fmt': [_P, _I, _I, _I, _P, _P, _I, _I, _P, _I, _P],
    'refvsr_conv_hr_last_fmt': [_P, _I, _I, _P, _F, _P, _I, _I, _P, _I, _P],
    'refvsr_convert_result': [_P, _Z, _I, _P, _P],
    'refvsr_conf_alpha': [_P, _P, _I, _I, _I, _P, _P, _F, _P, _I, _F, _P, _P, _P],
    'refvsr_spynet_level_input_batch': [_P, _P, _I, _P, _I, _I, _P, _P, _P],
    # multi-map launches (ABI 11): host arrays of `batch` device pointers
    'refvsr_resblock24_chain_batch': [_P, _I, _I, _I, _I, _P, _Z, _F, _P, _P, _P, _P],
    'refvsr_resblock48_chain_batch': [_P, _I, _I, _I, _I, _P, _Z, _F, _P, _P, _P, _P],      # ABI 12
    'refvsr_conv24_batch': [_P, _I, _P, _I, _I, _I, _I, _P, _F, _P, _P, _F, _P, _P],
    'refvsr_conv_shuffle2_batch': [_P, _I, _I, _I, _I, _P, _F, _P, _P],
    'refvsr_conf_alpha_batch': [_P, _P, _I, _I, _I, _I, _P, _P, _F, _P, _I, _F, _P, _P, _P],
    'refvsr_warp_nhwc16_batch': [_P, _I, _I, _I, _I, _P, _I, _I, _P, _P],
    'refvsr_warp_nhwc16_up2_batch': [_P, _I, _I, _I, _I, _P, _I, _I, _P, _P],
    'refvsr_warp_planar_batch': [_P, _I, _I, _I, _I, _P, _I, _I, _P, _P],
    'refvsr_match_patches': [_P, _I, _I, _P, _P, _P, _P],
    'refvsr_match_top2': [_P, _I, _P, _I, _I, _P, _P, _P],
    'refvsr_match_refine': [_P, _I, _I, _P, _I, _I, _P, _P, _P, _P, _I, _F, _P, _P, _P, _P],
    'refvsr_match_exact': [_P, _I, _I, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    'refvsr_match_naive': [_P, _I, _I, _P, _I, _I, _P, _P, _P],
    'refvsr_block_gather_nhwc16': [_P, _I, _I, _I, _P, _I, _I, _I, _P, _P],
    'refvsr_block_gather_rgb': [_P, _I, _I, _P, _I, _I, _I, _P, _P, _P],
    'refvsr_aligned_sample': [_P, _I, _I, _I, _I, _P, _P, _P],
    'refvsr_dcn_sample': [_P, _I, _I, _I, _P, _I, _P, _P],
    'refvsr_tsa_weight': [_P, _P, _P, _I, _I, _I, _P, _P],
    'refvsr_pool3s2_nhwc16': [_P, _I, _I, _I, _P, _I, _I, _I, _P],
    'refvsr_up2_bilinear_nhwc16': [_P, _I, _I, _I, _F, _P, _P],
    'refvsr_tsa_blend': [_P, _P, _P, _Z, _P, _P],
}
_SPECIAL = {'refvsr_abi_version': (C.c_int, []), 'refvsr_last_error': (C.c_char_p, [])}
EXPORTS = tuple(sorted(list(SIGNATURES) + list(_SPECIAL)))

_lib = None


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP library is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                'refvsr_amd: %s not found -- the HIP extension is required (no CPU fallback). '
                'Build it with `make -C refvsr_amd/csrc` or `python -c "import __graft_entry__ as g; g.build()"`.'
                % LIB_PATH)
        h = C.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(h, name)
            fn.argtypes = args
            fn.restype = C.c_int
        for name, (res, args) in _SPECIAL.items():
            fn = getattr(h, name)
            fn.argtypes = args
            fn.restype = res
        if h.refvsr_abi_version() != ABI_VERSION:
            raise RuntimeError('refvsr_amd: ABI mismatch (library %d, binding %d)' % (h.refvsr_abi_version(), ABI_VERSION))
        if h.refvsr_max_maps() != MAX_MAPS:
            raise RuntimeError('refvsr_amd: REFVSR_MAX_MAPS mismatch (library %d, binding %d)' % (h.refvsr_max_maps(), MAX_MAPS))
        _lib = h
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().refvsr_last_error()
        raise RuntimeError('refvsr_hip.%s failed (rc=%d): %s' % (what, rc, msg.decode() if msg else '?'))
