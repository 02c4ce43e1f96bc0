"""ctypes binding of liblh264.so (the C ABI of include/lh264.h)."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("LH264_SO", os.path.join(HERE, "liblh264.so"))   # LH264_SO: tuning experiments only


class LibraryMissing(RuntimeError):
    pass


# record layouts == include/lh264.h
MB_DTYPE = np.dtype([
    ("mb_type", "<u2"), ("cbp", "u1"), ("qp_y", "u1"), ("qp_c", "u1", (2,)), ("flags", "u1"),
    ("intra_avail", "u1"), ("intra_mode", "i1", (16,)), ("chroma_mode", "i1"), ("reserved0", "u1"),
    ("slice_id", "<u2"), ("sub_type", "u1", (4,)), ("ref_idx", "i1", (4,)), ("nzc", "u1", (24,)),
    ("mv", "<i2", (16, 2)), ("reserved1", "u1", (4,)),
])
SLICE_DTYPE = np.dtype([
    ("first_mb", "<i4"), ("n_mbs", "<i4"), ("slice_type", "u1"), ("deblock_idc", "u1"),
    ("alpha_c0_offset", "i1"), ("beta_offset", "i1"), ("weighted_pred", "u1"), ("luma_log2_denom", "u1"),
    ("chroma_log2_denom", "u1"), ("n_refs", "u1"),
    ("luma_weight", "<i2", (16,)), ("luma_offset", "<i2", (16,)),
    ("chroma_weight", "<i2", (16, 2)), ("chroma_offset", "<i2", (16, 2)),
    ("ref_slot", "i1", (16,)), ("luma_dc_weight", "u1"), ("reserved", "u1", (7,)),
])
JOB_DTYPE = np.dtype([
    ("mbs", "<u8"), ("coeffs", "<u8"), ("slices", "<u8"),
    ("dst", "<u8", (3,)), ("ref", "<u8", (16, 3)),
    ("mb_w", "<i4"), ("mb_h", "<i4"), ("stride_y", "<i4"), ("stride_c", "<i4"), ("n_slices", "<i4"), ("flags", "<i4"),
])
assert MB_DTYPE.itemsize == 128 and SLICE_DTYPE.itemsize == 232 and JOB_DTYPE.itemsize == 456

JOB_NO_EXPAND, JOB_NO_DEBLOCK = 1, 2
PAD_Y, PAD_C = 32, 16

_lib = None

_SIGS = {
    "lh264_abi_version": (C.c_int, []),
    "lh264_build_id": (C.c_char_p, []),
    "lh264_last_error": (C.c_char_p, []),
    "lh264_device_count": (C.c_int, []),
    "lh264_set_device": (C.c_int, [C.c_int]),
    "lh264_dev_malloc": (C.c_void_p, [C.c_size_t]),
    "lh264_dev_free": (C.c_int, [C.c_void_p]),
    "lh264_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "lh264_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "lh264_dev_memset": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]),
    "lh264_stream_sync": (C.c_int, [C.c_void_p]),
    "lh264_pic_bytes": (C.c_size_t, [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_size_t),
                                    C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "lh264_recon_frames": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "lh264_recon_chains": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "lh264_time_recon_chains": (C.c_double, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
}
FRAME_INFO_DTYPE = np.dtype([("id", "<i4"), ("mb_w", "<i4"), ("mb_h", "<i4"), ("n_slices", "<i4"), ("n_refs", "<i4"), ("frame_num", "<i4"),
                             ("crop_x", "<i4"), ("crop_y", "<i4"), ("crop_w", "<i4"), ("crop_h", "<i4"), ("is_ref", "<i4"), ("idr", "<i4"),
                             ("ref_ids", "<i4", (16,))])
_SIGS.update({
    "lh264_parser_create": (C.c_void_p, []),
    "lh264_parser_destroy": (None, [C.c_void_p]),
    "lh264_parser_feed": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int]),
    "lh264_pip_restore": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "lh264_restore_error": (C.c_char_p, []),
    "lh264_compress_batch": (C.c_int, [C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "lh264_compress_batch_devices": (C.c_int, [C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]),
    "lh264_compressed_status": (C.c_int, [C.c_void_p]),
    "lh264_compressed_error": (C.c_char_p, [C.c_void_p]),
    "lh264_compressed_main": (C.c_void_p, [C.c_void_p, C.POINTER(C.c_size_t)]),
    "lh264_compressed_tag": (C.c_void_p, [C.c_void_p, C.c_int, C.POINTER(C.c_size_t)]),
    "lh264_compressed_pictures": (C.c_int, [C.c_void_p]),
    "lh264_compressed_free": (None, [C.c_void_p]),
    "lh264_compress_release": (None, []),
    "lh264_pip_pack_bound": (C.c_size_t, [C.c_size_t, C.POINTER(C.c_size_t), C.c_int]),
    "lh264_pip_pack": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_int, C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "lh264_pip_restore_file": (C.c_int, [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "lh264_parse_batch": (C.c_int, [C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "lh264_parse_batch_discard": (C.c_int, [C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_int, C.c_int, C.POINTER(C.c_int64)]),
    "lh264_pip_restore_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "lh264_parser_feed_file": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t]),
    "lh264_parser_main_stream": (C.c_void_p, [C.c_void_p, C.POINTER(C.c_size_t)]),
    "lh264_parser_pcm_samples": (C.c_void_p, [C.c_void_p, C.POINTER(C.c_size_t)]),
    "lh264_parser_frame_count": (C.c_int, [C.c_void_p]),
    "lh264_parser_frame_info": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "lh264_parser_frame_mbs": (C.c_void_p, [C.c_void_p, C.c_int]),
    "lh264_parser_frame_coeffs": (C.c_void_p, [C.c_void_p, C.c_int]),
    "lh264_parser_frame_levels": (C.c_void_p, [C.c_void_p, C.c_int]),
    "lh264_parser_frame_slices": (C.c_void_p, [C.c_void_p, C.c_int]),
    "lh264_parser_frame_covered": (C.c_void_p, [C.c_void_p, C.c_int]),
    "lh264_parser_frame_syntax": (C.c_void_p, [C.c_void_p, C.c_int]),
    "lh264_parser_frame_slice_syntax": (C.c_void_p, [C.c_void_p, C.c_int]),
    "lh264_parser_frame_syn_symbols": (C.c_void_p, [C.c_void_p, C.c_int, C.c_void_p]),
    "lh264_parser_frame_syn_offsets": (C.c_void_p, [C.c_void_p, C.c_int]),
    "lh264_parser_error": (C.c_char_p, [C.c_void_p]),
})
CODE_JOB_DTYPE = np.dtype([("syn_syms", "<u8"), ("syn_off", "<u8"), ("ctx_syms", "<u8"), ("ctx_n_syms", "<u8"), ("n_mbs", "<i4"), ("reserved", "<i4"),
                           ("ctx_sym_off", "<u8"), ("ctx_sym_base", "<u8")])
CODE_STREAM_DTYPE = np.dtype([("hash_keys", "<u8"), ("hash_cells", "<u8"), ("out", "<u8"), ("out_len", "<u8"), ("hash_cap", "<u4"), ("out_cap", "<u4")])
N_TAG_SLOTS = 40
assert CODE_JOB_DTYPE.itemsize == 56 and CODE_STREAM_DTYPE.itemsize == 40
_SIGS["lh264_code_chains"] = (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_longlong, C.c_int, C.c_void_p])
_SIGS["lh264_code_binarise_chains"] = (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_longlong, C.c_int, C.c_void_p])
_SIGS["lh264_code_finish_chains"] = (C.c_int, [C.c_void_p, C.c_int, C.c_void_p])
_SIGS["lh264_code_last_totals"] = (C.c_int, [C.c_void_p, C.c_void_p])
EXPORTS = sorted(_SIGS)


def lib():
    """the loaded C-ABI library; raises LibraryMissing when it has not been built"""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise LibraryMissing("%s not built: run `python -c 'import __graft_entry__ as g; g.build()'`" % SO_PATH)
        try:
            import torch  # noqa: F401  -- load torch's bundled HIP runtime first so the process has exactly one
        except Exception:  # pragma: no cover
            pass
        L = C.CDLL(SO_PATH)
        for name, (res, args) in _SIGS.items():
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise RuntimeError("liblh264: error %d: %s" % (rc, lib().lh264_last_error().decode()))


def pic_geometry(mb_w, mb_h):
    sy, sc = C.c_int(), C.c_int()
    oy, ou, ov = C.c_size_t(), C.c_size_t(), C.c_size_t()
    total = lib().lh264_pic_bytes(mb_w, mb_h, C.byref(sy), C.byref(sc), C.byref(oy), C.byref(ou), C.byref(ov))
    return sy.value, sc.value, oy.value, ou.value, ov.value, total

# ---- context-index (row a8) ----------------------------------------------------------------------------------
# row a10 syntax record (lh264_mbsyn_t, byte-packed)
MBSYN_DTYPE = np.dtype([
    ("have", "u1"), ("slice_type", "u1"), ("t8", "u1"), ("cbp_c", "u1"), ("cbp_l", "u1"), ("chroma_mode", "u1"),
    ("luma16_mode", "u1"), ("luma_qp", "u1"), ("mb_type", "<u4"), ("num_ref_idx_l0", "<u4"), ("skip_run", "<i4"),
    ("ref_idx", "i1", (4,)), ("sub_type", "u1", (4,)), ("pred_mode", "i1", (16,)), ("mvd", "<i2", (16, 2)),
    ("delta_qp", "<i4"), ("last_mb_qp", "<i4"),
])
assert MBSYN_DTYPE.itemsize == 116
CTX_SYM_DTYPE = np.dtype([("prior", "<u4"), ("value", "<i2"), ("kind", "u1"), ("pad", "u1")])
CTX_JOB_DTYPE = np.dtype([("mbs", "<u8"), ("levels", "<u8"), ("slices", "<u8"), ("nnz_past", "<u8"), ("nnz_cur", "<u8"),
                          ("syms", "<u8"), ("n_syms", "<u8"), ("mb_w", "<i4"), ("mb_h", "<i4"),
                          ("sym_off", "<u8"), ("sym_base", "<u8"), ("syms_cap", "<u8")])
CTX_MAX_SYMS = 432
assert CTX_SYM_DTYPE.itemsize == 8 and CTX_JOB_DTYPE.itemsize == 88
_SIGS["lh264_ctx_index_chains"] = (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p])
_SIGS["lh264_ctx_count_chains"] = (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p])
CODE_JOB_DTYPE = np.dtype([("syn_syms", "<u8"), ("syn_off", "<u8"), ("ctx_syms", "<u8"), ("ctx_n_syms", "<u8"), ("n_mbs", "<i4"), ("reserved", "<i4"),
                           ("ctx_sym_off", "<u8"), ("ctx_sym_base", "<u8")])
CODE_STREAM_DTYPE = np.dtype([("hash_keys", "<u8"), ("hash_cells", "<u8"), ("out", "<u8"), ("out_len", "<u8"), ("hash_cap", "<u4"), ("out_cap", "<u4")])
N_TAG_SLOTS = 40
assert CODE_JOB_DTYPE.itemsize == 56 and CODE_STREAM_DTYPE.itemsize == 40
_SIGS["lh264_code_chains"] = (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_longlong, C.c_int, C.c_void_p])
EXPORTS = sorted(_SIGS)
