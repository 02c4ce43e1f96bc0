"""Host front end binding: Annex-B bytes -> list of frames (numpy views of the parser's records)."""
import ctypes as C

import numpy as np

from . import _lib as L


class ParsedFrame:
    pass


def parse_file(data, strict=False, pcm=False):
    """A whole Annex-B file fed chunk by chunk as the reference's console application does.
    -> (frames, error_text, main_stream): main_stream is the recompressor's default stream (the '.pip' file itself).
    pcm=True: a fourth element, the samples of the stream's I_PCM macroblocks (stream LH264_TAG_PCM of the container)."""
    return parse_stream(data, strict, _file=True, _pcm=pcm)


def parse_batch_time(datas, threads=0, keep=True):
    """parse a batch of Annex-B files on host threads and throw the records away -> (seconds, pictures parsed).
    keep=True: C ABI lh264_parse_batch (all pictures of all streams stay in memory until the end); keep=False:
    lh264_parse_batch_discard (a picture is released when complete: the steady state of a pipeline).  For bench.py."""
    import time
    lib = L.lib()
    n = len(datas)
    ptrs = (C.c_char_p * n)(*[bytes(d) for d in datas])
    lens = (C.c_size_t * n)(*[len(d) for d in datas])
    if not keep:
        pics = (C.c_int64 * n)()
        t0 = time.perf_counter()
        L.check(lib.lh264_parse_batch_discard(ptrs, lens, n, threads, pics))
        return time.perf_counter() - t0, int(sum(pics))
    outs = (C.c_void_p * n)()
    t0 = time.perf_counter()
    L.check(lib.lh264_parse_batch(ptrs, lens, n, threads, outs))
    dt = time.perf_counter() - t0
    pics = 0
    for i in range(n):
        pics += lib.lh264_parser_frame_count(outs[i])
        lib.lh264_parser_destroy(outs[i])
    return dt, pics


def parse_stream(data, strict=False, _file=False, _pcm=False):
    """-> (frames, error_text).  frames have the attributes ReconSession / CtxSession expect."""
    lib = L.lib()
    p = lib.lh264_parser_create()
    try:
        if _file:
            rc = lib.lh264_parser_feed_file(p, bytes(data), len(data))
        else:
            rc = lib.lh264_parser_feed(p, bytes(data), len(data), 1)
        err = lib.lh264_parser_error(p).decode()
        if rc != 0 and strict:
            raise RuntimeError("h264 parse error: " + err)
        frames = []
        info = np.zeros(1, dtype=L.FRAME_INFO_DTYPE)
        for i in range(lib.lh264_parser_frame_count(p)):
            L.check(lib.lh264_parser_frame_info(p, i, info.ctypes.data_as(C.c_void_p)))
            fi = info[0]
            f = ParsedFrame()
            f.id, f.mb_w, f.mb_h, f.frame_num = int(fi["id"]), int(fi["mb_w"]), int(fi["mb_h"]), int(fi["frame_num"])
            f.crop_x, f.crop_y, f.crop_w, f.crop_h = int(fi["crop_x"]), int(fi["crop_y"]), int(fi["crop_w"]), int(fi["crop_h"])
            f.is_ref, f.idr = bool(fi["is_ref"]), bool(fi["idr"])
            f.ref_ids = [int(x) for x in fi["ref_ids"][:int(fi["n_refs"])]]
            n, ns = f.mb_w * f.mb_h, int(fi["n_slices"])

            def arr(ptr, nbytes, dtype):
                return np.frombuffer(C.string_at(ptr, nbytes), dtype=dtype).copy()
            f.mbs = arr(lib.lh264_parser_frame_mbs(p, i), n * 128, L.MB_DTYPE)
            f.coeffs = arr(lib.lh264_parser_frame_coeffs(p, i), n * 768, "<i2").reshape(n, 384)
            f.levels = arr(lib.lh264_parser_frame_levels(p, i), n * 768, "<i2").reshape(n, 384)
            f.slices = arr(lib.lh264_parser_frame_slices(p, i), ns * 232, L.SLICE_DTYPE)
            f.covered = arr(lib.lh264_parser_frame_covered(p, i), n, np.uint8)
            f.syn = arr(lib.lh264_parser_frame_syntax(p, i), n * 116, L.MBSYN_DTYPE)
            f.slice_syn = arr(lib.lh264_parser_frame_slice_syntax(p, i), ns * 16, "<i4").reshape(ns, 4)
            cnt = C.c_int(0)
            ptr = lib.lh264_parser_frame_syn_symbols(p, i, C.byref(cnt))
            f.syn_syms = arr(ptr, cnt.value * 8, L.CTX_SYM_DTYPE) if cnt.value else np.zeros(0, L.CTX_SYM_DTYPE)
            f.syn_off = arr(lib.lh264_parser_frame_syn_offsets(p, i), (n + 1) * 4, "<u4")
            frames.append(f)
        if _file:
            ln = C.c_size_t(0)
            ptr = lib.lh264_parser_main_stream(p, C.byref(ln))
            main = C.string_at(ptr, ln.value) if ln.value else b""
            if _pcm:
                ptr = lib.lh264_parser_pcm_samples(p, C.byref(ln))
                return frames, err, main, (C.string_at(ptr, ln.value) if ln.value else b"")
            return frames, err, main
        return frames, err
    finally:
        lib.lh264_parser_destroy(p)
