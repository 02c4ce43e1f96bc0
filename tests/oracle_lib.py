"""ctypes binding of oracle/liboracle.so (the CPU restatement; test infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
PAD_Y, PAD_C = 32, 16


class OrcPic(C.Structure):
    _fields_ = [("y", C.c_void_p), ("u", C.c_void_p), ("v", C.c_void_p), ("stride_y", C.c_int), ("stride_c", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(ORACLE_DIR, "liboracle.so")
        srcs = [os.path.join(ORACLE_DIR, f) for f in ("oracle_recon.c", "oracle_recon.h", "oracle_model.c", "oracle_model.h", "oracle_coder.c", "oracle_coder.h")]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "oracle"], stdout=subprocess.DEVNULL)
        _lib = C.CDLL(so)
        _lib.orc_recon_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(OrcPic), C.POINTER(OrcPic),
                                         C.c_int, C.c_int, C.c_int]
        _lib.orc_recon_frame.restype = None
        _lib.orc_luma_dc_qmul.restype = C.c_int
    return _lib


def pic_geometry(mb_w, mb_h):
    """stride/offsets of the reference's padded picture layout (pic_queue.cpp:62-112)."""
    w, h = mb_w * 16, mb_h * 16
    sy = (w + 2 * PAD_Y + 31) & ~31
    sc = sy >> 1
    hy, hc = h + 2 * PAD_Y, (h >> 1) + 2 * PAD_C
    off_y = PAD_Y * sy + PAD_Y
    off_u = sy * hy + PAD_C * sc + PAD_C
    off_v = sy * hy + sc * hc + PAD_C * sc + PAD_C
    total = sy * hy + 2 * sc * hc
    return sy, sc, off_y, off_u, off_v, total


class HostPic:
    """a padded picture in host memory"""

    def __init__(self, mb_w, mb_h, fill=128):
        self.mb_w, self.mb_h = mb_w, mb_h
        self.sy, self.sc, self.off_y, self.off_u, self.off_v, total = pic_geometry(mb_w, mb_h)
        self.buf = np.full(total, fill, dtype=np.uint8)

    def struct(self):
        base = self.buf.ctypes.data
        return OrcPic(base + self.off_y, base + self.off_u, base + self.off_v, self.sy, self.sc)

    def plane(self, p):
        bs = 8 if p else 16
        st = self.sc if p else self.sy
        off = (self.off_y, self.off_u, self.off_v)[p]
        h, w = self.mb_h * bs, self.mb_w * bs
        return np.lib.stride_tricks.as_strided(self.buf[off:], shape=(h, w), strides=(st, 1))

    def padded_plane(self, p):
        bs, pad = (8, PAD_C) if p else (16, PAD_Y)
        st = self.sc if p else self.sy
        off = (self.off_y, self.off_u, self.off_v)[p] - pad * st - pad
        h, w = self.mb_h * bs + 2 * pad, self.mb_w * bs + 2 * pad
        return np.lib.stride_tricks.as_strided(self.buf[off:], shape=(h, w), strides=(st, 1))


def recon_frame(mbs, coeffs, slices, dst, refs, flags=0):
    """mbs/coeffs/slices: numpy arrays (refdump dtypes); dst: HostPic; refs: list of HostPic"""
    L = lib()
    mbs = np.ascontiguousarray(mbs)
    coeffs = np.ascontiguousarray(coeffs, dtype=np.int16)
    slices = np.ascontiguousarray(slices)
    d = dst.struct()
    arr = (OrcPic * 16)()
    for i, r in enumerate(refs[:16]):
        arr[i] = r.struct()
    L.orc_recon_frame(mbs.ctypes.data, coeffs.ctypes.data, slices.ctypes.data, len(slices), C.byref(d), arr,
                      dst.mb_w, dst.mb_h, flags)


NO_EXPAND, NO_DEBLOCK = 1, 2


# ---- context-model oracle (oracle/oracle_model.c) ---------------------------------------------------------------
ORC_SYM_DTYPE = np.dtype([("prior", "<u4"), ("value", "<i2"), ("kind", "u1"), ("pad", "u1")])


def model_nnz_images(frames, past_of):
    """the FreqImage as 24 counts per macroblock: a skipped MB inherits the PAST entry (decode_slice.cpp:3104-3108)"""
    L = lib()
    imgs = []
    for i, f in enumerate(frames):
        n = f.mb_w * f.mb_h
        img = np.zeros((n, 24), dtype=np.uint8)
        past = imgs[past_of[i]] if past_of[i] is not None else None
        lv = np.ascontiguousarray(f.levels, dtype=np.int16)
        for k in range(n):
            t = int(f.mbs["mb_type"][k])
            if t == 0x100 or t == 0:
                if past is not None:
                    img[k] = past[k]
            else:
                L.orc_model_nnz24(lv[k].ctypes.data_as(C.c_void_p), img[k].ctypes.data_as(C.c_void_p))
        imgs.append(img)
    return imgs


def model_frame_symbols(f, img, past):
    """-> list (per MB) of ORC_SYM_DTYPE arrays, exactly what the reference's model codes for the frame"""
    L = lib()
    out = np.zeros(432, dtype=ORC_SYM_DTYPE)
    res = []
    zero = np.zeros(24, dtype=np.uint8)
    lv = np.ascontiguousarray(f.levels, dtype=np.int16)
    for k in range(f.mb_w * f.mb_h):
        t = int(f.mbs["mb_type"][k])
        if t in (0x100, 0x200, 0):
            res.append(out[:0].copy())
            continue
        left = img[k - 1] if k % f.mb_w else zero
        above = img[k - f.mb_w] if k >= f.mb_w else zero
        pst = past[k] if past is not None else zero
        st = int(f.slices["slice_type"][f.mbs["slice_id"][k]])
        n = L.orc_model_mb_symbols(lv[k].ctypes.data_as(C.c_void_p), t, st, int(f.mbs["cbp"][k]), int(f.mbs["flags"][k]) & 1,
                                   np.ascontiguousarray(left).ctypes.data_as(C.c_void_p), np.ascontiguousarray(above).ctypes.data_as(C.c_void_p),
                                   np.ascontiguousarray(pst).ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        res.append(out[:n].copy())
    return res
