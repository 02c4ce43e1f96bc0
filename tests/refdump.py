"""Reader for the binary dumps written by oracle/ref_dump.cpp (test infrastructure).

The dump holds what the *reference* decoder handed to its reconstruct path
(lh264_mb_t / lh264_slice_t records + coefficients) and what came out
(pre-deblock and final planes); tests/golden/make_golden.py turns dumps into
the small committed fixtures.
"""
import numpy as np

MB_DTYPE = np.dtype([
    ("mb_type", "<u2"), ("cbp", "u1"), ("qp_y", "u1"), ("qp_c", "u1", (2,)), ("flags", "u1"),
    ("intra_avail", "u1"), ("intra_mode", "i1", (16,)), ("chroma_mode", "i1"), ("reserved0", "u1"),
    ("slice_id", "<u2"), ("sub_type", "u1", (4,)), ("ref_idx", "i1", (4,)), ("nzc", "u1", (24,)),
    ("mv", "<i2", (16, 2)), ("reserved1", "u1", (4,)),
])
assert MB_DTYPE.itemsize == 128

SLICE_DTYPE = np.dtype([
    ("first_mb", "<i4"), ("n_mbs", "<i4"), ("slice_type", "u1"), ("deblock_idc", "u1"),
    ("alpha_c0_offset", "i1"), ("beta_offset", "i1"), ("weighted_pred", "u1"), ("luma_log2_denom", "u1"),
    ("chroma_log2_denom", "u1"), ("n_refs", "u1"),
    ("luma_weight", "<i2", (16,)), ("luma_offset", "<i2", (16,)),
    ("chroma_weight", "<i2", (16, 2)), ("chroma_offset", "<i2", (16, 2)),
    ("ref_slot", "i1", (16,)), ("luma_dc_weight", "u1"), ("reserved", "u1", (7,)),
])
assert SLICE_DTYPE.itemsize == 232, SLICE_DTYPE.itemsize

# one context-model symbol as observed in the reference: kind 0 luma DC, 1 chroma DC, 2 nonzeros(4x4), 3 coefficient(4x4),
# 4 nonzeros(8x8), 5 coefficient(8x8); value = the coded integer; prior = flat index into the reference's prior table
SYM_DTYPE = np.dtype([("kind", "u1"), ("value", "<i2"), ("prior", "<u4")])
assert SYM_DTYPE.itemsize == 7


# the DecodedMacroblock fields the recompressor's per-macroblock emit code reads (decoded_macroblock.h:12-34), as packed
# by oracle/ref_dump.cpp:pack_rtd; have == 0 for skipped / uncovered macroblocks
RTD_DTYPE = np.dtype([
    ("have", "u1"), ("slice_type", "u1"), ("t8", "u1"), ("cbp_c", "u1"), ("cbp_l", "u1"), ("chroma_mode", "u1"),
    ("luma16_mode", "u1"), ("luma_qp", "u1"), ("mb_type", "<u4"), ("num_ref_idx_l0", "<u4"), ("skip_run", "<i4"),
    ("ref_idx", "i1", (4,)), ("sub_type", "u1", (4,)), ("pred_mode", "i1", (16,)), ("mvd", "<i2", (16, 2)),
    ("delta_qp", "<i4"), ("last_mb_qp", "<i4"),
])
assert RTD_DTYPE.itemsize == 116


class Frame:
    __slots__ = ("id", "mb_w", "mb_h", "crop_w", "crop_h", "has_final", "ref_ids", "slices", "mbs",
                 "coeffs", "covered", "pre", "fin", "levels", "nei", "syms", "frame_num", "rtd", "slice_extra")


def read_dump(path, max_frames=None):
    data = np.fromfile(path, dtype=np.uint8)
    assert bytes(data[:8]) == b"LH264DMP", "bad magic"
    ver, nfr = np.frombuffer(data[8:16].tobytes(), "<i4")
    assert ver == 5
    pos = 16
    frames = []
    for _ in range(nfr if max_frames is None else min(nfr, max_frames)):
        hdr = np.frombuffer(data[pos:pos + 32 + 64 + 4].tobytes(), "<i4")
        pos += 100
        f = Frame()
        f.id, f.mb_w, f.mb_h, nsl, f.crop_w, f.crop_h, f.has_final, nref = [int(x) for x in hdr[:8]]
        f.ref_ids = [int(x) for x in hdr[8:8 + nref]]
        f.frame_num = int(hdr[24])
        n = f.mb_w * f.mb_h
        f.slices = np.frombuffer(data[pos:pos + nsl * 232].tobytes(), SLICE_DTYPE).copy(); pos += nsl * 232
        # per slice: pad bit count, pad bits, PPS transform_8x8_mode_flag, entropy_coding_mode_flag
        f.slice_extra = np.frombuffer(data[pos:pos + nsl * 16].tobytes(), "<i4").reshape(nsl, 4).copy(); pos += nsl * 16
        f.mbs = np.frombuffer(data[pos:pos + n * 128].tobytes(), MB_DTYPE).copy(); pos += n * 128
        f.coeffs = np.frombuffer(data[pos:pos + n * 768].tobytes(), "<i2").reshape(n, 384).copy(); pos += n * 768
        f.covered = data[pos:pos + n].copy(); pos += n
        f.levels = np.frombuffer(data[pos:pos + n * 768].tobytes(), "<i2").reshape(n, 384).copy(); pos += n * 768
        f.nei = data[pos:pos + n * 75].reshape(n, 3, 25).copy(); pos += n * 75
        f.syms = []
        for _k in range(n):
            ln = int(np.frombuffer(data[pos:pos + 4].tobytes(), "<i4")[0]); pos += 4
            f.syms.append(np.frombuffer(data[pos:pos + ln].tobytes(), SYM_DTYPE).copy()); pos += ln
        f.rtd = np.frombuffer(data[pos:pos + n * 116].tobytes(), RTD_DTYPE).copy(); pos += n * 116
        f.pre, f.fin = [], []
        for p in range(3):
            bs = 8 if p else 16
            sz = n * bs * bs
            f.pre.append(data[pos:pos + sz].reshape(f.mb_h * bs, f.mb_w * bs).copy()); pos += sz
        if f.has_final:
            for p in range(3):
                bs = 8 if p else 16
                sz = n * bs * bs
                f.fin.append(data[pos:pos + sz].reshape(f.mb_h * bs, f.mb_w * bs).copy()); pos += sz
        frames.append(f)
    read_dump.tags = None
    if max_frames is None and bytes(data[pos:pos + 4]) == b"TAGS":
        pos += 4
        nt = int(np.frombuffer(data[pos:pos + 4].tobytes(), "<i4")[0]); pos += 4
        tags = {}
        for _ in range(nt):
            tag, ln = [int(x) for x in np.frombuffer(data[pos:pos + 8].tobytes(), "<i4")]; pos += 8
            tags[tag] = bytes(data[pos:pos + ln]); pos += ln
        read_dump.tags = tags        # the recompressor's output streams (tag 0x7fffffff = the default stream)
    return frames
