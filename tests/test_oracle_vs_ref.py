"""Pin the oracle (oracle/liboracle.so, our C restatement) against the REFERENCE's own C kernels.

oracle/_ref/libref_kernels.so is built by oracle/Makefile from the sources in /root/reference; it exists
only in the build container.  Same pattern as the reference's unit tests (random blocks vs. anchor):
test/decoder/DecUT_IdctResAddPred.cpp:7-108, DecUT_IntraPrediction.cpp, DecUT_DeblockCommon.cpp:258-415,
test/encoder/EncUT_MotionCompensation.cpp:12-260, test/common/ExpandPicture.cpp:111-199.
Bit-exact comparison (integer work).
"""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O

REFK = os.path.join(O.ORACLE_DIR, "_ref", "libref_kernels.so")
pytestmark = pytest.mark.skipif(not os.path.exists(REFK), reason="reference kernels not built (no /root/reference here)")


@pytest.fixture(scope="module")
def libs():
    ref = C.CDLL(REFK)          # NB: static init of the reference builds its 8.78 GiB model (~30 s)
    return O.lib(), ref


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_idct4x4(libs):
    orc, ref = libs
    rng = np.random.default_rng(1)
    for it in range(2000):
        amp = [16, 256, 2048, 32767][it % 4]
        coef = rng.integers(-amp, amp + 1, 16).astype(np.int16)
        if it % 7 == 0:
            coef[rng.integers(0, 16, 12)] = 0
        a = rng.integers(0, 256, (4, 32)).astype(np.uint8)
        b = a.copy()
        orc.orc_idct4x4_add(_p(a), 32, _p(coef.copy()))
        ref.refk_idct4x4_add(_p(b), 32, _p(coef.copy()))
        assert np.array_equal(a, b)


def test_idct8x8(libs):
    orc, ref = libs
    rng = np.random.default_rng(2)
    for it in range(1000):
        amp = [16, 256, 2048, 32767][it % 4]
        coef = rng.integers(-amp, amp + 1, 64).astype(np.int16)
        a = rng.integers(0, 256, (8, 32)).astype(np.uint8)
        b = a.copy()
        orc.orc_idct8x8_add(_p(a), 32, _p(coef.copy()))
        ref.refk_idct8x8_add(_p(b), 32, _p(coef.copy()))
        assert np.array_equal(a, b)


def test_dc_transforms(libs):
    orc, ref = libs
    rng = np.random.default_rng(3)
    for it in range(1000):
        qp = int(rng.integers(0, 52))
        blk = rng.integers(-2048, 2048, 384).astype(np.int16)
        a, b = blk.copy(), blk.copy()
        orc.orc_luma_dc_dequant_idct(_p(a), orc.orc_luma_dc_qmul(qp, 16))
        ref.refk_luma_dc_dequant_idct(_p(b), qp)
        assert np.array_equal(a, b)
        a, b = blk.copy(), blk.copy()
        orc.orc_chroma_dc_idct(_p(a[256:]))
        ref.refk_chroma_dc_idct(_p(b[256:]))
        assert np.array_equal(a, b)


@pytest.mark.parametrize("kind,nmodes,size", [("pred4x4", 14, 4), ("pred16x16", 7, 16), ("predc8x8", 7, 8)])
def test_intra_pred(libs, kind, nmodes, size):
    orc, ref = libs
    rng = np.random.default_rng(4)
    st = 64
    for it in range(300):
        for mode in range(nmodes):
            img = rng.integers(0, 256, (40, st)).astype(np.uint8)
            a, b = img.copy(), img.copy()
            off = 8 * st + 16
            getattr(orc, "orc_" + kind)(C.c_void_p(a.ctypes.data + off), st, mode)
            getattr(ref, "refk_" + kind)(C.c_void_p(b.ctypes.data + off), st, mode)
            assert np.array_equal(a, b), (kind, mode)


def test_intra_pred8x8l(libs):
    orc, ref = libs
    rng = np.random.default_rng(5)
    st = 64
    for it in range(200):
        for mode in range(14):
            for tl in (0, 1):
                for tr in (0, 1):
                    img = rng.integers(0, 256, (40, st)).astype(np.uint8)
                    a, b = img.copy(), img.copy()
                    off = 8 * st + 16
                    orc.orc_pred8x8l(C.c_void_p(a.ctypes.data + off), st, mode, tl, tr)
                    ref.refk_pred8x8l(C.c_void_p(b.ctypes.data + off), st, mode, tl, tr)
                    assert np.array_equal(a, b), (mode, tl, tr)


def test_mc(libs):
    orc, ref = libs
    rng = np.random.default_rng(6)
    st = 64
    sizes = [(16, 16), (16, 8), (8, 16), (8, 8), (8, 4), (4, 8), (4, 4)]
    for it in range(40):
        src = rng.integers(0, 256, (48, st)).astype(np.uint8)
        if it % 5 == 0:
            src[:] = rng.choice([0, 255], src.shape)      # saturating content
        for (w, h) in sizes:
            for mvx in range(4):
                for mvy in range(4):
                    a = np.zeros((16, 32), np.uint8)
                    b = a.copy()
                    sp = C.c_void_p(src.ctypes.data + 8 * st + 8)
                    orc.orc_mc_luma(sp, st, _p(a), 32, mvx, mvy, w, h)
                    ref.refk_mc_luma(sp, st, _p(b), 32, mvx, mvy, w, h)
                    assert np.array_equal(a, b), (w, h, mvx, mvy)
            for mvx in range(8):
                for mvy in range(8):
                    a = np.zeros((16, 32), np.uint8)
                    b = a.copy()
                    sp = C.c_void_p(src.ctypes.data + 8 * st + 8)
                    orc.orc_mc_chroma(sp, st, _p(a), 32, mvx, mvy, w >> 1, h >> 1)
                    ref.refk_mc_chroma(sp, st, _p(b), 32, mvx, mvy, w >> 1, h >> 1)
                    assert np.array_equal(a, b), (w, h, mvx, mvy)


def test_deblock_edge_filters(libs):
    orc, ref = libs
    rng = np.random.default_rng(7)
    st = 32
    for it in range(3000):
        base = int(rng.integers(0, 256))
        spread = [2, 6, 20, 255][it % 4]
        img = np.clip(base + rng.integers(-spread, spread + 1, (24, st)), 0, 255).astype(np.uint8)
        alpha, beta = int(rng.integers(0, 256)), int(rng.integers(0, 19))
        tc = rng.integers(-1, 26, 4).astype(np.int8)
        vert = it & 1
        off = 8 * st + 8
        for name, args in (("luma_lt4", True), ("luma_eq4", False), ("chroma_lt4", True), ("chroma_eq4", False)):
            a, b = img.copy(), img.copy()
            xs, ys = (1, st) if vert else (st, 1)
            if args:
                getattr(orc, "orc_deblock_" + name)(C.c_void_p(a.ctypes.data + off), xs, ys, alpha, beta, _p(tc))
                getattr(ref, "refk_deblock_" + name)(C.c_void_p(b.ctypes.data + off), st, vert, alpha, beta, _p(tc))
            else:
                getattr(orc, "orc_deblock_" + name)(C.c_void_p(a.ctypes.data + off), xs, ys, alpha, beta)
                getattr(ref, "refk_deblock_" + name)(C.c_void_p(b.ctypes.data + off), st, vert, alpha, beta)
            assert np.array_equal(a, b), (name, vert)


def test_deblock_macroblock_drivers(libs):
    """WelsDeblockingMb / DeblockingIntraMb / DeblockingInterMb with the boundary-strength derivation (deblocking.cpp:160-352,
    568-862) over whole synthetic pictures (the pattern of test/decoder/DecUT_DeblockCommon.cpp:417-979): random macroblock types,
    QPs with Cb != Cr, coefficients, vectors, reference indices, 8x8 transform, non-zero alpha / beta offsets, several slices with
    disable_deblocking_filter_idc 0 / 1 / 2.  Includes the reference's stale-iTc inner chroma edge of intra macroblocks (:786-807)."""
    import synth
    orc, ref = libs
    cases = [dict(seed=11, p_frames=False), dict(seed=12, p_frames=True), dict(seed=13, p_frames=True, t8=True),
             dict(seed=14, p_frames=True, n_slices=3, idc=3), dict(seed=15, p_frames=False, t8=True, n_slices=2, idc=2),
             dict(seed=16, p_frames=True, pcm=True)]
    n_stale = 0
    for kw in cases:
        seed = kw.pop("seed")
        for f in synth.make_stream(seed, 7, 5, 3, **kw):
            rng = np.random.default_rng(1000 + seed + f.id)
            a = O.HostPic(f.mb_w, f.mb_h, fill=0)
            base = rng.integers(40, 216)
            for p in range(3):      # smooth content with small steps at block edges: every branch of the filters is reached
                pl = a.plane(p)
                pl[:] = np.clip(base + rng.integers(-9, 10, pl.shape) + 6 * ((np.arange(pl.shape[1]) // 4) % 3)[None, :], 0, 255)
            b = O.HostPic(f.mb_w, f.mb_h, fill=0)
            b.buf[:] = a.buf
            mbs = np.ascontiguousarray(f.mbs); sl = np.ascontiguousarray(f.slices)
            sa = a.struct()
            for si in range(len(sl)):
                orc.orc_deblock_slice(_p(mbs), _p(sl), si, C.byref(sa), f.mb_w, f.mb_h)
            sb = b.struct()
            ref.refk_deblock_picture(_p(mbs), _p(sl), f.mb_w, f.mb_h, C.c_void_p(sb.y), C.c_void_p(sb.u), C.c_void_p(sb.v), sb.stride_y, sb.stride_c)
            for p in range(3):
                assert np.array_equal(a.plane(p), b.plane(p)), (seed, f.id, p)
            intra = (mbs["mb_type"] & 0x207) != 0
            n_stale += int(np.count_nonzero(intra & (mbs["qp_c"][:, 0] != mbs["qp_c"][:, 1])))
    assert n_stale > 50         # the quirk's precondition is exercised


def test_expand(libs):
    orc, ref = libs
    rng = np.random.default_rng(8)
    for (mb_w, mb_h) in [(1, 1), (3, 2), (11, 9), (20, 15)]:
        a = O.HostPic(mb_w, mb_h, fill=0)
        a.buf[:] = rng.integers(0, 256, a.buf.shape)
        b = O.HostPic(mb_w, mb_h, fill=0)
        b.buf[:] = a.buf
        sa = a.struct()
        orc.orc_expand_pic(C.byref(sa), mb_w, mb_h)
        sb = b.struct()
        ref.refk_expand_picture(C.c_void_p(sb.y), C.c_void_p(sb.u), C.c_void_p(sb.v), mb_w * 16, mb_h * 16, sb.stride_y, sb.stride_c)
        assert np.array_equal(a.buf, b.buf)
