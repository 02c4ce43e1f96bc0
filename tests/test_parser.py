"""Host front end (row f1) on CPU: our Annex-B / CAVLC parser against (a) the records the reference's own parser produced
for the same streams (fixtures from oracle/_ref/ref_dump), and (b) end to end, parse -> oracle reconstruct -> SHA-1 of
the cropped YUV against the table of the reference's decoder test (test/api/decoder_test.cpp:90-128)."""
import glob
import hashlib
import json
import os

import numpy as np
import pytest

import golden_io
import oracle_lib as O
import losslessh264_amd as lh

STREAMS = os.path.join(golden_io.GOLDEN_DIR, "streams")
SHA = json.load(open(os.path.join(golden_io.GOLDEN_DIR, "decoder_sha1.json")))
REF_RES = "/root/reference/res"


def _records_equal(f, r):
    cov = r.covered.astype(bool)
    t = r.mbs["mb_type"]
    for fld in ("mb_type", "cbp", "qp_y", "qp_c", "slice_id", "nzc"):
        assert np.array_equal(f.mbs[fld][cov], r.mbs[fld][cov]), fld
    assert np.array_equal(f.mbs["flags"][cov] & 1, r.mbs["flags"][cov] & 1)
    nxn = cov & ((t == 1) | (t == 4))
    assert np.array_equal(f.mbs["intra_mode"][nxn], r.mbs["intra_mode"][nxn])
    assert np.array_equal(f.mbs["intra_mode"][cov & (t == 2), 0], r.mbs["intra_mode"][cov & (t == 2), 0])
    assert np.array_equal(f.mbs["intra_avail"][cov & (t == 4)], r.mbs["intra_avail"][cov & (t == 4)])
    intra = cov & ((t & 7) != 0)
    assert np.array_equal(f.mbs["chroma_mode"][intra], r.mbs["chroma_mode"][intra])
    inter = cov & ((t & 0x1F8) != 0)
    for fld in ("ref_idx", "mv"):
        assert np.array_equal(f.mbs[fld][inter], r.mbs[fld][inter]), fld
    p8 = cov & ((t == 0x40) | (t == 0x80))
    assert np.array_equal(f.mbs["sub_type"][p8], r.mbs["sub_type"][p8])
    notpcm = cov & (t != 0x200)
    assert np.array_equal(f.coeffs[notpcm], r.coeffs[notpcm])
    assert np.array_equal(f.coeffs[cov & (t == 0x200)] & 0xFF, r.coeffs[cov & (t == 0x200)] & 0xFF)
    if hasattr(r, "levels"):
        assert np.array_equal(f.levels[notpcm], r.levels[notpcm])
    for fld in ("first_mb", "n_mbs", "slice_type", "deblock_idc", "alpha_c0_offset", "beta_offset", "n_refs", "luma_dc_weight"):
        assert np.array_equal(f.slices[fld], r.slices[fld]), fld
    assert f.ref_ids == r.ref_ids


@pytest.mark.parametrize("name", [n for n in golden_io.list_fixtures() if os.path.exists(os.path.join(STREAMS, n))])
def test_records_match_reference_parser(name):
    frames, err = lh.parse_stream(open(os.path.join(STREAMS, name), "rb").read())
    assert err == ""
    ref = golden_io.load(name)
    assert len(frames) >= len(ref)
    for f, r in zip(frames, ref):
        assert (f.mb_w, f.mb_h, f.frame_num) == (r.mb_w, r.mb_h, r.frame_num)
        _records_equal(f, r)


def _yuv_sha1(frames):
    h = hashlib.sha1()
    pics = {}
    for f in frames:
        dst = O.HostPic(f.mb_w, f.mb_h)
        O.recon_frame(f.mbs, f.coeffs, f.slices, dst, [pics[r] for r in f.ref_ids], 0)
        pics[f.id] = dst
        for p in range(3):
            s = 1 if p else 0
            pl = dst.plane(p)[f.crop_y >> s:(f.crop_y + f.crop_h) >> s, f.crop_x >> s:(f.crop_x + f.crop_w) >> s]
            h.update(np.ascontiguousarray(pl).tobytes())
        for k in [k for k in pics if k < f.id - 20 and k not in f.ref_ids]:
            pass
    return h.hexdigest()


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(STREAMS, "*"))), ids=os.path.basename)
def test_end_to_end_sha1(path):
    name = os.path.basename(path)
    if name not in SHA:
        pytest.skip("no entry in the reference's SHA-1 table")
    frames, err = lh.parse_stream(open(path, "rb").read())
    assert err == ""
    assert _yuv_sha1(frames) == SHA[name]


@pytest.mark.skipif(not os.path.isdir(REF_RES), reason="reference streams only exist in the build container")
@pytest.mark.parametrize("name", [n for n in sorted(SHA) if not n.startswith("_")])
def test_all_reference_streams_sha1(name):
    path = os.path.join(REF_RES, name)
    frames, err = lh.parse_stream(open(path, "rb").read())
    assert err == "", err
    assert _yuv_sha1(frames) == SHA[name]


def test_garbage_and_truncation_do_not_crash():
    rng = np.random.default_rng(5)
    data = open(os.path.join(STREAMS, "SVA_BA2_D.264"), "rb").read()
    for cut in (0, 1, 5, 37, 1000, len(data) - 3):
        lh.parse_stream(data[:cut])
    for _ in range(20):
        b = bytearray(data)
        for pos in rng.integers(40, len(b), 30):
            b[pos] = int(rng.integers(0, 256))
        lh.parse_stream(bytes(b))
    lh.parse_stream(bytes(rng.integers(0, 256, 5000, dtype=np.uint8)))


# ---- row a10: the syntax records the recompressor codes (lh264_mbsyn_t) against the reference's DecodedMacroblock -------
PIP_FIXTURES = sorted(os.path.basename(p)[4:-4] for p in glob.glob(os.path.join(golden_io.GOLDEN_DIR, "pip_*.npz")))


@pytest.mark.parametrize("name", PIP_FIXTURES)
def test_syntax_records_match_reference(name):
    """every field of every coded macroblock, and the per-slice alignment bits, as captured from the unmodified reference
    (tests/golden/make_golden_pip.py)"""
    import refdump
    z = np.load(os.path.join(golden_io.GOLDEN_DIR, "pip_" + name + ".npz"))
    rtd = z["rtd"].reshape(-1).view(refdump.RTD_DTYPE)
    frames, err = lh.parse_stream(open(os.path.join(golden_io.GOLDEN_DIR, "streams", name), "rb").read())
    assert err == ""
    hdr = z["hdr"]
    mb0 = s0 = 0
    for i in range(len(hdr)):
        n, nsl = int(hdr[i][0] * hdr[i][1]), int(hdr[i][3])
        f = frames[i]
        assert (f.mb_w, f.mb_h, f.frame_num) == tuple(int(x) for x in hdr[i][:3])
        r = rtd[mb0:mb0 + n]
        assert np.array_equal(r["have"], f.syn["have"])
        coded = r["have"] == 1
        for fld in refdump.RTD_DTYPE.names:
            assert np.array_equal(r[fld][coded], f.syn[fld][coded]), (name, i, fld)
        ref_sl = z["slices"][s0:s0 + nsl, 3:6].copy()
        ref_sl[:, 2] &= 1              # bit 1 of the fixture column = constrained_intra_pred_flag (make_golden_pip.py)
        assert np.array_equal(ref_sl, f.slice_syn[:, :3]), (name, i)
        assert np.array_equal((z["slices"][s0:s0 + nsl, 5] >> 1) & 1, (f.slice_syn[:, 3] >> 1) & 1)
        assert np.array_equal((z["slices"][s0:s0 + nsl, 5] >> 2) & 1, f.slice_syn[:, 3] & 1)
        mb0 += n
        s0 += nsl
