"""Parity tests proper: the HIP hot path (through the C ABI) against the oracle and the reference-generated
golden fixtures.  Bit-exact (u8 / int16 integer work)."""
import numpy as np
import pytest

import golden_io
import oracle_lib as O
import synth

pytestmark = pytest.mark.gpu


def _oracle_stream(frames, flags=0):
    pics, out = {}, []
    for f in frames:
        refs = [pics[r] for r in f.ref_ids]
        if flags & O.NO_DEBLOCK:
            # references must be the deblocked pictures; compare the unfiltered reconstruction separately
            dst_f = O.HostPic(f.mb_w, f.mb_h)
            O.recon_frame(f.mbs, f.coeffs, f.slices, dst_f, refs, 0)
            pics[f.id] = dst_f
            dst = O.HostPic(f.mb_w, f.mb_h)
            O.recon_frame(f.mbs, f.coeffs, f.slices, dst, refs, flags)
        else:
            dst = O.HostPic(f.mb_w, f.mb_h)
            O.recon_frame(f.mbs, f.coeffs, f.slices, dst, refs, flags)
            pics[f.id] = dst
        out.append(dst)
    return out


def _compare(sess, chain, frames, want, padded=False):
    for i, f in enumerate(frames):
        got = sess.picture(chain, i, padded=padded)
        for p in range(3):
            ref = want[i].padded_plane(p) if padded else want[i].plane(p)
            if not np.array_equal(got[p], ref):
                ys, xs = np.nonzero(got[p] != ref)
                bs = 8 if p else 16
                pad = (16 if p else 32) if padded else 0
                k = ((ys[0] - pad) // bs) * f.mb_w + (xs[0] - pad) // bs
                typ = f.mbs["mb_type"][k] if 0 <= k < len(f.mbs) else -1
                raise AssertionError("frame %d plane %d: %d samples differ, first at (%d,%d) mb %d type %#x got %d want %d"
                                     % (i, p, len(ys), xs[0] - pad, ys[0] - pad, k, typ, got[p][ys[0], xs[0]], ref[ys[0], xs[0]]))


@pytest.mark.parametrize("name", golden_io.list_fixtures())
def test_golden_stream_final(name):
    import losslessh264_amd as lh
    frames = golden_io.load(name)
    if not all(f.covered.all() for f in frames):
        pytest.skip("stream with lost macroblocks")
    sess = lh.ReconSession([frames])
    sess.run(); sess.synchronize()
    _compare(sess, 0, frames, _oracle_stream(frames))
    for i, f in enumerate(frames):        # and straight against the reference's planes
        got = sess.picture(0, i)
        if f.has_final:
            assert [golden_io.crc(g) for g in got] == f.crc_fin, (name, i)


@pytest.mark.parametrize("name", ["SVA_BA2_D.264", "tibby8x8cavlc.264", "CVPCMNL1_SVA_C.264"])
def test_golden_first_frame_pre_deblock(name):
    import losslessh264_amd as lh
    frames = golden_io.load(name)[:1]
    sess = lh.ReconSession([frames], flags=lh._lib.JOB_NO_DEBLOCK)
    sess.run(); sess.synchronize()
    got = sess.picture(0, 0)
    assert [golden_io.crc(g) for g in got] == frames[0].crc_pre


CASES = [
    dict(seed=1, mb_w=6, mb_h=5, n_frames=2, p_frames=False),                       # intra only
    dict(seed=2, mb_w=11, mb_h=9, n_frames=4),                                      # QCIF I+P
    dict(seed=3, mb_w=7, mb_h=6, n_frames=3, t8=True),                              # 8x8 transform + I8x8
    dict(seed=4, mb_w=5, mb_h=4, n_frames=3, pcm=True, n_slices=3, idc=3),          # PCM, slices, mixed idc
    dict(seed=5, mb_w=9, mb_h=7, n_frames=3, weighted=True),                        # weighted prediction (+ its quirk)
    dict(seed=6, mb_w=1, mb_h=1, n_frames=3),                                       # smallest picture
    dict(seed=7, mb_w=2, mb_h=19, n_frames=2),                                      # tall: more rows than waves
    dict(seed=8, mb_w=45, mb_h=3, n_frames=2, amp=32767, density=0.5),              # wide, saturating coefficients
    dict(seed=9, mb_w=20, mb_h=18, n_frames=3, n_slices=4, idc=2, t8=True),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "seed%d" % c["seed"])
def test_synthetic_vs_oracle(case):
    import losslessh264_amd as lh
    frames = synth.make_stream(**case)
    sess = lh.ReconSession([frames])
    sess.run(); sess.synchronize()
    _compare(sess, 0, frames, _oracle_stream(frames), padded=True)


def test_synthetic_pre_deblock():
    import losslessh264_amd as lh
    frames = synth.make_stream(seed=11, mb_w=8, mb_h=6, n_frames=1, p_frames=False, t8=True)
    sess = lh.ReconSession([frames], flags=lh._lib.JOB_NO_DEBLOCK | lh._lib.JOB_NO_EXPAND)
    sess.run(); sess.synchronize()
    _compare(sess, 0, frames, _oracle_stream(frames, O.NO_DEBLOCK | O.NO_EXPAND))


def test_batch_of_streams_and_rerun_idempotent():
    import losslessh264_amd as lh
    streams = [synth.make_stream(seed=20 + i, mb_w=4 + i, mb_h=3 + (i % 3), n_frames=3) for i in range(5)]
    sess = lh.ReconSession(streams, replicate=3)
    sess.run(); sess.run(); sess.synchronize()      # second pass over already-written pictures: same result
    for c in range(sess.n_chains):
        frames = streams[c % len(streams)]
        _compare(sess, c, frames, _oracle_stream(frames), padded=True)


def test_parser_to_gpu_sha1():
    """whole product path: our host front end -> HIP reconstruct -> cropped YUV SHA-1 of the reference's decoder test"""
    import glob, hashlib, json, os
    import losslessh264_amd as lh
    sha = json.load(open(os.path.join(golden_io.GOLDEN_DIR, "decoder_sha1.json")))
    for path in sorted(glob.glob(os.path.join(golden_io.GOLDEN_DIR, "streams", "*"))):
        name = os.path.basename(path)
        if name not in sha:
            continue
        frames, err = lh.parse_stream(open(path, "rb").read())
        assert err == ""
        sess = lh.ReconSession([frames])
        sess.run(); sess.synchronize()
        h = hashlib.sha1()
        for i, f in enumerate(frames):
            got = sess.picture(0, i)
            for p in range(3):
                s = 1 if p else 0
                h.update(np.ascontiguousarray(got[p][f.crop_y >> s:(f.crop_y + f.crop_h) >> s, f.crop_x >> s:(f.crop_x + f.crop_w) >> s]).tobytes())
        assert h.hexdigest() == sha[name], name


def test_big_geometries_match_oracle():
    """BASELINE.json configs[2] / configs[3] geometries: 720p all-intra with 4 slices per picture and 1080p I/P, encoded by the
    reference's encoder from a synthetic pattern (tests/golden/make_synth_streams.py): host front end -> HIP reconstruct, every
    padded plane against the oracle"""
    import os
    import losslessh264_amd as lh
    for name in ("syn720p_allI_4slices.264", "syn1080p_IP.264"):
        frames, err = lh.parse_stream(open(os.path.join(golden_io.GOLDEN_DIR, "streams", name), "rb").read())
        assert err == "" and len(frames) == 2
        sess = lh.ReconSession([frames])
        sess.run(); sess.synchronize()
        pics = {}
        for i, f in enumerate(frames):
            dst = O.HostPic(f.mb_w, f.mb_h)
            O.recon_frame(f.mbs, f.coeffs, f.slices, dst, [pics[r] for r in f.ref_ids], 0)
            pics[f.id] = dst
            got = sess.picture(0, i, padded=True)
            for p in range(3):
                assert np.array_equal(got[p], dst.padded_plane(p)), (name, i, p)


@pytest.mark.parametrize("name", ["syn720p_allI_4slices_8f.264", "syn1080p_IP_8f.264", "tibby.264"])
def test_bench_streams_match_oracle(name):
    """the streams bench.py runs (--config 2 / 3: the 8-picture 720p and 1080p streams; BASELINE.json configs[0]: tibby.264), all
    pictures: host front end -> HIP reconstruct, every padded plane of every picture against the oracle.  68 macroblock rows on 8
    waves over eight 1080p pictures: the cross-picture wait_prefix path of recon_chain_kernel beyond the first two pictures."""
    import os
    import losslessh264_amd as lh
    frames, err = lh.parse_stream(open(os.path.join(golden_io.GOLDEN_DIR, "streams", name), "rb").read())
    assert err == "" and len(frames) >= 8
    sess = lh.ReconSession([frames], replicate=2)
    sess.run(); sess.synchronize()
    pics = {}
    for i, f in enumerate(frames):
        dst = O.HostPic(f.mb_w, f.mb_h)
        O.recon_frame(f.mbs, f.coeffs, f.slices, dst, [pics[r] for r in f.ref_ids], 0)
        pics[f.id] = dst
        for c in (0, 1):
            got = sess.picture(c, i, padded=True)
            for p in range(3):
                assert np.array_equal(got[p], dst.padded_plane(p)), (name, c, i, p)
        for old in [k for k in pics if k not in f.ref_ids and k != f.id and all(k not in g.ref_ids for g in frames[i + 1:])]:
            del pics[old]


def test_chain_with_resolution_change():
    """frames of different sizes in one chain (the row cursor of the pipelined wavefront crosses pictures of different
    heights; a new intra picture starts each segment, as after an SPS change)"""
    import losslessh264_amd as lh
    a = synth.make_stream(seed=31, mb_w=6, mb_h=5, n_frames=3)
    b = synth.make_stream(seed=32, mb_w=4, mb_h=11, n_frames=3)
    c = synth.make_stream(seed=33, mb_w=9, mb_h=2, n_frames=2)
    frames = []
    for seg in (a, b, c):
        base = len(frames)
        for f in seg:
            f.id += base
            f.ref_ids = [r + base for r in f.ref_ids]
            frames.append(f)
    sess = lh.ReconSession([frames])
    sess.run(); sess.synchronize()
    _compare(sess, 0, frames, _oracle_stream(frames), padded=True)


@pytest.mark.parametrize("ring", [4, 5])
def test_picture_buffers_reused_as_a_ring(ring):
    """a DPB-ring host policy: with 4 buffers the picture a frame writes is still a reference of the previous frame (the
    kernel must not overlap the two frames), with 5 it is not (frames overlap)"""
    import losslessh264_amd as lh
    frames = synth.make_stream(seed=41, mb_w=7, mb_h=9, n_frames=9)
    sess = lh.ReconSession([frames], ring=ring)
    sess.run(); sess.synchronize()
    ref = _oracle_stream(frames)
    for i in range(len(frames) - ring, len(frames)):          # the pictures still resident
        got = sess.picture(0, i, padded=True)
        for p in range(3):
            assert np.array_equal(got[p], ref[i].padded_plane(p)), (ring, i, p)


def test_lost_macroblocks_pass_the_picture_through():
    """macroblocks no slice covers (mb_type 0) keep the picture's current samples and are not filtered"""
    import losslessh264_amd as lh
    frames = synth.make_stream(seed=51, mb_w=6, mb_h=4, n_frames=1, p_frames=False, n_slices=2)
    f = frames[0]
    lost = np.arange(f.slices["first_mb"][1], f.slices["first_mb"][1] + f.slices["n_mbs"][1])
    f.mbs["mb_type"][lost] = 0
    f.slices = f.slices[:1].copy()
    sess = lh.ReconSession([frames])
    sess.run(); sess.synchronize()
    _compare(sess, 0, frames, _oracle_stream(frames))
    got = sess.picture(0, 0)
    y0 = (int(lost[0]) // f.mb_w + 1) * 16
    assert (got[0][y0:] == 128).all()                          # rows entirely inside the lost slice keep the fill value
