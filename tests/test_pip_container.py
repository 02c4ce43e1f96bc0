"""The recompressor's default stream (the '.pip' file itself) and, further down, the restore direction
(row f2), against whole-stream outputs of the reference's console application (tests/golden/cli_*.npz, written by
tests/golden/make_golden_cli.py from the unmodified reference built in the build container)."""
import glob
import os

import numpy as np
import pytest

import golden_io
import losslessh264_amd as lh

STREAMS = os.path.join(golden_io.GOLDEN_DIR, "streams")
CLI = sorted(os.path.basename(p)[4:-4] for p in glob.glob(os.path.join(golden_io.GOLDEN_DIR, "cli_*.npz")))


TAG_PCM = 70
HAVE_PCM = {"QCIF_2P_I_allIPCM.264"}           # CABAC, two pictures of I_PCM macroblocks


def cli_fixture(name):
    z = np.load(os.path.join(golden_io.GOLDEN_DIR, "cli_" + name + ".npz"))
    return z["main"].tobytes(), {int(k[4:]): z[k].tobytes() for k in z.files if k.startswith("tag_")}


@pytest.mark.parametrize("name", CLI)
def test_default_stream_matches_reference_cli(name):
    data = open(os.path.join(STREAMS, name), "rb").read()
    frames, err, main = lh.parse_file(data)
    assert err == ""
    ref_main, _ = cli_fixture(name)
    assert main == ref_main


def test_feed_file_gives_the_same_pictures_as_feed():
    data = open(os.path.join(STREAMS, "test_vd_1d.264"), "rb").read()
    a, _ = lh.parse_stream(data)
    b, _, _ = lh.parse_file(data)
    assert len(a) == len(b)
    for f, g in zip(a, b):
        assert np.array_equal(f.mbs, g.mbs) and np.array_equal(f.coeffs, g.coeffs) and np.array_equal(f.syn, g.syn)


# ---- restore direction (row f2): reference-written files -> the original bytes ------------------------------------------
def _is_cabac(name):
    frames, _ = lh.parse_stream(open(os.path.join(STREAMS, name), "rb").read())
    return any(int(f.slice_syn[0, 3]) & 1 for f in frames)


@pytest.mark.parametrize("name", CLI)
def test_restore_reference_files_gives_original_stream(name):
    """the files the reference's compressor wrote (main + every tag) decode back to the input, bit for bit"""
    main, tags = cli_fixture(name)
    orig = open(os.path.join(STREAMS, name), "rb").read()
    if name in HAVE_PCM:
        # the reference does not store the samples of I_PCM macroblocks (and cannot restore such a stream); they are this code's one
        # addition to the file set, stream LH264_TAG_PCM, made by the front end
        with pytest.raises(RuntimeError, match="I_PCM"):
            lh.restore(main, tags)
        pcm = lh.parse_file(orig, pcm=True)[3]
        assert len(pcm) % 384 == 0 and len(pcm) > 0
        tags[TAG_PCM] = pcm
    out = lh.restore(main, tags)
    assert out == orig


def test_restore_reports_what_it_cannot_do():
    main, tags = cli_fixture("SVA_BA2_D.264")
    t2 = dict(tags)
    del t2[19]
    with pytest.raises(RuntimeError, match="missing"):
        lh.restore(main, t2)


def test_restore_survives_corrupt_input():
    """damaged tag streams or a damaged default stream must end in an error or in different bytes, never in a crash"""
    rng = np.random.default_rng(11)
    for name in ("SVA_BA1_B.264", "test_qcif_cabac.264"):
        _corrupt_trials(rng, name)


def _corrupt_trials(rng, name):
    main, tags = cli_fixture(name)
    orig = open(os.path.join(STREAMS, name), "rb").read()
    for trial in range(12):
        t2 = {t: bytearray(b) for t, b in tags.items()}
        m2 = bytearray(main)
        if trial % 3 == 2:
            for pos in rng.integers(5, len(m2), 3):
                m2[pos] = int(rng.integers(0, 256))
        else:
            for t in list(t2)[trial % 5::5]:
                b = t2[t]
                for pos in rng.integers(0, len(b), max(1, len(b) // 50)):
                    b[pos] = int(rng.integers(0, 256))
                if trial % 2:
                    del b[len(b) // 2:]
        try:
            out = lh.restore(bytes(m2), {t: bytes(b) for t, b in t2.items()})
            assert isinstance(out, bytes)
        except RuntimeError:
            pass
    assert lh.restore(main, tags) == orig


# ---- single-file container (row f3) ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["SVA_BA2_D.264", "test_qcif_cabac.264"])
def test_single_file_container_round_trip(name):
    main, tags = cli_fixture(name)
    orig = open(os.path.join(STREAMS, name), "rb").read()
    blob = lh.pack(main, tags)
    assert blob[:6] == b"LHPIP1" and len(blob) == 16 + 8 * (1 + len(tags)) + len(main) + sum(len(b) for b in tags.values())
    assert lh.restore_file(blob) == orig
    vb = lh.pack(orig, {}, lh.VERBATIM)
    assert len(vb) == len(orig) + 24 and lh.restore_file(vb) == orig


def test_single_file_container_rejects_damage():
    main, tags = cli_fixture("SVA_BA2_D.264")
    blob = lh.pack(main, tags)
    for bad in (b"", blob[:10], b"XXXXXXXX" + blob[8:], blob[:12] + b"\xff\xff\xff\x7f" + blob[16:], blob[:20] + b"\xff\xff\xff\x7f" + blob[24:]):
        with pytest.raises(RuntimeError):
            lh.restore_file(bad)
    rng = np.random.default_rng(3)
    for _ in range(10):                    # damaged payloads: an error or other bytes, never a crash
        b = bytearray(blob)
        for pos in rng.integers(16, len(b), 5):
            b[pos] = int(rng.integers(0, 256))
        try:
            lh.restore_file(bytes(b))
        except RuntimeError:
            pass
