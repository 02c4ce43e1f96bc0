"""Every stream the reference ships (res/ and roundtriptest/, 44 files) against what the reference's own console application did
with it (tests/golden/ref_sweep.json, written by tests/golden/make_ref_sweep.py from the unmodified reference built by
oracle/Makefile): per stream the SHA-1 of every file it wrote in compress mode, and whether it restored the input itself.

The exceptions are named, not counted away:
  * test_scalinglist_jm.264   the reference decodes nothing (weighted_bipred_idc), its "compressed" output is 184 bytes
  * Error_I_P.264, BA_MW_D_IDR_LOST.264   damaged streams: the reference conceals errors, the front end does not model that
  * CVPCMNL1_SVA_C.264, QCIF_2P_I_allIPCM.264   I_PCM: the reference does not carry the samples through the round trip; this code adds
    one stream for them (LH264_TAG_PCM) and restores both
"""
import hashlib
import json
import os

import numpy as np
import pytest

import golden_io

SWEEP = json.load(open(os.path.join(golden_io.GOLDEN_DIR, "ref_sweep.json")))
STREAMS = sorted(SWEEP)
# the default stream (the .pip file itself) differs from the reference's for exactly these
MAIN_DIFFERS = {"test_scalinglist_jm.264"}
# the tagged streams differ (or the stream is refused as a whole) for exactly these
TAGS_DIFFER = {"test_scalinglist_jm.264", "Error_I_P.264", "BA_MW_D_IDR_LOST.264"}
# restore (compress (stream)) is not the stream for exactly these (BA_MW_D_IDR_LOST and test_scalinglist_jm do come back: what the
# front end cannot model stays in the default stream).  Error_I_P has pictures with macroblocks no slice covers; the reference conceals
# them, this code does not: the compress call must REFUSE the stream (an error, so that callers store it verbatim), not hand out a
# representation that does not restore
NO_RESTORE = {"Error_I_P.264"}
REFUSED = {"Error_I_P.264"}
# our one addition to the reference's file set: the samples of I_PCM macroblocks (include/lh264.h LH264_TAG_PCM), which the reference
# does not store (its own restore aborts on such streams)
TAG_PCM = 70
HAVE_PCM = {"CVPCMNL1_SVA_C.264", "QCIF_2P_I_allIPCM.264"}


def _sha(b):
    return hashlib.sha1(bytes(b)).hexdigest()


def _data(name):
    return open(os.path.join(golden_io.GOLDEN_DIR, "streams", name), "rb").read()


def test_sweep_fixture_is_the_survey_table():
    """44 streams, 33 of which the reference itself round-trips; every stream is committed with the SHA-1 the sweep saw"""
    assert len(SWEEP) == 44
    assert sum(1 for v in SWEEP.values() if v["reference_roundtrip"]) == 33
    for name, v in SWEEP.items():
        d = _data(name)
        assert len(d) == v["bytes"] and _sha(d) == v["sha1"], name
    assert SWEEP["tibby.264"]["files"]["main"][0] + sum(v[0] for k, v in SWEEP["tibby.264"]["files"].items() if k != "main") == 111796
    assert sum(v[0] for v in SWEEP["test_cif_P_CABAC_slice.264"]["files"].values()) == 824799


def test_default_stream_equals_reference_on_all_but_the_named_streams():
    """host front end (no GPU): the .pip default stream, byte for byte, for 43 of the 44 streams"""
    import losslessh264_amd as lh
    differs = set()
    for name in STREAMS:
        frames, err, main = lh.parse_file(_data(name))
        ref = SWEEP[name]["files"].get("main")
        if ref is None or _sha(main) != ref[1]:
            differs.add(name)
    assert differs == MAIN_DIFFERS, sorted(differs ^ MAIN_DIFFERS)


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["sw", "wave"])
def test_compress_all_streams_on_the_gpu_and_restore(path, monkeypatch):
    """the whole compress direction (front end -> HIP context-index + coder kernels) over all 44 streams in one batch: every file
    equals the reference's (SHA-1) except for the named streams; what equals the reference's files restores to the input (so the
    reference's own files do), including 9 streams the reference itself aborts on (two of them with I_PCM macroblocks, whose
    samples travel in our additional stream LH264_TAG_PCM)"""
    import losslessh264_amd as lh
    monkeypatch.setenv("LH264_CODER_PATH", path)                 # both forms of the coder's first stages (csrc/lh264_capi.hip code_binarise)
    datas = [_data(n) for n in STREAMS]
    res = lh.compress_batch(datas, 16)
    tags_differ, no_restore, restored_ref_fails = set(), set(), set()
    assert {name for name, r in zip(STREAMS, res) if r[2] is not None} == REFUSED
    for name, data, (main, tags, err) in zip(STREAMS, datas, res):
        ref = SWEEP[name]["files"]
        assert err is not None or (TAG_PCM in tags) == (name in HAVE_PCM), name
        ours = {t: b for t, b in (tags or {}).items() if t != TAG_PCM}
        same = err is None and main is not None and set(str(t) for t in ours) == set(k for k in ref if k != "main") and \
            _sha(main) == ref["main"][1] and all(_sha(ours[t]) == ref[str(t)][1] for t in ours)
        if not same:
            tags_differ.add(name)
        ok = False
        if err is None:
            try:
                ok = lh.restore(main, tags) == data
            except Exception:
                ok = False
        if not ok:
            no_restore.add(name)
        elif same and not SWEEP[name]["reference_roundtrip"]:
            restored_ref_fails.add(name)
    assert tags_differ == TAGS_DIFFER, sorted(tags_differ ^ TAGS_DIFFER)
    assert no_restore == NO_RESTORE, sorted(no_restore ^ NO_RESTORE)
    # byte-identical files that the reference cannot restore but this code does
    assert restored_ref_fails == {"BASQP1_Sony_C.jsv", "CVFC1_Sony_C.jsv", "SVA_Base_B.264", "SVA_CL1_E.264", "SVA_FM1_E.264",
                                  "test_cif_I_CABAC_slice.264", "test_cif_P_CABAC_slice.264",
                                  "CVPCMNL1_SVA_C.264", "QCIF_2P_I_allIPCM.264"} - TAGS_DIFFER - NO_RESTORE, sorted(restored_ref_fails)
