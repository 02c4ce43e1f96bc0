"""Rows a9/a10/f4 on the device: product path .264 -> host front end (records + syntax symbols) -> HIP context-index
kernels (coefficient symbols) -> HIP coder.  Every tagged arithmetic-coded stream must equal, byte for byte, what the
reference's compressor wrote for the same stream (tests/golden/pip_*.npz, generated from the unmodified reference)."""
import glob
import os

import numpy as np
import pytest

import golden_io

pytestmark = pytest.mark.gpu
FIXTURES = sorted(os.path.basename(p)[4:-4] for p in glob.glob(os.path.join(golden_io.GOLDEN_DIR, "pip_*.npz")))


def _frames(name, z):
    import losslessh264_amd as lh
    frames, err = lh.parse_stream(open(os.path.join(golden_io.GOLDEN_DIR, "streams", name), "rb").read())
    assert err == ""
    return frames[:len(z["hdr"])]


@pytest.mark.parametrize("path", ["sw", "wave"])
def test_device_coder_equals_reference_bytes(path, monkeypatch):
    """both forms of the coder's first stages (stream per workgroup / wave per (stream, partition), csrc/lh264_capi.hip code_binarise)"""
    import losslessh264_amd as lh
    monkeypatch.setenv("LH264_CODER_PATH", path)
    streams, refs = [], []
    for name in FIXTURES:
        z = np.load(os.path.join(golden_io.GOLDEN_DIR, "pip_" + name + ".npz"))
        streams.append(_frames(name, z))
        refs.append({int(k[4:]): z[k].tobytes() for k in z.files if k.startswith("tag_")})
    ctx = lh.CtxSession(streams)
    ctx.run()
    coder = lh.CoderSession(ctx, out_cap=1 << 17)
    coder.run()
    ctx.synchronize()
    for c, name in enumerate(FIXTURES):
        got = coder.tags(c)
        assert sorted(got) == sorted(refs[c]), (name, sorted(got), sorted(refs[c]))
        for t in sorted(refs[c]):
            assert got[t] == refs[c][t], "%s tag %d: %d bytes, reference %d" % (name, t, len(got[t]), len(refs[c][t]))


@pytest.mark.parametrize("log2p,window", [(0, 3), (2, 0), (4, 1), (5, 3), (7, 0)])
def test_wave_form_partitions_and_window_do_not_change_the_bytes(log2p, window, monkeypatch):
    """the wave form with 1 .. 128 partitions per stream (buckets of cells dealt to the partitions by coder_balance_kernel; 128 = the
    buckets themselves) and with / without the hint that keeps a stream's waves together: ten streams, so that the workgroups of the
    resolve kernel cover every XCD residue and a second group of eight - every tag byte-identical to the reference's"""
    import ctypes as C
    import losslessh264_amd as lh
    from losslessh264_amd import _lib as L
    monkeypatch.setenv("LH264_CODER_PATH", "wave")
    monkeypatch.setenv("LH264_CODER_LOG2P", str(log2p))
    monkeypatch.setenv("LH264_CODER_WINDOW", str(window))
    streams, refs = [], []
    for name in FIXTURES:
        z = np.load(os.path.join(golden_io.GOLDEN_DIR, "pip_" + name + ".npz"))
        streams.append(_frames(name, z))
        refs.append({int(k[4:]): z[k].tobytes() for k in z.files if k.startswith("tag_")})
    ctx = lh.CtxSession(streams)
    ctx.run()
    coder = lh.CoderSession(ctx, out_cap=1 << 17)
    coder.run()
    ctx.synchronize()
    for c, name in enumerate(FIXTURES):
        assert coder.tags(c) == refs[c], name
    # the partitions of the first stream: as many as asked for, and between them every decision of the stream
    f = L.lib().lh264_debug_coder_parts
    f.restype = C.c_int
    out = (C.c_ulonglong * 128)()
    assert f(0, out, 128) == 1 << log2p
    parts = [out[i] for i in range(1 << log2p)]
    assert sum(parts) > 0
    if log2p == 4:
        assert max(parts) * 16 <= 2 * sum(parts), parts          # dealt evenly (a small stream: within 2x of the mean)


def test_replicas_are_identical_and_rerun_is_stable():
    import losslessh264_amd as lh
    name = "SVA_BA2_D.264"
    z = np.load(os.path.join(golden_io.GOLDEN_DIR, "pip_" + name + ".npz"))
    ctx = lh.CtxSession([_frames(name, z)], replicate=5)
    ctx.run()
    coder = lh.CoderSession(ctx)
    coder.run(); coder.run()
    ctx.synchronize()
    ref = {int(k[4:]): z[k].tobytes() for k in z.files if k.startswith("tag_")}
    for c in range(5):
        assert coder.tags(c) == ref


def test_coder_reports_what_does_not_fit():
    """an output buffer that is too small ends in status bit 4 (the other streams of the call are unharmed, nothing is written behind
    a stream's buffers); a spill table of one cell either still gives the reference's bytes or reports status bit 1"""
    import torch
    import losslessh264_amd as lh
    name = "SVA_BA1_B.264"
    z = np.load(os.path.join(golden_io.GOLDEN_DIR, "pip_" + name + ".npz"))
    ref = {int(k[4:]): z[k].tobytes() for k in z.files if k.startswith("tag_")}
    ctx = lh.CtxSession([_frames(name, z)], replicate=2)
    ctx.run()
    small = lh.CoderSession(ctx, out_cap=256)                         # the largest tag of this stream is a few KB
    guard = small.d_out.clone()
    small.run()
    ctx.synchronize()
    with pytest.raises(RuntimeError, match="status"):
        small.tags(0)
    n = lh._lib.N_TAG_SLOTS
    lens = small.d_len.cpu().numpy().reshape(2, n + 1)
    assert (lens[:, n] & 4).all() and (lens[:, :35].max(axis=1) > 256).all()      # the lengths say how much room was needed
    del guard
    tiny = lh.CoderSession(ctx, hash_cap=1)
    tiny.run()
    ctx.synchronize()
    st = tiny.d_len.cpu().numpy().reshape(2, n + 1)[:, n]
    assert ((st == 0) | ((st & 1) != 0)).all()
    if (st == 0).all():
        assert tiny.tags(0) == ref and tiny.tags(1) == ref


# ---- whole streams: compress on the device, compare with the reference CLI's files, restore, compare with the input -----
CLI = sorted(os.path.basename(p)[4:-4] for p in glob.glob(os.path.join(golden_io.GOLDEN_DIR, "cli_*.npz")))
CLI_CAVLC = [n for n in CLI if "cabac" not in n.lower()]


@pytest.mark.parametrize("path", ["sw", "wave"])
def test_whole_stream_compress_equals_reference_cli_and_round_trips(path, monkeypatch):
    """.264 -> (default stream from the host front end, tagged streams from the HIP coder) == the files the reference's console
    application writes; and those restore (csrc/host/pip_restore.cpp) to the input, bit for bit.  Both forms of the coder's first stages."""
    import losslessh264_amd as lh
    monkeypatch.setenv("LH264_CODER_PATH", path)
    datas, streams, mains, pcms = [], [], [], []
    for name in CLI:
        data = open(os.path.join(golden_io.GOLDEN_DIR, "streams", name), "rb").read()
        frames, err, main, pcm = lh.parse_file(data, pcm=True)
        assert err == ""
        datas.append(data); streams.append(frames); mains.append(main); pcms.append(pcm)
    ctx = lh.CtxSession(streams)
    ctx.run()
    coder = lh.CoderSession(ctx, hash_cap=1 << 18, out_cap=1 << 20)
    coder.run()
    ctx.synchronize()
    for c, name in enumerate(CLI):
        z = np.load(os.path.join(golden_io.GOLDEN_DIR, "cli_" + name + ".npz"))
        ref = {int(k[4:]): z[k].tobytes() for k in z.files if k.startswith("tag_")}
        got = coder.tags(c)
        assert mains[c] == z["main"].tobytes(), name
        assert sorted(got) == sorted(ref), (name, sorted(got), sorted(ref))
        for t in sorted(ref):
            assert got[t] == ref[t], "%s tag %d: %d bytes, reference %d" % (name, t, len(got[t]), len(ref[t]))
        if pcms[c]:
            got[70] = pcms[c]                         # LH264_TAG_PCM: the samples of I_PCM macroblocks, which the reference's files lack
        assert lh.restore(mains[c], got) == datas[c], name


def test_coder_in_two_calls_equals_one_call():
    """lh264_code_binarise_chains + lh264_code_finish_chains (with another kernel on a second stream between them) == lh264_code_chains;
    a finish without its binarise is refused"""
    import torch
    import losslessh264_amd as lh
    names = ["SVA_BA1_B.264", "test_qcif_cabac.264"]
    streams = [lh.parse_file(open(os.path.join(golden_io.GOLDEN_DIR, "streams", n), "rb").read())[0] for n in names]
    ctx = lh.CtxSession(streams, replicate=3)
    ctx.run()
    coder = lh.CoderSession(ctx)
    coder.run()
    ctx.synchronize()
    one = [coder.tags(c) for c in range(6)]
    rec = lh.ReconSession(streams, replicate=3)
    side = torch.cuda.Stream(ctx.dev)
    coder.binarise()
    with torch.cuda.stream(side):
        rec.run()
    coder.finish()
    torch.cuda.synchronize(ctx.dev)
    assert [coder.tags(c) for c in range(6)] == one
    with pytest.raises(RuntimeError, match="without the matching"):
        coder.finish()


def test_coder_calls_on_two_hip_streams_do_not_share_work_memory_at_the_same_time():
    """the coder's work memory (decision words, tag lists, sums) is one set per device: a call on another HIP stream must wait on the
    device for the call before it (an event), not only for the host-side lock - otherwise the second call's binarisation overwrites
    what the first call's kernels are still reading"""
    import torch
    import losslessh264_amd as lh
    names = ["SVA_BA1_B.264", "test_qcif_cabac.264"]
    streams = [lh.parse_file(open(os.path.join(golden_io.GOLDEN_DIR, "streams", n), "rb").read())[0] for n in names]
    a_ctx = lh.CtxSession(streams, replicate=40)                  # long enough to be still running when the second call arrives
    b_ctx = lh.CtxSession(streams[::-1], replicate=2)
    a_ctx.run(); b_ctx.run()
    a, b = lh.CoderSession(a_ctx), lh.CoderSession(b_ctx)
    a.run(); b.run()
    torch.cuda.synchronize(a_ctx.dev)
    want_a, want_b = [a.tags(c) for c in range(80)], [b.tags(c) for c in range(4)]
    s1, s2 = torch.cuda.Stream(a_ctx.dev), torch.cuda.Stream(a_ctx.dev)
    for _ in range(3):
        a.d_out.zero_(); b.d_out.zero_()
        torch.cuda.synchronize(a_ctx.dev)
        with torch.cuda.stream(s1):
            a.run()
        with torch.cuda.stream(s2):
            b.run()
        with torch.cuda.stream(s1):
            a.binarise()
        with torch.cuda.stream(s2):
            a.finish()                                               # the second half on another stream than the first
        torch.cuda.synchronize(a_ctx.dev)
        assert [a.tags(c) for c in range(80)] == want_a
        assert [b.tags(c) for c in range(4)] == want_b


def test_command_line_compress_and_restore(tmp_path):
    """`python -m losslessh264_amd in.264 out.pip out.yuv` writes the reference console application's files (names and bytes), the
    YUV dump has the SHA-1 of the reference's decoder test, and `... out.pip back.264` gives the input back"""
    import hashlib
    import json
    import subprocess
    import sys
    name = "SVA_BA1_B.264"
    src = os.path.join(golden_io.GOLDEN_DIR, "streams", name)
    pip, yuv, back = str(tmp_path / "out.pip"), str(tmp_path / "out.yuv"), str(tmp_path / "back.264")
    root = os.path.dirname(golden_io.GOLDEN_DIR.rstrip("/")).rsplit("/tests", 1)[0]
    env = dict(os.environ, PYTHONPATH=root)
    subprocess.check_call([sys.executable, "-m", "losslessh264_amd", src, pip, yuv], env=env, cwd=root)
    z = np.load(os.path.join(golden_io.GOLDEN_DIR, "cli_" + name + ".npz"))
    assert open(pip, "rb").read() == z["main"].tobytes()
    written = sorted(int(p.rsplit(".", 1)[1]) for p in glob.glob(pip + ".*"))
    assert written == sorted(int(k[4:]) for k in z.files if k.startswith("tag_"))
    for t in written:
        assert open("%s.%d" % (pip, t), "rb").read() == z["tag_%d" % t].tobytes(), t
    sha = json.load(open(os.path.join(golden_io.GOLDEN_DIR, "decoder_sha1.json")))
    assert hashlib.sha1(open(yuv, "rb").read()).hexdigest() == sha[name]
    subprocess.check_call([sys.executable, "-m", "losslessh264_amd", pip, back], env=env, cwd=root)
    assert open(back, "rb").read() == open(src, "rb").read()


def test_command_line_single_file_with_verbatim_fallback(tmp_path):
    """`... in.264 out.lhp` / `... out.lhp back.264`: one container file; a stream that does not get smaller (CABAC; I_PCM, whose
    samples are carried as they are) is stored verbatim and still restores"""
    import subprocess
    import sys
    root = os.path.dirname(golden_io.GOLDEN_DIR.rstrip("/")).rsplit("/tests", 1)[0]
    env = dict(os.environ, PYTHONPATH=root)
    for name, expect_gain in (("SVA_BA1_B.264", True), ("test_qcif_cabac.264", False), ("QCIF_2P_I_allIPCM.264", False)):
        src = os.path.join(golden_io.GOLDEN_DIR, "streams", name)
        lhp, back = str(tmp_path / (name + ".lhp")), str(tmp_path / (name + ".back"))
        subprocess.check_call([sys.executable, "-m", "losslessh264_amd", src, lhp], env=env, cwd=root)
        subprocess.check_call([sys.executable, "-m", "losslessh264_amd", lhp, back], env=env, cwd=root)
        assert open(back, "rb").read() == open(src, "rb").read()
        if expect_gain:
            assert os.path.getsize(lhp) < os.path.getsize(src)
        else:
            assert os.path.getsize(lhp) <= os.path.getsize(src) + 24      # CABAC barely compresses: verbatim when there is no gain


# ---- the compress direction behind one C call, and the C++ console application on top of it -----------------------------------
def test_compress_batch_c_api_equals_reference_cli():
    """lh264_compress_batch (host orchestration in C++: parse threads, staging, ctx + coder launches, download) on all fixture
    streams at once: default stream and every tagged stream equal the reference console application's files"""
    import losslessh264_amd as lh
    datas = [open(os.path.join(golden_io.GOLDEN_DIR, "streams", n), "rb").read() for n in CLI]
    res = lh.compress_batch(datas)
    for name, data, (main, tags, err) in zip(CLI, datas, res):
        assert err is None, (name, err)
        z = np.load(os.path.join(golden_io.GOLDEN_DIR, "cli_" + name + ".npz"))
        assert main == z["main"].tobytes(), name
        ref = {int(k[4:]): z[k].tobytes() for k in z.files if k.startswith("tag_")}
        pcm = tags.get(70)                            # LH264_TAG_PCM: our addition for I_PCM macroblocks, not one of the reference's files
        assert {t: b for t, b in tags.items() if t != 70} == ref, (name, sorted(tags), sorted(ref))
        assert (pcm is not None) == (name == "QCIF_2P_I_allIPCM.264") and (pcm is None or pcm == lh.parse_file(data, pcm=True)[3])
        assert lh.restore(main, tags) == data, name


def test_compress_batch_over_devices_equals_single_device():
    """lh264_compress_batch_devices: the batch cut into contiguous shares, one host thread per share; with every device of the box
    (on a one-GPU box: two shares on device 0, which exercises the share arithmetic and the per-device arena lock) the result is the
    single-call result, share boundaries included (more shares than streams, empty shares)"""
    import torch
    import losslessh264_amd as lh
    datas = [open(os.path.join(golden_io.GOLDEN_DIR, "streams", n), "rb").read() for n in CLI]
    one = lh.compress_batch(datas, 8)
    nd = torch.cuda.device_count()
    devs = list(range(nd)) if nd > 1 else [0, 0]
    assert lh.compress_batch(datas, 8, devices=devs) == one
    assert lh.compress_batch(datas, 8, devices=[0, 0, 0]) == one
    assert lh.compress_batch(datas[:2], 8, devices=[0] * 5) == one[:2]
    assert lh.compress_batch([], 8, devices=[0, 0]) == []


def test_cpp_console_application(tmp_path):
    """losslessh264_amd/lh264dec (C++, built by __graft_entry__.build against include/*.h): the reference console application's
    calling convention and files, the YUV dump, the single-file mode with its verbatim fallback, and the batch mode"""
    import hashlib
    import json
    import subprocess
    root = os.path.dirname(golden_io.GOLDEN_DIR.rstrip("/")).rsplit("/tests", 1)[0]
    exe = os.path.join(root, "losslessh264_amd", "lh264dec")
    assert os.path.exists(exe), "lh264dec not built"
    name = "SVA_BA1_B.264"
    src = os.path.join(golden_io.GOLDEN_DIR, "streams", name)
    pip, yuv, back = str(tmp_path / "out.pip"), str(tmp_path / "out.yuv"), str(tmp_path / "back.264")
    subprocess.check_call([exe, src, pip, yuv], timeout=300)
    z = np.load(os.path.join(golden_io.GOLDEN_DIR, "cli_" + name + ".npz"))
    assert open(pip, "rb").read() == z["main"].tobytes()
    written = sorted(int(p.rsplit(".", 1)[1]) for p in glob.glob(pip + ".*"))
    assert written == sorted(int(k[4:]) for k in z.files if k.startswith("tag_"))
    for t in written:
        assert open("%s.%d" % (pip, t), "rb").read() == z["tag_%d" % t].tobytes(), t
    sha = json.load(open(os.path.join(golden_io.GOLDEN_DIR, "decoder_sha1.json")))
    assert hashlib.sha1(open(yuv, "rb").read()).hexdigest() == sha[name]
    subprocess.check_call([exe, pip, back], timeout=300)
    assert open(back, "rb").read() == open(src, "rb").read()
    # one container per stream, several streams in one call; the I_PCM stream does not get smaller and is stored verbatim
    names = ["SVA_BA2_D.264", "tibby8x8cavlc.264", "QCIF_2P_I_allIPCM.264", "test_qcif_cabac.264"]
    outd = tmp_path / "batch"
    outd.mkdir()
    subprocess.check_call([exe, "--batch", str(outd)] + [os.path.join(golden_io.GOLDEN_DIR, "streams", n) for n in names], timeout=600)
    for n in names:
        lhp, b2 = str(outd / (n + ".lhp")), str(tmp_path / (n + ".back"))
        subprocess.check_call([exe, lhp, b2], timeout=300)
        orig = open(os.path.join(golden_io.GOLDEN_DIR, "streams", n), "rb").read()
        assert open(b2, "rb").read() == orig, n
        assert os.path.getsize(lhp) <= len(orig) + 24
    assert os.path.getsize(str(outd / "SVA_BA2_D.264.lhp")) < os.path.getsize(os.path.join(golden_io.GOLDEN_DIR, "streams", "SVA_BA2_D.264")) + 300
    assert os.path.getsize(str(outd / "tibby8x8cavlc.264.lhp")) < 140000


def test_compress_batch_mixed_good_and_bad_inputs():
    """one call with good streams, an empty one, garbage, a truncated stream and one the reference cannot decode either: the good ones
    come out right, the others carry a status and whatever default stream there is, nothing crashes"""
    import losslessh264_amd as lh
    rng = np.random.default_rng(9)
    good = open(os.path.join(golden_io.GOLDEN_DIR, "streams", "SVA_BA2_D.264"), "rb").read()
    other = open(os.path.join(golden_io.GOLDEN_DIR, "streams", "Static.264"), "rb").read()
    datas = [good, b"", bytes(rng.integers(0, 256, 3000, dtype=np.uint8)), good[:len(good) // 2], other, good]
    res = lh.compress_batch(datas, 4)
    z = np.load(os.path.join(golden_io.GOLDEN_DIR, "cli_SVA_BA2_D.264.npz"))
    ref = {int(k[4:]): z[k].tobytes() for k in z.files if k.startswith("tag_")}
    for i in (0, 5):
        assert res[i][2] is None and res[i][0] == z["main"].tobytes() and res[i][1] == ref
    assert res[4][2] is None and lh.restore(res[4][0], res[4][1]) == other
    assert res[1][1] == {} and res[2][1] == {}            # nothing to code in an empty / garbage input
    # the truncated stream: either compressed (then it must round-trip) or reported
    main, tags, err = res[3]
    if err is None:
        assert lh.restore(main, tags) == datas[3]
