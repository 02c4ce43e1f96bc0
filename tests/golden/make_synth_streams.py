#!/usr/bin/env python3
"""Synthetic 720p / 1080p streams of the configs whose source video is not shipped (BASELINE.json configs[2], configs[3]; SURVEY
section 8d / Appendix F): a numpy test pattern encoded by the REFERENCE's own encoder (oracle/_ref/h264enc, built by
oracle/Makefile from the sources where they lie) with the reference's testbin/welsenc.cfg + layer2.cfg.  The .264 files are the
corpus seed (a different numpy may change the noise): they are committed under tests/golden/streams/, the YUV and cfg copies
are not.  A few frames only - these are parity cases for the big geometries (multi-slice 720p all-intra, 1080p I/P)."""
import os
import shutil
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def pattern(path, W, H, N, coherent):
    rng = np.random.default_rng(1234)
    yy, xx = np.mgrid[0:H, 0:W]
    noise0 = rng.normal(0, 6, (H, W))
    with open(path, "wb") as f:
        for t in range(N):
            noise = np.roll(noise0, (3 * t, 5 * t), axis=(0, 1)) if coherent else rng.normal(0, 6, (H, W))
            y = (128 + 60 * np.sin((xx + 8 * t) / 37.0) + 50 * np.cos((yy - 5 * t) / 23.0) + noise).clip(0, 255).astype(np.uint8)
            u = (128 + 40 * np.sin((xx[::2, ::2] + 4 * t) / 51.0)).clip(0, 255).astype(np.uint8)
            v = (128 + 40 * np.cos((yy[::2, ::2] + 3 * t) / 45.0)).clip(0, 255).astype(np.uint8)
            f.write(y.tobytes()); f.write(u.tobytes()); f.write(v.tobytes())


def encode(tmp, yuv, W, H, N, out, iper, slc):
    cmd = [os.path.join(ROOT, "oracle", "_ref", "h264enc"), "welsenc.cfg", "-org", yuv, "-sw", str(W), "-sh", str(H), "-bf", out,
           "-numtl", "1", "-iper", str(iper), "-cabac", "0", "-frms", str(N), "-rc", "-1", "-ltr", "0", "-scene", "0", "-bgd", "0", "-aq", "0",
           "-lconfig", "0", "layer2.cfg", "-dw", "0", str(W), "-dh", "0", str(H), "-frout", "0", "30", "-lqp", "0", "26"] + slc
    subprocess.check_call(cmd, cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def main():
    tmp = tempfile.mkdtemp(prefix="lh264_syn_")
    for f in ("welsenc.cfg", "layer2.cfg"):
        shutil.copy(os.path.join(REF, "testbin", f), tmp)
    dst = os.path.join(HERE, "streams")
    pattern(os.path.join(tmp, "a.yuv"), 1280, 720, 2, False)
    encode(tmp, "a.yuv", 1280, 720, 2, os.path.join(dst, "syn720p_allI_4slices.264"), 1, ["-slcmd", "0", "1", "-slcnum", "0", "4"])
    pattern(os.path.join(tmp, "b.yuv"), 1920, 1080, 2, True)
    encode(tmp, "b.yuv", 1920, 1080, 2, os.path.join(dst, "syn1080p_IP.264"), 16, ["-slcmd", "0", "0"])
    # the bench workloads of configs[2] / configs[3]: the same patterns, 8 pictures (the first two pictures are NOT the files above:
    # the encoder's rate-free QP 26 path is deterministic, the noise of the all-intra pattern is drawn per picture from one generator)
    pattern(os.path.join(tmp, "c.yuv"), 1280, 720, 8, False)
    encode(tmp, "c.yuv", 1280, 720, 8, os.path.join(dst, "syn720p_allI_4slices_8f.264"), 1, ["-slcmd", "0", "1", "-slcnum", "0", "4"])
    pattern(os.path.join(tmp, "d.yuv"), 1920, 1080, 8, True)
    encode(tmp, "d.yuv", 1920, 1080, 8, os.path.join(dst, "syn1080p_IP_8f.264"), 16, ["-slcmd", "0", "0"])
    for n in ("syn720p_allI_4slices.264", "syn1080p_IP.264", "syn720p_allI_4slices_8f.264", "syn1080p_IP_8f.264"):
        print(n, os.path.getsize(os.path.join(dst, n)), "bytes")
    shutil.rmtree(tmp)


if __name__ == "__main__":
    main()
