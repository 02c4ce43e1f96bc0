"""Command line with the reference console application's calling convention (codec/console/dec/src/h264dec.cpp:150-178,
79-104): the file names decide the mode.

    python -m losslessh264_amd in.264  out.pip [out.yuv]    compress: out.pip = default stream, out.pip.<tag> = tagged streams
    python -m losslessh264_amd in.pip  out.264              restore the original bytes from in.pip + in.pip.<tag>
    python -m losslessh264_amd in.264  out.lhp              compress into ONE file (LHPIP1 container, include/lh264.h); the result is
                                                            restored and compared before it is written, and a stream the round trip
                                                            cannot carry (I_PCM, damaged or unsupported syntax) is stored verbatim
    python -m losslessh264_amd in.lhp  out.264              restore from the container

Compress runs the host front end and the HIP context-index + coder kernels (needs the GPU); the optional YUV dump runs the
HIP reconstruct kernel and writes the cropped I420 pictures like the reference's decoder.  Restore is host code.
"""
import glob
import os
import sys

import numpy as np


def compress_single(src, dst):
    import losslessh264_amd as lh
    data = open(src, "rb").read()
    blob = None
    why = ""
    try:
        frames, err, main, pcm = lh.parse_file(data, pcm=True)
        if err:
            why = err
        elif frames:
            ctx = lh.CtxSession([frames])
            ctx.run()
            coder = lh.CoderSession(ctx, hash_cap=1 << 18, out_cap=max(1 << 16, 2 * len(data)))
            coder.run()
            ctx.synchronize()
            tags = coder.tags(0)
            if pcm:
                tags[70] = pcm          # LH264_TAG_PCM: the samples of the I_PCM macroblocks travel as they are
            if lh.restore(main, tags) == data:
                blob = lh.pack(main, tags)
            else:
                why = "the restored stream differs"
        else:
            why = "no picture"
    except RuntimeError as e:
        why = str(e)
    if blob is None or len(blob) >= len(data) + 32:
        blob = lh.pack(data, {}, lh.VERBATIM)
        why = why or "no gain"
    with open(dst, "wb") as f:
        f.write(blob)
    print("%s: %d bytes -> %d bytes (%.4f)%s" % (src, len(data), len(blob), len(blob) / max(1, len(data)), "  [verbatim: %s]" % why if why else ""))


def restore_single(src, dst):
    import losslessh264_amd as lh
    out = lh.restore_file(open(src, "rb").read())
    with open(dst, "wb") as f:
        f.write(out)
    print("%s -> %s: %d bytes" % (src, dst, len(out)))


def compress(src, dst, yuv=None):
    import losslessh264_amd as lh
    data = open(src, "rb").read()
    frames, err, main, pcm = lh.parse_file(data, pcm=True)
    if err:
        raise SystemExit("cannot compress %s: %s" % (src, err))
    ctx = lh.CtxSession([frames])
    ctx.run()
    coder = lh.CoderSession(ctx, hash_cap=1 << 18, out_cap=max(1 << 16, 2 * len(data)))
    coder.run()
    ctx.synchronize()
    tags = coder.tags(0)
    if pcm:
        tags[70] = pcm                  # LH264_TAG_PCM (the reference writes no such file: its own restore fails on I_PCM streams)
    with open(dst, "wb") as f:
        f.write(main)
    for t, b in tags.items():
        with open("%s.%d" % (dst, t), "wb") as f:
            f.write(b)
    total = len(main) + sum(len(b) for b in tags.values())
    print("%s: %d bytes -> %d bytes (%.4f), %d pictures" % (src, len(data), total, total / max(1, len(data)), len(frames)))
    if yuv:
        sess = lh.ReconSession([frames])
        sess.run()
        sess.synchronize()
        with open(yuv, "wb") as f:
            for i, fr in enumerate(frames):
                pl = sess.picture(0, i)
                for p in range(3):
                    s = 1 if p else 0
                    f.write(np.ascontiguousarray(pl[p][fr.crop_y >> s:(fr.crop_y + fr.crop_h) >> s, fr.crop_x >> s:(fr.crop_x + fr.crop_w) >> s]).tobytes())


def restore(src, dst):
    import losslessh264_amd as lh
    main = open(src, "rb").read()
    tags = {}
    for p in glob.glob(glob.escape(src) + ".*"):
        ext = p[len(src) + 1:]
        if ext.isdigit():
            tags[int(ext)] = open(p, "rb").read()
    out = lh.restore(main, tags)
    with open(dst, "wb") as f:
        f.write(out)
    print("%s (+%d tagged streams) -> %s: %d bytes" % (src, len(tags), dst, len(out)))


def main(argv):
    if len(argv) < 3:
        print(__doc__)
        return 2
    if argv[1].endswith(".lhp"):
        restore_single(argv[1], argv[2])
    elif argv[2].endswith(".lhp"):
        compress_single(argv[1], argv[2])
    elif ".pip" in os.path.basename(argv[1]):      # as the reference decides (h264dec.cpp:167-173)
        restore(argv[1], argv[2])
    else:
        compress(argv[1], argv[2], argv[3] if len(argv) > 3 else None)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
