#!/usr/bin/env python3
"""Generate the committed golden fixtures from the REFERENCE itself (run in the build container only).

For each selected stream this runs oracle/_ref/ref_dump (our shim around the reference decoder built
by oracle/Makefile from /root/reference) and stores, per decoded frame, what the reference handed to
its reconstruct path (macroblock records, slice records, coefficients) and what came out (CRC32 of the
pre-deblock and final planes; full planes for a few frames).  The fixtures are data only.

    python tests/golden/make_golden.py            # regenerates tests/golden/*.npz

Fixture layout (np.savez_compressed): see tests/golden_io.py for the reader.
"""
import os
import subprocess
import sys
import tempfile
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from refdump import read_dump  # noqa: E402

REF = "/root/reference"
# (stream, max_frames)   -- kept small: the fixtures are committed
STREAMS = [
    ("res/SVA_BA2_D.264", 17),            # QCIF I/P CAVLC baseline
    ("res/BA_MW_D.264", 24),              # config #2 seed stream (I + P, multi-partition)
    ("res/SVA_BA1_B.264", 6),             # QCIF, intra-heavy
    ("res/MR1_BT_A.h264", 8),             # multiple reference frames
    ("res/CVPCMNL1_SVA_C.264", 2),        # I_PCM macroblocks (CIF)
    ("res/test_vd_1d.264", 3),            # 320x192 multi-slice
    ("roundtriptest/tibby8x8cavlc.264", 6),   # High profile, 8x8 transform + I8x8
    ("roundtriptest/tibbycabac.264", 4),  # CABAC-parsed records
    ("res/CI1_FT_B.264", 4),              # CIF CABAC interlace-free main profile
    ("res/test_qcif_cabac.264", 6),       # QCIF CABAC I/P
    ("res/CI_MW_D.264", 6),               # QCIF CABAC, multiple partitions
]


# full-length record sets for bench.py (config #2 seed stream); CRCs only, no planes
BENCH_STREAMS = [
    ("res/BA_MW_D.264", 100),
]


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def main():
    dump_bin = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
    if not os.path.exists(dump_bin):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    tmp = tempfile.mkdtemp(prefix="lh264_golden_")
    # one process per stream: the reference keeps its context-model history (FreqImage) in function statics, so a
    # second stream in the same process would inherit the first one's PAST (the reference CLI handles one stream
    # per process).  Each process builds the reference's 8.78 GiB model: run at most 3 at a time.
    todo = sorted(set(os.path.join(REF, s) for s, _ in STREAMS + BENCH_STREAMS))
    running = []
    while todo or running:
        while todo and len(running) < 3:
            running.append(subprocess.Popen([dump_bin, tmp, todo.pop()], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL))
        running[0].wait()
        assert running[0].returncode == 0
        running.pop(0)
    for stream, nmax in STREAMS + BENCH_STREAMS:
        bench = (stream, nmax) in BENCH_STREAMS
        base = os.path.basename(stream)
        frames = read_dump(os.path.join(tmp, base + ".dmp"), nmax)
        out = {"n_frames": np.int32(len(frames))}
        hdr, crcs = [], []
        for i, f in enumerate(frames):
            hdr.append([f.id, f.mb_w, f.mb_h, len(f.slices), f.crop_w, f.crop_h, f.has_final, len(f.ref_ids)]
                       + (f.ref_ids + [-1] * 16)[:16] + [f.frame_num])
            out["mbs_%d" % i] = f.mbs.view(np.uint8).reshape(-1, 128)
            out["slices_%d" % i] = f.slices.view(np.uint8).reshape(-1, 232)
            # coefficients are sparse: store (flat index, value) pairs
            nz = np.flatnonzero(f.coeffs)
            out["cidx_%d" % i] = nz.astype(np.uint32)
            out["cval_%d" % i] = f.coeffs.reshape(-1)[nz]
            out["covered_%d" % i] = f.covered
            # context-model observations (row a8): raw levels, the neighbours' nonzero counts the reference's model saw,
            # and every (kind, value, prior index) it coded for this frame
            lz = np.flatnonzero(f.levels)
            out["lidx_%d" % i] = lz.astype(np.uint32)
            out["lval_%d" % i] = f.levels.reshape(-1)[lz]
            out["nei_%d" % i] = f.nei
            out["symcnt_%d" % i] = np.array([len(s) for s in f.syms], dtype=np.uint16)
            out["syms_%d" % i] = np.concatenate(f.syms).view(np.uint8) if len(f.syms) else np.zeros(0, np.uint8)
            crcs.append([crc(p) for p in f.pre] + ([crc(p) for p in f.fin] if f.has_final else [0, 0, 0]))
            if not bench and i in (0, len(frames) - 1):       # full planes for the first and last frame (debugging aid)
                for p in range(3):
                    out["pre_%d_%d" % (i, p)] = f.pre[p]
                    if f.has_final:
                        out["fin_%d_%d" % (i, p)] = f.fin[p]
        out["hdr"] = np.array(hdr, dtype=np.int32)
        out["crc"] = np.array(crcs, dtype=np.uint32)
        out["stream_bytes"] = np.int64(os.path.getsize(os.path.join(REF, stream)))
        path = os.path.join(HERE, ("bench_" if bench else "") + base + ".npz")
        np.savez_compressed(path, **out)
        print("%s: %d frames -> %s (%d KB)" % (stream, len(frames), path, os.path.getsize(path) // 1024))


if __name__ == "__main__":
    main()
