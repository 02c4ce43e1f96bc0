#!/usr/bin/env python3
"""Generate the fixtures of the recompressor's coder: tests/golden/pip_<stream>.npz.

For each stream oracle/_ref/ref_dump (our shim around the unmodified reference, one process per stream) yields, per
macroblock, the DecodedMacroblock fields the reference's emit code read (captured from its FreqImage), the quantised
levels, per slice the pad bits it sent to the pad-byte tag, and - at the end - the byte string of every tagged
arithmetic-coded stream exactly as the reference's console application would write them to <out>.pip.<tag>.
tests/test_oracle_coder.py replays the records through oracle/oracle_coder.c and requires identical bytes.
Fixtures are data only; nothing of the reference's source is stored.
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import refdump  # noqa: E402

REF = "/root/reference"
# (stream, output pictures fed to the reference; 0 = the whole stream)
STREAMS = [
    ("res/SVA_BA2_D.264", 0),             # QCIF I/P, single slice
    ("res/SVA_BA1_B.264", 0),             # QCIF, intra-heavy
    ("res/Static.264", 0),                # long skip runs
    ("res/MR1_BT_A.h264", 12),            # several reference pictures: raw reference-index bits
    ("res/test_vd_1d.264", 0),            # several slices per picture
    ("res/BA_MW_D.264", 30),              # config #2 seed stream
    ("roundtriptest/tibby8x8cavlc.264", 5),   # 8x8 transform, I8x8
    ("res/CI_MW_D.264", 12),              # constrained_intra_pred: the other variant of the intra-mode cache
    ("res/test_qcif_cabac.264", 8),       # CABAC I/P (config #5 family): skip flags instead of skip runs
    ("roundtriptest/tibbycabac.264", 4),  # CABAC with the 8x8 transform
]


def main():
    dump_bin = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
    tmp = tempfile.mkdtemp(prefix="lh264_pip_")
    todo = list(STREAMS)
    running = []
    while todo or running:
        while todo and len(running) < 3:            # each reference process allocates 8.8 GiB of prior tables
            stream, nmax = todo.pop()
            env = dict(os.environ)
            if nmax:
                env["REF_DUMP_MAX_FRAMES"] = str(nmax)
            running.append(subprocess.Popen([dump_bin, tmp, os.path.join(REF, stream)], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL))
        running[0].wait()
        assert running[0].returncode == 0
        running.pop(0)
    for stream, nmax in STREAMS:
        base = os.path.basename(stream)
        frames = refdump.read_dump(os.path.join(tmp, base + ".dmp"))
        tags = refdump.read_dump.tags
        out = {"n_frames": np.int32(len(frames))}
        out["hdr"] = np.array([[f.mb_w, f.mb_h, f.frame_num, len(f.slices)] for f in frames], dtype=np.int32)
        out["mb_types"] = np.concatenate([f.mbs["mb_type"] for f in frames]).astype(np.uint16)
        lv = np.concatenate([f.levels.reshape(-1) for f in frames])
        nz = np.flatnonzero(lv)
        out["lidx"] = nz.astype(np.uint32)
        out["lval"] = lv[nz]
        out["rtd"] = np.concatenate([f.rtd for f in frames]).view(np.uint8).reshape(-1, refdump.RTD_DTYPE.itemsize)
        # per slice: first_mb, n_mbs, slice_type, pad bit count, pad bits, PPS transform_8x8_mode_flag (bit 0)
        out["slices"] = np.array([[int(s["first_mb"]), int(s["n_mbs"]), int(s["slice_type"]), int(e[0]), int(e[1]), int(e[2])]
                                  for f in frames for s, e in zip(f.slices, f.slice_extra)], dtype=np.int32)
        # bits 1 and 2 of the last column: the PPS's constrained_intra_pred_flag and entropy_coding_mode_flag (facts of the
        # bitstream, read with the product's header parser; the dump does not carry them)
        sys.path.insert(0, ROOT)
        import losslessh264_amd as lh
        pf, _ = lh.parse_stream(open(os.path.join(REF, stream), "rb").read())
        fl = np.array([int(x) for f in pf[:len(frames)] for x in f.slice_syn[:, 3]], dtype=np.int32)
        assert len(fl) == len(out["slices"])
        out["slices"][:, 5] |= (fl & 2) | ((fl & 1) << 2)       # bit 1 constrained_intra_pred_flag, bit 2 entropy_coding_mode_flag
        for t, b in tags.items():
            if t != 0x7fffffff:
                out["tag_%d" % t] = np.frombuffer(b, dtype=np.uint8)
        path = os.path.join(HERE, "pip_" + base + ".npz")
        np.savez_compressed(path, **out)
        print("%s: %d frames, %d tags, %d coded bytes -> %s (%d KB)" % (stream, len(frames), len(tags) - 1,
              sum(len(b) for t, b in tags.items() if t != 0x7fffffff), path, os.path.getsize(path) // 1024))


if __name__ == "__main__":
    main()
