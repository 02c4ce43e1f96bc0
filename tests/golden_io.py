"""Reader for the committed golden fixtures (tests/golden/*.npz, written by make_golden.py)."""
import glob
import os
import zlib

import numpy as np

from refdump import MB_DTYPE, SLICE_DTYPE, SYM_DTYPE

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class GFrame:
    pass


def list_fixtures():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))
                  if not os.path.basename(p).startswith(("bench_", "pip_", "cli_")))


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    frames = []
    load.stream_bytes = int(z["stream_bytes"]) if "stream_bytes" in z else 0
    for i in range(int(z["n_frames"])):
        h = z["hdr"][i]
        f = GFrame()
        f.id, f.mb_w, f.mb_h, nsl, f.crop_w, f.crop_h, f.has_final, nref = [int(x) for x in h[:8]]
        f.ref_ids = [int(x) for x in h[8:8 + nref]]
        f.mbs = z["mbs_%d" % i].reshape(-1).view(MB_DTYPE).copy()
        f.slices = z["slices_%d" % i].reshape(-1).view(SLICE_DTYPE).copy()
        coeffs = np.zeros(f.mb_w * f.mb_h * 384, dtype=np.int16)
        coeffs[z["cidx_%d" % i]] = z["cval_%d" % i]
        f.coeffs = coeffs.reshape(-1, 384)
        f.covered = z["covered_%d" % i]
        f.frame_num = int(h[24]) if len(h) > 24 else 0
        if ("lidx_%d" % i) in z:
            levels = np.zeros(f.mb_w * f.mb_h * 384, dtype=np.int16)
            levels[z["lidx_%d" % i]] = z["lval_%d" % i]
            f.levels = levels.reshape(-1, 384)
            f.nei = z["nei_%d" % i]
            cnt = z["symcnt_%d" % i].astype(np.int64)
            allsyms = z["syms_%d" % i].view(SYM_DTYPE)
            offs = np.concatenate([[0], np.cumsum(cnt)])
            f.syms = [allsyms[offs[k]:offs[k + 1]] for k in range(len(cnt))]
        f.crc_pre = [int(x) for x in z["crc"][i][:3]]
        f.crc_fin = [int(x) for x in z["crc"][i][3:]]
        f.pre = [z["pre_%d_%d" % (i, p)] for p in range(3)] if ("pre_%d_0" % i) in z else None
        f.fin = [z["fin_%d_%d" % (i, p)] for p in range(3)] if ("fin_%d_0" % i) in z else None
        frames.append(f)
    return frames


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF
