#!/usr/bin/env python3
"""phase shares of the resolve kernel (needs LH264_SO = a -DLH264_CODER_DEBUG build): N replicas of the bench stream"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import losslessh264_amd as lh
from losslessh264_amd import _lib as L
streams = int(sys.argv[1]) if len(sys.argv) > 1 else 512
data = open(os.path.join(ROOT, "tests", "golden", "streams", "BA_MW_D.264"), "rb").read()
frames, err = lh.parse_stream(data)
ctx = lh.CtxSession([frames], replicate=streams)
coder = lh.CoderSession(ctx)
ctx.run(); coder.run(); ctx.synchronize()
lens = coder.d_len.cpu().numpy().astype(np.int64).reshape(streams, L.N_TAG_SLOTS + 1)
names = ["pre-work + land", "ticket wait", "serial section", "probability + store", "lookup two steps ahead", "loop tail + word wait"]
lens[:, 40] = lens[:, 34]
tot = lens[:, 35:41].sum()
n_rounds = 683343 // 64 + 1
print("streams %d: wave k-cycles per stream %.0f; per round (8 waves -> /8 per wave): " % (streams, tot / streams) +
      ", ".join("%s %.0f cyc (%.0f%%)" % (names[i], 1024.0 * lens[:, 35 + i].sum() / streams / n_rounds, 100.0 * lens[:, 35 + i].sum() / tot) for i in range(6)))
