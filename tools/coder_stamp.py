#!/usr/bin/env python3
"""phase breakdown of the device coder (diagnostic build -DLH264_CODER_STAMP): s_memtime per phase, summed per wave"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import losslessh264_amd as lh
from losslessh264_amd import _lib as L
data = open(os.path.join(ROOT, "tests", "golden", "streams", "BA_MW_D.264"), "rb").read()
frames, err = lh.parse_stream(data)
streams = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = lh.CtxSession([frames], replicate=streams)
coder = lh.CoderSession(ctx)
ctx.run(); coder.run(); ctx.synchronize()
out = coder.d_out.cpu().numpy().reshape(streams, L.N_TAG_SLOTS, coder.out_cap)
names = ["fill", "binarise", "owner+rank", "probe+fetch", "serial path", "serial writeback", "tail", "touch+raw bits", "cell rows", "tag scans + writeback", "scatter + hand-off", "wait for the coding wave"]
acc = out[:, 39, :128].copy().view(np.uint64).astype(np.float64)     # [streams][16]
acc[:, 6] *= 1.0
tot = (acc[:, :6].sum(axis=1) + acc[:, 7:12].sum(axis=1)).mean()
for i, n in enumerate(names):
    print("%-16s %6.1f %%" % (n, 100 * acc[:, i].mean() / tot))
nb = acc[:, 12].mean()
print("parallel batches/stream %.0f  rounds/batch %.2f  max decisions of a symbol/batch %.1f  cell-row loop trips/batch %.1f  longest per-row chain/batch %.1f" % (nb, acc[:, 13].mean() / nb, acc[:, 14].mean() / nb, acc[:, 15].mean() / nb, acc[:, 6].mean() / nb))
