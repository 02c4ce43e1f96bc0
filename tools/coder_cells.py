#!/usr/bin/env python3
"""how many prior cells a stream touches (occupancy of its hash table after a run) and, with a -DLH264_CODER_DEBUG build, how
often the resolve kernel's LDS cache missed / was flushed"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import losslessh264_amd as lh
for name in sys.argv[1:] or ["BA_MW_D.264"]:
    data = open(os.path.join(ROOT, "tests", "golden", "streams", name), "rb").read()
    frames, err = lh.parse_stream(data)
    ctx = lh.CtxSession([frames], replicate=2)
    coder = lh.CoderSession(ctx, hash_cap=1 << 18, out_cap=1 << 20)
    ctx.run(); coder.run(); ctx.synchronize()
    keys = np.zeros(1)
    cells = coder.d_cells[:coder.hash_cap * 16].cpu().numpy().reshape(-1, 16)
    used = keys != 0
    print(name, "frames", len(frames), "mbs", sum(f.mb_w * f.mb_h for f in frames), "distinct cells", int(used.sum()),
          "distinct DynProbs (nonzero state)", int((cells != 0).sum()), flush=True)
    lens = coder.d_len[:41].cpu().numpy().astype(np.int64)
    names = ["pre-work + land", "ticket wait", "serial section", "probability + store", "lookup two steps ahead"]
    tot = max(1, int(lens[35:40].sum()))
    print("   debug build only (wave time, k-cycles):", ", ".join("%s %d (%.0f%%)" % (names[i], lens[35 + i], 100.0 * lens[35 + i] / tot) for i in range(5)), "status", int(lens[40]), flush=True)
