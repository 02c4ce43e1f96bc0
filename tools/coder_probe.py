#!/usr/bin/env python3
"""time of the device coder stage on the bench workload (BA_MW_D x N streams), and its output size"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import losslessh264_amd as lh
data = open(os.path.join(ROOT, "tests", "golden", "streams", "BA_MW_D.264"), "rb").read()
frames, err = lh.parse_stream(data)
assert err == ""
for streams in [int(a) for a in sys.argv[1:]] or [64, 512]:
    ctx = lh.CtxSession([frames], replicate=streams)
    coder = lh.CoderSession(ctx, hash_cap=1 << 16, out_cap=1 << 16)
    ctx.run(); coder.run(); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        coder.run()
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / 3
    tags = coder.tags(streams - 1)
    nsym = int(ctx.d_nsyms.sum().item()) // streams + len(sum([list(f.syn_syms) for f in frames[:1]], [])) * 0
    print("streams=%d  coder %.1f ms  (%.1f MB/s of .264)  coded bytes/stream %d of %d (%.4f)  ctx symbols/stream %d" % (
        streams, dt * 1e3, streams * len(data) / dt / 1e6, sum(len(v) for v in tags.values()), len(data), sum(len(v) for v in tags.values()) / len(data), nsym), flush=True)
    del coder, ctx
