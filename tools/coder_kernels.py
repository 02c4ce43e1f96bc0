#!/usr/bin/env python3
"""per-kernel device time of the coder stage on the bench workload (BA_MW_D x N streams), from hipEvents around whole runs and
from rocprofv3 when run under it: prints total ms per coder.run()"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import losslessh264_amd as lh
data = open(os.path.join(ROOT, "tests", "golden", "streams", "BA_MW_D.264"), "rb").read()
frames, err = lh.parse_stream(data)
streams = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = lh.CtxSession([frames], replicate=streams)
coder = lh.CoderSession(ctx)
ctx.run(); coder.run(); ctx.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    coder.run()
ctx.synchronize()
dt = (time.perf_counter() - t0) / 3
try:
    n = sum(len(v) for v in coder.tags(streams - 1).values())
except Exception as e:
    n = repr(e)
print("so=%s streams=%d coder %.2f ms coded=%s" % (os.environ.get("LH264_SO", "default"), streams, dt * 1e3, n), flush=True)
