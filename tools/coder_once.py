#!/usr/bin/env python3
"""one launch of the coder kernel on the bench workload (target for rocprofv3 --pmc)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import losslessh264_amd as lh
frames, err = lh.parse_stream(open(os.path.join(ROOT, "tests", "golden", "streams", "BA_MW_D.264"), "rb").read())
streams = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = lh.CtxSession([frames], replicate=streams)
coder = lh.CoderSession(ctx)
ctx.run(); coder.run(); ctx.synchronize()
print("done")
