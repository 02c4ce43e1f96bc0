#!/usr/bin/env python3
"""coded bytes per tag of stream 0 (how long the lists of the range stage are): tools/coder_taglens.py NAME STREAMS [FRAMES]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import losslessh264_amd as lh
name, streams = sys.argv[1], int(sys.argv[2])
frames, err = lh.parse_stream(open(os.path.join(ROOT, "tests", "golden", "streams", name), "rb").read())
if len(sys.argv) > 3:
    frames = frames[:int(sys.argv[3])]
ctx = lh.CtxSession([frames], replicate=streams)
coder = lh.CoderSession(ctx, out_cap=1 << 21)
ctx.run(); coder.run(); ctx.synchronize()
t = coder.tags(0)
print(" ".join("%d:%d" % (k, len(v)) for k, v in sorted(t.items()) if len(v)))
