#!/usr/bin/env python3
"""decisions per partition of one stream (how even the resolve kernel's waves are loaded): tools/coder_parts.py NAME STREAMS [FRAMES]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import losslessh264_amd as lh
from losslessh264_amd import _lib as L
name, streams = sys.argv[1], int(sys.argv[2])
frames, err = lh.parse_stream(open(os.path.join(ROOT, "tests", "golden", "streams", name), "rb").read())
if len(sys.argv) > 3:
    frames = frames[:int(sys.argv[3])]
ctx = lh.CtxSession([frames], replicate=streams)
coder = lh.CoderSession(ctx, out_cap=1 << 21)
ctx.run(); coder.run(); ctx.synchronize()
out = (C.c_ulonglong * 128)()
f = L.lib().lh264_debug_coder_parts
f.restype = C.c_int
P = f(0, out, 128)
v = [out[i] for i in range(max(P, 0))]
tot = sum(v)
print("partitions", P, "decisions", tot)
print(" ".join("%.1f%%" % (100.0 * x / tot) for x in v))
print("largest / mean = %.2f" % (max(v) * P / tot))
