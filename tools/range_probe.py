#!/usr/bin/env python3
"""diagnostic (build/variants/librangeprobe.so, -DLH264_RANGE_PROBE): distinct range states left after a lookback of 256 / 1024 / 4096
decisions in front of every coarse chunk, for one stream of a config.  LH264_SO=build/variants/librangeprobe.so python3 tools/range_probe.py NAME [frames]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import losslessh264_amd as lh
from losslessh264_amd import _lib as L
name = sys.argv[1]
frames, err = lh.parse_stream(open(os.path.join(ROOT, "tests", "golden", "streams", name), "rb").read())
if len(sys.argv) > 2:
    frames = frames[:int(sys.argv[2])]
ctx = lh.CtxSession([frames], replicate=2)
coder = lh.CoderSession(ctx, out_cap=1 << 22)
ctx.run(); coder.run(); ctx.synchronize()
out = np.zeros(1 << 20, dtype=np.uint32)
f = L.lib().lh264_debug_coder_seeds
f.restype = C.c_longlong
n = f(out.ctypes.data_as(C.c_void_p), C.c_longlong(len(out)))
s = out[:n]
s = s[s != 0]                 # (chunk 0 of a list has no lookback)
for k, w in enumerate((256, 1024, 4096)):
    d = (s >> (8 * k)) & 0xff
    print(name, "W=%d: chunks %d  distinct states: mean %.1f  median %d  p90 %d  max %d  ==1: %.1f%%  <=4: %.1f%%  <=8: %.1f%%" % (
        w, len(d), d.mean(), np.median(d), np.percentile(d, 90), d.max(), 100 * (d == 1).mean(), 100 * (d <= 4).mean(), 100 * (d <= 8).mean()))
