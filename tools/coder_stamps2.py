#!/usr/bin/env python3
"""diagnostic (build/variants/libcoderdbg.so, -DLH264_CODER_DEBUG): where the resolve kernel's wave time goes.
LH264_SO=build/variants/libcoderdbg.so python3 tools/coder_stamps2.py NAME STREAMS [FRAMES]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import losslessh264_amd as lh
from losslessh264_amd import _lib as L
name, streams = sys.argv[1], int(sys.argv[2])
frames, err = lh.parse_stream(open(os.path.join(ROOT, "tests", "golden", "streams", name), "rb").read())
if len(sys.argv) > 3:
    frames = frames[:int(sys.argv[3])]
ctx = lh.CtxSession([frames], replicate=streams)
coder = lh.CoderSession(ctx, out_cap=1 << 21)
out = np.zeros(16, dtype=np.uint64)
ctx.run(); coder.run(); ctx.synchronize()
L.lib().lh264_debug_read_rs_stamps(out.ctypes.data_as(C.c_void_p), 1)
coder.run(); ctx.synchronize()
L.lib().lh264_debug_read_rs_stamps(out.ctypes.data_as(C.c_void_p), 1)
names = ["wait words + ring read", "lookup (round + 2)", "match", "land", "counters / probability", "flush", "generate + request", "rounds", "flushes", "probe iterations (max over lanes, summed)"]
tot = float(out[:7].sum())
for i, n in enumerate(names):
    if i < 7:
        print("%-26s %6.1f %%  %8.0f cycles per round" % (n, 100.0 * out[i] / tot, out[i] / max(1, out[7])))
    else:
        print("%-26s %d" % (n, out[i]))
