#!/usr/bin/env python3
"""one launch of the context-index + coder kernels on a config's stream (target for rocprofv3 --pmc / --kernel-trace):
tools/coder_cfg_once.py NAME STREAMS [FRAMES]; prints decisions per stream"""
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
dw, le = C.c_ulonglong(), C.c_ulonglong()
L.lib().lh264_code_last_totals(C.byref(dw), C.byref(le))
print("decision words %d (%.0f per stream), list entries %d, macroblocks %d" % (dw.value, dw.value / streams, le.value, ctx.n_mbs_total))
