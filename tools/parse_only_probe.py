#!/usr/bin/env python3
"""the host side of lh264_compress_batch alone (LH264_COMPRESS_PARSE_ONLY=1 skips the device stage; LH264_TRACE_COMPRESS=1 prints
the time of every wave of parsed streams)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("LH264_COMPRESS_PARSE_ONLY", "1")
import torch  # noqa: F401  (HIP runtime first)
import losslessh264_amd as lh
data = open(os.path.join(ROOT, "tests", "golden", "streams", "BA_MW_D.264"), "rb").read()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
lh.compress_batch([data] * 4, 16)
t0 = time.perf_counter()
lh.compress_batch([data] * n, 16)
dt = time.perf_counter() - t0
print("%d streams through the host side of the pipeline: %.2f s (%.1f MB/s)" % (n, dt, n * len(data) / dt / 1e6))
