#!/usr/bin/env python3
"""chain-kernel time on BA_MW_D.264 x 512 (100 frames), HIP events; argv[1] = timed launches (default 8); LH264_FLAGS = job flags"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_io
import losslessh264_amd as lh
ba = golden_io.load("bench_BA_MW_D.264")
fl = int(os.environ.get("LH264_FLAGS", "0"))
s = lh.ReconSession([ba], replicate=512, share_records=False, flags=fl)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
if n > 1:
    s.time_kernel(2)
print("recon flags=%d %.3f ms" % (fl, s.time_kernel(n)), flush=True)
