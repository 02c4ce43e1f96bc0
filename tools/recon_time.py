#!/usr/bin/env python3
"""chain-kernel time on BA_MW_D.264 x 512 (100 frames), HIP events"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_io
import losslessh264_amd as lh
ba = golden_io.load("bench_BA_MW_D.264")
s = lh.ReconSession([ba], replicate=512, share_records=False)
s.time_kernel(2)
print("recon %.3f ms" % s.time_kernel(8), flush=True)
