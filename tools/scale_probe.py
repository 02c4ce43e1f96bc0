#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_io
import losslessh264_amd as lh
ba = golden_io.load("bench_BA_MW_D.264")[:20]
for streams in (1, 32, 128, 256, 512, 1024, 2048, 4096):
    s = lh.ReconSession([ba], replicate=streams, share_records=False)
    ms = s.time_kernel(3)
    print("streams=%5d  %.2f ms  %.1f M MB/s  %.1f us/frame" % (streams, ms, s.n_mbs_total / ms / 1e3, ms * 1e3 / len(ba)), flush=True)
    del s
