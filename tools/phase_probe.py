#!/usr/bin/env python3
"""Ablation probe: launch time of the chain kernel with phases switched off (debug flags) and on different content."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import golden_io, synth
import losslessh264_amd as lh

def run(name, frames, streams, flags):
    s = lh.ReconSession([frames], replicate=streams, share_records=False, flags=flags)
    ms = s.time_kernel(3)
    n = s.n_mbs_total
    print("%-40s flags=%d streams=%d mbs=%d  %.2f ms  %.1f M MB/s  %.2f us/frame/stream" % (
        name, flags, streams, n, ms, n / ms / 1e3, ms * 1e3 / len(frames)), flush=True)

streams = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ba = golden_io.load("bench_BA_MW_D.264")
for fl in (0, 2, 1, 3):
    run("BA_MW_D 100fr", ba, streams, fl)
if os.environ.get("PROBE_SHORT"):
    sys.exit(0)
run("BA_MW_D first frame only (I)", ba[:1], streams, 0)
run("BA_MW_D first frame only (I) nodeblock", ba[:1], streams, 3)
i16 = synth.make_stream(1, 11, 9, 4, p_frames=False)
run("synthetic intra QCIF", i16, streams, 0)
run("synthetic intra QCIF nodeblock", i16, streams, 3)
big = synth.make_stream(2, 80, 45, 2, p_frames=True)
run("synthetic 720p I+P", big, 64, 0)
run("synthetic 720p I+P nodeblock", big, 64, 3)
