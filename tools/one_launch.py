#!/usr/bin/env python3
"""One launch of the chain kernel on a chosen workload (target for rocprofv3 --pmc runs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_io, synth
import losslessh264_amd as lh
which = sys.argv[1] if len(sys.argv) > 1 else "ba"
streams = int(sys.argv[2]) if len(sys.argv) > 2 else 512
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
if which == "ba":
    frames = golden_io.load("bench_BA_MW_D.264")
elif which == "ba20":
    frames = golden_io.load("bench_BA_MW_D.264")[:20]
elif which == "intra":
    frames = synth.make_stream(1, 11, 9, 4, p_frames=False)
elif which == "720p":
    frames = synth.make_stream(2, 80, 45, 2, p_frames=True)
s = lh.ReconSession([frames], replicate=streams, share_records=False)
c = lh.CtxSession([frames], replicate=streams) if which.startswith("ba") else None
for _ in range(reps):
    s.run()
    if c is not None:
        c.run()
s.synchronize()
print("mbs per launch", s.n_mbs_total)
