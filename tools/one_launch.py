#!/usr/bin/env python3
"""A few launches of the chain kernel (and, for the BA_MW_D workloads, the context and coder kernels) on a chosen workload: the
target of the rocprofv3 --pmc runs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_io, synth
import losslessh264_amd as lh
which = sys.argv[1] if len(sys.argv) > 1 else "ba"
streams = int(sys.argv[2]) if len(sys.argv) > 2 else 512
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
if which in ("ba", "ba20"):
    frames, _ = lh.parse_stream(open(os.path.join(ROOT, "tests", "golden", "streams", "BA_MW_D.264"), "rb").read())
    if which == "ba20":
        frames = frames[:20]
elif which == "intra":
    frames = synth.make_stream(1, 11, 9, 4, p_frames=False)
elif which == "720p":
    frames = synth.make_stream(2, 80, 45, 2, p_frames=True)
s = lh.ReconSession([frames], replicate=streams, share_records=False)
c = lh.CtxSession([frames], replicate=streams) if which.startswith("ba") else None
k = lh.CoderSession(c) if c is not None else None
for _ in range(reps):
    s.run()
    if c is not None:
        c.run()
        k.run()
s.synchronize()
print("mbs per launch", s.n_mbs_total)
