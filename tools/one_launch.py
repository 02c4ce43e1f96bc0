#!/usr/bin/env python3
"""A few launches of every kernel of the device path (reconstruct, context index, coder) on one of bench.py's workloads: the target of
the rocprofv3 --pmc passes (counters are collected over a plain run, never beside tracing).
  tools/one_launch.py --config N [--streams S] [--reps R]        bench.py's --config N batch
  tools/one_launch.py ba|ba20|intra|720p STREAMS REPS            (older form: BA_MW_D / synthetic records, reconstruct only for the latter two)
Prints the macroblocks per launch and the library's build id."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import losslessh264_amd as lh
from losslessh264_amd import _lib as L

if len(sys.argv) > 1 and not sys.argv[1].startswith("--"):
    import synth
    which = sys.argv[1]
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
    order, rep = [frames], streams
    coded = which.startswith("ba")
else:
    import bench
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=1)
    ap.add_argument("--streams", type=int, default=0)
    ap.add_argument("--reps", type=int, default=2)
    a = ap.parse_args()
    cfg = bench.CONFIGS[a.config]
    order = []
    for name in cfg["streams"]:
        frames, err, _ = lh.parse_file(open(os.path.join(ROOT, "tests", "golden", "streams", name), "rb").read())
        assert err == "", err
        order.append(frames[:cfg["frames"]] if cfg["frames"] else frames)
    n = a.streams or cfg["n"]
    rep, reps, coded = max(1, n // len(order)), a.reps, True
s = lh.ReconSession(order, replicate=rep, share_records=False)
c = lh.CtxSession(order, replicate=rep) if coded else None
out_cap = 1 << 16
while coded and out_cap < 0.6 * max(sum(f.mb_w * f.mb_h for f in fr) for fr in order) * 40:      # generous: ~40 bytes per macroblock at most here
    out_cap <<= 1
k = lh.CoderSession(c, out_cap=out_cap) if coded else None
for _ in range(reps):
    s.run()
    if coded:
        c.run()
        k.run()
s.synchronize()
print("mbs per launch", s.n_mbs_total, "streams", s.n_chains, "build", L.lib().lh264_build_id().decode())
