#!/usr/bin/env python3
"""Phase shares from the diagnostic stamp build (build/variants/libstamp.so)."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["LH264_SO"] = os.path.join(ROOT, "build", "variants", "libstamp.so")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_io, synth
import losslessh264_amd as lh
from losslessh264_amd import _lib
names = ["0 stage records", "1 neighbours", "2 residual", "3 prediction", "4 publish", "5 deblock loads", "6 deblock filter", "7 window write", "8 wait for row above", "9 row end (stores+pad)", "10 frame switch/prefix"]
L = _lib.lib()
L.lh264_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 16)()
def run(name, frames, streams):
    s = lh.ReconSession([frames], replicate=streams, share_records=False)
    L.lh264_debug_read_stamps(buf, 1)
    s.run(); s.synchronize()
    L.lh264_debug_read_stamps(buf, 1)
    tot = sum(buf[i] for i in range(11))
    print("== %s streams=%d  cycles/MB(sum over phases)=%.0f" % (name, streams, tot / s.n_mbs_total))
    for i in range(11):
        print("   %-24s %6.1f%%  %8.0f cyc/MB" % (names[i], 100.0 * buf[i] / tot, buf[i] / s.n_mbs_total))
if len(sys.argv) > 1:            # tools/stamp_probe.py STREAM N [FRAMES]
    fr, err = lh.parse_stream(open(os.path.join(ROOT, "tests", "golden", "streams", sys.argv[1]), "rb").read())
    if len(sys.argv) > 3:
        fr = fr[:int(sys.argv[3])]
    run(sys.argv[1], fr, int(sys.argv[2]))
else:
    ba = golden_io.load("bench_BA_MW_D.264")[:20]
    run("BA_MW_D", ba, 1)
    run("BA_MW_D", ba, 512)
    run("intra", synth.make_stream(1, 11, 9, 4, p_frames=False), 512)
