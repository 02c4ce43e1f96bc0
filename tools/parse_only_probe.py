import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import losslessh264_amd as lh
data = open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests/golden/streams/BA_MW_D.264"), "rb").read()
lh.compress_batch([data] * 4, 16)
t0 = time.perf_counter(); lh.compress_batch([data] * 2048, 16); print("2048 streams parse-only pipeline: %.2f s" % (time.perf_counter() - t0))
