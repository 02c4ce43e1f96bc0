#!/usr/bin/env python3
"""the whole device compress (rows a1-a10) run back to back on one HIP stream, and with the reconstruct kernel on a second stream
beside the context-index + coder stages (they read different buffers): ms per step for N replicas of the bench stream"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import losslessh264_amd as lh
streams = int(sys.argv[1]) if len(sys.argv) > 1 else 512
data = open(os.path.join(ROOT, "tests", "golden", "streams", "BA_MW_D.264"), "rb").read()
frames, err, main = lh.parse_file(data)
sess = lh.ReconSession([frames], replicate=streams, share_records=False)
ctx = lh.CtxSession([frames], replicate=streams)
coder = lh.CoderSession(ctx)
dev = sess.dev
s2 = torch.cuda.Stream(dev)
def serial():
    sess.run(); ctx.run(); coder.run()
def overlapped():
    s2.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s2):
        sess.run()
    ctx.run(); coder.run()
    torch.cuda.current_stream(dev).wait_stream(s2)
def recon_second():
    cur = torch.cuda.current_stream(dev)
    ctx.run()
    s2.wait_stream(cur)
    with torch.cuda.stream(s2):
        sess.run()
    coder.run()
    cur.wait_stream(s2)
def recon_last():
    cur = torch.cuda.current_stream(dev)
    s2.wait_stream(cur)          # (the previous step's end)
    ctx.run(); coder.run()
    with torch.cuda.stream(s2):
        sess.run()
    cur.wait_stream(s2)
def split_coder():
    # the reconstruct kernel beside the second half of the coder only (lh264_code_binarise_chains / lh264_code_finish_chains)
    cur = torch.cuda.current_stream(dev)
    ctx.run(); coder.binarise()
    s2.wait_stream(cur)
    with torch.cuda.stream(s2):
        sess.run()
    coder.finish()
    cur.wait_stream(s2)
halves = [lh.ReconSession([frames], replicate=streams // 2, share_records=False) for _ in range(2)] if os.environ.get("PROBE_HALVES") else None
def recon_halves():
    # the reconstruct batch as two launches of half the streams each (one workgroup per CU at a time): slower by itself, but the
    # coder's kernels - the resolve kernel too - fit beside it
    cur = torch.cuda.current_stream(dev)
    s2.wait_stream(cur)
    with torch.cuda.stream(s2):
        halves[0].run(); halves[1].run()
    ctx.run(); coder.run()
    cur.wait_stream(s2)
def recon_halves_last():
    cur = torch.cuda.current_stream(dev)
    s2.wait_stream(cur)
    ctx.run(); coder.run()
    with torch.cuda.stream(s2):
        halves[0].run(); halves[1].run()
    cur.wait_stream(s2)
pipe = None
if os.environ.get("PROBE_PIPE"):
    # two batches in flight: while batch i is binarised, reconstructed and coded, the context indices of batch j are computed on a
    # third stream (behind the binarisation of i: beside the resolve kernel and the bool coder's kernels)
    ctx_b = lh.CtxSession([frames], replicate=streams)
    coder_b = lh.CoderSession(ctx_b)
    pipe = {"ctx": [ctx, ctx_b], "coder": [coder, coder_b], "done": [torch.cuda.Event(), torch.cuda.Event()], "ev": torch.cuda.Event(), "s3": torch.cuda.Stream(dev), "k": 0}
    ctx.run(); pipe["done"][0].record(torch.cuda.current_stream(dev))
def pipelined():
    cur = torch.cuda.current_stream(dev)
    i = pipe["k"] & 1; j = 1 - i
    pipe["k"] += 1
    s2.wait_stream(cur)
    cur.wait_event(pipe["done"][i])
    pipe["coder"][i].binarise()
    pipe["ev"].record(cur)
    with torch.cuda.stream(s2):
        sess.run()
    pipe["s3"].wait_event(pipe["ev"])
    with torch.cuda.stream(pipe["s3"]):
        pipe["ctx"][j].run()
        pipe["done"][j].record(pipe["s3"])
    pipe["coder"][i].finish()
    cur.wait_stream(s2)
def pipelined_early():
    # as pipelined, but the next batch's context indexing is enqueued at the start of the step (beside the binarisation and the
    # reconstruct kernel) instead of behind the binarisation
    cur = torch.cuda.current_stream(dev)
    i = pipe["k"] & 1; j = 1 - i
    pipe["k"] += 1
    s2.wait_stream(cur)
    pipe["s3"].wait_stream(cur)
    cur.wait_event(pipe["done"][i])
    with torch.cuda.stream(pipe["s3"]):
        pipe["ctx"][j].run()
        pipe["done"][j].record(pipe["s3"])
    pipe["coder"][i].binarise()
    with torch.cuda.stream(s2):
        sess.run()
    pipe["coder"][i].finish()
    cur.wait_stream(s2)
ev_emit = torch.cuda.Event()
def split_recon_main():
    # as split_coder, but the reconstruct kernel sits right behind the binarisation in the main queue and the coder's second half comes
    # from the other one: the reconstruct workgroups are placed first (two per CU), the resolve workgroups take what is left
    cur = torch.cuda.current_stream(dev)
    s2.wait_stream(cur)
    ctx.run(); coder.binarise()
    ev_emit.record(cur)
    sess.run()
    s2.wait_event(ev_emit)
    with torch.cuda.stream(s2):
        coder.finish()
    cur.wait_stream(s2)
def free_running():
    # no join per step: the reconstruct stream runs ahead of (or behind) the coder stream; joined by the caller's synchronize
    with torch.cuda.stream(s2):
        sess.run()
    ctx.run(); coder.run()
modes = (("one stream", serial), ("two streams", overlapped), ("recon last", recon_last), ("split coder", split_coder), ("split rmain", split_recon_main), ("recon last", recon_last))
if pipe:
    modes = (("one stream", serial), ("recon last", recon_last), ("pipelined", pipelined), ("pipe early", pipelined_early), ("pipelined", pipelined), ("pipe early", pipelined_early))
if halves:
    modes = (("one stream", serial), ("recon last", recon_last), ("halves first", recon_halves), ("halves last", recon_halves_last), ("halves first", recon_halves), ("halves last", recon_halves_last))
for name, fn in modes:
    fn(); torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(5):
        fn()
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / 5
    print("%-12s %.2f ms per step = %.0f MB/s of .264 (a1-a10), coded %d" % (name, dt * 1e3, streams * len(data) / dt / 1e6, sum(len(v) for v in coder.tags(streams - 1).values())), flush=True)
