#!/usr/bin/env python3
"""bench.py - one "step" = one pass of the whole device-side compress direction over one batch
of independent streams resident in HBM: inverse transforms, intra/inter prediction, in-loop deblocking and reference padding
(a1-a7, `recon_chain_kernel`), the recompressor's per-coefficient context-model prior lookup (a8, the ctx_* kernels) and the adaptive
binary arithmetic coder that produces the compressed bytes (a9/a10, the coder_* kernels).  `value` counts a stream as recompressed
only when its tagged byte streams exist in HBM.

Workloads (BASELINE.json configs; `--config`, default 1 = the configuration the metric is quoted on):
  1  res/BA_MW_D.264 (Baseline CAVLC, QCIF, 100 frames) x 512 independent streams per GPU
  2  1280x720 all-intra CAVLC, 4 slices per picture (the SURVEY's synthetic pattern through the reference's encoder) x 1024
  3  1920x1080 I/P Baseline (same generator, -iper 16), the largest batch one GPU holds comfortably
  4  the CABAC path: roundtriptest/tibbycabac.264 + res/test_cif_P_CABAC_slice.264 alternating, 1024 streams
Every replica owns its records and pictures in HBM.  The streams are parsed by the product's own host front end.

N > 1 (`--gpus N`): one process per GPU.  Under torchrun the ranks exist already; otherwise this script starts them itself (fresh
processes, before anything touches a GPU).  The global stream list (N x the per-GPU batch) is cut by macroblock count
(losslessh264_amd/shard.py), every rank compresses its share, the per-stream result records are gathered over RCCL and checked
once on rank 0.  Weak scaling; no data-path collective.

Prints ONE JSON line (the contract in the task description).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

INTRA_BYTES_PER_MB = 1280      # 768 coeff + 128 record read, 384 written           
INTER_BYTES_PER_MB = 1883      # + 603 reference samples for a 16x16 partition       
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)

CONFIGS = {
    1: {"what": "configs[1]: res/BA_MW_D.264 (QCIF, 100 frames) x %d independent streams per GPU", "streams": ["BA_MW_D.264"], "n": 512, "frames": None},
    2: {"what": "configs[2]: 1280x720 all-intra CAVLC, 4 slices per picture (synthetic pattern, reference encoder, QP 26), %d frames x %d streams per GPU",
        "streams": ["syn720p_allI_4slices_8f.264"], "n": 1024, "frames": 4},
    3: {"what": "configs[3]: 1920x1080 I/P Baseline (synthetic pattern, reference encoder, -iper 16), %d frames x %d streams per GPU",
        "streams": ["syn1080p_IP_8f.264"], "n": 256, "frames": 8},
    4: {"what": "configs[4]: CABAC path, roundtriptest/tibbycabac.264 + res/test_cif_P_CABAC_slice.264 alternating, first %d frames x %d streams per GPU",
        "streams": ["tibbycabac.264", "test_cif_P_CABAC_slice.264"], "n": 1024, "frames": 24},
}


# SURVEY 8(d), Config 2 (= BASELINE.json configs[1]): "res/BA_MW_D.264 (... also BANM_MW_D, BA1_Sony_D, BAMQ1/2_JVC_C, BA1_FT_C, SVA_BA1_B/BA2_D,
# MIDR/NRF/MPS_MW)"
MIXED_STREAMS = ["BA_MW_D.264", "BANM_MW_D.264", "BA1_Sony_D.jsv", "BAMQ1_JVC_C.264", "BAMQ2_JVC_C.264", "BA1_FT_C.264", "SVA_BA1_B.264",
                 "SVA_BA2_D.264", "MIDR_MW_D.264", "NRF_MW_E.264", "MPS_MW_A.264"]


def spawn_ranks(args):
    """--gpus N without a launcher: N fresh child processes (this one never touches a GPU), rank 0's line is ours"""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # a rank that dies (no such device, out of memory ...) must not leave the others waiting in a collective for ever
    import threading
    buf = []
    reader = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc = 0
    while any(p.poll() is None for p in procs):
        bad = [p.returncode for p in procs if p.poll() is not None and p.returncode != 0]
        if bad:
            rc = bad[0]
            for p in procs:
                if p.poll() is None:
                    p.kill()            # (exactly the children started above)
            break
        time.sleep(0.2)
    for p in procs:
        rc = rc or p.wait()
    reader.join(timeout=10)
    out = (buf[0] if buf else b"").decode()
    lines = [l for l in out.splitlines() if l.startswith("{")]        # (a collective library may chat on stdout)
    sys.stdout.write((lines[-1] if lines else out) + "\n")
    sys.exit(rc)


def cpu_pass_factory(frames):
    """one pass of the oracle (our C restatement of the reference path, rows a1-a8) over a stream; thread-safe (own buffers)"""
    import ctypes as C
    import numpy as np
    import oracle_lib as O
    from losslessh264_amd.ctx import past_policy
    L = O.lib()
    L.orc_model_frame_symbols.restype = C.c_long
    pol = past_policy(frames)
    nmax = max(f.mb_w * f.mb_h for f in frames)
    prep = [(np.ascontiguousarray(f.mbs), np.ascontiguousarray(f.slices), np.ascontiguousarray(f.levels, dtype=np.int16)) for f in frames]

    def one_pass():
        syms = np.zeros(nmax * 432, dtype=O.ORC_SYM_DTYPE)
        nsy = np.zeros(nmax, dtype=np.uint16)
        pics, imgs = {}, []
        for i, f in enumerate(frames):
            dst = O.HostPic(f.mb_w, f.mb_h)
            O.recon_frame(f.mbs, f.coeffs, f.slices, dst, [pics[r] for r in f.ref_ids], 0)        # rows a1-a7
            pics[f.id] = dst
            mbs, sl, lv = prep[i]                                                                  # row a8
            img = np.empty(f.mb_w * f.mb_h * 24, dtype=np.uint8)
            past = imgs[pol[i]] if pol[i] is not None else None
            L.orc_model_frame_nnz(mbs.ctypes.data_as(C.c_void_p), lv.ctypes.data_as(C.c_void_p), f.mb_w * f.mb_h,
                                  past.ctypes.data_as(C.c_void_p) if past is not None else None, img.ctypes.data_as(C.c_void_p))
            L.orc_model_frame_symbols(mbs.ctypes.data_as(C.c_void_p), sl.ctypes.data_as(C.c_void_p), lv.ctypes.data_as(C.c_void_p),
                                      f.mb_w, f.mb_h, img.ctypes.data_as(C.c_void_p),
                                      past.ctypes.data_as(C.c_void_p) if past is not None else None,
                                      syms.ctypes.data_as(C.c_void_p), nsy.ctypes.data_as(C.c_void_p))
            imgs.append(img)
    return one_pass


def cpu_baseline(frames, stream_bytes, seconds, threads):
    """the oracle on `threads` host threads (one stream each, as SURVEY 8d asks: P = min(cores, streams)); bounded sample"""
    from concurrent.futures import ThreadPoolExecutor
    one = cpu_pass_factory(frames)
    one()                                                  # builds the oracle library on first use
    t0 = time.perf_counter()

    def worker(_):
        n = 0
        while time.perf_counter() - t0 < seconds:
            one(); n += 1
        return n
    with ThreadPoolExecutor(threads) as ex:
        reps = sum(ex.map(worker, range(threads)))
    dt = time.perf_counter() - t0
    return reps * stream_bytes / dt / 1e6, reps, dt


def reference_cli_baseline(stream, stream_bytes, procs):
    """the unmodified reference's console application (oracle/_ref/h264dec, built in the build container by oracle/Makefile and
    shipped like our own .so) compressing the stream on this box: its whole pipeline (parse + reconstruct + model + coder), one core per
    process, `procs` processes at once (each allocates 8.8 GiB of prior tables).  In-call time from its own 'decode time' line."""
    import re
    import tempfile
    exe = os.path.join(ROOT, "oracle", "_ref", "h264dec")
    if not os.path.exists(exe):
        return None
    try:
        with tempfile.TemporaryDirectory() as d:
            t0 = time.perf_counter()
            ps = []
            for i in range(procs):
                wd = os.path.join(d, str(i)); os.makedirs(wd)
                ps.append(subprocess.Popen([exe, stream, os.path.join(wd, "o.pip")], cwd=wd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
            outs = [p.communicate(timeout=300)[0].decode(errors="replace") for p in ps]
            wall = time.perf_counter() - t0
            secs = [float(m.group(1)) for m in (re.search(r"decode time:\s*([0-9.]+) sec", o) for o in outs) if m]
            size = sum(os.path.getsize(os.path.join(d, "0", f)) for f in os.listdir(os.path.join(d, "0")) if f.startswith("o.pip"))
        if len(secs) != procs or any(p.returncode != 0 for p in ps):
            return None
        return {"value": sum(stream_bytes / s for s in secs) / 1e6, "unit": "MB/s", "cores": procs, "kind": "reference",
                "sample": "oracle/_ref/h264dec %s -> .pip, %d processes at once (one core each): whole pipeline, in-call %.3f s each on average "
                          "(wall %.1f s with the start-up of the 8.8 GiB tables), %d bytes written per stream" % (
                              os.path.basename(stream), procs, sum(secs) / procs, wall, size)}
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=1, choices=sorted(CONFIGS))
    ap.add_argument("--streams", type=int, default=0, help="independent streams per GPU (0: the config's batch)")
    ap.add_argument("--frames", type=int, default=0, help="pictures per stream (0: the config's)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-host", action="store_true", help="skip the host-stage (front end / restore / end-to-end) measurements")
    ap.add_argument("--no-pipeline", action="store_true", help="every batch on its own: no second batch's context indexing beside the coder")
    ap.add_argument("--mixed", action="store_true", help="configs[1] as a heterogeneous batch: the eleven Baseline streams SURVEY 8(d) lists, interleaved "
                                                         "(46 of each = 506 streams per GPU), the longest first within every group of eleven")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)                                   # does not return

    import numpy as np
    import torch
    import golden_io
    import losslessh264_amd as lh
    from losslessh264_amd import _lib as L
    from losslessh264_amd import shard

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: running %d rank(s)\n" % (args.gpus, world, world))
    dist = None
    rehearse = False
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # LH264_BENCH_REHEARSE=1: every rank on GPU 0 and the collectives over gloo - a rehearsal of the N > 1 path on a one-GPU box
        # (RCCL needs a GPU per rank); the line then says so and is no scaling measurement
        rehearse = os.environ.get("LH264_BENCH_REHEARSE") == "1"
        if rehearse:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    coll_dev = None if (world > 1 and os.environ.get("LH264_BENCH_REHEARSE") == "1") else dev
    collectives = "none (one rank)" if world == 1 else ("gloo (rehearsal)" if coll_dev is None else "nccl (RCCL)")
    if world > 1 and coll_dev is not None:
        # the data path has no collective; RCCL carries the barrier, the max of the step times and the gathered result records.  If its
        # first collective fails on this node, every rank sees the failure and they fall back together to gloo on host memory
        try:
            probe = torch.ones(1, device=dev)
            dist.all_reduce(probe)
            torch.cuda.synchronize(dev)
            assert int(probe.item()) == world
        except Exception as e:                          # noqa: BLE001
            sys.stderr.write("bench.py: RCCL collective failed on rank %d (%r): falling back to gloo for barrier / max / gather\n" % (rank, e))
            try:
                dist.destroy_process_group()
            except Exception:                           # noqa: BLE001
                pass
            os.environ["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 1)
            dist.init_process_group("gloo", rank=rank, world_size=world)
            coll_dev = None
            collectives = "gloo (RCCL failed on this node: %s)" % type(e).__name__

    cfg = dict(CONFIGS[args.config])
    if args.mixed:
        assert args.config == 1, "--mixed is configs[1]'s heterogeneous batch"
        cfg["streams"] = MIXED_STREAMS
        cfg["n"] = 46 * len(MIXED_STREAMS)
        cfg["what"] = "configs[1], heterogeneous: " + ", ".join(MIXED_STREAMS) + " interleaved, %d streams per GPU"
    per_gpu = args.streams or cfg["n"]
    n_frames = args.frames or cfg["frames"]
    # ---- the streams of this configuration, through the product's own host front end ------------------------------------------------------
    distinct, datas, mains = [], [], []
    for name in cfg["streams"]:
        data = open(os.path.join(ROOT, "tests", "golden", "streams", name), "rb").read()
        frames, perr, main_stream = lh.parse_file(data)
        assert perr == "", perr
        if n_frames:
            frames = frames[:n_frames]
        distinct.append(frames); datas.append(data); mains.append(main_stream)
    if args.mixed:                                          # the longest stream first within every group (a workgroup = a stream)
        idx = sorted(range(len(distinct)), key=lambda k: -sum(f.mb_w * f.mb_h for f in distinct[k]))
        distinct, datas, mains = [distinct[k] for k in idx], [datas[k] for k in idx], [mains[k] for k in idx]
        cfg["streams"] = [cfg["streams"][k] for k in idx]
    # input bytes a stream stands for: the whole file, or the share of the pictures used
    full = [lh.parse_file(d)[0] for d in datas] if n_frames else distinct
    stream_bytes = [len(d) * sum(f.mb_w * f.mb_h for f in fr) / max(1, sum(f.mb_w * f.mb_h for f in fu)) for d, fr, fu in zip(datas, distinct, full)]
    whole = [len(fr) == len(fu) for fr, fu in zip(distinct, full)]         # the stream is used as a whole: the reference's file sizes apply
    del full
    # ---- the global stream list and this rank's share (static partition by macroblock count, no exchange during work) ----------
    n_global = per_gpu * world
    kind_of = [g % len(distinct) for g in range(n_global)]
    work = [sum(f.mb_w * f.mb_h for f in distinct[k]) for k in kind_of]
    g0, g1 = shard.partition_by_work(work, world)[rank]
    my_kinds = kind_of[g0:g1]
    # sessions take a list of distinct streams + a replica count: chain c holds distinct[c % len(distinct)]
    assert all(my_kinds[i] == (my_kinds[0] + i) % len(distinct) for i in range(len(my_kinds)))
    order = [distinct[(my_kinds[0] + i) % len(distinct)] for i in range(len(distinct))] if my_kinds else distinct
    n_local = len(my_kinds)
    assert n_local % len(distinct) == 0 or len(distinct) == 1
    rep = max(1, n_local // len(distinct))
    sess = lh.ReconSession(order, device=local_rank, replicate=rep, share_records=False)
    ctx = lh.CtxSession(order, device=local_rank, replicate=rep)
    out_cap = 1 << 16
    while out_cap < 0.6 * max(stream_bytes):               # room for the largest tag of the largest stream
        out_cap <<= 1
    coder = lh.CoderSession(ctx, out_cap=out_cap)
    local_bytes = sum(stream_bytes[k] for k in my_kinds)
    # Two batches in flight (a pipeline of depth two over the queue of batches a service works through): while batch i is binarised,
    # reconstructed and coded, the context indices of batch j are computed on a third stream, behind the binarisation of i - beside the
    # resolve kernel and the bool coder's kernels, whose waves mostly wait.  Every step still launches every kernel of rows a1-a10 once;
    # the second batch is a second set of device buffers with the same streams (synthetic input).  Only where that second set fits easily.
    pipeline = not args.no_pipeline and ctx.n_mbs_total * 930 + ctx.n_syms_total * 8 < 40e9      # (records, levels, images + the symbol pool)
    ctxs, coders = [ctx], [coder]
    if pipeline:
        ctxs.append(lh.CtxSession(order, device=local_rank, replicate=rep))
        coders.append(lh.CoderSession(ctxs[1], out_cap=out_cap))

    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    k_ms = {"recon": 0.0, "ctx": 0.0, "coder": 0.0}

    # The reconstruct kernel (rows a1-a7) and the context-index + coder stages (a8-a10) read and write different buffers: a step
    # launches them on two HIP streams and joins them at its end (31.5 -> 28.4 ms).  What overlaps is the reconstruct kernel and the
    # binarisation (small workgroups that fit beside it); the resolve kernel does not fit beside two reconstruct workgroups per CU
    # (80 KB of LDS and 97 registers per lane against 84 KB and 4 x 104 taken) and runs behind them - tools/overlap_timeline.py shows
    # it.  (Per-stage times are taken from extra steps run on one stream, below.)
    side = torch.cuda.Stream(dev)
    third = torch.cuda.Stream(dev)
    ctx_done = [torch.cuda.Event(), torch.cuda.Event()]
    ev_binarised = torch.cuda.Event()
    recon_events = []
    state = {"k": 0, "pipelined": False}

    # the reconstruct kernel is enqueued first: it takes four wave slots of 104 registers on every SIMD, the coder's kernels fit one more
    # wave beside them.  Enqueued behind the coder's first half, whichever of the two reached a CU first kept it, and a change as small
    # as a fill kernel more or less moved the step between 26.6 and 34.5 ms (LH264_BENCH_RECON_FIRST=0: that order, for comparison)
    recon_first = os.environ.get("LH264_BENCH_RECON_FIRST", "1") == "1"
    # (experiment, measured and not adopted: the next batch's context indices enqueued at the top of the step instead of behind this
    # batch's binarisation - they then compete with the count / emit kernels the host waits for: 31.1 against 25.9 ms a step)
    ctx_early = os.environ.get("LH264_BENCH_CTX_EARLY", "0") == "1"
    # (experiment, measured and not adopted: the reconstruct chain and the coder chain not joined at the ends of a step - 32.4 against
    # 25.6 ms a step on the bench batch, the same on the CIF batch, 187 against 198 ms on the 720p batch)
    free_run = os.environ.get("LH264_BENCH_FREE_RUN", "0") == "1"

    def recon_on_side():
        with torch.cuda.stream(side):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()         # (on the stream the kernel is launched on: its duration in the timed steps, beside the coder)
            sess.run()          # rows a1-a7: reconstruct + deblock + pad (one launch of the dominant kernel); enqueued behind the coder's
                                # first half, it starts at once: that call returns when its counting kernels are done
            e1.record()
            recon_events.append((e0, e1))

    def step(timed=False):
        cur = torch.cuda.current_stream(dev)
        if timed:
            ev[0].record()
            sess.run()
            ev[1].record()
            ctx.run()
            ev[2].record()
            coder.run()
            ev[3].record()
            return
        if not (free_run and state["pipelined"]):
            side.wait_stream(cur)   # (the end of the step before)
        if not state["pipelined"]:
            if recon_first:
                recon_on_side()
            ctx.run()           # row a8: per-coefficient context-model prior indices
            coder.run()         # rows a9/a10: binarisation, adaptive probabilities, bool coders -> the tagged byte streams
            if not recon_first:
                recon_on_side()
            cur.wait_stream(side)
            return
        i = state["k"] & 1
        j = 1 - i
        state["k"] += 1
        cur.wait_event(ctx_done[i])                 # batch i's context indices (row a8), computed during the step before
        if recon_first:
            recon_on_side()                         # rows a1-a7
        if ctx_early:
            with torch.cuda.stream(third):
                ctxs[j].run()
                ctx_done[j].record(third)
        coders[i].binarise()                        # rows a9/a10, first half, batch i
        ev_binarised.record(cur)
        if not recon_first:
            recon_on_side()
        if not ctx_early:
            third.wait_event(ev_binarised)
            with torch.cuda.stream(third):
                ctxs[j].run()                       # row a8 of the next batch
                ctx_done[j].record(third)
        coders[i].finish()                          # rows a9/a10, second half, batch i
        if not free_run:
            cur.wait_stream(side)

    step(timed=True)            # untimed priming pass on one stream: the coder's work memory is allocated here, not beside the first kernel
    torch.cuda.synchronize(dev)
    # a few steps with every batch on its own (two streams, no second batch in flight): reported beside the pipelined figure
    step()
    torch.cuda.synchronize(dev)
    t_u = time.perf_counter()
    for _ in range(3):
        step()
    torch.cuda.synchronize(dev)
    unpipelined_ms = (time.perf_counter() - t_u) / 3 * 1e3
    if pipeline:
        state["pipelined"] = True
        ctxs[0].run()
        ctx_done[0].record(torch.cuda.current_stream(dev))
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    recon_events.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    state["pipelined"] = False
    torch.cuda.synchronize(dev)
    # per-stage device time, hipEvents on the launch stream (torch's current stream is the stream the C ABI launches on), a few extra steps
    n_ev = max(2, min(args.steps, 5))
    for _ in range(n_ev):
        step(timed=True)
        torch.cuda.synchronize(dev)
        k_ms["recon"] += ev[0].elapsed_time(ev[1]); k_ms["ctx"] += ev[1].elapsed_time(ev[2]); k_ms["coder"] += ev[2].elapsed_time(ev[3])
    for k in k_ms:
        k_ms[k] /= n_ev
    types = np.concatenate([f.mbs["mb_type"] for fr in order for f in fr])
    n_intra = int(np.count_nonzero(types & 0x207))
    n_inter = int(np.count_nonzero(types & 0x1F8))
    alg_bytes = (n_intra * INTRA_BYTES_PER_MB + n_inter * INTER_BYTES_PER_MB) * rep
    # the dominant kernel's launch duration over the timed region (where it runs beside the coder's kernels); k_ms["recon"] is the same
    # kernel with the machine to itself
    recon_ms = sum(a.elapsed_time(b) for a, b in recon_events[:args.steps]) / max(1, min(args.steps, len(recon_events)))
    achieved = alg_bytes / (recon_ms * 1e-3) / 1e9

    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process; the figure comes from the
    # committed rocprofv3 --pmc passes over the same launch (profiles/traffic.json), corrected as the guide prescribes
    traffic = coder_traffic = None
    traffic_note = "no counters committed for this configuration and batch size"
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(str(args.config))
        if tj is not None and not args.mixed:
            have = L.lib().lh264_build_id().decode()
            if tj["streams"] != n_local:
                traffic_note = "profiles/traffic.json holds the counters of a %d-stream batch, this run has %d" % (tj["streams"], n_local)
            elif tj["build_id"] != have:
                traffic_note = "profiles/traffic.json was taken on build %s, this library is build %s: not reported" % (tj["build_id"], have)
            else:
                t = tj["kernels"]["recon_chain_kernel"]
                traffic = (2.0 * t["fetch_size_kb"] + t["write_size_kb"]) * 1024.0
                coder_traffic = sum((2.0 * v.get("fetch_size_kb", 0.0) + v.get("write_size_kb", 0.0)) * 1024.0 for k, v in tj["kernels"].items() if k.startswith("coder_"))
                traffic_note = "rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE per launch, %s, build %s" % (tj["source"], have)
    except (OSError, KeyError, ValueError):
        pass

    # the coder's own memory-bound figure: what its stages have to move (symbols read twice, 8-byte decision words and 2-byte list
    # entries written once and read once, the bytes written)
    dw, le = L.C.c_ulonglong(), L.C.c_ulonglong()
    L.check(L.lib().lh264_code_last_totals(L.C.byref(dw), L.C.byref(le)))
    n_ctx_syms = int(ctx.d_nsyms.view(torch.int16).to(torch.int64).sum().item())
    n_syn_syms = sum(len(f.syn_syms) for fr in order for f in fr) * rep
    lens = coder.d_len.cpu().numpy().reshape(n_local, L.N_TAG_SLOTS + 1)
    assert not lens[:, L.N_TAG_SLOTS].any(), "device coder status %s" % sorted(set(lens[:, L.N_TAG_SLOTS].tolist()))
    coded_local = lens[:, :35].astype(np.int64).sum(axis=1)
    coder_bytes = 2 * 8 * (n_ctx_syms + n_syn_syms) + 2 * 8 * dw.value + 2 * 2 * le.value + int(coded_local.sum())

    # ---- parity outside the timed region ---------------------------------------------------------------------------------------------------
    # every stream's compressed size must be what the reference wrote (fixture: the files of its console application), the last
    # picture of the first and last replica must have the reference decoder's plane CRCs where the fixture has them
    # (a stream used whole: the files of the reference's console application, tests/golden/cli_*.npz; the first pictures of a longer
    # stream: what the reference wrote after exactly those pictures, tests/golden/bench_cut.json - length and SHA-1 of every tag)
    import hashlib
    cuts = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_cut.json")))
    sweep = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_sweep.json")))
    ref_tagged, ref_tags = [], []
    for k, name in enumerate(cfg["streams"]):
        p = os.path.join(ROOT, "tests", "golden", "cli_" + name + ".npz")
        cut = cuts.get("%s:%d" % (name, len(distinct[k])))
        if os.path.exists(p) and whole[k]:
            z = np.load(p)
            ref_tags.append({int(f[4:]): (len(z[f]), hashlib.sha1(z[f].tobytes()).hexdigest()) for f in z.files if f.startswith("tag_")})
        elif whole[k] and name in sweep and sweep[name].get("compress_rc") == 0:
            # (the sweep of the unmodified reference over every shipped stream, tests/golden/ref_sweep.json: size and SHA-1 of every file)
            ref_tags.append({int(t): (v[0], v[1]) for t, v in sweep[name]["files"].items() if t.isdigit()})
        elif cut is not None and cut["pictures"] == len(distinct[k]):
            ref_tags.append({int(t): (v[0], v[1]) for t, v in cut["tags"].items()})
        else:
            ref_tags.append(None)
        ref_tagged.append(None if ref_tags[-1] is None else sum(v[0] for v in ref_tags[-1].values()))
    # every batch that was coded in the timed steps: both sets of buffers when two batches are in flight
    checked = {"batches": 0, "streams_sizes": 0, "streams_sha1": 0}
    from losslessh264_amd.coder import TAG_OF_SLOT
    for cd in coders:
        ln = cd.d_len.cpu().numpy().reshape(n_local, L.N_TAG_SLOTS + 1)
        assert not ln[:, L.N_TAG_SLOTS].any(), "device coder status %s" % sorted(set(ln[:, L.N_TAG_SLOTS].tolist()))
        checked["batches"] += 1
        for c in range(n_local):
            want = ref_tags[(my_kinds[0] + c) % len(distinct)]
            if want is None:
                continue
            got = {TAG_OF_SLOT[sl]: int(ln[c, sl]) for sl in range(35) if ln[c, sl]}
            assert got == {t: v[0] for t, v in want.items() if v[0]}, "stream %d: tag sizes %s, the reference wrote %s" % (c, got, {t: v[0] for t, v in want.items()})
            checked["streams_sizes"] += 1
        for k in range(len(distinct)):                  # the bytes of the first and the last replica of every distinct stream
            if ref_tags[k] is None:
                continue
            reps = [i for i in range(n_local) if (my_kinds[0] + i) % len(distinct) == k]
            for c in sorted({reps[0], reps[-1]}) if reps else []:
                tg = cd.tags(c)
                assert {t: hashlib.sha1(b).hexdigest() for t, b in tg.items()} == {t: v[1] for t, v in ref_tags[k].items() if v[0]}, "stream %d: coded bytes differ from the reference's" % c
                checked["streams_sha1"] += 1
    roundtrip = None
    if args.config == 1 and not args.mixed:
        ref_frames = golden_io.load("bench_BA_MW_D.264")
        for c in (0, n_local - 1):
            got = sess.picture(c, len(order[0]) - 1)
            assert [golden_io.crc(g) for g in got] == ref_frames[len(order[0]) - 1].crc_fin, "bench output differs from the reference"
        tags = coder.tags(n_local - 1)
        coded = sum(len(v) for v in tags.values())
        assert len(mains[0]) + coded == 53739, "compressed size differs from the reference's 53,739 bytes"
        roundtrip = {"ratio": (len(mains[0]) + coded) / len(datas[0]), "reference_ratio": 53739 / 55885, "roundtrip_ok": lh.restore(mains[0], tags) == datas[0]}
        assert roundtrip["roundtrip_ok"], "restore(compress(stream)) differs from the stream"
    else:
        # ratio of what was coded: the tagged streams against the share of the input they stand for (the default stream is per file)
        roundtrip = {"tagged_bytes_per_input_byte": float(coded_local.sum()) / max(1.0, local_bytes),
                     "coded_bytes_equal_reference_files": [w is not None for w in ref_tagged]}
        if all(w is not None for w in ref_tagged) and all(whole):
            # whole streams: size and round trip of one replica of every distinct stream, as for configs[1]
            ok = True
            for k in range(len(distinct)):
                c = next(i for i in range(n_local) if (my_kinds[0] + i) % len(distinct) == k)
                ok = ok and lh.restore(mains[k], coder.tags(c)) == datas[k]
            # the reference's sizes: its console application's files (fixture), or the sweep's record of them
            def ref_size(name):
                q = os.path.join(ROOT, "tests", "golden", "cli_" + name + ".npz")
                if os.path.exists(q):
                    zz = np.load(q)
                    return sum(len(zz[f]) for f in zz.files)
                return sum(v[0] for v in sweep[name]["files"].values())
            roundtrip.update({"ratio": sum(len(m) + w for m, w in zip(mains, ref_tagged)) / sum(len(d) for d in datas),
                              "reference_ratio": sum(ref_size(name) for name in cfg["streams"]) / sum(len(d) for d in datas), "roundtrip_ok": ok})
            assert ok, "restore(compress(stream)) differs from the stream"

    # ---- multi-GPU: the per-stream result records of every rank, gathered and checked once on rank 0 ---------------------------------------
    outv = coder.d_out.view(n_local, -1)
    sums = torch.cat([outv[i:i + 8].sum(dim=1, dtype=torch.int64) for i in range(0, n_local, 8)]).cpu().numpy() if n_local else np.zeros(0, np.int64)
    records = np.stack([np.arange(g0, g1, dtype=np.int64), coded_local, sums], axis=1) if n_local else np.zeros((0, 3), np.int64)
    n_records = n_local
    if dist is not None:
        allrec = shard.gather_records(records, dist, device=coll_dev)
        n_records = len(allrec)
        if rank == 0:
            assert sorted(allrec[:, 0].tolist()) == list(range(n_global)), "a stream was compressed twice or not at all"
            for k in range(len(distinct)):
                rows = allrec[np.array([kind_of[int(g)] == k for g in allrec[:, 0]])]
                assert len(set(map(tuple, rows[:, 1:].tolist()))) == 1, "replicas of one stream differ between ranks"

    host_stages = None
    if rank == 0 and world == 1 and not args.no_cpu and not args.no_host and args.config == 1 and not args.mixed:
        # host stages on this box's cores (rows f1 / f2): the front end (parse + default stream + syntax symbols) and the restore
        # direction (adaptive decode + CAVLC writer), one stream per thread, bounded to a few seconds each; and the whole compress
        # direction behind one C call, host bytes in -> host bytes out (parse, staging, PCIe, kernels, download)
        data = datas[0]
        tags = coder.tags(n_local - 1)
        ncpu = min(16, len(os.sched_getaffinity(0)))      # a one-GPU box's CPU share is 16 cores
        nb = 8 * ncpu
        pt, npic = lh.parse_batch_time([data] * nb, ncpu, keep=False)
        assert npic == nb * len(order[0])
        t1 = time.perf_counter()
        outs = lh.restore_batch([(mains[0], tags)] * nb, ncpu)
        rt = time.perf_counter() - t1
        assert all(o == data for o in outs)
        # each size twice: the first call of a size also grows the device and page-locked arenas (reported apart), the second is the
        # steady state of a service that keeps them
        e2e_times = {}
        for mult in (4, 16):                              # 512 streams: four groups already overlap parsing with the device stage
            first = None
            for attempt in range(2):
                # the C call (host bytes in, handles to host bytes out); every stream's status and a few streams' bytes are checked
                e2e = lh.compress_batch_handles([data] * (mult * nb), ncpu)
                et = e2e.seconds
                assert all(e2e.status(i) == 0 for i in range(e2e.n))
                for i in (0, e2e.n // 2, e2e.n - 1):
                    m_i, t_i, err_i = e2e.result(i)
                    assert err_i is None and t_i == tags and m_i == mains[0]
                e2e.free()
                if attempt == 0:
                    first = et
            e2e_times[mult * nb] = (et, first)
        host_stages = {"threads": ncpu, "front_end_MB_per_s": nb * len(data) / pt / 1e6, "restore_MB_per_s": nb * len(data) / rt / 1e6}
        for ns, (et, first) in e2e_times.items():
            host_stages["compress_batch_end_to_end_%d_streams_MB_per_s" % ns] = ns * len(data) / et / 1e6
            host_stages["compress_batch_end_to_end_%d_streams_first_call_MB_per_s" % ns] = ns * len(data) / first / 1e6

    if rank == 0:
        ms = dt / args.steps * 1e3
        total_bytes = local_bytes * world * args.steps        # every rank holds the same mix (weak scaling)
        a18 = k_ms["recon"] + k_ms["ctx"]
        what = cfg["what"] % ((per_gpu,) if args.config == 1 else (len(order[0]), per_gpu))
        mixed = None
        if args.mixed:
            # how much of the launch is the longest stream's own chain: one workgroup (or one (stream, partition) wave) cannot go faster
            lone = lh.ReconSession([order[0]], device=local_rank)
            lone.run(); torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); lone.run(); e1.record(); torch.cuda.synchronize(dev)
            mbs = [sum(f.mb_w * f.mb_h for f in fr) for fr in order]
            mixed = {"streams": cfg["streams"], "macroblocks_per_stream": mbs, "order": "the longest stream first within every group of %d consecutive workgroups" % len(order),
                     "recon_chain_kernel_ms_longest_stream_alone": e0.elapsed_time(e1),
                     "note": "one workgroup per stream: the launch cannot end before its longest stream's chain does; compare with stage_ms.a1_a7_recon_chain_kernel"}
        out = {
            "metric": "MB/s .264 recompressed (bit-exact roundtrip) + ratio, 1/2/4/8 MI355X",
            "value": total_bytes / dt / 1e6, "unit": "MB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/int16", "data": "synthetic batch: reference-shipped / reference-encoded streams parsed by the host front end, replicated",
            "config": {"workload": what, "streams_per_gpu": per_gpu, "frames_per_stream": [len(fr) for fr in order],
                       "input_bytes_per_stream": stream_bytes, "mbs_per_step_per_gpu": sess.n_mbs_total,
                       "stages_in_timed_region": "a1-a7 (IDCT, intra/inter prediction, deblocking, reference padding) + a8 (context-model prior index per "
                                                 "coefficient symbol) + a9/a10 (binarisation, adaptive probabilities, bool coders): the whole compress direction "
                                                 "on the device, records in HBM -> tagged byte streams in HBM; the host CAVLC/CABAC parse is not in this step",
                       "parallelism": "reconstruct: one workgroup per stream, one wave per MB row; context indices: a wave per macroblock (compact symbol pool); "
                                      "coder, two forms (config.coder_form): a lane per decision into per-partition runs (a stream's cells dealt evenly to 8 or 16 "
                                      "partitions), one wave per (stream, partition) for the adaptive probabilities - or, a few hundred small streams, a thread "
                                      "per symbol and one workgroup per stream; the bool coders: range walk in coarse chunks from checked candidate "
                                      "start states, one lane per 256 decisions (sums), one wave per (stream, tag) (carries); the reconstruct kernel runs on a "
                                      "second HIP stream beside the context-index and coder kernels; streams sharded across GPUs",
                       "coder_form": ("stream per workgroup" if (os.environ.get("LH264_CODER_PATH") == "sw" or (os.environ.get("LH264_CODER_PATH") != "wave" and 384 <= n_local < 1024
                                                                         and sess.n_mbs_total / max(1, n_local) <= 12288)) else "wave per (stream, partition)"),
                       "stage_ms_note": "stage times are from steps run on one stream; in the timed steps the stages overlap, ms_per_step is less than their sum",
                       "pipeline": ("two batches in flight: the context-index kernels (row a8) of the next batch run on a third HIP stream beside the second half of "
                                    "this batch's coder; every step launches every kernel of rows a1-a10 once" if pipeline else "none: every batch on its own"),
                       "ms_per_step_one_batch_in_flight": unpipelined_ms,
                       "stage_ms": {"a1_a7_recon_chain_kernel": k_ms["recon"], "a8_ctx_kernels": k_ms["ctx"], "a9_a10_coder_kernels": k_ms["coder"]},
                       "a1_a8_only_MB_per_s": local_bytes / (a18 * 1e-3) / 1e6,
                       "compression": roundtrip, "parity_checked": checked, "mixed_batch": mixed, "host_stages": host_stages,
                       "multi_gpu": {"result_records_gathered": n_records, "global_streams": n_global, "collectives": collectives,
                                     "work_list": "no scatter: every rank derives its share of the global stream list locally (partition_by_work); one gather of result records",
                                     "measured_on_hardware": ("rehearsal: all ranks on GPU 0, gloo collectives - not a scaling measurement" if rehearse
                                                              else "this line" if world > 1 else "single GPU; N > 1 unmeasured in this run")}},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_note, "kernel": "recon_chain_kernel", "kernel_ms": recon_ms, "kernel_ms_alone": k_ms["recon"], "algorithmic_bytes_per_launch": alg_bytes,
                         "coder_stage": {"bound": "hbm", "achieved": coder_bytes / (k_ms["coder"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                         "frac": coder_bytes / (k_ms["coder"] * 1e-3) / 1e9 / HBM_PEAK_GBS, "ms": k_ms["coder"], "algorithmic_bytes": coder_bytes, "traffic": coder_traffic,
                                         "note": "symbols read twice + 8-byte decision words and 2-byte list entries written and read once + output; "
                                                 "the stage is bound by two serial chains (per adaptive probability, per tag), not by HBM"}},
        }
        if world == 1 and not args.no_cpu:
            ncpu = min(16, len(os.sched_getaffinity(0)))
            v, reps, secs = cpu_baseline(order[0], stream_bytes[0], args.cpu_seconds, ncpu)
            cpu_model = ""
            try:
                cpu_model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
            except (OSError, IndexError):
                pass
            out["cpu_baseline"] = {"value": v, "unit": "MB/s", "cores": ncpu, "kind": "port", "cpu": cpu_model,
                                   "sample": "oracle (C restatement of rows a1-a8; the reference's coder a9/a10 is in reference_cli) over %s x %d passes on %d threads, "
                                             "%.1f s" % (cfg["streams"][0], reps, ncpu, secs)}
            nproc = max(1, min(ncpu, 4, int(os.sysconf("SC_PHYS_PAGES") * os.sysconf("SC_PAGE_SIZE") / (12 << 30))))
            ref = reference_cli_baseline(os.path.join(ROOT, "tests", "golden", "streams", cfg["streams"][0]), len(datas[0]), nproc)
            if ref is not None:
                out["cpu_baseline"]["reference_cli"] = ref
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
