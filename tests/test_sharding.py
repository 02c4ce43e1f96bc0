"""Multi-GPU path on CPU: world-size-2 gloo run of the stream sharding + result fan-in used by bench.py --gpus N."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp

from losslessh264_amd.shard import partition_by_work, gather_records


def test_partition_covers_and_balances():
    work = [99] * 10 + [396] * 3 + [3600]
    for ws in (1, 2, 3, 4, 8):
        b = partition_by_work(work, ws)
        assert b[0][0] == 0 and b[-1][1] == len(work)
        assert all(b[i][1] == b[i + 1][0] for i in range(ws - 1))
    b = partition_by_work([100] * 512, 8)
    assert [e - s for s, e in b] == [64] * 8


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import golden_io
    import oracle_lib as O
    streams = [golden_io.load("SVA_BA2_D.264")[:2], golden_io.load("SVA_BA1_B.264")[:2], golden_io.load("SVA_BA2_D.264")[:3]]
    work = [sum(f.mb_w * f.mb_h for f in s) for s in streams]
    s0, s1 = partition_by_work(work, world)[rank]
    recs = []
    for sid in range(s0, s1):            # each rank reconstructs only its share (oracle stands in for the GPU here)
        pics = {}
        for f in streams[sid]:
            dst = O.HostPic(f.mb_w, f.mb_h)
            O.recon_frame(f.mbs, f.coeffs, f.slices, dst, [pics[r] for r in f.ref_ids], 0)
            pics[f.id] = dst
        last = streams[sid][-1]
        recs.append([sid, golden_io.crc(pics[last.id].plane(0)), last.crc_fin[0]])
    allrec = gather_records(np.array(recs, dtype=np.int64).reshape(-1, 3), dist)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        q.put(allrec.tolist())


def test_two_rank_gloo_sharding():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1, 2]          # every stream processed exactly once
    assert all(r[1] == r[2] for r in res)                   # and bit-exact with the reference's planes
