"""Stream-level data parallelism: streams are independent, so a node's GPUs each take a contiguous share
of the stream list balanced by macroblock count; the only communication is the fan-in of small per-stream result
records (torch.distributed: RCCL on GPUs, gloo in the CPU tests)."""
import numpy as np


def partition_by_work(work, world_size):
    """contiguous partition of `work` (macroblocks per stream) into world_size shares with near-equal sums.
    returns list of (start, end) per rank."""
    work = np.asarray(work, dtype=np.int64)
    total = int(work.sum())
    bounds, acc, start = [], 0, 0
    cum = np.cumsum(work)
    for r in range(world_size):
        target = total * (r + 1) / world_size
        end = int(np.searchsorted(cum, target - 1e-9, side="left")) + 1 if r < world_size - 1 else len(work)
        end = max(end, start)
        end = min(end, len(work))
        bounds.append((start, end))
        start = end
    return bounds


def gather_records(local_records, dist, device=None):
    """fixed-size int64 result records (e.g. stream id, bytes, checksum) of every rank -> one array on all ranks.
    device: where the collective's tensors live (the rank's GPU under RCCL; None = host memory, gloo)"""
    import torch
    rec = torch.as_tensor(np.asarray(local_records, dtype=np.int64))
    if rec.ndim == 1:
        rec = rec.reshape(-1, 1)
    rec = rec.to(device) if device is not None else rec
    n_local = torch.tensor([rec.shape[0]], dtype=torch.int64, device=device)
    counts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(dist.get_world_size())]
    dist.all_gather(counts, n_local)
    nmax = int(max(int(c.item()) for c in counts))
    pad = torch.zeros((nmax, rec.shape[1]), dtype=torch.int64, device=device)
    pad[:rec.shape[0]] = rec
    out = [torch.zeros_like(pad) for _ in range(dist.get_world_size())]
    dist.all_gather(out, pad)
    return np.concatenate([o[:int(c.item())].cpu().numpy() for o, c in zip(out, counts)], axis=0)
