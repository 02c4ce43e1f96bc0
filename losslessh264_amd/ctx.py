"""Host side of the context-index path (row a8): which earlier picture is PAST, HBM staging, launch."""
import numpy as np

from . import _lib as L


def past_policy(frames):
    """index of the picture the reference's FreqImage holds as PAST for each frame (None = no PAST).

    FreqImage keeps two buffers and flips when frame_num changes (decoded_macroblock.h:119-123, called at every slice
    start with pCtx->iFrameNum, decode_slice.cpp:3034); the current picture is written into frame[cur], PAST is
    frame[1-cur]."""
    cur, last_fn, slot, out = 0, 0, [None, None], []
    for i, f in enumerate(frames):
        fn = getattr(f, "frame_num", i)
        if fn != last_fn:
            cur ^= 1
            last_fn = fn
        out.append(slot[1 - cur])
        slot[cur] = i
    return out


class CtxSession:
    """streams: list of lists of frames with mb_w, mb_h, mbs, slices, levels (int16[n,384]), frame_num"""

    def __init__(self, streams, device=0, replicate=1):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("losslessh264_amd: no GPU visible (there is no CPU fallback)")
        self.torch, self.lib = torch, L.lib()
        L.check(self.lib.lh264_set_device(device))
        dev = self.dev = torch.device("cuda", device)
        self.streams, self.replicate = streams, replicate
        mbs, lev, sl, info = [], [], [], []
        mo = so = 0
        self.max_mbs = 1
        for st in streams:
            ii = []
            pol = past_policy(st)
            for i, f in enumerate(st):
                n = f.mb_w * f.mb_h
                mbs.append(np.ascontiguousarray(f.mbs).view(np.uint8).reshape(-1))
                lev.append(np.ascontiguousarray(f.levels, dtype=np.int16).reshape(-1))
                sl.append(np.ascontiguousarray(f.slices).view(np.uint8).reshape(-1))
                ii.append((mo, so, f.mb_w, f.mb_h, pol[i]))
                mo += n
                so += len(f.slices)
                self.max_mbs = max(self.max_mbs, n)
            info.append(ii)
        self.info = info
        nd, nsd = mo, so
        self.n_mbs_distinct = nd
        rep = replicate
        self.d_mbs = torch.from_numpy(np.concatenate(mbs)).to(dev).repeat(rep)
        self.d_levels = torch.from_numpy(np.concatenate(lev)).to(dev).repeat(rep)
        self.d_slices = torch.from_numpy(np.concatenate(sl)).to(dev).repeat(rep)
        total = nd * rep
        self.d_nnz = torch.zeros(total * 24, dtype=torch.uint8, device=dev)
        self.d_nsyms = torch.zeros(total, dtype=torch.int16, device=dev)
        # compact layout (lh264.h, ABI 3): the symbols of all pictures in one pool, a macroblock's behind its predecessors'; the count
        # pass says how large the pool must be (8 bytes per coded symbol instead of 3,456 per macroblock)
        self.d_symoff = torch.zeros(total, dtype=torch.int32, device=dev)
        n_chains = len(streams) * rep
        n_jobs = sum(len(s) for s in streams) * rep
        self.d_symbase = torch.zeros(n_jobs + 1, dtype=torch.int64, device=dev)
        jobs = np.zeros(n_jobs, dtype=L.CTX_JOB_DTYPE)
        first = np.zeros(n_chains + 1, dtype=np.int32)
        bm, bl, bs = self.d_mbs.data_ptr(), self.d_levels.data_ptr(), self.d_slices.data_ptr()
        bn, bc = self.d_nnz.data_ptr(), self.d_nsyms.data_ptr()
        bo, bb = self.d_symoff.data_ptr(), self.d_symbase.data_ptr()
        j = 0
        self.job_mb_off = []
        self.job_of = {}
        for c in range(n_chains):
            first[c] = j
            blk = c // len(streams)
            ii = info[c % len(streams)]
            for fi, (m0, s0, w, h, past) in enumerate(ii):
                g = m0 + blk * nd
                jb = jobs[j]
                jb["mbs"], jb["levels"], jb["slices"] = bm + g * 128, bl + g * 768, bs + (s0 + blk * nsd) * 232
                jb["nnz_cur"] = bn + g * 24
                jb["nnz_past"] = 0 if past is None else bn + (ii[past][0] + blk * nd) * 24
                jb["n_syms"] = bc + g * 2
                jb["sym_off"], jb["sym_base"] = bo + g * 4, bb + j * 8
                jb["mb_w"], jb["mb_h"] = w, h
                self.job_mb_off.append(g)
                self.job_of[(c, fi)] = j
                j += 1
        first[n_chains] = j
        self.n_chains, self.n_jobs, self.n_mbs_total = n_chains, n_jobs, total
        self.d_first = torch.from_numpy(first).to(dev)
        # the count pass, once (the inputs of a session do not change): how many symbols the pool holds
        self.d_jobs = torch.from_numpy(jobs.view(np.uint8).reshape(-1)).to(dev)
        d_total = torch.zeros(1, dtype=torch.int64, device=dev)
        L.check(self.lib.lh264_ctx_count_chains(self.d_jobs.data_ptr(), self.d_first.data_ptr(), n_chains, n_jobs, self.max_mbs,
                                                d_total.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
        self.n_syms_total = int(d_total.item())
        self.d_syms = torch.zeros(max(1, self.n_syms_total) * 8, dtype=torch.uint8, device=dev)
        jobs["syms"], jobs["syms_cap"] = self.d_syms.data_ptr(), self.n_syms_total
        self.d_jobs = torch.from_numpy(jobs.view(np.uint8).reshape(-1)).to(dev)
        torch.cuda.synchronize(dev)

    def run(self):
        L.check(self.lib.lh264_ctx_index_chains(self.d_jobs.data_ptr(), self.d_first.data_ptr(), self.n_chains, self.n_jobs,
                                                self.max_mbs, self.torch.cuda.current_stream(self.dev).cuda_stream))

    def synchronize(self):
        self.torch.cuda.synchronize(self.dev)

    def frame_symbols(self, chain, frame):
        """-> (n_syms[n_mb], syms[n_mb, 432]) of one frame (the compact pool unpacked into the reference's fixed slots)"""
        ii = self.info[chain % len(self.streams)]
        g = ii[frame][0] + (chain // len(self.streams)) * self.n_mbs_distinct
        n = ii[frame][2] * ii[frame][3]
        ns = self.d_nsyms[g:g + n].cpu().numpy().view(np.uint16)
        off = self.d_symoff[g:g + n].cpu().numpy().view(np.uint32).astype(np.int64)
        base = int(self.d_symbase[self.job_of[(chain, frame)]].item())
        tot = int(off[-1] + ns[-1]) if n else 0          # (every macroblock's run starts on a multiple of 8 symbols)
        pool = self.d_syms[base * 8:(base + tot) * 8].cpu().numpy().view(L.CTX_SYM_DTYPE)
        sy = np.zeros((n, L.CTX_MAX_SYMS), dtype=L.CTX_SYM_DTYPE)
        for k in range(n):
            sy[k, :ns[k]] = pool[off[k]:off[k] + ns[k]]
        return ns, sy

    def frame_nnz(self, chain, frame):
        ii = self.info[chain % len(self.streams)]
        g = ii[frame][0] + (chain // len(self.streams)) * self.n_mbs_distinct
        n = ii[frame][2] * ii[frame][3]
        return self.d_nnz[g * 24:(g + n) * 24].cpu().numpy().reshape(n, 24)
