"""Host side of the device coder (rows a9/a10/f4): stages the host symbol lists, allocates the per-stream prior hash tables
and output buffers in HBM, launches `lh264_code_chains` on the symbols of a CtxSession."""
import numpy as np

from . import _lib as L

TAG_OF_SLOT = list(range(34)) + [69]


class CoderSession:
    """ctx: a CtxSession over the same streams (frames must carry syn_syms / syn_off from parse_stream)"""

    def __init__(self, ctx, hash_cap=None, out_cap=1 << 16):
        if hash_cap is None:
            # the spill table of a stream (8 entries per "cell"): room for four adaptive probabilities per coded coefficient - measured:
            # 0.1 .. 0.5 per coefficient, the more the higher the bit rate
            nz = max(sum(int(np.count_nonzero(f.levels)) for f in st) for st in ctx.streams)
            hash_cap = 1 << 13
            while hash_cap * 2 < nz and hash_cap < (1 << 20):
                hash_cap <<= 1
        assert hash_cap <= 1 << 20 and hash_cap & (hash_cap - 1) == 0
        torch = ctx.torch
        self.ctx, self.torch, self.lib, dev = ctx, torch, ctx.lib, ctx.dev
        streams, rep = ctx.streams, ctx.replicate
        syms, offs, info = [], [], []
        so = oo = 0
        for st in streams:
            ii = []
            for f in st:
                n = f.mb_w * f.mb_h
                syms.append(np.ascontiguousarray(f.syn_syms).view(np.uint8).reshape(-1))
                offs.append(np.ascontiguousarray(f.syn_off, dtype=np.uint32))
                ii.append((so, oo, n))
                so += len(f.syn_syms)
                oo += n + 1
            info.append(ii)
        self.d_syn = torch.from_numpy(np.concatenate(syms)).to(dev)          # shared by replicas (read-only)
        self.d_off = torch.from_numpy(np.concatenate(offs)).to(dev)
        n_chains = ctx.n_chains
        self.hash_cap, self.out_cap = hash_cap, out_cap
        self.d_keys = torch.zeros(16, dtype=torch.int32, device=dev)          # unused by ABI 2
        self.d_cells = torch.zeros(n_chains * hash_cap * 16, dtype=torch.int32, device=dev)
        self.d_out = torch.zeros(n_chains * L.N_TAG_SLOTS * out_cap, dtype=torch.uint8, device=dev)
        self.d_len = torch.zeros(n_chains * (L.N_TAG_SLOTS + 1), dtype=torch.int32, device=dev)
        jobs = np.zeros(ctx.n_jobs, dtype=L.CODE_JOB_DTYPE)
        bs, bo = self.d_syn.data_ptr(), self.d_off.data_ptr()
        by, bc = ctx.d_syms.data_ptr(), ctx.d_nsyms.data_ptr()
        b_symoff, b_symbase = ctx.d_symoff.data_ptr(), ctx.d_symbase.data_ptr()
        j = 0
        for c in range(n_chains):
            ii = info[c % len(streams)]
            for (s0, o0, n) in ii:
                g = ctx.job_mb_off[j]
                jb = jobs[j]
                jb["syn_syms"], jb["syn_off"] = bs + s0 * 8, bo + o0 * 4
                jb["ctx_syms"], jb["ctx_n_syms"] = by, bc + g * 2               # (the compact layout: one pool for all pictures)
                jb["ctx_sym_off"], jb["ctx_sym_base"] = b_symoff + g * 4, b_symbase + j * 8
                jb["n_mbs"] = n
                j += 1
        st = np.zeros(n_chains, dtype=L.CODE_STREAM_DTYPE)
        for c in range(n_chains):
            st[c]["hash_keys"] = self.d_keys.data_ptr()
            st[c]["hash_cells"] = self.d_cells.data_ptr() + c * hash_cap * 64
            st[c]["out"] = self.d_out.data_ptr() + c * L.N_TAG_SLOTS * out_cap
            st[c]["out_len"] = self.d_len.data_ptr() + c * (L.N_TAG_SLOTS + 1) * 4
            st[c]["hash_cap"], st[c]["out_cap"] = hash_cap, out_cap
        self.d_jobs = torch.from_numpy(jobs.view(np.uint8).reshape(-1)).to(dev)
        self.d_streams = torch.from_numpy(st.view(np.uint8).reshape(-1)).to(dev)
        self.n_chains = n_chains
        torch.cuda.synchronize(dev)

    def run(self):
        """code every stream (the prior tables are adaptive state: cleared first)"""
        self.d_cells.zero_()
        L.check(self.lib.lh264_code_chains(self.d_jobs.data_ptr(), self.ctx.d_first.data_ptr(), self.d_streams.data_ptr(), self.n_chains,
                                           self.ctx.n_jobs, self.ctx.n_mbs_total, self.ctx.max_mbs,
                                           self.torch.cuda.current_stream(self.ctx.dev).cuda_stream))

    def binarise(self):
        """first half of run(): count, scan, emit the decision words (lh264_code_binarise_chains; synchronises the stream once)"""
        self.d_cells.zero_()
        L.check(self.lib.lh264_code_binarise_chains(self.d_jobs.data_ptr(), self.ctx.d_first.data_ptr(), self.d_streams.data_ptr(), self.n_chains,
                                                    self.ctx.n_jobs, self.ctx.n_mbs_total, self.ctx.max_mbs,
                                                    self.torch.cuda.current_stream(self.ctx.dev).cuda_stream))

    def finish(self):
        """second half of run(): adaptive probabilities and bool coders (lh264_code_finish_chains)"""
        L.check(self.lib.lh264_code_finish_chains(self.d_streams.data_ptr(), self.n_chains, self.torch.cuda.current_stream(self.ctx.dev).cuda_stream))

    def tags(self, chain):
        """-> {tag: bytes} of one stream (after run + synchronize)"""
        n = L.N_TAG_SLOTS
        lens = self.d_len[chain * (n + 1):(chain + 1) * (n + 1)].cpu().numpy()
        if lens[n] != 0:
            raise RuntimeError("device coder status %d (bits - 1: prior table full or invalid, 4: output overflow, 8: counter overflow, 16: internal)" % lens[n])
        base = chain * n * self.out_cap
        out = {}
        for slot in range(len(TAG_OF_SLOT)):
            if lens[slot]:
                out[TAG_OF_SLOT[slot]] = bytes(self.d_out[base + slot * self.out_cap: base + slot * self.out_cap + int(lens[slot])].cpu().numpy())
        return out
