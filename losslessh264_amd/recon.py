"""Host-side driver of the reconstruct hot path: stages streams of macroblock records into HBM and
runs `lh264_recon_chains` (one workgroup per stream, frames in decode order)."""
import numpy as np

from . import _lib as L


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("losslessh264_amd: no GPU visible (there is no CPU fallback)")
    return torch


class ReconSession:
    """A batch of independent streams resident in HBM.

    streams: list of streams; a stream is a list of frames; a frame is an object with
      mb_w, mb_h, mbs (MB_DTYPE[n]), coeffs (int16[n,384]), slices (SLICE_DTYPE[k]), id, ref_ids
    (exactly what tests/golden_io.py / tests/refdump.py return, or what the host parser emits).
    Every frame gets its own padded picture in HBM (sized for 288 GB parts); ring=N instead reuses N picture buffers
    cyclically per stream, the way a decoded-picture-buffer host policy would (frames of one size only).
    """

    def __init__(self, streams, device=0, flags=0, replicate=1, share_records=True, ring=0):
        torch = _torch()
        self.torch = torch
        self.lib = L.lib()
        L.check(self.lib.lh264_set_device(device))
        self.dev = torch.device("cuda", device)
        self.streams = streams
        self.replicate = replicate
        self.n_chains = len(streams) * replicate
        # ---- flatten records of the distinct streams (replicas share the read-only records)
        mbs, coeffs, slices = [], [], []
        self.frame_info = []     # per distinct stream: list of (mb_off, slice_off, n_slices, mb_w, mb_h, id, ref_ids)
        mb_off = sl_off = 0
        self.max_w = self.max_h = 1
        for st in streams:
            info = []
            for f in st:
                n = f.mb_w * f.mb_h
                mbs.append(np.ascontiguousarray(f.mbs).view(np.uint8).reshape(-1))
                coeffs.append(np.ascontiguousarray(f.coeffs, dtype=np.int16).reshape(-1))
                slices.append(np.ascontiguousarray(f.slices).view(np.uint8).reshape(-1))
                info.append((mb_off, sl_off, len(f.slices), f.mb_w, f.mb_h, f.id, list(f.ref_ids)))
                mb_off += n
                sl_off += len(f.slices)
                self.max_w, self.max_h = max(self.max_w, f.mb_w), max(self.max_h, f.mb_h)
            self.frame_info.append(info)
        self.n_mbs_distinct = mb_off
        self.d_mbs = torch.from_numpy(np.concatenate(mbs)).to(self.dev)
        self.d_coeffs = torch.from_numpy(np.concatenate(coeffs)).to(self.dev)
        self.d_slices = torch.from_numpy(np.concatenate(slices)).to(self.dev)
        n_sl_distinct = sl_off
        if not share_records and replicate > 1:
            # physically independent streams: every replica owns its records in HBM
            self.d_mbs = self.d_mbs.repeat(replicate)
            self.d_coeffs = self.d_coeffs.repeat(replicate)
            self.d_slices = self.d_slices.repeat(replicate)
        # ---- pictures: one per frame per chain
        self.pic_off = []        # [chain][frame] -> byte offset into d_pics
        total = 0
        self.geo = {}
        for c in range(self.n_chains):
            offs = []
            for fi_, (_, _, _, w, h, _, _) in enumerate(self.frame_info[c % len(streams)]):
                if (w, h) not in self.geo:
                    self.geo[(w, h)] = L.pic_geometry(w, h)
                if ring and fi_ >= ring:
                    offs.append(offs[fi_ - ring])
                    continue
                offs.append(total)
                total += (self.geo[(w, h)][5] + 255) & ~255
            self.pic_off.append(offs)
        self.d_pics = torch.full((total,), 128, dtype=torch.uint8, device=self.dev)
        # ---- job table
        n_jobs = sum(len(self.frame_info[c % len(streams)]) for c in range(self.n_chains))
        jobs = np.zeros(n_jobs, dtype=L.JOB_DTYPE)
        chain_first = np.zeros(self.n_chains + 1, dtype=np.int32)
        base_m, base_c, base_s, base_p = (self.d_mbs.data_ptr(), self.d_coeffs.data_ptr(), self.d_slices.data_ptr(),
                                          self.d_pics.data_ptr())
        j = 0
        self.n_mbs_total = 0
        for c in range(self.n_chains):
            chain_first[c] = j
            info = self.frame_info[c % len(streams)]
            id2idx = {fi[5]: i for i, fi in enumerate(info)}
            blk = 0 if share_records else c // len(streams)
            for i, (mo, so, ns, w, h, fid, refs) in enumerate(info):
                mo += blk * self.n_mbs_distinct
                so += blk * n_sl_distinct
                sy, sc, oy, ou, ov, _ = self.geo[(w, h)]
                jb = jobs[j]
                jb["mbs"] = base_m + mo * 128
                jb["coeffs"] = base_c + mo * 768
                jb["slices"] = base_s + so * 232
                p = base_p + self.pic_off[c][i]
                jb["dst"] = (p + oy, p + ou, p + ov)
                for k, rid in enumerate(refs[:16]):
                    ri = id2idx.get(rid, None)
                    q = base_p + self.pic_off[c][ri if ri is not None else i]
                    jb["ref"][k] = (q + oy, q + ou, q + ov)
                jb["mb_w"], jb["mb_h"], jb["stride_y"], jb["stride_c"] = w, h, sy, sc
                jb["n_slices"], jb["flags"] = ns, flags
                self.n_mbs_total += w * h
                j += 1
        chain_first[self.n_chains] = j
        self.n_jobs = n_jobs
        self.d_jobs = torch.from_numpy(jobs.view(np.uint8).reshape(-1)).to(self.dev)
        self.d_chain_first = torch.from_numpy(chain_first).to(self.dev)
        torch.cuda.synchronize(self.dev)

    def _stream(self):
        return self.torch.cuda.current_stream(self.dev).cuda_stream

    def run(self):
        """one pass of the hot path over the whole batch (asynchronous on torch's current stream)"""
        L.check(self.lib.lh264_recon_chains(self.d_jobs.data_ptr(), self.d_chain_first.data_ptr(), self.n_chains,
                                            self.max_w, self.max_h, self._stream()))

    def time_kernel(self, iters):
        """mean ms per launch of the chain kernel, hipEvents on the launch stream"""
        ms = self.lib.lh264_time_recon_chains(self.d_jobs.data_ptr(), self.d_chain_first.data_ptr(), self.n_chains,
                                              self.max_w, self.max_h, iters, self._stream())
        if ms < 0:
            raise RuntimeError("liblh264: %s" % self.lib.lh264_last_error().decode())
        return ms

    def synchronize(self):
        self.torch.cuda.synchronize(self.dev)

    def picture(self, chain, frame, padded=False):
        """download one reconstructed picture -> [Y, U, V] numpy planes (MB-aligned size)"""
        info = self.frame_info[chain % len(self.streams)][frame]
        w, h = info[3], info[4]
        sy, sc, oy, ou, ov, total = self.geo[(w, h)]
        off = self.pic_off[chain][frame]
        buf = self.d_pics[off:off + total].cpu().numpy()
        out = []
        for p, (o, st) in enumerate(((oy, sy), (ou, sc), (ov, sc))):
            bs, pad = (8, L.PAD_C) if p else (16, L.PAD_Y)
            if padded:
                o0 = o - pad * st - pad
                hh, ww = h * bs + 2 * pad, w * bs + 2 * pad
            else:
                o0, hh, ww = o, h * bs, w * bs
            out.append(np.lib.stride_tricks.as_strided(buf[o0:], shape=(hh, ww), strides=(st, 1)).copy())
        return out
