"""Seeded synthetic macroblock records: every reconstruct / deblock code path with random but *valid* inputs
(the kernels are defined on any record whose modes respect neighbour availability, exactly like the
reference's own random-block unit tests)."""
import numpy as np

from refdump import MB_DTYPE, SLICE_DTYPE

I4, I16, I8, P16, P16x8, P8x16, P8x8, P8x8R0, SKIP, IPCM = 1, 2, 4, 8, 0x10, 0x20, 0x40, 0x80, 0x100, 0x200


class SynFrame:
    pass


def _coeffs(rng, n, density, amp):
    c = np.zeros((n, 384), dtype=np.int16)
    mask = rng.random((n, 384)) < density
    vals = rng.integers(-amp, amp + 1, (n, 384))
    c[mask] = vals[mask]
    return c


def make_stream(seed, mb_w, mb_h, n_frames, p_frames=True, t8=False, pcm=False, n_slices=1, idc=0,
                amp=600, density=0.08, weighted=False):
    """returns a list of SynFrame (same attributes as golden frames); frame 0 is all-intra"""
    rng = np.random.default_rng(seed)
    frames = []
    n = mb_w * mb_h
    for fi in range(n_frames):
        f = SynFrame()
        f.id, f.mb_w, f.mb_h = fi, mb_w, mb_h
        is_p = p_frames and fi > 0
        nref = min(fi, 3) if is_p else 0
        f.ref_ids = [fi - 1 - k for k in range(nref)]
        # slices: split raster MBs into n_slices runs
        bounds = np.linspace(0, n, n_slices + 1).astype(int)
        f.slices = np.zeros(n_slices, dtype=SLICE_DTYPE)
        for s in range(n_slices):
            sl = f.slices[s]
            sl["first_mb"], sl["n_mbs"] = bounds[s], bounds[s + 1] - bounds[s]
            sl["slice_type"] = 0 if is_p else 2
            sl["deblock_idc"] = idc if idc != 3 else int(rng.integers(0, 3))
            sl["alpha_c0_offset"] = int(rng.integers(-3, 4)) * 2
            sl["beta_offset"] = int(rng.integers(-3, 4)) * 2
            sl["n_refs"] = max(nref, 1)
            sl["luma_dc_weight"] = 16
            sl["ref_slot"][:] = -1
            sl["ref_slot"][:nref] = np.arange(nref)
            if weighted and is_p:
                sl["weighted_pred"] = 1
                sl["luma_log2_denom"] = int(rng.integers(0, 7))
                sl["chroma_log2_denom"] = int(rng.integers(0, 7))
                sl["luma_weight"][:] = rng.integers(-20, 100, 16)
                sl["luma_offset"][:] = rng.integers(-20, 20, 16)
                sl["chroma_weight"][:] = rng.integers(-20, 100, (16, 2))
                sl["chroma_offset"][:] = rng.integers(-20, 20, (16, 2))
        slice_of = np.zeros(n, dtype=np.int32)
        for s in range(n_slices):
            slice_of[bounds[s]:bounds[s + 1]] = s
        mbs = np.zeros(n, dtype=MB_DTYPE)
        coeffs = _coeffs(rng, n, density, amp)
        for k in range(n):
            x, y = k % mb_w, k // mb_w
            m = mbs[k]
            sid = slice_of[k]
            m["slice_id"] = sid
            left = x > 0 and slice_of[k - 1] == sid
            top = y > 0 and slice_of[k - mb_w] == sid
            tl = x > 0 and y > 0 and slice_of[k - mb_w - 1] == sid
            tr = y > 0 and x + 1 < mb_w and slice_of[k - mb_w + 1] == sid
            qp = int(rng.integers(10, 52))
            m["qp_y"] = qp
            m["qp_c"] = (min(51, max(0, qp + int(rng.integers(-3, 4)))), min(51, max(0, qp + int(rng.integers(-3, 4)))))
            choices = [I4, I16] + ([I8] if t8 else []) + ([IPCM] if pcm else [])
            if is_p:
                choices += [P16, P16x8, P8x16, P8x8, P8x8R0, SKIP, SKIP, P16]
            typ = int(rng.choice(choices))
            m["mb_type"] = typ
            cbp_l, cbp_c = int(rng.integers(0, 16)), int(rng.integers(0, 3))
            if typ == I16:
                cbp_l = int(rng.choice([0, 15]))
            if typ in (SKIP, IPCM):
                cbp_l = cbp_c = 0
            m["cbp"] = cbp_l | (cbp_c << 4)
            use_t8 = t8 and typ in (I8, P16, P16x8, P8x16) and (typ == I8 or rng.random() < 0.5)
            if typ == I8:
                use_t8 = True
            m["flags"] = 1 if use_t8 else 0
            # coefficients only where cbp says so (what the reference parser leaves behind)
            c = coeffs[k]
            if typ == IPCM:
                c[:] = rng.integers(0, 256, 384)
                m["flags"] |= 2
            else:
                for b8 in range(4):
                    if not (cbp_l >> b8) & 1:
                        blk = c[b8 * 64:(b8 + 1) * 64]
                        if typ == I16:
                            dc = blk[::16].copy()
                            blk[:] = 0
                            blk[::16] = dc
                        else:
                            blk[:] = 0
                if cbp_c == 0:
                    c[256:] = 0
                elif cbp_c == 1:
                    dc = c[256::16].copy()
                    c[256:] = 0
                    c[256::16] = dc
                if typ == SKIP:
                    c[:] = 0
            # nzc: nonzero counts per 4x4 (raster layout of the reference), DC of I16 / chroma DC excluded
            nz = np.zeros(24, dtype=np.uint8)
            if typ != IPCM:
                for zb in range(16):
                    bx = (zb & 1) | ((zb >> 2) & 1) << 1
                    by = ((zb >> 1) & 1) | ((zb >> 3) & 1) << 1
                    blk = c[(zb >> 2) * 64:(zb >> 2) * 64 + 64] if use_t8 else c[zb * 16:zb * 16 + 16]
                    cnt = np.count_nonzero(blk[1:] if (typ == I16 and not use_t8) else blk)
                    nz[by * 4 + bx] = min(cnt, 16)
                cmap = [16, 17, 20, 21, 18, 19, 22, 23]
                for j in range(8):
                    nz[cmap[j]] = np.count_nonzero(c[256 + j * 16 + 1:256 + j * 16 + 16])
            else:
                nz[:] = 16
            m["nzc"] = nz
            # intra modes respecting availability
            if typ == I16:
                opts = [6] + ([0, 5] if top else []) + ([1, 4] if left else []) + ([2] if (top and left) else []) + \
                       ([3] if (top and left and tl) else [])
                m["intra_mode"][0] = int(rng.choice(opts))
            if typ in (I4, I8, I16, IPCM):
                copts = [6] + ([2, 5] if top else []) + ([1, 4] if left else []) + ([0] if (top and left) else []) + \
                        ([3] if (top and left and tl) else [])
                m["chroma_mode"] = int(rng.choice(copts))
            if typ == I4:
                for zb in range(16):
                    bx = (zb & 1) | ((zb >> 2) & 1) << 1
                    by = ((zb >> 1) & 1) | ((zb >> 3) & 1) << 1
                    a_l = left if bx == 0 else True
                    a_t = top if by == 0 else True
                    a_tl = (tl if (bx == 0 and by == 0) else (top if by == 0 else (left if bx == 0 else True)))
                    if by == 0:
                        a_tr = top if bx < 3 else tr
                    else:
                        # inside the MB the top-right block is available iff it precedes in z-order
                        zt = ((bx + 1) & 1) | (((by - 1) & 1) << 1) | (((bx + 1) >> 1) << 2) | (((by - 1) >> 1) << 3)
                        a_tr = bx < 3 and zt < zb
                    opts = [11]
                    if a_t:
                        opts += [0, 10, 12, 13]
                        if a_tr:
                            opts += [3, 7]
                    if a_l:
                        opts += [1, 9, 8]
                    if a_t and a_l:
                        opts += [2]
                        if a_tl:
                            opts += [4, 5, 6]
                    m["intra_mode"][by * 4 + bx] = int(rng.choice(opts))
            if typ == I8:
                av = (1 if top else 0) | (2 if tl else 0) | (4 if left else 0) | (8 if tr else 0)
                m["intra_avail"] = av
                for i8 in range(4):
                    bx, by = i8 & 1, i8 >> 1
                    a_l = left if bx == 0 else True
                    a_t = top if by == 0 else True
                    a_tl = [tl, top, left, True][i8]
                    a_tr = [top, tr, True, False][i8]
                    opts = [11]
                    if a_t:
                        opts += [0, 10, 12, 13]
                        if a_tr:
                            opts += [3, 7]
                    if a_l:
                        opts += [1, 9, 8]
                    if a_t and a_l:
                        opts += [2]
                        if a_tl:
                            opts += [4, 5, 6]
                    m["intra_mode"][by * 8 + bx * 2] = int(rng.choice(opts))
            if typ & 0x1F8:
                # motion: moderate vectors plus a few far outliers that hit the reference's MV clipping
                def rmv():
                    if rng.random() < 0.05:
                        return rng.integers(-4000, 4000, 2)
                    return rng.integers(-40, 41, 2)
                mv = np.zeros((16, 2), dtype=np.int16)
                ref = np.zeros(4, dtype=np.int8)
                sub = np.zeros(4, dtype=np.uint8)
                if typ in (P16, SKIP):
                    mv[:] = rmv()
                    ref[:] = 0 if typ == SKIP else int(rng.integers(0, nref))
                elif typ == P16x8:
                    mv[:8] = rmv(); mv[8:] = rmv()
                    ref[:2] = int(rng.integers(0, nref)); ref[2:] = int(rng.integers(0, nref))
                elif typ == P8x16:
                    a, b2 = rmv(), rmv()
                    for b4 in range(16):
                        mv[b4] = a if (b4 & 3) < 2 else b2
                    ref[0] = ref[2] = int(rng.integers(0, nref)); ref[1] = ref[3] = int(rng.integers(0, nref))
                else:
                    for q in range(4):
                        sub[q] = int(rng.choice([1, 2, 4, 8]))
                        ref[q] = 0 if typ == P8x8R0 else int(rng.integers(0, nref))
                        qx, qy = (q & 1) * 2, (q >> 1) * 2
                        vs = [rmv() for _ in range(4)]
                        for j in range(4):
                            jx, jy = j & 1, j >> 1
                            b4 = (qy + jy) * 4 + qx + jx
                            if sub[q] == 1:
                                mv[b4] = vs[0]
                            elif sub[q] == 2:
                                mv[b4] = vs[jy]
                            elif sub[q] == 4:
                                mv[b4] = vs[jx]
                            else:
                                mv[b4] = vs[j]
                m["mv"], m["ref_idx"], m["sub_type"] = mv, ref, sub
        f.mbs, f.coeffs = mbs, coeffs
        f.covered = np.ones(n, dtype=np.uint8)
        frames.append(f)
    return frames
