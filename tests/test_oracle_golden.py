"""The oracle (C restatement) against the committed golden fixtures, which were produced by the reference
decoder itself (tests/golden/make_golden.py).  Runs on CPU, here and on the GPU box.  Bit-exact."""
import numpy as np
import pytest

import golden_io
import oracle_lib as O


@pytest.mark.parametrize("name", golden_io.list_fixtures())
def test_oracle_matches_reference(name):
    frames = golden_io.load(name)
    pics = {}
    for f in frames:
        refs = [pics[r] for r in f.ref_ids]
        cov = np.kron(f.covered.reshape(f.mb_h, f.mb_w).astype(bool), np.ones((16, 16), bool))
        pre = O.HostPic(f.mb_w, f.mb_h)
        O.recon_frame(f.mbs, f.coeffs, f.slices, pre, refs, O.NO_DEBLOCK | O.NO_EXPAND)
        if f.covered.all():
            for p in range(3):
                assert golden_io.crc(pre.plane(p)) == f.crc_pre[p], (name, f.id, "pre-deblock plane", p)
        dst = O.HostPic(f.mb_w, f.mb_h)
        O.recon_frame(f.mbs, f.coeffs, f.slices, dst, refs, 0)
        pics[f.id] = dst
        if f.has_final:
            for p in range(3):
                assert golden_io.crc(dst.plane(p)) == f.crc_fin[p], (name, f.id, "final plane", p)
        if f.fin is not None:
            for p in range(3):
                assert np.array_equal(dst.plane(p), f.fin[p])
        del cov


def test_expand_is_nearest_sample():
    import synth
    f = synth.make_stream(3, 5, 4, 1, p_frames=False)[0]
    dst = O.HostPic(f.mb_w, f.mb_h, fill=7)
    O.recon_frame(f.mbs, f.coeffs, f.slices, dst, [], 0)
    for p in range(3):
        inner, padded = dst.plane(p), dst.padded_plane(p)
        pad = 16 if p else 32
        assert np.array_equal(padded, np.pad(inner, pad, mode="edge"))
