"""Row a8 on the GPU: the HIP ctx-index kernels (through the C ABI) against the oracle and the reference-generated
fixtures.  Bit-exact."""
import numpy as np
import pytest

import golden_io
import oracle_lib as O

pytestmark = pytest.mark.gpu


def _check_stream(sess, chain, frames, against_reference):
    import losslessh264_amd as lh
    pol = lh.past_policy(frames)
    imgs = O.model_nnz_images(frames, pol)
    for i, f in enumerate(frames):
        assert np.array_equal(sess.frame_nnz(chain, i), imgs[i]), ("nnz image", i)
        ns, sy = sess.frame_symbols(chain, i)
        past = imgs[pol[i]] if pol[i] is not None else None
        want = O.model_frame_symbols(f, imgs[i], past)
        for k in range(f.mb_w * f.mb_h):
            w = want[k]
            assert ns[k] == len(w), (i, k, ns[k], len(w))
            g = sy[k][:ns[k]]
            assert np.array_equal(g["kind"], w["kind"]) and np.array_equal(g["value"], w["value"]) and np.array_equal(g["prior"], w["prior"]), (i, k)
            if against_reference and f.covered[k]:
                r = f.syms[k]
                assert len(r) == ns[k] and np.array_equal(g["prior"], r["prior"]) and np.array_equal(g["value"], r["value"]), (i, k, "vs reference")


@pytest.mark.parametrize("name", golden_io.list_fixtures())
def test_golden_symbols(name):
    import losslessh264_amd as lh
    frames = golden_io.load(name)
    sess = lh.CtxSession([frames])
    sess.run(); sess.synchronize()
    _check_stream(sess, 0, frames, True)


def test_synthetic_levels_batch():
    import losslessh264_amd as lh
    import synth
    streams = []
    for i in range(3):
        fr = synth.make_stream(seed=40 + i, mb_w=5 + i, mb_h=4, n_frames=3, t8=(i == 1), density=0.15, amp=40)
        for j, f in enumerate(fr):
            f.levels = f.coeffs          # any int16 content is a valid level array for the index computation
            f.frame_num = j
        streams.append(fr)
    sess = lh.CtxSession(streams, replicate=2)
    sess.run(); sess.synchronize()
    for c in range(sess.n_chains):
        _check_stream(sess, c, streams[c % 3], False)
