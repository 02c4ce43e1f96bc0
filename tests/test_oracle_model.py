"""Row a8 on CPU: the context-index oracle + the host's PAST policy against what the reference's own MacroblockModel
did (fixtures from oracle/_ref/ref_dump's hooks on getNonzerosPrior*/getACPrior*/get*DCIntPrior).  Bit-exact."""
import numpy as np
import pytest

import golden_io
import oracle_lib as O
from losslessh264_amd.ctx import past_policy


@pytest.mark.parametrize("name", golden_io.list_fixtures())
def test_symbols_match_reference(name):
    frames = golden_io.load(name)
    pol = past_policy(frames)
    imgs = O.model_nnz_images(frames, pol)
    nsym = 0
    for i, f in enumerate(frames):
        past = imgs[pol[i]] if pol[i] is not None else None
        got = O.model_frame_symbols(f, imgs[i], past)
        for k in range(f.mb_w * f.mb_h):
            if not f.covered[k]:
                continue
            want = f.syms[k]
            assert len(got[k]) == len(want), (name, i, k)
            assert np.array_equal(got[k]["kind"], want["kind"]) and np.array_equal(got[k]["value"], want["value"]) \
                and np.array_equal(got[k]["prior"], want["prior"]), (name, i, k)
            nsym += len(want)
            # the neighbours the reference saw == our FreqImage restatement
            nei = f.nei[k]
            if nei[0][0]:
                assert np.array_equal(nei[0][1:], imgs[i][k - 1]), (name, i, k, "LEFT")
            if nei[1][0]:
                assert np.array_equal(nei[1][1:], imgs[i][k - f.mb_w]), (name, i, k, "ABOVE")
            if nei[2][0]:
                assert past is not None and np.array_equal(nei[2][1:], past[k]), (name, i, k, "PAST")
    assert nsym > 0
