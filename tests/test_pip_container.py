"""The recompressor's default stream (the '.pip' file itself) and, further down, the restore direction
(row f2), against whole-stream outputs of the reference's console application (tests/golden/cli_*.npz, written by
tests/golden/make_golden_cli.py from the unmodified reference built in the build container)."""
import glob
import os

import numpy as np
import pytest

import golden_io
import losslessh264_amd as lh

STREAMS = os.path.join(golden_io.GOLDEN_DIR, "streams")
CLI = sorted(os.path.basename(p)[4:-4] for p in glob.glob(os.path.join(golden_io.GOLDEN_DIR, "cli_*.npz")))


def cli_fixture(name):
    z = np.load(os.path.join(golden_io.GOLDEN_DIR, "cli_" + name + ".npz"))
    return z["main"].tobytes(), {int(k[4:]): z[k].tobytes() for k in z.files if k.startswith("tag_")}


@pytest.mark.parametrize("name", CLI)
def test_default_stream_matches_reference_cli(name):
    data = open(os.path.join(STREAMS, name), "rb").read()
    frames, err, main = lh.parse_file(data)
    assert err == ""
    ref_main, _ = cli_fixture(name)
    assert main == ref_main


def test_feed_file_gives_the_same_pictures_as_feed():
    data = open(os.path.join(STREAMS, "test_vd_1d.264"), "rb").read()
    a, _ = lh.parse_stream(data)
    b, _, _ = lh.parse_file(data)
    assert len(a) == len(b)
    for f, g in zip(a, b):
        assert np.array_equal(f.mbs, g.mbs) and np.array_equal(f.coeffs, g.coeffs) and np.array_equal(f.syn, g.syn)
