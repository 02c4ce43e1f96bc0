"""The C-ABI library loads and exports every symbol include/lh264.h declares (no compute calls: no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "lh264.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(lh264_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_header_symbols():
    import __graft_entry__ as g
    g.build()
    from losslessh264_amd import _lib
    L = ctypes.CDLL(_lib.SO_PATH)
    names = _declared()
    assert len(names) >= 14
    for n in names:
        assert hasattr(L, n), "missing export " + n
    assert sorted(_lib.EXPORTS) == names
    assert _lib.lib().lh264_abi_version() == 3


def test_record_layouts_match_header():
    # sizes asserted against a tiny C program compiled from the header
    import subprocess, tempfile
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "lh264.h"
int main(void){ printf("%zu %zu %zu %zu %zu %zu\n", sizeof(lh264_mb_t), sizeof(lh264_slice_t), sizeof(lh264_frame_job_t),
  offsetof(lh264_mb_t, mv), offsetof(lh264_slice_t, ref_slot), offsetof(lh264_frame_job_t, mb_w)); return 0; }
'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")])
        out = subprocess.check_output([os.path.join(d, "t")]).decode().split()
    from losslessh264_amd import MB_DTYPE, SLICE_DTYPE, JOB_DTYPE
    assert [int(x) for x in out] == [MB_DTYPE.itemsize, SLICE_DTYPE.itemsize, JOB_DTYPE.itemsize,
                                     MB_DTYPE.fields["mv"][1], SLICE_DTYPE.fields["ref_slot"][1], JOB_DTYPE.fields["mb_w"][1]]


def test_compute_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from losslessh264_amd import _lib
    L = _lib.lib()
    assert L.lh264_device_count() <= 0
    assert L.lh264_recon_frames(None, 1, 1, 1, None) != 0
    assert b"no HIP device" in L.lh264_last_error()
    import losslessh264_amd as lh
    import synth
    with pytest.raises(RuntimeError):
        lh.ReconSession([synth.make_stream(1, 2, 2, 1)])
