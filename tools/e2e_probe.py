#!/usr/bin/env python3
"""end-to-end wall time of lh264_compress_batch (parse on host threads + staging + upload + kernels + download) on N copies of the
bench stream, and of lh264_pip_restore_batch on the result"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401  (HIP runtime first)
import losslessh264_amd as lh
data = open(os.path.join(ROOT, "tests", "golden", "streams", "BA_MW_D.264"), "rb").read()
for n in [int(a) for a in sys.argv[1:]] or [64, 512]:
    t0 = time.perf_counter()
    lh.compress_batch([data] * n, 16)          # first call of this size: the device and page-locked arenas grow (reported apart)
    cold = time.perf_counter() - t0
    t0 = time.perf_counter()
    res = lh.compress_batch([data] * n, 16)
    dt = time.perf_counter() - t0
    # the C call alone (the Python wrapper above copies every tag of every stream into bytes objects)
    import ctypes as C
    lib = lh.lib()
    ptrs = (C.c_char_p * n)(*([data] * n)); lens = (C.c_size_t * n)(*([len(data)] * n)); outs = (C.c_void_p * n)()
    t0 = time.perf_counter()
    rc = lib.lh264_compress_batch(ptrs, lens, n, 16, outs)
    dc = time.perf_counter() - t0
    assert rc == 0
    for i in range(n):
        assert lib.lh264_compressed_status(outs[i]) == 0
        lib.lh264_compressed_free(outs[i])
    assert all(e is None for _, _, e in res) and sum(len(b) for b in res[-1][1].values()) == 52742
    t1 = time.perf_counter()
    outs = lh.restore_batch([(m, t) for m, t, _ in res], 16)
    dr = time.perf_counter() - t1
    assert all(o == data for o in outs)
    print("streams=%d  lh264_compress_batch %.3f s (%.1f MB/s end to end)  Python compress_batch %.3f s (%.1f MB/s; first call of this size %.3f s)   restore_batch %.2f s (%.1f MB/s)" % (
        n, dc, n * len(data) / dc / 1e6, dt, n * len(data) / dt / 1e6, cold, dr, n * len(data) / dr / 1e6), flush=True)
