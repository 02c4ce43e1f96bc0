"""losslessh264_amd - MI355X-native decode-reconstruct hot path of the lossless H.264 recompressor.

The product is the C-ABI library `liblh264.so` (include/lh264.h, hand-written gfx950 HIP kernels);
this package is the thin Python host side used by tests and bench.py: it only moves bytes into HBM
(torch tensors as plain device buffers) and calls the C ABI through ctypes.  There is no CPU
fallback: without the built library or without a GPU every compute call raises.
"""
from ._lib import lib, LibraryMissing, MB_DTYPE, SLICE_DTYPE, JOB_DTYPE, pic_geometry  # noqa: F401
from .recon import ReconSession  # noqa: F401
from .ctx import CtxSession, past_policy  # noqa: F401
from .parse import parse_stream, parse_file, parse_batch_time  # noqa: F401
from .coder import CoderSession  # noqa: F401
from .restore import restore, restore_batch, pack, restore_file, VERBATIM, compress_batch, compress_batch_handles  # noqa: F401

__all__ = ["lib", "LibraryMissing", "ReconSession", "CtxSession", "past_policy", "parse_stream", "parse_file", "CoderSession", "restore", "restore_batch", "pack", "restore_file", "VERBATIM", "compress_batch", "compress_batch_handles", "parse_batch_time", "MB_DTYPE", "SLICE_DTYPE", "JOB_DTYPE", "pic_geometry"]
