"""Restore direction: default stream + tagged streams -> the original H.264 bytes (C ABI
lh264_pip_restore; host code, see csrc/host/pip_restore.h)."""
import ctypes as C

from . import _lib as L

N_TAG_IDS = 72


def restore(main, tags, size_hint=None):
    """main: bytes of the default stream; tags: {tag id: bytes}.  Returns the restored Annex-B stream."""
    lib = L.lib()
    ptrs = (C.c_char_p * N_TAG_IDS)()
    lens = (C.c_size_t * N_TAG_IDS)()
    for t, b in tags.items():
        if 0 <= t < N_TAG_IDS:
            ptrs[t] = bytes(b)
            lens[t] = len(b)
    cap = size_hint or (4 * (len(main) + sum(len(b) for b in tags.values())) + 4096)
    for _ in range(2):
        out = C.create_string_buffer(cap)
        n = C.c_size_t(0)
        rc = lib.lh264_pip_restore(bytes(main), len(main), ptrs, lens, N_TAG_IDS, out, cap, C.byref(n))
        if rc == 0:
            return out.raw[:n.value]
        if rc == -2 and n.value > cap:
            cap = n.value
            continue
        break
    raise RuntimeError("restore failed: " + lib.lh264_restore_error().decode())
