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


RESTORE_ITEM = None


def _item_type():
    global RESTORE_ITEM
    if RESTORE_ITEM is None:
        class Item(C.Structure):
            _fields_ = [("main_stream", C.c_char_p), ("main_len", C.c_size_t), ("tags", C.POINTER(C.c_char_p)), ("tag_len", C.POINTER(C.c_size_t)),
                        ("n_tags", C.c_int32), ("status", C.c_int32), ("out", C.c_void_p), ("out_cap", C.c_size_t), ("out_len", C.c_size_t)]
        RESTORE_ITEM = Item
    return RESTORE_ITEM


def restore_batch(items, threads=0, out_cap=None):
    """items: list of (main bytes, {tag: bytes}) -> list of restored byte strings (None where a restore failed), one host
    thread per stream (C ABI lh264_pip_restore_batch)."""
    lib = L.lib()
    Item = _item_type()
    arr = (Item * len(items))()
    keep = []
    for i, (main, tags) in enumerate(items):
        ptrs = (C.c_char_p * N_TAG_IDS)()
        lens = (C.c_size_t * N_TAG_IDS)()
        for t, b in tags.items():
            if 0 <= t < N_TAG_IDS:
                ptrs[t] = bytes(b)
                lens[t] = len(b)
        cap = out_cap or (4 * (len(main) + sum(len(b) for b in tags.values())) + 4096)
        buf = C.create_string_buffer(cap)
        keep.append((ptrs, lens, buf))
        it = arr[i]
        it.main_stream, it.main_len, it.tags, it.tag_len, it.n_tags = bytes(main), len(main), ptrs, lens, N_TAG_IDS
        it.out, it.out_cap = C.addressof(buf), cap
    L.check(lib.lh264_pip_restore_batch(C.byref(arr), len(items), threads))
    return [keep[i][2].raw[:arr[i].out_len] if arr[i].status == 0 else None for i in range(len(items))]


VERBATIM = 1


def pack(main, tags, flags=0):
    """default stream + tagged streams (or, with VERBATIM, the input itself as `main`) -> one LHPIP1 container (C ABI lh264_pip_pack)"""
    lib = L.lib()
    ptrs = (C.c_char_p * N_TAG_IDS)()
    lens = (C.c_size_t * N_TAG_IDS)()
    for t, b in tags.items():
        if 0 <= t < N_TAG_IDS:
            ptrs[t] = bytes(b)
            lens[t] = len(b)
    cap = lib.lh264_pip_pack_bound(len(main), lens, N_TAG_IDS)
    out = C.create_string_buffer(cap)
    n = C.c_size_t(0)
    L.check(lib.lh264_pip_pack(bytes(main), len(main), ptrs, lens, N_TAG_IDS, flags, out, cap, C.byref(n)))
    return out.raw[:n.value]


def restore_file(blob, size_hint=None):
    """LHPIP1 container -> the original stream (C ABI lh264_pip_restore_file)"""
    lib = L.lib()
    cap = size_hint or (4 * len(blob) + 4096)
    for _ in range(2):
        out = C.create_string_buffer(cap)
        n = C.c_size_t(0)
        rc = lib.lh264_pip_restore_file(bytes(blob), len(blob), out, cap, C.byref(n))
        if rc == 0:
            return out.raw[:n.value]
        if rc == -2 and n.value > cap:
            cap = n.value
            continue
        break
    raise RuntimeError("restore failed: " + lib.lh264_restore_error().decode())


class CompressedBatch:
    """the handles lh264_compress_batch returned: results are copied out stream by stream on demand (the C call itself does not touch
    Python objects: `seconds` is its wall time)"""

    def __init__(self, handles, n, seconds):
        self._h, self.n, self.seconds = handles, n, seconds

    def result(self, i):
        """-> (main bytes, {tag: bytes}, error text or None) of stream i"""
        lib = L.lib()
        h = self._h[i]
        ln = C.c_size_t(0)
        p = lib.lh264_compressed_main(h, C.byref(ln))
        main = C.string_at(p, ln.value) if ln.value else b""
        tags = {}
        for t in range(N_TAG_IDS):
            p = lib.lh264_compressed_tag(h, t, C.byref(ln))
            if p:
                tags[t] = C.string_at(p, ln.value)
        err = None if lib.lh264_compressed_status(h) == 0 else lib.lh264_compressed_error(h).decode()
        return main, tags, err

    def status(self, i):
        return L.lib().lh264_compressed_status(self._h[i])

    def free(self):
        lib = L.lib()
        for i in range(self.n):
            if self._h[i]:
                lib.lh264_compressed_free(self._h[i])
                self._h[i] = None
        self.n = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def compress_batch_handles(datas, threads=0, devices=None):
    """lh264_compress_batch / lh264_compress_batch_devices -> CompressedBatch (call .free() when done)"""
    import time
    lib = L.lib()
    n = len(datas)
    ptrs = (C.c_char_p * n)(*[bytes(d) for d in datas])
    lens = (C.c_size_t * n)(*[len(d) for d in datas])
    outs = (C.c_void_p * n)()
    t0 = time.perf_counter()
    if devices:
        devs = (C.c_int * len(devices))(*devices)
        rc = lib.lh264_compress_batch_devices(ptrs, lens, n, threads, devs, len(devices), outs)
    else:
        rc = lib.lh264_compress_batch(ptrs, lens, n, threads, outs)
    dt = time.perf_counter() - t0
    L.check(rc)
    return CompressedBatch(outs, n, dt)


def compress_batch(datas, threads=0, devices=None):
    """the whole compress direction behind one C call (lh264_compress_batch): list of Annex-B byte strings ->
    list of (main bytes, {tag: bytes}, error text or None).  devices: list of device indices to shard the batch over
    (lh264_compress_batch_devices); default: the current device"""
    b = compress_batch_handles(datas, threads, devices)
    res = [b.result(i) for i in range(b.n)]
    b.free()
    return res
