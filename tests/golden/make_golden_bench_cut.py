#!/usr/bin/env python3
"""Fixture for the bench configurations that use the first pictures of a longer stream (bench.py --config 2 / 4):
tests/golden/bench_cut.json = per (stream, picture count) the length and SHA-1 of every tagged stream the unmodified reference
(oracle/_ref/ref_dump, our shim around it; REF_DUMP_MAX_FRAMES = the picture count) wrote after exactly those pictures.
bench.py compares the coded bytes of every timed batch with it.  Data only."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import refdump  # noqa: E402

CUTS = [("syn720p_allI_4slices_8f.264", 4), ("tibbycabac.264", 24), ("test_cif_P_CABAC_slice.264", 24)]


def main():
    dump_bin = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
    out = {}
    for name, n in CUTS:
        tmp = tempfile.mkdtemp(prefix="lh264_cut_")
        env = dict(os.environ, REF_DUMP_MAX_FRAMES=str(n))
        subprocess.check_call([dump_bin, tmp, os.path.join(HERE, "streams", name)], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        frames = refdump.read_dump(os.path.join(tmp, name + ".dmp"))
        tags = refdump.read_dump.tags
        out["%s:%d" % (name, n)] = {"pictures": len(frames),
                                    "tags": {str(t): [len(b), hashlib.sha1(b).hexdigest()] for t, b in sorted(tags.items()) if t != 0x7fffffff}}
        print(name, n, len(frames), sum(len(b) for t, b in tags.items() if t != 0x7fffffff))
    json.dump(out, open(os.path.join(HERE, "bench_cut.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
