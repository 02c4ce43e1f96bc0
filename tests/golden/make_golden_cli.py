#!/usr/bin/env python3
"""Whole-stream outputs of the reference's console application: tests/golden/cli_<stream>.npz.

For every stream under tests/golden/streams the unmodified reference CLI built by oracle/Makefile (oracle/_ref/h264dec) is
run in compress mode (`h264dec in.264 out.pip`) in this container; the fixture keeps the files it wrote: `main` = out.pip
(the default stream) and `tag_<n>` = out.pip.<n>.  They pin (a) the product's default stream (lh264_parser_main_stream),
(b) the device coder over whole streams, (c) the restore direction (reference-written files -> the original bytes).
Fixtures are data only.  Streams the reference cannot compress are skipped.
"""
import glob
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def main():
    cli = os.path.join(ROOT, "oracle", "_ref", "h264dec")
    streams = sorted(glob.glob(os.path.join(HERE, "streams", "*")))
    if len(sys.argv) > 1:                              # only the named streams
        streams = [s for s in streams if os.path.basename(s) in sys.argv[1:]]
    tmp = tempfile.mkdtemp(prefix="lh264_cli_")
    running = []
    todo = list(streams)
    while todo or running:
        while todo and len(running) < 3:            # each reference process allocates 8.8 GiB of prior tables
            s = todo.pop()
            wd = os.path.join(tmp, "w_" + os.path.basename(s))
            os.makedirs(wd)
            running.append(subprocess.Popen([cli, s, os.path.join(tmp, os.path.basename(s) + ".pip")], cwd=wd,
                                            stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL))
        running.pop(0).wait()
    for s in streams:
        base = os.path.basename(s)
        p = os.path.join(tmp, base + ".pip")
        if not os.path.exists(p):
            print(base, ": no output, skipped")
            continue
        out = {"main": np.frombuffer(open(p, "rb").read(), dtype=np.uint8)}
        for q in glob.glob(p + ".*"):
            out["tag_" + q.rsplit(".", 1)[1]] = np.frombuffer(open(q, "rb").read(), dtype=np.uint8)
        path = os.path.join(HERE, "cli_" + base + ".npz")
        np.savez_compressed(path, **out)
        print("%s: main %d B, %d tags, %d B in all (input %d B) -> %d KB" % (base, len(out["main"]), len(out) - 1,
              sum(len(v) for v in out.values()), os.path.getsize(s), os.path.getsize(path) // 1024))


if __name__ == "__main__":
    main()
