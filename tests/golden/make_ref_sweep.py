#!/usr/bin/env python3
"""The reference's console application over EVERY stream it ships (res/ and roundtriptest/): tests/golden/ref_sweep.json.

For each stream the unmodified reference CLI built by oracle/Makefile (oracle/_ref/h264dec) is run in this container in compress
mode (`h264dec in.264 out.pip`) and then in restore mode (`h264dec out.pip back.264`); the JSON keeps, per stream, the input's size
and SHA-1, the size and SHA-1 of every file the compressor wrote (`main` = out.pip, `<n>` = out.pip.<n>), whether the reference
itself restored the input bit for bit, and its exit codes.  The streams are copied to tests/golden/streams/ (data files the
reference's own tests hold).  Nothing of this runs on the GPU box: the tests read the JSON.
"""
import glob
import hashlib
import json
import os
import shutil
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def sha1(b):
    return hashlib.sha1(b).hexdigest()


def main():
    cli = os.path.join(ROOT, "oracle", "_ref", "h264dec")
    streams = sorted(glob.glob(os.path.join(REF, "res", "*.264")) + glob.glob(os.path.join(REF, "res", "*.jsv")) +
                     glob.glob(os.path.join(REF, "res", "*.h264")) + glob.glob(os.path.join(REF, "roundtriptest", "*.264")))
    tmp = tempfile.mkdtemp(prefix="lh264_sweep_")
    jobs = []
    for s in streams:
        base = os.path.basename(s)
        wd = os.path.join(tmp, base); os.makedirs(wd)
        jobs.append((s, base, wd))
    # each reference process allocates 8.8 GiB of prior tables: three at a time
    def run_all(make_cmd):
        running, todo, rcs = [], list(jobs), {}
        while todo or running:
            while todo and len(running) < 3:
                s, base, wd = todo.pop()
                running.append((base, subprocess.Popen(make_cmd(s, wd), cwd=wd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)))
            base, p = running.pop(0)
            try:
                rcs[base] = p.wait(timeout=900)
            except subprocess.TimeoutExpired:
                p.kill(); rcs[base] = -999
        return rcs
    rc_c = run_all(lambda s, wd: [cli, s, os.path.join(wd, "out.pip")])
    rc_r = run_all(lambda s, wd: [cli, os.path.join(wd, "out.pip"), os.path.join(wd, "back.264")])
    out = {}
    for s, base, wd in jobs:
        data = open(s, "rb").read()
        files = {}
        p = os.path.join(wd, "out.pip")
        if os.path.exists(p):
            b = open(p, "rb").read(); files["main"] = [len(b), sha1(b)]
        for q in glob.glob(p + ".*"):
            b = open(q, "rb").read(); files[q.rsplit(".", 1)[1]] = [len(b), sha1(b)]
        back = os.path.join(wd, "back.264")
        ok = os.path.exists(back) and open(back, "rb").read() == data
        out[base] = {"dir": os.path.basename(os.path.dirname(s)), "bytes": len(data), "sha1": sha1(data), "files": files,
                     "compress_rc": rc_c[base], "restore_rc": rc_r[base], "reference_roundtrip": bool(ok)}
        dst = os.path.join(HERE, "streams", base)
        if not os.path.exists(dst):
            shutil.copy(s, dst); os.chmod(dst, 0o644)
        print(base, len(data), "->", sum(v[0] for v in files.values()), "roundtrip" if ok else "NO roundtrip", flush=True)
    json.dump(out, open(os.path.join(HERE, "ref_sweep.json"), "w"), indent=1, sort_keys=True)
    shutil.rmtree(tmp)


if __name__ == "__main__":
    main()
