"""Robustness of the host front end (CPU): the parser + symbolizer, compiled with AddressSanitizer and UBSan, must survive
truncated, corrupted, empty and tiny inputs without any memory error (the reference's own tests feed lost-packet and error
streams through its decoder: test/api/decoder_test.cpp BA_MW_D_IDR_LOST / P_LOST / Error_I_P)."""
import glob
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parser_survives_damaged_streams(tmp_path):
    exe = str(tmp_path / "parser_stress")
    host = os.path.join(ROOT, "losslessh264_amd", "csrc", "host")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "native", "parser_stress.cpp"),
                           os.path.join(host, "h264_parser.cpp"), os.path.join(host, "pip_symbols.cpp"), "-o", exe])
    # every stream of up to 64 KB (27 of the 44: CAVLC and CABAC, lost packets, I_PCM, scaling lists, FMO/ASO headers) and three larger ones
    # with the 8x8 transform, CABAC P pictures and multiple reference pictures; 67 damaged variants of each.  (All 44 streams take eight
    # minutes under the sanitizers; the whole CPU suite is meant to run in a few.)
    streams = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "streams", "*")))
    streams = [s for s in streams if os.path.getsize(s) <= 65536 or os.path.basename(s) in ("tibbycabac.264", "tibby8x8cavlc.264", "MR1_BT_A.h264")]
    assert len(streams) >= 25
    r = subprocess.run([exe] + streams, capture_output=True, timeout=900,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert r.returncode == 0, (r.stdout.decode()[-2000:], r.stderr.decode()[-6000:])
    assert b"cases=" in r.stdout


def test_restore_survives_damaged_files(tmp_path):
    """the restore direction and the default-stream writer under ASan/UBSan: clean files of the reference's console application
    restore exactly; damaged, truncated and missing tag streams are rejected or decoded to something else without memory errors"""
    import numpy as np
    exe = str(tmp_path / "restore_stress")
    host = os.path.join(ROOT, "losslessh264_amd", "csrc", "host")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "native", "restore_stress.cpp"),
                           os.path.join(host, "h264_parser.cpp"), os.path.join(host, "pip_symbols.cpp"),
                           os.path.join(host, "pip_restore.cpp"), "-o", exe])
    bases = []
    for name in ("SVA_BA2_D.264", "SVA_BA1_B.264", "test_vd_1d.264", "CI_MW_D.264", "tibby8x8cavlc.264", "test_qcif_cabac.264", "tibbycabac.264"):
        z = np.load(os.path.join(ROOT, "tests", "golden", "cli_" + name + ".npz"))
        base = str(tmp_path / name.rsplit(".", 1)[0])
        with open(base + ".264", "wb") as f:
            f.write(open(os.path.join(ROOT, "tests", "golden", "streams", name), "rb").read())
        with open(base + ".pip", "wb") as f:
            f.write(z["main"].tobytes())
        for k in z.files:
            if k.startswith("tag_"):
                with open(base + ".pip." + k[4:], "wb") as f:
                    f.write(z[k].tobytes())
        bases.append(base)
    r = subprocess.run([exe] + bases, capture_output=True, timeout=900,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert r.returncode == 0, (r.stdout.decode()[-2000:], r.stderr.decode()[-6000:])
    assert b"cases=" in r.stdout
