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
    streams = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "streams", "*")))
    assert len(streams) >= 8
    r = subprocess.run([exe] + streams, capture_output=True, timeout=900,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert r.returncode == 0, (r.stdout.decode()[-2000:], r.stderr.decode()[-6000:])
    assert b"cases=" in r.stdout
