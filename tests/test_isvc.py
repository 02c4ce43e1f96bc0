"""The coarse boundary (include/lh264_isvc.h): an ISVCDecoder-compatible object in liblh264.so.

CPU part: exported symbols, structure layouts (against the reference's codec_api.h when /root/reference is present, and
against the committed numbers always), virtual-table order through the C view of the interface, loud failure without a GPU.
GPU part: a client application (tests/isvc_client.cpp) decodes the golden streams through the vtable and its YUV must hash to
the reference's decoder-test SHA-1 table (tests/golden/decoder_sha1.json, from test/api/decoder_test.cpp:82-126)."""
import ctypes
import glob
import hashlib
import json
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")
SO_DIR = os.path.join(ROOT, "losslessh264_amd")
REF_API = "/root/reference/codec/api/svc"

LAYOUT_PROBE = r'''
#include <stdio.h>
#include <stddef.h>
#include HEADER
int main() {
  printf("%zu %zu %zu %zu %zu %zu %zu ", sizeof(SDecodingParam), sizeof(SBufferInfo), sizeof(SSysMEMBuffer), sizeof(SDecoderCapability),
         sizeof(SParserBsInfo), sizeof(SDecoderStatistics), sizeof(OpenH264Version));
  printf("%zu %zu %zu %zu %zu %zu ", offsetof(SDecodingParam, eOutputColorFormat), offsetof(SDecodingParam, uiTargetDqLayer),
         offsetof(SDecodingParam, eEcActiveIdc), offsetof(SDecodingParam, bParseOnly), offsetof(SDecodingParam, sVideoProperty),
         offsetof(SBufferInfo, UsrData));
  printf("%zu %zu %zu ", offsetof(SBufferInfo, uiOutYuvTimeStamp), offsetof(SParserBsInfo, pDstBuff), offsetof(SDecoderStatistics, iAvgLumaQp));
  printf("%d %d %d %d %d %d %d\n", (int)dsInitialOptExpected, (int)dsDstBufNeedExpan, (int)DECODER_OPTION_GET_STATISTICS,
         (int)ERROR_CON_SLICE_MV_COPY_CROSS_IDR_FREEZE_RES_CHANGE, (int)videoFormatI420, (int)cmUnsupportedData, (int)FEEDBACK_UNKNOWN_NAL);
  return 0;
}
'''
# the numbers the probe prints for the reference's codec_api.h on x86-64 (regenerate: run the probe with REF_API)
LAYOUT_EXPECTED = "40 48 20 36 552 84 16 8 16 20 24 28 24 16 520 56 8192 32768 12 7 23 5 2"


def _build_lib():
    import __graft_entry__ as g
    g.build()


def _compile(src_path, out, extra=()):
    cmd = ["g++", "-O1", "-std=c++11", "-I", INC, src_path, "-o", out, os.path.join(SO_DIR, "liblh264.so"),
           "-Wl,--allow-shlib-undefined", "-Wl,-rpath," + SO_DIR, "-Wl,-rpath,/opt/rocm/lib"] + list(extra)
    subprocess.check_call(cmd)


def _probe(tmp_path, header, incdir):
    src = tmp_path / "probe.cpp"
    src.write_text(LAYOUT_PROBE.replace("HEADER", '"%s"' % header))
    exe = str(tmp_path / "probe")
    subprocess.check_call(["g++", "-std=c++11", "-I", incdir, str(src), "-o", exe])
    return subprocess.check_output([exe]).decode().strip()


def test_exports():
    _build_lib()
    txt = open(os.path.join(INC, "lh264_isvc.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = set(re.findall(r"\b(Wels[A-Za-z]+|lh264_isvc_[a-z0-9_]+)\s*\(", txt)) - {"WelsTraceCallback"}
    assert {"WelsCreateDecoder", "WelsDestroyDecoder", "WelsGetDecoderCapability", "WelsGetCodecVersion",
            "WelsGetCodecVersionEx"} <= names
    L = ctypes.CDLL(os.path.join(SO_DIR, "liblh264.so"))
    for n in sorted(names):
        assert hasattr(L, n), "missing export " + n


def test_layouts_match_committed_numbers(tmp_path):
    assert _probe(tmp_path, "lh264_isvc.h", INC) == LAYOUT_EXPECTED


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_API, "codec_api.h")), reason="reference tree not present")
def test_layouts_match_reference_header(tmp_path):
    assert _probe(tmp_path, "codec_api.h", REF_API) == LAYOUT_EXPECTED


C_VTABLE_CLIENT = r'''
#include <stdio.h>
#include <string.h>
#include "lh264_isvc.h"
int main(void) {
  ISVCDecoder* dec = NULL;               /* C view: a pointer to a pointer to the table of functions */
  if (WelsCreateDecoder (&dec) || !dec) return 1;
  int v = 0, lvl = 3; unsigned char* dst[3]; SBufferInfo info; SParserBsInfo pi; SDecoderCapability cap; OpenH264Version ver;
  memset (&info, 0, sizeof (info)); memset (&pi, 0, sizeof (pi));
  int a = 0, b = 0, c = 0, d = 0;
  printf ("%ld ", (*dec)->GetOption (dec, DECODER_OPTION_DATAFORMAT, &v));
  printf ("%d ", (int)(*dec)->DecodeFrame2 (dec, NULL, 0, dst, &info));
  printf ("%d ", (int)(*dec)->DecodeFrameNoDelay (dec, NULL, 0, dst, &info));
  printf ("%d ", (int)(*dec)->DecodeFrameEx (dec, NULL, 0, NULL, 0, &a, &b, &c, &d));
  printf ("%d ", (int)(*dec)->DecodeParser (dec, NULL, 0, &pi));
  printf ("%ld ", (*dec)->SetOption (dec, DECODER_OPTION_TRACE_LEVEL, &lvl));
  printf ("%ld ", (*dec)->SetOption (dec, DECODER_OPTION_DATAFORMAT, &v));
  printf ("%ld ", (*dec)->Initialize (dec, NULL));
  printf ("%ld ", (*dec)->Uninitialize (dec));
  WelsGetDecoderCapability (&cap); WelsGetCodecVersionEx (&ver);
  printf ("%d %d %u.%u.%u\n", cap.iProfileIdc, cap.iMaxFs, ver.uMajor, ver.uMinor, ver.uRevision);
  WelsDestroyDecoder (dec);
  return 0;
}
'''


def test_vtable_order_through_c_view(tmp_path):
    """every slot of the virtual table answers with the code only that method returns before Initialize()"""
    _build_lib()
    src = tmp_path / "c_client.c"
    src.write_text(C_VTABLE_CLIENT)
    exe = str(tmp_path / "c_client")
    subprocess.check_call(["gcc", "-std=c99", "-I", INC, str(src), "-o", exe, os.path.join(SO_DIR, "liblh264.so"),
                           "-Wl,--allow-shlib-undefined", "-Wl,-rpath," + SO_DIR, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.check_output([exe]).decode().split()
    #  GetOption->cmInitExpected(4); DecodeFrame2/NoDelay->dsInitialOptExpected; DecodeFrameEx->0; DecodeParser->0x2000;
    #  SetOption(trace) ok before init; SetOption(dataformat)->0x2000; Initialize(NULL)->cmInitParaError; Uninitialize->0
    assert out == ["4", "8192", "8192", "0", "8192", "0", "8192", "1", "0", "66", "5120", "1.4.1"]


def test_client_fails_loudly_without_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    _build_lib()
    exe = str(tmp_path / "isvc_client")
    _compile(os.path.join(ROOT, "tests", "isvc_client.cpp"), exe)
    stream = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "streams", "*")))[0]
    r = subprocess.run([exe, stream, str(tmp_path / "o.yuv")], capture_output=True)
    assert r.returncode == 3 and b"Initialize failed: 2" in r.stderr      # cmUnkonwReason: no CPU decode behind this object


def _sha_table():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "decoder_sha1.json")))


def _run_client(exe, stream, out, *args):
    r = subprocess.run([exe, stream, out] + list(args), capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    return hashlib.sha1(open(out, "rb").read()).hexdigest(), r.stdout.decode()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["frame2", "no-delay"])
def test_client_decodes_golden_streams(tmp_path, mode):
    exe = str(tmp_path / "isvc_client")
    _compile(os.path.join(ROOT, "tests", "isvc_client.cpp"), exe)
    sha = _sha_table()
    n = 0
    for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "streams", "*"))):
        name = os.path.basename(path)
        if name not in sha:
            continue
        got, log = _run_client(exe, path, str(tmp_path / "o.yuv"), *(["--no-delay"] if mode == "no-delay" else []))
        assert got == sha[name], (name, log)
        assert "state=0x0" in log and "version=1.4.1" in log
        n += 1
    assert n >= 4


@pytest.mark.gpu
def test_client_built_with_reference_header():
    """the same client compiled against the reference's own codec_api.h (oracle/Makefile, built where /root/reference exists)"""
    exe = os.path.join(ROOT, "oracle", "_ref", "isvc_client_refhdr")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/isvc_client_refhdr not built")
    sha = _sha_table()
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "streams", "*"))):
            name = os.path.basename(path)
            if name in sha:
                got, log = _run_client(exe, path, os.path.join(d, "o.yuv"))
                assert got == sha[name], (name, log)
