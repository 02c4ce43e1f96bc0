// lh264dec.cpp - console application with the calling convention of the reference's (codec/console/dec/src/h264dec.cpp:150-178,
// 79-121): the file names decide the mode.
//
//   lh264dec in.264 out.pip [out.yuv]   compress: out.pip = default stream, out.pip.<tag> = tagged streams (GPU); the optional
//                                       YUV dump = cropped I420 pictures through the ISVCDecoder object of the same library
//   lh264dec in.pip out.264             restore the original bytes from in.pip + in.pip.<tag>                  (host)
//   lh264dec in.264 out.lhp             compress into ONE container file; restored and compared before it is written, a
//                                       stream the round trip cannot carry is stored verbatim
//   lh264dec in.lhp out.264             restore from the container
//   lh264dec --batch out_dir a.264 b.264 ...   many streams in one lh264_compress_batch call -> out_dir/<name>.lhp
//
// Written against include/lh264.h and include/lh264_isvc.h only; links liblh264.so.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "lh264.h"
#include "lh264_isvc.h"

typedef std::vector<uint8_t> Bytes;

static bool load (const std::string& p, Bytes& v) {
  FILE* f = fopen (p.c_str(), "rb");
  if (!f) return false;
  fseek (f, 0, SEEK_END); const long n = ftell (f); fseek (f, 0, SEEK_SET);
  v.resize (n > 0 ? (size_t)n : 0);
  const bool ok = n <= 0 || fread (v.data(), 1, (size_t)n, f) == (size_t)n;
  fclose (f);
  return ok;
}
static bool save (const std::string& p, const uint8_t* d, size_t n) {
  FILE* f = fopen (p.c_str(), "wb");
  if (!f) return false;
  const bool ok = n == 0 || fwrite (d, 1, n, f) == n;
  fclose (f);
  return ok;
}
static bool ends_with (const std::string& s, const char* e) { const size_t n = strlen (e); return s.size() >= n && s.compare (s.size() - n, n, e) == 0; }
static std::string base_name (const std::string& p) { const size_t k = p.find_last_of ('/'); return k == std::string::npos ? p : p.substr (k + 1); }

struct Compressed {
  lh264_compressed_t* h = nullptr;
  ~Compressed() { if (h) lh264_compressed_free (h); }
};

// the tagged streams of a compressed stream as the arrays lh264_pip_pack / lh264_pip_restore take
static void tag_arrays (const lh264_compressed_t* h, const uint8_t* tags[72], size_t len[72]) {
  for (int t = 0; t < 72; t++) tags[t] = lh264_compressed_tag (h, t, &len[t]);
}

static int dump_yuv (const Bytes& bs, const std::string& path) {
  FILE* out = fopen (path.c_str(), "wb");
  if (!out) { perror (path.c_str()); return 2; }
  ISVCDecoder* dec = NULL;
  if (WelsCreateDecoder (&dec) || !dec) { fprintf (stderr, "WelsCreateDecoder failed\n"); fclose (out); return 1; }
  SDecodingParam param; memset (&param, 0, sizeof (param));
  param.eOutputColorFormat = videoFormatI420; param.uiTargetDqLayer = (unsigned char) - 1; param.eEcActiveIdc = ERROR_CON_SLICE_COPY;
  param.sVideoProperty.size = sizeof (param.sVideoProperty); param.sVideoProperty.eVideoBsType = VIDEO_BITSTREAM_DEFAULT;
  if (dec->Initialize (&param)) { fprintf (stderr, "decoder Initialize failed (no GPU?)\n"); WelsDestroyDecoder (dec); fclose (out); return 3; }
  auto emit = [&] (unsigned char** dst, const SBufferInfo& info) {
    if (info.iBufferStatus != 1) return;
    const int w = info.UsrData.sSystemBuffer.iWidth, h = info.UsrData.sSystemBuffer.iHeight;
    for (int p = 0; p < 3; p++) {
      const int pw = p ? w / 2 : w, ph = p ? h / 2 : h, st = info.UsrData.sSystemBuffer.iStride[p ? 1 : 0];
      for (int y = 0; y < ph; y++) fwrite (dst[p] + (size_t)y * st, 1, (size_t)pw, out);
    }
  };
  size_t pos = 0;
  while (pos < bs.size()) {                     // one start-code-delimited chunk per call, as h264dec.cpp:246-272
    size_t i;
    for (i = 1; pos + i + 2 < bs.size(); i++)
      if (bs[pos + i] == 0 && bs[pos + i + 1] == 0 && (bs[pos + i + 2] == 1 || (pos + i + 3 < bs.size() && bs[pos + i + 2] == 0 && bs[pos + i + 3] == 1))) break;
    if (pos + i + 2 >= bs.size()) i = bs.size() - pos;
    unsigned char* dst[3] = {0, 0, 0}; SBufferInfo info; memset (&info, 0, sizeof (info));
    if (i >= 4) { dec->DecodeFrameNoDelay (bs.data() + pos, (int)i, dst, &info); emit (dst, info); }
    pos += i;
  }
  unsigned char* dst[3] = {0, 0, 0}; SBufferInfo info; memset (&info, 0, sizeof (info));
  dec->DecodeFrame2 (NULL, 0, dst, &info); emit (dst, info);
  dec->Uninitialize(); WelsDestroyDecoder (dec);
  fclose (out);
  return 0;
}

static int compress_files (const std::string& src, const std::string& dst, const char* yuv) {
  Bytes in;
  if (!load (src, in)) { perror (src.c_str()); return 2; }
  const uint8_t* d = in.data(); const size_t n = in.size();
  Compressed c;
  if (lh264_compress_batch (&d, &n, 1, 0, &c.h) != LH264_OK || lh264_compressed_status (c.h) != LH264_OK) {
    fprintf (stderr, "cannot compress %s: %s\n", src.c_str(), c.h ? lh264_compressed_error (c.h) : lh264_last_error());
    return 1;
  }
  size_t ml; const uint8_t* m = lh264_compressed_main (c.h, &ml);
  if (!save (dst, m, ml)) { perror (dst.c_str()); return 2; }
  size_t total = ml;
  for (int t = 0; t < 72; t++) {
    size_t tl; const uint8_t* p = lh264_compressed_tag (c.h, t, &tl);
    if (p) { if (!save (dst + "." + std::to_string (t), p, tl)) { perror (dst.c_str()); return 2; } total += tl; }
  }
  printf ("%s: %zu bytes -> %zu bytes (%.4f), %d pictures\n", src.c_str(), n, total, n ? (double)total / (double)n : 0.0, lh264_compressed_pictures (c.h));
  return yuv ? dump_yuv (in, yuv) : 0;
}

static int restore_files (const std::string& src, const std::string& dst) {
  Bytes m;
  if (!load (src, m)) { perror (src.c_str()); return 2; }
  std::vector<Bytes> tb (72); const uint8_t* tags[72]; size_t len[72]; int nt = 0;
  for (int t = 0; t < 72; t++) { const bool h = load (src + "." + std::to_string (t), tb[t]); tags[t] = h ? (tb[t].empty() ? (const uint8_t*)"" : tb[t].data()) : nullptr; len[t] = tb[t].size(); nt += h; }
  size_t need = 0;
  Bytes out (64);
  int rc = lh264_pip_restore (m.data(), m.size(), tags, len, 72, out.data(), out.size(), &need);
  if (rc == LH264_E_ARG && need > out.size()) { out.resize (need); rc = lh264_pip_restore (m.data(), m.size(), tags, len, 72, out.data(), out.size(), &need); }
  if (rc != LH264_OK) { fprintf (stderr, "cannot restore %s: %s\n", src.c_str(), lh264_restore_error()); return 1; }
  if (!save (dst, out.data(), need)) { perror (dst.c_str()); return 2; }
  printf ("%s (+%d tagged streams) -> %s: %zu bytes\n", src.c_str(), nt, dst.c_str(), need);
  return 0;
}

// container for one already compressed (or failed) stream: verified, verbatim when the round trip does not hold or does not pay
static bool make_container (const Bytes& in, const lh264_compressed_t* h, Bytes& blob, std::string& why) {
  why.clear();
  if (h && lh264_compressed_status (h) == LH264_OK && lh264_compressed_pictures (h) > 0) {
    size_t ml; const uint8_t* m = lh264_compressed_main (h, &ml);
    const uint8_t* tags[72]; size_t len[72];
    tag_arrays (h, tags, len);
    Bytes back (in.size() + 64); size_t bl = 0;
    if (lh264_pip_restore (m, ml, tags, len, 72, back.data(), back.size(), &bl) == LH264_OK && bl == in.size() && memcmp (back.data(), in.data(), bl) == 0) {
      blob.resize (lh264_pip_pack_bound (ml, len, 72));
      size_t n = 0;
      if (lh264_pip_pack (m, ml, tags, len, 72, 0, blob.data(), blob.size(), &n) == LH264_OK) { blob.resize (n); if (n < in.size() + 32) return true; why = "no gain"; }
    } else why = "the restored stream differs";
  } else why = h ? lh264_compressed_error (h) : "not compressed";
  if (why.empty()) why = "no picture";
  blob.resize (in.size() + 64);
  size_t n = 0;
  if (lh264_pip_pack (in.data(), in.size(), nullptr, nullptr, 0, LH264_PIP_VERBATIM, blob.data(), blob.size(), &n) != LH264_OK) return false;
  blob.resize (n);
  return true;
}

static int compress_single (const std::vector<std::string>& srcs, const std::vector<std::string>& dsts) {
  const int n = (int)srcs.size();
  std::vector<Bytes> in (n);
  std::vector<const uint8_t*> d (n); std::vector<size_t> l (n);
  for (int i = 0; i < n; i++) { if (!load (srcs[i], in[i])) { perror (srcs[i].c_str()); return 2; } d[i] = in[i].data(); l[i] = in[i].size(); }
  std::vector<lh264_compressed_t*> h (n, nullptr);
  // every device of the node takes a share of the batch
  const int nd = lh264_device_count();
  std::vector<int> devs;
  for (int i = 0; i < nd && i < n; i++) devs.push_back (i);
  const int rc = devs.size() > 1 ? lh264_compress_batch_devices (d.data(), l.data(), n, 0, devs.data(), (int)devs.size(), h.data())
                                 : lh264_compress_batch (d.data(), l.data(), n, 0, h.data());
  int ret = 0;
  for (int i = 0; i < n; i++) {
    Bytes blob; std::string why;
    if (!make_container (in[i], rc == LH264_OK ? h[i] : nullptr, blob, why) || !save (dsts[i], blob.data(), blob.size())) { fprintf (stderr, "cannot write %s\n", dsts[i].c_str()); ret = 2; }
    else printf ("%s: %zu bytes -> %zu bytes (%.4f)%s%s%s\n", srcs[i].c_str(), in[i].size(), blob.size(), in[i].empty() ? 0.0 : (double)blob.size() / (double)in[i].size(),
                 why.empty() ? "" : "  [verbatim: ", why.c_str(), why.empty() ? "" : "]");
    if (h[i]) lh264_compressed_free (h[i]);
  }
  return ret;
}

static int restore_single (const std::string& src, const std::string& dst) {
  Bytes f;
  if (!load (src, f)) { perror (src.c_str()); return 2; }
  size_t need = 0;
  Bytes out (64);
  int rc = lh264_pip_restore_file (f.data(), f.size(), out.data(), out.size(), &need);
  if (rc == LH264_E_ARG && need > out.size()) { out.resize (need); rc = lh264_pip_restore_file (f.data(), f.size(), out.data(), out.size(), &need); }
  if (rc != LH264_OK) { fprintf (stderr, "cannot restore %s: %s\n", src.c_str(), lh264_restore_error()); return 1; }
  if (!save (dst, out.data(), need)) { perror (dst.c_str()); return 2; }
  printf ("%s -> %s: %zu bytes\n", src.c_str(), dst.c_str(), need);
  return 0;
}

int main (int argc, char** argv) {
  if (argc >= 4 && !strcmp (argv[1], "--batch")) {
    std::vector<std::string> srcs, dsts;
    for (int i = 3; i < argc; i++) { srcs.push_back (argv[i]); dsts.push_back (std::string (argv[2]) + "/" + base_name (argv[i]) + ".lhp"); }
    return compress_single (srcs, dsts);
  }
  if (argc < 3) {
    fprintf (stderr, "usage: %s in.264 out.pip [out.yuv] | in.pip out.264 | in.264 out.lhp | in.lhp out.264 | --batch out_dir in.264...\n", argv[0]);
    return 2;
  }
  const std::string a = argv[1], b = argv[2];
  if (ends_with (a, ".lhp")) return restore_single (a, b);
  if (ends_with (b, ".lhp")) return compress_single ({a}, {b});
  if (base_name (a).find (".pip") != std::string::npos) return restore_files (a, b);      // as the reference decides (h264dec.cpp:167-173)
  return compress_files (a, b, argc > 3 ? argv[3] : nullptr);
}
