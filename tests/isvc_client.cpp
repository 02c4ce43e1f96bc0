// isvc_client.cpp - a small decoder application written against the ISVCDecoder interface, used by tests to drive
// liblh264.so exactly as an application drives the reference's decoder library: one NAL per DecodeFrame2 call, a final
// flush call with (NULL, 0), cropped I420 written per output picture (the calling pattern of the reference's console
// decoder, codec/console/dec/src/h264dec.cpp:244-350, and of its decoder test, test/api/BaseDecoderTest.cpp).
//
// Built twice: against include/lh264_isvc.h (tests/test_isvc.py, anywhere) and, with -DLH264_USE_REFERENCE_HEADER, against
// the reference's own codec_api.h by oracle/Makefile (only where /root/reference exists) - the second binary proves that a
// client compiled with the reference's declarations runs on this library without recompiling anything else.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#ifdef LH264_USE_REFERENCE_HEADER
#include "codec_api.h"
#else
#include "lh264_isvc.h"
#endif

static void write_plane (FILE* f, const unsigned char* p, int stride, int w, int h) {
  for (int y = 0; y < h; y++) fwrite (p + (size_t)y * stride, 1, (size_t)w, f);
}

static int emit (FILE* out, unsigned char** dst, const SBufferInfo& info) {
  if (info.iBufferStatus != 1) return 0;
  const int w = info.UsrData.sSystemBuffer.iWidth, h = info.UsrData.sSystemBuffer.iHeight;
  write_plane (out, dst[0], info.UsrData.sSystemBuffer.iStride[0], w, h);
  write_plane (out, dst[1], info.UsrData.sSystemBuffer.iStride[1], w / 2, h / 2);
  write_plane (out, dst[2], info.UsrData.sSystemBuffer.iStride[1], w / 2, h / 2);
  return 1;
}

int main (int argc, char** argv) {
  if (argc < 3) { fprintf (stderr, "usage: %s in.264 out.yuv [--no-delay]\n", argv[0]); return 2; }
  const bool no_delay = argc > 3 && !strcmp (argv[3], "--no-delay");
  FILE* in = fopen (argv[1], "rb");
  if (!in) { perror (argv[1]); return 2; }
  std::vector<unsigned char> bs;
  unsigned char tmp[65536]; size_t n;
  while ((n = fread (tmp, 1, sizeof (tmp), in)) > 0) bs.insert (bs.end(), tmp, tmp + n);
  fclose (in);
  FILE* out = fopen (argv[2], "wb");
  if (!out) { perror (argv[2]); return 2; }

  ISVCDecoder* dec = NULL;
  if (WelsCreateDecoder (&dec) || !dec) { fprintf (stderr, "WelsCreateDecoder failed\n"); return 1; }
  SDecodingParam param; memset (&param, 0, sizeof (param));
  param.eOutputColorFormat = videoFormatI420;
  param.uiTargetDqLayer = (unsigned char) - 1;
  param.eEcActiveIdc = ERROR_CON_SLICE_COPY;
  param.sVideoProperty.size = sizeof (param.sVideoProperty);
  param.sVideoProperty.eVideoBsType = VIDEO_BITSTREAM_DEFAULT;
  const long irc = dec->Initialize (&param);
  if (irc) { fprintf (stderr, "Initialize failed: %ld\n", irc); WelsDestroyDecoder (dec); return 3; }
  int fmt = (int)videoFormatI420;
  if (dec->SetOption (DECODER_OPTION_DATAFORMAT, &fmt)) { fprintf (stderr, "SetOption failed\n"); return 1; }

  int frames = 0, state_or = 0;
  size_t pos = 0;
  unsigned long long ts = 0;
  while (pos < bs.size()) {
    // next chunk: from this start code up to the next one (h264dec.cpp:255-272)
    size_t next = pos + 3;
    for (; next + 3 <= bs.size(); next++)
      if (bs[next] == 0 && bs[next + 1] == 0 && (bs[next + 2] == 1 || (next + 3 < bs.size() && bs[next + 2] == 0 && bs[next + 3] == 1))) break;
    if (next + 3 > bs.size()) next = bs.size();
    unsigned char* dst[3] = {NULL, NULL, NULL};
    SBufferInfo info; memset (&info, 0, sizeof (info));
    info.uiInBsTimeStamp = ++ts;
    const int st = no_delay ? (int)dec->DecodeFrameNoDelay (&bs[pos], (int) (next - pos), dst, &info)
                   : (int)dec->DecodeFrame2 (&bs[pos], (int) (next - pos), dst, &info);
    state_or |= st;
    frames += emit (out, dst, info);
    pos = next;
  }
  for (;;) {      // end of stream: drain
    int eos = 1;
    dec->SetOption (DECODER_OPTION_END_OF_STREAM, &eos);
    unsigned char* dst[3] = {NULL, NULL, NULL};
    SBufferInfo info; memset (&info, 0, sizeof (info));
    state_or |= (int)dec->DecodeFrame2 (NULL, 0, dst, &info);
    if (!emit (out, dst, info)) break;
    frames++;
  }
  int frame_num = -2; dec->GetOption (DECODER_OPTION_FRAME_NUM, &frame_num);
  SDecoderStatistics stats; memset (&stats, 0, sizeof (stats));
  dec->GetOption (DECODER_OPTION_GET_STATISTICS, &stats);
  OpenH264Version v = WelsGetCodecVersion();
  printf ("frames=%d state=0x%x last_frame_num=%d stats_frames=%u %ux%u version=%u.%u.%u\n", frames, state_or, frame_num,
          stats.uiDecodedFrameCount, stats.uiWidth, stats.uiHeight, v.uMajor, v.uMinor, v.uRevision);
  dec->Uninitialize();
  WelsDestroyDecoder (dec);
  fclose (out);
  return 0;
}
