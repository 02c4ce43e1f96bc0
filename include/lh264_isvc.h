/* lh264_isvc.h - the coarse drop-in boundary: an ISVCDecoder-compatible decoder object whose reconstruct path runs on
 * the GPU (host front end: losslessh264_amd/csrc/host/h264_parser.cpp; reconstruction: lh264_recon_chains).
 *
 * Binary compatibility is with the reference's public decoder interface, codec/api/svc/codec_api.h:345-573
 * (class ISVCDecoder / struct ISVCDecoderVtbl, WelsCreateDecoder .. WelsGetCodecVersionEx) and the plain-data types of
 * codec/api/svc/codec_app_def.h and codec_def.h that cross it.  An application compiled against the reference's own
 * codec_api.h can link liblh264.so instead of libopenh264/libdecoder: the exported symbol names, the virtual-table
 * order and every structure layout below are the reference's.  This header exists so that this repository's tests (and
 * applications without the reference tree) have the declarations; it declares only the decoder half of the API.
 *
 * Error behaviour follows codec_app_def.h:77-100 (DECODING_STATE bit mask) and codec_def.h:80-87 (CM_RETURN).  There is
 * no CPU reconstruct path behind this object: Initialize() fails with cmUnkonwReason when no GPU is visible. */
#ifndef LH264_ISVC_H_
#define LH264_ISVC_H_

#ifdef WELS_VIDEO_CODEC_SVC_API_H__
#error "include either the reference's codec_api.h or lh264_isvc.h, not both"
#endif

#ifndef __cplusplus
#include <stdbool.h>
#endif
#ifdef __cplusplus
extern "C" {
#endif

/* codec_app_def.h:67-72 */
typedef struct _tagVersion { unsigned int uMajor, uMinor, uRevision, uReserved; } OpenH264Version;

/* codec_app_def.h:77-100 */
typedef enum {
  dsErrorFree = 0x00, dsFramePending = 0x01, dsRefLost = 0x02, dsBitstreamError = 0x04, dsDepLayerLost = 0x08,
  dsNoParamSets = 0x10, dsDataErrorConcealed = 0x20,
  dsInvalidArgument = 0x1000, dsInitialOptExpected = 0x2000, dsOutOfMemory = 0x4000, dsDstBufNeedExpan = 0x8000
} DECODING_STATE;

/* codec_app_def.h:150-166 */
typedef enum {
  DECODER_OPTION_DATAFORMAT = 0, DECODER_OPTION_END_OF_STREAM, DECODER_OPTION_VCL_NAL, DECODER_OPTION_TEMPORAL_ID,
  DECODER_OPTION_FRAME_NUM, DECODER_OPTION_IDR_PIC_ID, DECODER_OPTION_LTR_MARKING_FLAG,
  DECODER_OPTION_LTR_MARKED_FRAME_NUM, DECODER_OPTION_ERROR_CON_IDC, DECODER_OPTION_TRACE_LEVEL,
  DECODER_OPTION_TRACE_CALLBACK, DECODER_OPTION_TRACE_CALLBACK_CONTEXT, DECODER_OPTION_GET_STATISTICS
} DECODER_OPTION;

/* codec_app_def.h:171-180 */
typedef enum {
  ERROR_CON_DISABLE = 0, ERROR_CON_FRAME_COPY, ERROR_CON_SLICE_COPY, ERROR_CON_FRAME_COPY_CROSS_IDR,
  ERROR_CON_SLICE_COPY_CROSS_IDR, ERROR_CON_SLICE_COPY_CROSS_IDR_FREEZE_RES_CHANGE, ERROR_CON_SLICE_MV_COPY_CROSS_IDR,
  ERROR_CON_SLICE_MV_COPY_CROSS_IDR_FREEZE_RES_CHANGE
} ERROR_CON_IDC;

/* codec_app_def.h:184-188 */
typedef enum { FEEDBACK_NON_VCL_NAL = 0, FEEDBACK_VCL_NAL, FEEDBACK_UNKNOWN_NAL } FEEDBACK_VCL_NAL_IN_AU;

/* codec_app_def.h:212-216 */
typedef enum { VIDEO_BITSTREAM_AVC = 0, VIDEO_BITSTREAM_SVC = 1, VIDEO_BITSTREAM_DEFAULT = VIDEO_BITSTREAM_SVC } VIDEO_BITSTREAM_TYPE;

/* codec_def.h:43-63 (only the values a decoder reports) */
typedef enum { videoFormatI420 = 23, videoFormatYV12 = 24, videoFormatInternal = 25, videoFormatNV12 = 26,
               videoFormatVFlip = 0x80000000 } EVideoFormatType;

/* codec_def.h:80-87 */
typedef enum { cmResultSuccess, cmInitParaError, cmUnkonwReason, cmMallocMemeError, cmInitExpected, cmUnsupportedData } CM_RETURN;

/* codec_app_def.h:477-496 */
typedef struct { unsigned int size; VIDEO_BITSTREAM_TYPE eVideoBsType; } SVideoProperty;
typedef struct TagSVCDecodingParam {
  char* pFileNameRestructed;
  EVideoFormatType eOutputColorFormat;
  unsigned int uiCpuLoad;
  unsigned char uiTargetDqLayer;
  ERROR_CON_IDC eEcActiveIdc;
  bool bParseOnly;
  SVideoProperty sVideoProperty;
} SDecodingParam, *PDecodingParam;

/* codec_def.h:187-204 */
typedef struct TagSysMemBuffer { int iWidth, iHeight, iFormat, iStride[2]; } SSysMEMBuffer;
typedef struct TagBufferInfo {
  int iBufferStatus;                       /* 1: ppDst[] hold a picture */
  unsigned long long uiInBsTimeStamp, uiOutYuvTimeStamp;
  union { SSysMEMBuffer sSystemBuffer; } UsrData;
} SBufferInfo;

/* codec_app_def.h:590-613 */
typedef struct TagDecoderCapability {
  int iProfileIdc, iProfileIop, iLevelIdc, iMaxMbps, iMaxFs, iMaxCpb, iMaxDpb, iMaxBr;
  bool bRedPicCap;
} SDecoderCapability;
#define MAX_NAL_UNITS_IN_LAYER 128         /* codec_app_def.h:49 */
typedef struct TagParserBsInfo {
  int iNalNum;
  int iNalLenInByte[MAX_NAL_UNITS_IN_LAYER];
  unsigned char* pDstBuff;
  int iSpsWidthInPixel, iSpsHeightInPixel;
  unsigned long long uiInBsTimeStamp, uiOutBsTimeStamp;
} SParserBsInfo, *PParserBsInfo;

/* codec_app_def.h:644-669 */
typedef struct TagVideoDecoderStatistics {
  unsigned int uiWidth, uiHeight;
  float fAverageFrameSpeedInMs, fActualAverageFrameSpeedInMs;
  unsigned int uiDecodedFrameCount, uiResolutionChangeTimes, uiIDRCorrectNum, uiAvgEcRatio, uiAvgEcPropRatio, uiEcIDRNum,
           uiEcFrameNum, uiIDRLostNum, uiFreezingIDRNum, uiFreezingNonIDRNum;
  int iAvgLumaQp, iSpsReportErrorNum, iSubSpsReportErrorNum, iPpsReportErrorNum, iSpsNoExistNalNum, iSubSpsNoExistNalNum,
      iPpsNoExistNalNum;
} SDecoderStatistics;

typedef void (*WelsTraceCallback) (void* ctx, int level, const char* string);

#ifdef __cplusplus
}   /* extern "C" */

/* codec_api.h:345-421: same virtual functions in the same order (the virtual destructor last) */
class ISVCDecoder {
 public:
  virtual long Initialize (const SDecodingParam* pParam) = 0;
  virtual long Uninitialize() = 0;
  virtual DECODING_STATE DecodeFrame (const unsigned char* pSrc, const int iSrcLen, unsigned char** ppDst, int* pStride,
                                      int& iWidth, int& iHeight) = 0;
  virtual DECODING_STATE DecodeFrameNoDelay (const unsigned char* pSrc, const int iSrcLen, unsigned char** ppDst,
      SBufferInfo* pDstInfo) = 0;
  virtual DECODING_STATE DecodeFrame2 (const unsigned char* pSrc, const int iSrcLen, unsigned char** ppDst,
                                       SBufferInfo* pDstInfo) = 0;
  virtual DECODING_STATE DecodeParser (const unsigned char* pSrc, const int iSrcLen, SParserBsInfo* pDstInfo) = 0;
  virtual DECODING_STATE DecodeFrameEx (const unsigned char* pSrc, const int iSrcLen, unsigned char* pDst, int iDstStride,
                                        int& iDstLen, int& iWidth, int& iHeight, int& iColorFormat) = 0;
  virtual long SetOption (DECODER_OPTION eOptionId, void* pOption) = 0;
  virtual long GetOption (DECODER_OPTION eOptionId, void* pOption) = 0;
  virtual ~ISVCDecoder() {}
};
extern "C" {
#else
/* codec_api.h:500-540: the C view of the same object */
typedef struct ISVCDecoderVtbl ISVCDecoderVtbl;
typedef const ISVCDecoderVtbl* ISVCDecoder;
struct ISVCDecoderVtbl {
  long (*Initialize) (ISVCDecoder*, const SDecodingParam* pParam);
  long (*Uninitialize) (ISVCDecoder*);
  DECODING_STATE (*DecodeFrame) (ISVCDecoder*, const unsigned char* pSrc, const int iSrcLen, unsigned char** ppDst,
                                 int* pStride, int* iWidth, int* iHeight);
  DECODING_STATE (*DecodeFrameNoDelay) (ISVCDecoder*, const unsigned char* pSrc, const int iSrcLen, unsigned char** ppDst,
                                        SBufferInfo* pDstInfo);
  DECODING_STATE (*DecodeFrame2) (ISVCDecoder*, const unsigned char* pSrc, const int iSrcLen, unsigned char** ppDst,
                                  SBufferInfo* pDstInfo);
  DECODING_STATE (*DecodeParser) (ISVCDecoder*, const unsigned char* pSrc, const int iSrcLen, SParserBsInfo* pDstInfo);
  DECODING_STATE (*DecodeFrameEx) (ISVCDecoder*, const unsigned char* pSrc, const int iSrcLen, unsigned char* pDst,
                                   int iDstStride, int* iDstLen, int* iWidth, int* iHeight, int* iColorFormat);
  long (*SetOption) (ISVCDecoder*, DECODER_OPTION eOptionId, void* pOption);
  long (*GetOption) (ISVCDecoder*, DECODER_OPTION eOptionId, void* pOption);
};
#endif

/* codec_api.h:548-571 */
int  WelsGetDecoderCapability (SDecoderCapability* pDecCapability);
long WelsCreateDecoder (ISVCDecoder** ppDecoder);
void WelsDestroyDecoder (ISVCDecoder* pDecoder);
OpenH264Version WelsGetCodecVersion (void);
void WelsGetCodecVersionEx (OpenH264Version* pVersion);

/* ---- flat C entry points over the same object, for FFI hosts that cannot call through a C++ vtable (ctypes, cgo, JNI).
 * Thin forwards to the virtual functions above; not part of the reference API. */
long lh264_isvc_initialize (ISVCDecoder* dec, const SDecodingParam* param);
long lh264_isvc_uninitialize (ISVCDecoder* dec);
int  lh264_isvc_decode_frame2 (ISVCDecoder* dec, const unsigned char* src, int len, unsigned char** dst3, SBufferInfo* info);
int  lh264_isvc_decode_frame_no_delay (ISVCDecoder* dec, const unsigned char* src, int len, unsigned char** dst3, SBufferInfo* info);
long lh264_isvc_set_option (ISVCDecoder* dec, int option, void* value);
long lh264_isvc_get_option (ISVCDecoder* dec, int option, void* value);

#ifdef __cplusplus
}
#endif
#endif /* LH264_ISVC_H_ */
