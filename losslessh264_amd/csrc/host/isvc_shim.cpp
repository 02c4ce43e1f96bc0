// isvc_shim.cpp - the coarse boundary of include/lh264_isvc.h: an object binary-compatible with the reference's
// ISVCDecoder (codec/api/svc/codec_api.h:345-421; the reference's implementation is CWelsDecoder,
// codec/decoder/plus/src/welsDecoderExt.cpp:186-640).  The bitstream is parsed on the host by lh264host::Parser into the
// flat records of include/lh264.h; every completed access unit is reconstructed on the GPU with lh264_recon_chains (a
// chain of one frame) and the padded picture is copied back for the caller, as DecodeFrameConstruction
// (codec/decoder/core/src/decoder_core.cpp:56-200) hands out pointers into the reference's own padded picture.
//
// One access unit per launch is the latency-bound way to use the GPU; the throughput route is the batched C ABI
// (lh264_recon_chains over many streams).  This object exists so that code written against the reference's decoder keeps
// working unchanged.  There is no CPU reconstruct path: without a GPU Initialize() fails.
#include <stdlib.h>
#include <string.h>
#include <deque>
#include <map>
#include <memory>
#include <new>
#include <string>
#include <vector>
#include "../../../include/lh264.h"
#include "../../../include/lh264_isvc.h"
#include "h264_parser.h"

namespace {

using lh264host::FrameOut;
using lh264host::Parser;

struct DevPic {
  uint8_t* base = nullptr;
  size_t bytes = 0, off[3] = {0, 0, 0};
  int mb_w = 0, mb_h = 0, stride_y = 0, stride_c = 0;
};

class GpuDecoder : public ISVCDecoder {
 public:
  GpuDecoder() { memset (&stats_, 0, sizeof (stats_)); stats_.iAvgLumaQp = -1; }
  ~GpuDecoder() override { Uninitialize(); }

  long Initialize (const SDecodingParam* p) override {
    if (!p) return cmInitParaError;
    Uninitialize();
    if (p->bParseOnly) return cmUnsupportedData;           // DecodeParser's bitstream rewriting is not provided
    if (lh264_device_count() <= 0) return cmUnkonwReason;   // no GPU: fail loudly, never decode on the CPU
    param_ = *p; param_.pFileNameRestructed = nullptr;
    ec_idc_ = (int)p->eEcActiveIdc;
    parser_.reset (new Parser());
    d_chain_first_ = (int32_t*)lh264_dev_malloc (2 * sizeof (int32_t));
    d_job_ = (lh264_frame_job_t*)lh264_dev_malloc (sizeof (lh264_frame_job_t));
    if (!d_chain_first_ || !d_job_) { Uninitialize(); return cmMallocMemeError; }
    const int32_t cf[2] = {0, 1};
    if (lh264_memcpy_h2d (d_chain_first_, cf, sizeof (cf), nullptr) || lh264_stream_sync (nullptr)) { Uninitialize(); return cmUnkonwReason; }
    inited_ = true;
    return cmResultSuccess;
  }

  long Uninitialize() override {
    for (auto& kv : pics_) lh264_dev_free (kv.second.base);
    pics_.clear();
    for (auto* b : free_pics_) { lh264_dev_free (b->base); delete b; }
    free_pics_.clear();
    if (d_chain_first_) lh264_dev_free (d_chain_first_);
    if (d_job_) lh264_dev_free (d_job_);
    if (d_mbs_) lh264_dev_free (d_mbs_);
    if (d_coeffs_) lh264_dev_free (d_coeffs_);
    if (d_slices_) lh264_dev_free (d_slices_);
    d_chain_first_ = nullptr; d_job_ = nullptr; d_mbs_ = nullptr; d_coeffs_ = nullptr; d_slices_ = nullptr;
    cap_mbs_ = cap_slices_ = 0;
    parser_.reset(); pending_.clear(); host_out_.clear();
    inited_ = false; eos_ = false;
    return 0;
  }

  DECODING_STATE DecodeFrame (const unsigned char* src, const int len, unsigned char** dst, int* stride, int& w, int& h) override {
    SBufferInfo info; memset (&info, 0, sizeof (info));
    DECODING_STATE st = DecodeFrame2 (src, len, dst, &info);
    if (st == dsErrorFree) {
      stride[0] = info.UsrData.sSystemBuffer.iStride[0]; stride[1] = info.UsrData.sSystemBuffer.iStride[1];
      w = info.UsrData.sSystemBuffer.iWidth; h = info.UsrData.sSystemBuffer.iHeight;
    }
    return st;
  }

  DECODING_STATE DecodeFrameNoDelay (const unsigned char* src, const int len, unsigned char** dst, SBufferInfo* info) override {
    // welsDecoderExt.cpp:408-430: decode, then flush with (NULL, 0) so the picture comes out with its last slice instead
    // of with the first slice of the next access unit.  A picture whose slices arrive in separate calls is flushed once
    // it is whole (the reference would conceal the missing part; this decoder waits for it).
    int rc = (int)DecodeFrame2 (src, len, dst, info);
    if (!inited_ || !info || info->iBufferStatus == 1) return (DECODING_STATE)rc;
    if (src && len > 0 && pending_.empty() && !parser_->picture_in_progress_is_whole()) return (DECODING_STATE)rc;
    rc |= (int)DecodeFrame2 (nullptr, 0, dst, info);
    return (DECODING_STATE)rc;
  }

  DECODING_STATE DecodeFrame2 (const unsigned char* src, const int len, unsigned char** dst, SBufferInfo* info) override {
    if (!inited_) return dsInitialOptExpected;
    if (!dst || !info) return dsInvalidArgument;
    dst[0] = dst[1] = dst[2] = nullptr;
    const unsigned long long ts = info->uiInBsTimeStamp;
    memset (info, 0, sizeof (*info));
    info->uiInBsTimeStamp = ts;
    int state = dsErrorFree;
    vcl_in_au_ = FEEDBACK_UNKNOWN_NAL;
    if (src && len > 0) {
      eos_ = false;
      for (int i = 0; i + 3 < len; i++)
        if (src[i] == 0 && src[i + 1] == 0 && src[i + 2] == 1) {
          const int t = src[i + 3] & 31;
          if (t == 1 || t == 5) vcl_in_au_ = FEEDBACK_VCL_NAL; else if (vcl_in_au_ != FEEDBACK_VCL_NAL) vcl_in_au_ = FEEDBACK_NON_VCL_NAL;
        }
      if (parser_->feed (src, (size_t)len) < 0) {
        state |= parser_->error().find ("missing") != std::string::npos ? dsNoParamSets : dsBitstreamError;
        trace (1, parser_->error().c_str());
        parser_->clear_error();
      }
    } else {
      eos_ = true;
      parser_->flush();
    }
    auto& fr = parser_->frames();
    for (auto& f : fr) pending_.push_back ({std::move (f), ts});
    fr.clear();
    if (!pending_.empty()) {
      std::unique_ptr<FrameOut> f = std::move (pending_.front().frame);
      const unsigned long long fts = pending_.front().ts;
      pending_.pop_front();
      state |= reconstruct_and_output (*f, dst, info, fts);
    }
    return (DECODING_STATE)state;
  }

  DECODING_STATE DecodeParser (const unsigned char*, const int, SParserBsInfo*) override {
    return inited_ ? dsInvalidArgument : dsInitialOptExpected;   // parse-only mode (bParseOnly) is not provided
  }

  DECODING_STATE DecodeFrameEx (const unsigned char*, const int, unsigned char*, int, int&, int&, int&, int&) override {
    return dsErrorFree;                                           // a no-op in the reference too (welsDecoderExt.cpp:641-652)
  }

  long SetOption (DECODER_OPTION id, void* v) override {
    if (!inited_ && id != DECODER_OPTION_TRACE_LEVEL && id != DECODER_OPTION_TRACE_CALLBACK && id != DECODER_OPTION_TRACE_CALLBACK_CONTEXT)
      return dsInitialOptExpected;
    switch (id) {
    case DECODER_OPTION_DATAFORMAT:
      if (!v) return cmInitParaError;
      return * (int*)v == (int)videoFormatI420 ? cmResultSuccess : cmInitParaError;
    case DECODER_OPTION_END_OF_STREAM:
      if (!v) return cmInitParaError;
      eos_ = * (int*)v != 0;
      return cmResultSuccess;
    case DECODER_OPTION_ERROR_CON_IDC:
      if (!v) return cmInitParaError;
      ec_idc_ = * (int*)v;
      return cmResultSuccess;
    case DECODER_OPTION_TRACE_LEVEL: if (v) trace_level_ = * (int*)v; return cmResultSuccess;
    case DECODER_OPTION_TRACE_CALLBACK: if (v) trace_cb_ = * (WelsTraceCallback*)v; return cmResultSuccess;
    case DECODER_OPTION_TRACE_CALLBACK_CONTEXT: if (v) trace_ctx_ = * (void**)v; return cmResultSuccess;
    default: return cmInitParaError;
    }
  }

  long GetOption (DECODER_OPTION id, void* v) override {
    if (!inited_) return cmInitExpected;
    if (!v) return cmInitParaError;
    switch (id) {
    case DECODER_OPTION_DATAFORMAT: * (int*)v = (int)videoFormatI420; return cmResultSuccess;
    case DECODER_OPTION_END_OF_STREAM: * (int*)v = eos_; return cmResultSuccess;
    case DECODER_OPTION_IDR_PIC_ID: * (int*)v = last_idr_pic_id_; return cmResultSuccess;
    case DECODER_OPTION_FRAME_NUM: * (int*)v = last_frame_num_; return cmResultSuccess;
    case DECODER_OPTION_LTR_MARKING_FLAG: * (int*)v = 0; return cmResultSuccess;
    case DECODER_OPTION_LTR_MARKED_FRAME_NUM: * (int*)v = 0; return cmResultSuccess;
    case DECODER_OPTION_VCL_NAL: * (int*)v = vcl_in_au_; return cmResultSuccess;
    case DECODER_OPTION_TEMPORAL_ID: * (int*)v = vcl_in_au_ == FEEDBACK_VCL_NAL ? 0 : -1; return cmResultSuccess;
    case DECODER_OPTION_ERROR_CON_IDC: * (int*)v = ec_idc_; return cmResultSuccess;
    case DECODER_OPTION_GET_STATISTICS: memcpy (v, &stats_, sizeof (stats_)); return cmResultSuccess;
    default: return cmInitParaError;
    }
  }

 private:
  struct Pending { std::unique_ptr<FrameOut> frame; unsigned long long ts; };

  void trace (int level, const char* msg) { if (trace_cb_ && level <= trace_level_) trace_cb_ (trace_ctx_, level, msg); }

  DevPic* acquire_pic (int mb_w, int mb_h) {
    for (size_t i = 0; i < free_pics_.size(); i++)
      if (free_pics_[i]->mb_w == mb_w && free_pics_[i]->mb_h == mb_h) { DevPic* p = free_pics_[i]; free_pics_.erase (free_pics_.begin() + i); return p; }
    for (auto* b : free_pics_) { lh264_dev_free (b->base); delete b; }      // resolution change: drop the old pool
    free_pics_.clear();
    std::unique_ptr<DevPic> p (new DevPic());
    p->mb_w = mb_w; p->mb_h = mb_h;
    p->bytes = lh264_pic_bytes (mb_w, mb_h, &p->stride_y, &p->stride_c, &p->off[0], &p->off[1], &p->off[2]);
    p->base = (uint8_t*)lh264_dev_malloc (p->bytes);
    if (!p->base) return nullptr;
    return p.release();
  }

  int ensure_staging (size_t n_mbs, size_t n_slices) {
    if (n_mbs > cap_mbs_) {
      if (d_mbs_) lh264_dev_free (d_mbs_);
      if (d_coeffs_) lh264_dev_free (d_coeffs_);
      d_mbs_ = (lh264_mb_t*)lh264_dev_malloc (n_mbs * sizeof (lh264_mb_t));
      d_coeffs_ = (int16_t*)lh264_dev_malloc (n_mbs * 384 * sizeof (int16_t));
      cap_mbs_ = (d_mbs_ && d_coeffs_) ? n_mbs : 0;
      if (!cap_mbs_) return -1;
    }
    if (n_slices > cap_slices_) {
      if (d_slices_) lh264_dev_free (d_slices_);
      d_slices_ = (lh264_slice_t*)lh264_dev_malloc (n_slices * sizeof (lh264_slice_t));
      cap_slices_ = d_slices_ ? n_slices : 0;
      if (!cap_slices_) return -1;
    }
    return 0;
  }

  int reconstruct_and_output (FrameOut& f, unsigned char** dst, SBufferInfo* info, unsigned long long ts) {
    const size_t n = (size_t)f.mb_w * f.mb_h;
    last_frame_num_ = f.frame_num; if (f.idr) last_idr_pic_id_ = f.idr_pic_id;
    size_t covered = 0;
    for (size_t k = 0; k < n; k++) covered += f.covered[k] != 0;
    if (covered != n || f.slices.empty()) {
      // the reference would conceal the missing macroblocks (error_concealment.cpp); this decoder reports the loss instead
      trace (2, "lh264: access unit incomplete, no picture produced");
      release_unreferenced (f, nullptr);
      return dsBitstreamError;
    }
    DevPic* pic = acquire_pic (f.mb_w, f.mb_h);
    if (!pic || ensure_staging (n, f.slices.size())) { if (pic) free_pics_.push_back (pic); return dsOutOfMemory; }
    const DevPic P = *pic;                 // geometry survives handing `pic` to the reference map below
    lh264_frame_job_t job; memset (&job, 0, sizeof (job));
    job.mbs_dev = d_mbs_; job.coeffs_dev = d_coeffs_; job.slices_dev = d_slices_;
    job.dst.y_dev = pic->base + pic->off[0]; job.dst.u_dev = pic->base + pic->off[1]; job.dst.v_dev = pic->base + pic->off[2];
    int state = dsErrorFree;
    for (size_t i = 0; i < LH264_MAX_REFS; i++) {
      const DevPic* r = pic;
      if (i < f.ref_ids.size()) {
        auto it = pics_.find (f.ref_ids[i]);
        if (it != pics_.end() && it->second.mb_w == f.mb_w && it->second.mb_h == f.mb_h) r = &it->second;
        else state |= dsRefLost;
      }
      job.ref[i].y_dev = r->base + r->off[0]; job.ref[i].u_dev = r->base + r->off[1]; job.ref[i].v_dev = r->base + r->off[2];
    }
    job.mb_w = f.mb_w; job.mb_h = f.mb_h; job.stride_y = pic->stride_y; job.stride_c = pic->stride_c;
    job.n_slices = (int32_t)f.slices.size();
    job.flags = f.is_ref ? 0 : LH264_JOB_NO_EXPAND;
    host_out_.resize (pic->bytes);
    int rc = lh264_memcpy_h2d (d_mbs_, f.mbs.data(), n * sizeof (lh264_mb_t), nullptr);
    rc = rc ? rc : lh264_memcpy_h2d (d_coeffs_, f.coeffs.data(), n * 384 * sizeof (int16_t), nullptr);
    rc = rc ? rc : lh264_memcpy_h2d (d_slices_, f.slices.data(), f.slices.size() * sizeof (lh264_slice_t), nullptr);
    rc = rc ? rc : lh264_memcpy_h2d (d_job_, &job, sizeof (job), nullptr);
    rc = rc ? rc : lh264_recon_chains (d_job_, d_chain_first_, 1, f.mb_w, f.mb_h, nullptr);
    rc = rc ? rc : lh264_memcpy_d2h (host_out_.data(), pic->base, pic->bytes, nullptr);
    rc = rc ? rc : lh264_stream_sync (nullptr);
    if (rc) { trace (1, lh264_last_error()); free_pics_.push_back (pic); return dsOutOfMemory; }
    release_unreferenced (f, pic);
    pic = nullptr;
    // DecodeFrameConstruction (decoder_core.cpp:150-200): pointers at the cropped origin inside the padded picture
    dst[0] = host_out_.data() + P.off[0] + (size_t)f.crop_y * P.stride_y + f.crop_x;
    dst[1] = host_out_.data() + P.off[1] + (size_t) (f.crop_y >> 1) * P.stride_c + (f.crop_x >> 1);
    dst[2] = host_out_.data() + P.off[2] + (size_t) (f.crop_y >> 1) * P.stride_c + (f.crop_x >> 1);
    info->iBufferStatus = 1;
    info->uiOutYuvTimeStamp = ts;
    info->UsrData.sSystemBuffer.iFormat = (int)videoFormatI420;
    info->UsrData.sSystemBuffer.iWidth = f.crop_w; info->UsrData.sSystemBuffer.iHeight = f.crop_h;
    info->UsrData.sSystemBuffer.iStride[0] = P.stride_y; info->UsrData.sSystemBuffer.iStride[1] = P.stride_c;
    if (stats_.uiWidth != (unsigned)f.crop_w || stats_.uiHeight != (unsigned)f.crop_h) {
      stats_.uiResolutionChangeTimes++; stats_.uiWidth = f.crop_w; stats_.uiHeight = f.crop_h;
    }
    stats_.uiDecodedFrameCount++;
    if (f.idr) stats_.uiIDRCorrectNum++;
    return state;
  }

  // keep exactly the pictures the parser's DPB still marks as used for reference (plus `cur` when it is one)
  void release_unreferenced (const FrameOut& f, DevPic* cur) {
    for (auto it = pics_.begin(); it != pics_.end();) {
      bool keep = false;
      for (int id : f.dpb_ids) keep |= id == it->first;
      if (!keep) { free_pics_.push_back (new DevPic (it->second)); it = pics_.erase (it); } else ++it;
    }
    if (cur && f.is_ref) { pics_[f.id] = *cur; delete cur; }
    else if (cur) free_pics_.push_back (cur);
    while (free_pics_.size() > 4) { lh264_dev_free (free_pics_.back()->base); delete free_pics_.back(); free_pics_.pop_back(); }
  }

  bool inited_ = false, eos_ = false;
  SDecodingParam param_;
  std::unique_ptr<Parser> parser_;
  std::deque<Pending> pending_;
  std::map<int, DevPic> pics_;            // frame id -> reference picture resident in HBM
  std::vector<DevPic*> free_pics_;
  int32_t* d_chain_first_ = nullptr;
  lh264_frame_job_t* d_job_ = nullptr;
  lh264_mb_t* d_mbs_ = nullptr; int16_t* d_coeffs_ = nullptr; lh264_slice_t* d_slices_ = nullptr;
  size_t cap_mbs_ = 0, cap_slices_ = 0;
  std::vector<uint8_t> host_out_;
  SDecoderStatistics stats_;
  int ec_idc_ = 0, trace_level_ = 0, vcl_in_au_ = FEEDBACK_UNKNOWN_NAL, last_frame_num_ = -1, last_idr_pic_id_ = 0;
  WelsTraceCallback trace_cb_ = nullptr; void* trace_ctx_ = nullptr;
};

}  // namespace

extern "C" {

int WelsGetDecoderCapability (SDecoderCapability* c) {      // welsDecoderExt.cpp:663-677: the level 3.2 baseline figures
  if (!c) return 1;
  memset (c, 0, sizeof (*c));
  c->iProfileIdc = 66; c->iProfileIop = 0xE0; c->iLevelIdc = 32; c->iMaxMbps = 216000; c->iMaxFs = 5120;
  c->iMaxCpb = 20000; c->iMaxDpb = 20480; c->iMaxBr = 20000; c->bRedPicCap = false;
  return 0;
}
long WelsCreateDecoder (ISVCDecoder** pp) {
  if (!pp) return 1;                                         // ERR_INVALID_PARAMETERS
  *pp = new (std::nothrow) GpuDecoder();
  return *pp ? 0 : 2;
}
void WelsDestroyDecoder (ISVCDecoder* p) { delete static_cast<GpuDecoder*> (p); }
OpenH264Version WelsGetCodecVersion (void) { OpenH264Version v = {1, 4, 1, 0}; return v; }   // codec_ver.h:7
void WelsGetCodecVersionEx (OpenH264Version* v) { if (v) *v = WelsGetCodecVersion(); }

long lh264_isvc_initialize (ISVCDecoder* d, const SDecodingParam* p) { return d ? d->Initialize (p) : cmInitParaError; }
long lh264_isvc_uninitialize (ISVCDecoder* d) { return d ? d->Uninitialize() : cmInitParaError; }
int lh264_isvc_decode_frame2 (ISVCDecoder* d, const unsigned char* s, int n, unsigned char** dst, SBufferInfo* info) {
  return d ? (int)d->DecodeFrame2 (s, n, dst, info) : (int)dsInvalidArgument;
}
int lh264_isvc_decode_frame_no_delay (ISVCDecoder* d, const unsigned char* s, int n, unsigned char** dst, SBufferInfo* info) {
  return d ? (int)d->DecodeFrameNoDelay (s, n, dst, info) : (int)dsInvalidArgument;
}
long lh264_isvc_set_option (ISVCDecoder* d, int o, void* v) { return d ? d->SetOption ((DECODER_OPTION)o, v) : cmInitParaError; }
long lh264_isvc_get_option (ISVCDecoder* d, int o, void* v) { return d ? d->GetOption ((DECODER_OPTION)o, v) : cmInitParaError; }

}  // extern "C"
