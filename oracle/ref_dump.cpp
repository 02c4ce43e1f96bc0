// TEST INFRASTRUCTURE ONLY (oracle/): our own shim around the *reference* decoder.
//
// Links against the reference libraries built by oracle/Makefile from the sources
// in /root/reference (nothing of the reference is copied here) and decodes one or
// more .264 files through ISVCDecoder::DecodeFrameNoDelay exactly like the
// reference CLI does (codec/console/dec/src/h264dec.cpp:246-364).  Two link-time
// hooks (ld --wrap) observe the reference's hot path without modifying it:
//
//   WelsDec::WelsTargetSliceConstruction (decode_slice.cpp:110)  -> per-slice dump of the
//        SDqLayer arrays (dec_frame.h:60-97) BEFORE reconstruction, i.e. what the reference
//        hands to its reconstruct kernels, in lh264_mb_t / lh264_slice_t form;
//   WelsDec::WelsDeblockingFilterSlice  (deblocking.cpp:872)      -> the slice's pixels after
//        reconstruction and before the in-loop filter.
//
// Output: one binary file per input stream (see tests/golden/make_golden.py for the
// reader) holding, per decoded frame: slices, macroblock records, coefficients,
// pre-deblock planes and final (deblocked) planes.
//
// usage: ref_dump out_dir in1.264 [in2.264 ...]
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <map>

#include "codec_api.h"
#include "codec_app_def.h"
#include "codec_def.h"
#include "decoder_context.h"
#include "dec_frame.h"
#include "slice.h"
#include "picture.h"
#include "wels_common_defs.h"
#include "macroblock_model.h"
#include "decoded_macroblock.h"

#include "../include/lh264.h"

using namespace WelsDec;

namespace {

struct Frame {
  int id = -1, mb_w = 0, mb_h = 0, crop_w = 0, crop_h = 0, has_final = 0, frame_num = 0;
  const void* dec_buf = nullptr;
  PPicture pic = nullptr;
  int last_first_mb = -1;
  std::vector<lh264_slice_t> slices;
  std::vector<lh264_mb_t> mbs;
  std::vector<int16_t> coeffs;
  std::vector<int32_t> ref_ids;          // union over slices, index = job ref slot
  std::vector<uint8_t> pre[3], fin[3];
  std::vector<uint8_t> covered;          // MB covered by some slice
  std::vector<int16_t> levels;           // pScaledTCoeffQuant (raw levels the context model reads)
  std::vector<uint8_t> nei;              // per MB: 3 x (present, nnz[24]) for LEFT, ABOVE, PAST as the model saw them
  std::vector<std::vector<uint8_t> > syms;   // per MB: packed symbols {u8 kind, i16 value, u32 prior}
  std::vector<int32_t> slice_extra;      // per slice: pad bit count, pad bit value, PPS transform_8x8_mode_flag, entropy_coding_mode_flag
  std::vector<uint8_t> rtd;              // per MB: RTD_BYTES of the DecodedMacroblock the recompressor coded (have=0: skipped)
};

// what the recompressor's per-macroblock emit code reads of DecodedMacroblock (decoded_macroblock.h:12-34), packed
enum { RTD_BYTES = 116 };
void pack_rtd (const DecodedMacroblock& d, uint8_t* o) {
  memset (o, 0, RTD_BYTES);
  o[0] = 1;                                   // have
  o[1] = d.eSliceType; o[2] = d.pTransformSize8x8Flag; o[3] = (uint8_t)d.uiCbpC; o[4] = (uint8_t)d.uiCbpL;
  o[5] = d.uiChmaI8x8Mode; o[6] = d.uiLumaI16x16Mode; o[7] = (uint8_t)d.uiLumaQp;
  uint32_t t = d.uiMbType; memcpy (o + 8, &t, 4);
  uint32_t nr = d.uiNumRefIdxL0Active; memcpy (o + 12, &nr, 4);
  int32_t sk = d.iMbSkipRun; memcpy (o + 16, &sk, 4);
  for (int i = 0; i < 4; i++) { o[20 + i] = (uint8_t)d.iRefIdx[i]; o[24 + i] = (uint8_t)d.uiSubMbType[i]; }
  for (int i = 0; i < 16; i++) o[28 + i] = (uint8_t) (int8_t)d.iBestIntra4x4PredMode[i];
  memcpy (o + 44, d.sMbMvp, 64);
  int32_t dq = (int32_t)d.cachedDeltaLumaQp; memcpy (o + 108, &dq, 4);
  int32_t lq = d.iLastMbQp; memcpy (o + 112, &lq, 4);
}

// context-model observations collected while the reference parses a slice (before its reconstruction hook fires)
struct MbObs { std::vector<uint8_t> syms; uint8_t nei[75]; bool have; uint8_t rtd[116]; };
const FreqImage* g_img = nullptr;
int g_prev_mb = -1;
std::map<int, MbObs> g_obs;     // mb index -> observation of the frame being parsed
int g_obs_mb = -1;
void obs_sym (int kind, int value, long prior) {
  if (g_obs_mb < 0) return;
  std::vector<uint8_t>& v = g_obs[g_obs_mb].syms;
  uint8_t b[7];
  b[0] = (uint8_t)kind; int16_t vv = (int16_t)value; uint32_t pp = (uint32_t)prior;
  memcpy (b + 1, &vv, 2); memcpy (b + 3, &pp, 4);
  v.insert (v.end(), b, b + 7);
}
void obs_nnz (const DecodedMacroblock* d, uint8_t* out) {
  out[0] = d ? 1 : 0;
  for (int i = 0; i < 24; i++) out[1 + i] = 0;
  if (!d) return;
  for (int i = 0; i < 16; i++) out[1 + i] = (uint8_t)d->countSubblockNonzeros (0, i);
  for (int i = 0; i < 4; i++) out[17 + i] = (uint8_t)d->countSubblockNonzeros (1, i);
  for (int i = 4; i < 8; i++) out[17 + i] = (uint8_t)d->countSubblockNonzeros (2, i);
}

FILE* g_out = nullptr;
int g_nframes = 0;
Frame g_cur;
bool g_have_cur = false;
std::map<const void*, int> g_buf_to_frame;   // pData[0] -> id of the last frame reconstructed there
int g_slice_first = 0, g_slice_n = 0;

void put32 (int32_t v) { fwrite (&v, 4, 1, g_out); }

void flush_frame() {
  if (!g_have_cur) return;
  Frame& f = g_cur;
  put32 (f.id); put32 (f.mb_w); put32 (f.mb_h); put32 ((int)f.slices.size());
  put32 (f.crop_w); put32 (f.crop_h); put32 (f.has_final); put32 ((int)f.ref_ids.size());
  for (int i = 0; i < LH264_MAX_REFS; i++) put32 (i < (int)f.ref_ids.size() ? f.ref_ids[i] : -1);
  put32 (f.frame_num);
  fwrite (f.slices.data(), sizeof (lh264_slice_t), f.slices.size(), g_out);
  fwrite (f.slice_extra.data(), 4, f.slice_extra.size(), g_out);
  fwrite (f.mbs.data(), sizeof (lh264_mb_t), f.mbs.size(), g_out);
  fwrite (f.coeffs.data(), 2, f.coeffs.size(), g_out);
  fwrite (f.covered.data(), 1, f.covered.size(), g_out);
  fwrite (f.levels.data(), 2, f.levels.size(), g_out);
  fwrite (f.nei.data(), 1, f.nei.size(), g_out);
  for (size_t k = 0; k < f.syms.size(); k++) { put32 ((int)f.syms[k].size()); fwrite (f.syms[k].data(), 1, f.syms[k].size(), g_out); }
  fwrite (f.rtd.data(), 1, f.rtd.size(), g_out);
  for (int p = 0; p < 3; p++) fwrite (f.pre[p].data(), 1, f.pre[p].size(), g_out);
  for (int p = 0; p < 3; p++) fwrite (f.fin[p].data(), 1, f.fin[p].size(), g_out);
  g_nframes++;
  g_have_cur = false;
}

void copy_mb_pixels (PWelsDecoderContext pCtx, Frame& f, int first, int n) {
  PPicture pic = pCtx->pDec;
  for (int k = first; k < first + n && k < f.mb_w * f.mb_h; k++) {
    int mx = k % f.mb_w, my = k / f.mb_w;
    for (int p = 0; p < 3; p++) {
      int bs = p ? 8 : 16, W = f.mb_w * bs;
      int ls = pic->iLinesize[p ? 1 : 0];
      for (int r = 0; r < bs; r++)
        memcpy (&f.pre[p][(my * bs + r) * W + mx * bs], pic->pData[p] + (my * bs + r) * ls + mx * bs, bs);
    }
  }
}

void capture_final (Frame& f, int cw, int ch) {
  PPicture pic = f.pic;
  if (!pic || pic->pData[0] != f.dec_buf) {
    // output picture is not the one we are tracking (error concealment / reordering): no final planes
    return;
  }
  for (int p = 0; p < 3; p++) {
    int bs = p ? 8 : 16, W = f.mb_w * bs, H = f.mb_h * bs;
    int ls = pic->iLinesize[p ? 1 : 0];
    f.fin[p].resize ((size_t)W * H);
    for (int r = 0; r < H; r++) memcpy (&f.fin[p][(size_t)r * W], pic->pData[p] + (size_t)r * ls, W);
  }
  f.crop_w = cw; f.crop_h = ch; f.has_final = 1;
}

}  // namespace

// The DecodedMacroblock of macroblock k is final only after its emit code ran; the reference then stores it into its
// FreqImage (decode_slice.cpp:3101-3109), where we read it at the next hook (next macroblock, or the slice's
// reconstruction).  A skipped macroblock's entry is a copy of PAST and is not recorded.
void capture_prev (WelsDec::PWelsDecoderContext ctx) {
  if (g_prev_mb < 0 || !g_img) return;
  MbObs& o = g_obs[g_prev_mb];
  const uint32_t t = ctx->pCurDqLayer->pMbType[g_prev_mb];
  if (t != MB_TYPE_SKIP && t != 0) pack_rtd (g_img->at (g_prev_mb), o.rtd);
  g_prev_mb = -1;
}

// ---- link-time hooks ------------------------------------------------------------------------
namespace WelsDec {
int32_t WelsTargetSliceConstruction (PWelsDecoderContext pCtx);
void WelsDeblockingFilterSlice (PWelsDecoderContext pCtx, PDeblockingFilterMbFunc pDeblockMb);
}
extern "C" {
int32_t __real__ZN7WelsDec27WelsTargetSliceConstructionEPNS_21TagWelsDecoderContextE (PWelsDecoderContext);
void __real__ZN7WelsDec25WelsDeblockingFilterSliceEPNS_21TagWelsDecoderContextEPFvPNS_10TagDqLayerEPNS_19tagDeblockingFilterEiE (
  PWelsDecoderContext, PDeblockingFilterMbFunc);

int32_t __wrap__ZN7WelsDec27WelsTargetSliceConstructionEPNS_21TagWelsDecoderContextE (PWelsDecoderContext pCtx) {
  capture_prev (pCtx);
  PDqLayer L = pCtx->pCurDqLayer;
  PSlice pSlice = &L->sLayerInfo.sSliceInLayer;
  PSliceHeader sh = &pSlice->sSliceHeaderExt.sSliceHeader;
  const int first = sh->iFirstMbInSlice, n = pSlice->iTotalMbInCurSlice;
  const int mbw = L->iMbWidth, mbh = L->iMbHeight;
  const void* buf = pCtx->pDec ? pCtx->pDec->pData[0] : nullptr;

  bool new_frame = !g_have_cur || g_cur.dec_buf != buf || first <= g_cur.last_first_mb
                   || g_cur.mb_w != mbw || g_cur.mb_h != mbh;
  if (new_frame) {
    flush_frame();
    g_cur = Frame();
    g_cur.id = g_nframes; g_cur.mb_w = mbw; g_cur.mb_h = mbh; g_cur.dec_buf = buf; g_cur.pic = pCtx->pDec; g_cur.frame_num = pCtx->iFrameNum;
    g_cur.mbs.assign ((size_t)mbw * mbh, lh264_mb_t());
    memset (g_cur.mbs.data(), 0, g_cur.mbs.size() * sizeof (lh264_mb_t));
    g_cur.coeffs.assign ((size_t)mbw * mbh * 384, 0);
    g_cur.covered.assign ((size_t)mbw * mbh, 0);
    g_cur.levels.assign ((size_t)mbw * mbh * 384, 0);
    g_cur.nei.assign ((size_t)mbw * mbh * 75, 0);
    g_cur.rtd.assign ((size_t)mbw * mbh * RTD_BYTES, 0);
    g_cur.syms.assign ((size_t)mbw * mbh, std::vector<uint8_t>());
    for (int p = 0; p < 3; p++) g_cur.pre[p].assign ((size_t)mbw * mbh * (p ? 64 : 256), 0);
    g_have_cur = true;
    g_buf_to_frame[buf] = g_cur.id;
  }
  Frame& f = g_cur;
  f.last_first_mb = first;

  lh264_slice_t s;
  memset (&s, 0, sizeof (s));
  s.first_mb = first; s.n_mbs = n;
  s.slice_type = (uint8_t)pSlice->eSliceType;
  s.deblock_idc = (uint8_t)sh->uiDisableDeblockingFilterIdc;
  s.alpha_c0_offset = (int8_t)sh->iSliceAlphaC0Offset;
  s.beta_offset = (int8_t)sh->iSliceBetaOffset;
  s.weighted_pred = L->bUseWeightPredictionFlag ? 1 : 0;
  s.n_refs = (uint8_t)sh->uiRefCount[0];
  s.luma_dc_weight = pCtx->bUseScalingList ? (uint8_t)(pCtx->pDequant_coeff4x4[0][0][0] / 10) : 16; // [list 0 = Intra-Y][qp 0][pos 0] = weight * 10
  if (L->pPredWeightTable) {
    s.luma_log2_denom = (uint8_t)L->pPredWeightTable->uiLumaLog2WeightDenom;
    s.chroma_log2_denom = (uint8_t)L->pPredWeightTable->uiChromaLog2WeightDenom;
    for (int i = 0; i < LH264_MAX_REFS; i++) {
      s.luma_weight[i] = (int16_t)L->pPredWeightTable->sPredList[0].iLumaWeight[i];
      s.luma_offset[i] = (int16_t)L->pPredWeightTable->sPredList[0].iLumaOffset[i];
      for (int c = 0; c < 2; c++) {
        s.chroma_weight[i][c] = (int16_t)L->pPredWeightTable->sPredList[0].iChromaWeight[i][c];
        s.chroma_offset[i][c] = (int16_t)L->pPredWeightTable->sPredList[0].iChromaOffset[i][c];
      }
    }
  }
  for (int i = 0; i < LH264_MAX_REFS; i++) {
    s.ref_slot[i] = -1;
    if (pSlice->eSliceType == P_SLICE && i < (int)sh->uiRefCount[0]) {
      PPicture r = pCtx->sRefPic.pRefList[LIST_0][i];
      if (!r) r = pCtx->sRefPic.pRefList[LIST_0][0];   // rec_mb.cpp:238-242 fallback
      if (!r) continue;
      int fid = -1;
      std::map<const void*, int>::iterator it = g_buf_to_frame.find (r->pData[0]);
      if (it != g_buf_to_frame.end()) fid = it->second;
      int slot = -1;
      for (size_t k = 0; k < f.ref_ids.size(); k++) if (f.ref_ids[k] == fid) slot = (int)k;
      if (slot < 0 && f.ref_ids.size() < LH264_MAX_REFS) { f.ref_ids.push_back (fid); slot = (int)f.ref_ids.size() - 1; }
      s.ref_slot[i] = (int8_t)slot;
    }
  }
  const int sid = (int)f.slices.size();
  f.slices.push_back (s);
  {   // what WelsDecodeSlice sends to the pad-byte tag after the slice's last macroblock (decode_slice.cpp:3133-3148)
    PBitStringAux pBs = L->pBitStringAux;
    const int nPad = 7 - (pBs->iLeftBits & 0x7);
    f.slice_extra.push_back (nPad);
    f.slice_extra.push_back (nPad ? (pBs->pEndBuf[-1] & ((1 << nPad) - 1)) : 0);
    f.slice_extra.push_back (pCtx->pPps->bTransform8x8ModeFlag ? 1 : 0);
    f.slice_extra.push_back (pCtx->pPps->bEntropyCodingModeFlag ? 1 : 0);
  }

  for (int k = first; k < first + n && k < mbw * mbh; k++) {
    lh264_mb_t& m = f.mbs[k];
    memset (&m, 0, sizeof (m));
    m.mb_type = (uint16_t)L->pMbType[k];
    m.cbp = (uint8_t)L->pCbp[k];
    m.qp_y = (uint8_t)L->pLumaQp[k];
    m.qp_c[0] = (uint8_t)L->pChromaQp[k][0];
    m.qp_c[1] = (uint8_t)L->pChromaQp[k][1];
    m.flags = L->pTransformSize8x8Flag[k] ? LH264_MBF_T8x8 : 0;
    m.intra_avail = L->pIntraNxNAvailFlag[k];
    for (int i = 0; i < 16; i++) m.intra_mode[i] = L->pIntra4x4FinalMode[k][i];
    if (m.mb_type == MB_TYPE_INTRA16x16) m.intra_mode[0] = L->pIntraPredMode[k][7];
    m.chroma_mode = L->pChromaPredMode[k];
    m.slice_id = (uint16_t)sid;
    for (int i = 0; i < 4; i++) {
      m.sub_type[i] = (uint8_t)L->pSubMbType[k][i];
      m.ref_idx[i] = L->pRefIndex[0][k][((i >> 1) << 3) + ((i & 1) << 1)];
    }
    for (int i = 0; i < 24; i++) m.nzc[i] = (uint8_t)L->pNzc[k][i];
    for (int i = 0; i < 16; i++) { m.mv[i][0] = L->pMv[0][k][i][0]; m.mv[i][1] = L->pMv[0][k][i][1]; }
    memcpy (&f.coeffs[(size_t)k * 384], L->pScaledTCoeff[k], 768);
    memcpy (&f.levels[(size_t)k * 384], L->pScaledTCoeffQuant[k], 768);
    { std::map<int, MbObs>::iterator it = g_obs.find (k);
      if (it != g_obs.end()) { f.syms[k] = it->second.syms; memcpy (&f.nei[(size_t)k * 75], it->second.nei, 75); memcpy (&f.rtd[(size_t)k * RTD_BYTES], it->second.rtd, RTD_BYTES); g_obs.erase (it); } }
    f.covered[k] = 1;
    if (m.mb_type == MB_TYPE_INTRA_PCM) {
      // the reference writes I_PCM samples into the frame while parsing (decode_slice.cpp:3213-3263);
      // carry them in the (otherwise unused) coefficient slot
      PPicture pic = pCtx->pDec;
      int mx = k % mbw, my = k / mbw;
      int16_t* c = &f.coeffs[(size_t)k * 384];
      for (int r = 0; r < 16; r++) for (int x = 0; x < 16; x++)
          c[r * 16 + x] = pic->pData[0][(my * 16 + r) * pic->iLinesize[0] + mx * 16 + x];
      for (int p = 1; p < 3; p++) for (int r = 0; r < 8; r++) for (int x = 0; x < 8; x++)
            c[256 + (p - 1) * 64 + r * 8 + x] = pic->pData[p][(my * 8 + r) * pic->iLinesize[1] + mx * 8 + x];
      m.flags |= LH264_MBF_PCM_IN_COEFF;
    }
  }

  g_slice_first = first; g_slice_n = n;
  int32_t rc = __real__ZN7WelsDec27WelsTargetSliceConstructionEPNS_21TagWelsDecoderContextE (pCtx);
  bool deblock_ran = (pSlice->eSliceType == I_SLICE || pSlice->eSliceType == P_SLICE)
                     && sh->uiDisableDeblockingFilterIdc != 1 && n > 0;
  if (!deblock_ran) copy_mb_pixels (pCtx, f, first, n);
  return rc;
}

void __wrap__ZN7WelsDec25WelsDeblockingFilterSliceEPNS_21TagWelsDecoderContextEPFvPNS_10TagDqLayerEPNS_19tagDeblockingFilterEiE (
  PWelsDecoderContext pCtx, PDeblockingFilterMbFunc fn) {
  if (g_have_cur) copy_mb_pixels (pCtx, g_cur, g_slice_first, g_slice_n);
  __real__ZN7WelsDec25WelsDeblockingFilterSliceEPNS_21TagWelsDecoderContextEPFvPNS_10TagDqLayerEPNS_19tagDeblockingFilterEiE (pCtx, fn);
}

// ---- context-model hooks (macroblock_model.cpp): record which prior every coefficient symbol used -------------
#define MM_INIT _ZN15MacroblockModel21initCurrentMacroblockEP17DecodedMacroblockPN7WelsDec21TagWelsDecoderContextEPK9FreqImageii
#define MM_NZ4  _ZN15MacroblockModel19getNonzerosPrior4x4Eii
#define MM_NZ8  _ZN15MacroblockModel19getNonzerosPrior8x8Eii
#define MM_AC4  _ZN15MacroblockModel13getACPrior4x4EiiiRKSt6vectorIiSaIiEEi
#define MM_AC8  _ZN15MacroblockModel13getACPrior8x8EiiiRKSt6vectorIiSaIiEEi
#define MM_LDC  _ZN15MacroblockModel17getLumaDCIntPriorEm
#define MM_CDC  _ZN15MacroblockModel19getChromaDCIntPriorEm
#define REAL_(x) __real_##x
#define REAL(x) REAL_(x)
#define WRAP_(x) __wrap_##x
#define WRAP(x) WRAP_(x)
void REAL(MM_INIT) (MacroblockModel*, DecodedMacroblock*, PWelsDecoderContext, const FreqImage*, int, int);
MacroblockModel::NonzerosPrior* REAL(MM_NZ4) (MacroblockModel*, int, int);
MacroblockModel::NonzerosPrior* REAL(MM_NZ8) (MacroblockModel*, int, int);
MacroblockModel::ACPrior* REAL(MM_AC4) (MacroblockModel*, int, int, int, const std::vector<int>&, int);
MacroblockModel::ACPrior* REAL(MM_AC8) (MacroblockModel*, int, int, int, const std::vector<int>&, int);
MacroblockModel::DCPrior* REAL(MM_LDC) (MacroblockModel*, size_t);
MacroblockModel::DCPrior* REAL(MM_CDC) (MacroblockModel*, size_t);

void WRAP(MM_INIT) (MacroblockModel* self, DecodedMacroblock* mb, PWelsDecoderContext ctx, const FreqImage* f, int x, int y) {
  capture_prev (ctx);
  REAL(MM_INIT) (self, mb, ctx, f, x, y);
  g_obs_mb = y * (int)f->width + x;
  g_prev_mb = g_obs_mb; g_img = f;
  MbObs& o = g_obs[g_obs_mb];
  o.syms.clear();
  memset (o.rtd, 0, sizeof (o.rtd));
  obs_nnz (self->n[Nei::LEFT], o.nei); obs_nnz (self->n[Nei::ABOVE], o.nei + 25); obs_nnz (self->n[Nei::PAST], o.nei + 50);
}
MacroblockModel::NonzerosPrior* WRAP(MM_NZ4) (MacroblockModel* self, int color, int idx) {
  MacroblockModel::NonzerosPrior* p = REAL(MM_NZ4) (self, color, idx);
  const int16_t* ac = self->mb->getAC (color, idx);
  int nz = 0;   // the value coded with this prior (decode_slice.cpp:2061-2064)
  const bool emit_dc = color ? !(self->mb->uiCbpC == 1 || self->mb->uiCbpC == 2) : (self->mb->uiMbType != MB_TYPE_INTRA16x16);
  for (int i = emit_dc ? 0 : 1; i < 16; i++) if (ac[i]) nz++;
  obs_sym (2, nz, (long) (p - &self->nonzerosPriors.at (0, 0, 0, 0, 0, 0)));
  return p;
}
MacroblockModel::NonzerosPrior* WRAP(MM_NZ8) (MacroblockModel* self, int color, int idx) {
  MacroblockModel::NonzerosPrior* p = REAL(MM_NZ8) (self, color, idx);
  const int16_t* ac = self->mb->getAC (color, idx);
  int nz = 0;
  for (int i = (self->mb->uiMbType != MB_TYPE_INTRA16x16) ? 0 : 1; i < 64; i++) if (ac[i]) nz++;
  obs_sym (4, nz, (long) (p - &self->nonzerosPriors8x8.at (0, 0, 0, 0, 0, 0)));
  return p;
}
MacroblockModel::ACPrior* WRAP(MM_AC4) (MacroblockModel* self, int idx, int coef, int color, const std::vector<int>& em, int nz) {
  MacroblockModel::ACPrior* p = REAL(MM_AC4) (self, idx, coef, color, em, nz);
  const int16_t* ac = self->mb->getAC (color, idx);
  obs_sym (3, ac[coef], (long) (p - &self->acPriors.at (0, 0, 0, 0).at (0, 0, 0, 0, 0)));
  return p;
}
MacroblockModel::ACPrior* WRAP(MM_AC8) (MacroblockModel* self, int idx, int coef, int color, const std::vector<int>& em, int nz) {
  MacroblockModel::ACPrior* p = REAL(MM_AC8) (self, idx, coef, color, em, nz);
  const int16_t* ac = self->mb->getAC (color, idx);
  obs_sym (5, ac[coef], (long) (p - &self->acPriors8x8.at (0, 0, 0, 0).at (0, 0, 0, 0, 0)));
  return p;
}
MacroblockModel::DCPrior* WRAP(MM_LDC) (MacroblockModel* self, size_t i) {
  MacroblockModel::DCPrior* p = REAL(MM_LDC) (self, i);
  obs_sym (0, self->mb->odata.lumaDC[i], (long) (p - &self->lumaDCIntPriors.at (0, 0, 0)));
  return p;
}
MacroblockModel::DCPrior* WRAP(MM_CDC) (MacroblockModel* self, size_t i) {
  MacroblockModel::DCPrior* p = REAL(MM_CDC) (self, i);
  obs_sym (1, self->mb->odata.chromaDC[i], (long) (p - &self->chromaDCIntPriors.at (0, 0, 0)));
  return p;
}
}  // extern "C"

static int decode_one (const char* in, const char* outdir) {
  FILE* fi = fopen (in, "rb");
  if (!fi) { fprintf (stderr, "cannot open %s\n", in); return 1; }
  fseek (fi, 0, SEEK_END); long sz = ftell (fi); fseek (fi, 0, SEEK_SET);
  std::vector<uint8_t> buf (sz + 4);
  if (fread (buf.data(), 1, sz, fi) != (size_t)sz) { fclose (fi); return 1; }
  fclose (fi);
  static const uint8_t sc[4] = {0, 0, 0, 1};
  memcpy (&buf[sz], sc, 4);   // h264dec.cpp:227,238 appends a start code

  std::string base = in;
  size_t sl = base.find_last_of ('/');
  if (sl != std::string::npos) base = base.substr (sl + 1);
  std::string outp = std::string (outdir) + "/" + base + ".dmp";
  g_out = fopen (outp.c_str(), "wb");
  if (!g_out) { fprintf (stderr, "cannot write %s\n", outp.c_str()); return 1; }
  fwrite ("LH264DMP", 1, 8, g_out);
  put32 (5); put32 (0);
  g_nframes = 0; g_have_cur = false; g_buf_to_frame.clear(); g_obs.clear(); g_obs_mb = -1;

  ISVCDecoder* dec = nullptr;
  if (WelsCreateDecoder (&dec) || !dec) { fprintf (stderr, "WelsCreateDecoder failed\n"); return 1; }
  SDecodingParam p;
  memset (&p, 0, sizeof (p));
  p.eOutputColorFormat = videoFormatI420;
  p.uiTargetDqLayer = (uint8_t) - 1;
  p.eEcActiveIdc = ERROR_CON_SLICE_COPY;
  p.sVideoProperty.eVideoBsType = VIDEO_BITSTREAM_DEFAULT;
  if (dec->Initialize (&p)) { fprintf (stderr, "Initialize failed\n"); return 1; }

  long pos = 0;
  int nout = 0;
  const char* mf = getenv ("REF_DUMP_MAX_FRAMES");       // stop feeding after this many output pictures (short fixtures)
  const int max_out = mf ? atoi (mf) : 0;
  while (pos < sz) {
    if (max_out > 0 && nout >= max_out) break;
    // next start-code-delimited chunk, as h264dec.cpp:246-272 does
    long i;
    for (i = 0; i < sz - pos; i++) {
      if ((buf[pos + i] == 0 && buf[pos + i + 1] == 0 && buf[pos + i + 2] == 0 && buf[pos + i + 3] == 1 && i > 0) ||
          (buf[pos + i] == 0 && buf[pos + i + 1] == 0 && buf[pos + i + 2] == 1 && i > 0)) break;
    }
    long slice = i;
    if (slice < 4) { pos += slice; continue; }
    uint8_t* dst[3] = {0, 0, 0};
    SBufferInfo info;
    memset (&info, 0, sizeof (info));
    dec->DecodeFrameNoDelay (buf.data() + pos, (int)slice, dst, &info);
    if (info.iBufferStatus == 1) {
      if (g_have_cur) capture_final (g_cur, info.UsrData.sSystemBuffer.iWidth, info.UsrData.sSystemBuffer.iHeight);
      nout++;
    }
    pos += slice;
  }
  flush_frame();
  {
    // the recompressor's output as it would be written by the console app (h264dec.cpp:79-121): one byte string per tag
    struct Capture : public CompressedWriter {
      std::map<int, std::vector<uint8_t> > streams;
      std::pair<uint32_t, H264Error> Write (int streamId, const uint8_t* data, unsigned int size) {
        std::vector<uint8_t>& v = streams[streamId];
        v.insert (v.end(), data, data + size);
        return std::pair<uint32_t, H264Error> (size, 0);
      }
      void Close() {}
    } cap;
    oMovie().flushToWriter (cap);
    fwrite ("TAGS", 1, 4, g_out);
    put32 ((int)cap.streams.size());
    for (auto& kv : cap.streams) { put32 (kv.first); put32 ((int)kv.second.size()); fwrite (kv.second.data(), 1, kv.second.size(), g_out); }
  }
  fseek (g_out, 12, SEEK_SET); put32 (g_nframes);
  fclose (g_out); g_out = nullptr;
  dec->Uninitialize();
  WelsDestroyDecoder (dec);
  fprintf (stderr, "%s: %d frames dumped, %d output -> %s\n", in, g_nframes, nout, outp.c_str());
  return 0;
}

int main (int argc, char** argv) {
  if (argc < 3) { fprintf (stderr, "usage: ref_dump out_dir in.264 [...]\n"); return 2; }
  int rc = 0;
  for (int i = 2; i < argc; i++) rc |= decode_one (argv[i], argv[1]);
  return rc;
}
