// TEST INFRASTRUCTURE ONLY (oracle/): flat C entry points onto the REFERENCE's own C kernels,
// linked from the libraries oracle/Makefile builds out of /root/reference.  tests/test_oracle_vs_ref.py
// drives these and oracle/liboracle.so with the same random inputs (the pattern of the reference's own
// unit tests: test/decoder/DecUT_IdctResAddPred.cpp, DecUT_IntraPrediction.cpp, DecUT_DeblockCommon.cpp,
// test/encoder/EncUT_MotionCompensation.cpp).  Nothing here is product code and no reference source is copied.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "decoder_context.h"
#include "decode_mb_aux.h"
#include "decode_slice.h"
#include "get_intra_predictor.h"
#include "deblocking_common.h"
#include "expand_pic.h"
#include "mc.h"
#include "deblocking.h"
#include "../include/lh264.h"   /* our record layout (input of the shim) */
#include <vector>

using namespace WelsDec;

extern "C" {

void refk_idct4x4_add (uint8_t* dst, int stride, int16_t* coef) { IdctResAddPred_c (dst, stride, coef); }
void refk_idct8x8_add (uint8_t* dst, int stride, int16_t* coef) { IdctResAddPred8x8_c (dst, stride, coef); }
void refk_chroma_dc_idct (int16_t* blk) { WelsChromaDcIdct (blk); }
void refk_luma_dc_dequant_idct (int16_t* blk, int qp) {
  // the function only reads bUseScalingList / pDequant_coeff4x4 of the context (decode_slice.cpp:272)
  static SWelsDecoderContext* ctx = (SWelsDecoderContext*)calloc (1, sizeof (SWelsDecoderContext));
  ctx->bUseScalingList = false;
  WelsLumaDcDequantIdct (blk, qp, ctx);
}

void refk_pred4x4 (uint8_t* dst, int stride, int mode) {
  // table order of decoder.cpp:919-964 == wels_common_defs.h:314-331
  static PGetIntraPredFunc f[14] = {
    WelsI4x4LumaPredV_c, WelsI4x4LumaPredH_c, WelsI4x4LumaPredDc_c, WelsI4x4LumaPredDDL_c, WelsI4x4LumaPredDDR_c,
    WelsI4x4LumaPredVR_c, WelsI4x4LumaPredHD_c, WelsI4x4LumaPredVL_c, WelsI4x4LumaPredHU_c, WelsI4x4LumaPredDcLeft_c,
    WelsI4x4LumaPredDcTop_c, WelsI4x4LumaPredDcNA_c, WelsI4x4LumaPredDDLTop_c, WelsI4x4LumaPredVLTop_c
  };
  f[mode] (dst, stride);
}
void refk_pred8x8l (uint8_t* dst, int stride, int mode, int tl, int tr) {
  static PGetIntraPred8x8Func f[14] = {
    WelsI8x8LumaPredV_c, WelsI8x8LumaPredH_c, WelsI8x8LumaPredDc_c, WelsI8x8LumaPredDDL_c, WelsI8x8LumaPredDDR_c,
    WelsI8x8LumaPredVR_c, WelsI8x8LumaPredHD_c, WelsI8x8LumaPredVL_c, WelsI8x8LumaPredHU_c, WelsI8x8LumaPredDcLeft_c,
    WelsI8x8LumaPredDcTop_c, WelsI8x8LumaPredDcNA_c, WelsI8x8LumaPredDDLTop_c, WelsI8x8LumaPredVLTop_c
  };
  f[mode] (dst, stride, tl != 0, tr != 0);
}
void refk_pred16x16 (uint8_t* dst, int stride, int mode) {
  static PGetIntraPredFunc f[7] = {
    WelsI16x16LumaPredV_c, WelsI16x16LumaPredH_c, WelsI16x16LumaPredDc_c, WelsI16x16LumaPredPlane_c,
    WelsI16x16LumaPredDcLeft_c, WelsI16x16LumaPredDcTop_c, WelsI16x16LumaPredDcNA_c
  };
  f[mode] (dst, stride);
}
void refk_predc8x8 (uint8_t* dst, int stride, int mode) {
  static PGetIntraPredFunc f[7] = {
    WelsIChromaPredDc_c, WelsIChromaPredH_c, WelsIChromaPredV_c, WelsIChromaPredPlane_c,
    WelsIChromaPredDcLeft_c, WelsIChromaPredDcTop_c, WelsIChromaPredDcNA_c
  };
  f[mode] (dst, stride);
}

static SMcFunc* mcf() {
  static SMcFunc f;
  static bool init = false;
  if (!init) { WelsCommon::InitMcFunc (&f, 0); init = true; }
  return &f;
}
void refk_mc_luma (const uint8_t* src, int sstride, uint8_t* dst, int dstride, int mvx, int mvy, int w, int h) {
  mcf()->pMcLumaFunc (src, sstride, dst, dstride, (int16_t)mvx, (int16_t)mvy, w, h);
}
void refk_mc_chroma (const uint8_t* src, int sstride, uint8_t* dst, int dstride, int mvx, int mvy, int w, int h) {
  mcf()->pMcChromaFunc (src, sstride, dst, dstride, (int16_t)mvx, (int16_t)mvy, w, h);
}

// xstride/ystride form (as DeblockLumaLt4_c): vertical edge = (1, stride), horizontal = (stride, 1)
void refk_deblock_luma_lt4 (uint8_t* pix, int stride, int vertical_edge, int alpha, int beta, int8_t* tc) {
  if (vertical_edge) DeblockLumaLt4H_c (pix, stride, alpha, beta, tc); else DeblockLumaLt4V_c (pix, stride, alpha, beta, tc);
}
void refk_deblock_luma_eq4 (uint8_t* pix, int stride, int vertical_edge, int alpha, int beta) {
  if (vertical_edge) DeblockLumaEq4H_c (pix, stride, alpha, beta); else DeblockLumaEq4V_c (pix, stride, alpha, beta);
}
void refk_deblock_chroma_lt4 (uint8_t* pix, int stride, int vertical_edge, int alpha, int beta, int8_t* tc) {
  if (vertical_edge) DeblockChromaLt4H2_c (pix, stride, alpha, beta, tc); else DeblockChromaLt4V2_c (pix, stride, alpha, beta, tc);
}
void refk_deblock_chroma_eq4 (uint8_t* pix, int stride, int vertical_edge, int alpha, int beta) {
  if (vertical_edge) DeblockChromaEq4H2_c (pix, stride, alpha, beta); else DeblockChromaEq4V2_c (pix, stride, alpha, beta);
}

// ExpandReferencingPicture through the reference's own function table (expand_pic.cpp:116-174)
void refk_expand_picture (uint8_t* y, uint8_t* u, uint8_t* v, int w, int h, int stride_y, int stride_c) {
  static SExpandPicFunc f;
  static bool init = false;
  if (!init) { InitExpandPictureFunc (&f, 0); init = true; }
  uint8_t* data[3] = {y, u, v};
  int32_t st[3] = {stride_y, stride_c, stride_c};
  ExpandReferencingPicture (data, w, h, st, f.pfExpandLumaPicture, f.pfExpandChromaPicture);
}

// The reference's per-macroblock deblocking drivers (WelsDeblockingMb -> DeblockingIntraMb / DeblockingInterMb with the boundary
// strength derivation, deblocking.cpp:160-352,568-862) over a whole synthetic picture in raster order, the way
// WelsDeblockingFilterSlice (:872-934) walks a slice: the layer's per-macroblock arrays are filled from our records, the planes are
// filtered in place.  Pattern of the reference's own test/decoder/DecUT_DeblockCommon.cpp:417-979.
void refk_deblock_picture (const lh264_mb_t* mbs, const lh264_slice_t* slices, int mb_w, int mb_h, uint8_t* y, uint8_t* u, uint8_t* v,
                           int stride_y, int stride_c) {
  const int n = mb_w * mb_h;
  SDqLayer L; memset (&L, 0, sizeof (L));
  std::vector<int16_t> type (n); std::vector<int32_t> sidc (n); std::vector<int8_t> qp (n);
  std::vector<int8_t> cqp (2 * n), nzc (24 * n), ref (16 * n); std::vector<int16_t> mv (32 * n);
  bool* t8 = (bool*)calloc (n, sizeof (bool));
  for (int k = 0; k < n; k++) {
    const lh264_mb_t& m = mbs[k];
    type[k] = (int16_t)m.mb_type; sidc[k] = m.slice_id; qp[k] = (int8_t)m.qp_y; cqp[2 * k] = (int8_t)m.qp_c[0]; cqp[2 * k + 1] = (int8_t)m.qp_c[1];
    t8[k] = (m.flags & LH264_MBF_T8x8) != 0;
    for (int i = 0; i < 24; i++) nzc[24 * k + i] = m.nzc[i] != 0;      // as the reconstruction leaves them (pWelsSetNonZeroCountFunc, decode_slice.cpp:266-267)
    for (int b = 0; b < 16; b++) {
      mv[32 * k + 2 * b] = m.mv[b][0]; mv[32 * k + 2 * b + 1] = m.mv[b][1];
      ref[16 * k + b] = m.ref_idx[((b >> 3) << 1) + ((b & 3) >> 1)];
    }
  }
  L.pMbType = type.data(); L.pSliceIdc = sidc.data(); L.pLumaQp = qp.data();
  L.pChromaQp = (int8_t (*)[2])cqp.data(); L.pNzc = (int8_t (*)[24])nzc.data();
  L.pMv[0] = (int16_t (*)[16][2])mv.data(); L.pRefIndex[0] = (int8_t (*)[16])ref.data();
  L.pTransformSize8x8Flag = t8; L.iMbWidth = mb_w; L.iMbHeight = mb_h;
  SDeblockingFunc fn; DeblockingInit (&fn, 0);
  SDeblockingFilter F; memset (&F, 0, sizeof (F));
  F.pCsData[0] = y; F.pCsData[1] = u; F.pCsData[2] = v; F.iCsStride[0] = stride_y; F.iCsStride[1] = stride_c;
  F.pLoopf = &fn;
  for (int k = 0; k < n; k++) {
    const lh264_slice_t& sl = slices[mbs[k].slice_id];          // the slice header fields WelsDeblockingFilterSlice copies (:893-899)
    if (sl.deblock_idc == 1) continue;
    F.iSliceAlphaC0Offset = sl.alpha_c0_offset; F.iSliceBetaOffset = sl.beta_offset;
    L.iMbX = k % mb_w; L.iMbY = k / mb_w; L.iMbXyIndex = k;
    WelsDeblockingMb (&L, &F, DeblockingAvailableNoInterlayer (&L, sl.deblock_idc));
  }
  free (t8);
}

}  // extern "C"
