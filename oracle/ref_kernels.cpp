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

}  // extern "C"
