/* TEST INFRASTRUCTURE ONLY - see oracle_recon.c */
#ifndef ORACLE_RECON_H_
#define ORACLE_RECON_H_
#include <stdint.h>
#include "../include/lh264.h"

#ifdef __cplusplus
extern "C" {
#endif

/* pixel kernels (in place on a plane, neighbours read from the plane like the reference does) */
void orc_idct4x4_add (uint8_t* dst, int stride, const int16_t* coef);
void orc_idct8x8_add (uint8_t* dst, int stride, const int16_t* coef);
void orc_luma_dc_dequant_idct (int16_t* mb_coef, int qmul);
void orc_chroma_dc_idct (int16_t* plane_coef);
void orc_pred4x4 (uint8_t* dst, int stride, int mode);
void orc_pred8x8l (uint8_t* dst, int stride, int mode, int tl_avail, int tr_avail);
void orc_pred16x16 (uint8_t* dst, int stride, int mode);
void orc_predc8x8 (uint8_t* dst, int stride, int mode);
void orc_mc_luma (const uint8_t* src, int sstride, uint8_t* dst, int dstride, int mvx, int mvy, int w, int h);
void orc_mc_chroma (const uint8_t* src, int sstride, uint8_t* dst, int dstride, int mvx, int mvy, int w, int h);
void orc_deblock_luma_lt4 (uint8_t* pix, int xstride, int ystride, int alpha, int beta, const int8_t* tc);
void orc_deblock_luma_eq4 (uint8_t* pix, int xstride, int ystride, int alpha, int beta);
void orc_deblock_chroma_lt4 (uint8_t* pix, int xstride, int ystride, int alpha, int beta, const int8_t* tc);
void orc_deblock_chroma_eq4 (uint8_t* pix, int xstride, int ystride, int alpha, int beta);
int  orc_luma_dc_qmul (int qp, int weight);

/* a picture in host memory, same layout as lh264_pic_t (plane pointers at pixel (0,0), padded) */
typedef struct orc_pic {
  uint8_t* y; uint8_t* u; uint8_t* v;
  int stride_y, stride_c;
} orc_pic_t;

/* frame-level drivers, same contract as lh264_recon_frames for one job */
void orc_recon_slice (const lh264_mb_t* mbs, const int16_t* coeffs, const lh264_slice_t* slices, int slice_idx,
                      orc_pic_t* dst, const orc_pic_t* refs, int mb_w, int mb_h);
void orc_deblock_slice (const lh264_mb_t* mbs, const lh264_slice_t* slices, int slice_idx,
                        orc_pic_t* dst, int mb_w, int mb_h);
void orc_expand_pic (orc_pic_t* pic, int mb_w, int mb_h);
void orc_recon_frame (const lh264_mb_t* mbs, const int16_t* coeffs, const lh264_slice_t* slices, int n_slices,
                      orc_pic_t* dst, const orc_pic_t* refs, int mb_w, int mb_h, int flags);

#ifdef __cplusplus
}
#endif
#endif
