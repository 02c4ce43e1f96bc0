/* oracle_coder.h - TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Plain-C restatement of SURVEY.md section 8 rows a9 (adaptive probabilities, binarisers, bool coder) and a10 (the
 * per-macroblock syntax symbols and their prior tables) of the reference's recompressor, compress direction:
 * given, per macroblock, the DecodedMacroblock fields the reference's emit code reads (tests/refdump.py:RTD_DTYPE) and
 * the quantised levels, it produces the byte string of every tagged arithmetic-coded stream (".pip.<tag>").
 * Pinned by tests/test_oracle_coder.py against byte streams and bit-level decision traces produced by the reference itself.
 */
#ifndef ORACLE_CODER_H_
#define ORACLE_CODER_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_coder orc_coder_t;

/* rtd record, layout == tests/refdump.py:RTD_DTYPE == oracle/ref_dump.cpp:pack_rtd */
#pragma pack(push, 1)
typedef struct orc_rtd {
  uint8_t have, slice_type, t8, cbp_c, cbp_l, chroma_mode, luma16_mode, luma_qp;
  uint32_t mb_type, num_ref_idx_l0;
  int32_t skip_run;
  int8_t ref_idx[4];
  uint8_t sub_type[4];
  int8_t pred_mode[16];
  int16_t mvd[16][2];
  int32_t delta_qp, last_mb_qp;
} orc_rtd_t;
#pragma pack(pop)

/* one slice of a picture: consecutive macroblocks first_mb .. first_mb + n_mbs - 1 */
typedef struct orc_slice_info {
  int32_t first_mb, n_mbs;
  int32_t slice_type;        /* 0 P, 2 I (eSliceType) */
  int32_t pad_bits, pad_value; /* bits left in the slice's last byte after its last macroblock, and their value */
  int32_t transform8x8_pps;  /* bit 0: PPS transform_8x8_mode_flag, bit 1: constrained_intra_pred_flag, bit 2: entropy_coding_mode_flag */
} orc_slice_info_t;

orc_coder_t* orc_coder_new (int keep_trace);
void orc_coder_free (orc_coder_t* c);
/* one picture.  mb_types[k]: the decoder's pMbType (MB_TYPE_SKIP = 0x100 for skipped macroblocks), levels: [n][384]
 * quantised levels (pScaledTCoeffQuant), rtd: [n] records (have == 0 for skipped macroblocks), avail: [n] the
 * neighbour availability bits of intra NxN macroblocks (lh264_mb_t.intra_avail: T 1, TL 2, L 4) */
int orc_coder_picture (orc_coder_t* c, int mb_w, int mb_h, int frame_num, const uint16_t* mb_types, const int16_t* levels,
                       const orc_rtd_t* rtd, const uint8_t* avail, const orc_slice_info_t* slices, int n_slices);
/* code a flat list of symbols given in the product's record format (include/lh264.h lh264_ctx_sym_t, LH264_SYM_* kinds 0..10) */
struct orc_sym;
int orc_coder_symbols (orc_coder_t* c, const struct orc_sym* syms, long n);
/* vpx_stop_encode on every tag; afterwards orc_coder_tag returns the final bytes */
void orc_coder_finish (orc_coder_t* c);
int orc_coder_tag (orc_coder_t* c, int tag, const uint8_t** bytes);     /* length, 0 if the tag was never used */
/* decision trace (keep_trace != 0): n entries of {tag, probability, bit} */
long orc_coder_trace (orc_coder_t* c, const uint8_t** tags, const uint8_t** probs, const uint8_t** bits);
const char* orc_coder_error (orc_coder_t* c);

#ifdef __cplusplus
}
#endif
#endif
