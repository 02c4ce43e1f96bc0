/* TEST INFRASTRUCTURE ONLY - see oracle_model.c */
#ifndef ORACLE_MODEL_H_
#define ORACLE_MODEL_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* one context-model symbol: which prior (flat index into the reference's table) codes which integer */
typedef struct orc_sym { uint32_t prior; int16_t value; uint8_t kind; uint8_t pad; } orc_sym_t;
enum { ORC_SYM_LUMA_DC = 0, ORC_SYM_CHROMA_DC = 1, ORC_SYM_NZ4 = 2, ORC_SYM_AC4 = 3, ORC_SYM_NZ8 = 4, ORC_SYM_AC8 = 5 };
#define ORC_MAX_SYMS 432   /* 16 + 8 + 24 + 384 */

int  orc_model_mb_type_code (int mb_type);
void orc_model_nnz24 (const int16_t levels[384], uint8_t nnz[24]);
/* symbols of one macroblock in emission order; nnz_* = per-4x4 nonzero counts of the LEFT / ABOVE / PAST
 * macroblocks as the model sees them (NULL = neighbour absent). returns the number of symbols. */
int  orc_model_mb_symbols (const int16_t levels[384], int mb_type, int slice_type, int cbp, int t8,
                           const uint8_t* nnz_left, const uint8_t* nnz_above, const uint8_t* nnz_past, orc_sym_t* out);
struct lh264_mb; struct lh264_slice;
void orc_model_frame_nnz (const struct lh264_mb* mbs, const int16_t* levels, int n_mbs, const uint8_t* past, uint8_t* cur);
long orc_model_frame_symbols (const struct lh264_mb* mbs, const struct lh264_slice* slices, const int16_t* levels, int mb_w, int mb_h,
                              const uint8_t* cur, const uint8_t* past, orc_sym_t* out, uint16_t* n_out);
#ifdef __cplusplus
}
#endif
#endif
