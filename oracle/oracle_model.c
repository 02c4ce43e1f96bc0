/*
 * TEST INFRASTRUCTURE ONLY.  Plain-C restatement of the recompressor's per-coefficient context-index
 * computation: which adaptive prior codes each coefficient symbol of a
 * macroblock.  Checker for the HIP ctx-index kernel; never linked into the product.
 *
 * Pinning: tests/test_oracle_model.py compares, symbol by symbol (kind, value, flat prior index), with what
 * the reference's own MacroblockModel returned while compressing whole streams (oracle/_ref/ref_dump hooks
 * getNonzerosPrior4x4/8x8, getACPrior4x4/8x8, getLumaDCIntPrior, getChromaDCIntPrior), and with the committed
 * fixtures generated from those runs.
 */
#include <string.h>
#include "oracle_model.h"
#include "../include/lh264.h"

/* zig-zag scans, decode_slice.cpp:2034-2049 (kzz16 / kzz64) = H.264 Table 8-? frame scans */
static const uint8_t kZz16[16] = {0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15};
static const uint8_t kZz64[64] = {
  0, 1, 5, 6, 14, 15, 27, 28, 2, 4, 7, 13, 16, 26, 29, 42, 3, 8, 12, 17, 25, 30, 41, 43, 9, 11, 18, 24, 31, 40, 44, 53,
  10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60, 21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63
};

/* MacroblockModel::encodeMacroblockType, macroblock_model.cpp:647-679 */
int orc_model_mb_type_code (int t) {
  switch (t) {
  case LH264_MB_I4x4: return 0;
  case LH264_MB_I16x16: return 1;
  case LH264_MB_I8x8: return 2;
  case LH264_MB_P16x16: return 3;
  case LH264_MB_P16x8: return 4;
  case LH264_MB_P8x16: return 5;
  case LH264_MB_P8x8: return 6;
  case LH264_MB_P8x8REF0: return 7;
  case LH264_MB_IPCM: return 8;
  case 0x400: return 9;
  case 0x4000: return 10;
  default: return 11;      /* SKIP and 0 */
  }
}

/* DecodedMacroblock::countSubblockNonzeros, decoded_macroblock.h:84-91: all 16 entries of the block */
void orc_model_nnz24 (const int16_t levels[384], uint8_t nnz[24]) {
  for (int b = 0; b < 24; b++) {
    int n = 0;
    for (int i = 0; i < 16; i++) n += levels[b * 16 + i] != 0;
    nnz[b] = (uint8_t)n;
  }
}

static inline int min_i (int a, int b) { return a < b ? a : b; }
static inline int clamp04 (int v) { return v < 0 ? 0 : (v > 4 ? 4 : v); }
static inline int cnt8 (const uint8_t* n, int i) { return n[i] + n[i + 1] + n[i + 2] + n[i + 3]; }

/* encode4x4<N>, decode_slice.cpp:2059-2094 with getNonzerosPrior{4x4,8x8} (macroblock_model.cpp:474-548) and
 * getACPrior{4x4,8x8} (:550-594).  priorCoef's left/above are always 0 (its body is commented out, :210-250),
 * so the last two prior coordinates are the constant 2. */
static int emit_block (const int16_t* ac, int n, int emit_dc, int st, int mbc, int color, int past, int left, int above, orc_sym_t* out) {
  const uint8_t* zz = n == 16 ? kZz16 : kZz64;
  int nonzeros = 0, k = 0;
  for (int i = emit_dc ? 0 : 1; i < n; i++) nonzeros += ac[zz[i]] != 0;
  out[k].kind = n == 16 ? ORC_SYM_NZ4 : ORC_SYM_NZ8;
  out[k].value = (int16_t)nonzeros;
  out[k].prior = (uint32_t) ((((((st * 16 + mbc) * 3 + color) * 3 + min_i (2, past)) * 3 + min_i (2, left)) * 3) + min_i (2, above));
  out[k].pad = 0;
  k++;
  int left_nz = nonzeros, prev = 0, prev2 = 0, emitted = 0;
  for (int i = 0; i < n; i++) {
    if (i == 0 && !emit_dc) continue;
    if (left_nz == 0) continue;
    const int c = ac[zz[i]];
    const uint32_t outer = (uint32_t) ((((st * 16 + mbc) * 3 + color) * n) + emitted);
    const uint32_t inner = (uint32_t) ((((min_i (4, left_nz) * 5 + clamp04 (prev + 2)) * 5 + clamp04 (prev2 + 2)) * 5 + 2) * 5 + 2);
    out[k].kind = n == 16 ? ORC_SYM_AC4 : ORC_SYM_AC8;
    out[k].value = (int16_t)c;
    out[k].prior = outer * 3125u + inner;
    out[k].pad = 0;
    k++;
    prev2 = prev; prev = c; emitted++;
    if (c) left_nz--;
  }
  return k;
}

/* the coefficient part of WelsDecodeSliceForNonRecoding, decode_slice.cpp:2393-2434 */
int orc_model_mb_symbols (const int16_t levels[384], int mb_type, int slice_type, int cbp, int t8,
                          const uint8_t* nl, const uint8_t* na, const uint8_t* np, orc_sym_t* out) {
  uint8_t cur[24];
  static const uint8_t zero[24] = {0};
  orc_model_nnz24 (levels, cur);
  const uint8_t* L = nl ? nl : zero, *A = na ? na : zero, *P = np ? np : zero;
  const int mbc = orc_model_mb_type_code (mb_type), st = slice_type;
  const int cbpl = cbp & 15, cbpc = cbp >> 4;
  int k = 0;
  const int i16 = mb_type == LH264_MB_I16x16;
  if (i16) for (int i = 0; i < 16; i++) {          /* getLumaDCIntPrior: lumaDCIntPriors[i][slice][mbtype] */
      out[k].kind = ORC_SYM_LUMA_DC; out[k].value = levels[i * 16]; out[k].prior = (uint32_t) ((i * 5 + st) * 16 + mbc); out[k].pad = 0; k++;
    }
  const int cdc = (cbpc == 1 || cbpc == 2);
  if (cdc) for (int i = 0; i < 8; i++) {
      out[k].kind = ORC_SYM_CHROMA_DC; out[k].value = levels[256 + i * 16]; out[k].prior = (uint32_t) ((i * 5 + st) * 16 + mbc); out[k].pad = 0; k++;
    }
  for (int i8 = 0; i8 < 4; i8++) {
    if (!(cbpl & (1 << i8))) continue;
    if (t8) {
      const int s = i8;     /* getNonzerosPrior8x8: neighbours by 8x8 index */
      const int past = cnt8 (P, i8 * 4);
      const int left = (s & 1) == 0 ? cnt8 (L, (s + 1) * 4) : cnt8 (cur, (s - 1) * 4);
      const int above = (s & 2) == 0 ? cnt8 (A, (s + 2) * 4) : cnt8 (cur, (s - 2) * 4);
      k += emit_block (levels + i8 * 64, 64, !i16, st, mbc, 0, past, left, above, out + k);
    } else {
      for (int j = 0; j < 4; j++) {
        const int i = i8 * 4 + j;   /* storage (z-order) index, used by the reference as if it were raster */
        const int past = P[i];
        const int left = (i & 3) == 0 ? L[i + 3] : cur[i - 1];
        const int above = i < 4 ? A[i + 12] : cur[i - 4];
        k += emit_block (levels + i * 16, 16, !i16, st, mbc, 0, past, left, above, out + k);
      }
    }
  }
  if (cbpc == 2) {
    for (int i = 0; i < 8; i++) {
      const int color = i < 4 ? 1 : 2;
      const int past = P[16 + i];
      const int left = (i & 1) == 0 ? L[16 + i + 1] : cur[16 + i - 1];
      const int above = (i & 2) == 0 ? A[16 + i + 2] : cur[16 + i - 2];
      k += emit_block (levels + 256 + i * 16, 16, !cdc, st, mbc, color, past, left, above, out + k);
    }
  }
  return k;
}

/* whole-frame drivers (used by bench.py's cpu_baseline so the timed CPU path has no Python in its loop) */
void orc_model_frame_nnz (const lh264_mb_t* mbs, const int16_t* levels, int n_mbs, const uint8_t* past, uint8_t* cur) {
  for (int k = 0; k < n_mbs; k++) {
    const int t = mbs[k].mb_type;
    if (t == LH264_MB_SKIP || t == 0) {
      if (past) memcpy (cur + k * 24, past + k * 24, 24); else memset (cur + k * 24, 0, 24);
    } else orc_model_nnz24 (levels + (size_t)k * 384, cur + k * 24);
  }
}
long orc_model_frame_symbols (const lh264_mb_t* mbs, const lh264_slice_t* slices, const int16_t* levels, int mb_w, int mb_h,
                              const uint8_t* cur, const uint8_t* past, orc_sym_t* out /* [n][432] */, uint16_t* n_out) {
  long total = 0;
  for (int k = 0; k < mb_w * mb_h; k++) {
    const int t = mbs[k].mb_type;
    int n = 0;
    if (!(t == LH264_MB_SKIP || t == LH264_MB_IPCM || t == 0))
      n = orc_model_mb_symbols (levels + (size_t)k * 384, t, slices[mbs[k].slice_id].slice_type, mbs[k].cbp, mbs[k].flags & LH264_MBF_T8x8,
                                (k % mb_w) ? cur + (k - 1) * 24 : 0, k >= mb_w ? cur + (k - mb_w) * 24 : 0, past ? past + k * 24 : 0,
                                out + (size_t)k * ORC_MAX_SYMS);
    n_out[k] = (uint16_t)n;
    total += n;
  }
  return total;
}
