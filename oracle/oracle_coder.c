/* oracle_coder.c - TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Plain-C restatement of the compress direction of the reference's recompressor.
 * References are to /root/reference/codec/decoder/core/:
 *   DynProb, Branch<n>, the int priors                   inc/compression_stream.h:87-241
 *   ArithmeticCodedOutput::emitBit / emitBits / emitUnary / emitBitsZeroToPow2Inclusive   :407-487
 *   CompressionStream::emitInt / emitUEGkInt             :523-591
 *   vpx_write / vpx_start_encode / vpx_stop_encode       inc/bitwriter.h:35-105, src/bitwriter.cpp:17-37
 *   the per-macroblock emit code                          src/decode_slice.cpp:2174-2473 (WelsDecodeSliceForNonRecoding)
 *   encode4x4                                             src/decode_slice.cpp:2059-2094
 *   the prior tables and their indices                   inc/macroblock_model.h:36-136, src/macroblock_model.cpp:370-760
 *   FreqImage (PAST / LEFT / ABOVE)                       inc/decoded_macroblock.h:106-192, Neighbors::init macroblock_model.cpp:9-44
 * The reference keeps its prior tables dense (several GB); here a table cell is created when first touched.
 */
#include "oracle_coder.h"
#include "oracle_model.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* billing.h:6-55 */
enum { T_SKIP = 1, T_SKIP_END = 2, T_CBPC = 3, T_CBPL = 4, T_QPL = 6, T_MB_TYPE = 7, T_T8 = 8, T_REF = 9, T_8x8 = 10, T_16x16 = 11,
       T_PRED_MODE = 13, T_SUB_MB = 14, T_MVX = 15, T_MVY = 16, T_LDC = 17, T_CRDC = 18,
       T_LAC_0_EOB = 19, T_LAC_0_BITMASK, T_LAC_0_EXP, T_LAC_0_RES, T_LAC_0_SIGN,
       T_LAC_N_EOB = 24, T_LAC_N_BITMASK, T_LAC_N_EXP, T_LAC_N_RES, T_LAC_N_SIGN,
       T_CRAC_EOB = 29, T_CRAC_BITMASK, T_CRAC_EXP, T_CRAC_RES, T_CRAC_SIGN, T_PADBYTE = 69, N_TAGS = 72 };

/* wels_common_defs.h */
enum { MBT_I4x4 = 0x01, MBT_I16x16 = 0x02, MBT_I8x8 = 0x04, MBT_16x16 = 0x08, MBT_16x8 = 0x10, MBT_8x16 = 0x20, MBT_8x8 = 0x40,
       MBT_8x8_REF0 = 0x80, MBT_SKIP = 0x100, MBT_IPCM = 0x200 };
enum { SUB_8x8 = 1, SUB_8x4 = 2, SUB_4x8 = 4, SUB_4x4 = 8 };

/* ---- DynProb (compression_stream.h:87-115) ------------------------------------------------------------------------ */
typedef struct { uint32_t c[2]; uint8_t prob; } dynprob_t;
static void dp_init (dynprob_t* p) { p->c[0] = p->c[1] = 0; p->prob = 128; }
static void dp_update (dynprob_t* p, int bit) {
  p->c[bit]++;
  p->prob = (uint8_t) ((256 * (p->c[0] + 1)) / (p->c[1] + p->c[0] + 2));
  if (p->c[0] + p->c[1] > 512) { p->c[0] = (p->c[0] + 1) >> 1; p->c[1] = (p->c[1] + 1) >> 1; }
}

/* ---- bool coder (libvpx, bitwriter.h) ------------------------------------------------------------------------------ */
typedef struct { unsigned lowvalue, range; int count; unsigned pos; uint8_t* buf; unsigned cap; int used; } writer_t;
static const uint8_t k_norm[256] = {
  0, 7, 6, 6, 5, 5, 5, 5, 4, 4, 4, 4, 4, 4, 4, 4, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3,
  2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2,
  1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
  1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
  0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
  0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
  0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
  0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0
};
static void w_start (writer_t* w) { w->lowvalue = 0; w->range = 255; w->count = -24; w->pos = 0; w->used = 1; }
static void w_write (writer_t* w, int bit, int probability) {
  if (w->pos + 16 > w->cap) { w->cap = w->cap ? w->cap * 2 : 1024; w->buf = (uint8_t*)realloc (w->buf, w->cap); }
  unsigned split = 1 + (((w->range - 1) * (unsigned)probability) >> 8);
  int count = w->count;
  unsigned range = split, lowvalue = w->lowvalue;
  if (bit) { lowvalue += split; range = w->range - split; }
  unsigned shift = k_norm[range];
  range <<= shift;
  count += (int)shift;
  if (count >= 0) {
    int offset = (int)shift - count;
    if ((lowvalue << (offset - 1)) & 0x80000000u) {
      int x = (int)w->pos - 1;
      while (x >= 0 && w->buf[x] == 0xff) { w->buf[x] = 0; x--; }
      w->buf[x] += 1;
    }
    w->buf[w->pos++] = (uint8_t) (lowvalue >> (24 - offset));
    lowvalue <<= offset;
    shift = (unsigned)count;
    lowvalue &= 0xffffff;
    count -= 8;
  }
  lowvalue <<= shift;
  w->count = count; w->lowvalue = lowvalue; w->range = range;
}
static void w_stop (writer_t* w) {
  for (int i = 0; i < 32; i++) w_write (w, 0, 128);
  if ((w->buf[w->pos - 1] & 0xe0) == 0xc0) w->buf[w->pos++] = 0;
}

/* ---- sparse prior tables ------------------------------------------------------------------------------------------- */
enum { TB_MBTYPE, TB_MVD, TB_MODE8, TB_LDC, TB_CDC, TB_NZ4, TB_NZ8, TB_AC4, TB_AC8, TB_SKIPRUN, TB_QPL, TB_SUBMB, TB_NUMREF, TB_CBPC,
       TB_CBPL, TB_STOP, TB_T8, TB_PREDMODE, TB_COUNT };
/* DynProbs per table cell */
static const int k_cell[TB_COUNT] = { 15, 14, 8, 9, 9, 8, 8, 13, 13, 511, 128, 255, 15, 3, 15, 1, 1, 15 };

typedef struct { uint64_t key; uint32_t off; } slot_t;
typedef struct {
  slot_t* slots; uint32_t n_slots, n_used;
  dynprob_t* pool; uint32_t pool_n, pool_cap;
} store_t;

static dynprob_t* store_get (store_t* s, int table, uint32_t index) {
  const uint64_t key = ((uint64_t) (table + 1) << 40) | index;
  if (s->n_used * 2 >= s->n_slots) {            /* grow */
    uint32_t nn = s->n_slots ? s->n_slots * 2 : 4096;
    slot_t* ns = (slot_t*)calloc (nn, sizeof (slot_t));
    for (uint32_t i = 0; i < s->n_slots; i++) if (s->slots[i].key) {
      uint32_t h = (uint32_t) ((s->slots[i].key * 0x9E3779B97F4A7C15ull) >> 32) & (nn - 1);
      while (ns[h].key) h = (h + 1) & (nn - 1);
      ns[h] = s->slots[i];
    }
    free (s->slots); s->slots = ns; s->n_slots = nn;
  }
  uint32_t h = (uint32_t) ((key * 0x9E3779B97F4A7C15ull) >> 32) & (s->n_slots - 1);
  while (s->slots[h].key && s->slots[h].key != key) h = (h + 1) & (s->n_slots - 1);
  if (!s->slots[h].key) {
    const int n = k_cell[table];
    if (s->pool_n + n > s->pool_cap) { s->pool_cap = s->pool_cap ? s->pool_cap * 2 : 65536; while (s->pool_n + n > s->pool_cap) s->pool_cap *= 2; s->pool = (dynprob_t*)realloc (s->pool, s->pool_cap * sizeof (dynprob_t)); }
    s->slots[h].key = key; s->slots[h].off = s->pool_n; s->n_used++;
    for (int i = 0; i < n; i++) dp_init (&s->pool[s->pool_n + i]);
    s->pool_n += n;
  }
  return s->pool + s->slots[h].off;    /* valid until the next store_get */
}

/* ---- the FreqImage: what the model keeps of every macroblock (decoded_macroblock.h:4-34) ---------------------------- */
typedef struct {
  uint8_t initialized, zeroed, is_skipped;
  uint8_t cbp_c, cbp_l, chroma_mode, luma16_mode;
  uint8_t nnz[24];
  uint16_t cached_skips;
  uint32_t mb_type, num_ref;
  int8_t ipm[8];             /* pIntraPredMode[mb][0..7] of the decoder (dec_frame.h), for PredIntra4x4Mode */
} cell_t;

struct orc_coder {
  store_t st;
  writer_t w[N_TAGS];
  dynprob_t test_prob;       /* ArithmeticCodedOutput::TEST_PROB: ONE adaptive probability shared by every raw bit of every tag */
  cell_t* img[2]; int img_w, img_h, cur, last_frame_id, prior_valid;
  int8_t* ipm; uint8_t* mbclass; int ipm_n;       /* decoder-side pIntraPredMode[][8] and intra NxN flag, persistent */
  int keep_trace; long tr_n, tr_cap; uint8_t* tr_tag; uint8_t* tr_prob; uint8_t* tr_bit;
  char err[256];
};

orc_coder_t* orc_coder_new (int keep_trace) {
  orc_coder_t* c = (orc_coder_t*)calloc (1, sizeof (*c));
  dp_init (&c->test_prob);
  c->keep_trace = keep_trace;
  return c;
}
void orc_coder_free (orc_coder_t* c) {
  if (!c) return;
  for (int i = 0; i < N_TAGS; i++) free (c->w[i].buf);
  free (c->st.slots); free (c->st.pool); free (c->img[0]); free (c->img[1]); free (c->ipm); free (c->mbclass);
  free (c->tr_tag); free (c->tr_prob); free (c->tr_bit);
  free (c);
}
const char* orc_coder_error (orc_coder_t* c) { return c->err; }

/* ---- emit primitives ------------------------------------------------------------------------------------------------ */
static void emit_bit (orc_coder_t* c, int tag, int bit, dynprob_t* p) {          /* ArithmeticCodedOutput::emitBit :407-436 */
  writer_t* w = &c->w[tag];
  if (!w->used) w_start (w);
  if (c->keep_trace) {
    if (c->tr_n == c->tr_cap) {
      c->tr_cap = c->tr_cap ? c->tr_cap * 2 : 1 << 16;
      c->tr_tag = (uint8_t*)realloc (c->tr_tag, c->tr_cap); c->tr_prob = (uint8_t*)realloc (c->tr_prob, c->tr_cap); c->tr_bit = (uint8_t*)realloc (c->tr_bit, c->tr_cap);
    }
    c->tr_tag[c->tr_n] = (uint8_t)tag; c->tr_prob[c->tr_n] = p->prob; c->tr_bit[c->tr_n] = (uint8_t)bit; c->tr_n++;
  }
  w_write (w, bit, p->prob);
  dp_update (p, bit);
}
static void emit_raw (orc_coder_t* c, int tag, int bit) { emit_bit (c, tag, bit, &c->test_prob); }
static void emit_raw_bits (orc_coder_t* c, int tag, unsigned data, int nbits) {   /* emitBits(uint16_t, int) :441-448 */
  for (int i = 0; i < nbits; i++) emit_raw (c, tag, (data >> (nbits - 1 - i)) & 1);
}
/* Branch<nbits>: MSB first; a node's array holds itself, then its whole 0-subtree, then its 1-subtree (:117-166) */
static void emit_tree (orc_coder_t* c, int tag, unsigned data, int nbits, dynprob_t* arr) {
  unsigned off = 0;
  for (int n = nbits; n >= 1; n--) {
    const int bit = (data >> (n - 1)) & 1;
    emit_bit (c, tag, bit, arr + off);
    const unsigned children = (1u << (n - 1)) - 1;
    off += bit ? 1 + children : 1;
  }
}
/* emitBitsZeroToPow2Inclusive<nbits> :455-463: priors has 1 << nbits entries */
static void emit_pow2 (orc_coder_t* c, int tag, unsigned data, int nbits, dynprob_t* priors, unsigned preferred) {
  emit_bit (c, tag, data != preferred, priors);
  if (data != preferred) emit_tree (c, tag, data > preferred ? data - 1 : data, nbits, priors + 1);
}
/* UnaryIntPrior<N>::at :170-186; N == 0: a fresh probability every time, never remembered */
static void emit_unary (orc_coder_t* c, int tag, int data, dynprob_t* pri, int n, int early_termination) {
  for (int i = 0; i < data; i++) {
    if (n == 0) { dynprob_t t; dp_init (&t); emit_bit (c, tag, 1, &t); }
    else emit_bit (c, tag, 1, pri + (i < n - 1 ? i : n - 1));
    if (i == early_termination - 1) return;
  }
  if (n == 0) { dynprob_t t; dp_init (&t); emit_bit (c, tag, 0, &t); }
  else emit_bit (c, tag, 0, pri + (data < n - 1 ? data : n - 1));
}
/* a prior of the IntPrior family laid out as: [zero][sign] exponent[E] mantissa[M] */
typedef struct { dynprob_t* zero; dynprob_t* sign; dynprob_t* exponent; int E; dynprob_t* mantissa; int M; int order; } intprior_t;
static void emit_int (orc_coder_t* c, int data, const intprior_t* p, int tag_exp, int tag_man, int tag_zero, int tag_sign) {   /* :523-572 */
  if (p->zero) { emit_bit (c, tag_zero, data == 0, p->zero); if (data == 0) return; }
  if (p->sign) { emit_bit (c, tag_sign, data > 0, p->sign); if (data < 0) data = -data; }
  data--;
  int log2 = 0;
  const int data_high = 1 + (data >> p->order);
  while ((2 << log2) <= data_high) log2++;
  emit_unary (c, tag_exp, log2, p->exponent, p->E, -1);
  int bits[40], nb = 0;
  for (int i = log2 - 1; i >= 0; i--) bits[nb++] = (data_high >> i) & 1;
  for (int i = p->order - 1; i >= 0; i--) bits[nb++] = (data >> i) & 1;
  int lo = 0, hi = p->M;
  for (int i = 0; i < nb; i++) {
    if (hi > lo) {
      const int mid = (hi + lo) / 2;
      emit_bit (c, tag_man, bits[i], p->mantissa + mid);
      if (bits[i]) lo = mid + 1; else hi = mid;
    } else emit_raw (c, tag_man, bits[i]);
  }
}
/* UEGkIntPrior<N, M, E, Mant, Order> laid out as: zero, sign, first[M], second = {zero, exponent[E], mantissa[Mant]} */
static void emit_uegk (orc_coder_t* c, int data, dynprob_t* cell, int N, int M, int E, int Mant, int order,
                       int tag_exp, int tag_man, int tag_zero, int tag_sign) {      /* :575-591 */
  emit_bit (c, tag_zero, data == 0, cell + 0);
  if (data == 0) return;
  emit_bit (c, tag_sign, data < 0, cell + 1);
  if (data < 0) data = -data;
  emit_unary (c, tag_man, data - 1, cell + 2, M, N);
  if (data - 1 >= N) {
    intprior_t p; p.zero = cell + 2 + M; p.sign = NULL; p.exponent = cell + 2 + M + 1; p.E = E; p.mantissa = p.exponent + E; p.M = Mant; p.order = order;
    emit_int (c, data - 1 - N, &p, tag_exp, tag_man, tag_zero, tag_sign);
  }
}

/* ---- model helpers ------------------------------------------------------------------------------------------------- */
static int type_code (unsigned t) { return orc_model_mb_type_code ((int)t); }        /* encodeMacroblockType macroblock_model.cpp:647-679 */
static unsigned swizzle_sign (int v) { return v >= 0 ? ((unsigned)v << 1) & 0xffff : ((((unsigned) (-v - 1)) << 1) | 1) & 0xffff; }

/* FreqImage::updateFrame decoded_macroblock.h:119-166: flip on a new frame id, then recompute isSkipped / cachedSkips of
 * the PREVIOUS picture from its coefficients (every slice calls this) */
static void image_update_frame (orc_coder_t* c, int frame_id) {
  if (frame_id != c->last_frame_id) { c->cur = c->cur ? 0 : 1; c->last_frame_id = frame_id; }
  cell_t* f = c->img[1 - c->cur];
  const int n = c->img_w * c->img_h;
  unsigned contiguous = 0;
  for (int i = 0; i < n; i++) {
    if (f[i].zeroed) { f[i].is_skipped = 1; contiguous++; }
    else {
      f[i].is_skipped = 0;
      for (unsigned j = 0; j < contiguous; j++) f[i - j].cached_skips = (uint16_t)contiguous;
      contiguous = 0;
    }
  }
}

static const uint8_t k_scan8[16] = { 9, 10, 17, 18, 11, 12, 19, 20, 25, 26, 33, 34, 27, 28, 35, 36 };   /* g_kuiScan8: 1 + bx + 8 * (1 + by), z-order */
static const uint8_t k_cache30[16] = { 7, 8, 13, 14, 9, 10, 15, 16, 19, 20, 25, 26, 21, 22, 27, 28 };    /* g_kuiCache30ScanIdx */
static const uint8_t k_scan4[16] = { 0, 1, 4, 5, 2, 3, 6, 7, 8, 9, 12, 13, 10, 11, 14, 15 };             /* g_kuiScan4 */

int orc_coder_picture (orc_coder_t* c, int mb_w, int mb_h, int frame_num, const uint16_t* mb_types, const int16_t* levels,
                       const orc_rtd_t* rtd, const uint8_t* avail, const orc_slice_info_t* slices, int n_slices) {
  const int n = mb_w * mb_h;
  if (c->ipm_n != n) {
    free (c->ipm); free (c->mbclass);
    c->ipm = (int8_t*)calloc ((size_t)n, 8); c->mbclass = (uint8_t*)calloc ((size_t)n, 1); c->ipm_n = n;
  }
  for (int s = 0; s < n_slices; s++) {
    const orc_slice_info_t* S = &slices[s];
    /* WelsDecodeSlice decode_slice.cpp:3031-3046 */
    c->prior_valid = 1;
    image_update_frame (c, frame_num);
    if (c->img_w != mb_w || c->img_h != mb_h) {
      c->prior_valid = 0;
      c->img_w = mb_w; c->img_h = mb_h;
      for (int b = 0; b < 2; b++) { free (c->img[b]); c->img[b] = (cell_t*)calloc ((size_t)n, sizeof (cell_t)); }
    }
    cell_t* cur = c->img[c->cur];
    cell_t* last = c->img[1 - c->cur];
    int skip_state = -1, orig_skipped = -1, mb_in_slice = 0;
    unsigned cached_qp = 0; int last_nonzero_dqp = 0;
    const int is_p = S->slice_type == 0;
    for (int k = S->first_mb; k < S->first_mb + S->n_mbs && k < n; k++, mb_in_slice++) {
      const int x = k % mb_w, y = k / mb_w;
      /* Neighbors::init */
      const cell_t* nl = (x > 0 && cur[k - 1].initialized) ? &cur[k - 1] : NULL;
      const cell_t* na = (y > 0 && cur[k - mb_w].initialized) ? &cur[k - mb_w] : NULL;
      const cell_t* np = (c->prior_valid && last[k].initialized) ? &last[k] : NULL;
      const int write_skip_run = skip_state == -1;
      const unsigned type = mb_types[k];
      const int skipped = is_p && type == MBT_SKIP;
      /* WelsDecodeMbCavlcPSlice :3894-3915: how the run counter moves */
      int mb_skip_run = 0;
      if (is_p && ((S->transform8x8_pps >> 2) & 1)) {
        /* CABAC (WelsDecodeMbCabacPSlice :1164-1200, :2208-2210): a skip flag per macroblock; the "run" is 0 or 1 and is written
         * for every macroblock (write_skip_run is forced below by leaving skip_state at -1) */
        mb_skip_run = skipped ? 1 : 0;
      } else if (is_p) {
        if (skip_state == -1) {       /* mb_skip_run read here: the number of skipped macroblocks from this one on */
          int run = 0;
          while (k + run < S->first_mb + S->n_mbs && mb_types[k + run] == MBT_SKIP) run++;
          skip_state = run;
        }
        mb_skip_run = skip_state;
        if (skip_state-- == 0) { /* coded macroblock: state is -1 again */ }
      }
      const int has_stop = (k == S->first_mb + S->n_mbs - 1);
      const int initial_skip = mb_skip_run > 0 && orig_skipped == -1;
      const int final_skip = mb_skip_run == 0 && orig_skipped != -1;
      if (write_skip_run) {
        /* getSkipRunPrior macroblock_model.cpp:374-387; mb->uiMbType is still 0 here -> type code 11 */
        int prior = 0;
        if (np) prior = np->cached_skips / 8 + (np->cached_skips % 8 ? 1 : 0);
        emit_tree (c, T_SKIP, (unsigned)mb_skip_run, 9, store_get (&c->st, TB_SKIPRUN, (uint32_t) (prior * 16 + 11)));
      }
      if (mb_skip_run == 1) emit_bit (c, T_SKIP_END, has_stop, store_get (&c->st, TB_STOP, (uint32_t) (mb_in_slice < 2048 ? mb_in_slice : 2047)));
      const int write_block = mb_skip_run == 0;
      if (initial_skip) orig_skipped = mb_skip_run;
      if (final_skip) orig_skipped = -1;
      (void)skipped;
      if (write_block) {
        const orc_rtd_t* R = &rtd[k];
        if (!R->have) { snprintf (c->err, sizeof (c->err), "macroblock %d is coded but has no record", k); return -1; }
        const int mbc = type_code (R->mb_type);
        const int16_t* lv = levels + (size_t)k * 384;
        emit_bit (c, T_SKIP_END, has_stop, store_get (&c->st, TB_STOP, (uint32_t) (mb_in_slice < 2048 ? mb_in_slice : 2047)));
        {   /* getMacroblockTypePrior :441-465 */
          int prior = 15, prev = 15;
          if (na) prior = type_code (na->mb_type);
          if (nl) prior = type_code (nl->mb_type);
          if (np) prev = type_code (np->mb_type);
          prior += prev;
          emit_tree (c, T_MB_TYPE, (unsigned)mbc, 4, store_get (&c->st, TB_MBTYPE, (uint32_t) (prior * 2 + is_p)));
        }
        emit_tree (c, T_CBPL, R->cbp_c, 2, store_get (&c->st, TB_CBPC, (uint32_t) ((np ? np->cbp_c : 0) * 16 + mbc)));
        emit_tree (c, T_CBPL, R->cbp_l, 4, store_get (&c->st, TB_CBPL, (uint32_t) ((np ? np->cbp_l : 0) * 16 + mbc)));
        {
          const int dqp = (int)R->luma_qp - (int)cached_qp;
          const int sidx = last_nonzero_dqp < 0 ? 0 : (last_nonzero_dqp == 0 ? 1 : 2);
          emit_pow2 (c, T_QPL, swizzle_sign (dqp), 7, store_get (&c->st, TB_QPL, (uint32_t) ((mb_in_slice == 0) * 3 + sidx)), 0);
          cached_qp = R->luma_qp;
          if (dqp) last_nonzero_dqp = dqp;
        }
        emit_tree (c, T_REF, R->num_ref_idx_l0, 4, store_get (&c->st, TB_NUMREF, (uint32_t) ((np ? np->num_ref : 0) * 16 + mbc)));
        int ref_bits = 0;
        while ((1u << ref_bits) < R->num_ref_idx_l0) ref_bits++;        /* ceil(log2(n)) */
        {   /* getChromaI8x8ModePrior / getLumaI16x16ModePrior :611-645 (both use chromaI8x8ModePriors) */
          int pr = 7;
          if (np) { pr = np->chroma_mode; if (pr >= 6) pr = 6; }
          emit_pow2 (c, T_8x8, R->chroma_mode, 3, store_get (&c->st, TB_MODE8, (uint32_t)pr), (unsigned)pr);
          pr = 7;
          if (np) { pr = np->luma16_mode; if (pr >= 6) pr = 6; }
          emit_pow2 (c, T_16x16, R->luma16_mode, 3, store_get (&c->st, TB_MODE8, (uint32_t)pr), (unsigned)pr);
        }
        int8_t* my_ipm = c->ipm + (size_t)k * 8;
        if (R->mb_type == MBT_I4x4 || R->mb_type == MBT_I8x8) {
          /* the decoder's intra mode cache: WelsFillCacheConstrain0IntraNxN parse_mb_syn_cavlc.cpp:204-248 */
          int8_t cache[48];
          memset (cache, 0, sizeof (cache));
          /* GetNeighborAvailMbType: a neighbour counts when it belongs to the same slice (consecutive macroblocks, no FMO) */
          (void)avail;
          int left_av = x > 0 && k - 1 >= S->first_mb, top_av = k - mb_w >= S->first_mb,
              topleft_av = x > 0 && k - mb_w - 1 >= S->first_mb;
          if (!((S->transform8x8_pps >> 1) & 1)) {
            if (top_av && c->mbclass[k - mb_w]) memcpy (cache + 1, c->ipm + (size_t) (k - mb_w) * 8, 4);
            else memset (cache + 1, top_av ? 2 : -1, 4);
            if (left_av && c->mbclass[k - 1]) {
              const int8_t* li = c->ipm + (size_t) (k - 1) * 8;
              cache[8] = li[4]; cache[16] = li[5]; cache[24] = li[6]; cache[32] = li[3];
            } else cache[8] = cache[16] = cache[24] = cache[32] = (int8_t) (left_av ? 2 : -1);
          } else {
            /* constrained_intra_pred: WelsFillCacheConstrain1IntraNxN parse_mb_syn_cavlc.cpp:157-202 (only an I4x4 neighbour lends
             * its modes; I16x16 / I_PCM count as DC, the rest as unavailable) and WelsMapNxNNeighToSampleConstrain1
             * decode_slice.cpp:419-438 (samples from intra neighbours only) */
            const unsigned lt = left_av ? mb_types[k - 1] : 0, tt = top_av ? mb_types[k - mb_w] : 0, tlt = topleft_av ? mb_types[k - mb_w - 1] : 0;
            const unsigned intra = MBT_I4x4 | MBT_I16x16 | MBT_I8x8 | MBT_IPCM;
            if (top_av && tt == MBT_I4x4) memcpy (cache + 1, c->ipm + (size_t) (k - mb_w) * 8, 4);
            else memset (cache + 1, (tt == MBT_I16x16 || tt == MBT_IPCM) ? 2 : -1, 4);
            if (left_av && lt == MBT_I4x4) {
              const int8_t* li = c->ipm + (size_t) (k - 1) * 8;
              cache[8] = li[4]; cache[16] = li[5]; cache[24] = li[6]; cache[32] = li[3];
            } else cache[8] = cache[16] = cache[24] = cache[32] = (int8_t) ((lt == MBT_I16x16 || lt == MBT_IPCM) ? 2 : -1);
            left_av = left_av && (lt & intra); top_av = top_av && (tt & intra); topleft_av = topleft_av && (tlt & intra);
          }
          if (R->mb_type == MBT_I4x4) {
            int sample_av[30];
            memset (sample_av, 0, sizeof (sample_av));
            /* WelsMapNxNNeighToSampleNormal: row 0 = top neighbours (0 = top-left, 1..4 top), column 0 = left */
            sample_av[0] = topleft_av;
            for (int i = 1; i <= 4; i++) sample_av[i] = top_av;
            for (int i = 1; i <= 4; i++) sample_av[6 * i] = left_av;
            for (int i = 0; i < 16; i++) {
              const int top_mode = cache[k_scan8[i] - 8], left_mode = cache[k_scan8[i] - 1];
              const int pred = (left_mode == -1 || top_mode == -1) ? 2 : (left_mode < top_mode ? left_mode : top_mode);   /* PredIntra4x4Mode */
              const int idx = k_cache30[i];
              sample_av[idx] = 1;
              const int la = sample_av[idx - 1], ta = sample_av[idx - 6], lta = sample_av[idx - 7];
              const int avail_idx = (!!la << 2) | (!!ta << 1) | !!lta;
              emit_tree (c, T_PRED_MODE, (unsigned) (uint16_t)R->pred_mode[i] & 15, 4,
                         store_get (&c->st, TB_PREDMODE, (uint32_t) ((mbc * 8 + avail_idx) * 9 + pred)));
              cache[k_scan8[i]] = R->pred_mode[i];
            }
          } else {
            for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) cache[k_scan8[(i << 2) + j]] = R->pred_mode[i];
          }
          memcpy (my_ipm, cache + 1 + 8 * 4, 4);
          my_ipm[4] = cache[4 + 8 * 1]; my_ipm[5] = cache[4 + 8 * 2]; my_ipm[6] = cache[4 + 8 * 3];
          c->mbclass[k] = 1;
        } else c->mbclass[k] = 0;
        if (R->mb_type == MBT_I8x8) {
          for (int i = 0; i < 4; i++)
            emit_tree (c, T_PRED_MODE, (unsigned) (uint8_t)R->pred_mode[i] & 15, 4, store_get (&c->st, TB_PREDMODE, (uint32_t) ((mbc * 8 + 6) * 9 + 1)));
          for (int i = 0; i < 4; i++) emit_tree (c, T_SUB_MB, R->sub_type[i], 8, store_get (&c->st, TB_SUBMB, (uint32_t)mbc));
          for (int i = 0; i < 4; i++) emit_raw_bits (c, T_REF, (uint8_t)R->ref_idx[i], ref_bits);
        } else if (R->mb_type == MBT_8x8 || R->mb_type == MBT_8x8_REF0) {
          for (int i = 0; i < 4; i++) emit_tree (c, T_SUB_MB, R->sub_type[i], 8, store_get (&c->st, TB_SUBMB, (uint32_t)mbc));
          if (R->mb_type == MBT_8x8) for (int i = 0; i < 4; i++) emit_raw_bits (c, T_REF, (uint8_t)R->ref_idx[i], ref_bits);
          for (int i = 0; i < 4; i++) {
            int idxs[4], ni = 0;
            switch (R->sub_type[i]) {
            case SUB_8x8: idxs[ni++] = k_scan4[i << 2]; break;
            case SUB_8x4: for (int j = 0; j < 2; j++) idxs[ni++] = k_scan4[(i << 2) + (j << 1)]; break;
            case SUB_4x8: for (int j = 0; j < 2; j++) idxs[ni++] = k_scan4[(i << 2) + j]; break;
            case SUB_4x4: for (int j = 0; j < 4; j++) idxs[ni++] = k_scan4[(i << 2) + j]; break;
            default: snprintf (c->err, sizeof (c->err), "macroblock %d: sub type %d", k, R->sub_type[i]); return -1;
            }
            for (int j = 0; j < ni; j++) {      /* writeMv :2126-2133; the MVD prior is indexed by the RAW mb type and the block */
              emit_uegk (c, R->mvd[idxs[j]][0], store_get (&c->st, TB_MVD, (uint32_t) (R->mb_type * 16 + idxs[j])), 9, 4, 3, 4, 3, T_MVX, T_MVX, T_MVX, T_MVX);
              emit_uegk (c, R->mvd[idxs[j]][1], store_get (&c->st, TB_MVD, (uint32_t) (R->mb_type * 16 + idxs[j])) , 9, 4, 3, 4, 3, T_MVY, T_MVY, T_MVY, T_MVY);
            }
          }
        } else if (R->mb_type == MBT_8x16 || R->mb_type == MBT_16x8) {
          for (int i = 0; i < 2; i++) emit_raw_bits (c, T_REF, (uint8_t)R->ref_idx[i], ref_bits);
          for (int i = 0; i < 2; i++) {
            const int bi = R->mb_type == MBT_16x8 ? i * 8 : i * 2;
            emit_uegk (c, R->mvd[bi][0], store_get (&c->st, TB_MVD, (uint32_t) (R->mb_type * 16 + bi)), 9, 4, 3, 4, 3, T_MVX, T_MVX, T_MVX, T_MVX);
            emit_uegk (c, R->mvd[bi][1], store_get (&c->st, TB_MVD, (uint32_t) (R->mb_type * 16 + bi)), 9, 4, 3, 4, 3, T_MVY, T_MVY, T_MVY, T_MVY);
          }
        } else if (R->mb_type == MBT_16x16) {
          emit_raw_bits (c, T_REF, (uint8_t)R->ref_idx[0], ref_bits);
          emit_uegk (c, R->mvd[0][0], store_get (&c->st, TB_MVD, (uint32_t) (R->mb_type * 16)), 9, 4, 3, 4, 3, T_MVX, T_MVX, T_MVX, T_MVX);
          emit_uegk (c, R->mvd[0][1], store_get (&c->st, TB_MVD, (uint32_t) (R->mb_type * 16)), 9, 4, 3, 4, 3, T_MVY, T_MVY, T_MVY, T_MVY);
        }
        {   /* needParseTransformSize8x8 decoded_macroblock.h:72-86 */
          int no_sub_lt8 = 1;
          if (R->mb_type == MBT_8x8 || R->mb_type == MBT_8x8_REF0) for (int i = 0; i < 4; i++) no_sub_lt8 &= R->sub_type[i] == SUB_8x8;
          const int is_inter = (R->mb_type & (MBT_16x16 | MBT_16x8 | MBT_8x16 | MBT_8x8 | MBT_8x8_REF0 | MBT_SKIP)) != 0;
          if ((((R->mb_type >= MBT_16x16 && R->mb_type <= MBT_8x16) || no_sub_lt8) && is_inter && R->cbp_l > 0 && (S->transform8x8_pps & 1)))
            emit_bit (c, T_T8, R->t8, store_get (&c->st, TB_T8, (uint32_t) (mbc * 128 + R->luma_qp)));
        }
        /* coefficients: the a8 restatement supplies (kind, prior index, value) in emission order */
        orc_sym_t syms[ORC_MAX_SYMS];
        const int ns = orc_model_mb_symbols (lv, (int)R->mb_type, S->slice_type, R->cbp_l | (R->cbp_c << 4), R->t8,
                                             nl ? nl->nnz : NULL, na ? na->nnz : NULL, np ? np->nnz : NULL, syms);
        for (int i = 0; i < ns; i++) {
          const orc_sym_t* sy = &syms[i];
          if (sy->kind == ORC_SYM_LUMA_DC || sy->kind == ORC_SYM_CHROMA_DC) {
            dynprob_t* cell = store_get (&c->st, sy->kind == ORC_SYM_LUMA_DC ? TB_LDC : TB_CDC, sy->prior);
            intprior_t p; p.exponent = cell; p.E = 3; p.mantissa = cell + 3; p.M = 4; p.zero = cell + 7; p.sign = cell + 8; p.order = 0;
            const int t = sy->kind == ORC_SYM_LUMA_DC ? T_LDC : T_CRDC;
            emit_int (c, sy->value, &p, t, t, t, t);
          } else if (sy->kind == ORC_SYM_NZ4 || sy->kind == ORC_SYM_NZ8) {
            dynprob_t* cell = store_get (&c->st, sy->kind == ORC_SYM_NZ4 ? TB_NZ4 : TB_NZ8, sy->prior);
            intprior_t p; p.exponent = cell; p.E = 3; p.mantissa = cell + 3; p.M = 4; p.zero = cell + 7; p.sign = NULL; p.order = 0;
            const int color = (int) ((sy->prior / 27) % 3);
            const int t = color ? T_CRAC_EOB : T_LAC_0_EOB;
            emit_int (c, sy->value, &p, t, t, t, t);
          } else {
            const int nco = sy->kind == ORC_SYM_AC4 ? 16 : 64;
            const uint32_t outer = sy->prior / 3125;
            const int emitted = (int) (outer % nco), color = (int) ((outer / nco) % 3), code = (int) ((outer / nco / 3) % 16);
            const int first = color == 0 && emitted == 0 && code != 1;     /* scan position 0 (only when the DC is coded here) */
            const int base = color ? T_CRAC_EOB : (first ? T_LAC_0_EOB : T_LAC_N_EOB);
            if (!c->w[base + 2].used) w_start (&c->w[base + 2]);     /* encode4x4 bills to tag(..._EXP) before coding: the stream exists even if no escape ever uses it */
            emit_uegk (c, sy->value, store_get (&c->st, sy->kind == ORC_SYM_AC4 ? TB_AC4 : TB_AC8, sy->prior), 14, 4, 2, 4, 0,
                       base + 2 /*EXP*/, base + 3 /*RES*/, base + 1 /*BITMASK*/, base + 4 /*SIGN*/);
          }
        }
      }
      /* decode_slice.cpp:3098-3109: the image entry */
      if (!write_block) {
        cur[k] = last[k];
        c->mbclass[k] = 0;
      } else {
        const orc_rtd_t* R = &rtd[k];
        cell_t* e = &cur[k];
        memset (e, 0, sizeof (*e));
        e->initialized = 1;
        e->cbp_c = R->cbp_c; e->cbp_l = R->cbp_l; e->chroma_mode = R->chroma_mode; e->luma16_mode = R->luma16_mode;
        e->mb_type = R->mb_type; e->num_ref = R->num_ref_idx_l0;
        const int16_t* lv = levels + (size_t)k * 384;
        orc_model_nnz24 (lv, e->nnz);
        e->zeroed = 1;
        for (int i = 0; i < 384; i++) if (lv[i]) { e->zeroed = 0; break; }
      }
    }
    if (S->pad_bits) emit_raw_bits (c, T_PADBYTE, (unsigned)S->pad_value, S->pad_bits);     /* decode_slice.cpp:3133-3148 */
  }
  return 0;
}

/* Code a flat list of symbols in the product's record format (include/lh264.h: lh264_ctx_sym_t with the LH264_SYM_* kinds
 * 0..10; SPLICE markers must have been resolved by the caller).  The checker of the product's host symbolizer. */
int orc_coder_symbols (orc_coder_t* c, const orc_sym_t* syms, long n) {
  static const int tree_bits[TB_COUNT] = { 4, 0, 3, 0, 0, 0, 0, 0, 0, 9, 7, 8, 4, 2, 4, 0, 0, 4 };
  for (long i = 0; i < n; i++) {
    const orc_sym_t* sy = &syms[i];
    const int table = (int) (sy->prior >> 27);
    const uint32_t index = sy->prior & 0x7ffffffu;
    const int tag = sy->pad;
    switch (sy->kind) {
    case ORC_SYM_LUMA_DC: case ORC_SYM_CHROMA_DC: {
      dynprob_t* cell = store_get (&c->st, sy->kind == ORC_SYM_LUMA_DC ? TB_LDC : TB_CDC, sy->prior);
      intprior_t p; p.exponent = cell; p.E = 3; p.mantissa = cell + 3; p.M = 4; p.zero = cell + 7; p.sign = cell + 8; p.order = 0;
      const int t = sy->kind == ORC_SYM_LUMA_DC ? T_LDC : T_CRDC;
      emit_int (c, sy->value, &p, t, t, t, t);
      break; }
    case ORC_SYM_NZ4: case ORC_SYM_NZ8: {
      dynprob_t* cell = store_get (&c->st, sy->kind == ORC_SYM_NZ4 ? TB_NZ4 : TB_NZ8, sy->prior);
      intprior_t p; p.exponent = cell; p.E = 3; p.mantissa = cell + 3; p.M = 4; p.zero = cell + 7; p.sign = NULL; p.order = 0;
      const int t = ((sy->prior / 27) % 3) ? T_CRAC_EOB : T_LAC_0_EOB;
      emit_int (c, sy->value, &p, t, t, t, t);
      break; }
    case ORC_SYM_AC4: case ORC_SYM_AC8: {
      const int nco = sy->kind == ORC_SYM_AC4 ? 16 : 64;
      const uint32_t outer = sy->prior / 3125;
      const int emitted = (int) (outer % nco), color = (int) ((outer / nco) % 3), code = (int) ((outer / nco / 3) % 16);
      const int first = color == 0 && emitted == 0 && code != 1;
      const int base = color ? T_CRAC_EOB : (first ? T_LAC_0_EOB : T_LAC_N_EOB);
      if (!c->w[base + 2].used) w_start (&c->w[base + 2]);
      emit_uegk (c, sy->value, store_get (&c->st, sy->kind == ORC_SYM_AC4 ? TB_AC4 : TB_AC8, sy->prior), 14, 4, 2, 4, 0, base + 2, base + 3, base + 1, base + 4);
      break; }
    case 6: /* TREE */ emit_tree (c, tag, (unsigned) (uint16_t)sy->value, tree_bits[table], store_get (&c->st, table, index)); break;
    case 7: /* POW2 */ emit_pow2 (c, tag, (unsigned) (uint16_t)sy->value, tree_bits[table], store_get (&c->st, table, index), table == TB_MODE8 ? index : 0); break;
    case 8: /* BIT */ emit_bit (c, tag, sy->value != 0, store_get (&c->st, table, index)); break;
    case 9: /* RAW */ emit_raw_bits (c, tag, (unsigned) (uint16_t)sy->value, (int)sy->prior); break;
    case 10: /* MVD */ emit_uegk (c, sy->value, store_get (&c->st, TB_MVD, index), 9, 4, 3, 4, 3, tag, tag, tag, tag); break;
    default: snprintf (c->err, sizeof (c->err), "symbol %ld: unknown kind %d", i, sy->kind); return -1;
    }
  }
  return 0;
}

void orc_coder_finish (orc_coder_t* c) {
  for (int t = 0; t < N_TAGS; t++) if (c->w[t].used) w_stop (&c->w[t]);
}
int orc_coder_tag (orc_coder_t* c, int tag, const uint8_t** bytes) {
  if (tag < 0 || tag >= N_TAGS || !c->w[tag].used) { *bytes = NULL; return 0; }
  *bytes = c->w[tag].buf;
  return (int)c->w[tag].pos;
}
long orc_coder_trace (orc_coder_t* c, const uint8_t** tags, const uint8_t** probs, const uint8_t** bits) {
  *tags = c->tr_tag; *probs = c->tr_prob; *bits = c->tr_bit;
  return c->tr_n;
}
