// pip_symbols.cpp - see pip_symbols.h.  Fresh implementation; line references are to the reference's
// decoder/core/src/decode_slice.cpp (DS), decoder/core/src/macroblock_model.cpp (MM), decoder/core/inc/decoded_macroblock.h (DM).
#include "pip_symbols.h"
#include <string.h>

namespace lh264host {
namespace {

// tag ids, billing.h:6-55
enum { TAG_SKIP = 1, TAG_SKIP_END = 2, TAG_CBPL = 4, TAG_QPL = 6, TAG_MB_TYPE = 7, TAG_T8 = 8, TAG_REF = 9, TAG_8x8 = 10, TAG_16x16 = 11,
       TAG_PRED_MODE = 13, TAG_SUB_MB = 14, TAG_MVX = 15, TAG_MVY = 16, TAG_PADBYTE = 69 };

int type_code (uint32_t t) {          // MacroblockModel::encodeMacroblockType MM:647-679
  switch (t) {
  case LH264_MB_I4x4: return 0;  case LH264_MB_I16x16: return 1;  case LH264_MB_I8x8: return 2;
  case LH264_MB_P16x16: return 3;  case LH264_MB_P16x8: return 4;  case LH264_MB_P8x16: return 5;
  case LH264_MB_P8x8: return 6;  case LH264_MB_P8x8REF0: return 7;  case LH264_MB_IPCM: return 8;
  default: return 11;
  }
}
uint32_t rd32 (const uint8_t* p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }

struct Out {
  PoolVec<lh264_ctx_sym_t>& v;
  void put (int kind, int table, uint32_t index, int value, int tag) {
    lh264_ctx_sym_t s; s.prior = LH264_PRIOR (table, index); s.value = (int16_t)value; s.kind = (uint8_t)kind; s.pad = (uint8_t)tag;
    v.push_back (s);
  }
  void raw (int value, int nbits, int tag) {
    if (nbits <= 0) return;
    lh264_ctx_sym_t s; s.prior = (uint32_t)nbits; s.value = (int16_t)value; s.kind = LH264_SYM_RAW; s.pad = (uint8_t)tag;
    v.push_back (s);
  }
};

const uint8_t kScan8[16] = {9, 10, 17, 18, 11, 12, 19, 20, 25, 26, 33, 34, 27, 28, 35, 36};   // 1 + bx + 8 * (1 + by), blocks in z-order
const uint8_t kCache30[16] = {7, 8, 13, 14, 9, 10, 15, 16, 19, 20, 25, 26, 21, 22, 27, 28};    // 1 + bx + 6 * (1 + by)
const uint8_t kZ2Raster[16] = {0, 1, 4, 5, 2, 3, 6, 7, 8, 9, 12, 13, 10, 11, 14, 15};

}  // namespace

// FreqImage::updateFrame DM:119-166: the buffers flip when frame_num changes; then isSkipped / cachedSkips of the PREVIOUS
// picture are recomputed from its coefficients (on every slice)
void Symbolizer::update_frame (int frame_id) {
  if (frame_id != last_frame_id_) { cur_ = cur_ ? 0 : 1; last_frame_id_ = frame_id; }
  std::vector<Cell>& f = img_[1 - cur_];
  unsigned run = 0;
  for (size_t i = 0; i < f.size(); i++) {
    if (f[i].zeroed) run++;
    else {
      for (unsigned j = 0; j < run; j++) f[i - j].cached_skips = (uint16_t)run;   // (sic) the run's first entry is not reached, entry i is
      run = 0;
    }
  }
}

void Symbolizer::picture (FrameOut& f) {
  const int w = f.mb_w, n = f.mb_w * f.mb_h;
  if ((int)ipm_.size() != n * 8) { ipm_.assign ((size_t)n * 8, 0); nxn_.assign (n, 0); }
  // symbols in coding order; a macroblock's run is [start, start + cnt)
  // straight into the picture's list: slices in raster order (all but ASO streams) leave it in macroblock order already
  PoolVec<lh264_ctx_sym_t>& flat = f.syn_syms; flat.clear();
  start_.assign (n, 0); cnt_.assign (n, 0);
  for (size_t si = 0; si < f.slices.size(); si++) {
    const lh264_slice_t& S = f.slices[si];
    const SliceSyn& X = f.slice_syn[si];
    // WelsDecodeSlice DS:3031-3046
    bool prior_valid = true;
    update_frame (f.frame_num);
    if (img_w_ != f.mb_w || img_h_ != f.mb_h) {
      prior_valid = false;
      img_w_ = f.mb_w; img_h_ = f.mb_h;
      img_[0].assign (n, Cell()); img_[1].assign (n, Cell());
    }
    std::vector<Cell>& cur = img_[cur_];
    std::vector<Cell>& last = img_[1 - cur_];
    const bool is_p = S.slice_type == 0, cabac = (X.flags & 1) != 0;
    const int end = S.first_mb + S.n_mbs;
    int skip_state = -1, mb_in_slice = 0, cached_qp = 0, last_nonzero_dqp = 0;
    for (int k = S.first_mb; k < end && k < n; k++, mb_in_slice++) {
      Out o = {flat};
      start_[k] = (uint32_t)flat.size();
      struct Close { PoolVec<lh264_ctx_sym_t>& v; uint32_t& s; uint32_t& c; ~Close() { c = (uint32_t)v.size() - s; } } close_run = {flat, start_[k], cnt_[k]};
      const int x = k % w;
      const Cell* nl = (x > 0 && cur[k - 1].initialized) ? &cur[k - 1] : nullptr;              // Neighbors::init MM:9-44
      const Cell* na = (k >= w && cur[k - w].initialized) ? &cur[k - w] : nullptr;
      const Cell* np = (prior_valid && last[k].initialized) ? &last[k] : nullptr;
      const bool write_skip_run = skip_state == -1;
      int mb_skip_run = 0;
      if (is_p && cabac) {                           // CABAC: a skip flag per macroblock; the run is written every time, DS:1186,2208-2210
        mb_skip_run = f.mbs[k].mb_type == LH264_MB_SKIP ? 1 : 0;
      } else if (is_p) {                             // WelsDecodeMbCavlcPSlice DS:3894-3915
        if (skip_state == -1) { int run = 0; while (k + run < end && f.mbs[k + run].mb_type == LH264_MB_SKIP) run++; skip_state = run; }
        mb_skip_run = skip_state;
        skip_state--;                                // a coded macroblock leaves -1: the next one reads a new run
      }
      const int has_stop = k == end - 1;
      const uint32_t stop_idx = (uint32_t) (mb_in_slice < 2048 ? mb_in_slice : 2047);
      if (write_skip_run) {                          // getSkipRunPrior MM:374-387; the macroblock type is still unset: code 11
        const int pr = np ? np->cached_skips / 8 + (np->cached_skips % 8 ? 1 : 0) : 0;
        o.put (LH264_SYM_TREE, LH264_TB_SKIPRUN, (uint32_t) (pr * 16 + 11), mb_skip_run, TAG_SKIP);
      }
      if (mb_skip_run == 1) o.put (LH264_SYM_BIT, LH264_TB_STOP, stop_idx, has_stop, TAG_SKIP_END);
      if (mb_skip_run != 0) {                        // skipped: the image entry is the PAST one, DS:3104-3105
        cur[k] = last[k];
        nxn_[k] = 0;
        continue;
      }
      const MbSyn& R = f.syn[k];
      const uint32_t type = R.mb_type;
      const int mbc = type_code (type);
      o.put (LH264_SYM_BIT, LH264_TB_STOP, stop_idx, has_stop, TAG_SKIP_END);
      {                                              // getMacroblockTypePrior MM:441-465
        int prior = 15, prev = 15;
        if (na) prior = type_code (na->mb_type);
        if (nl) prior = type_code (nl->mb_type);
        if (np) prev = type_code (np->mb_type);
        o.put (LH264_SYM_TREE, LH264_TB_MBTYPE, (uint32_t) ((prior + prev) * 2 + (is_p ? 1 : 0)), mbc, TAG_MB_TYPE);
      }
      o.put (LH264_SYM_TREE, LH264_TB_CBPC, (uint32_t) ((np ? np->cbp_c : 0) * 16 + mbc), R.cbp_c, TAG_CBPL);   // (sic) billed to CBPL, DS:2261
      o.put (LH264_SYM_TREE, LH264_TB_CBPL, (uint32_t) ((np ? np->cbp_l : 0) * 16 + mbc), R.cbp_l, TAG_CBPL);
      {                                              // DS:2268-2276, getQPLPrior MM:388-391
        const int dqp = (int)R.luma_qp - cached_qp;
        const int sidx = last_nonzero_dqp < 0 ? 0 : (last_nonzero_dqp == 0 ? 1 : 2);
        const unsigned sw = dqp >= 0 ? ((unsigned)dqp << 1) & 0xffff : ((((unsigned) (-dqp - 1)) << 1) | 1) & 0xffff;   // swizzle_sign MM:719-725
        o.put (LH264_SYM_POW2, LH264_TB_QPL, (uint32_t) ((mb_in_slice == 0 ? 1 : 0) * 3 + sidx), (int)sw, TAG_QPL);
        cached_qp = R.luma_qp;
        if (dqp) last_nonzero_dqp = dqp;
      }
      o.put (LH264_SYM_TREE, LH264_TB_NUMREF, (uint32_t) ((np ? np->num_ref : 0) * 16 + mbc), (int)R.num_ref_idx_l0, TAG_REF);
      int ref_bits = 0;
      while ((1u << ref_bits) < R.num_ref_idx_l0) ref_bits++;
      {                                              // getChromaI8x8ModePrior / getLumaI16x16ModePrior MM:611-645: one table for both
        int pr = 7;
        if (np) { pr = np->chroma_mode; if (pr >= 6) pr = 6; }
        o.put (LH264_SYM_POW2, LH264_TB_MODE8, (uint32_t)pr, R.chroma_mode, TAG_8x8);
        pr = 7;
        if (np) { pr = np->luma16_mode; if (pr >= 6) pr = 6; }
        o.put (LH264_SYM_POW2, LH264_TB_MODE8, (uint32_t)pr, R.luma16_mode, TAG_16x16);
      }
      int8_t* my_ipm = &ipm_[(size_t)k * 8];
      if (type == LH264_MB_I4x4 || type == LH264_MB_I8x8) {
        // the decoder's intra-mode cache (WelsFillCacheConstrain0IntraNxN / ...Constrain1IntraNxN parse_mb_syn_cavlc.cpp:157-248): a
        // neighbour counts when it lies in the same slice; with constrained_intra_pred only an I4x4 neighbour lends its modes,
        // I16x16 / I_PCM count as DC and everything else as unavailable, and samples are available from intra neighbours only
        // (WelsMapNxNNeighToSampleConstrain1 DS:419-438)
        int8_t cache[48];
        memset (cache, 0, sizeof (cache));
        const bool cip = (X.flags & 2) != 0;
        bool left_av = x > 0 && k - 1 >= S.first_mb, top_av = k - w >= S.first_mb, topleft_av = x > 0 && k - w - 1 >= S.first_mb;
        const uint32_t lt = left_av ? f.mbs[k - 1].mb_type : 0, tt = top_av ? f.mbs[k - w].mb_type : 0, tlt = topleft_av ? f.mbs[k - w - 1].mb_type : 0;
        if (!cip) {
          if (top_av && nxn_[k - w]) memcpy (cache + 1, &ipm_[(size_t) (k - w) * 8], 4);
          else memset (cache + 1, top_av ? 2 : -1, 4);
          if (left_av && nxn_[k - 1]) {
            const int8_t* li = &ipm_[(size_t) (k - 1) * 8];
            cache[8] = li[4]; cache[16] = li[5]; cache[24] = li[6]; cache[32] = li[3];
          } else cache[8] = cache[16] = cache[24] = cache[32] = (int8_t) (left_av ? 2 : -1);
        } else {
          if (top_av && tt == LH264_MB_I4x4) memcpy (cache + 1, &ipm_[(size_t) (k - w) * 8], 4);
          else memset (cache + 1, (tt == LH264_MB_I16x16 || tt == LH264_MB_IPCM) ? 2 : -1, 4);
          if (left_av && lt == LH264_MB_I4x4) {
            const int8_t* li = &ipm_[(size_t) (k - 1) * 8];
            cache[8] = li[4]; cache[16] = li[5]; cache[24] = li[6]; cache[32] = li[3];
          } else cache[8] = cache[16] = cache[24] = cache[32] = (int8_t) ((lt == LH264_MB_I16x16 || lt == LH264_MB_IPCM) ? 2 : -1);
          left_av = left_av && (lt & LH264_MB_INTRA); top_av = top_av && (tt & LH264_MB_INTRA); topleft_av = topleft_av && (tlt & LH264_MB_INTRA);
        }
        if (type == LH264_MB_I4x4) {                 // DS:2301-2318
          int sample_av[30];
          memset (sample_av, 0, sizeof (sample_av));
          sample_av[0] = topleft_av;
          for (int i = 1; i <= 4; i++) { sample_av[i] = top_av; sample_av[6 * i] = left_av; }
          for (int i = 0; i < 16; i++) {
            const int top_mode = cache[kScan8[i] - 8], left_mode = cache[kScan8[i] - 1];
            const int pred = (left_mode == -1 || top_mode == -1) ? 2 : (left_mode < top_mode ? left_mode : top_mode);
            const int idx = kCache30[i];
            sample_av[idx] = 1;
            const int avail_idx = (sample_av[idx - 1] ? 4 : 0) | (sample_av[idx - 6] ? 2 : 0) | (sample_av[idx - 7] ? 1 : 0);
            o.put (LH264_SYM_TREE, LH264_TB_PREDMODE, (uint32_t) ((mbc * 8 + avail_idx) * 9 + pred), R.pred_mode[i] & 15, TAG_PRED_MODE);
            cache[kScan8[i]] = R.pred_mode[i];
          }
        } else {
          for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) cache[kScan8[(i << 2) + j]] = R.pred_mode[i];
        }
        memcpy (my_ipm, cache + 1 + 8 * 4, 4);
        my_ipm[4] = cache[4 + 8 * 1]; my_ipm[5] = cache[4 + 8 * 2]; my_ipm[6] = cache[4 + 8 * 3];
        nxn_[k] = 1;
      } else nxn_[k] = 0;
      auto mvd = [&] (int blk) {                     // writeMv DS:2126-2133: the prior is indexed by the RAW type and the block
        o.put (LH264_SYM_MVD, LH264_TB_MVD, type * 16 + (uint32_t)blk, R.mvd[blk][0], TAG_MVX);
        o.put (LH264_SYM_MVD, LH264_TB_MVD, type * 16 + (uint32_t)blk, R.mvd[blk][1], TAG_MVY);
      };
      if (type == LH264_MB_I8x8) {                   // DS:2320-2335
        for (int i = 0; i < 4; i++) o.put (LH264_SYM_TREE, LH264_TB_PREDMODE, (uint32_t) ((mbc * 8 + 6) * 9 + 1), R.pred_mode[i] & 15, TAG_PRED_MODE);
        for (int i = 0; i < 4; i++) o.put (LH264_SYM_TREE, LH264_TB_SUBMB, (uint32_t)mbc, R.sub_type[i], TAG_SUB_MB);
        for (int i = 0; i < 4; i++) o.raw ((uint8_t)R.ref_idx[i], ref_bits, TAG_REF);
      } else if (type == LH264_MB_P8x8 || type == LH264_MB_P8x8REF0) {
        for (int i = 0; i < 4; i++) o.put (LH264_SYM_TREE, LH264_TB_SUBMB, (uint32_t)mbc, R.sub_type[i], TAG_SUB_MB);
        if (type == LH264_MB_P8x8) for (int i = 0; i < 4; i++) o.raw ((uint8_t)R.ref_idx[i], ref_bits, TAG_REF);
        for (int i = 0; i < 4; i++) {
          switch (R.sub_type[i]) {
          case LH264_SUB_8x8: mvd (kZ2Raster[i << 2]); break;
          case LH264_SUB_8x4: for (int j = 0; j < 2; j++) mvd (kZ2Raster[(i << 2) + (j << 1)]); break;
          case LH264_SUB_4x8: for (int j = 0; j < 2; j++) mvd (kZ2Raster[(i << 2) + j]); break;
          default: for (int j = 0; j < 4; j++) mvd (kZ2Raster[(i << 2) + j]); break;
          }
        }
      } else if (type == LH264_MB_P8x16 || type == LH264_MB_P16x8) {
        for (int i = 0; i < 2; i++) o.raw ((uint8_t)R.ref_idx[i], ref_bits, TAG_REF);
        for (int i = 0; i < 2; i++) mvd (type == LH264_MB_P16x8 ? i * 8 : i * 2);
      } else if (type == LH264_MB_P16x16) {
        o.raw ((uint8_t)R.ref_idx[0], ref_bits, TAG_REF);
        mvd (0);
      }
      {                                              // needParseTransformSize8x8 DM:72-86
        bool no_sub_lt8 = true;
        if (type == LH264_MB_P8x8 || type == LH264_MB_P8x8REF0) for (int i = 0; i < 4; i++) no_sub_lt8 = no_sub_lt8 && R.sub_type[i] == LH264_SUB_8x8;
        const bool is_inter = (type & (LH264_MB_P16x16 | LH264_MB_P16x8 | LH264_MB_P8x16 | LH264_MB_P8x8 | LH264_MB_P8x8REF0 | LH264_MB_SKIP)) != 0;
        if (((type >= LH264_MB_P16x16 && type <= LH264_MB_P8x16) || no_sub_lt8) && is_inter && R.cbp_l > 0 && X.transform8x8_pps)
          o.put (LH264_SYM_BIT, LH264_TB_T8, (uint32_t) (mbc * 128 + R.luma_qp), R.t8, TAG_T8);
      }
      { lh264_ctx_sym_t s; s.prior = 0; s.value = 0; s.kind = LH264_SYM_SPLICE; s.pad = 0; flat.push_back (s); }
      // the image entry, DS:3098-3109
      Cell e;
      e.initialized = 1; e.cbp_c = R.cbp_c; e.cbp_l = R.cbp_l; e.chroma_mode = R.chroma_mode; e.luma16_mode = R.luma16_mode;
      e.mb_type = type; e.num_ref = R.num_ref_idx_l0; e.cached_skips = 0;
      e.zeroed = f.lev_nonzero[k] ? 0 : 1;
      cur[k] = e;
    }
    // the alignment bits after the slice's stop bit go to the pad-byte tag, DS:3133-3148
    if (X.pad_bits && end - 1 < n && end - 1 >= S.first_mb && start_[end - 1] + cnt_[end - 1] == flat.size()) {
      Out o = {flat}; o.raw (X.pad_value, X.pad_bits, TAG_PADBYTE); cnt_[end - 1]++;
    }
  }
  f.syn_off.assign ((size_t)n + 1, 0);
  bool in_order = true;
  uint32_t off = 0;
  for (int k = 0; k < n; k++) {
    if (cnt_[k] && start_[k] != off) { in_order = false; break; }
    f.syn_off[k] = off;
    off += cnt_[k];
  }
  if (in_order && off == flat.size()) { f.syn_off[n] = off; return; }
  // slices out of raster order: the runs are put into macroblock order
  flat_.assign (flat.begin(), flat.end());
  f.syn_syms.clear();
  for (int k = 0; k < n; k++) {
    f.syn_off[k] = (uint32_t)f.syn_syms.size();
    f.syn_syms.insert (f.syn_syms.end(), flat_.begin() + start_[k], flat_.begin() + start_[k] + cnt_[k]);
  }
  f.syn_off[n] = (uint32_t)f.syn_syms.size();
}

}  // namespace lh264host
