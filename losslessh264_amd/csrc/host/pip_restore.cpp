// pip_restore.cpp - see pip_restore.h.  Fresh implementation; line references are to the reference's
// decoder/core/src/decode_slice.cpp (DS), decoder/core/src/macroblock_model.cpp (MM), decoder/core/inc/compression_stream.h (CS),
// decoder/core/inc/decoded_macroblock.h (DM), decoder/core/src/decoder.cpp (DC).
#include "pip_restore.h"
#include <stdexcept>
#include <string.h>
#include <algorithm>
#include "h264_parser.h"
#include "h264_tables.h"
#include "h264_vlc_tables.h"
#include "h264_cabac_tables.h"

namespace lh264host {
namespace {

// tag ids, billing.h:6-55
enum { TAG_SKIP = 1, TAG_SKIP_END = 2, TAG_CBPL = 4, TAG_QPL = 6, TAG_MB_TYPE = 7, TAG_T8 = 8, TAG_REF = 9, TAG_8x8 = 10, TAG_16x16 = 11,
       TAG_PRED_MODE = 13, TAG_SUB_MB = 14, TAG_MVX = 15, TAG_MVY = 16, TAG_LDC = 17, TAG_CRDC = 18, TAG_LAC_0 = 19, TAG_LAC_N = 24,
       TAG_CRAC = 29, TAG_PADBYTE = 69, N_TAGS = 72 };

// ---- DynProb, CS:87-115 -------------------------------------------------------------------------------------------------
// floor (num / den) for num < 2^18, 2 <= den <= 516 by a reciprocal table (the division is a third of a decision's cost)
struct RecipTable { uint32_t inv[520]; RecipTable() { inv[0] = inv[1] = 0; for (uint32_t d = 2; d < 520; d++) inv[d] = (uint32_t) (((1ull << 32) + d - 1) / d); } };
inline uint8_t div_prob (uint32_t num, uint32_t den) {
  static const RecipTable T;
  uint32_t q = (uint32_t) (((uint64_t)num * T.inv[den]) >> 32);
  if (q * den > num) q--;                            // ceil(2^32/den) overestimates by less than num/2^32 * ... : at most one too many
  return (uint8_t)q;
}
struct DynProb {
  uint16_t c0 = 0, c1 = 0; uint8_t prob = 128;
  inline void update (int bit) {
    if (bit) c1++; else c0++;
    prob = div_prob (256u * (c0 + 1u), c0 + c1 + 2u);
    if (c0 + c1 > 512) { c0 = (uint16_t) ((c0 + 1) >> 1); c1 = (uint16_t) ((c1 + 1) >> 1); }
  }
};

// ---- the bool decoder (libvpx dboolhuff as vendored by the reference: bitreader.h:77-136, bitreader.cpp:43-108) -----------
struct BoolReader {
  const uint8_t* p = nullptr; const uint8_t* end = nullptr;
  uint64_t value = 0; int count = -8; uint32_t range = 255; bool present = false;
  void fill() {
    int shift = 64 - 8 - (count + 8);
    while (shift >= 0) {
      if (p < end) { count += 8; value |= (uint64_t) (*p++) << shift; shift -= 8; }
      else { count += 0x40000000; break; }               // past the end: zeros
    }
  }
  inline int read (int prob) {
    const uint32_t split = 1 + (((range - 1) * (uint32_t)prob) >> 8);
    if (count < 0) fill();
    const uint64_t bigsplit = (uint64_t)split << 56;
    int bit = 0;
    uint32_t r = split;
    if (value >= bigsplit) { r = range - split; value -= bigsplit; bit = 1; }
    const int shift = r >= 128 ? 0 : __builtin_clz (r) - 24;       // vpx_norm[r]
    r <<= shift;
    range = r; value <<= shift; count -= shift;
    return bit;
  }
};

// ---- the adaptive priors: a cell per (table, index), created when first touched ------------------------------------------
const int kCell[LH264_TB_COUNT] = {15, 14, 8, 9, 9, 8, 8, 13, 13, 511, 128, 255, 15, 3, 15, 1, 1, 15};
const int kTreeBits[LH264_TB_COUNT] = {4, 0, 3, 0, 0, 0, 0, 0, 0, 9, 7, 8, 4, 2, 4, 0, 0, 4};

class PriorStore {
 public:
  PriorStore() { slots_.assign (1u << 16, Slot{0, 0}); }
  DynProb* get (int table, uint32_t index) {          // valid until the next get
    const uint32_t key = LH264_PRIOR (table, index) + 1u;
    if (used_ * 2 >= slots_.size()) grow();
    const size_t mask = slots_.size() - 1;
    size_t h = hash (key) & mask;
    while (slots_[h].key && slots_[h].key != key) h = (h + 1) & mask;
    if (!slots_[h].key) {
      slots_[h].key = key; slots_[h].off = (uint32_t)pool_.size(); used_++;
      pool_.resize (pool_.size() + (size_t)kCell[table]);
    }
    return pool_.data() + slots_[h].off;
  }
 private:
  struct Slot { uint32_t key, off; };                  // key and offset side by side: one cache line per probe
  static inline uint32_t hash (uint32_t k) { return (uint32_t) ((k * 0x9E3779B97F4A7C15ull) >> 32); }
  void grow() {
    std::vector<Slot> s2 (slots_.size() * 2, Slot{0, 0});
    for (const Slot& s : slots_) if (s.key) {
        size_t h = hash (s.key) & (s2.size() - 1);
        while (s2[h].key) h = (h + 1) & (s2.size() - 1);
        s2[h] = s;
      }
    slots_.swap (s2);
  }
  std::vector<Slot> slots_;
  std::vector<DynProb> pool_;
  size_t used_ = 0;
};

// ---- CABAC arithmetic encoding engine, ITU-T H.264 9.3.4.2 (the reference borrows its encoder's: encoder/core/src/set_mb_syn_cabac.cpp) --
struct CabacEnc {
  std::vector<uint8_t> bytes; uint32_t acc = 0; int nacc = 0;
  uint32_t low = 0, range = 510; int outstanding = 0; bool first = true;
  uint8_t state[460];
  void init (int col, int qp) {
    bytes.clear(); acc = 0; nacc = 0; low = 0; range = 510; outstanding = 0; first = true;
    qp = std::min (51, std::max (0, qp));
    for (int i = 0; i < 460; i++) {
      const int m = kCabacInit[i][col][0], n = kCabacInit[i][col][1];
      const int pre = std::min (126, std::max (1, ((m * qp) >> 4) + n));
      state[i] = pre <= 63 ? (uint8_t) ((63 - pre) << 1) : (uint8_t) (((pre - 64) << 1) | 1);
    }
  }
  inline void write_bit (int b) { acc = (acc << 1) | (uint32_t) (b & 1); if (++nacc == 8) { bytes.push_back ((uint8_t)acc); acc = 0; nacc = 0; } }
  inline void put (int b) {                          // PutBit, 9.3.4.2 figure 9-9
    if (first) first = false; else write_bit (b);
    while (outstanding > 0) { write_bit (1 - b); outstanding--; }
  }
  inline void renorm() {
    while (range < 256) {
      if (low < 256) put (0);
      else if (low >= 512) { low -= 512; put (1); }
      else { low -= 256; outstanding++; }
      range <<= 1; low <<= 1;
    }
  }
  inline void encode (int ctx, int bin) {
    uint8_t& s = state[ctx];
    const int st = s >> 1; int mps = s & 1;
    const uint32_t lps = kCabacRangeLps[st][(range >> 6) & 3];
    range -= lps;
    if ((bin & 1) != mps) {
      low += range; range = lps;
      if (st == 0) mps = !mps;
      s = (uint8_t) ((kCabacNextLps[st] << 1) | mps);
    } else s = (uint8_t) ((kCabacNextMps[st] << 1) | mps);
    renorm();
  }
  inline void bypass (int bin) {
    low <<= 1;
    if (bin & 1) low += range;
    if (low >= 1024) { put (1); low -= 1024; }
    else if (low < 512) put (0);
    else { low -= 512; outstanding++; }
  }
  inline void terminate (int bin) {
    range -= 2;
    if (bin) {
      low += range;
      range = 2; renorm();                           // EncodeFlush
      put ((low >> 9) & 1);
      write_bit ((low >> 8) & 1); write_bit (1);       // ((low >> 7) & 3) | 1: the last bit is the rbsp stop bit
      while (nacc) write_bit (0);
    } else renorm();
  }
};
const uint8_t kSig8x8[63] = {0, 1, 2, 3, 4, 5, 5, 4, 4, 3, 3, 4, 4, 4, 5, 5, 4, 4, 4, 4, 3, 3, 6, 7, 7, 7, 8, 9, 10, 9, 8, 7, 7, 6, 11, 12, 13, 11, 6, 7, 8, 9,
                             14, 10, 9, 8, 6, 11, 12, 13, 11, 6, 9, 14, 10, 9, 11, 12, 13, 11, 14, 10, 12};      // Table 9-43
const uint8_t kLast8x8[63] = {0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 3,
                              4, 4, 4, 4, 4, 4, 4, 4, 5, 5, 5, 5, 6, 6, 6, 6, 7, 7, 7, 7, 8, 8, 8};
const int kCatCbf[5] = {0, 4, 8, 12, 16}, kCatMap[5] = {0, 15, 29, 44, 47}, kCatAbs[5] = {0, 10, 20, 30, 39};

inline int z2x (int z) { return (z & 1) | ((z >> 2) & 1) << 1; }
inline int z2y (int z) { return ((z >> 1) & 1) | ((z >> 3) & 1) << 1; }
const uint8_t kScan8[16] = {9, 10, 17, 18, 11, 12, 19, 20, 25, 26, 33, 34, 27, 28, 35, 36};
const uint8_t kCache30[16] = {7, 8, 13, 14, 9, 10, 15, 16, 19, 20, 25, 26, 21, 22, 27, 28};
const uint8_t kZ2Raster[16] = {0, 1, 4, 5, 2, 3, 6, 7, 8, 9, 12, 13, 10, 11, 14, 15};
const int kChromaNzc[2][4] = {{16, 17, 20, 21}, {18, 19, 22, 23}};
// scan position -> index into the block's level array, as the reference's encode4x4 / decode4x4 use them (DS:2034-2052): the
// 4x4 table is the zig-zag scan, the 8x8 table is (sic) the inverse permutation
const uint8_t kZz16[16] = {0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15};
const uint8_t kZz64[64] = {
  0, 1, 5, 6, 14, 15, 27, 28, 2, 4, 7, 13, 16, 26, 29, 42, 3, 8, 12, 17, 25, 30, 41, 43, 9, 11, 18, 24, 31, 40, 44, 53,
  10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60, 21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63
};

int type_code (uint32_t t) {          // MacroblockModel::encodeMacroblockType MM:647-679
  switch (t) {
  case LH264_MB_I4x4: return 0;  case LH264_MB_I16x16: return 1;  case LH264_MB_I8x8: return 2;
  case LH264_MB_P16x16: return 3;  case LH264_MB_P16x8: return 4;  case LH264_MB_P8x16: return 5;
  case LH264_MB_P8x8: return 6;  case LH264_MB_P8x8REF0: return 7;  case LH264_MB_IPCM: return 8;
  default: return 11;
  }
}
const uint32_t kCodeType[9] = {LH264_MB_I4x4, LH264_MB_I16x16, LH264_MB_I8x8, LH264_MB_P16x16, LH264_MB_P16x8, LH264_MB_P8x16,
                               LH264_MB_P8x8, LH264_MB_P8x8REF0, LH264_MB_IPCM};
inline int min2 (int v) { return v < 2 ? v : 2; }
inline int clamp04 (int v) { return v < 0 ? 0 : (v > 4 ? 4 : v); }

struct Cell {                         // what the model remembers of a macroblock (DM:4-34); nnz = countSubblockNonzeros
  uint8_t initialized = 0, zeroed = 0, cbp_c = 0, cbp_l = 0, chroma_mode = 0, luma16_mode = 0;
  uint16_t cached_skips = 0;
  uint32_t mb_type = 0, num_ref = 0;
  uint8_t nnz[24] = {0};
};
struct WState {                       // what the CAVLC writer must know of the macroblocks around (9.2.1, 8.3.1.1)
  int32_t slice = -1; uint32_t mb_type = 0; uint8_t type_class = 0; int8_t ipm[16]; uint8_t nzc[24];
  // CABAC context selection (9.3.3.1.1)
  uint8_t skip = 0, t8 = 0, cbp = 0, chroma_pred = 0; int8_t ref[4]; uint8_t mvd[16][2]; uint32_t cbf = 0;
};
struct MbDec {                        // one decoded macroblock
  uint32_t type = 0; int cbp_c = 0, cbp_l = 0, luma_qp = 0, num_ref = 0, chroma_mode = 0, luma16_mode = 0, t8 = 0;
  int pred_mode[16]; int sub_type[4]; int ref_idx[4]; int mvd[16][2];
  int16_t lev[384];
};

class Restorer {
 public:
  Restorer (const uint8_t* const* tags, const size_t* tag_len, int n_tags, std::string& err) : err_ (err) {
    if (n_tags > LH264_TAG_PCM && tags[LH264_TAG_PCM]) { pcm_ = tags[LH264_TAG_PCM]; pcm_end_ = pcm_ + tag_len[LH264_TAG_PCM]; }
    for (int t = 0; t < N_TAGS && t < n_tags; t++) if (t != LH264_TAG_PCM && tags[t]) { rd_[t].p = tags[t]; rd_[t].end = tags[t] + tag_len[t]; rd_[t].present = true; rd_[t].fill(); }
    build_vlc();
  }
  int run (const uint8_t* d, size_t n, std::vector<uint8_t>& out);

 private:
  std::string& err_;
  bool failed_ = false;
  BoolReader rd_[N_TAGS];
  const uint8_t* pcm_ = nullptr; const uint8_t* pcm_end_ = nullptr;     // samples of the I_PCM macroblocks still to be written
  DynProb test_prob_;                 // ArithmeticCodedInput::TEST_PROB: one adaptive probability shared by the raw bits of all tags
  PriorStore store_;
  Parser hdr_;
  MainStreamWriter w_;
  // model state (as csrc/host/pip_symbols.cpp keeps it on the compress side)
  std::vector<Cell> img_[2];
  int img_w_ = 0, img_h_ = 0, cur_ = 0, last_frame_id_ = 0;
  std::vector<int8_t> ipm_;
  std::vector<uint8_t> nxn_;
  // writer state
  std::vector<WState> ws_;
  int ws_n_ = 0, sid_ = 0;
  // coeff_token by (table, total_coeff, trailing_ones) -> length, code
  uint8_t tok_len_[5][17][4]; uint16_t tok_code_[5][17][4];

  void fail (const std::string& m) { if (!failed_) { failed_ = true; err_ = m; } }

  // ---- scan primitives (the inverses of ArithmeticCodedOutput::emit*, CS:289-351 and CompressionStream::scanInt / scanUEGkInt :607-676)
  inline int scan_bit (int tag, DynProb* p) {
    BoolReader& r = rd_[tag];
    if (!r.present) { fail ("tag stream " + std::to_string (tag) + " is missing"); return 0; }
    const int bit = r.read (p->prob);
    p->update (bit);
    return bit;
  }
  inline int scan_raw (int tag) { return scan_bit (tag, &test_prob_); }
  unsigned scan_raw_bits (int tag, int n) { unsigned v = 0; for (int i = 0; i < n; i++) v = (v << 1) | (unsigned)scan_raw (tag); return v; }
  unsigned scan_tree (int tag, int nbits, DynProb* arr) {      // Branch<n>: a node, its whole 0-subtree, then its 1-subtree
    unsigned off = 0, v = 0;
    for (int n = nbits; n >= 1; n--) {
      const int bit = scan_bit (tag, arr + off);
      v = (v << 1) | (unsigned)bit;
      off += bit ? 1u + ((1u << (n - 1)) - 1u) : 1u;
    }
    return v;
  }
  unsigned scan_pow2 (int tag, int nbits, DynProb* priors, unsigned preferred) {
    if (!scan_bit (tag, priors)) return preferred;
    const unsigned d = scan_tree (tag, nbits, priors + 1);
    return d >= preferred ? d + 1 : d;
  }
  int scan_unary (int tag, DynProb* pri, int n, int early_termination) {
    int i = 0;
    for (;;) {
      int bit;
      if (n == 0) { DynProb t; bit = scan_bit (tag, &t); }
      else bit = scan_bit (tag, pri + (i < n - 1 ? i : n - 1));
      if (!bit) return i;
      i++;
      if (i == early_termination) return i;
      if (i > 70000 || failed_) { fail ("corrupt unary code"); return 0; }
    }
  }
  struct IntPrior { DynProb* zero; DynProb* sign; DynProb* exponent; int E; DynProb* mantissa; int M; int order; };
  int scan_int (const IntPrior& p, int tag_exp, int tag_man, int tag_zero, int tag_sign) {
    if (p.zero && scan_bit (tag_zero, p.zero)) return 0;
    bool positive = true;
    if (p.sign) positive = scan_bit (tag_sign, p.sign) != 0;
    const int log2 = scan_unary (tag_exp, p.exponent, p.E, -1);
    if (log2 > 30) { fail ("corrupt integer code"); return 0; }
    int lo = 0, hi = p.M;
    uint32_t data_high = 1, low = 0;
    for (int i = 0; i < log2 + p.order; i++) {
      int bit;
      if (hi > lo) {
        const int mid = (hi + lo) / 2;
        bit = scan_bit (tag_man, p.mantissa + mid);
        if (bit) lo = mid + 1; else hi = mid;
      } else bit = scan_raw (tag_man);
      if (i < log2) data_high = (data_high << 1) | (uint32_t)bit; else low = (low << 1) | (uint32_t)bit;
    }
    const int data = (int) (((data_high - 1) << p.order) | low) + 1;
    return positive ? data : -data;
  }
  int scan_uegk (DynProb* cell, int N, int M, int E, int Mant, int order, int tag_exp, int tag_man, int tag_zero, int tag_sign) {
    if (scan_bit (tag_zero, cell + 0)) return 0;
    const int neg = scan_bit (tag_sign, cell + 1);
    int v = scan_unary (tag_man, cell + 2, M, N);
    if (v >= N) {
      IntPrior p; p.zero = cell + 2 + M; p.sign = nullptr; p.exponent = cell + 2 + M + 1; p.E = E; p.mantissa = p.exponent + E; p.M = Mant; p.order = order;
      v = N + scan_int (p, tag_exp, tag_man, tag_zero, tag_sign);
    }
    v += 1;
    return neg ? -v : v;
  }
  unsigned tree (int tag, int table, uint32_t index) { return scan_tree (tag, kTreeBits[table], store_.get (table, index)); }

  // ---- model ---------------------------------------------------------------------------------------------------------------
  void update_frame (int frame_id);
  bool decode_slice (const Parser::HeaderInfo& H);
  void decode_coeffs (MbDec& m, int st, int mbc, const Cell* nl, const Cell* na, const Cell* np, Cell& e);
  // ---- CAVLC writer ----------------------------------------------------------------------------------------------------------
  void build_vlc();
  void put_ue (uint32_t v) { int n = 0; while (((v + 1) >> n) > 1) n++; w_.emit_bits (0, n); w_.emit_bits (v + 1, n + 1); }
  void put_se (int v) { put_ue (v > 0 ? (uint32_t) (2 * v - 1) : (uint32_t) (-2 * v)); }
  int pred_intra_mode (int k, int bx, int by, int w, int sid, bool cip) const;
  void write_residual_block (const int* lv, int maxc, int nC, int& total_out);
  void write_mb (const Parser::HeaderInfo& H, int k, const MbDec& m, int& qp_prev);
  CabacEnc ce_;
  void cabac_residual (int k, int cat, int blk, int plane, bool cur_intra, const int* lv, int maxc);
  void write_mb_cabac (const Parser::HeaderInfo& H, int k, const MbDec* m /* null: P_Skip */, int& qp_prev, int& last_dqp);
};

// FreqImage::updateFrame DM:119-166
void Restorer::update_frame (int frame_id) {
  if (frame_id != last_frame_id_) { cur_ = cur_ ? 0 : 1; last_frame_id_ = frame_id; }
  std::vector<Cell>& f = img_[1 - cur_];
  unsigned run = 0;
  for (size_t i = 0; i < f.size(); i++) {
    if (f[i].zeroed) run++;
    else {
      for (unsigned j = 0; j < run; j++) f[i - j].cached_skips = (uint16_t)run;
      run = 0;
    }
  }
}

// the coefficient symbols of one macroblock, in the order and with the priors of csrc/lh264_ctx.hip:ctx_symbols_kernel
// (decode4x4 DS:2096-2124, getNonzerosPrior / getACPrior / get*DCIntPrior MM:466-594)
void Restorer::decode_coeffs (MbDec& m, int st, int mbc, const Cell* nl, const Cell* na, const Cell* np, Cell& e) {
  static const uint8_t kZero[24] = {0};
  const uint8_t* Lf = nl ? nl->nnz : kZero; const uint8_t* Ab = na ? na->nnz : kZero; const uint8_t* Pa = np ? np->nnz : kZero;
  uint8_t* C = e.nnz;
  int16_t* lev = m.lev;
  const bool i16 = m.type == LH264_MB_I16x16;
  const bool cdc = m.cbp_c == 1 || m.cbp_c == 2;
  auto dc = [&] (int table, int i, int tag) {
    DynProb* cell = store_.get (table, (uint32_t) ((i * 5 + st) * 16 + mbc));
    IntPrior p; p.exponent = cell; p.E = 3; p.mantissa = cell + 3; p.M = 4; p.zero = cell + 7; p.sign = cell + 8; p.order = 0;
    return scan_int (p, tag, tag, tag, tag);
  };
  if (i16) for (int i = 0; i < 16; i++) lev[i * 16] = (int16_t)dc (LH264_TB_LDC, i, TAG_LDC);
  if (cdc) for (int i = 0; i < 8; i++) lev[256 + i * 16] = (int16_t)dc (LH264_TB_CDC, i, TAG_CRDC);
  // e.nnz[b] = the nonzero levels of 4x4 block b (a separately coded DC included), counted as they are decoded
  for (int b = 0; b < 24; b++) C[b] = (uint8_t) (lev[b * 16] != 0);
  for (int b = 0; b < 24; b++) {
    const bool luma = b < 16;
    const bool big = luma && m.t8;
    bool coded = luma ? ((m.cbp_l >> (b >> 2)) & 1) != 0 : m.cbp_c == 2;
    if (big && (b & 3)) coded = false;
    if (coded && !failed_) {
      const bool emit_dc = luma ? !i16 : !cdc;
      const int start = emit_dc ? 0 : 1, color = luma ? 0 : (b < 20 ? 1 : 2), nco = big ? 64 : 16;
      int past, left, above;
      if (big) {
        const int s = b >> 2;
        auto c8 = [] (const uint8_t* p, int i) { return p[i] + p[i + 1] + p[i + 2] + p[i + 3]; };
        past = c8 (Pa, b);
        left = (s & 1) == 0 ? c8 (Lf, (s + 1) * 4) : c8 (C, (s - 1) * 4);
        above = (s & 2) == 0 ? c8 (Ab, (s + 2) * 4) : c8 (C, (s - 2) * 4);
      } else if (luma) {
        past = Pa[b];
        left = (b & 3) == 0 ? Lf[b + 3] : C[b - 1];
        above = b < 4 ? Ab[b + 12] : C[b - 4];
      } else {
        const int i = b - 16;
        past = Pa[b];
        left = (i & 1) == 0 ? Lf[b + 1] : C[b - 1];
        above = (i & 2) == 0 ? Ab[b + 2] : C[b - 2];
      }
      int nonzeros;
      {
        DynProb* cell = store_.get (big ? LH264_TB_NZ8 : LH264_TB_NZ4,
                                    (uint32_t) ((((((st * 16 + mbc) * 3 + color) * 3 + min2 (past)) * 3 + min2 (left)) * 3) + min2 (above)));
        IntPrior p; p.exponent = cell; p.E = 3; p.mantissa = cell + 3; p.M = 4; p.zero = cell + 7; p.sign = nullptr; p.order = 0;
        const int t = color ? TAG_CRAC : TAG_LAC_0;
        nonzeros = scan_int (p, t, t, t, t);
      }
      if (nonzeros < 0 || nonzeros > nco - start) { fail ("corrupt nonzero count"); return; }
      const uint32_t outer0 = (uint32_t) (((st * 16 + mbc) * 3 + color) * nco);
      int left_nz = nonzeros, prev = 0, prev2 = 0, emitted = 0;
      for (int pos = start; pos < nco && left_nz > 0 && !failed_; pos++) {
        const uint32_t inner = (uint32_t) ((((std::min (4, left_nz) * 5 + clamp04 (prev + 2)) * 5 + clamp04 (prev2 + 2)) * 5 + 2) * 5 + 2);
        const bool first = color == 0 && emitted == 0 && mbc != 1;
        const int base = color ? TAG_CRAC : (first ? TAG_LAC_0 : TAG_LAC_N);
        const int v = scan_uegk (store_.get (big ? LH264_TB_AC8 : LH264_TB_AC4, (outer0 + (uint32_t)emitted) * 3125u + inner), 14, 4, 2, 4, 0,
                                 base + 2, base + 3, base + 1, base + 4);
        if (v < -32768 || v > 32767) { fail ("corrupt coefficient"); return; }
        const int at = big ? kZz64[pos] : kZz16[pos];
        lev[b * 16 + at] = (int16_t)v;
        prev2 = prev; prev = v; emitted++;
        if (v) { left_nz--; C[b + (at >> 4)]++; }
      }
    }
  }
}

// ---- CAVLC writer ---------------------------------------------------------------------------------------------------------
void Restorer::build_vlc() {
  memset (tok_len_, 0, sizeof (tok_len_)); memset (tok_code_, 0, sizeof (tok_code_));
  for (int t = 0; t < 5; t++) for (int i = 0; i < kCoeffTokenCount[t]; i++) {
      const VlcTok& v = kCoeffToken[t][i];
      if (t == 3) continue;
      tok_len_[t][v.total_coeff][v.trailing_ones] = v.len; tok_code_[t][v.total_coeff][v.trailing_ones] = v.code;
    }
}

// Intra4x4PredMode / Intra8x8PredMode prediction, 8.3.1.1 / 8.3.2.1, on 4x4 granularity (as the front end's parser derives it)
int Restorer::pred_intra_mode (int k, int bx, int by, int w, int sid, bool cip) const {
  int modeA = 2, modeB = 2; bool dcpred = false;
  auto avail = [&] (int kk) { return kk >= 0 && ws_[kk].slice == sid && (!cip || ws_[kk].type_class == 1 || ws_[kk].type_class == 2); };
  {
    int kk = k, x = bx - 1, y = by;
    if (x < 0) { kk = (k % w) ? k - 1 : -1; x = 3; }
    if (kk != k && !avail (kk)) dcpred = true;
    else modeA = (kk == k || ws_[kk].type_class == 1) ? ws_[kk].ipm[y * 4 + x] : 2;
  }
  {
    int kk = k, x = bx, y = by - 1;
    if (y < 0) { kk = k >= w ? k - w : -1; y = 3; }
    if (kk != k && !avail (kk)) dcpred = true;
    else modeB = (kk == k || ws_[kk].type_class == 1) ? ws_[kk].ipm[y * 4 + x] : 2;
  }
  return dcpred ? 2 : std::min (modeA, modeB);
}

// residual_block_cavlc, 7.3.5.3.2 / 9.2: lv = the block's levels in scan order
void Restorer::write_residual_block (const int* lv, int maxc, int nC, int& total_out) {
  int coef[16], pos_of[16], total = 0;
  for (int i = maxc - 1; i >= 0; i--) if (lv[i]) { coef[total] = lv[i]; pos_of[total] = i; total++; }   // highest frequency first
  total_out = total;
  int t1 = 0;
  while (t1 < total && t1 < 3 && (coef[t1] == 1 || coef[t1] == -1)) t1++;
  const int tab = nC < 0 ? 4 : nC < 2 ? 0 : nC < 4 ? 1 : nC < 8 ? 2 : 3;
  if (tab == 3) w_.emit_bits (total == 0 ? 3u : (uint32_t) (((total - 1) << 2) | t1), 6);
  else w_.emit_bits (tok_code_[tab][total][t1], tok_len_[tab][total][t1]);
  if (total == 0) return;
  for (int i = 0; i < t1; i++) w_.emit_bit (coef[i] < 0);
  int suffix_len = (total > 10 && t1 < 3) ? 1 : 0;
  for (int i = t1; i < total; i++) {
    const int level = coef[i];
    int code = level > 0 ? 2 * level - 2 : -2 * level - 1;
    if (i == t1 && t1 < 3) code -= 2;
    const int base15 = (15 << suffix_len) + (suffix_len == 0 ? 15 : 0);
    if (suffix_len == 0 && code < 14) { w_.emit_bits (1, code + 1); }
    else if (suffix_len == 0 && code < 30) { w_.emit_bits (1, 15); w_.emit_bits ((uint32_t) (code - 14), 4); }
    else if (suffix_len > 0 && (code >> suffix_len) < 15) { w_.emit_bits (1, (code >> suffix_len) + 1); w_.emit_bits ((uint32_t)code & ((1u << suffix_len) - 1), suffix_len); }
    else {
      const int v = code - base15;
      if (v < 4096) { w_.emit_bits (1, 16); w_.emit_bits ((uint32_t)v, 12); }
      else {                                                  // level_prefix >= 16 (escape for very large levels)
        int p = 16;
        while (v - ((1 << (p - 3)) - 4096) >= (1 << (p - 3))) p++;
        w_.emit_bits (0, p - 16); w_.emit_bits (1, 17);       // p zeros and a one
        w_.emit_bits ((uint32_t) (v - ((1 << (p - 3)) - 4096)), p - 3);
      }
    }
    if (suffix_len == 0) suffix_len = 1;
    if (std::abs (level) > (3 << (suffix_len - 1)) && suffix_len < 6) suffix_len++;
  }
  int zeros_left = 0;
  if (total < maxc) {
    zeros_left = pos_of[0] + 1 - total;                       // zeros below the highest-frequency coefficient
    const VlcSym* tbl = nC < 0 ? kTotalZerosChromaDc[total] : kTotalZeros[total];
    const int cnt = nC < 0 ? kTotalZerosChromaDcCount[total] : kTotalZerosCount[total];
    for (int i = 0; i < cnt; i++) if (tbl[i].sym == zeros_left) { w_.emit_bits (tbl[i].code, tbl[i].len); break; }
  }
  for (int i = 0; i < total - 1 && zeros_left > 0; i++) {
    const int run = pos_of[i] - pos_of[i + 1] - 1;
    const int zl = std::min (zeros_left, 7);
    for (int q = 0; q < kRunBeforeCount[zl]; q++) if (kRunBefore[zl][q].sym == run) { w_.emit_bits (kRunBefore[zl][q].code, kRunBefore[zl][q].len); break; }
    zeros_left -= run;
  }
}

// macroblock_layer, 7.3.5
void Restorer::write_mb (const Parser::HeaderInfo& H, int k, const MbDec& m, int& qp_prev) {
  const int w = H.mb_w, sid = sid_;
  const bool is_p = H.sh.slice_type == 0;
  WState& s = ws_[k];
  s.slice = sid; memset (s.nzc, 0, 24); for (int i = 0; i < 16; i++) s.ipm[i] = 2;
  const uint32_t type = m.type;
  s.mb_type = type;
  const bool intra = (type & LH264_MB_INTRA) != 0;
  const bool i16 = type == LH264_MB_I16x16;
  const int cbp = m.cbp_l | (m.cbp_c << 4);
  if (type == LH264_MB_IPCM) {                          // 7.3.5: mb_type 25, alignment zeros, 256 + 2 x 64 samples; QP prediction is not touched
    put_ue (25u + (is_p ? 5u : 0u));
    while (w_.bits_in_byte() & 7) w_.emit_bit (0);
    for (int i = 0; i < 384; i++) w_.emit_bits (pcm_[i], 8);
    pcm_ += 384;
    s.type_class = 2;
    memset (s.nzc, 16, 24);
    return;
  }
  if (intra) {
    uint32_t mbt;
    if (i16) {
      static const int kRaw16[7] = {0, 1, 2, 3, 2, 2, 2};
      mbt = 1u + (uint32_t)kRaw16[std::min (m.luma16_mode, 6)] + 4u * (uint32_t)m.cbp_c + (m.cbp_l ? 12u : 0u);
      s.type_class = 2;
    } else { mbt = 0; s.type_class = 1; }
    put_ue (mbt + (is_p ? 5u : 0u));
    if (!i16) {
      const bool t8 = type == LH264_MB_I8x8;
      if (H.transform_8x8) w_.emit_bit (t8);
      const int nblk = t8 ? 4 : 16;
      for (int i = 0; i < nblk; i++) {
        const int bx = t8 ? (i & 1) * 2 : z2x (i), by = t8 ? (i >> 1) * 2 : z2y (i);
        const int pred = pred_intra_mode (k, bx, by, w, sid, H.constrained_intra_pred);
        const int mode = m.pred_mode[i];
        if (mode == pred) w_.emit_bit (1);
        else { w_.emit_bit (0); w_.emit_bits ((uint32_t) (mode < pred ? mode : mode - 1), 3); }
        const int n = t8 ? 2 : 1;
        for (int yy = 0; yy < n; yy++) for (int x = 0; x < n; x++) s.ipm[(by + yy) * 4 + bx + x] = (int8_t)mode;
      }
    }
    static const int kRawChroma[7] = {0, 1, 2, 3, 0, 0, 0};
    put_ue ((uint32_t)kRawChroma[std::min (m.chroma_mode, 6)]);
    if (!i16) { for (uint32_t ci = 0; ci < 48; ci++) if (kCbpIntra[ci] == cbp) { put_ue (ci); break; } }
  } else {
    s.type_class = 3;
    const int nref = H.sh.num_ref_idx_l0;
    auto put_ref = [&] (int r) { if (nref <= 1) return; if (nref == 2) w_.emit_bit (r ? 0 : 1); else put_ue ((uint32_t)r); };
    auto put_mvd = [&] (int blk) { put_se (m.mvd[blk][0]); put_se (m.mvd[blk][1]); };
    if (type == LH264_MB_P16x16) { put_ue (0); put_ref (m.ref_idx[0]); put_mvd (0); }
    else if (type == LH264_MB_P16x8) { put_ue (1); put_ref (m.ref_idx[0]); put_ref (m.ref_idx[1]); put_mvd (0); put_mvd (8); }
    else if (type == LH264_MB_P8x16) { put_ue (2); put_ref (m.ref_idx[0]); put_ref (m.ref_idx[1]); put_mvd (0); put_mvd (2); }
    else {
      put_ue (type == LH264_MB_P8x8 ? 3 : 4);
      for (int q = 0; q < 4; q++) put_ue (m.sub_type[q] == LH264_SUB_8x8 ? 0u : m.sub_type[q] == LH264_SUB_8x4 ? 1u : m.sub_type[q] == LH264_SUB_4x8 ? 2u : 3u);
      if (type == LH264_MB_P8x8) for (int q = 0; q < 4; q++) put_ref (m.ref_idx[q]);
      for (int q = 0; q < 4; q++) {
        switch (m.sub_type[q]) {
        case LH264_SUB_8x8: put_mvd (kZ2Raster[q << 2]); break;
        case LH264_SUB_8x4: for (int j = 0; j < 2; j++) put_mvd (kZ2Raster[(q << 2) + (j << 1)]); break;
        case LH264_SUB_4x8: for (int j = 0; j < 2; j++) put_mvd (kZ2Raster[(q << 2) + j]); break;
        default: for (int j = 0; j < 4; j++) put_mvd (kZ2Raster[(q << 2) + j]); break;
        }
      }
    }
    for (uint32_t ci = 0; ci < 48; ci++) if (kCbpInter[ci] == cbp) { put_ue (ci); break; }
    bool no_sub_lt8 = true;
    if (type == LH264_MB_P8x8 || type == LH264_MB_P8x8REF0) for (int q = 0; q < 4; q++) if (m.sub_type[q] != LH264_SUB_8x8) no_sub_lt8 = false;
    if (m.cbp_l && H.transform_8x8 && no_sub_lt8) w_.emit_bit (m.t8);
  }
  if (!(cbp || i16)) return;
  {
    const int d = (((m.luma_qp - qp_prev) + 26 + 104) % 52) - 26;
    put_se (d);
    qp_prev = m.luma_qp;
  }
  // residual, 7.3.5.3
  int lv[16], tot;
  // nC of 9.2.1 from the total_coeff of the blocks to the left and above (macroblocks of other slices are not available)
  auto luma_nC = [&] (int bx, int by) {
    int nA = 0, nB = 0; bool aA = true, aB = true;
    if (bx == 0) { const int kk = (k % w) ? k - 1 : -1; aA = kk >= 0 && ws_[kk].slice == sid; if (aA) nA = ws_[kk].nzc[by * 4 + 3]; } else nA = s.nzc[by * 4 + bx - 1];
    if (by == 0) { const int kk = k >= w ? k - w : -1; aB = kk >= 0 && ws_[kk].slice == sid; if (aB) nB = ws_[kk].nzc[12 + bx]; } else nB = s.nzc[(by - 1) * 4 + bx];
    return (aA && aB) ? (nA + nB + 1) >> 1 : aA ? nA : aB ? nB : 0;
  };
  if (i16) {
    for (int i = 0; i < 16; i++) { const int r = kZigzag4x4[i]; lv[i] = m.lev[(((r & 3) & 1) | (((r >> 2) & 1) << 1) | (((r & 3) >> 1) << 2) | (((r >> 2) >> 1) << 3)) * 16]; }
    write_residual_block (lv, 16, luma_nC (0, 0), tot);
  }
  for (int i8 = 0; i8 < 4; i8++) {
    if (!((m.cbp_l >> i8) & 1)) continue;
    for (int j = 0; j < 4; j++) {
      const int z = i8 * 4 + j, bx = z2x (z), by = z2y (z);
      const int maxc = i16 ? 15 : 16;
      for (int i = 0; i < maxc; i++) lv[i] = m.t8 ? m.lev[i8 * 64 + kZigzag8x8[4 * i + j]] : m.lev[z * 16 + kZigzag4x4[i16 ? i + 1 : i]];
      write_residual_block (lv, maxc, luma_nC (bx, by), tot);
      s.nzc[by * 4 + bx] = (uint8_t)tot;
    }
  }
  if (m.cbp_c) {
    for (int p = 0; p < 2; p++) {
      for (int i = 0; i < 4; i++) lv[i] = m.lev[256 + p * 64 + i * 16];
      write_residual_block (lv, 4, -1, tot);
    }
    if (m.cbp_c == 2) {
      for (int p = 0; p < 2; p++) for (int j = 0; j < 4; j++) {
          const int bx = j & 1, by = j >> 1;
          int nA = 0, nB = 0; bool aA = true, aB = true;
          if (bx == 0) { const int kk = (k % w) ? k - 1 : -1; aA = kk >= 0 && ws_[kk].slice == sid; if (aA) nA = ws_[kk].nzc[kChromaNzc[p][by * 2 + 1]]; } else nA = s.nzc[kChromaNzc[p][by * 2]];
          if (by == 0) { const int kk = k >= w ? k - w : -1; aB = kk >= 0 && ws_[kk].slice == sid; if (aB) nB = ws_[kk].nzc[kChromaNzc[p][2 + bx]]; } else nB = s.nzc[kChromaNzc[p][bx]];
          const int nC = (aA && aB) ? (nA + nB + 1) >> 1 : aA ? nA : aB ? nB : 0;
          for (int i = 0; i < 15; i++) lv[i] = m.lev[256 + p * 64 + j * 16 + kZigzag4x4[i + 1]];
          write_residual_block (lv, 15, nC, tot);
          s.nzc[kChromaNzc[p][j]] = (uint8_t)tot;
        }
    }
  }
}

// ---- CABAC macroblock layer writer: 7.3.5 with the binarisations and context selection of 9.3.2 / 9.3.3, the mirror image of
// the front end's parse_mb_cabac (csrc/host/h264_parser.cpp) -----------------------------------------------------------------
void Restorer::cabac_residual (int k, int cat, int blk, int plane, bool cur_intra, const int* lv, int maxc) {
  const int w = img_w_, sid = sid_;
  WState& s = ws_[k];
  int last_nz = -1, n_sig = 0;
  for (int i = 0; i < maxc; i++) if (lv[i]) { last_nz = i; n_sig++; }
  if (cat != 5) {                                          // coded_block_flag, 9.3.3.1.1.9
    int bit, bitA, bitB; int kA = k, kB = k;
    if (cat == 0) { bit = bitA = bitB = 16; kA = -2; kB = -2; }
    else if (cat == 3) { bit = bitA = bitB = 17 + plane; kA = -2; kB = -2; }
    else if (cat == 4) {
      const int cx = blk & 1, cy = blk >> 1;
      bit = 19 + plane * 4 + blk;
      if (cx == 0) { kA = -2; bitA = 19 + plane * 4 + cy * 2 + 1; } else bitA = bit - 1;
      if (cy == 0) { kB = -2; bitB = 19 + plane * 4 + 2 + cx; } else bitB = bit - 2;
    } else {
      const int bx = blk & 3, by = blk >> 2;
      bit = blk;
      if (bx == 0) { kA = -2; bitA = by * 4 + 3; } else bitA = blk - 1;
      if (by == 0) { kB = -2; bitB = 12 + bx; } else bitB = blk - 4;
    }
    if (kA == -2) kA = ((k % w) && ws_[k - 1].slice == sid) ? k - 1 : -1;
    if (kB == -2) kB = (k >= w && ws_[k - w].slice == sid) ? k - w : -1;
    const int cA = kA < 0 ? (cur_intra ? 1 : 0) : (int) ((ws_[kA].cbf >> bitA) & 1);
    const int cBf = kB < 0 ? (cur_intra ? 1 : 0) : (int) ((ws_[kB].cbf >> bitB) & 1);
    ce_.encode (85 + kCatCbf[cat] + cA + 2 * cBf, n_sig != 0);
    if (!n_sig) return;
    s.cbf |= 1u << bit;
  }
  const int sig_base = cat == 5 ? 402 : 105 + kCatMap[cat], last_base = cat == 5 ? 417 : 166 + kCatMap[cat];
  const int abs_base = cat == 5 ? 426 : 227 + kCatAbs[cat];
  for (int i = 0; i < maxc - 1; i++) {
    const int inc_s = cat == 5 ? kSig8x8[i] : cat == 3 ? std::min (i, 2) : i;
    const int inc_l = cat == 5 ? kLast8x8[i] : cat == 3 ? std::min (i, 2) : i;
    const int sig = lv[i] != 0;
    ce_.encode (sig_base + inc_s, sig);
    if (sig) {
      ce_.encode (last_base + inc_l, i == last_nz);
      if (i == last_nz) break;
    }
  }
  int num_eq1 = 0, num_gt1 = 0;
  for (int i = maxc - 1; i >= 0; i--) {
    if (!lv[i]) continue;
    const int mag = std::abs (lv[i]), v = mag - 1;
    int inc = num_gt1 ? 0 : std::min (4, 1 + num_eq1);
    ce_.encode (abs_base + inc, v > 0);
    if (v > 0) {
      inc = 5 + std::min (4 - (cat == 3 ? 1 : 0), num_gt1);
      int cnt = 1;
      while (cnt < 14) { const int bin = v > cnt; ce_.encode (abs_base + inc, bin); if (!bin) break; cnt++; }
      if (v >= 14) {                                         // Exp-Golomb order 0 suffix, bypass coded
        int rem = v - 14, kk = 0;
        while (rem >= (1 << kk)) { ce_.bypass (1); rem -= 1 << kk; kk++; }
        ce_.bypass (0);
        while (kk--) ce_.bypass ((rem >> kk) & 1);
      }
    }
    ce_.bypass (lv[i] < 0);
    if (mag == 1) num_eq1++; else num_gt1++;
  }
}

void Restorer::write_mb_cabac (const Parser::HeaderInfo& H, int k, const MbDec* mp, int& qp_prev, int& last_dqp) {
  const int w = H.mb_w, sid = sid_;
  const bool is_p = H.sh.slice_type == 0;
  WState& s = ws_[k];
  const int kA = ((k % w) && ws_[k - 1].slice == sid) ? k - 1 : -1, kB = (k >= w && ws_[k - w].slice == sid) ? k - w : -1;
  if (is_p) ce_.encode (11 + (kA >= 0 && !ws_[kA].skip) + (kB >= 0 && !ws_[kB].skip), mp == nullptr);     // mb_skip_flag
  s.slice = sid; memset (s.nzc, 0, 24); for (int i = 0; i < 16; i++) { s.ipm[i] = 2; s.mvd[i][0] = s.mvd[i][1] = 0; }
  s.skip = 0; s.t8 = 0; s.cbp = 0; s.chroma_pred = 0; s.cbf = 0;
  for (int i = 0; i < 4; i++) s.ref[i] = -1;
  if (!mp) {
    s.mb_type = LH264_MB_SKIP; s.type_class = 3; s.skip = 1;
    for (int i = 0; i < 4; i++) s.ref[i] = 0;
    last_dqp = 0;
    return;
  }
  const MbDec& m = *mp;
  const uint32_t type = m.type;
  s.mb_type = type;
  const bool intra = (type & LH264_MB_INTRA) != 0;
  const bool i16 = type == LH264_MB_I16x16;
  const int cbp = m.cbp_l | (m.cbp_c << 4);
  auto i_type = [&] (bool islice, int mbt) {               // 9.3.2.5, Table 9-36
    const int ctx0 = islice ? 3 + (kA >= 0 && ws_[kA].type_class != 1) + (kB >= 0 && ws_[kB].type_class != 1) : 17;
    if (mbt == 0) { ce_.encode (ctx0, 0); return; }
    ce_.encode (ctx0, 1);
    ce_.terminate (0);
    const int v = mbt - 1, pm = v & 3, chroma = (v >> 2) % 3, luma = v >= 12;
    const int base = islice ? 3 : 17;
    ce_.encode (base + (islice ? 3 : 1), luma);
    ce_.encode (base + (islice ? 4 : 2), chroma != 0);
    if (chroma) ce_.encode (base + (islice ? 5 : 2), chroma == 2);
    ce_.encode (base + (islice ? 6 : 3), pm >> 1);
    ce_.encode (base + (islice ? 7 : 3), pm & 1);
  };
  auto t8_flag = [&] (int v) { ce_.encode (399 + (kA >= 0 && ws_[kA].t8) + (kB >= 0 && ws_[kB].t8), v); };
  bool t8 = false;
  if (type == LH264_MB_IPCM) {
    // mb_type 25: the prefix bin, then the terminating bin set, which flushes the engine (9.3.4.5) and leaves the stream byte aligned;
    // the samples follow as they are and the engine starts afresh behind them, the context states kept (9.3.1.2)
    if (is_p) ce_.encode (14, 1);
    ce_.encode (is_p ? 17 : 3 + (kA >= 0 && ws_[kA].type_class != 1) + (kB >= 0 && ws_[kB].type_class != 1), 1);
    ce_.terminate (1);
    ce_.bytes.insert (ce_.bytes.end(), pcm_, pcm_ + 384);
    pcm_ += 384;
    ce_.low = 0; ce_.range = 510; ce_.outstanding = 0; ce_.first = true;
    s.type_class = 2; s.cbf = 0xffffffffu; s.cbp = 0x2f;
    memset (s.nzc, 16, 24);
    last_dqp = 0;
    return;
  }
  if (intra) {
    int mbt = 0;
    if (i16) {
      static const int kRaw16[7] = {0, 1, 2, 3, 2, 2, 2};
      mbt = 1 + kRaw16[std::min (m.luma16_mode, 6)] + 4 * m.cbp_c + (m.cbp_l ? 12 : 0);
      s.type_class = 2;
    } else s.type_class = 1;
    if (is_p) ce_.encode (14, 1);
    i_type (!is_p, mbt);
    if (!i16) {
      t8 = type == LH264_MB_I8x8;
      if (H.transform_8x8) t8_flag (t8);
      s.t8 = t8;
      const int nblk = t8 ? 4 : 16;
      for (int i = 0; i < nblk; i++) {
        const int bx = t8 ? (i & 1) * 2 : z2x (i), by = t8 ? (i >> 1) * 2 : z2y (i);
        const int pred = pred_intra_mode (k, bx, by, w, sid, H.constrained_intra_pred);
        const int mode = m.pred_mode[i];
        ce_.encode (68, mode == pred);
        if (mode != pred) { const int rem = mode < pred ? mode : mode - 1; ce_.encode (69, rem & 1); ce_.encode (69, (rem >> 1) & 1); ce_.encode (69, (rem >> 2) & 1); }
        const int n = t8 ? 2 : 1;
        for (int yy = 0; yy < n; yy++) for (int x = 0; x < n; x++) s.ipm[(by + yy) * 4 + bx + x] = (int8_t)mode;
      }
    }
    static const int kRawChroma[7] = {0, 1, 2, 3, 0, 0, 0};
    const int cm = kRawChroma[std::min (m.chroma_mode, 6)];
    {                                                        // intra_chroma_pred_mode, 9.3.3.1.1.8
      const int cA = kA >= 0 && ws_[kA].type_class != 3 && ws_[kA].chroma_pred != 0;
      const int cBn = kB >= 0 && ws_[kB].type_class != 3 && ws_[kB].chroma_pred != 0;
      ce_.encode (64 + cA + cBn, cm != 0);
      if (cm) { ce_.encode (67, cm != 1); if (cm != 1) ce_.encode (67, cm == 3); }
    }
    s.chroma_pred = (uint8_t)cm;
  } else {
    s.type_class = 3;
    const int nref = H.sh.num_ref_idx_l0;
    ce_.encode (14, 0);
    if (type == LH264_MB_P16x16) { ce_.encode (15, 0); ce_.encode (16, 0); }
    else if (type == LH264_MB_P8x8 || type == LH264_MB_P8x8REF0) { ce_.encode (15, 0); ce_.encode (16, 1); }
    else if (type == LH264_MB_P16x8) { ce_.encode (15, 1); ce_.encode (17, 1); }
    else { ce_.encode (15, 1); ce_.encode (17, 0); }
    auto ref_gt0 = [&] (int bx, int by) -> int {
      int kk = k, x = bx, yy = by;
      if (x < 0) { kk = kA; x = 3; } else if (yy < 0) { kk = kB; yy = 3; }
      if (kk < 0) return 0;
      const WState& t = ws_[kk];
      if (t.type_class != 3 || t.skip) return 0;
      return t.ref[(yy >> 1) * 2 + (x >> 1)] > 0;
    };
    auto put_ref = [&] (int bx, int by, int v) {
      if (nref <= 1) return;
      int inc = ref_gt0 (bx - 1, by) + 2 * ref_gt0 (bx, by - 1);
      for (int i = 0; i < v; i++) { ce_.encode (54 + inc, 1); inc = i == 0 ? 4 : 5; }
      ce_.encode (54 + inc, 0);
    };
    auto abs_mvd = [&] (int bx, int by, int comp) -> int {
      int kk = k, x = bx, yy = by;
      if (x < 0) { kk = kA; x = 3; } else if (yy < 0) { kk = kB; yy = 3; }
      if (kk < 0) return 0;
      return ws_[kk].mvd[yy * 4 + x][comp];
    };
    auto put_mvd1 = [&] (int bx, int by, int comp, int d) {     // UEG3, uCoff 9, signed (9.3.2.3, 9.3.3.1.1.7)
      const int base = comp ? 47 : 40;
      const int sum = abs_mvd (bx - 1, by, comp) + abs_mvd (bx, by - 1, comp);
      int inc = sum < 3 ? 0 : sum > 32 ? 2 : 1;
      const int a = std::abs (d);
      ce_.encode (base + inc, a != 0);
      if (!a) return;
      int v = 1;
      inc = 3;
      while (v < 9) { const int bin = a > v; ce_.encode (base + inc, bin); if (!bin) break; v++; if (inc < 6) inc++; }
      if (a >= 9) {
        int rem = a - 9, kk = 3;
        while (rem >= (1 << kk)) { ce_.bypass (1); rem -= 1 << kk; kk++; }
        ce_.bypass (0);
        while (kk--) ce_.bypass ((rem >> kk) & 1);
      }
      ce_.bypass (d < 0);
    };
    auto part = [&] (int bx, int by, int bw, int bh) {          // the partition whose motion vector difference sits at (bx,by)
      const int dx = m.mvd[by * 4 + bx][0], dy = m.mvd[by * 4 + bx][1];
      put_mvd1 (bx, by, 0, dx); put_mvd1 (bx, by, 1, dy);
      const uint8_t ax = (uint8_t)std::min (255, std::abs (dx)), ay = (uint8_t)std::min (255, std::abs (dy));
      for (int yy = by; yy < by + bh; yy++) for (int x = bx; x < bx + bw; x++) { s.mvd[yy * 4 + x][0] = ax; s.mvd[yy * 4 + x][1] = ay; }
    };
    if (type == LH264_MB_P16x16 || type == LH264_MB_P16x8 || type == LH264_MB_P8x16) {
      const int mbt = type == LH264_MB_P16x16 ? 0 : type == LH264_MB_P16x8 ? 1 : 2;
      const int np = mbt == 0 ? 1 : 2;
      for (int i = 0; i < np; i++) {
        const int bx = mbt == 2 ? i * 2 : 0, by = mbt == 1 ? i * 2 : 0;
        put_ref (bx, by, m.ref_idx[i]);
        for (int q = 0; q < 4; q++) {
          const bool in = mbt == 0 || (mbt == 1 ? (q >> 1) == i : (q & 1) == i);
          if (in) s.ref[q] = (int8_t)m.ref_idx[i];
        }
      }
      for (int i = 0; i < np; i++) {
        int bx = 0, by = 0, bw = 4, bh = 4;
        if (mbt == 1) { bh = 2; by = i * 2; } else if (mbt == 2) { bw = 2; bx = i * 2; }
        part (bx, by, bw, bh);
      }
    } else {
      int sub[4];
      for (int q = 0; q < 4; q++) {                           // sub_mb_type, Table 9-37
        sub[q] = m.sub_type[q] == LH264_SUB_8x8 ? 0 : m.sub_type[q] == LH264_SUB_8x4 ? 1 : m.sub_type[q] == LH264_SUB_4x8 ? 2 : 3;
        if (sub[q] == 0) ce_.encode (21, 1);
        else { ce_.encode (21, 0); if (sub[q] == 1) ce_.encode (22, 0); else { ce_.encode (22, 1); ce_.encode (23, sub[q] == 2); } }
      }
      for (int q = 0; q < 4; q++) { put_ref ((q & 1) * 2, (q >> 1) * 2, m.ref_idx[q]); s.ref[q] = (int8_t)m.ref_idx[q]; }
      for (int q = 0; q < 4; q++) {
        const int qx = (q & 1) * 2, qy = (q >> 1) * 2;
        const int nsp = sub[q] == 0 ? 1 : sub[q] == 3 ? 4 : 2;
        for (int j = 0; j < nsp; j++) {
          int bx = qx, by = qy, bw = 2, bh = 2;
          if (sub[q] == 1) { bh = 1; by += j; } else if (sub[q] == 2) { bw = 1; bx += j; } else if (sub[q] == 3) { bw = bh = 1; bx += j & 1; by += j >> 1; }
          part (bx, by, bw, bh);
        }
      }
    }
  }
  if (!i16) {                                                // coded_block_pattern, 9.3.2.6 / 9.3.3.1.1.4
    auto luma_bit = [&] (int kk, int b8) -> int {
      if (kk < 0) return 0;
      if (ws_[kk].skip) return 1;
      return ((ws_[kk].cbp >> b8) & 1) ? 0 : 1;
    };
    int cl = 0;
    for (int b8 = 0; b8 < 4; b8++) {
      const int cA = (b8 & 1) ? (((cl >> (b8 - 1)) & 1) ? 0 : 1) : luma_bit (kA, b8 + 1);
      const int cBn = (b8 & 2) ? (((cl >> (b8 - 2)) & 1) ? 0 : 1) : luma_bit (kB, b8 + 2);
      const int bit = (m.cbp_l >> b8) & 1;
      ce_.encode (73 + cA + 2 * cBn, bit);
      cl |= bit << b8;
    }
    auto chroma_nz = [&] (int kk, int lvl) -> int {
      if (kk < 0) return 0;
      if (ws_[kk].skip) return 0;
      return (ws_[kk].cbp >> 4) >= lvl;
    };
    ce_.encode (77 + chroma_nz (kA, 1) + 2 * chroma_nz (kB, 1), m.cbp_c != 0);
    if (m.cbp_c) ce_.encode (77 + 4 + chroma_nz (kA, 2) + 2 * chroma_nz (kB, 2), m.cbp_c == 2);
    if (!intra) {
      bool no_sub_lt8 = true;
      if (type == LH264_MB_P8x8 || type == LH264_MB_P8x8REF0) for (int q = 0; q < 4; q++) if (m.sub_type[q] != LH264_SUB_8x8) no_sub_lt8 = false;
      if (m.cbp_l && H.transform_8x8 && no_sub_lt8) { t8 = m.t8 != 0; t8_flag (t8); }
    }
  }
  s.cbp = (uint8_t)cbp;
  if (t8) s.t8 = 1;
  if (!(cbp || i16)) { last_dqp = 0; return; }
  {                                                          // mb_qp_delta, 9.3.2.7 / 9.3.3.1.1.5
    const int d = (((m.luma_qp - qp_prev) + 26 + 104) % 52) - 26;
    const int v = d > 0 ? 2 * d - 1 : -2 * d;
    ce_.encode (60 + (last_dqp != 0 ? 1 : 0), v != 0);
    if (v) {
      ce_.encode (62, v >= 2);
      if (v >= 2) { for (int j = 2; j < v; j++) ce_.encode (63, 1); ce_.encode (63, 0); }
    }
    last_dqp = d;
    qp_prev = m.luma_qp;
  }
  int lv[64];
  if (i16) {
    for (int i = 0; i < 16; i++) { const int r = kZigzag4x4[i]; lv[i] = m.lev[(((r & 3) & 1) | (((r >> 2) & 1) << 1) | (((r & 3) >> 1) << 2) | (((r >> 2) >> 1) << 3)) * 16]; }
    cabac_residual (k, 0, 0, 0, true, lv, 16);
  }
  for (int i8 = 0; i8 < 4; i8++) {
    if (!((m.cbp_l >> i8) & 1)) continue;
    if (t8) {
      for (int i = 0; i < 64; i++) lv[i] = m.lev[i8 * 64 + kZigzag8x8[i]];
      cabac_residual (k, 5, i8, 0, intra, lv, 64);
      for (int j = 0; j < 4; j++) { const int z = i8 * 4 + j; s.cbf |= 1u << (z2y (z) * 4 + z2x (z)); }
      continue;
    }
    for (int j = 0; j < 4; j++) {
      const int z = i8 * 4 + j, bx = z2x (z), by = z2y (z);
      const int maxc = i16 ? 15 : 16;
      for (int i = 0; i < maxc; i++) lv[i] = m.lev[z * 16 + kZigzag4x4[i16 ? i + 1 : i]];
      cabac_residual (k, i16 ? 1 : 2, by * 4 + bx, 0, intra, lv, maxc);
    }
  }
  if (m.cbp_c) {
    for (int p = 0; p < 2; p++) {
      for (int i = 0; i < 4; i++) lv[i] = m.lev[256 + p * 64 + i * 16];
      cabac_residual (k, 3, 0, p, intra, lv, 4);
    }
    if (m.cbp_c == 2) {
      for (int p = 0; p < 2; p++) for (int j = 0; j < 4; j++) {
          for (int i = 0; i < 15; i++) lv[i] = m.lev[256 + p * 64 + j * 16 + kZigzag4x4[i + 1]];
          cabac_residual (k, 4, j, p, intra, lv, 15);
        }
    }
  }
}

// one slice: the macroblocks in scan order (WelsDecodeSliceForRecoding DS:2476-2830), each written out as CAVLC at once
bool Restorer::decode_slice (const Parser::HeaderInfo& H) {
  const int w = H.mb_w, n = H.mb_w * H.mb_h;
  if (n <= 0 || H.sh.first_mb < 0 || H.sh.first_mb >= n) { fail ("bad slice geometry"); return false; }
  if ((int)ipm_.size() != n * 8) { ipm_.assign ((size_t)n * 8, 0); nxn_.assign (n, 0); }
  if (ws_n_ != n) { ws_.assign (n, WState()); ws_n_ = n; }
  sid_++;
  bool prior_valid = true;
  update_frame (H.sh.frame_num);
  if (img_w_ != H.mb_w || img_h_ != H.mb_h) {
    prior_valid = false;
    img_w_ = H.mb_w; img_h_ = H.mb_h;
    img_[0].assign (n, Cell()); img_[1].assign (n, Cell());
  }
  std::vector<Cell>& cur = img_[cur_];
  std::vector<Cell>& last = img_[1 - cur_];
  const bool is_p = H.sh.slice_type == 0, cabac = H.cabac;
  const int st = H.sh.slice_type;
  int skip_state = -1, mb_in_slice = 0, cached_qp = 0, last_nonzero_dqp = 0, qp_prev = H.sh.slice_qp, last_dqp = 0;
  if (cabac) {                                       // cabac_alignment_one_bit, then the arithmetic codeword (7.3.4, 9.3.1)
    while (w_.bits_in_byte()) w_.emit_bit (1);
    ce_.init (is_p ? 1 + H.sh.cabac_init_idc : 0, H.sh.slice_qp);
  }
  uint32_t pending_skips = 0;
  MbDec m;
  for (int k = H.sh.first_mb; ; k++, mb_in_slice++) {
    if (k >= n) { fail ("slice runs past the picture"); return false; }
    if (failed_) return false;
    const int x = k % w;
    const Cell* nl = (x > 0 && cur[k - 1].initialized) ? &cur[k - 1] : nullptr;
    const Cell* na = (k >= w && cur[k - w].initialized) ? &cur[k - w] : nullptr;
    const Cell* np = (prior_valid && last[k].initialized) ? &last[k] : nullptr;
    int mb_skip_run = 0;
    const uint32_t stop_idx = (uint32_t) (mb_in_slice < 2048 ? mb_in_slice : 2047);
    if (skip_state == -1 || cabac) {                 // CABAC: a run of 0 or 1 for every macroblock (DS:2505-2516)
      const int pr = np ? np->cached_skips / 8 + (np->cached_skips % 8 ? 1 : 0) : 0;
      const int run = (int)tree (TAG_SKIP, LH264_TB_SKIPRUN, (uint32_t) (pr * 16 + 11));
      if (is_p && !cabac) skip_state = run; else mb_skip_run = run;
      if (cabac && run > 1) { fail ("corrupt skip flag"); return false; }
    }
    if (is_p && !cabac) { mb_skip_run = skip_state; skip_state--; }
    bool has_stop = false;
    if (mb_skip_run == 1) has_stop = scan_bit (TAG_SKIP_END, store_.get (LH264_TB_STOP, stop_idx)) != 0;
    if (mb_skip_run != 0) {
      if (!is_p) { fail ("skip run in an I slice"); return false; }
      cur[k] = last[k];
      nxn_[k] = 0;
      WState& s = ws_[k];
      s.slice = sid_; s.mb_type = LH264_MB_SKIP; s.type_class = 3; memset (s.nzc, 0, 24); for (int i = 0; i < 16; i++) s.ipm[i] = 2;
      if (cabac) { write_mb_cabac (H, k, nullptr, qp_prev, last_dqp); ce_.terminate (has_stop); }
      else pending_skips++;
      if (has_stop) break;
      continue;
    }
    has_stop = scan_bit (TAG_SKIP_END, store_.get (LH264_TB_STOP, stop_idx)) != 0;
    memset (&m, 0, sizeof (m));
    {
      int prior = 15, prev = 15;
      if (na) prior = type_code (na->mb_type);
      if (nl) prior = type_code (nl->mb_type);
      if (np) prev = type_code (np->mb_type);
      const unsigned code = tree (TAG_MB_TYPE, LH264_TB_MBTYPE, (uint32_t) ((prior + prev) * 2 + (is_p ? 1 : 0)));
      if (code > 8) { fail ("corrupt macroblock type"); return false; }
      if (code == 8 && pcm_end_ - pcm_ < 384) { fail ("I_PCM macroblock without its samples (stream LH264_TAG_PCM)"); return false; }
      m.type = kCodeType[code];
    }
    const uint32_t type = m.type;
    const int mbc = type_code (type);
    if (!is_p && !(type & LH264_MB_INTRA)) { fail ("inter macroblock in an I slice"); return false; }
    m.cbp_c = (int)tree (TAG_CBPL, LH264_TB_CBPC, (uint32_t) ((np ? np->cbp_c : 0) * 16 + mbc));
    m.cbp_l = (int)tree (TAG_CBPL, LH264_TB_CBPL, (uint32_t) ((np ? np->cbp_l : 0) * 16 + mbc));
    if (m.cbp_c > 2) { fail ("corrupt chroma cbp"); return false; }
    {
      const int sidx = last_nonzero_dqp < 0 ? 0 : (last_nonzero_dqp == 0 ? 1 : 2);
      const unsigned sw = scan_pow2 (TAG_QPL, 7, store_.get (LH264_TB_QPL, (uint32_t) ((mb_in_slice == 0 ? 1 : 0) * 3 + sidx)), 0);
      const int dqp = (sw & 1) ? - (int) (sw >> 1) - 1 : (int) (sw >> 1);       // unswizzle_sign MM:726-732
      m.luma_qp = (cached_qp + dqp) & 0xff;
      cached_qp = m.luma_qp;
      if (dqp) last_nonzero_dqp = dqp;
      if (m.luma_qp > 51) { fail ("corrupt QP"); return false; }
    }
    m.num_ref = (int)tree (TAG_REF, LH264_TB_NUMREF, (uint32_t) ((np ? np->num_ref : 0) * 16 + mbc));
    int ref_bits = 0;
    while ((1 << ref_bits) < m.num_ref) ref_bits++;
    {
      int pr = 7;
      if (np) { pr = np->chroma_mode; if (pr >= 6) pr = 6; }
      m.chroma_mode = (int)scan_pow2 (TAG_8x8, 3, store_.get (LH264_TB_MODE8, (uint32_t)pr), (unsigned)pr);
      pr = 7;
      if (np) { pr = np->luma16_mode; if (pr >= 6) pr = 6; }
      m.luma16_mode = (int)scan_pow2 (TAG_16x16, 3, store_.get (LH264_TB_MODE8, (uint32_t)pr), (unsigned)pr);
    }
    int8_t* my_ipm = &ipm_[(size_t)k * 8];
    if (type == LH264_MB_I4x4 || type == LH264_MB_I8x8) {
      int8_t cache[48];
      memset (cache, 0, sizeof (cache));
      // the decoder's intra-mode cache and sample availability, normal / constrained_intra_pred variants (see pip_symbols.cpp)
      const bool cip = H.constrained_intra_pred;
      bool left_av = x > 0 && k - 1 >= H.sh.first_mb, top_av = k - w >= H.sh.first_mb, topleft_av = x > 0 && k - w - 1 >= H.sh.first_mb;
      const uint32_t lt = left_av ? ws_[k - 1].mb_type : 0, tt = top_av ? ws_[k - w].mb_type : 0, tlt = topleft_av ? ws_[k - w - 1].mb_type : 0;
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
      if (type == LH264_MB_I4x4) {
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
          m.pred_mode[i] = (int)tree (TAG_PRED_MODE, LH264_TB_PREDMODE, (uint32_t) ((mbc * 8 + avail_idx) * 9 + pred));
          if (m.pred_mode[i] > 8) { fail ("corrupt intra mode"); return false; }
          cache[kScan8[i]] = (int8_t)m.pred_mode[i];
        }
      } else {
        for (int i = 0; i < 4; i++) {
          m.pred_mode[i] = (int)tree (TAG_PRED_MODE, LH264_TB_PREDMODE, (uint32_t) ((mbc * 8 + 6) * 9 + 1));
          if (m.pred_mode[i] > 8) { fail ("corrupt intra mode"); return false; }
        }
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) cache[kScan8[(i << 2) + j]] = (int8_t)m.pred_mode[i];
      }
      memcpy (my_ipm, cache + 1 + 8 * 4, 4);
      my_ipm[4] = cache[4 + 8 * 1]; my_ipm[5] = cache[4 + 8 * 2]; my_ipm[6] = cache[4 + 8 * 3];
      nxn_[k] = 1;
    } else nxn_[k] = 0;
    auto mvd = [&] (int blk) {
      m.mvd[blk][0] = scan_uegk (store_.get (LH264_TB_MVD, type * 16 + (uint32_t)blk), 9, 4, 3, 4, 3, TAG_MVX, TAG_MVX, TAG_MVX, TAG_MVX);
      m.mvd[blk][1] = scan_uegk (store_.get (LH264_TB_MVD, type * 16 + (uint32_t)blk), 9, 4, 3, 4, 3, TAG_MVY, TAG_MVY, TAG_MVY, TAG_MVY);
    };
    auto sub = [&] (int i) { m.sub_type[i] = (int)tree (TAG_SUB_MB, LH264_TB_SUBMB, (uint32_t)mbc); };
    auto ref = [&] (int i) { m.ref_idx[i] = (int)scan_raw_bits (TAG_REF, ref_bits); };
    if (type == LH264_MB_I8x8) {
      for (int i = 0; i < 4; i++) sub (i);
      for (int i = 0; i < 4; i++) ref (i);
    } else if (type == LH264_MB_P8x8 || type == LH264_MB_P8x8REF0) {
      for (int i = 0; i < 4; i++) sub (i);
      if (type == LH264_MB_P8x8) for (int i = 0; i < 4; i++) ref (i);
      for (int i = 0; i < 4; i++) {
        switch (m.sub_type[i]) {
        case LH264_SUB_8x8: mvd (kZ2Raster[i << 2]); break;
        case LH264_SUB_8x4: for (int j = 0; j < 2; j++) mvd (kZ2Raster[(i << 2) + (j << 1)]); break;
        case LH264_SUB_4x8: for (int j = 0; j < 2; j++) mvd (kZ2Raster[(i << 2) + j]); break;
        case LH264_SUB_4x4: for (int j = 0; j < 4; j++) mvd (kZ2Raster[(i << 2) + j]); break;
        default: fail ("corrupt sub-macroblock type"); return false;
        }
      }
    } else if (type == LH264_MB_P8x16 || type == LH264_MB_P16x8) {
      for (int i = 0; i < 2; i++) ref (i);
      for (int i = 0; i < 2; i++) mvd (type == LH264_MB_P16x8 ? i * 8 : i * 2);
    } else if (type == LH264_MB_P16x16) {
      ref (0);
      mvd (0);
    }
    {
      bool no_sub_lt8 = true;
      if (type == LH264_MB_P8x8 || type == LH264_MB_P8x8REF0) for (int i = 0; i < 4; i++) no_sub_lt8 = no_sub_lt8 && m.sub_type[i] == LH264_SUB_8x8;
      const bool is_inter = (type & LH264_MB_INTER) != 0;
      if (((type >= LH264_MB_P16x16 && type <= LH264_MB_P8x16) || no_sub_lt8) && is_inter && m.cbp_l > 0 && H.transform_8x8)
        m.t8 = scan_bit (TAG_T8, store_.get (LH264_TB_T8, (uint32_t) (mbc * 128 + m.luma_qp)));
      else m.t8 = type == LH264_MB_I8x8;
    }
    Cell e;
    e.initialized = 1; e.cbp_c = (uint8_t)m.cbp_c; e.cbp_l = (uint8_t)m.cbp_l; e.chroma_mode = (uint8_t)m.chroma_mode; e.luma16_mode = (uint8_t)m.luma16_mode;
    e.mb_type = type; e.num_ref = (uint32_t)m.num_ref; e.cached_skips = 0;
    decode_coeffs (m, st, mbc, x > 0 ? &cur[k - 1] : nullptr, k >= w ? &cur[k - w] : nullptr, np, e);
    if (failed_) return false;
    e.zeroed = 1;
    for (int i = 0; i < 384; i++) if (m.lev[i]) { e.zeroed = 0; break; }
    cur[k] = e;
    // the bits
    if (cabac) { write_mb_cabac (H, k, &m, qp_prev, last_dqp); ce_.terminate (has_stop); }
    else {
      if (is_p) { put_ue (pending_skips); pending_skips = 0; }
      write_mb (H, k, m, qp_prev);
    }
    if (has_stop) break;
  }
  if (cabac) {
    // the arithmetic codeword ends with the rbsp stop bit; the reference then overwrites the low 7 bits of its last byte with
    // what the compressor saw there (copySBitStringAux DS:1297-1333 with the 7 pad bits of DS:3133-3148)
    const unsigned pad_value = scan_raw_bits (TAG_PADBYTE, 7);
    if (ce_.bytes.empty()) { fail ("empty CABAC slice"); return false; }
    ce_.bytes.back() = (uint8_t) ((ce_.bytes.back() & 0x80) | pad_value);
    w_.append_bytes (ce_.bytes.data(), ce_.bytes.size());
    return !failed_;
  }
  if (pending_skips) put_ue (pending_skips);
  // rbsp_slice_trailing_bits: the stop bit, then the alignment bits as the compressor saw them (DS:3133-3148)
  const int pad_bits = 7 - (w_.bits_in_byte() & 7);
  const unsigned pad_value = pad_bits ? scan_raw_bits (TAG_PADBYTE, pad_bits) : 0;
  w_.emit_bit (1);
  w_.emit_bits (pad_value, pad_bits);
  return !failed_;
}

// the default stream, chunk by chunk as the console application feeds it back (h264dec.cpp:246-272, DC:658-860)
int Restorer::run (const uint8_t* d, size_t n, std::vector<uint8_t>& out) {
  size_t pos = 0;
  std::vector<uint8_t> nal;
  auto at = [&] (size_t i) -> int { return i < n ? d[i] : (i == n + 3 ? 1 : 0); };
  while (pos < n && !failed_) {
    size_t i;
    for (i = 0; i < n; i++) {
      if (i > 0 && at (pos + i) == 0 && at (pos + i + 1) == 0 && ((at (pos + i + 2) == 0 && at (pos + i + 3) == 1) || at (pos + i + 2) == 1)) break;
    }
    const size_t len = i;
    if (len < 4) { w_.append_bytes (d + pos, std::min (len, n - pos)); pos += len; continue; }
    const uint8_t* c = d + pos;
    pos += len;
    size_t off = 0; bool found = false;
    for (size_t q = 0, zeros = 0; q < len; q++) {
      if (c[q] == 0) { zeros++; continue; }
      if (c[q] == 1 && zeros >= 2) { off = q + 1; found = true; break; }
      zeros = 0;
    }
    if (!found) continue;
    w_.append_bytes (c, off);
    Parser::unescape (c + off, len - off, nal);
    w_.start_escape();
    size_t tz = 0;
    while (tz < nal.size() && nal[nal.size() - 1 - tz] == 0) tz++;
    if (!nal.empty() && !(nal[0] & 0x80)) {
      const int type = nal[0] & 31;
      w_.append_byte (nal[0]);
      std::vector<uint8_t> esc (c + off, c + len);            // the NAL as it stands in the stream, without trailing zero bytes
      while (!esc.empty() && esc.back() == 0) esc.pop_back();
      if (type == 1 || type == 5) {
        Parser::HeaderInfo H;
        // (a CABAC slice leaves only its header in the default stream, zero-padded to the byte: trailing zero bytes are header bits)
        std::vector<uint8_t> hb (c + off, c + len);
        hb.insert (hb.end(), 4, 0);
        if (hdr_.parse_headers (hb.data(), hb.size(), H) < 0 || !H.is_slice) { fail ("cannot parse a slice header of the default stream (" + hdr_.error() + ")"); break; }
        if (H.cabac) {
          // The header of a CABAC slice is all its NAL keeps in the default stream, zero-padded to the byte.  When it ends in zero
          // bytes, cutting the stream at start codes hands those to the following chunk as if they were zeros before a start
          // code (the reference then writes them in the wrong place and cannot restore such streams): take them back.
          size_t need = 1 + ((size_t)H.hdr_bits + 7) / 8;
          for (size_t have = nal.size(); have < need && pos < n && d[pos] == 0; have++) pos++;
        }
        const std::vector<uint8_t>& rb = hdr_.last_rbsp();
        for (int b = 0; b < H.hdr_bits; b++) w_.emit_bit ((rb[(size_t)b >> 3] >> (7 - (b & 7))) & 1);
        if (!decode_slice (H)) break;
      } else {
        if (type == 7 || type == 8) { Parser::HeaderInfo H; hdr_.parse_headers (esc.data(), esc.size(), H); }
        if (nal.size() > 1 + tz) w_.append_bytes (nal.data() + 1, nal.size() - 1 - tz);
        for (size_t q = 0; q < tz; q++) w_.append_byte (0);
      }
    }
    w_.stop_escape();
  }
  if (failed_) return -1;
  w_.pad_to_byte();
  out.swap (w_.buffer);
  return 0;
}

}  // namespace

int pip_restore (const uint8_t* main_stream, size_t main_len, const uint8_t* const* tags, const size_t* tag_len, int n_tags,
                 std::vector<uint8_t>& out, std::string& err) {
  err.clear();
  if (!main_stream || !tags || !tag_len) { err = "null argument"; return -1; }
  try {
    Restorer r (tags, tag_len, n_tags, err);
    return r.run (main_stream, main_len, out);
  } catch (const std::exception& e) {      // allocation failures etc. on hostile input: reported, never thrown across the C ABI
    err = std::string ("internal: ") + e.what();
    return -1;
  }
}

}  // namespace lh264host
