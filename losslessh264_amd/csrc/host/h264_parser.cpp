// h264_parser.cpp - see h264_parser.h.  ITU-T H.264 clauses 7.3 (syntax), 8.2.4 (reference lists), 8.3.1.1 / 8.4.1
// (intra mode and motion vector prediction), 9.2 (CAVLC).  Frame (progressive) pictures, I and P slices, one slice
// group -- the subset the reference decoder itself supports.
#include "h264_parser.h"
#include <mutex>
#include <new>
#include <stdexcept>
#include <atomic>
#include <thread>
#include <functional>
#include "pip_symbols.h"
#include "h264_cabac_tables.h"
#include <string.h>
#include <algorithm>
#include "h264_tables.h"
#include "h264_vlc_tables.h"

namespace lh264host {

namespace {
// released coefficient planes, matched by exact size.  Per thread first (no lock); what a thread cannot keep, and what it
// leaves behind when it ends, goes to a shared pool so that batches parsed by short-lived worker threads still recycle.
struct SharedZeroPool {
  std::mutex m;
  std::map<size_t, std::vector<void*>> by_size;
  size_t bytes = 0;
  ~SharedZeroPool() { for (auto& kv : by_size) for (void* p : kv.second) free (p); }
  void* get (size_t b) {
    std::lock_guard<std::mutex> g (m);
    auto it = by_size.find (b);
    if (it == by_size.end() || it->second.empty()) return nullptr;
    void* p = it->second.back(); it->second.pop_back(); bytes -= b;
    return p;
  }
  bool put (void* p, size_t b) {
    std::lock_guard<std::mutex> g (m);
    if (bytes + b > ((size_t)4 << 30) / 16) return false;
    by_size[b].push_back (p); bytes += b;
    return true;
  }
};
// sixteen of them: released blocks are dealt round robin, a thread looks in "its" shard first - the pictures of a batch are
// released by one thread and taken by sixteen, and one lock would serialise them
enum { kShards = 16 };
struct ShardedPool {
  SharedZeroPool shard[kShards];
  std::atomic<unsigned> rr {0};
  void* get (size_t b) {
    static thread_local unsigned mine = (unsigned)std::hash<std::thread::id>() (std::this_thread::get_id());
    for (unsigned i = 0; i < kShards; i++) if (void* p = shard[(mine + i) % kShards].get (b)) return p;
    return nullptr;
  }
  bool put (void* p, size_t b) { return shard[rr.fetch_add (1, std::memory_order_relaxed) % kShards].put (p, b); }
};
ShardedPool& shared_pool() { static ShardedPool p; return p; }
struct ZeroCache {
  enum { kKeep = 256 };
  struct E { void* p; size_t bytes; } e[kKeep];
  int n = 0;
  ~ZeroCache() { for (int i = 0; i < n; i++) if (!shared_pool().put (e[i].p, e[i].bytes)) free (e[i].p); }
};
thread_local ZeroCache g_zero_cache;
}
void* zerobuf_get (size_t bytes, bool zero) {
  ZeroCache& c = g_zero_cache;
  void* p = nullptr;
  for (int i = c.n - 1; i >= 0; i--) if (c.e[i].bytes == bytes) { p = c.e[i].p; c.e[i] = c.e[--c.n]; break; }
  if (!p && bytes >= 4096) p = shared_pool().get (bytes);
  if (p) { if (zero) memset (p, 0, bytes); return p; }
  p = zero ? calloc (1, bytes) : malloc (bytes);
  if (!p) throw std::bad_alloc();            // caught at the parser's entry points (feed_nal / flush), never crosses the C ABI
  return p;
}
void zerobuf_put (void* p, size_t bytes) {
  ZeroCache& c = g_zero_cache;
  if (c.n < ZeroCache::kKeep && bytes <= (size_t)8 << 20) { c.e[c.n].p = p; c.e[c.n].bytes = bytes; c.n++; }
  else if (bytes < 4096 || !shared_pool().put (p, bytes)) free (p);
}

StreamArena*& current_stream_arena() { static thread_local StreamArena* a = nullptr; return a; }
StreamArena::~StreamArena() { for (auto& c : chunks) zerobuf_put (c.first, c.second); }
void* StreamArena::alloc (size_t bytes) {
  bytes = (bytes + 63) & ~ (size_t)63;
  if (chunks.empty() || used + bytes > chunks.back().second) {
    const size_t cap = std::max<size_t> ((size_t)1 << 20, bytes);
    chunks.emplace_back ((char*)zerobuf_get (cap, false), cap);
    used = 0;
  }
  void* p = chunks.back().first + used;
  used += bytes;
  return p;
}

bool BitReader::more_rbsp_data() const {
  if (pos >= nbits) return false;
  // find the last set bit of the payload (the rbsp stop bit)
  size_t last = nbits;
  while (last > pos) {
    size_t q = last - 1;
    if ((p[q >> 3] >> (7 - (q & 7))) & 1) return q > pos;
    last--;
  }
  return false;
}

namespace {

// z-order 4x4 index <-> raster (x,y)
inline int z2x (int z) { return (z & 1) | ((z >> 2) & 1) << 1; }
inline int z2y (int z) { return ((z >> 1) & 1) | ((z >> 3) & 1) << 1; }
inline int xy2z (int x, int y) { return (x & 1) | ((y & 1) << 1) | ((x >> 1) << 2) | ((y >> 1) << 3); }
inline bool zidx_before (int x, int y, int z) { return xy2z (x, y) < z; }
const int kChromaNzcIdx[2][4] = {{16, 17, 20, 21}, {18, 19, 22, 23}};   // reference nzc layout (common_tables.cpp:39-47)

struct DpbPic {
  int frame_id = -1, frame_num = 0, frame_num_wrap = 0, long_idx = -1;
  bool is_long = false;
};

struct MbState {       // per-macroblock parse state of the current picture (neighbour context)
  int16_t slice = -1;
  uint8_t type_class = 0;   // 0 not decoded, 1 intra NxN, 2 intra other, 3 inter
  int8_t ipm[16];           // Intra4x4PredMode per raster 4x4 (I8x8: replicated), 2 otherwise
  int8_t ref[4];
  int16_t mv[16][2];
  // CABAC context inputs (9.3.3.1.1)
  uint8_t skip = 0, pcm = 0, t8 = 0, cbp = 0, chroma_pred = 0;
  uint8_t mvd[16][2];       // |mvd| per 4x4 block, saturated
  uint32_t cbf = 0;         // coded_block_flag: bits 0..15 luma 4x4 (raster), 16 luma DC, 17/18 chroma DC, 19..22 Cb AC, 23..26 Cr AC
};

// ---- CABAC arithmetic decoding engine (9.3.1.2, 9.3.3.2) -----------------------------------------------------------------
struct Cabac {
  const uint8_t* p = nullptr; size_t nbits = 0, pos = 0;
  uint32_t range = 510, offset = 0;
  uint8_t state[460];        // pStateIdx << 1 | valMPS
  bool err = false;
  inline uint32_t bit() { if (pos >= nbits) { pos++; if (pos > nbits + 64) err = true; return 0; } const uint32_t b = (p[pos >> 3] >> (7 - (pos & 7))) & 1; pos++; return b; }
  void start (const uint8_t* d, size_t bytes, size_t bitpos) {
    p = d; nbits = bytes * 8; pos = bitpos; range = 510; offset = 0;
    for (int i = 0; i < 9; i++) offset = (offset << 1) | bit();
  }
  void init_contexts (int col, int qp) {
    qp = std::min (51, std::max (0, qp));
    for (int i = 0; i < 460; i++) {
      const int m = kCabacInit[i][col][0], n = kCabacInit[i][col][1];
      const int pre = std::min (126, std::max (1, ((m * qp) >> 4) + n));
      state[i] = pre <= 63 ? (uint8_t) ((63 - pre) << 1) : (uint8_t) (((pre - 64) << 1) | 1);
    }
  }
  inline int decode (int ctx) {
    uint8_t& s = state[ctx];
    const int st = s >> 1; int mps = s & 1;
    const uint32_t lps = kCabacRangeLps[st][(range >> 6) & 3];
    range -= lps;
    int b;
    if (offset >= range) {
      b = !mps; offset -= range; range = lps;
      if (st == 0) mps = !mps;
      s = (uint8_t) ((kCabacNextLps[st] << 1) | mps);
    } else { b = mps; s = (uint8_t) ((kCabacNextMps[st] << 1) | mps); }
    while (range < 256) { range <<= 1; offset = (offset << 1) | bit(); }
    return b;
  }
  inline int bypass() { offset = (offset << 1) | bit(); if (offset >= range) { offset -= range; return 1; } return 0; }
  inline int terminate() {
    range -= 2;
    if (offset >= range) return 1;
    while (range < 256) { range <<= 1; offset = (offset << 1) | bit(); }
    return 0;
  }
};

}  // namespace

struct Parser::Impl {
  Parser* self;
  std::map<int, Sps> sps;
  std::map<int, Pps> pps;
  std::vector<DpbPic> dpb;
  int next_frame_id = 0;
  // picture in progress
  std::unique_ptr<FrameOut> cur;
  std::vector<MbState> st;
  const Sps* csps = nullptr; const Pps* cpps = nullptr;
  SliceHeader first_sh;
  int last_first_mb = -1;
  int prev_ref_frame_num = 0;
  bool cur_is_long = false; int cur_long_idx = -1; bool had_mmco5 = false;
  std::vector<uint8_t> rbsp;
  // the decoder's persistent per-position arrays that the recompressor reads for every coded macroblock
  std::vector<uint8_t> persist_chroma, persist_l16, persist_sub;
  int persist_w = 0, persist_h = 0;
  int slice_cached_qp = 0, slice_run_before = 0;
  Symbolizer symbolizer;
  int last_hdr_bits = -1; bool last_cabac = false;
  int16_t no_coef[384];                                  // where the dequantised coefficients go when nobody wants them
  alignas (8) int16_t lev_scratch[384] = {};                  // sparse mode: the macroblock in hand, turned into list entries when it is done
  // a coded macroblock is done: note whether it has any nonzero level; in sparse mode list them (picture-relative index << 16 | value)
  uint32_t lev_mask = 0;                                 // sparse mode: the 16-coefficient blocks of lev_scratch written for the macroblock in hand
  void finish_levels (int k) {
    bool any = false;
    if (self->sparse_levels_) {
      // only the blocks that were written are looked at, and they are zero again afterwards
      for (uint32_t msk = lev_mask; msk; msk &= msk - 1) {
        int16_t* b = lev_scratch + 16 * __builtin_ctz (msk);
        const size_t at = (size_t)k * 384 + (size_t) (b - lev_scratch);
        for (int j = 0; j < 16; j++) if (b[j]) { any = true; cur->sparse.push_back (((uint64_t) (at + j) << 16) | (uint16_t)b[j]); b[j] = 0; }
      }
      lev_mask = 0;
    } else {
      const int16_t* lv = &cur->levels[(size_t)k * 384];
      for (int i = 0; i < 96 && !any; i++) { uint64_t q; memcpy (&q, lv + 4 * i, 8); any = q != 0; }
    }
    cur->lev_nonzero[k] = any ? 1 : 0;
  }       // the slice NAL just handled: header length in bits, entropy mode

  explicit Impl (Parser* s) : self (s) {}

  // ---- helpers -------------------------------------------------------------------------------------------------
  void fail (const std::string& m) { if (self->err_.empty()) self->err_ = m; }

  static void unescape (const uint8_t* d, size_t n, std::vector<uint8_t>& out) {
    out.clear(); out.reserve (n);
    int zeros = 0;
    for (size_t i = 0; i < n; i++) {
      if (zeros >= 2 && d[i] == 3) { zeros = 0; continue; }
      out.push_back (d[i]);
      zeros = d[i] == 0 ? zeros + 1 : 0;
    }
  }

  static void parse_scaling_list (BitReader& br, uint8_t* dst, int n, const uint8_t* fallback_raster, const uint8_t* def_zz, bool& use_default) {
    int last = 8, next = 8;
    use_default = false;
    const uint8_t* zz = n == 16 ? kZigzag4x4 : kZigzag8x8;
    for (int j = 0; j < n; j++) {
      if (next != 0) {
        int delta = br.se();
        next = (last + delta + 256) % 256;
        if (j == 0 && next == 0) { use_default = true; break; }
      }
      dst[zz[j]] = (uint8_t) (next == 0 ? last : next);
      last = dst[zz[j]];
    }
    if (use_default) for (int j = 0; j < n; j++) dst[zz[j]] = def_zz[j];
    (void)fallback_raster;
  }
  // scaling matrices of an SPS (fall-back rule A) or PPS (fall-back rule B = the SPS's lists), 7.3.2.1.1 / Table 7-2
  static void parse_scaling_matrix (BitReader& br, uint8_t sl4[6][16], uint8_t sl8[2][64], int n_lists, const Sps* fb_sps) {
    for (int i = 0; i < n_lists; i++) {
      bool present = br.u1();
      bool use_def = false;
      if (i < 6) {
        if (present) parse_scaling_list (br, sl4[i], 16, nullptr, kDefaultScaling4x4[i < 3 ? 0 : 1], use_def);
        else if (i == 0 || i == 3) {
          if (fb_sps) memcpy (sl4[i], fb_sps->sl4[i], 16);
          else for (int j = 0; j < 16; j++) sl4[i][kZigzag4x4[j]] = kDefaultScaling4x4[i < 3 ? 0 : 1][j];
        } else memcpy (sl4[i], sl4[i - 1], 16);
      } else {
        const int k = i - 6;
        if (present) parse_scaling_list (br, sl8[k], 64, nullptr, kDefaultScaling8x8[k], use_def);
        else if (fb_sps) memcpy (sl8[k], fb_sps->sl8[k], 64);
        else for (int j = 0; j < 64; j++) sl8[k][kZigzag8x8[j]] = kDefaultScaling8x8[k][j];
      }
    }
  }

  void parse_sps (BitReader& br) {
    Sps s;
    s.profile_idc = br.u (8); br.u (8); s.level_idc = br.u (8);
    const uint32_t id = br.ue();
    if (id > 31) { fail ("seq_parameter_set_id out of range"); return; }
    memset (s.sl4, 16, sizeof (s.sl4)); memset (s.sl8, 16, sizeof (s.sl8));
    if (s.profile_idc == 100 || s.profile_idc == 110 || s.profile_idc == 122 || s.profile_idc == 244 || s.profile_idc == 44 ||
        s.profile_idc == 83 || s.profile_idc == 86 || s.profile_idc == 118 || s.profile_idc == 128) {
      const uint32_t cfi = br.ue();
      if (cfi > 3) { fail ("chroma_format_idc out of range"); return; }
      s.chroma_format_idc = (int)cfi;
      if (s.chroma_format_idc == 3) br.u1();
      const uint32_t bdl = br.ue(), bdc = br.ue();
      br.u1();
      if (s.chroma_format_idc != 1 || bdl || bdc) { self->n_unsupported_++; fail ("unsupported chroma format / bit depth"); return; }
      s.scaling_matrix_present = br.u1();
      if (s.scaling_matrix_present) parse_scaling_matrix (br, s.sl4, s.sl8, 8, nullptr);
    }
    // every Exp-Golomb value is range-checked as unsigned BEFORE it is narrowed (H.264 7.4.2.1.1 limits)
    const uint32_t l2fn = br.ue();
    if (l2fn > 12) { fail ("log2_max_frame_num out of range"); return; }
    s.log2_max_frame_num = (int)l2fn + 4;
    const uint32_t pt = br.ue();
    if (pt > 2) { fail ("pic_order_cnt_type out of range"); return; }
    s.poc_type = (int)pt;
    if (s.poc_type == 0) {
      const uint32_t l2p = br.ue();
      if (l2p > 12) { fail ("log2_max_pic_order_cnt_lsb out of range"); return; }
      s.log2_max_poc_lsb = (int)l2p + 4;
    } else if (s.poc_type == 1) {
      s.delta_pic_order_always_zero = br.u1();
      s.offset_for_non_ref_pic = br.se(); s.offset_for_top_to_bottom = br.se();
      const uint32_t nc = br.ue();
      if (nc > 255) { fail ("num_ref_frames_in_pic_order_cnt_cycle out of range"); return; }
      s.num_ref_frames_in_poc_cycle = (int)nc;
      for (int i = 0; i < s.num_ref_frames_in_poc_cycle; i++) s.offset_for_ref_frame.push_back (br.se());
    }
    const uint32_t nrf = br.ue();
    if (nrf > 16) { fail ("max_num_ref_frames out of range"); return; }
    s.num_ref_frames = (int)nrf;
    s.gaps_allowed = br.u1();
    const uint32_t mbw = br.ue(), mbh = br.ue();
    // level 6.2 allows 139,264 macroblocks per frame and 16,384 samples per side (A.3.1): nothing larger is a picture
    if (mbw >= 1024 || mbh >= 1024 || (uint64_t) (mbw + 1) * (mbh + 1) > 139264u) { fail ("picture size out of range"); return; }
    s.mb_w = (int)mbw + 1;
    s.mb_h = (int)mbh + 1;
    s.frame_mbs_only = br.u1();
    if (!s.frame_mbs_only) { br.u1(); self->n_unsupported_++; fail ("interlaced coding is not supported"); return; }
    s.direct_8x8 = br.u1();
    if (br.u1()) {
      const uint32_t cl = br.ue(), cr = br.ue(), ct = br.ue(), cb = br.ue();
      if ((uint64_t)cl + cr >= (uint64_t)s.mb_w * 8 || (uint64_t)ct + cb >= (uint64_t)s.mb_h * 8) { fail ("frame cropping outside the picture"); return; }
      s.crop_l = (int)cl; s.crop_r = (int)cr; s.crop_t = (int)ct; s.crop_b = (int)cb;
    }
    if (br.err) { fail ("truncated SPS"); return; }
    s.valid = true;
    sps[id] = s;
  }

  void parse_pps (BitReader& br) {
    Pps p;
    const uint32_t id = br.ue(), sid = br.ue();
    if (id > 255 || sid > 31) { fail ("parameter set id out of range"); return; }
    p.sps_id = (int)sid;
    p.cabac = br.u1();
    p.pic_order_present = br.u1();
    const uint32_t nsg = br.ue();
    if (nsg > 7) { fail ("num_slice_groups out of range"); return; }
    p.num_slice_groups = (int)nsg + 1;
    if (p.num_slice_groups > 1) { self->n_unsupported_++; fail ("FMO (slice groups) is not supported"); return; }
    const uint32_t nr0 = br.ue(), nr1 = br.ue();
    if (nr0 > 31 || nr1 > 31) { fail ("num_ref_idx_default_active out of range"); return; }
    p.num_ref_idx_l0 = (int)nr0 + 1; p.num_ref_idx_l1 = (int)nr1 + 1;
    p.weighted_pred = br.u1(); p.weighted_bipred_idc = br.u (2);
    p.pic_init_qp = 26 + br.se(); p.pic_init_qs = 26 + br.se();
    p.chroma_qp_offset[0] = p.chroma_qp_offset[1] = br.se();
    p.deblocking_control = br.u1(); p.constrained_intra_pred = br.u1(); p.redundant_pic_cnt = br.u1();
    memset (p.sl4, 16, sizeof (p.sl4)); memset (p.sl8, 16, sizeof (p.sl8));
    auto it = sps.find (p.sps_id);
    if (it != sps.end()) { memcpy (p.sl4, it->second.sl4, sizeof (p.sl4)); memcpy (p.sl8, it->second.sl8, sizeof (p.sl8)); }
    if (br.more_rbsp_data()) {
      p.transform_8x8 = br.u1();
      p.scaling_matrix_present = br.u1();
      if (p.scaling_matrix_present)
        parse_scaling_matrix (br, p.sl4, p.sl8, 6 + (p.transform_8x8 ? 2 : 0), (it != sps.end() && it->second.scaling_matrix_present) ? &it->second : nullptr);
      p.chroma_qp_offset[1] = br.se();
    }
    if (br.err) { fail ("truncated PPS"); return; }
    p.valid = true;
    pps[id] = p;
  }

  bool parse_slice_header (BitReader& br, int nal_type, int nal_ref_idc, SliceHeader& sh) {
    sh.idr = nal_type == 5; sh.nal_ref_idc = nal_ref_idc;
    const uint32_t fmb = br.ue();
    if (fmb >= 139264u) { fail ("first_mb_in_slice out of range"); return false; }
    sh.first_mb = (int)fmb;
    const uint32_t stu = br.ue();
    if (stu > 9) { fail ("slice_type out of range"); return false; }
    int st = (int)stu;
    if (st >= 5) st -= 5;
    if (st != 0 && st != 2) { self->n_unsupported_++; fail ("only I and P slices are supported"); return false; }
    sh.slice_type = st;
    const uint32_t pid = br.ue();
    if (pid > 255) { fail ("pic_parameter_set_id out of range"); return false; }
    sh.pps_id = (int)pid;
    auto pit = pps.find (sh.pps_id);
    if (pit == pps.end() || !pit->second.valid) { fail ("slice refers to a missing PPS"); return false; }
    const Pps& P = pit->second;
    auto sit = sps.find (P.sps_id);
    if (sit == sps.end() || !sit->second.valid) { fail ("slice refers to a missing SPS"); return false; }
    const Sps& S = sit->second;
    sh.frame_num = br.u (S.log2_max_frame_num);
    if (sh.idr) { const uint32_t v = br.ue(); if (v > 65535) { fail ("idr_pic_id out of range"); return false; } sh.idr_pic_id = (int)v; }
    if (S.poc_type == 0) { sh.poc_lsb = br.u (S.log2_max_poc_lsb); if (P.pic_order_present) sh.delta_poc_bottom = br.se(); }
    else if (S.poc_type == 1 && !S.delta_pic_order_always_zero) { sh.delta_poc[0] = br.se(); if (P.pic_order_present) sh.delta_poc[1] = br.se(); }
    if (P.redundant_pic_cnt) { const uint32_t v = br.ue(); if (v > 127) { fail ("redundant_pic_cnt out of range"); return false; } sh.redundant_pic_cnt = (int)v; }
    sh.num_ref_idx_l0 = P.num_ref_idx_l0;
    if (st == 0) {
      if (br.u1()) {
        const uint32_t nr = br.ue();
        if (nr > 31) { fail ("num_ref_idx_l0_active out of range"); return false; }
        sh.num_ref_idx_l0 = (int)nr + 1;
      }
      if (br.u1()) {            // ref_pic_list_modification_flag_l0
        for (;;) {
          const uint32_t idcu = br.ue();
          if (idcu == 3 || br.err) break;
          if (idcu > 3) { fail ("bad ref list modification"); return false; }
          const int idc = (int)idcu;
          const uint32_t rv = br.ue();
          if (rv > 0x7fffffffu) { fail ("bad ref list modification"); return false; }
          sh.reorder.push_back ({idc, rv});
          if (sh.reorder.size() > 64) { fail ("bad ref list modification"); return false; }
        }
      }
      if (P.weighted_pred) {
        sh.has_weights = true;
        const uint32_t ld = br.ue(), cd = br.ue();
        if (ld > 7 || cd > 7) { fail ("weight denominator out of range"); return false; }
        sh.luma_log2_denom = (int)ld; sh.chroma_log2_denom = (int)cd;
        for (int i = 0; i < sh.num_ref_idx_l0 && i < 32; i++) {
          sh.luma_weight[i] = 1 << sh.luma_log2_denom; sh.luma_offset[i] = 0;
          for (int c = 0; c < 2; c++) { sh.chroma_weight[i][c] = 1 << sh.chroma_log2_denom; sh.chroma_offset[i][c] = 0; }
          if (br.u1()) { sh.luma_weight[i] = br.se(); sh.luma_offset[i] = br.se(); }
          if (br.u1()) for (int c = 0; c < 2; c++) { sh.chroma_weight[i][c] = br.se(); sh.chroma_offset[i][c] = br.se(); }
        }
      }
    }
    if (nal_ref_idc) {
      if (sh.idr) { sh.no_output_of_prior = br.u1(); sh.long_term_reference = br.u1(); }
      else if ((sh.adaptive_marking = br.u1())) {
        for (;;) {
          const uint32_t opu = br.ue();
          if (opu == 0 || br.err) break;
          if (opu > 6) { fail ("bad mmco list"); return false; }
          const int op = (int)opu;
          SliceHeader::Mmco m = {op, 0, 0};
          auto small = [&] () -> int { const uint32_t v = br.ue(); if (v > 0x00ffffffu) br.err = true; return (int) (v & 0x00ffffffu); };
          if (op == 1 || op == 3) m.a = small();
          if (op == 2) m.a = small();
          if (op == 3 || op == 6) m.b = small();
          if (op == 4) m.a = small();
          sh.mmco.push_back (m);
          if (sh.mmco.size() > 66) { fail ("bad mmco list"); return false; }
        }
      }
    }
    if (P.cabac && st != 2) { const uint32_t ci = br.ue(); sh.cabac_init_idc = (int) (ci & 3); if (ci > 2) { fail ("cabac_init_idc out of range"); return false; } }
    sh.slice_qp = P.pic_init_qp + br.se();
    if (P.deblocking_control) {
      const uint32_t di = br.ue();
      if (di > 2) { fail ("disable_deblocking_filter_idc out of range"); return false; }
      sh.deblock_idc = (int)di;
      if (sh.deblock_idc != 1) {
        const int ao = br.se(), bo = br.se();
        if (ao < -6 || ao > 6 || bo < -6 || bo > 6) { fail ("deblocking offsets out of range"); return false; }
        sh.alpha_off = ao * 2; sh.beta_off = bo * 2;
      }
    }
    return !br.err;
  }

  // ---- reference lists (8.2.4) ----------------------------------------------------------------------------------
  void build_ref_list (const SliceHeader& sh, const Sps& S, std::vector<int>& list /* frame ids, -1 = none */) {
    const int max_fn = 1 << S.log2_max_frame_num;
    std::vector<DpbPic*> sterm, lterm;
    for (auto& d : dpb) {
      if (d.is_long) lterm.push_back (&d);
      else { d.frame_num_wrap = d.frame_num > sh.frame_num ? d.frame_num - max_fn : d.frame_num; sterm.push_back (&d); }
    }
    std::sort (sterm.begin(), sterm.end(), [] (DpbPic * a, DpbPic * b) { return a->frame_num_wrap > b->frame_num_wrap; });
    std::sort (lterm.begin(), lterm.end(), [] (DpbPic * a, DpbPic * b) { return a->long_idx < b->long_idx; });
    std::vector<DpbPic*> l;
    l.insert (l.end(), sterm.begin(), sterm.end()); l.insert (l.end(), lterm.begin(), lterm.end());
    const int n = sh.num_ref_idx_l0;
    l.resize (std::max ((size_t)n + 1, l.size()), nullptr);
    int pred = sh.frame_num, idx = 0;
    for (const auto& r : sh.reorder) {
      DpbPic* target = nullptr;
      if (r.idc == 0 || r.idc == 1) {
        int nowrap = r.idc == 0 ? pred - (int) (r.val + 1) : pred + (int) (r.val + 1);
        if (r.idc == 0 && nowrap < 0) nowrap += max_fn;
        if (r.idc == 1 && nowrap >= max_fn) nowrap -= max_fn;
        pred = nowrap;
        const int picnum = nowrap > sh.frame_num ? nowrap - max_fn : nowrap;
        for (auto* d : sterm) if (d->frame_num_wrap == picnum) target = d;
      } else if (r.idc == 2) {
        for (auto* d : lterm) if (d->long_idx == (int)r.val) target = d;
      }
      if (!target || idx > n) continue;
      l.insert (l.begin() + idx, target);
      int w = idx + 1;
      for (size_t k = idx + 1; k < l.size(); k++) if (l[k] != target) l[w++] = l[k];
      l.resize (w);
      l.resize (std::max ((size_t)n + 1, l.size()), nullptr);
      idx++;
    }
    list.clear();
    for (int i = 0; i < n; i++) list.push_back (i < (int)l.size() && l[i] ? l[i]->frame_id : -1);
  }

  void mark_reference (const SliceHeader& sh, const Sps& S) {
    const int max_fn = 1 << S.log2_max_frame_num;
    if (sh.idr) {
      dpb.clear();
      DpbPic d; d.frame_id = cur->id; d.frame_num = sh.frame_num;
      if (sh.long_term_reference) { d.is_long = true; d.long_idx = 0; }
      dpb.push_back (d);
      return;
    }
    bool cur_long = false; int cur_lidx = -1;
    if (sh.adaptive_marking) {
      for (auto& d : dpb) if (!d.is_long) d.frame_num_wrap = d.frame_num > sh.frame_num ? d.frame_num - max_fn : d.frame_num;
      for (const auto& m : sh.mmco) {
        const int picx = sh.frame_num - (int) (m.a + 1);
        if (m.op == 1) { for (size_t i = 0; i < dpb.size(); i++) if (!dpb[i].is_long && dpb[i].frame_num_wrap == picx) { dpb.erase (dpb.begin() + i); break; } }
        else if (m.op == 2) { for (size_t i = 0; i < dpb.size(); i++) if (dpb[i].is_long && dpb[i].long_idx == (int)m.a) { dpb.erase (dpb.begin() + i); break; } }
        else if (m.op == 3) {
          for (size_t i = 0; i < dpb.size(); i++) if (dpb[i].is_long && dpb[i].long_idx == (int)m.b) { dpb.erase (dpb.begin() + i); break; }
          for (auto& d : dpb) if (!d.is_long && d.frame_num_wrap == picx) { d.is_long = true; d.long_idx = (int)m.b; break; }
        } else if (m.op == 4) {
          for (size_t i = 0; i < dpb.size();) { if (dpb[i].is_long && dpb[i].long_idx >= (int)m.a) dpb.erase (dpb.begin() + i); else i++; }
        } else if (m.op == 5) { dpb.clear(); had_mmco5 = true; }
        else if (m.op == 6) {
          for (size_t i = 0; i < dpb.size(); i++) if (dpb[i].is_long && dpb[i].long_idx == (int)m.b) { dpb.erase (dpb.begin() + i); break; }
          cur_long = true; cur_lidx = (int)m.b;
        }
      }
    } else {
      const size_t cap = (size_t)std::max (S.num_ref_frames, 1);
      while (dpb.size() >= cap) {       // sliding window: drop the short-term picture with the smallest FrameNumWrap
        int best = -1;
        for (size_t i = 0; i < dpb.size(); i++) {
          if (dpb[i].is_long) continue;
          dpb[i].frame_num_wrap = dpb[i].frame_num > sh.frame_num ? dpb[i].frame_num - max_fn : dpb[i].frame_num;
          if (best < 0 || dpb[i].frame_num_wrap < dpb[best].frame_num_wrap) best = (int)i;
        }
        if (best < 0) break;
        dpb.erase (dpb.begin() + best);
      }
    }
    DpbPic d; d.frame_id = cur->id; d.frame_num = had_mmco5 ? 0 : sh.frame_num; d.is_long = cur_long; d.long_idx = cur_lidx;
    dpb.push_back (d);
    const size_t cap = (size_t)std::max (S.num_ref_frames, 1);
    while (dpb.size() > cap) {          // never exceed the DPB size the stream declared
      size_t best = 0; bool found = false;
      for (size_t i = 0; i + 1 < dpb.size(); i++) if (!dpb[i].is_long && (!found || dpb[i].frame_num_wrap < dpb[best].frame_num_wrap)) { best = i; found = true; }
      dpb.erase (dpb.begin() + (found ? best : 0));
    }
  }

  // ---- picture management ---------------------------------------------------------------------------------------
  void finish_picture() {
    if (!cur) return;
    if (first_sh.nal_ref_idc) { mark_reference (first_sh, *csps); cur->is_ref = true; prev_ref_frame_num = had_mmco5 ? 0 : first_sh.frame_num; }
    had_mmco5 = false;
    for (const auto& d : dpb) cur->dpb_ids.push_back (d.frame_id);
    {   // the row-a10 symbol lists (every slice of the picture must have been parsed to its end)
      if (cur->slice_syn.size() == cur->slices.size()) symbolizer.picture (*cur);
      else { cur->syn_off.assign ((size_t)cur->mb_w * cur->mb_h + 1, 0); cur->syn_syms.clear(); }
    }
    for (uint8_t c : cur->covered) if (!c) { self->damaged_ = true; break; }
    cur->complete = true;
    self->pictures_done_++;
    if (self->keep_frames_) self->frames_.push_back (std::move (cur));
    cur.reset();
    last_first_mb = -1;
  }

  void start_picture (const SliceHeader& sh, const Sps& S, const Pps& P) {
    cur.reset (new FrameOut());
    cur->id = next_frame_id++;
    cur->mb_w = S.mb_w; cur->mb_h = S.mb_h; cur->frame_num = sh.frame_num; cur->idr = sh.idr;
    cur->idr_pic_id = sh.idr_pic_id; cur->nal_ref_idc = sh.nal_ref_idc;
    cur->crop_x = 2 * S.crop_l; cur->crop_y = 2 * S.crop_t;
    cur->crop_w = S.mb_w * 16 - 2 * (S.crop_l + S.crop_r); cur->crop_h = S.mb_h * 16 - 2 * (S.crop_t + S.crop_b);
    const size_t n = (size_t)S.mb_w * S.mb_h;
    cur->mbs.assign (n, lh264_mb_t()); memset (cur->mbs.data(), 0, n * sizeof (lh264_mb_t));
    if (self->want_coeffs_) cur->coeffs.assign_zero ((size_t)n * 384);
    if (!self->sparse_levels_) cur->levels.assign_zero ((size_t)n * 384, !self->lazy_levels_);
    cur->lev_nonzero.assign (n, 0); cur->covered.assign (n, 0);
    cur->syn.assign (n, MbSyn()); memset (cur->syn.data(), 0, n * sizeof (MbSyn));
    if (persist_w != S.mb_w || persist_h != S.mb_h) {       // the decoder re-allocates (zeroed) on a resolution change
      persist_w = S.mb_w; persist_h = S.mb_h;
      persist_chroma.assign (n, 0); persist_l16.assign (n, 0); persist_sub.assign (n * 4, 0);
    }
    st.assign (n, MbState());
    csps = &S; cpps = &P; first_sh = sh;
    if (sh.idr) { /* the DPB is cleared when the IDR picture is marked */ }
  }

  // ---- CAVLC residual block (9.2) ------------------------------------------------------------------------------------
  struct TokLut { uint16_t t[5][1024]; };          // total_coeff << 8 | trailing_ones << 4 | length, by the next 10 bits
  static const TokLut& tok_lut() {
    static const TokLut lut = [] {
      TokLut L; memset (&L, 0, sizeof (L));
      for (int tab = 0; tab < 5; tab++) if (tab != 3) for (int i = 0; i < kCoeffTokenCount[tab]; i++) {
            const VlcTok& t = kCoeffToken[tab][i];
            if (t.len == 0 || t.len > 10) continue;
            const uint32_t lo = (uint32_t)t.code << (10 - t.len);
            for (uint32_t q = 0; q < (1u << (10 - t.len)); q++) L.t[tab][lo + q] = (uint16_t) (t.total_coeff << 8 | t.trailing_ones << 4 | t.len);
          }
      return L;
    }();
    return lut;
  }
  // returns total_coeff, writes levels in scan order positions start..; -1 on error
  int residual_block (BitReader& br, int nC, int max_coeff, int* level /*[16] in scan order (index = position in the block's scan)*/) {
    const int tab = nC < 0 ? 4 : nC < 2 ? 0 : nC < 4 ? 1 : nC < 8 ? 2 : 3;
    int total = -1, t1 = 0;
    if (tab == 3) {                     // 6-bit fixed length code
      uint32_t v = br.u (6);
      if (v == 3) { total = 0; t1 = 0; } else { total = (v >> 2) + 1; t1 = v & 3; }
    } else {
      const uint32_t bits = br.peek (16);
      const TokLut& L = tok_lut();
      const uint16_t e = L.t[tab][bits >> 6];           // codes of up to 10 bits resolve in one look-up
      if (e) { total = e >> 8; t1 = (e >> 4) & 3; br.skip (e & 15); }
      else for (int i = 0; i < kCoeffTokenCount[tab]; i++) {
        const VlcTok& t = kCoeffToken[tab][i];
        if ((bits >> (16 - t.len)) == t.code) { total = t.total_coeff; t1 = t.trailing_ones; br.skip (t.len); break; }
      }
    }
    for (int i = 0; i < 16; i++) level[i] = 0;
    if (total < 0 || br.err) return -1;
    if (total == 0) return 0;
    if (total > max_coeff) return -1;
    int lv[16];
    int suffix_len = (total > 10 && t1 < 3) ? 1 : 0;
    for (int i = 0; i < total; i++) {
      if (i < t1) { lv[i] = br.u1() ? -1 : 1; continue; }
      int prefix = 0;
      {
        const uint32_t w = br.peek (32);
        if (w) { prefix = __builtin_clz (w); br.skip (prefix + 1); if (br.err) return -1; }
        else { while (!br.u1()) { if (br.err || ++prefix > 32) return -1; } }
      }
      int code = std::min (15, prefix) << suffix_len;
      int ssize = (prefix == 14 && suffix_len == 0) ? 4 : (prefix >= 15 ? prefix - 3 : suffix_len);
      if (ssize > 0) code += (int)br.u (ssize);
      if (prefix >= 15 && suffix_len == 0) code += 15;
      if (prefix >= 16) code += (1 << (prefix - 3)) - 4096;
      if (i == t1 && t1 < 3) code += 2;
      lv[i] = (code & 1) ? (-code - 1) >> 1 : (code + 2) >> 1;
      if (suffix_len == 0) suffix_len = 1;
      if (std::abs (lv[i]) > (3 << (suffix_len - 1)) && suffix_len < 6) suffix_len++;
    }
    int zeros_left = 0;
    if (total < max_coeff) {
      const uint32_t bits = br.peek (9);
      bool ok = false;
      if (nC < 0) {
        for (int i = 0; i < kTotalZerosChromaDcCount[total]; i++) {
          const VlcSym& s = kTotalZerosChromaDc[total][i];
          if ((bits >> (9 - s.len)) == s.code) { zeros_left = s.sym; br.skip (s.len); ok = true; break; }
        }
      } else {
        for (int i = 0; i < kTotalZerosCount[total]; i++) {
          const VlcSym& s = kTotalZeros[total][i];
          if ((bits >> (9 - s.len)) == s.code) { zeros_left = s.sym; br.skip (s.len); ok = true; break; }
        }
      }
      if (!ok) return -1;
    }
    if (zeros_left + total > max_coeff) return -1;
    int pos = zeros_left + total - 1;      // scan position of the highest-frequency coefficient
    for (int i = 0; i < total; i++) {
      int run = 0;
      if (i < total - 1 && zeros_left > 0) {
        const int zl = std::min (zeros_left, 7);
        const uint32_t bits = br.peek (11);
        bool ok = false;
        for (int k = 0; k < kRunBeforeCount[zl]; k++) {
          const VlcSym& s = kRunBefore[zl][k];
          if ((bits >> (11 - s.len)) == s.code) { run = s.sym; br.skip (s.len); ok = true; break; }
        }
        if (!ok || run > zeros_left) return -1;
      } else if (i == total - 1) run = zeros_left;
      level[pos] = lv[i];
      pos -= run + 1;
      zeros_left -= run;
    }
    return br.err ? -1 : total;
  }

  // ---- slice data ---------------------------------------------------------------------------------------------------
  struct SliceCtx {
    const Sps* S; const Pps* P; const SliceHeader* sh; int sid;
    std::vector<int> ref_frames;      // ref_idx -> frame id
  };

  inline bool mb_avail (int k, int sid) const { return k >= 0 && st[k].slice == sid; }
  inline bool intra_nb_avail (int k, int sid, bool constrained) const {
    return mb_avail (k, sid) && (!constrained || st[k].type_class == 1 || st[k].type_class == 2);
  }

  // total_coeff of the 4x4 block at luma raster position (bx,by may be -1 / 4 meaning the neighbouring MB)
  int nz_luma (int k, int bx, int by, int sid, bool& avail) const {
    const int w = cur->mb_w;
    int kk = k;
    if (bx < 0) { kk = (k % w) ? k - 1 : -1; bx = 3; }
    if (by < 0) { kk = k >= w ? k - w : -1; by = 3; }
    avail = kk == k || mb_avail (kk, sid);
    if (!avail) return 0;
    return cur->mbs[kk].nzc[by * 4 + bx];
  }
  int nz_chroma (int k, int c, int bx, int by, int sid, bool& avail) const {
    const int w = cur->mb_w;
    int kk = k;
    if (bx < 0) { kk = (k % w) ? k - 1 : -1; bx = 1; }
    if (by < 0) { kk = k >= w ? k - w : -1; by = 1; }
    avail = kk == k || mb_avail (kk, sid);
    if (!avail) return 0;
    return cur->mbs[kk].nzc[kChromaNzcIdx[c][by * 2 + bx]];
  }
  static int nC_of (int nA, bool aA, int nB, bool aB) {
    if (aA && aB) return (nA + nB + 1) >> 1;
    if (aA) return nA;
    if (aB) return nB;
    return 0;
  }

  // dequantisation exactly as the reference's parser does it (parse_mb_syn_cavlc.cpp:945-993, :1104-1106)
  struct Dq { const Pps* P; };
  inline int dq4 (const Pps& P, bool use_sl, int list, int qp, int j, int level) const {
    const int x = j & 3, y = j >> 2;
    const int cls = ((x & 1) == 0 && (y & 1) == 0) ? 0 : ((x & 1) && (y & 1)) ? 1 : 2;
    const int d = kNormAdjust4x4[qp % 6][cls] << (qp / 6);
    return use_sl ? (level * (P.sl4[list][j] * d)) >> 4 : level * d;
  }
  static inline int cls8 (int x, int y) {
    if ((x & 3) == 0 && (y & 3) == 0) return 0;
    if ((x & 1) && (y & 1)) return 1;
    if ((x & 3) == 2 && (y & 3) == 2) return 2;
    if (((x & 3) == 0 && (y & 1)) || ((x & 1) && (y & 3) == 0)) return 3;
    if (((x & 3) == 0 && (y & 3) == 2) || ((x & 3) == 2 && (y & 3) == 0)) return 4;
    return 5;
  }
  inline int dq8 (const Pps& P, bool use_sl, int list8, int qp, int j, int level) const {
    const int d = (use_sl ? P.sl8[list8][j] : 16) * kNormAdjust8x8[qp % 6][cls8 (j & 7, j >> 3)];
    return qp >= 36 ? (level * d) * (1 << (qp / 6 - 6)) : (level * d + (1 << (5 - qp / 6))) >> (6 - qp / 6);
  }

  // median motion vector prediction (8.4.1.3) on the 4x4-granular state
  struct Nb { bool avail; int ref; int mvx, mvy; };
  Nb nb_block (int k, int bx, int by, int sid, uint32_t filled) const {    // bx,by in -1..4 relative to MB k
    const int w = cur->mb_w;
    int kk = k, x = bx, y = by;
    if (x < 0) { kk = (kk % w) ? kk - 1 : -1; x += 4; } else if (x > 3) { kk = ((kk % w) + 1 < w) ? kk + 1 : -1; x -= 4; }
    if (kk >= 0 && y < 0) { kk = kk >= w ? kk - w : -1; y += 4; }
    Nb n = {false, -1, 0, 0};
    if (kk < 0) return n;
    if (kk == k) {
      if (bx < 0 || bx > 3 || by < 0) return n;
      if (!((filled >> (y * 4 + x)) & 1)) return n;
    } else {
      if (!mb_avail (kk, sid)) return n;
      if (kk > k) return n;
    }
    n.avail = true;
    const MbState& s = kk == k ? st[k] : st[kk];
    n.ref = s.ref[(y >> 1) * 2 + (x >> 1)];
    n.mvx = s.mv[y * 4 + x][0]; n.mvy = s.mv[y * 4 + x][1];
    return n;
  }
  static int median3 (int a, int b, int c) { return std::max (std::min (a, b), std::min (std::max (a, b), c)); }
  void predict_mv (int k, int sid, uint32_t filled, int bx, int by, int bw /*in 4x4 units*/, int ref, int shape /*0 generic,1 16x8 top,2 16x8 bottom,3 8x16 left,4 8x16 right*/,
                   int& px, int& py) const {
    Nb A = nb_block (k, bx - 1, by, sid, filled), B = nb_block (k, bx, by - 1, sid, filled), C = nb_block (k, bx + bw, by - 1, sid, filled);
    if (!C.avail) C = nb_block (k, bx - 1, by - 1, sid, filled);
    if (shape == 1 && B.avail && B.ref == ref) { px = B.mvx; py = B.mvy; return; }
    if (shape == 2 && A.avail && A.ref == ref) { px = A.mvx; py = A.mvy; return; }
    if (shape == 3 && A.avail && A.ref == ref) { px = A.mvx; py = A.mvy; return; }
    if (shape == 4 && C.avail && C.ref == ref) { px = C.mvx; py = C.mvy; return; }
    if (!B.avail && !C.avail && A.avail) { px = A.mvx; py = A.mvy; return; }
    const int m = (A.avail && A.ref == ref) + (B.avail && B.ref == ref) + (C.avail && C.ref == ref);
    if (m == 1) {
      if (A.avail && A.ref == ref) { px = A.mvx; py = A.mvy; }
      else if (B.avail && B.ref == ref) { px = B.mvx; py = B.mvy; }
      else { px = C.mvx; py = C.mvy; }
      return;
    }
    px = median3 (A.avail ? A.mvx : 0, B.avail ? B.mvx : 0, C.avail ? C.mvx : 0);
    py = median3 (A.avail ? A.mvy : 0, B.avail ? B.mvy : 0, C.avail ? C.mvy : 0);
  }

  bool parse_slice_data_cavlc (BitReader& br, SliceCtx& c);
  bool parse_slice_data_cabac (BitReader& br, SliceCtx& c);
  bool parse_mb_cabac (Cabac& cb, SliceCtx& c, int k, int& qp_prev, int& last_dqp, bool is_skip);
  int cabac_residual (Cabac& cb, int k, int sid, int cat, int blk /*luma raster 4x4, chroma block 0..3, or plane for DC*/, int plane, bool cur_intra, int* lv, int maxc);
  bool parse_mb_cavlc (BitReader& br, SliceCtx& c, int k, int& qp_prev, bool is_skip);
  void finalize_intra_modes (SliceCtx& c, int k, const int* raw4 /*16 raster or null*/, bool is8, int i16mode, int chroma_mode);

  int handle_nal (const uint8_t* nal, size_t len);
};

// map parsed (standard-numbered) intra modes to the reference's availability-resolved "final" modes
// (CheckIntraNxNPredMode / CheckIntra16x16PredMode / CheckIntraChromaPredMode, parse_mb_syn_cavlc.cpp:519-611)
void Parser::Impl::finalize_intra_modes (SliceCtx& c, int k, const int* raw, bool is8, int i16mode, int chroma_mode) {
  const int w = cur->mb_w, sid = c.sid;
  const bool cip = c.P->constrained_intra_pred;
  const bool L = (k % w) && intra_nb_avail (k - 1, sid, cip);
  const bool T = k >= w && intra_nb_avail (k - w, sid, cip);
  const bool TL = (k % w) && k >= w && intra_nb_avail (k - w - 1, sid, cip);
  const bool TR = k >= w && ((k % w) + 1 < w) && intra_nb_avail (k - w + 1, sid, cip);
  lh264_mb_t& m = cur->mbs[k];
  m.intra_avail = (uint8_t) ((T ? LH264_AVAIL_T : 0) | (TL ? LH264_AVAIL_TL : 0) | (L ? LH264_AVAIL_L : 0) | (TR ? LH264_AVAIL_TR : 0));
  auto dcmap = [] (bool l, bool t, int dc, int dcl, int dct, int dc128) { return l && t ? dc : l ? dcl : t ? dct : dc128; };
  if (raw) {
    if (!is8) {
      for (int z = 0; z < 16; z++) {
        const int bx = z2x (z), by = z2y (z);
        const bool l = bx > 0 || L, t = by > 0 || T;
        bool tr;
        if (by == 0) tr = bx < 3 ? T : TR;
        else tr = bx < 3 && zidx_before (bx + 1, by - 1, z);
        int mode = raw[by * 4 + bx];
        if (mode == 2) mode = dcmap (l, t, LH264_I4_DC, LH264_I4_DC_L, LH264_I4_DC_T, LH264_I4_DC_128);
        else if (mode == 3 && !tr) mode = LH264_I4_DDL_TOP;
        else if (mode == 7 && !tr) mode = LH264_I4_VL_TOP;
        m.intra_mode[by * 4 + bx] = (int8_t)mode;
      }
    } else {
      for (int i8 = 0; i8 < 4; i8++) {
        const int bx = i8 & 1, by = i8 >> 1;
        const bool l = bx > 0 || L, t = by > 0 || T;
        const bool tr = i8 == 0 ? T : i8 == 1 ? TR : i8 == 2;
        int mode = raw[by * 8 + bx * 2];
        if (mode == 2) mode = dcmap (l, t, LH264_I4_DC, LH264_I4_DC_L, LH264_I4_DC_T, LH264_I4_DC_128);
        else if (mode == 3 && !tr) mode = LH264_I4_DDL_TOP;
        else if (mode == 7 && !tr) mode = LH264_I4_VL_TOP;
        for (int j = 0; j < 4; j++) m.intra_mode[(by * 2 + (j >> 1)) * 4 + bx * 2 + (j & 1)] = (int8_t)mode;
      }
    }
  } else if (i16mode >= 0) {
    int mode = i16mode;
    if (mode == 2) mode = dcmap (L, T, LH264_I16_DC, LH264_I16_DC_L, LH264_I16_DC_T, LH264_I16_DC_128);
    m.intra_mode[0] = (int8_t)mode;
  }
  if (chroma_mode >= 0) {
    int mode = chroma_mode;
    if (mode == 0) mode = dcmap (L, T, LH264_C_DC, LH264_C_DC_L, LH264_C_DC_T, LH264_C_DC_128);
    m.chroma_mode = (int8_t)mode;
  }
}

// ---- one macroblock (7.3.5) ------------------------------------------------------------------------------------------
bool Parser::Impl::parse_mb_cavlc (BitReader& br, SliceCtx& c, int k, int& qp_prev, bool is_skip) {
  const Sps& S = *c.S; const Pps& P = *c.P; const SliceHeader& sh = *c.sh;
  const int sid = c.sid, w = cur->mb_w;
  if (k < 0 || k >= cur->mb_w * cur->mb_h) { fail ("macroblock address out of range"); return false; }
  lh264_mb_t& m = cur->mbs[k];
  MbState& s = st[k];
  memset (&m, 0, sizeof (m));
  m.slice_id = (uint16_t)sid;
  s.slice = (int16_t)sid;
  cur->covered[k] = 1;
  for (int i = 0; i < 16; i++) { s.ipm[i] = 2; s.mv[i][0] = s.mv[i][1] = 0; }
  for (int i = 0; i < 4; i++) { s.ref[i] = -1; m.ref_idx[i] = -1; }
  int16_t* coef = self->want_coeffs_ ? &cur->coeffs[(size_t)k * 384] : no_coef;
  int16_t* lev = self->sparse_levels_ ? lev_scratch : &cur->levels[(size_t)k * 384];
  if (self->lazy_levels_ && !self->sparse_levels_ && !is_skip) memset (lev, 0, 768);     // (the sparse scratch is left zero by finish_levels)
  const bool use_sl = S.scaling_matrix_present || P.scaling_matrix_present;
  auto set_qp = [&] (int qp) {
    m.qp_y = (uint8_t)qp;
    for (int p = 0; p < 2; p++) m.qp_c[p] = kChromaQp[std::min (51, std::max (0, qp + P.chroma_qp_offset[p]))];
  };
  auto fill_part = [&] (int bx, int by, int bw, int bh, int ref, int mvx, int mvy, uint32_t& filled) {
    for (int y = by; y < by + bh; y++) for (int x = bx; x < bx + bw; x++) {
        s.mv[y * 4 + x][0] = (int16_t)mvx; s.mv[y * 4 + x][1] = (int16_t)mvy;
        m.mv[y * 4 + x][0] = (int16_t)mvx; m.mv[y * 4 + x][1] = (int16_t)mvy;
        filled |= 1u << (y * 4 + x);
      }
    (void)ref;
  };
  MbSyn& y = cur->syn[k];
  memset (&y, 0, sizeof (y));
  if (is_skip) {                                        // P_Skip: inferred motion (8.4.1.1)
    m.mb_type = LH264_MB_SKIP; s.type_class = 3;
    for (int i = 0; i < 4; i++) { s.ref[i] = 0; m.ref_idx[i] = 0; }
    Nb A = nb_block (k, -1, 0, sid, 0), B = nb_block (k, 0, -1, sid, 0);
    int px = 0, py = 0;
    if (A.avail && B.avail && !(A.ref == 0 && A.mvx == 0 && A.mvy == 0) && !(B.ref == 0 && B.mvx == 0 && B.mvy == 0))
      predict_mv (k, sid, 0, 0, 0, 4, 0, 0, px, py);
    uint32_t filled = 0;
    fill_part (0, 0, 4, 4, 0, px, py, filled);
    set_qp (qp_prev);
    return true;
  }
  y.have = 1; y.slice_type = (uint8_t)sh.slice_type; y.num_ref_idx_l0 = (uint32_t)sh.num_ref_idx_l0;
  y.skip_run = slice_run_before; y.last_mb_qp = qp_prev;
  // finish the record whichever way the macroblock ends
  struct SynDone {
    Impl* d; MbSyn& y; lh264_mb_t& m; int k;
    ~SynDone() {
      y.mb_type = m.mb_type; y.t8 = (m.flags & LH264_MBF_T8x8) ? 1 : 0; y.cbp_c = m.cbp >> 4; y.cbp_l = m.cbp & 15; y.luma_qp = m.qp_y;
      if (m.mb_type == LH264_MB_I4x4 || m.mb_type == LH264_MB_I8x8 || m.mb_type == LH264_MB_I16x16) d->persist_chroma[k] = (uint8_t)m.chroma_mode;
      if (m.mb_type == LH264_MB_I16x16) d->persist_l16[k] = (uint8_t)m.intra_mode[0];
      if (m.mb_type == LH264_MB_P8x8 || m.mb_type == LH264_MB_P8x8REF0) memcpy (&d->persist_sub[(size_t)k * 4], m.sub_type, 4);
      y.chroma_mode = d->persist_chroma[k]; y.luma16_mode = d->persist_l16[k];
      memcpy (y.sub_type, &d->persist_sub[(size_t)k * 4], 4);
      y.delta_qp = (int)m.qp_y - d->slice_cached_qp;
      d->slice_cached_qp = m.qp_y;
      // (a destructor: an allocation failure in here must not escape - it would end the process - but fail the stream)
      try { if (m.mb_type != LH264_MB_IPCM) d->finish_levels (k); } catch (const std::exception& e) { d->fail (std::string ("internal: ") + e.what()); }
    }
  } syn_done = {this, y, m, k};
  uint32_t mbt = br.ue();
  bool intra = true;
  if (sh.slice_type == 0) { if (mbt < 5) intra = false; else mbt -= 5; }
  if (br.err) return false;
  int cbp = 0;
  bool t8 = false;
  int raw_modes[16]; bool have_raw = false; int i16mode = -1, chroma_mode = -1;
  if (intra) {
    if (mbt > 25) { fail ("invalid mb_type"); return false; }
    if (mbt == 25) {                                    // I_PCM (7.3.5: pcm alignment + 384 samples)
      m.mb_type = LH264_MB_IPCM; s.type_class = 2;
      while (!br.byte_aligned()) br.u1();
      for (int i = 0; i < 384; i++) { coef[i] = (int16_t)br.u (8); self->pcm_.push_back ((uint8_t)coef[i]); }
      m.flags |= LH264_MBF_PCM_IN_COEFF;
      memset (m.nzc, 16, 24);
      m.qp_y = 0; m.qp_c[0] = m.qp_c[1] = 0;          // the reference stores QP 0 for I_PCM (decode_slice.cpp:3257-3258)
      finalize_intra_modes (c, k, nullptr, false, -1, -1);
      return !br.err;
    }
    if (mbt == 0) {                                     // I_NxN
      if (P.transform_8x8) t8 = br.u1();
      m.mb_type = t8 ? LH264_MB_I8x8 : LH264_MB_I4x4; s.type_class = 1;
      const int nblk = t8 ? 4 : 16;
      for (int i = 0; i < nblk; i++) {
        const int bx = t8 ? (i & 1) * 2 : z2x (i), by = t8 ? (i >> 1) * 2 : z2y (i);
        // neighbouring modes (8.3.1.1 / 8.3.2.1)
        int modeA = 2, modeB = 2; bool dcpred = false;
        {
          int kk = k, x = bx - 1, y = by;
          if (x < 0) { kk = (k % w) ? k - 1 : -1; x = 3; }
          if (kk != k && !intra_nb_avail (kk, sid, P.constrained_intra_pred)) dcpred = true;
          else modeA = (kk == k || st[kk].type_class == 1) ? st[kk].ipm[y * 4 + x] : 2;
        }
        {
          int kk = k, x = bx, y = by - 1;
          if (y < 0) { kk = k >= w ? k - w : -1; y = 3; }
          if (kk != k && !intra_nb_avail (kk, sid, P.constrained_intra_pred)) dcpred = true;
          else modeB = (kk == k || st[kk].type_class == 1) ? st[kk].ipm[y * 4 + x] : 2;
        }
        const int pred = dcpred ? 2 : std::min (modeA, modeB);
        int mode = pred;
        if (!br.u1()) { const int rem = (int)br.u (3); mode = rem < pred ? rem : rem + 1; }
        const int n = t8 ? 2 : 1;
        y.pred_mode[i] = (int8_t)mode;
        for (int yy = 0; yy < n; yy++) for (int x = 0; x < n; x++) { s.ipm[(by + yy) * 4 + bx + x] = (int8_t)mode; raw_modes[(by + yy) * 4 + bx + x] = mode; }
      }
      have_raw = true;
      chroma_mode = (int)br.ue();
      const uint32_t ci = br.ue();
      if (ci > 47 || chroma_mode < 0 || chroma_mode > 3) { fail ("invalid cbp / chroma mode"); return false; }
      cbp = kCbpIntra[ci];
    } else {                                            // Intra16x16
      m.mb_type = LH264_MB_I16x16; s.type_class = 2;
      i16mode = (mbt - 1) & 3;
      cbp = (((mbt - 1) >> 2) % 3) << 4 | ((mbt - 1) >= 12 ? 15 : 0);
      chroma_mode = (int)br.ue();
      if (chroma_mode < 0 || chroma_mode > 3) { fail ("invalid chroma mode"); return false; }
    }
  } else {                                              // P macroblocks
    s.type_class = 3;
    static const uint16_t kType[5] = {LH264_MB_P16x16, LH264_MB_P16x8, LH264_MB_P8x16, LH264_MB_P8x8, LH264_MB_P8x8REF0};
    m.mb_type = kType[mbt];
    const int nref = sh.num_ref_idx_l0;
    auto read_ref = [&] () -> int { if (nref <= 1) return 0; if (nref == 2) return br.u1() ? 0 : 1; return (int)br.ue(); };
    uint32_t filled = 0;
    if (mbt <= 2) {
      const int np = mbt == 0 ? 1 : 2;
      int ref[2];
      for (int i = 0; i < np; i++) { ref[i] = read_ref(); if (ref[i] < 0 || ref[i] >= nref) { fail ("ref_idx out of range"); return false; } }
      for (int i = 0; i < np; i++) {
        int bx = 0, by = 0, bw = 4, bh = 4, shape = 0;
        if (mbt == 1) { bh = 2; by = i * 2; shape = 1 + i; } else if (mbt == 2) { bw = 2; bx = i * 2; shape = 3 + i; }
        for (int q = 0; q < 4; q++) if ((q >> 1) * 2 >= by && (q >> 1) * 2 < by + bh && (q & 1) * 2 >= bx && (q & 1) * 2 < bx + bw) { s.ref[q] = (int8_t)ref[i]; m.ref_idx[q] = (int8_t)ref[i]; }
        int px, py;
        predict_mv (k, sid, filled, bx, by, bw, ref[i], shape, px, py);
        const int dx = br.se(), dy = br.se();
        y.mvd[by * 4 + bx][0] = (int16_t)dx; y.mvd[by * 4 + bx][1] = (int16_t)dy; y.ref_idx[i] = (int8_t)ref[i];
        const int mvx = px + dx, mvy = py + dy;
        fill_part (bx, by, bw, bh, ref[i], mvx, mvy, filled);
      }
    } else {
      int sub[4], ref[4] = {0, 0, 0, 0};
      for (int q = 0; q < 4; q++) { sub[q] = (int)br.ue(); if (sub[q] > 3) { fail ("invalid sub_mb_type"); return false; } m.sub_type[q] = (uint8_t) (1 << sub[q]); }
      if (mbt == 3) for (int q = 0; q < 4; q++) { ref[q] = read_ref(); if (ref[q] < 0 || ref[q] >= nref) { fail ("ref_idx out of range"); return false; } }
      for (int q = 0; q < 4; q++) { s.ref[q] = (int8_t)ref[q]; m.ref_idx[q] = (int8_t)ref[q]; y.ref_idx[q] = (int8_t)ref[q]; }
      for (int q = 0; q < 4; q++) {
        const int qx = (q & 1) * 2, qy = (q >> 1) * 2;
        const int nsp = sub[q] == 0 ? 1 : sub[q] == 3 ? 4 : 2;
        for (int j = 0; j < nsp; j++) {
          int bx = qx, by = qy, bw = 2, bh = 2;
          if (sub[q] == 1) { bh = 1; by += j; } else if (sub[q] == 2) { bw = 1; bx += j; } else if (sub[q] == 3) { bw = bh = 1; bx += j & 1; by += j >> 1; }
          int px, py;
          predict_mv (k, sid, filled, bx, by, bw, ref[q], 0, px, py);
          const int dx = br.se(), dy = br.se();
          y.mvd[by * 4 + bx][0] = (int16_t)dx; y.mvd[by * 4 + bx][1] = (int16_t)dy;
          const int mvx = px + dx, mvy = py + dy;
          fill_part (bx, by, bw, bh, ref[q], mvx, mvy, filled);
        }
      }
    }
    const uint32_t ci = br.ue();
    if (ci > 47) { fail ("invalid cbp"); return false; }
    cbp = kCbpInter[ci];
    bool no_sub_lt8 = true;
    if (mbt >= 3) for (int q = 0; q < 4; q++) if (m.sub_type[q] != LH264_SUB_8x8) no_sub_lt8 = false;
    if ((cbp & 15) && P.transform_8x8 && no_sub_lt8) t8 = br.u1();
  }
  if (br.err) return false;
  m.cbp = (uint8_t)cbp;
  if (t8) m.flags |= LH264_MBF_T8x8;
  if (intra) finalize_intra_modes (c, k, have_raw ? raw_modes : nullptr, t8, i16mode, chroma_mode);
  int qp = qp_prev;
  const bool i16 = m.mb_type == LH264_MB_I16x16;
  if (cbp || i16) {
    const int dqp = br.se();
    if (dqp < -26 || dqp > 25) { fail ("mb_qp_delta out of range"); return false; }
    qp = ((qp_prev + dqp) % 52 + 52) % 52;
  }
  set_qp (qp);
  qp_prev = qp;
  if (!(cbp || i16)) return !br.err;

  // ---- residual (7.3.5.3) -------------------------------------------------------------------------------------------
  int lv[16];
  const int ylist = intra ? 0 : 3;
  auto luma_nC = [&] (int bx, int by) { bool aA, aB; const int nA = nz_luma (k, bx - 1, by, sid, aA), nB = nz_luma (k, bx, by - 1, sid, aB); return nC_of (nA, aA, nB, aB); };
  if (i16) {
    if (residual_block (br, luma_nC (0, 0), 16, lv) < 0) { fail ("CAVLC error (I16 DC)"); return false; }
    for (int i = 0; i < 16; i++) if (lv[i]) {
        const int r = kZigzag4x4[i], zb = xy2z (r & 3, r >> 2);
        coef[zb * 16] = lev[zb * 16] = (int16_t)lv[i]; lev_mask |= 1u << zb;           // dequantised by the DC transform on the device
      }
  }
  for (int i8 = 0; i8 < 4; i8++) {
    if (!((cbp >> i8) & 1)) continue;
    for (int j = 0; j < 4; j++) {
      const int z = i8 * 4 + j, bx = z2x (z), by = z2y (z);
      const int maxc = i16 ? 15 : 16;
      const int tot = residual_block (br, luma_nC (bx, by), maxc, lv);
      if (tot < 0) { fail ("CAVLC error (luma)"); return false; }
      m.nzc[by * 4 + bx] = (uint8_t)tot;
      for (int i = 0; i < maxc; i++) if (lv[i]) {
          if (t8) {
            const int pos = kZigzag8x8[4 * i + j];
            { const int li_ = i8 * 64 + pos; lev[li_] = (int16_t)lv[i]; lev_mask |= 1u << (li_ >> 4); }
            coef[i8 * 64 + pos] = (int16_t)dq8 (P, use_sl, intra ? 0 : 1, qp, pos, lv[i]);
          } else {
            const int pos = kZigzag4x4[i16 ? i + 1 : i];
            { const int li_ = z * 16 + pos; lev[li_] = (int16_t)lv[i]; lev_mask |= 1u << (li_ >> 4); }
            coef[z * 16 + pos] = (int16_t)dq4 (P, use_sl, ylist, qp, pos, lv[i]);
          }
        }
    }
  }
  const int cbpc = cbp >> 4;
  if (cbpc) {
    for (int p = 0; p < 2; p++) {                       // chroma DC, nC = -1
      if (residual_block (br, -1, 4, lv) < 0) { fail ("CAVLC error (chroma DC)"); return false; }
      const int qc = m.qp_c[p];
      const int d0 = kNormAdjust4x4[qc % 6][0] << (qc / 6);
      for (int i = 0; i < 4; i++) if (lv[i]) {
          { const int li_ = 256 + p * 64 + i * 16; lev[li_] = (int16_t)lv[i]; lev_mask |= 1u << (li_ >> 4); }
          coef[256 + p * 64 + i * 16] = (int16_t) (use_sl ? (lv[i] * (P.sl4[ylist + 1 + p][0] * d0)) >> 4 : lv[i] * d0);
        }
    }
    if (cbpc == 2) {
      for (int p = 0; p < 2; p++) for (int j = 0; j < 4; j++) {
          const int bx = j & 1, by = j >> 1;
          bool aA, aB;
          const int nA = nz_chroma (k, p, bx - 1, by, sid, aA), nB = nz_chroma (k, p, bx, by - 1, sid, aB);
          const int tot = residual_block (br, nC_of (nA, aA, nB, aB), 15, lv);
          if (tot < 0) { fail ("CAVLC error (chroma AC)"); return false; }
          m.nzc[kChromaNzcIdx[p][j]] = (uint8_t)tot;
          for (int i = 0; i < 15; i++) if (lv[i]) {
              const int pos = kZigzag4x4[i + 1];
              { const int li_ = 256 + p * 64 + j * 16 + pos; lev[li_] = (int16_t)lv[i]; lev_mask |= 1u << (li_ >> 4); }
              coef[256 + p * 64 + j * 16 + pos] = (int16_t)dq4 (P, use_sl, ylist + 1 + p, m.qp_c[p], pos, lv[i]);
            }
        }
    }
  }
  return !br.err;
}

bool Parser::Impl::parse_slice_data_cavlc (BitReader& br, SliceCtx& c) {
  const int n = cur->mb_w * cur->mb_h;
  int k = c.sh->first_mb, qp_prev = c.sh->slice_qp, count = 0;
  bool more = true;
  slice_cached_qp = 0; slice_run_before = 0;
  while (more && k < n) {
    slice_run_before = 0;
    if (c.sh->slice_type != 2) {
      uint32_t run = br.ue();
      if (br.err || k < 0 || run > (uint32_t) (n - k)) { fail ("invalid mb_skip_run"); return false; }
      for (uint32_t i = 0; i < run; i++, k++, count++) if (!parse_mb_cavlc (br, c, k, qp_prev, true)) return false;
      slice_run_before = (int)run;
      more = br.more_rbsp_data();
      if (!more || k >= n) break;
    }
    if (!parse_mb_cavlc (br, c, k, qp_prev, false)) return false;
    k++; count++;
    more = br.more_rbsp_data();
  }
  cur->slices[c.sid].n_mbs = count;
  {   // what follows the stop bit in its byte
    SliceSyn ss;
    const size_t stop = br.pos;                      // position of rbsp_stop_one_bit
    ss.pad_bits = 7 - (int) (stop & 7);
    ss.pad_value = (ss.pad_bits && (stop >> 3) < rbsp.size()) ? (rbsp[stop >> 3] & ((1 << ss.pad_bits) - 1)) : 0;
    ss.transform8x8_pps = c.P->transform_8x8 ? 1 : 0;
    ss.flags = (c.P->cabac ? 1 : 0) | (c.P->constrained_intra_pred ? 2 : 0);
    if (cur->slice_syn.size() <= (size_t)c.sid) cur->slice_syn.resize ((size_t)c.sid + 1);
    cur->slice_syn[c.sid] = ss;
  }
  return true;
}

// ---- CABAC macroblock layer (7.3.5 with the binarisations and context selection of 9.3.2 / 9.3.3) ------------------------
namespace {
// ctxIdxInc of significant_coeff_flag (frame coded) and last_significant_coeff_flag for 8x8 blocks, Table 9-43
const uint8_t kSig8x8[63] = {0, 1, 2, 3, 4, 5, 5, 4, 4, 3, 3, 4, 4, 4, 5, 5, 4, 4, 4, 4, 3, 3, 6, 7, 7, 7, 8, 9, 10, 9, 8, 7, 7, 6, 11, 12, 13, 11, 6, 7, 8, 9,
                             14, 10, 9, 8, 6, 11, 12, 13, 11, 6, 9, 14, 10, 9, 11, 12, 13, 11, 14, 10, 12};
const uint8_t kLast8x8[63] = {0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 3,
                              4, 4, 4, 4, 4, 4, 4, 4, 5, 5, 5, 5, 6, 6, 6, 6, 7, 7, 7, 7, 8, 8, 8};
const int kCatCbf[5] = {0, 4, 8, 12, 16}, kCatMap[5] = {0, 15, 29, 44, 47}, kCatAbs[5] = {0, 10, 20, 30, 39};
}  // namespace

// one residual block (7.3.5.3.3): returns the number of nonzero coefficients, levels in scan order in lv[0..maxc-1]
int Parser::Impl::cabac_residual (Cabac& cb, int k, int sid, int cat, int blk, int plane, bool cur_intra, int* lv, int maxc) {
  const int w = cur->mb_w;
  MbState& s = st[k];
  for (int i = 0; i < maxc; i++) lv[i] = 0;
  if (cat != 5) {
    // coded_block_flag, ctxIdxInc = condTermFlagA + 2 condTermFlagB (9.3.3.1.1.9)
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
    if (kA == -2) kA = ((k % w) && mb_avail (k - 1, sid)) ? k - 1 : -1;
    if (kB == -2) kB = (k >= w && mb_avail (k - w, sid)) ? k - w : -1;
    const int cA = kA < 0 ? (cur_intra ? 1 : 0) : (int) ((st[kA].cbf >> bitA) & 1);
    const int cBf = kB < 0 ? (cur_intra ? 1 : 0) : (int) ((st[kB].cbf >> bitB) & 1);
    if (!cb.decode (85 + kCatCbf[cat] + cA + 2 * cBf)) return 0;
    s.cbf |= 1u << bit;
  }
  const int sig_base = cat == 5 ? 402 : 105 + kCatMap[cat], last_base = cat == 5 ? 417 : 166 + kCatMap[cat];
  const int abs_base = cat == 5 ? 426 : 227 + kCatAbs[cat];
  uint8_t sig[64];
  int n_sig = 0, i;
  for (i = 0; i < maxc - 1; i++) {
    const int inc_s = cat == 5 ? kSig8x8[i] : cat == 3 ? std::min (i, 2) : i;
    const int inc_l = cat == 5 ? kLast8x8[i] : cat == 3 ? std::min (i, 2) : i;
    if (cb.decode (sig_base + inc_s)) {
      sig[n_sig++] = (uint8_t)i;
      if (cb.decode (last_base + inc_l)) break;
    }
  }
  if (i == maxc - 1) sig[n_sig++] = (uint8_t) (maxc - 1);      // the last position is significant by inference
  int num_eq1 = 0, num_gt1 = 0;
  for (int j = n_sig - 1; j >= 0; j--) {
    int inc = num_gt1 ? 0 : std::min (4, 1 + num_eq1);
    int v = 0;                                                   // coeff_abs_level_minus1: TU prefix (cMax 14) + EG0 suffix
    if (cb.decode (abs_base + inc)) {
      v = 1;
      inc = 5 + std::min (4 - (cat == 3 ? 1 : 0), num_gt1);
      while (v < 14 && cb.decode (abs_base + inc)) v++;
      if (v == 14) {
        int kk = 0;
        while (cb.bypass()) { v += 1 << kk; kk++; if (kk > 24) { cb.err = true; break; } }
        while (kk--) v += cb.bypass() << kk;
      }
    }
    const int mag = v + 1;
    lv[sig[j]] = cb.bypass() ? -mag : mag;
    if (mag == 1) num_eq1++; else num_gt1++;
  }
  return n_sig;
}

bool Parser::Impl::parse_mb_cabac (Cabac& cb, SliceCtx& c, int k, int& qp_prev, int& last_dqp, bool is_skip) {
  const Sps& S = *c.S; const Pps& P = *c.P; const SliceHeader& sh = *c.sh;
  const int sid = c.sid, w = cur->mb_w;
  if (k < 0 || k >= cur->mb_w * cur->mb_h) { fail ("macroblock address out of range"); return false; }
  lh264_mb_t& m = cur->mbs[k];
  MbState& s = st[k];
  memset (&m, 0, sizeof (m));
  m.slice_id = (uint16_t)sid;
  s.slice = (int16_t)sid;
  cur->covered[k] = 1;
  for (int i = 0; i < 16; i++) { s.ipm[i] = 2; s.mv[i][0] = s.mv[i][1] = 0; s.mvd[i][0] = s.mvd[i][1] = 0; }
  for (int i = 0; i < 4; i++) { s.ref[i] = -1; m.ref_idx[i] = -1; }
  s.skip = 0; s.pcm = 0; s.t8 = 0; s.cbp = 0; s.chroma_pred = 0; s.cbf = 0;
  int16_t* coef = self->want_coeffs_ ? &cur->coeffs[(size_t)k * 384] : no_coef;
  int16_t* lev = self->sparse_levels_ ? lev_scratch : &cur->levels[(size_t)k * 384];
  if (self->lazy_levels_ && !self->sparse_levels_ && !is_skip) memset (lev, 0, 768);     // (the sparse scratch is left zero by finish_levels)
  const bool use_sl = S.scaling_matrix_present || P.scaling_matrix_present;
  const int kA = ((k % w) && mb_avail (k - 1, sid)) ? k - 1 : -1, kB = (k >= w && mb_avail (k - w, sid)) ? k - w : -1;
  auto set_qp = [&] (int qp) {
    m.qp_y = (uint8_t)qp;
    for (int p = 0; p < 2; p++) m.qp_c[p] = kChromaQp[std::min (51, std::max (0, qp + P.chroma_qp_offset[p]))];
  };
  auto fill_part = [&] (int bx, int by, int bw, int bh, int mvx, int mvy, int dx, int dy, uint32_t& filled) {
    const uint8_t ax = (uint8_t)std::min (255, std::abs (dx)), ay = (uint8_t)std::min (255, std::abs (dy));
    for (int y = by; y < by + bh; y++) for (int x = bx; x < bx + bw; x++) {
        s.mv[y * 4 + x][0] = (int16_t)mvx; s.mv[y * 4 + x][1] = (int16_t)mvy;
        m.mv[y * 4 + x][0] = (int16_t)mvx; m.mv[y * 4 + x][1] = (int16_t)mvy;
        s.mvd[y * 4 + x][0] = ax; s.mvd[y * 4 + x][1] = ay;
        filled |= 1u << (y * 4 + x);
      }
  };
  MbSyn& y = cur->syn[k];
  memset (&y, 0, sizeof (y));
  if (c.sh->slice_type == 0) persist_chroma[k] = 0;           // WelsDecodeMbCabacPSlice resets pChromaPredMode (decode_slice.cpp:1174)
  if (is_skip) {                                        // P_Skip
    m.mb_type = LH264_MB_SKIP; s.type_class = 3; s.skip = 1;
    for (int i = 0; i < 4; i++) { s.ref[i] = 0; m.ref_idx[i] = 0; }
    Nb A = nb_block (k, -1, 0, sid, 0), B = nb_block (k, 0, -1, sid, 0);
    int px = 0, py = 0;
    if (A.avail && B.avail && !(A.ref == 0 && A.mvx == 0 && A.mvy == 0) && !(B.ref == 0 && B.mvx == 0 && B.mvy == 0))
      predict_mv (k, sid, 0, 0, 0, 4, 0, 0, px, py);
    uint32_t filled = 0;
    fill_part (0, 0, 4, 4, px, py, 0, 0, filled);
    set_qp (qp_prev);
    last_dqp = 0;
    return true;
  }
  y.have = 1; y.slice_type = (uint8_t)sh.slice_type; y.num_ref_idx_l0 = (uint32_t)sh.num_ref_idx_l0;
  y.skip_run = slice_run_before ? 1 : 0; y.last_mb_qp = qp_prev;     // CABAC: the reference's run counter never passes 1 (decode_slice.cpp:1186,2205-2238)
  struct SynDone {
    Impl* d; MbSyn& y; lh264_mb_t& m; int k;
    ~SynDone() {
      y.mb_type = m.mb_type; y.t8 = (m.flags & LH264_MBF_T8x8) ? 1 : 0; y.cbp_c = m.cbp >> 4; y.cbp_l = m.cbp & 15; y.luma_qp = m.qp_y;
      if (m.mb_type == LH264_MB_I4x4 || m.mb_type == LH264_MB_I8x8 || m.mb_type == LH264_MB_I16x16) d->persist_chroma[k] = (uint8_t)m.chroma_mode;
      if (m.mb_type == LH264_MB_I16x16) d->persist_l16[k] = (uint8_t)m.intra_mode[0];
      if (m.mb_type == LH264_MB_P8x8 || m.mb_type == LH264_MB_P8x8REF0) memcpy (&d->persist_sub[(size_t)k * 4], m.sub_type, 4);
      y.chroma_mode = d->persist_chroma[k]; y.luma16_mode = d->persist_l16[k];
      memcpy (y.sub_type, &d->persist_sub[(size_t)k * 4], 4);
      y.delta_qp = (int)m.qp_y - d->slice_cached_qp;
      d->slice_cached_qp = m.qp_y;
      // (a destructor: an allocation failure in here must not escape - it would end the process - but fail the stream)
      try { if (m.mb_type != LH264_MB_IPCM) d->finish_levels (k); } catch (const std::exception& e) { d->fail (std::string ("internal: ") + e.what()); }
    }
  } syn_done = {this, y, m, k};

  // ---- mb_type (9.3.2.5, 9.3.3.1.1.3) --------------------------------------------------------------------------------------
  auto i_type = [&] (bool islice) -> int {           // I-slice binarisation, or the suffix of an intra macroblock in a P slice
    int b0;
    if (islice) {
      const int cA = kA >= 0 && st[kA].type_class != 1, cBn = kB >= 0 && st[kB].type_class != 1;       // neighbour is not I_NxN
      b0 = cb.decode (3 + cA + cBn);
    } else b0 = cb.decode (17);
    if (!b0) return 0;
    if (cb.terminate()) return 25;
    const int base = islice ? 3 : 17;
    const int luma = cb.decode (base + (islice ? 3 : 1));
    const int b3 = cb.decode (base + (islice ? 4 : 2));
    int chroma = 0;
    if (b3) chroma = cb.decode (base + (islice ? 5 : 2)) ? 2 : 1;
    const int p0 = cb.decode (base + (islice ? (b3 ? 6 : 6) : 3)), p1 = cb.decode (base + (islice ? 7 : 3));
    return 1 + (p0 << 1 | p1) + 4 * chroma + 12 * luma;
  };
  uint32_t mbt;
  bool intra = true;
  if (sh.slice_type == 2) mbt = (uint32_t)i_type (true);
  else if (cb.decode (14)) mbt = (uint32_t)i_type (false);
  else {
    intra = false;
    if (!cb.decode (15)) mbt = cb.decode (16) ? 3 : 0;
    else mbt = cb.decode (17) ? 1 : 2;
  }
  if (cb.err) return false;
  int cbp = 0;
  bool t8 = false;
  int raw_modes[16]; bool have_raw = false; int i16mode = -1, chroma_mode = -1;
  auto chroma_pred = [&] () -> int {                  // intra_chroma_pred_mode, 9.3.3.1.1.8
    const int cA = kA >= 0 && st[kA].type_class != 3 && !st[kA].pcm && st[kA].chroma_pred != 0;
    const int cBn = kB >= 0 && st[kB].type_class != 3 && !st[kB].pcm && st[kB].chroma_pred != 0;
    if (!cb.decode (64 + cA + cBn)) return 0;
    if (!cb.decode (67)) return 1;
    return cb.decode (67) ? 3 : 2;
  };
  if (intra) {
    if (mbt == 25) {                                    // I_PCM: the samples follow byte aligned, then the engine restarts (9.3.1.2)
      m.mb_type = LH264_MB_IPCM; s.type_class = 2; s.pcm = 1; s.cbf = 0xffffffffu; s.cbp = 0x2f;
      size_t bp = (cb.pos + 7) & ~ (size_t)7;
      for (int i = 0; i < 384; i++, bp += 8) { coef[i] = (int16_t) (bp + 8 <= cb.nbits ? cb.p[bp >> 3] : 0); self->pcm_.push_back ((uint8_t)coef[i]); }
      if (bp > cb.nbits) cb.err = true;
      cb.start (cb.p, cb.nbits / 8, bp);
      m.flags |= LH264_MBF_PCM_IN_COEFF;
      memset (m.nzc, 16, 24);
      m.qp_y = 0; m.qp_c[0] = m.qp_c[1] = 0;
      finalize_intra_modes (c, k, nullptr, false, -1, -1);
      last_dqp = 0;
      return !cb.err;
    }
    if (mbt == 0) {                                     // I_NxN
      if (P.transform_8x8) {
        const int cA = kA >= 0 && st[kA].t8, cBn = kB >= 0 && st[kB].t8;
        t8 = cb.decode (399 + cA + cBn);
      }
      m.mb_type = t8 ? LH264_MB_I8x8 : LH264_MB_I4x4; s.type_class = 1; s.t8 = t8;
      const int nblk = t8 ? 4 : 16;
      for (int i = 0; i < nblk; i++) {
        const int bx = t8 ? (i & 1) * 2 : z2x (i), by = t8 ? (i >> 1) * 2 : z2y (i);
        int modeA = 2, modeB = 2; bool dcpred = false;
        {
          int kk = k, x = bx - 1, yy = by;
          if (x < 0) { kk = (k % w) ? k - 1 : -1; x = 3; }
          if (kk != k && !intra_nb_avail (kk, sid, P.constrained_intra_pred)) dcpred = true;
          else modeA = (kk == k || st[kk].type_class == 1) ? st[kk].ipm[yy * 4 + x] : 2;
        }
        {
          int kk = k, x = bx, yy = by - 1;
          if (yy < 0) { kk = k >= w ? k - w : -1; yy = 3; }
          if (kk != k && !intra_nb_avail (kk, sid, P.constrained_intra_pred)) dcpred = true;
          else modeB = (kk == k || st[kk].type_class == 1) ? st[kk].ipm[yy * 4 + x] : 2;
        }
        const int pred = dcpred ? 2 : std::min (modeA, modeB);
        int mode = pred;
        if (!cb.decode (68)) { const int rem = cb.decode (69) | cb.decode (69) << 1 | cb.decode (69) << 2; mode = rem < pred ? rem : rem + 1; }
        const int n = t8 ? 2 : 1;
        y.pred_mode[i] = (int8_t)mode;
        for (int yy = 0; yy < n; yy++) for (int x = 0; x < n; x++) { s.ipm[(by + yy) * 4 + bx + x] = (int8_t)mode; raw_modes[(by + yy) * 4 + bx + x] = mode; }
      }
      have_raw = true;
      chroma_mode = chroma_pred();
    } else {                                            // Intra16x16
      m.mb_type = LH264_MB_I16x16; s.type_class = 2;
      i16mode = (mbt - 1) & 3;
      cbp = (((mbt - 1) >> 2) % 3) << 4 | ((mbt - 1) >= 12 ? 15 : 0);
      chroma_mode = chroma_pred();
    }
    s.chroma_pred = (uint8_t)chroma_mode;
  } else {                                              // P macroblocks
    s.type_class = 3;
    static const uint16_t kType[4] = {LH264_MB_P16x16, LH264_MB_P16x8, LH264_MB_P8x16, LH264_MB_P8x8};
    m.mb_type = kType[mbt];
    const int nref = sh.num_ref_idx_l0;
    uint32_t filled = 0;
    // ref_idx_l0 of the partition whose first 4x4 block is (bx,by): ctxIdxInc from the partitions to the left and above
    auto ref_gt0 = [&] (int bx, int by) -> int {
      int kk = k, x = bx, yy = by;
      if (x < 0) { kk = kA; x = 3; } else if (yy < 0) { kk = kB; yy = 3; }
      if (kk < 0) return 0;
      const MbState& t = st[kk];
      if (t.type_class != 3 || t.skip) return 0;
      return t.ref[(yy >> 1) * 2 + (x >> 1)] > 0;
    };
    auto read_ref = [&] (int bx, int by) -> int {
      if (nref <= 1) return 0;
      int inc = ref_gt0 (bx - 1, by) + 2 * ref_gt0 (bx, by - 1), v = 0;
      while (cb.decode (54 + inc)) { v++; inc = v == 1 ? 4 : 5; if (v > 32) { cb.err = true; break; } }
      return v;
    };
    auto abs_mvd = [&] (int bx, int by, int comp) -> int {
      int kk = k, x = bx, yy = by;
      if (x < 0) { kk = kA; x = 3; } else if (yy < 0) { kk = kB; yy = 3; }
      if (kk < 0) return 0;
      return st[kk].mvd[yy * 4 + x][comp];
    };
    auto read_mvd = [&] (int bx, int by, int comp) -> int {        // UEG3, signedValFlag 1, uCoff 9 (9.3.2.3, 9.3.3.1.1.7)
      const int base = comp ? 47 : 40;
      const int sum = abs_mvd (bx - 1, by, comp) + abs_mvd (bx, by - 1, comp);
      int inc = sum < 3 ? 0 : sum > 32 ? 2 : 1;
      if (!cb.decode (base + inc)) return 0;
      int v = 1;
      inc = 3;
      while (v < 9 && cb.decode (base + inc)) { v++; if (inc < 6) inc++; }
      if (v == 9) {
        int kk = 3;
        while (cb.bypass()) { v += 1 << kk; kk++; if (kk > 24) { cb.err = true; break; } }
        while (kk--) v += cb.bypass() << kk;
      }
      return cb.bypass() ? -v : v;
    };
    if (mbt <= 2) {
      const int np = mbt == 0 ? 1 : 2;
      int ref[2];
      for (int i = 0; i < np; i++) {
        const int bx = mbt == 2 ? i * 2 : 0, by = mbt == 1 ? i * 2 : 0;
        ref[i] = read_ref (bx, by);
        if (ref[i] < 0 || ref[i] >= nref) { fail ("ref_idx out of range"); return false; }
        for (int q = 0; q < 4; q++) {
          const bool in = mbt == 0 || (mbt == 1 ? (q >> 1) == i : (q & 1) == i);
          if (in) { s.ref[q] = (int8_t)ref[i]; m.ref_idx[q] = (int8_t)ref[i]; }
        }
      }
      for (int i = 0; i < np; i++) {
        int bx = 0, by = 0, bw = 4, bh = 4, shape = 0;
        if (mbt == 1) { bh = 2; by = i * 2; shape = 1 + i; } else if (mbt == 2) { bw = 2; bx = i * 2; shape = 3 + i; }
        int px, py;
        predict_mv (k, sid, filled, bx, by, bw, ref[i], shape, px, py);
        const int dx = read_mvd (bx, by, 0), dy = read_mvd (bx, by, 1);
        y.mvd[by * 4 + bx][0] = (int16_t)dx; y.mvd[by * 4 + bx][1] = (int16_t)dy; y.ref_idx[i] = (int8_t)ref[i];
        fill_part (bx, by, bw, bh, px + dx, py + dy, dx, dy, filled);
      }
    } else {
      int sub[4], ref[4] = {0, 0, 0, 0};
      for (int q = 0; q < 4; q++) {                       // sub_mb_type, Table 9-37
        if (cb.decode (21)) sub[q] = 0;
        else if (!cb.decode (22)) sub[q] = 1;
        else sub[q] = cb.decode (23) ? 2 : 3;
        m.sub_type[q] = (uint8_t) (1 << sub[q]);
      }
      for (int q = 0; q < 4; q++) {
        ref[q] = read_ref ((q & 1) * 2, (q >> 1) * 2);
        if (ref[q] < 0 || ref[q] >= nref) { fail ("ref_idx out of range"); return false; }
        s.ref[q] = (int8_t)ref[q]; m.ref_idx[q] = (int8_t)ref[q]; y.ref_idx[q] = (int8_t)ref[q];
      }
      for (int q = 0; q < 4; q++) {
        const int qx = (q & 1) * 2, qy = (q >> 1) * 2;
        const int nsp = sub[q] == 0 ? 1 : sub[q] == 3 ? 4 : 2;
        for (int j = 0; j < nsp; j++) {
          int bx = qx, by = qy, bw = 2, bh = 2;
          if (sub[q] == 1) { bh = 1; by += j; } else if (sub[q] == 2) { bw = 1; bx += j; } else if (sub[q] == 3) { bw = bh = 1; bx += j & 1; by += j >> 1; }
          int px, py;
          predict_mv (k, sid, filled, bx, by, bw, ref[q], 0, px, py);
          const int dx = read_mvd (bx, by, 0), dy = read_mvd (bx, by, 1);
          y.mvd[by * 4 + bx][0] = (int16_t)dx; y.mvd[by * 4 + bx][1] = (int16_t)dy;
          fill_part (bx, by, bw, bh, px + dx, py + dy, dx, dy, filled);
        }
      }
    }
  }
  if (cb.err) return false;
  const bool i16 = m.mb_type == LH264_MB_I16x16;
  if (!i16) {                                           // coded_block_pattern, 9.3.2.6 / 9.3.3.1.1.4
    auto luma_bit = [&] (int kk, int b8) -> int {     // condTermFlagN for the 8x8 block b8 of macroblock kk
      if (kk < 0) return 0;
      if (st[kk].pcm) return 0;
      if (st[kk].skip) return 1;
      return ((st[kk].cbp >> b8) & 1) ? 0 : 1;
    };
    int cl = 0;
    for (int b8 = 0; b8 < 4; b8++) {
      const int cA = (b8 & 1) ? (((cl >> (b8 - 1)) & 1) ? 0 : 1) : luma_bit (kA, b8 + 1);
      const int cBn = (b8 & 2) ? (((cl >> (b8 - 2)) & 1) ? 0 : 1) : luma_bit (kB, b8 + 2);
      cl |= cb.decode (73 + cA + 2 * cBn) << b8;
    }
    auto chroma_nz = [&] (int kk, int lvl) -> int {
      if (kk < 0) return 0;
      if (st[kk].pcm) return 1;
      if (st[kk].skip) return 0;
      return (st[kk].cbp >> 4) >= lvl;
    };
    int cc = 0;
    if (cb.decode (77 + chroma_nz (kA, 1) + 2 * chroma_nz (kB, 1))) cc = cb.decode (77 + 4 + chroma_nz (kA, 2) + 2 * chroma_nz (kB, 2)) ? 2 : 1;
    cbp = cl | (cc << 4);
    if (!intra) {
      bool no_sub_lt8 = true;
      if (mbt == 3) for (int q = 0; q < 4; q++) if (m.sub_type[q] != LH264_SUB_8x8) no_sub_lt8 = false;
      if ((cbp & 15) && P.transform_8x8 && no_sub_lt8) {
        const int cA = kA >= 0 && st[kA].t8, cBn = kB >= 0 && st[kB].t8;
        t8 = cb.decode (399 + cA + cBn);
      }
    }
  }
  if (cb.err) return false;
  m.cbp = (uint8_t)cbp; s.cbp = (uint8_t)cbp;
  if (t8) { m.flags |= LH264_MBF_T8x8; s.t8 = 1; }
  if (intra) finalize_intra_modes (c, k, have_raw ? raw_modes : nullptr, t8, i16mode, chroma_mode);
  int qp = qp_prev;
  if (cbp || i16) {                                     // mb_qp_delta, 9.3.2.7 / 9.3.3.1.1.5
    int v = 0;
    if (cb.decode (60 + (last_dqp != 0 ? 1 : 0))) {
      v = 1;
      if (cb.decode (62)) { v = 2; while (cb.decode (63)) { v++; if (v > 120) { cb.err = true; break; } } }
    }
    const int dqp = (v & 1) ? (v + 1) >> 1 : - (v >> 1);
    if (dqp < -26 || dqp > 25) { fail ("mb_qp_delta out of range"); return false; }
    qp = ((qp_prev + dqp) % 52 + 52) % 52;
    last_dqp = dqp;
  } else last_dqp = 0;
  set_qp (qp);
  qp_prev = qp;
  if (!(cbp || i16)) return !cb.err;

  // ---- residual ------------------------------------------------------------------------------------------------------------
  int lv[64];
  const int ylist = intra ? 0 : 3;
  if (i16) {
    cabac_residual (cb, k, sid, 0, 0, 0, true, lv, 16);
    for (int i = 0; i < 16; i++) if (lv[i]) {
        const int r = kZigzag4x4[i], zb = xy2z (r & 3, r >> 2);
        coef[zb * 16] = lev[zb * 16] = (int16_t)lv[i]; lev_mask |= 1u << zb;
      }
  }
  for (int i8 = 0; i8 < 4; i8++) {
    if (!((cbp >> i8) & 1)) continue;
    if (t8) {
      const int tot = cabac_residual (cb, k, sid, 5, i8, 0, intra, lv, 64);
      for (int j = 0; j < 4; j++) { const int z = i8 * 4 + j; m.nzc[z2y (z) * 4 + z2x (z)] = (uint8_t)tot; s.cbf |= 1u << (z2y (z) * 4 + z2x (z)); }
      for (int i = 0; i < 64; i++) if (lv[i]) {
          const int pos = kZigzag8x8[i];
          { const int li_ = i8 * 64 + pos; lev[li_] = (int16_t)lv[i]; lev_mask |= 1u << (li_ >> 4); }
          coef[i8 * 64 + pos] = (int16_t)dq8 (P, use_sl, intra ? 0 : 1, qp, pos, lv[i]);
        }
      continue;
    }
    for (int j = 0; j < 4; j++) {
      const int z = i8 * 4 + j, bx = z2x (z), by = z2y (z);
      const int maxc = i16 ? 15 : 16;
      const int tot = cabac_residual (cb, k, sid, i16 ? 1 : 2, by * 4 + bx, 0, intra, lv, maxc);
      m.nzc[by * 4 + bx] = (uint8_t)tot;
      for (int i = 0; i < maxc; i++) if (lv[i]) {
          const int pos = kZigzag4x4[i16 ? i + 1 : i];
          { const int li_ = z * 16 + pos; lev[li_] = (int16_t)lv[i]; lev_mask |= 1u << (li_ >> 4); }
          coef[z * 16 + pos] = (int16_t)dq4 (P, use_sl, ylist, qp, pos, lv[i]);
        }
    }
  }
  const int cbpc = cbp >> 4;
  if (cbpc) {
    for (int p = 0; p < 2; p++) {
      cabac_residual (cb, k, sid, 3, 0, p, intra, lv, 4);
      const int qc = m.qp_c[p];
      const int d0 = kNormAdjust4x4[qc % 6][0] << (qc / 6);
      for (int i = 0; i < 4; i++) if (lv[i]) {
          { const int li_ = 256 + p * 64 + i * 16; lev[li_] = (int16_t)lv[i]; lev_mask |= 1u << (li_ >> 4); }
          coef[256 + p * 64 + i * 16] = (int16_t) (use_sl ? (lv[i] * (P.sl4[ylist + 1 + p][0] * d0)) >> 4 : lv[i] * d0);
        }
    }
    if (cbpc == 2) {
      for (int p = 0; p < 2; p++) for (int j = 0; j < 4; j++) {
          const int tot = cabac_residual (cb, k, sid, 4, j, p, intra, lv, 15);
          m.nzc[kChromaNzcIdx[p][j]] = (uint8_t)tot;
          for (int i = 0; i < 15; i++) if (lv[i]) {
              const int pos = kZigzag4x4[i + 1];
              { const int li_ = 256 + p * 64 + j * 16 + pos; lev[li_] = (int16_t)lv[i]; lev_mask |= 1u << (li_ >> 4); }
              coef[256 + p * 64 + j * 16 + pos] = (int16_t)dq4 (P, use_sl, ylist + 1 + p, m.qp_c[p], pos, lv[i]);
            }
        }
    }
  }
  return !cb.err;
}

bool Parser::Impl::parse_slice_data_cabac (BitReader& br, SliceCtx& c) {
  const int n = cur->mb_w * cur->mb_h;
  while (!br.byte_aligned()) br.u1();                   // cabac_alignment_one_bit
  Cabac cb;
  cb.init_contexts (c.sh->slice_type == 2 ? 0 : 1 + c.sh->cabac_init_idc, c.sh->slice_qp);
  cb.start (rbsp.data(), rbsp.size(), br.pos);
  int k = c.sh->first_mb, qp_prev = c.sh->slice_qp, count = 0, last_dqp = 0;
  slice_cached_qp = 0; slice_run_before = 0;
  const int w = cur->mb_w, sid = c.sid;
  while (k < n) {
    bool skip = false;
    if (c.sh->slice_type != 2) {                        // mb_skip_flag, 9.3.3.1.1.1
      const int kA = ((k % w) && mb_avail (k - 1, sid)) ? k - 1 : -1, kB = (k >= w && mb_avail (k - w, sid)) ? k - w : -1;
      skip = cb.decode (11 + (kA >= 0 && !st[kA].skip) + (kB >= 0 && !st[kB].skip)) != 0;
    }
    if (!parse_mb_cabac (cb, c, k, qp_prev, last_dqp, skip)) { if (self->err_.empty()) fail ("CABAC error"); return false; }
    slice_run_before = skip ? slice_run_before + 1 : 0;
    k++; count++;
    if (cb.terminate()) break;                          // end_of_slice_flag
    if (cb.err) { fail ("CABAC error (ran out of data)"); return false; }
  }
  cur->slices[c.sid].n_mbs = count;
  {
    // what the reference sends to the pad-bit tag for a CABAC slice (decode_slice.cpp:3133-3148): InitCabacDecEngineFromBS leaves
    // iLeftBits = 0, so always 7 bits, taken from the last byte of its bit buffer = the byte holding the rbsp stop bit
    // (DecInitBits bit_stream.cpp:72-90 with the size ParseNalHeader computes, au_parser.cpp:420-421)
    SliceSyn ss; ss.pad_bits = 7; ss.pad_value = 0; ss.transform8x8_pps = c.P->transform_8x8 ? 1 : 0; ss.flags = 1 | (c.P->constrained_intra_pred ? 2 : 0);
    if (!rbsp.empty()) ss.pad_value = rbsp.back() & 0x7f;
    if (cur->slice_syn.size() <= (size_t)c.sid) cur->slice_syn.resize ((size_t)c.sid + 1);
    cur->slice_syn[c.sid] = ss;
  }
  return true;
}

int Parser::Impl::handle_nal (const uint8_t* nal, size_t len) {
  if (len < 1) return 0;
  const int type = nal[0] & 31, ref_idc = (nal[0] >> 5) & 3;
  if (type == 7 || type == 8 || type == 1 || type == 5) {
    unescape (nal + 1, len - 1, rbsp);
    BitReader br; br.init (rbsp.data(), rbsp.size());
    if (type == 7) { parse_sps (br); return 0; }
    if (type == 8) { parse_pps (br); return 0; }
    SliceHeader sh;
    if (!parse_slice_header (br, type, ref_idc, sh)) return -1;
    if (sh.redundant_pic_cnt > 0) return 0;
    const Pps& P = pps[sh.pps_id]; const Sps& S = sps[P.sps_id];

    const bool new_pic = !cur || sh.frame_num != first_sh.frame_num || sh.idr != first_sh.idr || sh.pps_id != first_sh.pps_id ||
                         (sh.idr && sh.idr_pic_id != first_sh.idr_pic_id) || ((sh.nal_ref_idc == 0) != (first_sh.nal_ref_idc == 0)) ||
                         sh.poc_lsb != first_sh.poc_lsb || sh.delta_poc[0] != first_sh.delta_poc[0] || sh.first_mb <= last_first_mb ||
                         cur->mb_w != S.mb_w || cur->mb_h != S.mb_h;
    if (new_pic) { finish_picture(); start_picture (sh, S, P); }
    last_first_mb = sh.first_mb;
    SliceCtx c; c.S = &S; c.P = &P; c.sh = &sh; c.sid = (int)cur->slices.size();
    lh264_slice_t sl; memset (&sl, 0, sizeof (sl));
    sl.first_mb = sh.first_mb; sl.slice_type = (uint8_t)sh.slice_type; sl.deblock_idc = (uint8_t)sh.deblock_idc;
    sl.alpha_c0_offset = (int8_t)sh.alpha_off; sl.beta_offset = (int8_t)sh.beta_off;
    sl.n_refs = (uint8_t)sh.num_ref_idx_l0;
    sl.luma_dc_weight = (S.scaling_matrix_present || P.scaling_matrix_present) ? P.sl4[0][0] : 16;
    for (int i = 0; i < LH264_MAX_REFS; i++) sl.ref_slot[i] = -1;
    if (sh.slice_type == 0) {
      build_ref_list (sh, S, c.ref_frames);
      for (size_t i = 0; i < c.ref_frames.size() && i < LH264_MAX_REFS; i++) {
        const int fid = c.ref_frames[i];
        if (fid < 0) continue;
        int slot = -1;
        for (size_t q = 0; q < cur->ref_ids.size(); q++) if (cur->ref_ids[q] == fid) slot = (int)q;
        if (slot < 0 && cur->ref_ids.size() < LH264_MAX_REFS) { cur->ref_ids.push_back (fid); slot = (int)cur->ref_ids.size() - 1; }
        sl.ref_slot[i] = (int8_t)slot;
      }
      if (sh.has_weights) {
        sl.weighted_pred = 1; sl.luma_log2_denom = (uint8_t)sh.luma_log2_denom; sl.chroma_log2_denom = (uint8_t)sh.chroma_log2_denom;
        for (int i = 0; i < LH264_MAX_REFS && i < sh.num_ref_idx_l0; i++) {
          sl.luma_weight[i] = (int16_t)sh.luma_weight[i]; sl.luma_offset[i] = (int16_t)sh.luma_offset[i];
          for (int q = 0; q < 2; q++) { sl.chroma_weight[i][q] = (int16_t)sh.chroma_weight[i][q]; sl.chroma_offset[i][q] = (int16_t)sh.chroma_offset[i][q]; }
        }
      }
    }
    cur->slices.push_back (sl);
    if (sh.first_mb < 0 || (uint32_t)sh.first_mb >= (uint32_t) (S.mb_w * S.mb_h)) { fail ("first_mb_in_slice out of range"); return -1; }
    last_hdr_bits = (int)br.pos; last_cabac = P.cabac;
    // (a slice that fails is not modelled: the I_PCM samples it appended so far would shift every later macroblock's in the PCM stream)
    const size_t pcm_mark = self->pcm_.size();
    if (P.cabac ? !parse_slice_data_cabac (br, c) : !parse_slice_data_cavlc (br, c)) { self->pcm_.resize (pcm_mark); return -1; }
    return 0;
  }
  if (type == 10 || type == 11) finish_picture();
  return 0;
}

Parser::Parser() : d_ (new Impl (this)) {}
Parser::~Parser() {}
// the two places every entry point goes through: nothing thrown below them (allocation failures, length errors of the standard
// containers) crosses the C ABI - the stream is reported as failed instead
int Parser::feed_nal (const uint8_t* nal, size_t len) {
  try { return d_->handle_nal (nal, len); }
  catch (const std::exception& e) { d_->fail (std::string ("internal: ") + e.what()); d_->cur.reset(); return -1; }
}
void Parser::flush() {
  try { d_->finish_picture(); }
  catch (const std::exception& e) { d_->fail (std::string ("internal: ") + e.what()); d_->cur.reset(); }
}
bool Parser::picture_in_progress_is_whole() const {
  if (!d_->cur) return false;
  for (uint8_t c : d_->cur->covered) if (!c) return false;
  return true;
}

int Parser::feed (const uint8_t* d, size_t n) {
  // Annex B: NAL units are delimited by 00 00 01 start codes (B.1)
  size_t i = 0, start = (size_t) - 1;
  int rc = 0;
  while (i + 2 < n) {
    if (d[i] == 0 && d[i + 1] == 0 && d[i + 2] == 1) {
      if (start != (size_t) - 1) {
        size_t end = i;
        while (end > start && d[end - 1] == 0) end--;     // trailing_zero_8bits / the leading zero of a 4-byte start code
        if (end > start && feed_nal (d + start, end - start) < 0) rc = -1;
      }
      start = i + 3; i += 3;
    } else i++;
  }
  if (start != (size_t) - 1 && start < n) {
    size_t end = n;
    while (end > start && d[end - 1] == 0) end--;
    if (end > start && feed_nal (d + start, end - start) < 0) rc = -1;
  }
  return rc;
}

int Parser::parse_headers (const uint8_t* nal, size_t len, HeaderInfo& o) {
  if (len < 1) return -1;
  o = HeaderInfo();
  o.nal_type = nal[0] & 31;
  const int ref_idc = (nal[0] >> 5) & 3;
  Impl::unescape (nal + 1, len - 1, d_->rbsp);
  BitReader br; br.init (d_->rbsp.data(), d_->rbsp.size());
  if (o.nal_type == 7) { d_->parse_sps (br); return 0; }
  if (o.nal_type == 8) { d_->parse_pps (br); return 0; }
  if (o.nal_type != 1 && o.nal_type != 5) return 0;
  if (!d_->parse_slice_header (br, o.nal_type, ref_idc, o.sh)) return -1;
  const Pps& P = d_->pps[o.sh.pps_id]; const Sps& S = d_->sps[P.sps_id];
  o.is_slice = true; o.hdr_bits = (int)br.pos; o.mb_w = S.mb_w; o.mb_h = S.mb_h;
  o.cabac = P.cabac; o.transform_8x8 = P.transform_8x8; o.constrained_intra_pred = P.constrained_intra_pred;
  return 0;
}
const std::vector<uint8_t>& Parser::last_rbsp() const { return d_->rbsp; }
void Parser::unescape (const uint8_t* d, size_t n, std::vector<uint8_t>& out) { Impl::unescape (d, n, out); }

// ---- the recompressor's default stream ---------------------------------------------------------------------------
void MainStreamWriter::append_byte (uint8_t x) {
  if (!escaping_) { buffer.push_back (x); return; }
  if (x <= 3 && esc_n_ == 2 && esc_[0] == 0 && esc_[1] == 0) {
    const uint8_t e[4] = {0, 0, 3, x};
    buffer.insert (buffer.end(), e, e + 4);
    esc_n_ = 0;
  } else if (esc_n_ == 2) {
    buffer.push_back (esc_[0]); esc_[0] = esc_[1]; esc_[1] = x;
  } else esc_[esc_n_++] = x;
}
void MainStreamWriter::emit_bit (uint32_t bit) {
  bits_ = (bits_ << 1) | (bit & 1);
  if (++n_bits_ == 8) { const uint8_t b = (uint8_t)bits_; bits_ = 0; n_bits_ = 0; append_byte (b); }
}
void MainStreamWriter::stop_escape() {
  pad_to_byte();
  escaping_ = false;
  for (int i = 0; i < esc_n_; i++) buffer.push_back (esc_[i]);
  esc_n_ = 0;
}

// What the reference leaves in the default stream for each chunk the console application hands to DecodeFrameNoDelay
// (WelsDecodeBs decoder.cpp:658-860 for the chunk, then once more with no data): the bytes up to and including the
// start code; then, escaped: the NAL header byte (au_parser.cpp:143), for SPS/PPS/SEI the RBSP without its trailing
// zero bytes (au_parser.cpp:588), one zero byte per trailing zero byte (decoder.cpp:610-627), for a slice the bits of its
// header (decode_slice.cpp:2974-2980) and, for CAVLC, a single 1 bit (decoder.cpp:837-845); zero bits up to the byte.
int Parser::feed_file (const uint8_t* d, size_t n) {
  struct Scope { StreamArena*& slot; StreamArena* prev; Scope (StreamArena* a) : slot (current_stream_arena()), prev (slot) { slot = a; } ~Scope() { slot = prev; } } scope (arena_.get());
  int rc = 0;
  size_t pos = 0;
  std::vector<uint8_t> nal;
  auto at = [&] (size_t i) -> int { return i < n ? d[i] : (i == n + 3 ? 1 : 0); };   // the application appends 00 00 00 01
  while (pos < n) {
    size_t i;
    for (i = 0; i < n; i++) {
      if (i > 0 && at (pos + i) == 0 && at (pos + i + 1) == 0 && ((at (pos + i + 2) == 0 && at (pos + i + 3) == 1) || at (pos + i + 2) == 1)) break;
    }
    const size_t len = i;
    if (len < 4) { main_.append_bytes (d + pos, std::min (len, n - pos)); pos += len; continue; }
    const uint8_t* c = d + pos;
    pos += len;
    // first start code prefix (DetectStartCodePrefix au_parser.cpp:64-87)
    size_t off = 0; bool found = false;
    for (size_t q = 0, zeros = 0; q < len; q++) {
      if (c[q] == 0) { zeros++; continue; }
      if (c[q] == 1 && zeros >= 2) { off = q + 1; found = true; break; }
      zeros = 0;
    }
    if (!found) continue;
    main_.append_bytes (c, off);
    Impl::unescape (c + off, len - off, nal);
    main_.start_escape();
    size_t tz = 0;
    while (tz < nal.size() && nal[nal.size() - 1 - tz] == 0) tz++;
    bool slice_ok = false;
    if (!nal.empty() && !(nal[0] & 0x80)) {
      const int type = nal[0] & 31;
      main_.append_byte (nal[0]);
      const bool have_sps = !d_->sps.empty(), have_pps = !d_->pps.empty();
      if (type == 6 || type == 7 || (type == 8 && have_sps)) {
        if (nal.size() > 1 + tz) main_.append_bytes (nal.data() + 1, nal.size() - 1 - tz);
      }
      if ((type == 1 || type == 5 || type == 7 || type == 8) && (type == 7 || have_sps) && (type == 7 || type == 8 || have_pps)) {
        size_t end = len;
        while (end > off && c[end - 1] == 0) end--;
        d_->last_hdr_bits = -1;
        if (end > off && feed_nal (c + off, end - off) < 0) rc = -1;
        slice_ok = (type == 1 || type == 5) && d_->last_hdr_bits >= 0;
      }
    }
    for (size_t q = 0; q < tz; q++) main_.append_byte (0);
    if (slice_ok) {
      for (int b = 0; b < d_->last_hdr_bits; b++) main_.emit_bit ((d_->rbsp[(size_t)b >> 3] >> (7 - (b & 7))) & 1);
      if (!d_->last_cabac) main_.emit_bit (1);
    }
    main_.stop_escape();
  }
  flush();
  main_.pad_to_byte();
  return rc;
}

}  // namespace lh264host
