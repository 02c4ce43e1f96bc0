// h264_parser.h - host front end: Annex-B / NAL / SPS / PPS / slice header / CAVLC macroblock parse producing the flat
// macroblock records of include/lh264.h.  Fresh implementation from ITU-T H.264; what it
// must reproduce of the reference is the *content of the records* (the SDqLayer arrays after the reference's parser,
// dec_frame.h:60-97), which tests compare field by field with oracle/_ref/ref_dump fixtures.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <map>
#include <memory>
#include <string>
#include <vector>
#include "../../../include/lh264.h"

namespace lh264host {

struct BitReader {
  const uint8_t* p = nullptr;
  size_t nbits = 0, pos = 0, nbytes = 0;
  bool err = false;
  void init (const uint8_t* d, size_t bytes) { p = d; nbytes = bytes; nbits = bytes * 8; pos = 0; err = false; }
  // the next n <= 32 bits, zero-extended past the end
  inline uint32_t peek (int n) const {
    const size_t b = pos >> 3;
    uint64_t v;
    if (b + 8 <= nbytes) { memcpy (&v, p + b, 8); v = __builtin_bswap64 (v); }
    else { v = 0; for (size_t i = 0; i < 8; i++) v = (v << 8) | (b + i < nbytes ? p[b + i] : 0); }
    return n ? (uint32_t) ((v << (pos & 7)) >> (64 - n)) : 0;
  }
  inline void skip (int n) { pos += n; if (pos > nbits) err = true; }
  inline uint32_t u1() { if (pos >= nbits) { err = true; return 0; } uint32_t b = (p[pos >> 3] >> (7 - (pos & 7))) & 1; pos++; return b; }
  inline uint32_t u (int n) {             // n <= 32
    if (pos + (size_t)n > nbits) { err = true; pos = nbits; return 0; }
    const uint32_t v = peek (n); pos += n; return v;
  }
  uint32_t ue() {
    const uint32_t w = peek (32);
    if (w >> 16) {                        // at most 15 leading zeros: the whole code lies in the window
      const int z = __builtin_clz (w);
      const int len = 2 * z + 1;
      if (pos + (size_t)len > nbits) { err = true; pos = nbits; return 0; }
      pos += len;
      return (w >> (32 - len)) - 1;
    }
    int z = 0; while (!u1()) { if (err || ++z > 31) { err = true; return 0; } } return z == 0 ? 0 : ((1u << z) - 1 + u (z));
  }
  int32_t se() { uint32_t k = ue(); return (k & 1) ? (int32_t) ((k + 1) >> 1) : - (int32_t) (k >> 1); }
  bool byte_aligned() const { return (pos & 7) == 0; }
  bool more_rbsp_data() const;
};

struct Sps {
  bool valid = false;
  int profile_idc = 0, level_idc = 0, chroma_format_idc = 1;
  int log2_max_frame_num = 4, poc_type = 0, log2_max_poc_lsb = 4;
  bool delta_pic_order_always_zero = false;
  int offset_for_non_ref_pic = 0, offset_for_top_to_bottom = 0, num_ref_frames_in_poc_cycle = 0;
  std::vector<int> offset_for_ref_frame;
  int num_ref_frames = 0;
  bool gaps_allowed = false, frame_mbs_only = true, direct_8x8 = false;
  int mb_w = 0, mb_h = 0;
  int crop_l = 0, crop_r = 0, crop_t = 0, crop_b = 0;
  bool scaling_matrix_present = false;
  uint8_t sl4[6][16], sl8[2][64];       // raster order, after fall-back rule A
};
struct Pps {
  bool valid = false;
  int sps_id = 0;
  bool cabac = false, pic_order_present = false;
  int num_slice_groups = 1;
  int num_ref_idx_l0 = 1, num_ref_idx_l1 = 1;
  bool weighted_pred = false; int weighted_bipred_idc = 0;
  int pic_init_qp = 26, pic_init_qs = 26, chroma_qp_offset[2] = {0, 0};
  bool deblocking_control = false, constrained_intra_pred = false, redundant_pic_cnt = false;
  bool transform_8x8 = false, scaling_matrix_present = false;
  uint8_t sl4[6][16], sl8[2][64];
};

struct SliceHeader {
  int first_mb = 0, slice_type = 0 /*0 P, 2 I*/, pps_id = 0, frame_num = 0, idr_pic_id = 0;
  int poc_lsb = 0, delta_poc_bottom = 0, delta_poc[2] = {0, 0}, redundant_pic_cnt = 0;
  int num_ref_idx_l0 = 1;
  bool idr = false; int nal_ref_idc = 0;
  struct Reorder { int idc; uint32_t val; };
  std::vector<Reorder> reorder;
  bool has_weights = false; int luma_log2_denom = 0, chroma_log2_denom = 0;
  int luma_weight[32], luma_offset[32], chroma_weight[32][2], chroma_offset[32][2];
  bool no_output_of_prior = false, long_term_reference = false, adaptive_marking = false;
  struct Mmco { int op; uint32_t a, b; };
  std::vector<Mmco> mmco;
  int cabac_init_idc = 0, slice_qp = 26, deblock_idc = 0, alpha_off = 0, beta_off = 0;
};

// The macroblock syntax the recompressor codes beyond what lh264_mb_t carries: the fields of
// the reference's DecodedMacroblock that its emit code reads (decoded_macroblock.h:12-34, filled by the parser and by
// initRTDFromDecoderState decode_slice.cpp:81-109).  Several of them are whatever the decoder's persistent per-position
// arrays hold (pChromaPredMode, pIntraPredMode[][7], pSubMbType: only written by the macroblock types that own them).
#pragma pack(push, 1)
struct MbSyn {
  uint8_t have;              // 1: a coded (non-skipped) macroblock
  uint8_t slice_type, t8, cbp_c, cbp_l;
  uint8_t chroma_mode;       // pChromaPredMode[mb]: final chroma mode of the last intra macroblock at this position
  uint8_t luma16_mode;       // pIntraPredMode[mb][7]: final I16x16 mode of the last I16x16 macroblock at this position
  uint8_t luma_qp;
  uint32_t mb_type, num_ref_idx_l0;
  int32_t skip_run;          // length of the skip run that ended at this macroblock
  int8_t ref_idx[4];         // as parsed, per partition
  uint8_t sub_type[4];       // pSubMbType[mb]
  int8_t pred_mode[16];      // Intra4x4PredMode per block in z-order (I8x8: entries 0..3)
  int16_t mvd[16][2];        // motion vector differences as parsed, at the raster index of each partition's first block
  int32_t delta_qp;          // luma QP minus the previous coded macroblock's (0 before the first one of a slice)
  int32_t last_mb_qp;        // QP predictor in force when the macroblock was parsed
};
#pragma pack(pop)
static_assert (sizeof (MbSyn) == 116, "MbSyn layout");
struct SliceSyn { int32_t pad_bits, pad_value, transform8x8_pps, flags /* bit 0 entropy_coding_mode_flag, bit 1 constrained_intra_pred_flag */; };   // alignment bits after the slice's stop bit (decode_slice.cpp:3133-3148)

// zero-initialised array for the per-picture coefficient planes (768 bytes per macroblock, ~300x the size of the bitstream
// they come from).  Released blocks go to a small per-thread cache and are cleared on reuse: handing them back to the C library
// means fresh page faults for the next picture, and in a process that parses several streams side by side those serialise
// on the address-space lock (measured: 8 threads 1.8x one thread before, see DESIGN.md host front end).
void* zerobuf_get (size_t bytes, bool zero = true);
void zerobuf_put (void* p, size_t bytes);
template <typename T> class ZeroBuf {
 public:
  ZeroBuf() {}
  ~ZeroBuf() { if (p_) zerobuf_put (p_, n_ * sizeof (T)); }
  ZeroBuf (const ZeroBuf&) = delete;
  ZeroBuf& operator= (const ZeroBuf&) = delete;
  void assign_zero (size_t n, bool zero = true) { if (p_) zerobuf_put (p_, n_ * sizeof (T)); p_ = n ? (T*)zerobuf_get (n * sizeof (T), zero) : nullptr; n_ = p_ ? n : 0; }
  T* data() { return p_; }
  const T* data() const { return p_; }
  size_t size() const { return n_; }
  T& operator[] (size_t i) { return p_[i]; }
  const T& operator[] (size_t i) const { return p_[i]; }
 private:
  T* p_ = nullptr; size_t n_ = 0;
};

// std::vector on the same recycled blocks (sizes rounded up to 4 KB so that pictures of one stream reuse each other's blocks):
// the per-picture arrays of a batch are released by another thread than the one that parses the next streams, and handing
// them to the C library means trimmed heaps and fresh page faults for every wave of streams
// A parser can also carve the small per-picture arrays out of megabyte chunks of its own (StreamArena, set while it parses):
// nothing is returned block by block then - the chunks go back when the parser is destroyed - which is what keeps the release
// of a batch's pictures from competing with the parsing of the next ones.  A 16-byte header in front of every block says
// where it came from.
struct StreamArena {
  std::vector<std::pair<char*, size_t>> chunks;
  size_t used = 0;
  ~StreamArena();
  void* alloc (size_t bytes);
};
StreamArena*& current_stream_arena();               // thread local
template <typename T> struct PoolAlloc {
  typedef T value_type;
  PoolAlloc() {}
  template <typename U> PoolAlloc (const PoolAlloc<U>&) {}
  static size_t round (size_t n) { return ((n * sizeof (T)) + 16 + 4095) & ~ (size_t)4095; }
  T* allocate (size_t n) {
    char* p;
    if (StreamArena* a = current_stream_arena()) { p = (char*)a->alloc (n * sizeof (T) + 16); * (uint64_t*)p = 1; }
    else { p = (char*)zerobuf_get (round (n), false); * (uint64_t*)p = 0; }
    return (T*) (p + 16);
  }
  void deallocate (T* q, size_t n) {
    char* p = (char*)q - 16;
    if (* (uint64_t*)p == 0) zerobuf_put (p, round (n));
  }
  template <typename U> bool operator== (const PoolAlloc<U>&) const { return true; }
  template <typename U> bool operator!= (const PoolAlloc<U>&) const { return false; }
};
template <typename T> using PoolVec = std::vector<T, PoolAlloc<T>>;

// one parsed picture: exactly what lh264_recon_chains / lh264_ctx_index_chains consume
struct FrameOut {
  int id = 0, mb_w = 0, mb_h = 0, frame_num = 0, crop_w = 0, crop_h = 0, crop_x = 0, crop_y = 0;
  bool idr = false, is_ref = false, complete = false;
  PoolVec<lh264_mb_t> mbs;
  ZeroBuf<int16_t> coeffs, levels;
  std::vector<lh264_slice_t> slices;
  std::vector<int> ref_ids;             // ids of the pictures this one references (its job's ref slots)
  std::vector<int> dpb_ids;             // ids still marked 'used for reference' once this picture is done (others may be freed)
  int idr_pic_id = 0, nal_ref_idc = 0;
  std::vector<uint8_t> covered;
  PoolVec<MbSyn> syn;                   // per macroblock (row a10)
  std::vector<SliceSyn> slice_syn;      // per slice
  PoolVec<lh264_ctx_sym_t> syn_syms;       // the picture's row-a10 symbols (Symbolizer), macroblock after macroblock
  PoolVec<uint32_t> syn_off;               // mb_w*mb_h + 1 offsets
  PoolVec<uint8_t> lev_nonzero;            // per macroblock: it has a nonzero level (FreqImage's 'zeroed' test)
  PoolVec<uint64_t> sparse;                // sparse mode instead of `levels`: (macroblock * 384 + position) << 16 | level, nonzero levels only
};

// The recompressor's default stream (".pip" itself, stream id 0x7fffffff): the Annex-B input minus its slice data.
// Byte-level behaviour of the reference's BitStream (compression_stream.cpp:40-120, compression_stream.h:71-84): while
// escaping is on, bytes pass through a two-byte window that re-inserts emulation-prevention bytes.
class MainStreamWriter {
 public:
  std::vector<uint8_t> buffer;
  void append_byte (uint8_t x);
  void append_bytes (const uint8_t* d, size_t n) { for (size_t i = 0; i < n; i++) append_byte (d[i]); }
  void emit_bit (uint32_t bit);
  void emit_bits (uint32_t v, int n) {           // n <= 32, MSB first
    uint64_t acc = ((uint64_t)bits_ << n) | (n >= 32 ? (uint64_t)v : ((uint64_t)v & ((1ull << n) - 1)));
    int k = n_bits_ + n;
    while (k >= 8) { append_byte ((uint8_t) (acc >> (k - 8))); k -= 8; }
    bits_ = (uint32_t) (acc & ((1u << k) - 1u)); n_bits_ = k;
  }
  int bits_in_byte() const { return n_bits_; }
  void start_escape() { escaping_ = true; }
  void stop_escape();
  void pad_to_byte() { while (n_bits_ & 7) emit_bit (0); }
 private:
  uint32_t bits_ = 0; int n_bits_ = 0; bool escaping_ = false; uint8_t esc_[2] = {0, 0}; int esc_n_ = 0;
};

class Parser {
 public:
  Parser();
  ~Parser();
  // a whole Annex-B file, cut and fed the way the reference's console application does (h264dec.cpp:246-272, one
  // DecodeFrameNoDelay per start-code-delimited chunk); also builds the recompressor's default stream, main_stream()
  int feed_file (const uint8_t* data, size_t len);
  // headers only (the restore direction reads them from the default stream): SPS / PPS are remembered, for a slice NAL the
  // header is parsed and described; no picture is started.  nal = one NAL unit without start code.  <0: not parseable
  struct HeaderInfo {
    int nal_type = 0; bool is_slice = false; SliceHeader sh; int hdr_bits = 0, mb_w = 0, mb_h = 0;
    bool cabac = false, transform_8x8 = false, constrained_intra_pred = false;
  };
  int parse_headers (const uint8_t* nal, size_t len, HeaderInfo& out);
  const std::vector<uint8_t>& last_rbsp() const;     // the unescaped payload of the NAL handled last
  static void unescape (const uint8_t* d, size_t n, std::vector<uint8_t>& out);
  const std::vector<uint8_t>& main_stream() const { return main_.buffer; }
  // the samples of the stream's I_PCM macroblocks, 384 bytes each in decoding order: the reference's compressed representation does
  // not carry them (its own restore fails on such streams); ours does, as one more stream of the container (LH264_TAG_PCM)
  const std::vector<uint8_t>& pcm_samples() const { return pcm_; }
  // feed a whole Annex-B byte stream (or a piece that ends on a NAL boundary); completed pictures are appended to frames()
  int feed (const uint8_t* data, size_t len);
  int feed_nal (const uint8_t* nal, size_t len);      // one NAL unit without start code (with emulation prevention bytes)
  void flush();                                        // end of stream: completes the picture in progress
  bool picture_in_progress_is_whole() const;           // every macroblock of the picture being parsed has been covered by a slice
  std::vector<std::unique_ptr<FrameOut>>& frames() { return frames_; }
  const std::string& error() const { return err_; }       // first error since construction / clear_error()
  void clear_error() { err_.clear(); }
  int unsupported_count() const { return n_unsupported_; }
  // keep == false: a completed picture is counted and released at once (its buffers go back to the per-thread cache), as a
  // pipeline does after handing the records to the device; frames() then stays empty
  void set_keep_frames (bool keep) { keep_frames_ = keep; }
  // want == false: FrameOut::coeffs (the dequantised coefficients, only the reconstruct kernel reads them) stays empty -
  // the compress direction needs the raw levels only, and the planes are what the front end spends its memory bandwidth on
  void set_want_coeffs (bool want) { want_coeffs_ = want; }
  // lazy == true: FrameOut::levels is cleared macroblock by macroblock as coded macroblocks are parsed; the levels of skipped
  // or lost macroblocks are then undefined (no kernel reads them: lh264_ctx.hip returns on their type)
  void set_lazy_levels (bool lazy) { lazy_levels_ = lazy; }
  // sparse == true: FrameOut::levels stays empty and FrameOut::sparse lists the nonzero levels (a few per macroblock instead of
  // 768 bytes: what travels to the device, where lh264_compress_batch expands it)
  void set_sparse_levels (bool sparse) { sparse_levels_ = sparse; }
  // on: the per-picture arrays come from chunks owned by this parser (see StreamArena)
  void set_stream_arena (bool on) { if (on && !arena_) arena_.reset (new StreamArena()); else if (!on) arena_.reset(); }
  long pictures_done() const { return pictures_done_; }
  // a completed picture had macroblocks no slice covers (lost slices): the reference conceals them (error_concealment.cpp), which is not
  // modelled - the recompressed form of such a stream does not restore; callers that promise a round trip store it verbatim
  bool damaged() const { return damaged_; }

 private:
  struct Impl;
  std::unique_ptr<StreamArena> arena_;      // declared first: destroyed after the pictures that live in it
  std::unique_ptr<Impl> d_;
  std::vector<std::unique_ptr<FrameOut>> frames_;
  std::string err_;
  int n_unsupported_ = 0;
  bool keep_frames_ = true, want_coeffs_ = true, lazy_levels_ = false, sparse_levels_ = false; long pictures_done_ = 0; bool damaged_ = false;
  MainStreamWriter main_;
  std::vector<uint8_t> pcm_;
  friend struct Impl;
};

}  // namespace lh264host
