// lh264_coder.hip - the recompressor's adaptive binary arithmetic coder on the device.
//
// One workgroup = two wave64 per stream.  The symbol wave walks the stream's symbols in coding order (host list of syntax
// symbols per macroblock with the coefficient symbols of lh264_ctx_index_chains spliced in at the marker), 64 at a time:
// every lane binarises its own symbol (emitInt / emitUEGkInt / Branch<n> / emitBitsZeroToPow2Inclusive,
// /root/reference/codec/decoder/core/inc/compression_stream.h:117-166,455-591) and fetches its prior's cell - the adaptive
// probabilities (DynProb :87-115), one 64-byte cell of a per-stream open-addressing hash table in HBM - into an LDS row; the
// rows are advanced by their users in parallel, every decision taking the probability it is coded with; the decisions are
// sorted by tag and handed to the coding wave, whose lane t owns the libvpx bool coder (bitwriter.h:35-105) of tag slot t.
// The raw-bit probability TEST_PROB (:363,441-448) is shared by all tags and lives in a scalar of the symbol wave.  A DynProb is
// packed into 32 bits (two 10-bit counts and the probability the next decision will use, which is NOT derivable from the counts
// after a rescale), biased so that zero-filled memory is the initial state.  See DESIGN.md section 4.3.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/lh264.h"

namespace lh264 {

#define GLB __attribute__ ((address_space (1)))
typedef uint32_t u32x4 __attribute__ ((ext_vector_type (4)));
template <typename T> __device__ __forceinline__ GLB T* glb (const void* p) { return (GLB T*) (uintptr_t)p; }
__device__ __forceinline__ int uniform (int v) { return __builtin_amdgcn_readfirstlane (v); }

// ---- wave-wide inclusive scan / OR over 64 lanes with DPP: Hillis-Steele inside each row of 16 (row_shr 1, 2, 4, 8), then lane 15
// of rows 0 and 2 into rows 1 and 3 (row_bcast:15), then lane 31 into rows 2 and 3 (row_bcast:31).  Lanes without a source add 0.
template <int CTRL, int ROWS> __device__ __forceinline__ int dpp0 (int x) { return __builtin_amdgcn_update_dpp (0, x, CTRL, ROWS, 0xf, false); }
__device__ __forceinline__ int wave_scan_add (int x) {
  x += dpp0<0x111, 0xf> (x); x += dpp0<0x112, 0xf> (x); x += dpp0<0x114, 0xf> (x); x += dpp0<0x118, 0xf> (x);
  x += dpp0<0x142, 0xa> (x); x += dpp0<0x143, 0xc> (x);
  return x;
}
__device__ __forceinline__ uint32_t wave_or (uint32_t v) {
  int x = (int)v;
  x |= dpp0<0x111, 0xf> (x); x |= dpp0<0x112, 0xf> (x); x |= dpp0<0x114, 0xf> (x); x |= dpp0<0x118, 0xf> (x);
  x |= dpp0<0x142, 0xa> (x); x |= dpp0<0x143, 0xc> (x);
  return (uint32_t)__builtin_amdgcn_readlane (x, 63);
}

// ---- DynProb, packed -------------------------------------------------------------------------------------------------
__device__ __forceinline__ int dp_prob (uint32_t s) { return (int) (((s >> 20) + 128u) & 255u); }
__device__ __forceinline__ uint32_t dp_update (uint32_t s, int bit) {
  uint32_t c0 = s & 1023u, c1 = (s >> 10) & 1023u;
  if (bit) c1++; else c0++;
  // floor (256 (c0+1) / (c0+c1+2)) < 256: numerator < 2^18,
  // divisor <= 516: a float quotient is within one of the exact one
  const uint32_t num = 256u * (c0 + 1u), den = c0 + c1 + 2u;
  uint32_t prob = (uint32_t) ((float)num * __builtin_amdgcn_rcpf ((float)den));
  if (prob * den > num) prob--;
  else if ((prob + 1u) * den <= num) prob++;
  if (c0 + c1 > 512u) { c0 = (c0 + 1u) >> 1; c1 = (c1 + 1u) >> 1; }
  return c0 | (c1 << 10) | (((prob + 128u) & 255u) << 20);
}

// ---- per-lane bool coder (vpx_writer) with the carry resolved in registers instead of by re-reading the output ----------
struct Bc {
  uint32_t low, range, pos, ffrun;
  int count, pending, used, last;
};
__device__ __forceinline__ void bc_put (Bc& b, GLB uint8_t* out, uint32_t cap, int byte) {
  if (b.pos < cap) out[b.pos] = (uint8_t)byte;
  b.pos++;
  b.last = byte;
}
__device__ __forceinline__ void bc_byte (Bc& b, GLB uint8_t* out, uint32_t cap, int byte, bool carry) {
  if (carry) {                 // the pending byte takes the carry, the 0xff run behind it turns into zeros
    bc_put (b, out, cap, b.pending + 1);
    for (uint32_t i = 0; i < b.ffrun; i++) bc_put (b, out, cap, 0);
    b.ffrun = 0; b.pending = byte;
  } else if (b.pending < 0) b.pending = byte;
  else if (byte == 0xff) b.ffrun++;
  else {
    bc_put (b, out, cap, b.pending);
    for (uint32_t i = 0; i < b.ffrun; i++) bc_put (b, out, cap, 0xff);
    b.ffrun = 0; b.pending = byte;
  }
}
__device__ __forceinline__ void bc_write (Bc& b, GLB uint8_t* out, uint32_t cap, int bit, int prob) {      // vpx_write
  if (!b.used) { b.low = 0; b.range = 255; b.count = -24; b.pos = 0; b.ffrun = 0; b.pending = -1; b.used = 1; b.last = 0; }
  const uint32_t split = 1u + (((b.range - 1u) * (uint32_t)prob) >> 8);
  uint32_t range = split, low = b.low;
  if (bit) { low += split; range = b.range - split; }
  int shift = range < 128u ? __clz ((int)range) - 24 : 0;          // vpx_norm[range]
  range <<= shift;
  int count = b.count + shift;
  if (count >= 0) {
    const int offset = shift - count;
    const bool carry = ((low << (offset - 1)) & 0x80000000u) != 0;
    bc_byte (b, out, cap, (int) ((low >> (24 - offset)) & 0xffu), carry);
    low <<= offset;
    shift = count;
    low &= 0xffffffu;
    count -= 8;
  }
  low <<= shift;
  b.count = count; b.low = low; b.range = range;
}
__device__ __forceinline__ void bc_finish (Bc& b, GLB uint8_t* out, uint32_t cap) {      // vpx_stop_encode + flush of the held bytes
  for (int i = 0; i < 32; i++) bc_write (b, out, cap, 0, 128);
  if (b.pending >= 0) {
    bc_put (b, out, cap, b.pending);
    for (uint32_t i = 0; i < b.ffrun; i++) bc_put (b, out, cap, 0xff);
  }
  if ((b.last & 0xe0) == 0xc0) bc_put (b, out, cap, 0);
}

#ifdef LH264_CODER_STAMP
#define STAMP_FIELDS uint64_t st_t, st_acc[16];
#define STAMP_COUNT(c, i, v) (c).st_acc[i] += (uint64_t) (v);
#define STAMP(c, i) { const uint64_t st_n = __builtin_amdgcn_s_memtime(); (c).st_acc[i] += st_n - (c).st_t; (c).st_t = st_n; }
#else
#define STAMP_FIELDS
#define STAMP(c, i)
#define STAMP_COUNT(c, i, v)
#endif
// ---- what the symbol wave hands to the coding wave: a batch's decisions, probabilities resolved, sorted by tag --------------------
#ifndef LH264_CODER_BUFS
#define LH264_CODER_BUFS 1      // 2: one more batch in flight, 14 KB more LDS per stream (measured: no faster, fewer streams per CU)
#endif
struct Handoff {
  uint32_t sorted[LH264_CODER_BUFS][64 * 56];   // [buffer][decision words: bit 8 the bit, bits 24..31 the probability]
  uint32_t segtot[LH264_CODER_BUFS][64];        // [buffer][tag slot]: first word | number of words << 16
  uint32_t touch[LH264_CODER_BUFS][2];          // [buffer]: tag slots whose stream comes into existence with this batch (64-bit mask)
  int full[LH264_CODER_BUFS];    // buffer handed over, not yet coded
  int done;                      // no more batches; `pstatus` is final
  int pstatus;
};
// ---- the wave's coding context ---------------------------------------------------------------------------------------------
struct Coder {
  GLB uint32_t* keys; GLB uint32_t* cells; uint32_t mask;
  GLB uint8_t* out; uint32_t cap;
  int lane;
  uint32_t cellv;          // lanes 0..15: the 16 packed DynProbs of the cell in hand
  uint32_t cur_key, cur_slot; bool have_cell;
  uint32_t test_prob;      // TEST_PROB, wave-uniform
  int status;
  STAMP_FIELDS
  Handoff* H; int buf;        // hand-off buffers (LDS) and the one to fill next
};

__device__ __forceinline__ int tag_slot (int tag) { return tag == 69 ? 34 : tag; }

__device__ __forceinline__ void cell_flush (Coder& c) {
  if (c.have_cell && c.lane < 16) c.cells[(size_t)c.cur_slot * 16 + c.lane] = c.cellv;
  c.have_cell = false;
}
// make the cell of `key` the one in hand (find or insert)
__device__ __forceinline__ void cell_get (Coder& c, uint32_t key) {
#ifdef LH264_CODER_ABL_NOMEM
  key &= 63u;               // timing ablation: (nearly) always the cell in hand or an L1 hit
#endif
  if (c.have_cell && c.cur_key == key) return;
  cell_flush (c);
  uint32_t h = (key * 0x9E3779B1u) >> 7;
  for (int tries = 0; tries < 64; tries++, h += 64) {
    const uint32_t s = (h + (uint32_t)c.lane) & c.mask;
    const uint32_t kv = c.keys[s];
    const unsigned long long match = __ballot (kv == key + 1u), empty = __ballot (kv == 0u);
    if (match | empty) {
      const int p = __ffsll ((long long) (match | empty)) - 1;
      const uint32_t slot = (h + (uint32_t)p) & c.mask;
      if (! ((match >> p) & 1ull) && c.lane == 0) c.keys[slot] = key + 1u;      // a fresh cell: zero-filled memory is the initial state
      c.cur_slot = slot; c.cur_key = key; c.have_cell = true;
      c.cellv = c.lane < 16 ? c.cells[(size_t)slot * 16 + c.lane] : 0u;
      return;
    }
  }
  c.status = 1;             // table full
  c.cur_key = key; c.cur_slot = 0; c.have_cell = false; c.cellv = 0;
}

// ---- binarisation: a symbol becomes a short list of decisions ---------------------------------------------------------------
// Done by 64 lanes at once, one symbol per lane, into the lane's row of an LDS table.  A decision is one word: bits 0..7
// the place of its DynProb in the cell in hand (0xff: a raw bit, coded with TEST_PROB), bit 8 the bit, bits 16..23 the tag.
// A word with bit 31 set switches cells: the next word is the key (+1) of the cell the following decisions use (priors
// that are trees of more than 16 nodes span several cells).
#define DL_STRIDE 56
struct DList { uint32_t* row; int n; unsigned long long raw; bool sw; };            // row: LDS; raw: which entries are raw bits; sw: a cell switch occurs
__device__ __forceinline__ void push (DList& d, int j, int bit, int tag) {
  if (d.n < DL_STRIDE) { d.row[d.n] = (uint32_t) (j & 0xff) | ((uint32_t) (bit & 1) << 8) | ((uint32_t)tag << 16); if ((j & 0xff) == 0xff) d.raw |= 1ull << d.n; }
  d.n++;
}
__device__ __forceinline__ void push_switch (DList& d, uint32_t key) {
  if (d.n + 1 < DL_STRIDE) { d.row[d.n] = 0x80000000u; d.row[d.n + 1] = key + 1u; }
  d.n += 2; d.sw = true;
}
// UnaryIntPrior<n>::at(i) = prior[min(i, n-1)]; emitUnary compression_stream.h:465-474
__device__ __forceinline__ void b_unary (DList& d, int data, int base, int n, int early, int tag) {
  for (int i = 0; i < data; i++) {
    push (d, base + (i < n - 1 ? i : n - 1), 1, tag);
    if (i == early - 1) return;
  }
  push (d, base + (data < n - 1 ? data : n - 1), 0, tag);
}
// emitInt :523-572 with the prior's parts at fixed places of the cell (zero / sign < 0: the prior has none)
__device__ __forceinline__ void b_int (DList& d, int data, int zero, int sign, int ebase, int E, int mbase, int M, int order,
                                       int tag_exp, int tag_man, int tag_zero, int tag_sign) {
  if (zero >= 0) { push (d, zero, data == 0, tag_zero); if (data == 0) return; }
  if (sign >= 0) { push (d, sign, data > 0, tag_sign); if (data < 0) data = -data; }
  data--;
  const int data_high = 1 + (data >> order);
  const int log2 = 31 - __clz (data_high);               // largest l with (1 << l) <= data_high
  b_unary (d, log2, ebase, E, -1, tag_exp);
  int lo = 0, hi = M;
  const int nb = log2 + order;
  for (int i = 0; i < nb; i++) {
    const int bit = i < log2 ? (data_high >> (log2 - 1 - i)) & 1 : (data >> (order - 1 - (i - log2))) & 1;
    if (hi > lo) {
      const int mid = (hi + lo) / 2;
      push (d, mbase + mid, bit, tag_man);
      if (bit) lo = mid + 1; else hi = mid;
    } else push (d, 0xff, bit, tag_man);
  }
}
// emitUEGkInt :575-591; cell: zero 0, sign 1, first 2..2+M-1, second = {zero, exponent[E], mantissa[Mant]}
__device__ __forceinline__ void b_uegk (DList& d, int data, int N, int M, int E, int Mant, int order, int tag_exp, int tag_man, int tag_zero, int tag_sign) {
  push (d, 0, data == 0, tag_zero);
  if (data == 0) return;
  push (d, 1, data < 0, tag_sign);
  if (data < 0) data = -data;
  b_unary (d, data - 1, 2, M, N, tag_man);
  if (data - 1 >= N) b_int (d, data - 1 - N, 2 + M, -1, 2 + M + 1, E, 2 + M + 1 + E, Mant, order, tag_exp, tag_man, tag_zero, tag_sign);
}
// Branch<nbits> (:117-166): a node's array = itself, its 0-subtree, its 1-subtree.  With more than 16 nodes, node n lives
// in cell index * groups + n / 16, place n % 16.
__device__ __forceinline__ void b_tree (DList& d, uint32_t prior, int groups, unsigned off, unsigned data, int nbits, int tag, int cur_group) {
  const uint32_t index = prior & 0x7ffffffu;
  for (int n = nbits; n >= 1; n--) {
    const int bit = (data >> (n - 1)) & 1;
    if (groups > 1 && (int) (off >> 4) != cur_group) { cur_group = (int) (off >> 4); push_switch (d, (prior & 0xf8000000u) | (index * (uint32_t)groups + (off >> 4))); }
    push (d, (int) (off & 15u), bit, tag);
    off += bit ? 1u + ((1u << (n - 1)) - 1u) : 1u;
  }
}

// one symbol -> its decision list; returns the key (+1) of its (first) cell, 0 if it has none (raw bits);
// touch: the tag the symbol brings into existence (or -1)
__device__ __forceinline__ uint32_t build_symbol (DList& d, uint32_t prior, int value, int kind, int pad, int& touch) {
  enum { T_LDC = 17, T_CRDC = 18, T_LAC_0_EOB = 19, T_LAC_N_EOB = 24, T_CRAC_EOB = 29 };
  const int table = (int) (prior >> 27);
  const uint32_t index = prior & 0x7ffffffu;
  touch = -1;
  switch (kind) {
  case LH264_SYM_LUMA_DC: case LH264_SYM_CHROMA_DC: {      // IntPrior<3,4>: exponent 0..2, mantissa 3..6, zero 7, sign 8
    const int t = kind == LH264_SYM_LUMA_DC ? T_LDC : T_CRDC;
    b_int (d, value, 7, 8, 0, 3, 3, 4, 0, t, t, t, t);
    return LH264_PRIOR (kind == LH264_SYM_LUMA_DC ? LH264_TB_LDC : LH264_TB_CDC, prior) + 1u; }
  case LH264_SYM_NZ4: case LH264_SYM_NZ8: {                 // UnsignedIntPrior<3,4>
    const int t = ((prior / 27u) % 3u) ? T_CRAC_EOB : T_LAC_0_EOB;
    b_int (d, value, 7, -1, 0, 3, 3, 4, 0, t, t, t, t);
    return LH264_PRIOR (kind == LH264_SYM_NZ4 ? LH264_TB_NZ4 : LH264_TB_NZ8, prior) + 1u; }
  case LH264_SYM_AC4: case LH264_SYM_AC8: {                 // UEGkIntPrior<14,4,2,4,0>; tags by colour / first scan position (encode4x4)
    const uint32_t nco = kind == LH264_SYM_AC4 ? 16u : 64u;
    const uint32_t outer = prior / 3125u;
    const int emitted = (int) (outer % nco), color = (int) ((outer / nco) % 3u), code = (int) ((outer / nco / 3u) % 16u);
    const int first = color == 0 && emitted == 0 && code != 1;
    const int base = color ? T_CRAC_EOB : (first ? T_LAC_0_EOB : T_LAC_N_EOB);
    touch = base + 2;                                        // encode4x4 bills to tag(..._EXP): the stream exists from then on
    b_uegk (d, value, 14, 4, 2, 4, 0, base + 2, base + 3, base + 1, base + 4);
    return LH264_PRIOR (kind == LH264_SYM_AC4 ? LH264_TB_AC4 : LH264_TB_AC8, prior) + 1u; }
  case LH264_SYM_BIT:
    push (d, 0, value != 0, pad);
    return prior + 1u;
  case LH264_SYM_RAW:
    for (int i = 0; i < (int)prior; i++) push (d, 0xff, (value >> ((int)prior - 1 - i)) & 1, pad);
    return 0u;
  case LH264_SYM_MVD:                                       // UEGkIntPrior<9,4,3,4,3>
    b_uegk (d, value, 9, 4, 3, 4, 3, pad, pad, pad, pad);
    return prior + 1u;
  case LH264_SYM_TREE: {
    int nbits = 4, groups = 1;
    if (table == LH264_TB_SKIPRUN) { nbits = 9; groups = 32; } else if (table == LH264_TB_SUBMB) { nbits = 8; groups = 16; }
    else if (table == LH264_TB_CBPC) nbits = 2;
    b_tree (d, prior, groups, 0, (unsigned) (uint16_t)value, nbits, pad, 0);
    return ((prior & 0xf8000000u) | (index * (uint32_t)groups)) + 1u; }     // the cell of the tree's first 16 nodes
  case LH264_SYM_POW2: {                                    // emitBitsZeroToPow2Inclusive<nbits>: priors[0], then the tree in priors[1..]
    const bool qpl = table == LH264_TB_QPL;
    const int groups = qpl ? 8 : 1;
    const unsigned preferred = qpl ? 0u : index, data = (unsigned) (uint16_t)value;
    push (d, 0, data != preferred, pad);
    if (data != preferred) b_tree (d, prior, groups, 1, data > preferred ? data - 1u : data, qpl ? 7 : 3, pad, 0);
    return ((prior & 0xf8000000u) | (index * (uint32_t)groups)) + 1u; }
  default: return 0u;
  }
}

// one decision on the serial path: adaptive update of the cell in hand (or of TEST_PROB); the word, with the probability it
// is coded with, goes to its tag's list (lane t keeps the write position of tag slot t)
__device__ __forceinline__ void decide (Coder& c, uint32_t w, uint32_t* out, int& my_at) {
  const int j = (int) (w & 0xffu), bit = (int) ((w >> 8) & 1u), tag = (int) ((w >> 16) & 0xffu);
  int prob;
  if (j == 0xff) { prob = dp_prob (c.test_prob); c.test_prob = dp_update (c.test_prob, bit); }
  else {
    const uint32_t s = (uint32_t)__builtin_amdgcn_readlane ((int)c.cellv, j);
    prob = dp_prob (s);
    const uint32_t ns = dp_update (s, bit);
    if (c.lane == j) c.cellv = ns;
  }
  if (c.lane == tag_slot (tag)) { out[my_at] = (w & 0x00ffffffu) | ((uint32_t)prob << 24); my_at++; }
}

// Code a batch of up to 64 symbols held one per lane (sym = the 8-byte record).  Binarisation and memory latency are paid
// once per batch: every lane builds the decision list of its own symbol, looks up the symbol's cell in the hash table and
// fetches it into the wave's LDS rows, all in parallel; the decisions are then executed strictly in order, a cell shared by
// several symbols of the batch living in the row of the first of them; cells not found (new priors) and priors spanning
// several cells go to the table serially.
__device__ __forceinline__ void code_batch (Coder& c, uint64_t sym, int count, uint32_t* bcell /* LDS [64][16] */, uint32_t* dl /* LDS [64][DL_STRIDE] */) {
  const int lane = c.lane;
  const uint32_t prior = (uint32_t)sym, hi = (uint32_t) (sym >> 32);
  DList d; d.row = dl + lane * DL_STRIDE; d.n = 0; d.raw = 0ull; d.sw = false;
  int touch = -1;
  uint32_t key = 0;
  if (lane < count) key = build_symbol (d, prior, (int) (int16_t) (hi & 0xffffu), (int) ((hi >> 16) & 0xffu), (int) (hi >> 24), touch);
  if (d.n > DL_STRIDE) c.status = 8;
  const int nd = d.n;
  STAMP (c, 1)
  // the row of a cell = the first lane of the batch that uses it
  int owner = lane, rank = 0;                        // rank: how many earlier symbols of the batch use the same cell
  for (int j = 0; j < count; j++) {
    const uint32_t kj = (uint32_t)__builtin_amdgcn_readlane ((int)key, j);
    if (kj == key && j < owner) owner = j;
    if (kj == key && j < lane) rank++;
  }
  STAMP (c, 2)
  // parallel probe + fetch by the owners
  uint32_t slot = 0; bool found = false;
  if (key != 0u && owner == lane) {
    uint32_t h = ((key - 1u) * 0x9E3779B1u) >> 7;
    bool fresh = false;
    for (int p = 0; p < 16; p++) {
      const uint32_t s = (h + (uint32_t)p) & c.mask;
      uint32_t kv = c.keys[s];
      if (kv == 0u) kv = atomicCAS ((uint32_t*) (uintptr_t) (c.keys + s), 0u, key);      // a new prior: claim the slot (other lanes insert too)
      if (kv == key) { slot = s; found = true; break; }
      if (kv == 0u) { slot = s; found = true; fresh = true; break; }
    }
    if (found) {
      u32x4* dst = (u32x4*) (bcell + lane * 16);
      if (fresh) { const u32x4 z = {0u, 0u, 0u, 0u}; dst[0] = z; dst[1] = z; dst[2] = z; dst[3] = z; }      // zero = the initial state
      else {
        const GLB u32x4* src = (const GLB u32x4*) (c.cells + (size_t)slot * 16);
        dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
      }
    }
  }
  unsigned long long valid = __ballot (found);
  __builtin_amdgcn_wave_barrier();
  STAMP (c, 3)
#ifndef LH264_CODER_SERIAL
  // ---- the parallel way: no cell switch in the batch, every cell found --------------------------------------------------
  // The adaptive state of a DynProb depends only on the decisions made with that DynProb, the bool coder of a tag only on the
  // (probability, bit) pairs sent to that tag: so (1) TEST_PROB walks the raw bits of the batch in order, (2) every cell row is
  // advanced by the lanes whose symbols use it, one user after the other, all rows at once, each decision word taking the
  // probability it is coded with, (3) the words are sorted by tag, order kept, (4) lane t codes the list of tag slot t.
  const bool is_user = lane < count && key != 0u;
  if (!__ballot (d.sw || nd > DL_STRIDE || (is_user && !((valid >> owner) & 1ull)))) {
    uint32_t* row = dl + lane * DL_STRIDE;
    unsigned long long tmask = 0ull;                     // a stream exists once one of its symbols was billed (EXP tags)
    {
      unsigned long long tl = __ballot (touch >= 0);
      while (tl) {
        const int i = __ffsll ((long long)tl) - 1;
        const int ti = __builtin_amdgcn_readlane (touch, i);
        tmask |= 1ull << tag_slot (ti);
        tl &= ~__ballot (touch == ti);
      }
    }
    {                                                  // (1)
      unsigned long long rl = __ballot (d.raw != 0ull);
      while (rl) {
        const int i = __ffsll ((long long)rl) - 1;
        rl &= rl - 1ull;
        unsigned long long m = ((unsigned long long) (uint32_t)__builtin_amdgcn_readlane ((int) (d.raw >> 32), i) << 32) |
                               (uint32_t)__builtin_amdgcn_readlane ((int) (d.raw & 0xffffffffu), i);
        while (m) {
          const int t = __ffsll ((long long)m) - 1;
          m &= m - 1ull;
          const uint32_t w = dl[i * DL_STRIDE + t];
          const int prob = dp_prob (c.test_prob);
          c.test_prob = dp_update (c.test_prob, (int) ((w >> 8) & 1u));
          if (lane == 0) dl[i * DL_STRIDE + t] = (w & 0x00ffffffu) | ((uint32_t)prob << 24);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    STAMP (c, 7)
    // (2) and the tags a symbol uses (at most four), with their counts
    int s0 = -1, s1 = -1, s2 = -1, s3 = -1, n0 = 0, n1 = 0, n2 = 0, n3 = 0;
    bool many = false;
    {
      int maxrank = 0;
      for (unsigned long long b = __ballot (is_user && rank > 0); b; ) { maxrank++; b = __ballot (is_user && rank > maxrank); }
      uint32_t* cr = bcell + owner * 16;
      STAMP_COUNT (c, 12, 1) STAMP_COUNT (c, 13, maxrank + 1)
      { int mx = nd; for (int dd = 32; dd; dd >>= 1) mx = max (mx, __shfl_xor (mx, dd)); STAMP_COUNT (c, 14, mx) }
#ifdef LH264_CODER_STAMP
      {   // diagnostic: loop trips of the round scheme against the longest per-row chain (what an owner-driven pass would take)
        int ideal = 0;
        for (int j = 0; j < count; j++) { const int oj = __builtin_amdgcn_readlane (owner, j), nj = __builtin_amdgcn_readlane (nd, j); const uint32_t kj = (uint32_t)__builtin_amdgcn_readlane ((int)key, j); if (kj && lane == oj) ideal += nj; }
        for (int dd = 32; dd; dd >>= 1) ideal = max (ideal, __shfl_xor (ideal, dd));
        STAMP_COUNT (c, 6, ideal)
        for (int rnd = 0; rnd <= maxrank; rnd++) { int mx = ((is_user && rank == rnd) || (rnd == 0 && !is_user)) ? nd : 0; for (int dd = 32; dd; dd >>= 1) mx = max (mx, __shfl_xor (mx, dd)); STAMP_COUNT (c, 15, mx) }
      }
#endif
      for (int rnd = 0; rnd <= maxrank; rnd++) {
        if ((is_user && rank == rnd) || (rnd == 0 && !is_user)) {
          uint32_t wn = row[0];                          // next word fetched ahead: the compiler may not move it across the stores below
          for (int t = 0; t < nd; t++) {
            uint32_t w = wn;
            wn = row[t + 1 < DL_STRIDE ? t + 1 : t];
            const int j = (int) (w & 0xffu);
            if (j != 0xff) {
              const uint32_t sv = cr[j];
              w = (w & 0x00ffffffu) | ((uint32_t)dp_prob (sv) << 24);
              cr[j] = dp_update (sv, (int) ((w >> 8) & 1u));
              row[t] = w;
            }
            const int sl = tag_slot ((int) ((w >> 16) & 0xffu));
            if (sl == s0) n0++; else if (sl == s1) n1++; else if (sl == s2) n2++; else if (sl == s3) n3++;
            else if (s0 < 0) { s0 = sl; n0 = 1; } else if (s1 < 0) { s1 = sl; n1 = 1; } else if (s2 < 0) { s2 = sl; n2 = 1; }
            else if (s3 < 0) { s3 = sl; n3 = 1; } else many = true;
          }
        }
        asm volatile ("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
      }
    }
    STAMP (c, 8)
    if (__ballot (many)) c.status = 16;                // cannot happen with the binarisations above
    // (3) per tag: where each symbol's decisions go
    int b0 = 0, b1 = 0, b2 = 0, b3 = 0, my_seg = 0, my_tot = 0, running = 0;
    {
      const unsigned long long mine = (s0 >= 0 ? 1ull << s0 : 0ull) | (s1 >= 0 ? 1ull << s1 : 0ull) | (s2 >= 0 ? 1ull << s2 : 0ull) | (s3 >= 0 ? 1ull << s3 : 0ull);
      unsigned long long used = (unsigned long long)wave_or ((uint32_t)mine) | ((unsigned long long)wave_or ((uint32_t) (mine >> 32)) << 32);
      while (used) {
        const int T = __ffsll ((long long)used) - 1;
        used &= used - 1ull;
        const int v = (s0 == T ? n0 : 0) + (s1 == T ? n1 : 0) + (s2 == T ? n2 : 0) + (s3 == T ? n3 : 0);
        const int incl = wave_scan_add (v);
        const int tot = __builtin_amdgcn_readlane (incl, 63);
        const int at = running + incl - v;
        if (s0 == T) b0 = at; else if (s1 == T) b1 = at; else if (s2 == T) b2 = at; else if (s3 == T) b3 = at;
        if (lane == T) { my_seg = running; my_tot = tot; }
        running += tot;
      }
    }
    STAMP (c, 9)
    // the cell rows are final: back to the table while the words are sorted
    if (key != 0u && owner == lane && ((valid >> lane) & 1ull)) {
      GLB u32x4* dst = (GLB u32x4*) (c.cells + (size_t)slot * 16);
      const u32x4* src = (const u32x4*) (bcell + lane * 16);
      dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
    }
    // (4) hand the sorted words to the coding wave
    Handoff& H = *c.H;
    const int buf = c.buf;
    { volatile int* f = H.full; while (f[buf]) __builtin_amdgcn_s_sleep (1); }
    asm volatile ("" ::: "memory");
    STAMP (c, 11)
    uint32_t* sorted = H.sorted[buf];
    for (int t = 0; t < nd; t++) {
      const uint32_t w = row[t];
      const int sl = tag_slot ((int) ((w >> 16) & 0xffu));
      int at;
      if (sl == s0) at = b0++; else if (sl == s1) at = b1++; else if (sl == s2) at = b2++; else at = b3++;
      sorted[at] = w;
    }
    H.segtot[buf][lane] = (uint32_t)my_seg | ((uint32_t)my_tot << 16);
    if (lane == 0) { H.touch[buf][0] = (uint32_t)tmask; H.touch[buf][1] = (uint32_t) (tmask >> 32); }
    asm volatile ("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence (__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) { volatile int* f = H.full; f[buf] = 1; }
    c.buf = (buf + 1) % LH264_CODER_BUFS;
    STAMP (c, 10)
    return;
  }
#endif
  // ---- the serial way (a tree prior leaves its first cell, or a cell was not found by the parallel probe) ----------------------
  // first the size of every tag's list (lane t counts tag slot t), then the decisions one after the other, each word going to
  // its tag's list with the probability it is coded with
  Handoff& H = *c.H;
  const int buf = c.buf;
  int my_cnt = 0;
  unsigned long long tmask = 0ull;
  for (int i = 0; i < count; i++) {
    const int ni = __builtin_amdgcn_readlane (nd, i);
    const int ti = __builtin_amdgcn_readlane (touch, i);
    if (ti >= 0) tmask |= 1ull << tag_slot (ti);
    const uint32_t words = lane < ni ? dl[i * DL_STRIDE + lane] : 0u;
    for (int t = 0; t < ni; t++) {
      const uint32_t w = (uint32_t)__builtin_amdgcn_readlane ((int)words, t);
      if (w & 0x80000000u) t++;
      else if (lane == tag_slot ((int) ((w >> 16) & 0xffu))) my_cnt++;
    }
  }
  const int my_seg = wave_scan_add (my_cnt) - my_cnt;
  int my_at = my_seg;
  { volatile int* f = H.full; while (f[buf]) __builtin_amdgcn_s_sleep (1); }
  asm volatile ("" ::: "memory");
  uint32_t* sorted = H.sorted[buf];
  for (int i = 0; i < count; i++) {
    const uint32_t ki = (uint32_t)__builtin_amdgcn_readlane ((int)key, i);
    const int r = __builtin_amdgcn_readlane (owner, i);
    const int ni = __builtin_amdgcn_readlane (nd, i);
    const uint32_t words = lane < ni ? dl[i * DL_STRIDE + lane] : 0u;      // the symbol's decisions, one per lane
    if (ki) {
      if ((valid >> r) & 1ull) c.cellv = lane < 16 ? bcell[r * 16 + lane] : 0u;
      else {                       // not in the table yet (or probed too far away): find / insert serially
        c.have_cell = false;
        cell_get (c, ki - 1u);
        const uint32_t sl = c.cur_slot;
        if (lane == r) slot = sl;
        valid |= 1ull << r;
        c.have_cell = false;
      }
    }
    bool in_row = ki != 0u;
    for (int t = 0; t < ni; t++) {
      const uint32_t w = (uint32_t)__builtin_amdgcn_readlane ((int)words, t);
      if (w & 0x80000000u) {       // the tree leaves its first cell: the others go straight to the table and back
        const uint32_t k2 = (uint32_t)__builtin_amdgcn_readlane ((int)words, t + 1);
        if (in_row) { if (lane < 16) bcell[r * 16 + lane] = c.cellv; in_row = false; c.have_cell = false; }
        cell_get (c, k2 - 1u);      // (writes a previous out-of-row cell back first)
        t++;
      } else decide (c, w, sorted, my_at);
    }
    if (in_row) { if (lane < 16) bcell[r * 16 + lane] = c.cellv; }
    else cell_flush (c);
  }
  H.segtot[buf][lane] = (uint32_t)my_seg | ((uint32_t)my_cnt << 16);
  if (lane == 0) { H.touch[buf][0] = (uint32_t)tmask; H.touch[buf][1] = (uint32_t) (tmask >> 32); }
  asm volatile ("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence (__ATOMIC_RELEASE, "workgroup");
  if (lane == 0) { volatile int* f = H.full; f[buf] = 1; }
  c.buf = (buf + 1) % LH264_CODER_BUFS;
  STAMP (c, 4)
  // write the rows back
  if (key != 0u && owner == lane && ((valid >> lane) & 1ull)) {
    GLB u32x4* dst = (GLB u32x4*) (c.cells + (size_t)slot * 16);
    const u32x4* src = (const u32x4*) (bcell + lane * 16);
    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
  }
  __builtin_amdgcn_wave_barrier();
  STAMP (c, 5)
}

// Two waves per stream.  Wave 0 turns symbols into decisions with their probabilities (everything adaptive), wave 1 owns the 35
// bool coders (lane t = tag slot t) and codes the lists wave 0 hands over through an LDS buffer: while one batch is being
// coded the next one is being prepared (wave 0 needs the buffer again only when it has sorted the next batch).
__global__ void __launch_bounds__ (128)
coder_chain_kernel (const lh264_code_job_t* __restrict__ jobs, const int32_t* __restrict__ chain_first,
                    const lh264_code_stream_t* __restrict__ streams, int n_chains) {
  __shared__ uint32_t bcell[64 * 16];
  __shared__ uint32_t dl[64 * DL_STRIDE];
  __shared__ Handoff H;
  __shared__ uint64_t queue[128];
  const int chain = blockIdx.x;
  if (chain >= n_chains) return;
  const lh264_code_stream_t* S = streams + chain;
  const int wave = (int)threadIdx.x >> 6;
  if (threadIdx.x < LH264_CODER_BUFS) { H.full[threadIdx.x] = 0; }
  if (threadIdx.x == 0) { H.done = 0; H.pstatus = 0; }
  __syncthreads();
  if (wave == 1) {
    // ---- the coding wave ---------------------------------------------------------------------------------------------------
    const int lane = (int)threadIdx.x & 63;
    GLB uint8_t* out = glb<uint8_t> (S->out_dev);
    const uint32_t cap = S->out_cap;
    GLB uint8_t* o = out + (size_t)lane * cap;
    Bc bc;
    bc.used = 0; bc.pos = 0; bc.low = 0; bc.range = 255; bc.count = -24; bc.ffrun = 0; bc.pending = -1; bc.last = 0;
    volatile int* full = H.full;
    volatile int* done = &H.done;
    int buf = 0;
    for (;;) {
      while (!full[buf] && !*done) __builtin_amdgcn_s_sleep (1);
      if (!full[buf]) {                                 // done was seen: a batch handed over before it must still be coded
        __builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "workgroup");
        if (!full[buf]) break;
      }
      __builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "workgroup");
      asm volatile ("" ::: "memory");
      const uint32_t st = H.segtot[buf][lane];
      const int seg = (int) (st & 0xffffu), tot = (int) (st >> 16);
      const unsigned long long tm = (unsigned long long)H.touch[buf][0] | ((unsigned long long)H.touch[buf][1] << 32);
      if (((tm >> lane) & 1ull) && !bc.used) { bc.low = 0; bc.range = 255; bc.count = -24; bc.pos = 0; bc.ffrun = 0; bc.pending = -1; bc.used = 1; bc.last = 0; }
      const uint32_t* sorted = H.sorted[buf];
      uint32_t wn = tot > 0 ? sorted[seg] : 0u;
      for (int q = 0; __ballot (q < tot); q++) {
        if (q < tot) {
          const uint32_t w = wn;
          if (q + 1 < tot) wn = sorted[seg + q + 1];
#ifndef LH264_CODER_ABL_NOBC
          bc_write (bc, o, cap, (int) ((w >> 8) & 1u), (int) (w >> 24));
#else
          bc.low += w >> 24;
#endif
        }
      }
      asm volatile ("" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      if (lane == 0) full[buf] = 0;
      buf = (buf + 1) % LH264_CODER_BUFS;
    }
    GLB uint32_t* lens = glb<uint32_t> (S->out_len_dev);
    if (lane < LH264_N_TAG_SLOTS) {
      if (bc.used) bc_finish (bc, o, cap);
      lens[lane] = bc.used ? bc.pos : 0u;
    }
    const unsigned long long over = __ballot (lane < LH264_N_TAG_SLOTS && bc.used && bc.pos > cap);
    if (lane == 0) lens[LH264_N_TAG_SLOTS] = (uint32_t) (H.pstatus | (over ? 4 : 0));
    return;
  }
  // ---- the symbol wave -------------------------------------------------------------------------------------------------------
  Coder c;
  c.keys = glb<uint32_t> (S->hash_keys_dev); c.cells = glb<uint32_t> (S->hash_cells_dev); c.mask = S->hash_cap - 1u;
  c.out = glb<uint8_t> (S->out_dev); c.cap = S->out_cap;
  c.lane = (int)threadIdx.x;
  c.cellv = 0; c.cur_key = 0; c.cur_slot = 0; c.have_cell = false; c.test_prob = 0; c.status = 0;
  c.H = &H; c.buf = 0;
#ifdef LH264_CODER_STAMP
  c.st_t = __builtin_amdgcn_s_memtime(); for (int i = 0; i < 16; i++) c.st_acc[i] = 0;
#endif
  const int lane = c.lane;
  // One loop, one copy of the coder: fill the queue from the stream's symbol sources (host list of macroblock k, with the
  // coefficient symbols of macroblock k in place of the marker), then code a batch of up to 64.
  const int first = chain_first[chain], last = chain_first[chain + 1];
  int ji = first, k = 0, n = 0, mc = 0, cb = 0;
  uint32_t base = 0, o1 = 0;
  bool in_ctx = false, have_job = false, have_mb = false;
  const GLB uint64_t* hs = nullptr; const GLB uint32_t* off = nullptr; const GLB uint64_t* cs = nullptr; const GLB uint16_t* cn = nullptr;
  int qn = 0;
  for (;;) {
    // ---- fill ---------------------------------------------------------------------------------------------------------
    while (qn < 64) {
      if (!have_job) {
        if (ji >= last) break;
        const lh264_code_job_t* J = jobs + ji;
        hs = glb<const uint64_t> (J->syn_syms_dev); off = glb<const uint32_t> (J->syn_off_dev);
        cs = glb<const uint64_t> (J->ctx_syms_dev); cn = glb<const uint16_t> (J->ctx_n_syms_dev);
        n = J->n_mbs; k = 0; have_job = true; have_mb = false;
      }
      if (!have_mb) {
        if (k >= n) { have_job = false; ji++; continue; }
        base = (uint32_t)uniform ((int)off[k]); o1 = (uint32_t)uniform ((int)off[k + 1]);
        have_mb = true; in_ctx = false;
      }
      const int space = 128 - qn;
      if (in_ctx) {
        int t2 = mc - cb; if (t2 > 64) t2 = 64; if (t2 > space) t2 = space;
        if (t2 > 0) {
          const uint64_t cv = lane < t2 ? cs[(size_t)k * LH264_CTX_MAX_SYMS + cb + lane] : 0ull;
          if (lane < t2) queue[qn + lane] = cv;
          qn += t2; cb += t2;
        }
        if (cb >= mc) in_ctx = false;
        continue;
      }
      if (base >= o1) { have_mb = false; k++; continue; }
      int m = (int) (o1 - base); if (m > 64) m = 64; if (m > space) m = space;
      const uint64_t hv = lane < m ? hs[base + lane] : 0ull;
      const unsigned long long spl = __ballot (lane < m && ((hv >> 48) & 0xffull) == (unsigned long long)LH264_SYM_SPLICE);
      const int take = spl ? __ffsll ((long long)spl) - 1 : m;          // symbols before the marker (or all of them)
      if (lane < take) queue[qn + lane] = hv;
      qn += take; base += (uint32_t)take;
      if (spl) { base++; mc = uniform ((int)cn[k]); cb = 0; in_ctx = mc > 0; }
    }
    if (qn == 0) break;
    STAMP (c, 0)
    // ---- code ---------------------------------------------------------------------------------------------------------
    __builtin_amdgcn_wave_barrier();
    const int m = qn < 64 ? qn : 64;
    const uint64_t sym = lane < m ? queue[lane] : 0ull;
    const uint64_t mv = queue[64 + lane];
    __builtin_amdgcn_wave_barrier();
    code_batch (c, sym, m, bcell, dl);
    if (qn > 64) queue[lane] = mv;                   // what is left moves to the front
    qn -= m;
    __builtin_amdgcn_wave_barrier();
  }
  cell_flush (c);
  STAMP (c, 6)
#ifdef LH264_CODER_STAMP
  if (c.lane == 0) { GLB uint64_t* dbg = (GLB uint64_t*) (c.out + (size_t)39 * c.cap); for (int i = 0; i < 16; i++) dbg[i] = c.st_acc[i]; }
#endif
  if (c.lane == 0) H.pstatus = c.status;
  asm volatile ("" ::: "memory");
  __builtin_amdgcn_fence (__ATOMIC_RELEASE, "workgroup");
  if (c.lane == 0) { volatile int* dn = &H.done; *dn = 1; }
}

}  // namespace lh264
