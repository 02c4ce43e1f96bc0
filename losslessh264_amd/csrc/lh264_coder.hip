// lh264_coder.hip - the recompressor's adaptive binary arithmetic coder on the device.
//
// One workgroup = one wave64 per stream.  The wave walks the stream's symbols in coding order (host list of syntax
// symbols per macroblock with the coefficient symbols of lh264_ctx_index_chains spliced in at the marker); the
// binarisation of a symbol (emitInt / emitUEGkInt / Branch<n> / emitBitsZeroToPow2Inclusive,
// /root/reference/codec/decoder/core/inc/compression_stream.h:117-166,455-591) is wave-uniform scalar work, the adaptive
// probabilities of the symbol's prior (DynProb :87-115) are one 64-byte cell of a per-stream open-addressing hash table
// in HBM held in lanes 0..15 while the symbol is coded, and lane t owns the libvpx bool coder (bitwriter.h:35-105) of tag
// slot t: a decision for tag t is coded by lane t alone.  The raw-bit probability TEST_PROB (:363,441-448) is shared by
// all tags and lives in a scalar.  A DynProb is packed into 32 bits (two 10-bit counts and the probability the next
// decision will use, which is NOT derivable from the counts after a rescale), biased so that zero-filled memory is the
// initial state.
//
// Serial by nature: throughput comes from the number of streams (one wave each) - the dependent chain per symbol is one
// hash probe + one cell fetch; see DESIGN.md.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/lh264.h"

namespace lh264 {

#define GLB __attribute__ ((address_space (1)))
template <typename T> __device__ __forceinline__ GLB T* glb (const void* p) { return (GLB T*) (uintptr_t)p; }
__device__ __forceinline__ int uniform (int v) { return __builtin_amdgcn_readfirstlane (v); }

// ---- DynProb, packed -------------------------------------------------------------------------------------------------
__device__ __forceinline__ int dp_prob (uint32_t s) { return (int) (((s >> 20) + 128u) & 255u); }
__device__ __forceinline__ uint32_t dp_update (uint32_t s, int bit) {
  uint32_t c0 = s & 1023u, c1 = (s >> 10) & 1023u;
  if (bit) c1++; else c0++;
  const uint32_t prob = (256u * (c0 + 1u)) / (c0 + c1 + 2u);
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

// ---- the wave's coding context ---------------------------------------------------------------------------------------------
struct Coder {
  GLB uint32_t* keys; GLB uint32_t* cells; uint32_t mask;
  GLB uint8_t* out; uint32_t cap;
  int lane;
  uint32_t cellv;          // lanes 0..15: the 16 packed DynProbs of the cell in hand
  uint32_t cur_key, cur_slot; bool have_cell;
  uint32_t test_prob;      // TEST_PROB, wave-uniform
  int status;
  Bc bc;                   // this lane's tag
};

__device__ __forceinline__ int tag_slot (int tag) { return tag == 69 ? 34 : tag; }

__device__ __forceinline__ void cell_flush (Coder& c) {
  if (c.have_cell && c.lane < 16) c.cells[(size_t)c.cur_slot * 16 + c.lane] = c.cellv;
  c.have_cell = false;
}
// make the cell of `key` the one in hand (find or insert)
__device__ __forceinline__ void cell_get (Coder& c, uint32_t key) {
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

// one decision with DynProb j of the cell in hand, coded into tag `tag`
__device__ __forceinline__ void decide (Coder& c, int j, int bit, int tag) {
  const uint32_t s = (uint32_t)__builtin_amdgcn_readlane ((int)c.cellv, j);
  const int prob = dp_prob (s);
  const uint32_t ns = dp_update (s, bit);
  if (c.lane == j) c.cellv = ns;
  const int slot = tag_slot (tag);
  if (c.lane == slot) bc_write (c.bc, c.out + (size_t)slot * c.cap, c.cap, bit, prob);
}
__device__ __forceinline__ void decide_raw (Coder& c, int bit, int tag) {        // emitBit(bit): the shared TEST_PROB
  const int prob = dp_prob (c.test_prob);
  c.test_prob = dp_update (c.test_prob, bit);
  const int slot = tag_slot (tag);
  if (c.lane == slot) bc_write (c.bc, c.out + (size_t)slot * c.cap, c.cap, bit, prob);
}
__device__ __forceinline__ void touch_tag (Coder& c, int tag) {       // a stream exists once tag() was called for it
  const int slot = tag_slot (tag);
  if (c.lane == slot && !c.bc.used) { c.bc.low = 0; c.bc.range = 255; c.bc.count = -24; c.bc.pos = 0; c.bc.ffrun = 0; c.bc.pending = -1; c.bc.used = 1; c.bc.last = 0; }
}

// UnaryIntPrior<n>::at(i) = prior[min(i, n-1)]; emitUnary compression_stream.h:465-474
__device__ __forceinline__ void emit_unary (Coder& c, int data, int base, int n, int early, int tag) {
  for (int i = 0; i < data; i++) {
    decide (c, base + (i < n - 1 ? i : n - 1), 1, tag);
    if (i == early - 1) return;
  }
  decide (c, base + (data < n - 1 ? data : n - 1), 0, tag);
}
// emitInt :523-572 with the prior's parts at fixed places of the cell (zero / sign < 0: the prior has none)
__device__ __forceinline__ void emit_int (Coder& c, int data, int zero, int sign, int ebase, int E, int mbase, int M, int order,
                                          int tag_exp, int tag_man, int tag_zero, int tag_sign) {
  if (zero >= 0) { decide (c, zero, data == 0, tag_zero); if (data == 0) return; }
  if (sign >= 0) { decide (c, sign, data > 0, tag_sign); if (data < 0) data = -data; }
  data--;
  int log2 = 0;
  const int data_high = 1 + (data >> order);
  while ((2 << log2) <= data_high) log2++;
  emit_unary (c, log2, ebase, E, -1, tag_exp);
  int lo = 0, hi = M;
  const int nb = log2 + order;
  for (int i = 0; i < nb; i++) {
    const int bit = i < log2 ? (data_high >> (log2 - 1 - i)) & 1 : (data >> (order - 1 - (i - log2))) & 1;
    if (hi > lo) {
      const int mid = (hi + lo) / 2;
      decide (c, mbase + mid, bit, tag_man);
      if (bit) lo = mid + 1; else hi = mid;
    } else decide_raw (c, bit, tag_man);
  }
}
// emitUEGkInt :575-591; cell: zero 0, sign 1, first 2..2+M-1, second = {zero, exponent[E], mantissa[Mant]}
__device__ __forceinline__ void emit_uegk (Coder& c, int data, int N, int M, int E, int Mant, int order, int tag_exp, int tag_man, int tag_zero, int tag_sign) {
  decide (c, 0, data == 0, tag_zero);
  if (data == 0) return;
  decide (c, 1, data < 0, tag_sign);
  if (data < 0) data = -data;
  emit_unary (c, data - 1, 2, M, N, tag_man);
  if (data - 1 >= N) emit_int (c, data - 1 - N, 2 + M, -1, 2 + M + 1, E, 2 + M + 1 + E, Mant, order, tag_exp, tag_man, tag_zero, tag_sign);
}
// Branch<nbits> (:117-166): a node's array = itself, its 0-subtree, its 1-subtree.  Tables with more than 16 nodes span several
// cells: node n lives in cell (index * groups + n / 16), place n % 16.
__device__ __forceinline__ void emit_tree (Coder& c, uint32_t base_key, int groups, int first_node, unsigned data, int nbits, int tag) {
  unsigned off = (unsigned)first_node;
  for (int n = nbits; n >= 1; n--) {
    const int bit = (data >> (n - 1)) & 1;
    if (groups > 1) cell_get (c, (base_key & 0xf8000000u) | ((base_key & 0x7ffffffu) * (uint32_t)groups + (off >> 4)));
    decide (c, (int) (off & 15u), bit, tag);
    const unsigned children = (1u << (n - 1)) - 1u;
    off += bit ? 1u + children : 1u;
  }
}

__device__ __forceinline__ void code_symbol (Coder& c, uint32_t prior, int value, int kind, int pad) {
  const int table = (int) (prior >> 27);
  const uint32_t index = prior & 0x7ffffffu;
  enum { T_LDC = 17, T_CRDC = 18, T_LAC_0_EOB = 19, T_LAC_N_EOB = 24, T_CRAC_EOB = 29 };
  switch (kind) {
  case LH264_SYM_LUMA_DC: case LH264_SYM_CHROMA_DC: {      // IntPrior<3,4>: exponent 0..2, mantissa 3..6, zero 7, sign 8
    cell_get (c, LH264_PRIOR (kind == LH264_SYM_LUMA_DC ? LH264_TB_LDC : LH264_TB_CDC, prior));
    const int t = kind == LH264_SYM_LUMA_DC ? T_LDC : T_CRDC;
    emit_int (c, value, 7, 8, 0, 3, 3, 4, 0, t, t, t, t);
    break; }
  case LH264_SYM_NZ4: case LH264_SYM_NZ8: {                 // UnsignedIntPrior<3,4>
    cell_get (c, LH264_PRIOR (kind == LH264_SYM_NZ4 ? LH264_TB_NZ4 : LH264_TB_NZ8, prior));
    const int t = ((prior / 27u) % 3u) ? T_CRAC_EOB : T_LAC_0_EOB;
    emit_int (c, value, 7, -1, 0, 3, 3, 4, 0, t, t, t, t);
    break; }
  case LH264_SYM_AC4: case LH264_SYM_AC8: {                 // UEGkIntPrior<14,4,2,4,0>; tags by colour / first scan position (encode4x4)
    const uint32_t nco = kind == LH264_SYM_AC4 ? 16u : 64u;
    const uint32_t outer = prior / 3125u;
    const int emitted = (int) (outer % nco), color = (int) ((outer / nco) % 3u), code = (int) ((outer / nco / 3u) % 16u);
    const int first = color == 0 && emitted == 0 && code != 1;
    const int base = color ? T_CRAC_EOB : (first ? T_LAC_0_EOB : T_LAC_N_EOB);
    touch_tag (c, base + 2);
    cell_get (c, LH264_PRIOR (kind == LH264_SYM_AC4 ? LH264_TB_AC4 : LH264_TB_AC8, prior));
    emit_uegk (c, value, 14, 4, 2, 4, 0, base + 2, base + 3, base + 1, base + 4);
    break; }
  case LH264_SYM_TREE: {
    int nbits = 4, groups = 1;
    if (table == LH264_TB_SKIPRUN) { nbits = 9; groups = 32; } else if (table == LH264_TB_SUBMB) { nbits = 8; groups = 16; }
    else if (table == LH264_TB_CBPC) nbits = 2;
    if (groups == 1) cell_get (c, prior);
    emit_tree (c, prior, groups, 0, (unsigned) (uint16_t)value, nbits, pad);
    break; }
  case LH264_SYM_POW2: {                                    // emitBitsZeroToPow2Inclusive<nbits>: priors[0], then the tree in priors[1..]
    const bool qpl = table == LH264_TB_QPL;
    const int nbits = qpl ? 7 : 3, groups = qpl ? 8 : 1;
    const unsigned preferred = qpl ? 0u : index, data = (unsigned) (uint16_t)value;
    cell_get (c, (prior & 0xf8000000u) | (index * (uint32_t)groups));
    decide (c, 0, data != preferred, pad);
    if (data != preferred) emit_tree (c, prior, groups, 1, data > preferred ? data - 1u : data, nbits, pad);
    break; }
  case LH264_SYM_BIT:
    cell_get (c, prior);
    decide (c, 0, value != 0, pad);
    break;
  case LH264_SYM_RAW:
    for (int i = 0; i < (int)prior; i++) decide_raw (c, (value >> ((int)prior - 1 - i)) & 1, pad);
    break;
  case LH264_SYM_MVD:                                       // UEGkIntPrior<9,4,3,4,3>
    cell_get (c, prior);
    emit_uegk (c, value, 9, 4, 3, 4, 3, pad, pad, pad, pad);
    break;
  default: c.status = 2; break;
  }
}

__global__ void __launch_bounds__ (64)
coder_chain_kernel (const lh264_code_job_t* __restrict__ jobs, const int32_t* __restrict__ chain_first,
                    const lh264_code_stream_t* __restrict__ streams, int n_chains) {
  const int chain = blockIdx.x;
  if (chain >= n_chains) return;
  const lh264_code_stream_t* S = streams + chain;
  Coder c;
  c.keys = glb<uint32_t> (S->hash_keys_dev); c.cells = glb<uint32_t> (S->hash_cells_dev); c.mask = S->hash_cap - 1u;
  c.out = glb<uint8_t> (S->out_dev); c.cap = S->out_cap;
  c.lane = (int)threadIdx.x;
  c.cellv = 0; c.cur_key = 0; c.cur_slot = 0; c.have_cell = false; c.test_prob = 0; c.status = 0;
  c.bc.used = 0; c.bc.pos = 0; c.bc.low = 0; c.bc.range = 255; c.bc.count = -24; c.bc.ffrun = 0; c.bc.pending = -1; c.bc.last = 0;
  const int first = chain_first[chain], last = chain_first[chain + 1];
  for (int ji = first; ji < last; ji++) {
    const lh264_code_job_t* J = jobs + ji;
    const GLB lh264_ctx_sym_t* hs = glb<const lh264_ctx_sym_t> (J->syn_syms_dev);
    const GLB uint32_t* off = glb<const uint32_t> (J->syn_off_dev);
    const GLB lh264_ctx_sym_t* cs = glb<const lh264_ctx_sym_t> (J->ctx_syms_dev);
    const GLB uint16_t* cn = glb<const uint16_t> (J->ctx_n_syms_dev);
    const int n = J->n_mbs;
    for (int k = 0; k < n; k++) {
      const uint32_t o0 = (uint32_t)uniform ((int)off[k]), o1 = (uint32_t)uniform ((int)off[k + 1]);
      for (uint32_t si = o0; si < o1; si++) {
        const uint64_t raw = * (const GLB uint64_t*) (hs + si);
        const uint32_t prior = (uint32_t)uniform ((int) (uint32_t)raw);
        const uint32_t hi = (uint32_t)uniform ((int) (uint32_t) (raw >> 32));
        const int kind = (int) ((hi >> 16) & 0xffu);
        if (kind == LH264_SYM_SPLICE) {
          const int m = uniform ((int)cn[k]);
          const GLB lh264_ctx_sym_t* q = cs + (size_t)k * LH264_CTX_MAX_SYMS;
          for (int ci = 0; ci < m; ci++) {
            const uint64_t r2 = * (const GLB uint64_t*) (q + ci);
            const uint32_t p2 = (uint32_t)uniform ((int) (uint32_t)r2), h2 = (uint32_t)uniform ((int) (uint32_t) (r2 >> 32));
            code_symbol (c, p2, (int) (int16_t) (h2 & 0xffffu), (int) ((h2 >> 16) & 0xffu), 0);
          }
        } else code_symbol (c, prior, (int) (int16_t) (hi & 0xffffu), kind, (int) (hi >> 24));
      }
    }
  }
  cell_flush (c);
  GLB uint32_t* lens = glb<uint32_t> (S->out_len_dev);
  if (c.lane < LH264_N_TAG_SLOTS) {
    if (c.bc.used) bc_finish (c.bc, c.out + (size_t)c.lane * c.cap, c.cap);
    lens[c.lane] = c.bc.used ? c.bc.pos : 0u;
  }
  const unsigned long long over = __ballot (c.lane < LH264_N_TAG_SLOTS && c.bc.used && c.bc.pos > c.cap);
  if (c.lane == 0) lens[LH264_N_TAG_SLOTS] = (uint32_t) (c.status | (over ? 4 : 0));
}

}  // namespace lh264
