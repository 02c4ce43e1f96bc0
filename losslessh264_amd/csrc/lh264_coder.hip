// lh264_coder.hip - the recompressor's adaptive binary arithmetic coder on the device.
//
// The reference codes a stream strictly serially: symbol -> binarisation (emitInt / emitUEGkInt / Branch<n> /
// emitBitsZeroToPow2Inclusive, /root/reference/codec/decoder/core/inc/compression_stream.h:117-166,455-591) -> per decision an
// adaptive probability (DynProb :87-115) -> the libvpx bool coder of the decision's tag (bitwriter.h:35-105).  Only two things in
// that chain are really sequential: the state of ONE DynProb over the decisions made with it, and the state of ONE tag's bool coder
// over the decisions sent to it.  Everything else is data parallel, so the work is cut into kernels along those two lines.  This file
// is the form that scales INSIDE a stream (round 3); lh264_coder_sw.hip keeps round 2's first stages (a stream per workgroup), and
// lh264_capi.hip:code_binarise picks one per call.  Launch order:
//
//   coder_jobs_kernel         segments (<= 64 consecutive macroblocks of a picture) before every picture; picture -> stream    (small)
//   coder_count_kernel        a wave per segment, a lane per symbol: decisions per tag and per bucket of cells, in closed form
//                             (sym_count); the segment's symbols swept as two dense ranges                                  (parallel)
//   coder_balance_kernel      per stream: its 128 buckets of cells dealt to its 8 / 16 partitions, largest first              (small)
//   coder_partoff_kernel      per segment: where each partition's run of decision words starts                              (small)
//   coder_scan_kernel, coder_bases_kernel   where each segment's words and its entries of every tag's list start; stream bases (small)
//   coder_emit_kernel         a wave per segment, a lane per DECISION: the decision in closed form (decision_at) as a 64-bit word - key
//                             of the DynProb's cell, place, bit, the entry of its tag's list it will fill - into its partition's run
//                                                                                                                            (parallel)
//   coder_resolve_kernel      a wave per (stream, partition), 64 decisions per round: a DynProb is two counters, so the probability
//                             a decision is coded with follows from the counters before the round and PREFIX COUNTS of the earlier
//                             decisions of the round on the same DynProb; lanes holding the same DynProb find each other with
//                             ballots.  Counters in a keyed LDS cache in front of a spill table in HBM.  Output: (probability of
//                             the bit that occurred, bit) at the word's entry of its tag's list.  A stream's waves share one
//                             XCD's L2 and keep near one another (the 2-byte entries of a sector come from all of them)
//                                                                                              (serial per partition, 64 wide)
//   coder_chunkmap_kernel, coder_range_seed / first / link / walk / scan_kernel   the bool coder's range recurrence per tag list,
//                             in coarse chunks from checked candidate start states                        (serial per chunk)
//   coder_accum_kernel, coder_bytes_kernel   the bool coder's `low`: addends summed per output byte position by chunks of the
//                             list, then the carries and the bytes                                                          (parallel)
//
// Halving is lazy: the table holds the un-halved pair of counters and the reader halves when their sum has passed 512, so the
// probability always follows from the pair; zero-filled memory is the initial state.  See DESIGN.md section 4.3.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/lh264.h"
#include "lh264_coder.h"

namespace lh264 {

#define GLB __attribute__ ((address_space (1)))
#define LDS __attribute__ ((address_space (3)))
typedef uint32_t u32x4 __attribute__ ((ext_vector_type (4)));
template <typename T> __device__ __forceinline__ GLB T* glb (const void* p) { return (GLB T*) (uintptr_t)p; }
__device__ __forceinline__ int uniform (int v) { return __builtin_amdgcn_readfirstlane (v); }

// ---- wave-wide inclusive scan over 64 lanes with DPP: Hillis-Steele inside each row of 16 (row_shr 1, 2, 4, 8), then lane 15
// of rows 0 and 2 into rows 1 and 3 (row_bcast:15), then lane 31 into rows 2 and 3 (row_bcast:31).  Lanes without a source add 0.
template <int CTRL, int ROWS> __device__ __forceinline__ int dpp0 (int x) { return __builtin_amdgcn_update_dpp (0, x, CTRL, ROWS, 0xf, false); }
__device__ __forceinline__ int wave_scan_add (int x) {
  x += dpp0<0x111, 0xf> (x); x += dpp0<0x112, 0xf> (x); x += dpp0<0x114, 0xf> (x); x += dpp0<0x118, 0xf> (x);
  x += dpp0<0x142, 0xa> (x); x += dpp0<0x143, 0xc> (x);
  return x;
}

// ---- DynProb (compression_stream.h:87-115): the probability from the two counters -------------------------------------------
// floor (256 (c0+1) / (c0+c1+2)) < 256: numerator < 2^18, divisor <= 516: a float quotient is within one of the exact one
__device__ __forceinline__ uint32_t dp_ratio (uint32_t c0, uint32_t c1) {
  const uint32_t num = 256u * (c0 + 1u), den = c0 + c1 + 2u;
  uint32_t prob = (uint32_t) ((float)num * __builtin_amdgcn_rcpf ((float)den));
  if (prob * den > num) prob--;
  else if ((prob + 1u) * den <= num) prob++;
  return prob;
}

__device__ __forceinline__ int tag_slot (int tag) { return tag == 69 ? 34 : tag; }

// the tag of a coefficient / nonzero-count symbol: the context-index kernel leaves it in the symbol's pad byte (lh264_ctx.hip mk_sym);
// symbols from elsewhere (pad 0) have it taken out of the prior: colour, first scan position and macroblock class (encode4x4)
__device__ __forceinline__ int ac_tag_base (uint32_t prior, int kind, int pad) {
  if (pad) return pad;
  const uint32_t nco = kind == LH264_SYM_AC4 ? 16u : 64u;
  const uint32_t outer = prior / 3125u;
  const int emitted = (int) (outer % nco), color = (int) ((outer / nco) % 3u), code = (int) ((outer / nco / 3u) % 16u);
  const int first = color == 0 && emitted == 0 && code != 1;
  return color ? 29 : (first ? 19 : 24);
}
__device__ __forceinline__ int nz_tag (uint32_t prior, int pad) { return pad ? pad : (((prior / 27u) % 3u) ? 29 : 19); }
// the key raw bits are filed under: the shared TEST_PROB (compression_stream.h:363,441-448) as a cell of its own
#define CODER_RAW_KEY 0xf8000000u
// ---- how many decisions a symbol becomes, per tag, without walking its binarisation (the walk: lh264_coder_sw.hip binarize) -------
// n: all decisions; up to four (tag slot, count) pairs (-1: unused); tch: tag brought into existence; raws: how many of the decisions are
// raw bits (the shared TEST_PROB); key: the cell the others use (trees over several cells: the first one).
// The lanes of a wave hold symbols of all kinds: the integer-like ones (DC, nonzero count, coefficient, motion vector difference -
// emitInt / emitUEGkInt with different constants) go through ONE instruction stream with the constants in registers, not through a
// branch per kind (round-3 counters: 19 of 64 lanes were active on average in the per-kind version).
struct SymCount { int n, s0, s1, s2, s3, n0, n1, n2, n3, tch, raws; uint32_t key; };
template <bool CTX_ONLY = false> __device__ __forceinline__ SymCount sym_count (uint32_t prior, int value, int kind, int pad) {
  enum { T_LDC = 17, T_CRDC = 18, T_LAC_0_EOB = 19, T_LAC_N_EOB = 24, T_CRAC_EOB = 29 };
  SymCount c; c.n = 0; c.s0 = c.s1 = c.s2 = c.s3 = -1; c.n0 = c.n1 = c.n2 = c.n3 = 0; c.tch = -1; c.raws = 0; c.key = prior;
  const int table = (int) (prior >> 27);
  // CTX_ONLY: the symbol is known to come from the context-index kernel (a coefficient, a nonzero count or a DC level)
  const bool ac = kind == LH264_SYM_AC4 || kind == LH264_SYM_AC8, mvd = !CTX_ONLY && kind == LH264_SYM_MVD;
  const bool dc = kind == LH264_SYM_LUMA_DC || kind == LH264_SYM_CHROMA_DC, nzk = kind == LH264_SYM_NZ4 || kind == LH264_SYM_NZ8;
  if (CTX_ONLY || ac || mvd || dc || nzk) {
    // [zero flag] [sign] [unary up to N ones; beyond it: zero flag of the escape] [emitInt tail: l2 + 1 exponent, l2 + order mantissa decisions]
    const int N = ac ? 14 : mvd ? 9 : 0, order = mvd ? 3 : 0;
    const int av = value < 0 ? -value : value;
    int nz = 1, ns = 0, ne = 0, nm = 0, data = 0;
    if (value != 0) {
      ns = nzk ? 0 : 1;
      if (N) { const int u = av - 1; nm = u >= N ? N : u + 1; if (u >= N) { nz++; data = u - N; } }
      else data = av;
    }
    if (data > 0) {
      const int a = data - 1, high = 1 + (a >> order), l2 = 31 - __clz (high), nb = l2 + order;
      ne = l2 + 1; nm += nb;
      // raw bits: the mantissa decisions beyond the binary search over its four priors (man_place < 0)
      const int b0 = 0 < l2 ? (high >> (l2 - 1)) & 1 : (a >> ((order - 1) & 31)) & 1, b1 = 1 < l2 ? (high >> ((l2 - 2) & 31)) & 1 : (a >> ((order - 2 + l2) & 31)) & 1;
      c.raws = nb <= 2 ? 0 : nb - 2 - ((!b0 && !b1) ? 1 : 0);
    }
    c.n = nz + ns + ne + nm;
    c.key = mvd ? prior : LH264_PRIOR (kind == LH264_SYM_LUMA_DC ? LH264_TB_LDC : kind == LH264_SYM_CHROMA_DC ? LH264_TB_CDC : kind == LH264_SYM_NZ4 ? LH264_TB_NZ4 :
                                       kind == LH264_SYM_NZ8 ? LH264_TB_NZ8 : kind == LH264_SYM_AC4 ? LH264_TB_AC4 : LH264_TB_AC8, prior);
    if (ac) {                                                   // tags by colour / first scan position (encode4x4)
      const int base = ac_tag_base (prior, kind, pad);
      c.tch = tag_slot (base + 2);
      c.s0 = tag_slot (base + 1); c.n0 = nz;
      if (ns) { c.s1 = tag_slot (base + 4); c.n1 = ns; }
      if (nm) { c.s2 = tag_slot (base + 3); c.n2 = nm; }
      if (ne) { c.s3 = tag_slot (base + 2); c.n3 = ne; }
    } else {
      c.s0 = tag_slot (mvd ? pad : kind == LH264_SYM_LUMA_DC ? T_LDC : kind == LH264_SYM_CHROMA_DC ? T_CRDC : nz_tag (prior, pad));
      c.n0 = c.n;
    }
    return c;
  }
  const bool qpl = table == LH264_TB_QPL;
  const unsigned preferred = qpl ? 0u : (prior & 0x7ffffffu);
  c.n = kind == LH264_SYM_BIT ? 1 : kind == LH264_SYM_RAW ? ((int)prior > 0 ? (int)prior : 0) :
        kind == LH264_SYM_TREE ? (table == LH264_TB_SKIPRUN ? 9 : table == LH264_TB_SUBMB ? 8 : table == LH264_TB_CBPC ? 2 : 4) :
        kind == LH264_SYM_POW2 ? 1 + ((unsigned) (uint16_t)value != preferred ? (qpl ? 7 : 3) : 0) : 0;
  if (kind == LH264_SYM_RAW) { c.raws = c.n; c.key = CODER_RAW_KEY; }
  // trees over several cells: the first cell (cell_part gives all of them the same partition)
  if (kind == LH264_SYM_TREE && (table == LH264_TB_SKIPRUN || table == LH264_TB_SUBMB)) c.key = (prior & 0xf8000000u) | ((prior & 0x7ffffffu) << (table == LH264_TB_SKIPRUN ? 5 : 4));
  if (kind == LH264_SYM_POW2 && qpl) c.key = (prior & 0xf8000000u) | ((prior & 0x7ffffffu) << 3);
  if (c.n > 0) { c.s0 = tag_slot (pad); c.n0 = c.n; }          // the host's symbols name their tag
  return c;
}

// ---- decision j of a symbol in closed form (the same cases as sym_count above, without walking the binarisation) ----
// The parallel binarisation works a lane per DECISION: the j-th decision of a symbol follows from kind, value and j alone.
// key: the cell of the DynProb (LH264_PRIOR form), place: its place in the cell; raw bits (coded with the shared TEST_PROB,
// compression_stream.h:363,441-448) carry CODER_RAW_KEY, place 0.
struct Decision { uint32_t key; int place, bit, tag; };
// place of mantissa decision i in emitInt's binary search over 4 mantissa priors (:559-571), given the two bits before it; -1: raw
__device__ __forceinline__ int man_place (int i, int b0, int b1) { return i == 0 ? 2 : i == 1 ? (b0 ? 3 : 1) : (i == 2 && !b0 && !b1) ? 0 : -1; }
// emitInt behind its zero flag and sign: decision j of data >= 1 - unary exponent on ebase.. (E places), then mantissa on mbase..
__device__ __forceinline__ void int_tail_at (Decision& d, int j, int data, int order, int ebase, int E, int mbase, int tag_exp, int tag_man) {
  const int a = data - 1, high = 1 + (a >> order), l2 = 31 - __clz (high);
  if (j <= l2) { d.place = ebase + min (j, E - 1); d.bit = j < l2; d.tag = tag_exp; return; }
  const int i = j - l2 - 1;
  auto mbit = [&] (int k) -> int { return k < l2 ? (high >> (l2 - 1 - k)) & 1 : (a >> ((order - 1 - (k - l2)) & 31)) & 1; };     // (k beyond the mantissa: unused)
  const int pl = man_place (i, mbit (0), mbit (1));
  d.bit = mbit (i); d.tag = tag_man;
  if (pl < 0) { d.key = CODER_RAW_KEY; d.place = 0; } else d.place = mbase + pl;
}
// emitUEGkInt (:575-591): zero 0, sign 1, unary on 2..5 up to N ones, escape = emitInt (zero at 6, exponent 7.., mantissa 7+E..)
__device__ __forceinline__ void uegk_at (Decision& d, int j, int value, int N, int E, int order, int tag_exp, int tag_man, int tag_zero, int tag_sign) {
  if (j == 0) { d.place = 0; d.bit = value == 0; d.tag = tag_zero; return; }
  if (j == 1) { d.place = 1; d.bit = value < 0; d.tag = tag_sign; return; }
  const int u = (value < 0 ? -value : value) - 1, nu = u >= N ? N : u + 1;
  int i = j - 2;
  if (i < nu) { d.place = 2 + min (i, 3); d.bit = i < u; d.tag = tag_man; return; }
  i -= nu;                                                    // the escape: u >= N
  if (i == 0) { d.place = 6; d.bit = u - N == 0; d.tag = tag_zero; return; }
  int_tail_at (d, i - 1, u - N, order, 7, E, 7 + E, tag_exp, tag_man);
}
// Branch<nbits> (:117-166): node offset after the first j bits = one per zero bit + the subtree sizes skipped by the one bits
__device__ __forceinline__ void tree_at (Decision& d, uint32_t prior, int groups, unsigned off0, unsigned data, int nbits, int j) {
  data &= (1u << nbits) - 1u;
  const unsigned top = j ? data >> (nbits - j) : 0u;
  const unsigned off = off0 + (unsigned)j - (unsigned)__popc (top) + (top << (nbits - j));
  d.key = (prior & 0xf8000000u) | ((prior & 0x7ffffffu) * (uint32_t)groups + (off >> 4));
  d.place = (int) (off & 15u);
  d.bit = (int) ((data >> (nbits - 1 - j)) & 1u);
}
__device__ __forceinline__ Decision decision_at (uint32_t prior, int value, int kind, int pad, int j) {
  enum { T_LDC = 17, T_CRDC = 18, T_LAC_0_EOB = 19, T_LAC_N_EOB = 24, T_CRAC_EOB = 29 };
  Decision d; d.key = prior; d.place = 0; d.bit = 0; d.tag = pad;
  const int table = (int) (prior >> 27);
  switch (kind) {
  case LH264_SYM_LUMA_DC: case LH264_SYM_CHROMA_DC: case LH264_SYM_NZ4: case LH264_SYM_NZ8: {      // IntPrior<3,4> / UnsignedIntPrior<3,4>
    const bool dc = kind == LH264_SYM_LUMA_DC || kind == LH264_SYM_CHROMA_DC;
    d.tag = kind == LH264_SYM_LUMA_DC ? T_LDC : kind == LH264_SYM_CHROMA_DC ? T_CRDC : nz_tag (prior, pad);
    d.key = LH264_PRIOR (kind == LH264_SYM_LUMA_DC ? LH264_TB_LDC : kind == LH264_SYM_CHROMA_DC ? LH264_TB_CDC : kind == LH264_SYM_NZ4 ? LH264_TB_NZ4 : LH264_TB_NZ8, prior);
    if (j == 0) { d.place = 7; d.bit = value == 0; break; }
    if (dc && j == 1) { d.place = 8; d.bit = value > 0; break; }
    int_tail_at (d, j - (dc ? 2 : 1), value < 0 ? -value : value, 0, 0, 3, 3, d.tag, d.tag);
    break; }
  case LH264_SYM_AC4: case LH264_SYM_AC8: {                 // UEGkIntPrior<14,4,2,4,0>; tags by colour / first scan position (encode4x4)
    const int base = ac_tag_base (prior, kind, pad);
    d.key = LH264_PRIOR (kind == LH264_SYM_AC4 ? LH264_TB_AC4 : LH264_TB_AC8, prior);
    uegk_at (d, j, value, 14, 2, 0, base + 2, base + 3, base + 1, base + 4);
    break; }
  case LH264_SYM_BIT: d.bit = value != 0; break;
  case LH264_SYM_RAW: d.key = CODER_RAW_KEY; d.bit = (value >> ((int)prior - 1 - j)) & 1; break;
  case LH264_SYM_MVD: uegk_at (d, j, value, 9, 3, 3, pad, pad, pad, pad); break;      // UEGkIntPrior<9,4,3,4,3>
  case LH264_SYM_TREE: {
    int nbits = 4, groups = 1;
    if (table == LH264_TB_SKIPRUN) { nbits = 9; groups = 32; } else if (table == LH264_TB_SUBMB) { nbits = 8; groups = 16; }
    else if (table == LH264_TB_CBPC) nbits = 2;
    tree_at (d, prior, groups, 0u, (unsigned) (uint16_t)value, nbits, j);
    break; }
  case LH264_SYM_POW2: {                                    // emitBitsZeroToPow2Inclusive<nbits>: priors[0], then the tree in priors[1..]
    const bool qpl = table == LH264_TB_QPL;
    const int groups = qpl ? 8 : 1;
    const unsigned preferred = qpl ? 0u : (prior & 0x7ffffffu), data = (unsigned) (uint16_t)value;
    if (j == 0) { d.key = (prior & 0xf8000000u) | ((prior & 0x7ffffffu) * (uint32_t)groups); d.bit = data != preferred; break; }
    tree_at (d, prior, groups, 1u, data > preferred ? data - 1u : data, qpl ? 7 : 3, j - 1);
    break; }
  default: break;
  }
  d.tag = tag_slot (d.tag);
  return d;
}
// which of the stream's P partitions (a power of two) the DynProbs of a cell belong to: every partition is resolved by a wave of its own
// The cells a Branch tree spreads over (skip run: 32, sub-macroblock type: 16, the QP delta: 8 cells per prior, tree_at) share the partition
// of the tree's prior: every symbol then has all its modelled decisions in ONE partition (and its raw bits in TEST_PROB's).
__device__ __forceinline__ uint32_t cell_part (uint32_t key, int log2p) {
  const uint32_t table = key >> 27;
  const uint32_t sh = table == LH264_TB_SKIPRUN ? 5u : table == LH264_TB_SUBMB ? 4u : table == LH264_TB_QPL ? 3u : 0u;
  const uint32_t k = (key & 0xf8000000u) | ((key & 0x7ffffffu) >> sh);
  return log2p ? ((k ^ (k >> 15)) * 0x2C1B3C6Du) >> (32 - log2p) : 0u;
}

// a decision word (64 bits), as the binarisation leaves it for the resolve kernel: bits 0..31 the key of the DynProb's cell (LH264_PRIOR
// form; CODER_RAW_KEY: the shared TEST_PROB), 32..35 the place in the cell, 36 the bit, 37..63 the entry of the stream's tag lists the
// decision becomes (a stream's lists hold fewer than 2^27 entries per call; coder_scan_kernel checks it)
#define CODER_QPOS_BITS 27
// the cells are hashed into this many buckets; a stream's partitions are made of its buckets (coder_balance_kernel)
#define CODER_LOG2_BUCKETS LH264_CODER_MAX_LOG2P
#define CODER_BUCKETS (1 << CODER_LOG2_BUCKETS)

// order LDS traffic between the lanes of one wave: the DS instructions of a wave execute in issue order, so all that is needed is to
// keep the COMPILER from moving memory operations across this point
__device__ __forceinline__ void wsync() { asm volatile ("" ::: "memory"); __builtin_amdgcn_wave_barrier(); asm volatile ("" ::: "memory"); }

// ---- segments: the unit of the parallel binarisation -----------------------------------------------------------------------------
// A segment = up to CODER_SEG consecutive macroblocks of one picture, worked on by ONE wave (a workgroup = four segments; no
// workgroup barrier anywhere).  Its symbols in coding order are, macroblock after macroblock, the host list with the coefficient
// symbols in place of the marker; the wave lays that order out once in LDS (where each macroblock's symbols start, where its marker is).
#define CODER_SEG LH264_CODER_SEG_MBS
struct SegLds {
  uint32_t hoff[CODER_SEG + 1];      // host symbols of macroblock k start here (offsets into the picture's list)
  uint32_t sbase[CODER_SEG + 1];     // symbols of the segment before macroblock k, in coding order
  uint16_t mc[CODER_SEG];            // coefficient symbols of macroblock k
  uint16_t p[CODER_SEG];             // position of the marker in macroblock k's host list (0xffff: none)
};
struct Seg { const lh264_code_job_t* J; int job, k0, n; uint32_t total; };
// which picture and which macroblocks segment `b` is: seg0[] = segments before picture j, seg_job[] = the picture of every segment
__device__ __forceinline__ bool seg_locate (const lh264_code_job_t* jobs, const uint32_t* seg0, const uint32_t* seg_job, int n_jobs, uint32_t b, Seg& S) {
  if (b >= seg0[n_jobs]) return false;
  const uint32_t j = seg_job[b];
  S.job = (int)j; S.J = jobs + j;
  S.k0 = (int) (b - seg0[j]) * CODER_SEG;
  S.n = min (CODER_SEG, S.J->n_mbs - S.k0);
  return S.n > 0;
}
// lay the segment out (one wave); afterwards L.sbase[S.n] = S.total symbols
__device__ __forceinline__ void seg_layout (LDS SegLds& L, Seg& S, int lane) {
  const GLB uint32_t* off = glb<const uint32_t> (S.J->syn_off_dev) + S.k0;
  const GLB uint16_t* cn = glb<const uint16_t> (S.J->ctx_n_syms_dev) + S.k0;
  if (lane < S.n) { L.hoff[lane] = off[lane]; L.mc[lane] = cn[lane]; L.p[lane] = 0xffffu; }
  if (lane == 0) L.hoff[S.n] = off[S.n];
  wsync();
  // the markers: every host symbol of the segment is looked at once
  const GLB uint64_t* hs = glb<const uint64_t> (S.J->syn_syms_dev);
  const uint32_t h0 = L.hoff[0], h1 = L.hoff[S.n];
  for (uint32_t h = h0 + (uint32_t)lane; h < h1; h += 64u) {
    if (((hs[h] >> 48) & 0xffull) == (unsigned long long)LH264_SYM_SPLICE) {
      uint32_t lo = 0, hi = (uint32_t)S.n;               // the macroblock whose list holds position h
      while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (L.hoff[mid] <= h) lo = mid; else hi = mid; }
      L.p[lo] = (uint16_t) (h - L.hoff[lo]);
    }
  }
  wsync();
  // symbols per macroblock, running sum (at most 64 macroblocks: one per lane)
  uint32_t v = 0;
  if (lane < S.n) { const uint32_t nh = L.hoff[lane + 1] - L.hoff[lane]; v = L.p[lane] != 0xffffu ? nh - 1u + L.mc[lane] : nh; }
  const uint32_t incl = (uint32_t)wave_scan_add ((int)v);
  if (lane < S.n) L.sbase[lane] = incl - v;
  if (lane == 63) L.sbase[S.n] = incl;
  wsync();
  S.total = L.sbase[S.n];
}
// where the coefficient symbols of macroblock k of a picture start (in symbols behind ctx_syms_dev): its fixed slot, or - compact
// layout - the picture's first symbol in the pool + the macroblock's offset
__device__ __forceinline__ size_t ctx_sym_at (const lh264_code_job_t* J, int k) {
  return J->ctx_sym_off_dev ? (size_t)*glb<const unsigned long long> (J->ctx_sym_base_dev) + glb<const uint32_t> (J->ctx_sym_off_dev)[k] : (size_t)k * LH264_CTX_MAX_SYMS;
}
// symbols s0 + 64 q + lane, q = 0 .. 3, of the segment (0 beyond its end): the four searches advance together, so that a step waits for
// LDS once, not four times, and the four loads are under way together
__device__ __forceinline__ void seg_symbol4 (const LDS SegLds& L, const Seg& S, uint32_t s0, int lane, uint64_t out[4]) {
  uint32_t s[4], lo[4], hi[4];
#pragma unroll
  for (int q = 0; q < 4; q++) { s[q] = s0 + 64u * (uint32_t)q + (uint32_t)lane; lo[q] = 0; hi[q] = (uint32_t)S.n; }
  for (int it = 0; it < 7; it++) {                        // S.n <= 64: six halvings (a seventh step changes nothing)
    uint32_t v[4];
#pragma unroll
    for (int q = 0; q < 4; q++) v[q] = L.sbase[(lo[q] + hi[q]) >> 1];
#pragma unroll
    for (int q = 0; q < 4; q++) { const uint32_t mid = (lo[q] + hi[q]) >> 1; if (hi[q] - lo[q] > 1u) { if (v[q] <= s[q]) lo[q] = mid; else hi[q] = mid; } }
  }
  const GLB uint64_t* hsb = glb<const uint64_t> (S.J->syn_syms_dev);
  const GLB uint64_t* cs = glb<const uint64_t> (S.J->ctx_syms_dev);
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const uint32_t i = s[q] - L.sbase[lo[q]], p = L.p[lo[q]], mc = p != 0xffffu ? L.mc[lo[q]] : 0u;
    const GLB uint64_t* hs = hsb + L.hoff[lo[q]];
    const GLB uint64_t* src = (i < p || p == 0xffffu) ? hs + i : i < p + mc ? cs + ctx_sym_at (S.J, S.k0 + (int)lo[q]) + (i - p) : hs + (i - mc + 1u);
    out[q] = s[q] < S.total ? *src : 0ull;
  }
}

// ---- kernel 0: segments before each picture; which stream a picture belongs to ----------------------------------------------------
__global__ void __launch_bounds__ (CODER_ONE_WG)
coder_jobs_kernel (const lh264_code_job_t* __restrict__ jobs, const int32_t* __restrict__ chain_first, int n_jobs, int n_chains, unsigned seg_bound,
                   uint32_t* __restrict__ seg0, uint32_t* __restrict__ seg_job, uint32_t* __restrict__ job_chain, uint32_t* __restrict__ chain_info) {
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int j0 = 0; j0 < n_jobs; j0 += CODER_ONE_WG) {
    const int j = j0 + tid;
    const int v = j < n_jobs ? (max (jobs[j].n_mbs, 0) + CODER_SEG - 1) / CODER_SEG : 0;
    const int incl = wave_scan_add (v);
    if (lane == 63) wsum[wave] = (uint32_t)incl;
    __syncthreads();
    uint32_t before = carry;
    for (int w = 0; w < wave; w++) before += wsum[w];
    if (j < n_jobs) {
      seg0[j] = before + (uint32_t) (incl - v);
      for (uint32_t q = before + (uint32_t) (incl - v); q < before + (uint32_t)incl && q < seg_bound; q++) seg_job[q] = (uint32_t)j;
    }
    __syncthreads();
    if (tid == CODER_ONE_WG - 1) carry = before + (uint32_t)incl;
    __syncthreads();
  }
  if (tid == 0) seg0[n_jobs] = carry;
  // more segments than the caller's total_mbs made room for: the grids of the segment kernels would not reach all of them
  const uint32_t st0 = carry > seg_bound ? (uint32_t)LH264_CODER_ST_COUNT : 0u;
  for (int c = tid; c < n_chains; c += CODER_ONE_WG) {
    for (int j = chain_first[c]; j < chain_first[c + 1]; j++) job_chain[j] = (uint32_t)c;
    chain_info[(size_t)c * LH264_CODER_INFO_WORDS + LH264_CODER_INFO_STATUS] = st0;
    for (int q = 90; q < 96; q++) chain_info[(size_t)c * LH264_CODER_INFO_WORDS + q] = 0;
  }
}

// ---- kernel 1: decisions per tag and per partition of every segment -----------------------------------------------------------------
// seg_cnt[segment][0 .. LH264_N_TAG_SLOTS-1] decisions per tag slot (bit 31: the segment brings the tag's stream into existence),
// [LH264_N_TAG_SLOTS] all decisions; seg_part[segment][p] = decisions of the segment in partitions < p ([P]: all), i.e. where partition
// p's run starts inside the segment's decision words.
// Counters: one 16-bit column per lane in LDS ([slot][lane], two lanes to a dword) - the symbols of a step mostly count towards the
// same few tags, and 64 lanes adding to one LDS word take 64 turns (that was the round-2 kernel's whole time); a lane's column is
// its own bank.  A lane sees at most 64 x 528 / 64 symbols of at most 46 decisions: the 16-bit columns cannot overflow.
#define CNT_SLOTS (LH264_N_TAG_SLOTS + 1)            // columns: the tag slots, then all decisions
struct CountLds { SegLds seg; uint32_t col[CNT_SLOTS * 32]; uint32_t pcnt[LH264_CODER_MAX_PARTS]; };
__global__ void __launch_bounds__ (64 * LH264_CODER_WG_WAVES)
coder_count_kernel (const lh264_code_job_t* __restrict__ jobs, const uint32_t* __restrict__ seg0, const uint32_t* __restrict__ seg_job, int n_jobs,
                    uint32_t* __restrict__ seg_cnt, uint32_t* __restrict__ seg_bkt) {
  const int log2p = CODER_LOG2_BUCKETS;            // (the cells are counted per bucket; coder_balance_kernel makes partitions of the buckets)
  __shared__ CountLds Lg[LH264_CODER_WG_WAVES];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  LDS CountLds& L = * (LDS CountLds*) (uintptr_t) (uint32_t) (uintptr_t)&Lg[wave];
  const uint32_t seg = blockIdx.x * (uint32_t)LH264_CODER_WG_WAVES + (uint32_t)wave;
  Seg S;
  if (!seg_locate (jobs, seg0, seg_job, n_jobs, seg, S)) return;
  for (int i = lane; i < CNT_SLOTS * 32; i += 64) L.col[i] = 0;
  for (int i = lane; i < LH264_CODER_MAX_PARTS; i += 64) L.pcnt[i] = 0;
  seg_layout (L.seg, S, lane);
  LDS uint32_t* mycol = &L.col[lane >> 1];
  const uint32_t sh = 16u * (uint32_t) (lane & 1);
  auto add = [&] (int slot, uint32_t n) { __hip_atomic_fetch_add (mycol + slot * 32, n << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
  // (the partitions: one counter each for the wave - the lanes of a step spread over them)
  auto padd = [&] (uint32_t part, uint32_t n) { __hip_atomic_fetch_add (&L.pcnt[part], n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
  unsigned long long touch = 0;
  const uint32_t praw = cell_part (CODER_RAW_KEY, log2p);
  auto account = [&] (const SymCount c) {
    if (c.s0 >= 0) add (c.s0, (uint32_t)c.n0);
    if (c.s1 >= 0) add (c.s1, (uint32_t)c.n1);
    if (c.s2 >= 0) add (c.s2, (uint32_t)c.n2);
    if (c.s3 >= 0) add (c.s3, (uint32_t)c.n3);
    if (c.tch >= 0) touch |= 1ull << c.tch;
    add (LH264_N_TAG_SLOTS, (uint32_t)c.n);
    // partitions: the symbol's cell, TEST_PROB's for its raw bits
    if (c.raws) padd (praw, (uint32_t)c.raws);
    if (c.n > c.raws) padd (cell_part (c.key, log2p), (uint32_t) (c.n - c.raws));
  };
  // Counting does not need the coding order.  With the compact pool the segment's symbols are two dense ranges - the host's lists and
  // the context-index kernel's symbols - which are swept with coalesced loads (no search for the macroblock of a symbol).  Not so:
  // fixed slots per macroblock, a pool with gaps, or coefficient symbols of a macroblock whose host list has no marker (they are not coded).
  bool dense = S.J->ctx_sym_off_dev != 0;
  uint32_t off0 = 0, nctx = 0;
  if (dense) {
    const GLB uint32_t* so = glb<const uint32_t> (S.J->ctx_sym_off_dev) + S.k0;
    const uint32_t myoff = lane < S.n ? so[lane] : 0u, mymc = lane < S.n ? (uint32_t)L.seg.mc[lane] : 0u;
    const uint32_t nextoff = (uint32_t)__shfl_down ((int)myoff, 1);
    const bool bad = lane < S.n && ((mymc != 0u && L.seg.p[lane] == 0xffffu) || (lane + 1 < S.n && myoff + mymc != nextoff));
    dense = __ballot (bad) == 0ull;
    off0 = (uint32_t)__builtin_amdgcn_readfirstlane ((int)myoff);
    nctx = (uint32_t)__builtin_amdgcn_readlane ((int) (myoff + mymc), S.n - 1) - off0;
  }
  if (dense) {
    const GLB uint64_t* hs = glb<const uint64_t> (S.J->syn_syms_dev) + L.seg.hoff[0];
    const uint32_t nh = L.seg.hoff[S.n] - L.seg.hoff[0];
    for (uint32_t i0 = 0; i0 < nh; i0 += 256u) {
      uint64_t sy[4];
#pragma unroll
      for (int q = 0; q < 4; q++) { const uint32_t i = i0 + 64u * (uint32_t)q + (uint32_t)lane; sy[q] = i < nh ? hs[i] : 0ull; }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const uint32_t prior = (uint32_t)sy[q], hi = (uint32_t) (sy[q] >> 32);
        const int kind = (int) ((hi >> 16) & 0xffu);
        if (i0 + 64u * (uint32_t)q + (uint32_t)lane >= nh || kind == LH264_SYM_SPLICE) continue;
        account (sym_count (prior, (int) (int16_t) (hi & 0xffffu), kind, (int) (hi >> 24)));
      }
    }
    const GLB uint64_t* cs = glb<const uint64_t> (S.J->ctx_syms_dev) + (size_t)*glb<const unsigned long long> (S.J->ctx_sym_base_dev) + off0;
    for (uint32_t i0 = 0; i0 < nctx; i0 += 256u) {
      uint64_t sy[4];
#pragma unroll
      for (int q = 0; q < 4; q++) { const uint32_t i = i0 + 64u * (uint32_t)q + (uint32_t)lane; sy[q] = i < nctx ? cs[i] : 0ull; }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const uint32_t prior = (uint32_t)sy[q], hi = (uint32_t) (sy[q] >> 32);
        if (i0 + 64u * (uint32_t)q + (uint32_t)lane >= nctx) continue;
        account (sym_count<true> (prior, (int) (int16_t) (hi & 0xffffu), (int) ((hi >> 16) & 0xffu), (int) (hi >> 24)));
      }
    }
  } else
  for (uint32_t s0 = 0; s0 < S.total; s0 += 256u) {
    // in coding order, four symbols per lane and step: their loads are under way together
    uint64_t sy[4];
    seg_symbol4 (L.seg, S, s0, lane, sy);
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const uint64_t sym = sy[q];
      const uint32_t prior = (uint32_t)sym, hi = (uint32_t) (sym >> 32);
      if (s0 + 64u * (uint32_t)q + (uint32_t)lane >= S.total) continue;
      account (sym_count (prior, (int) (int16_t) (hi & 0xffffu), (int) ((hi >> 16) & 0xffu), (int) (hi >> 24)));
    }
  }
  wsync();
  // tags some symbol of the segment brought into existence: the same mask in every lane
  for (int m = 1; m < 64; m <<= 1) touch |= (unsigned long long)__shfl_xor ((long long)touch, m);
  // column sums: lane t adds up column t's 64 halves, every lane starting at a bank of its own
  {
    const int t = lane;
    uint32_t sum = 0;
    if (t < CNT_SLOTS)
      for (int i = 0; i < 32; i++) { const uint32_t v = L.col[t * 32 + ((i + lane) & 31)]; sum += (v & 0xffffu) + (v >> 16); }
    if (t < CNT_SLOTS) seg_cnt[(size_t)seg * LH264_CODER_CNT_STRIDE + t] = sum | (t < LH264_N_TAG_SLOTS ? (uint32_t) ((touch >> t) & 1ull) << 31 : 0u);
  }
  // decisions per bucket of cells
  GLB uint32_t* sb = glb<uint32_t> (seg_bkt) + (size_t)seg * CODER_BUCKETS;
  for (int p0 = 0; p0 < CODER_BUCKETS; p0 += 64) sb[p0 + lane] = L.pcnt[p0 + lane];
}

// ---- kernel 1b: the partitions of a stream = its buckets of cells dealt out evenly ------------------------------------------------------
// One DynProb belongs to one wave of the resolve kernel, and the kernel takes as long as its busiest wave.  The decisions of a stream are
// far from even over the cells (TEST_PROB alone - every raw bit - had 5.5 % of a 1080p stream's decisions, single cells 2-3 %): with the cells
// hashed straight into 16 partitions the largest one held 1.7-1.9 times the mean.  So the cells are hashed into CODER_BUCKETS = 128
// buckets, the stream's decisions per bucket are added up over its segments, and the buckets are dealt to the P partitions largest first,
// each to the partition with the least so far.  chain_map[stream][bucket] = partition.  One wave per stream.
__global__ void __launch_bounds__ (64)
coder_balance_kernel (const uint32_t* __restrict__ seg0, const int32_t* __restrict__ chain_first, const uint32_t* __restrict__ seg_bkt, int n_chains, int log2p,
                      uint8_t* __restrict__ chain_map) {
  static_assert (CODER_BUCKETS == 128, "two buckets per lane");
  const int c = blockIdx.x, lane = threadIdx.x;
  if (c >= n_chains) return;
  GLB uint8_t* map = glb<uint8_t> (chain_map) + (size_t)c * CODER_BUCKETS;
  const int P = 1 << log2p;
  if (P >= CODER_BUCKETS) { map[lane] = (uint8_t)lane; map[lane + 64] = (uint8_t) (lane + 64); return; }
  const size_t m0 = seg0[chain_first[c]], m1 = seg0[chain_first[c + 1]];
  unsigned long long t0 = 0, t1 = 0;
  const GLB uint32_t* sb = glb<const uint32_t> (seg_bkt) + lane;
  for (size_t g0 = m0; g0 < m1; g0 += 8) {                 // (eight segments' rows under way at a time)
    uint32_t a[8], b[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { a[k] = g0 + k < m1 ? sb[(g0 + k) * CODER_BUCKETS] : 0u; b[k] = g0 + k < m1 ? sb[(g0 + k) * CODER_BUCKETS + 64] : 0u; }
#pragma unroll
    for (int k = 0; k < 8; k++) { t0 += a[k]; t1 += b[k]; }
  }
  // largest remaining bucket -> least loaded partition; ties by index (the result depends on the counts alone)
  unsigned long long load = 0;                             // lanes < P: decisions dealt to partition `lane`
  bool done0 = false, done1 = false;
  for (int it = 0; it < CODER_BUCKETS; it++) {
    // (count, 127 - bucket) as one 64-bit key: the maximum is the largest count, the lowest bucket among equals
    unsigned long long k0 = done0 ? 0ull : (t0 << 8 | (unsigned long long) (255 - lane)), k1 = done1 ? 0ull : (t1 << 8 | (unsigned long long) (255 - (lane + 64)));
    unsigned long long best = k0 > k1 ? k0 : k1;
    for (int m = 1; m < 64; m <<= 1) { const unsigned long long o = (unsigned long long)__shfl_xor ((long long)best, m); best = o > best ? o : best; }
    const int bucket = 255 - (int) (best & 0xffull);
    const unsigned long long cnt = best >> 8;
    unsigned long long lk = lane < P ? (load << 8 | (unsigned long long)lane) : ~0ull;
    for (int m = 1; m < 64; m <<= 1) { const unsigned long long o = (unsigned long long)__shfl_xor ((long long)lk, m); lk = o < lk ? o : lk; }
    const int part = (int) (lk & 0xffull);
    if (lane == part) load += cnt;
    if (bucket == lane) { done0 = true; map[lane] = (uint8_t)part; }
    if (bucket == lane + 64) { done1 = true; map[lane + 64] = (uint8_t)part; }
  }
}

// ---- kernel 1c: where each partition's run starts inside a segment's decision words ------------------------------------------------------
// seg_part[segment][p] = decisions of the segment in partitions < p ([P]: all).  One wave per segment.
__global__ void __launch_bounds__ (64 * LH264_CODER_WG_WAVES)
coder_partoff_kernel (const uint32_t* __restrict__ seg0, const uint32_t* __restrict__ seg_job, const uint32_t* __restrict__ job_chain, int n_jobs, int log2p,
                      const uint32_t* __restrict__ seg_bkt, const uint8_t* __restrict__ chain_map, uint32_t* __restrict__ seg_part) {
  __shared__ uint32_t pc[LH264_CODER_WG_WAVES][LH264_CODER_MAX_PARTS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t seg = blockIdx.x * (uint32_t)LH264_CODER_WG_WAVES + (uint32_t)wave;
  if (seg >= seg0[n_jobs]) return;
  LDS uint32_t* pcnt = (LDS uint32_t*) (uintptr_t) (uint32_t) (uintptr_t)&pc[wave][0];
  const int P = 1 << log2p;
  for (int i = lane; i < P; i += 64) pcnt[i] = 0;
  wsync();
  const GLB uint8_t* map = glb<const uint8_t> (chain_map) + (size_t)job_chain[seg_job[seg]] * CODER_BUCKETS;
  const GLB uint32_t* sb = glb<const uint32_t> (seg_bkt) + (size_t)seg * CODER_BUCKETS;
  for (int b = lane; b < CODER_BUCKETS; b += 64) {
    const uint32_t v = sb[b];
    if (v) __hip_atomic_fetch_add (pcnt + map[b], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  wsync();
  GLB uint32_t* sp = glb<uint32_t> (seg_part) + (size_t)seg * (size_t) (P + 1);
  uint32_t carry = 0;
  for (int p0 = 0; p0 < P; p0 += 64) {
    const uint32_t v = p0 + lane < P ? pcnt[p0 + lane] : 0u;
    const uint32_t incl = (uint32_t)wave_scan_add ((int)v);
    if (p0 + lane < P) sp[p0 + lane + 1] = carry + incl;
    carry += (uint32_t)__builtin_amdgcn_readlane ((int)incl, 63);
  }
  if (lane == 0) sp[0] = 0;
}

// ---- kernel 2: per stream, where each segment's decisions start, where its entries of every tag's list start, the size of every list ---
// seg_cnt[segment][t] becomes the number of entries of tag t's list in front of the segment (within the stream)
__global__ void __launch_bounds__ (64)
coder_scan_kernel (const uint32_t* __restrict__ seg0, const int32_t* __restrict__ chain_first, uint32_t* __restrict__ seg_cnt,
                   uint32_t* __restrict__ seg_doff, uint32_t* __restrict__ chain_info, int n_chains) {
  const int c = blockIdx.x, lane = threadIdx.x;
  if (c >= n_chains) return;
  const size_t m0 = seg0[chain_first[c]], m1 = seg0[chain_first[c + 1]];
  uint32_t acc = 0, touched = 0;
  bool big = false;
  const int t = lane <= LH264_N_TAG_SLOTS ? lane : LH264_N_TAG_SLOTS;
  GLB uint32_t* p = glb<uint32_t> (seg_cnt) + t;
  for (size_t g0 = m0; g0 < m1; g0 += 8) {                 // (eight segments' counts under way at a time)
    uint32_t v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = g0 + k < m1 ? p[(g0 + k) * LH264_CODER_CNT_STRIDE] : 0u;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      if (g0 + k >= m1) break;
      if (lane == LH264_N_TAG_SLOTS) { seg_doff[g0 + k] = acc; big = big || acc + v[k] < acc; acc += v[k]; }
      else if (lane < LH264_N_TAG_SLOTS) { p[(g0 + k) * LH264_CODER_CNT_STRIDE] = acc; acc += v[k] & 0x7fffffffu; touched |= v[k] >> 31; }
    }
  }
  uint32_t* I = chain_info + (size_t)c * LH264_CODER_INFO_WORDS;
  // tag lists are padded to 8 entries (16 bytes): the coding kernel reads them 16 bytes at a time
  const uint32_t mine = lane < LH264_N_TAG_SLOTS ? ((acc + 7u) & ~7u) : 0u;
  const uint32_t incl = (uint32_t)wave_scan_add ((int)mine);
  if (lane < LH264_N_TAG_SLOTS) { I[LH264_CODER_INFO_TAGBASE + lane] = incl - mine; I[LH264_CODER_INFO_TAGCNT + lane] = acc; }
  const unsigned long long tm = __ballot (lane < LH264_N_TAG_SLOTS && touched != 0);
  if (lane == LH264_N_TAG_SLOTS) { I[LH264_CODER_INFO_NDEC] = acc; I[LH264_CODER_INFO_TOUCH] = (uint32_t)tm; I[LH264_CODER_INFO_TOUCH + 1] = (uint32_t) (tm >> 32); }
  if (lane == 63) I[LH264_CODER_INFO_NQ] = incl;
  // more than 2^32 decisions in a stream, or more list entries than a decision word can address
  const uint32_t nq = (uint32_t)__builtin_amdgcn_readlane ((int)incl, 63);
  if ((__ballot (big) || nq >= (1u << CODER_QPOS_BITS)) && lane == 0) atomicOr (&I[LH264_CODER_INFO_STATUS], (uint32_t)LH264_CODER_ST_COUNT);
}

// ---- kernel 3: where each stream's decision words and tag lists start (prefix over the streams); the totals for the host -------
__global__ void __launch_bounds__ (CODER_ONE_WG)
coder_bases_kernel (uint32_t* __restrict__ chain_info, int n_chains, unsigned long long* __restrict__ totals) {
  __shared__ unsigned long long wd[16], wq[16];
  __shared__ unsigned long long cd, cq;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) { cd = 0; cq = 0; }
  __syncthreads();
  for (int c0 = 0; c0 < n_chains; c0 += CODER_ONE_WG) {
    const int c = c0 + tid;
    uint32_t* I = chain_info + (size_t) (c < n_chains ? c : 0) * LH264_CODER_INFO_WORDS;
    // a stream's decision words start on a 256-byte line; one spare wave step of words is readable behind them
    const unsigned long long nd = c < n_chains ? (((unsigned long long)I[LH264_CODER_INFO_NDEC] + 63ull) & ~63ull) : 0ull;
    const unsigned long long nq = c < n_chains ? (unsigned long long)I[LH264_CODER_INFO_NQ] : 0ull;
    unsigned long long sd = nd, sq = nq;
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned long long od = __shfl_up (sd, d), oq = __shfl_up (sq, d);
      if (lane >= d) { sd += od; sq += oq; }
    }
    if (lane == 63) { wd[wave] = sd; wq[wave] = sq; }
    __syncthreads();
    unsigned long long bd = cd, bq = cq;
    for (int w = 0; w < wave; w++) { bd += wd[w]; bq += wq[w]; }
    if (c < n_chains) {
      const unsigned long long d0 = bd + sd - nd, q0 = bq + sq - nq;
      I[LH264_CODER_INFO_DBASE] = (uint32_t)d0; I[LH264_CODER_INFO_DBASE + 1] = (uint32_t) (d0 >> 32);
      I[LH264_CODER_INFO_QBASE] = (uint32_t)q0; I[LH264_CODER_INFO_QBASE + 1] = (uint32_t) (q0 >> 32);
    }
    __syncthreads();
    if (tid == CODER_ONE_WG - 1) { cd = bd + sd; cq = bq + sq; }
    __syncthreads();
  }
  if (tid == 0) { totals[0] = cd; totals[1] = cq; }
}

// which lanes of the wave hold the same `nbits`-bit key as this lane (valid lanes only)
template <int NBITS> __device__ __forceinline__ void wave_match (uint32_t key, unsigned long long valid, uint32_t& lo, uint32_t& hi) {
  uint32_t dlo = 0, dhi = 0;
#pragma unroll
  for (int b = 0; b < NBITS; b++) {
    const int xb = __builtin_amdgcn_sbfe ((int)key, b, 1);             // 0 or -1
    const unsigned long long m = __ballot (xb != 0);
    dlo |= (uint32_t)m ^ (uint32_t)xb; dhi |= (uint32_t) (m >> 32) ^ (uint32_t)xb;
  }
  lo = ~dlo & (uint32_t)valid; hi = ~dhi & (uint32_t) (valid >> 32);
}
__device__ __forceinline__ int below (uint32_t lo, uint32_t hi) { return (int)__builtin_amdgcn_mbcnt_hi (hi, __builtin_amdgcn_mbcnt_lo (lo, 0u)); }

// ---- kernel 4: the decision words ------------------------------------------------------------------------------------------------
// One wave per segment, a lane per DECISION: the wave takes 256 symbols of the segment at a time, lays their decision counts out as a
// running sum, and then every lane of a step finds the symbol its decision belongs to (binary search in LDS) and computes the decision
// in closed form (decision_at) - no lane walks a binarisation while its neighbours with shorter symbols idle.  A decision word goes to
// the run of its partition inside the segment's words (stable: coding order inside a partition), and carries the entry of its tag's
// list it will fill; both places are running counts in coding order: lanes with the same partition / tag find each other with ballots
// (wave_match), cursors per partition and per tag live in the wave's LDS.
#define EMIT_BATCH 256
struct EmitLds {
  SegLds seg;
  uint64_t bsym[EMIT_BATCH];         // the batch's symbols (those with decisions)
  uint16_t bS[EMIT_BATCH];           // decisions of the batch in front of symbol i
  uint32_t starts[EMIT_BATCH * 46 / 32 + 2];      // bit d: a symbol's decisions start at decision d of the batch
  uint32_t tcur[LH264_N_TAG_SLOTS];  // next entry of each tag's list (within the stream's lists)
  uint32_t pcur[LH264_CODER_MAX_PARTS];   // next word of each partition's run (within the segment's words)
  uint32_t map[CODER_BUCKETS / 4];        // bucket of cells -> partition (a byte each)
};
__global__ void __launch_bounds__ (64 * LH264_CODER_WG_WAVES)
coder_emit_kernel (const lh264_code_job_t* __restrict__ jobs, const uint32_t* __restrict__ seg0, const uint32_t* __restrict__ seg_job, const uint32_t* __restrict__ job_chain,
                   int n_jobs, int log2p, const uint32_t* __restrict__ seg_doff, const uint32_t* __restrict__ seg_cnt, const uint32_t* __restrict__ seg_part,
                   const uint8_t* __restrict__ chain_map, const uint32_t* __restrict__ chain_info, uint64_t* __restrict__ D) {
  __shared__ EmitLds Lg[LH264_CODER_WG_WAVES];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  LDS EmitLds& L = * (LDS EmitLds*) (uintptr_t) (uint32_t) (uintptr_t)&Lg[wave];
  const uint32_t seg = blockIdx.x * (uint32_t)LH264_CODER_WG_WAVES + (uint32_t)wave;
  Seg S;
  if (!seg_locate (jobs, seg0, seg_job, n_jobs, seg, S)) return;
  seg_layout (L.seg, S, lane);
  const uint32_t* I = chain_info + (size_t)job_chain[S.job] * LH264_CODER_INFO_WORDS;
  GLB uint64_t* Dseg = glb<uint64_t> (D) + ((unsigned long long)I[LH264_CODER_INFO_DBASE] | (unsigned long long)I[LH264_CODER_INFO_DBASE + 1] << 32) + seg_doff[seg];
  if (lane < LH264_N_TAG_SLOTS) L.tcur[lane] = I[LH264_CODER_INFO_TAGBASE + lane] + seg_cnt[(size_t)seg * LH264_CODER_CNT_STRIDE + lane];
  for (int i = lane; i < (1 << log2p); i += 64) L.pcur[i] = seg_part[(size_t)seg * (size_t) ((1 << log2p) + 1) + i];
  if (lane < CODER_BUCKETS / 4) L.map[lane] = ((const uint32_t*)chain_map)[(size_t)job_chain[S.job] * (CODER_BUCKETS / 4) + lane];
  wsync();
  for (uint32_t b0 = 0; b0 < S.total; b0 += EMIT_BATCH) {
    // the batch: symbols b0 .. b0 + 255 (four per lane, their loads under way together); those with decisions are kept, in coding
    // order, with the number of the batch's decisions in front of each, and a bit is set where each one's decisions start
    for (int i = lane; i < EMIT_BATCH * 46 / 32 + 2; i += 64) L.starts[i] = 0;
    uint64_t sy[EMIT_BATCH / 64];
    static_assert (EMIT_BATCH == 256, "seg_symbol4");
    seg_symbol4 (L.seg, S, b0, lane, sy);
    wsync();
    uint32_t run = 0, kept = 0;
#pragma unroll
    for (int q = 0; q < EMIT_BATCH / 64; q++) {
      const uint64_t sym = sy[q];
      const uint32_t hi = (uint32_t) (sym >> 32);
      const int n = b0 + 64u * (uint32_t)q + (uint32_t)lane < S.total ? sym_count ((uint32_t)sym, (int) (int16_t) (hi & 0xffffu), (int) ((hi >> 16) & 0xffu), (int) (hi >> 24)).n : 0;
      const int incl = wave_scan_add (n);
      const unsigned long long km = __ballot (n > 0);
      if (n > 0) {
        const uint32_t k = kept + (uint32_t)below ((uint32_t)km, (uint32_t) (km >> 32)), st = run + (uint32_t) (incl - n);
        L.bsym[k] = sym; L.bS[k] = (uint16_t)st;
        __hip_atomic_fetch_or (&L.starts[st >> 5], 1u << (st & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      kept += (uint32_t)__popcll (km);
      run += (uint32_t)__builtin_amdgcn_readlane (incl, 63);
    }
    wsync();
    const uint32_t T = run;                                 // decisions of the batch (< 2^16: 256 symbols of at most 46)
    uint32_t kbase = 0;                                     // symbols that start in front of the step's first decision
    for (uint32_t d0 = 0; d0 < T; d0 += 64u) {
      const bool valid = d0 + (uint32_t)lane < T;
      // the symbol of this lane's decision: the last one that starts at or before it = a count over the start bits
      const uint32_t mlo = * (volatile LDS uint32_t*)&L.starts[d0 >> 5], mhi = * (volatile LDS uint32_t*)&L.starts[(d0 >> 5) + 1u];
      const uint32_t upto = (uint32_t)__builtin_amdgcn_mbcnt_hi (mhi, __builtin_amdgcn_mbcnt_lo (mlo, 0u)) + (uint32_t) (((lane < 32 ? mlo >> lane : mhi >> (lane - 32)) & 1u));
      const uint32_t lo = min (kbase + upto - 1u, kept - 1u);
      kbase += (uint32_t)__popc (mlo) + (uint32_t)__popc (mhi);
      const uint32_t d = valid ? d0 + (uint32_t)lane : (uint32_t)L.bS[lo];
      const uint64_t sym = L.bsym[lo];
      const uint32_t shi = (uint32_t) (sym >> 32);
      const Decision dc = decision_at ((uint32_t)sym, (int) (int16_t) (shi & 0xffffu), (int) ((shi >> 16) & 0xffu), (int) (shi >> 24), (int) (d - (uint32_t)L.bS[lo]));
      const uint32_t part = ((const LDS uint8_t*)L.map)[cell_part (dc.key, CODER_LOG2_BUCKETS)];
      const unsigned long long vm = __ballot (valid);
      uint32_t tlo, thi, plo, phi;
      wave_match<6> ((uint32_t)dc.tag, vm, tlo, thi);
      if (log2p <= 4) wave_match<4> (part, vm, plo, phi); else wave_match<LH264_CODER_MAX_LOG2P> (part, vm, plo, phi);
      const int trank = below (tlo, thi), tn = __popc (tlo) + __popc (thi), prank = below (plo, phi), pn = __popc (plo) + __popc (phi);
      volatile LDS uint32_t* tc = &L.tcur[dc.tag];
      volatile LDS uint32_t* pc = &L.pcur[part];
      uint32_t tb = 0, pb = 0;
      if (valid) { tb = *tc; pb = *pc; }
      asm volatile ("" ::: "memory");
      if (valid && trank == tn - 1) *tc = tb + (uint32_t)tn;
      if (valid && prank == pn - 1) *pc = pb + (uint32_t)pn;
      if (valid)
        Dseg[pb + (uint32_t)prank] = (uint64_t)dc.key | (uint64_t) ((uint32_t)dc.place | (uint32_t)dc.bit << 4 | ((tb + (uint32_t)trank) & ((1u << CODER_QPOS_BITS) - 1u)) << 5) << 32;
    }
    wsync();
  }
}

// ---- kernel 5: the probability every decision is coded with -----------------------------------------------------------------
// One wave per (stream, partition): the DynProbs of a partition belong to that wave alone, so nothing is shared between waves - no
// ticket, no workgroup barrier (round 2 resolved a stream with one workgroup whose waves took turns on shared counters: 46 % of a
// wave's time was waiting for its turn).  A partition's decision words are the runs the binarisation left in every segment of the
// stream, in order; the wave packs them into rounds of 64 (a round may span runs).  Per round: lanes holding the same DynProb find each
// other with ballots; the probability a decision is coded with follows from the counters before the round plus the PREFIX COUNTS of the
// earlier decisions of the round on the same DynProb; the last lane of a group writes the counters back.  The entry (probability of the
// bit that occurred, bit) goes to the place of the tag's list the word names.
// The counters of a DynProb as kept here: c0 | c1 << 10, NOT yet halved when their sum has reached 513 - the reference computes the next
// probability before it halves (DynProb::update, :101-113), so the probability of the next decision always follows from the stored pair,
// and the halving is done by the next reader.  An entry of the wave's LDS cache (and of the spill table in HBM) is 64 bits: counters in
// bits 0..19, the DynProb's key (cell key << 4 | place, 36 bits) in bits 20..55, bit 63 set.  All zero = free.
#define R2_WAVES LH264_CODER_WG_WAVES
#define R2_SYNC_EVERY 32u        // rounds between two looks at the other waves of the stream
#define R2_SYNC_LOOKS 64         // at most this many looks (with a sleep between them) before going on regardless
#ifndef R2_DMA_MOD
#define R2_DMA_MOD ""          // cache policy of the decision word reads (LDS-DMA)
#endif
#ifndef R2_LOG2_BUCKETS
#define R2_LOG2_BUCKETS 8
#endif
#define R2_SLOTS (4 << R2_LOG2_BUCKETS)        // DynProbs in a wave's LDS cache: 1024
// Decision words are requested R2_DEPTH + 3 rounds before they are resolved
#ifndef R2_DEPTH
#define R2_DEPTH 3      // (measured: 6 slots cost a workgroup's worth of LDS per CU and bought nothing - 36 vs 29 ms on 512 x 4 720p pictures)
#endif
#define R2_CHECK 4               // the fill of the cache is looked at every R2_CHECK rounds (that many rounds insert <= 256)
#ifndef R2_FLUSH
#define R2_FLUSH 600             // everything goes to the spill table and the cache starts over above this many
#endif
// diagnostic build (-DLH264_CODER_DEBUG): shader-clock stamps of the resolve kernel's phases, summed over all waves (never in product builds)
#ifdef LH264_CODER_DEBUG
__device__ unsigned long long g_rs_stamps[16];
#define RS_STAMP_DECL unsigned long long st_t = __builtin_amdgcn_s_memtime(), st_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define RS_STAMP(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_t; st_t = t_; }
#define RS_STAMP_FLUSH if (lane == 0) for (int q_ = 0; q_ < 10; q_++) atomicAdd (&g_rs_stamps[q_], st_acc[q_]);
#else
#define RS_STAMP_DECL
#define RS_STAMP(i)
#define RS_STAMP_FLUSH
#endif
struct ResolveLds {
  unsigned long long ent[R2_SLOTS];
  uint32_t ring[R2_DEPTH][2][64];   // the decision words of R2_DEPTH future rounds (low dwords, high dwords), filled by LDS-DMA
  uint32_t scratch[64];
  uint32_t nres, pad[3];
};
// An entry of a wave's LDS cache (and of the spill table): low dword = the key of the DynProb's cell, high dword = 0x10 | place in
// bits 27..31 (never zero for an entry in use), the counters in bits 0..19.  All zero = free.
typedef unsigned long long u64;
#define RS_KEYMASK 0xf8000000ffffffffull
__device__ __forceinline__ u64 ent_make (uint32_t klo, uint32_t place, uint32_t st) { return (u64)klo | (u64) ((0x10u | place) << 27 | st) << 32; }
__device__ __forceinline__ uint32_t rs_hash (uint32_t klo, uint32_t place) { return (klo + place * 0x61C88647u) * 0x9E3779B1u; }
__device__ __forceinline__ uint32_t ent_hash (u64 e) { return rs_hash ((uint32_t)e, (uint32_t) (e >> 59) & 15u); }

// the spill table of a partition: open addressing over its share of the stream's `hash_cells_dev` memory (zero-filled by the caller),
// entries as above.  Only this wave touches it (two of its lanes may want the same free slot: compare-and-swap); its accesses go to L2
// (agent scope), never through this CU's L1.
__device__ __forceinline__ bool spill_put (GLB u64* T, uint32_t tmask, u64 e, uint32_t skip) {
  uint32_t h = (ent_hash (e) >> 8) + skip;
  for (uint32_t tries = 0; tries <= tmask; tries++, h++) {
    GLB u64* p = T + (h & tmask);
    u64 cur = __hip_atomic_load (p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == 0ull) { u64 expect = 0ull; if (__hip_atomic_compare_exchange_strong (p, &expect, e, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return true; cur = expect; }
    if (((cur ^ e) & RS_KEYMASK) == 0ull) { __hip_atomic_store (p, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return true; }
    if (tries >= 4096u) break;
  }
  return false;                  // the table is (as good as) full
}
__device__ __forceinline__ uint32_t spill_get (const GLB u64* T, uint32_t tmask, u64 key, u64 first) {
  uint32_t h = ent_hash (key) >> 8;
  u64 cur = first;                                  // the entry at the key's home slot, requested a round ago
  for (uint32_t tries = 0; tries <= tmask; tries++) {
    if (cur == 0ull) return 0u;
    if (((cur ^ key) & RS_KEYMASK) == 0ull) return (uint32_t) (cur >> 32) & 0xfffffu;
    h++;
    cur = __hip_atomic_load (T + (h & tmask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return 0u;
}

struct SlotRef { int idx; bool miss, inserted; u64 first;
#ifdef LH264_CODER_DEBUG
  int iters;
#endif
};
// the cache entry of the DynProb of decision word (lo, hi): inserted with fresh counters if absent; if counters may have been spilled,
// the inserting lane asks the spill table (answer taken by rs_land)
__device__ __forceinline__ void rs_lookup (LDS ResolveLds& S, const GLB u64* T, uint32_t tmask, uint32_t lo, uint32_t hi, bool valid, bool spilled, SlotRef& R) {
#ifdef LH264_CODER_DEBUG
  R.iters = 0;
#endif
  R.idx = 0; R.miss = false; R.inserted = false; R.first = 0ull;
  if (valid) {
    const uint32_t place = hi & 15u, kt = (0x10u | place) << 27;
    const u64 fresh = ent_make (lo, place, 0u);
    // buckets of four entries (32 bytes, read at once): a probe looks at a whole bucket, so the longest probe sequence among the 64
    // lanes of a wave - which is what the wave waits for - stays short
    const uint32_t hh = rs_hash (lo, place);
    uint32_t bkt = hh >> (32 - R2_LOG2_BUCKETS), h = 0;
    for (int tries = 0; tries < R2_SLOTS / 4; tries++, bkt++) {      // (the flush policy keeps the cache at most 7/8 full: bounded anyway)
      bkt &= R2_SLOTS / 4 - 1;
#ifdef LH264_CODER_DEBUG
      R.iters++;
#endif
      const LDS u32x4* bp = (const LDS u32x4*)&S.ent[4u * bkt];
      const u32x4 a = * (volatile const LDS u32x4*)bp, b = * (volatile const LDS u32x4*) (bp + 1);
      const bool m0 = a.x == lo && (a.y & 0xf8000000u) == kt, m1 = a.z == lo && (a.w & 0xf8000000u) == kt,
                 m2 = b.x == lo && (b.y & 0xf8000000u) == kt, m3 = b.z == lo && (b.w & 0xf8000000u) == kt;
      if (m0 || m1 || m2 || m3) { h = 4u * bkt + (m0 ? 0u : m1 ? 1u : m2 ? 2u : 3u); break; }
      const int empty = a.y == 0u ? 0 : a.w == 0u ? 1 : b.y == 0u ? 2 : b.w == 0u ? 3 : -1;
      if (empty >= 0) {
        // take the first free entry of the bucket; if another lane is quicker, look at the bucket again (it may have put this very key there)
        u64 expect = 0ull;
        if (__hip_atomic_compare_exchange_strong (&S.ent[4u * bkt + (uint32_t)empty], &expect, fresh, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
          h = 4u * bkt + (uint32_t)empty; R.miss = true; break;
        }
        bkt--;
      }
    }
    R.idx = (int)h;
    R.inserted = R.miss;
    R.miss = R.miss && spilled;                                // before the first flush a new DynProb is simply fresh
    // (the spill table is asked in rs_land, a round later, and waited for there: a load the compiler sees under way ACROSS iterations
    // makes it drain every request under way - "s_waitcnt vmcnt(0)" - in every iteration, needed or not)
  }
}
// the counters the spill table holds for an entry inserted by rs_lookup go into the entry (it has not been used yet)
__device__ __forceinline__ void rs_land (LDS ResolveLds& S, const GLB u64* T, uint32_t tmask, uint32_t lo, uint32_t hi, SlotRef& R) {
  if (R.miss) {
    const u64 key = ent_make (lo, hi & 15u, 0u);
    const uint32_t st = spill_get (T, tmask, key, __hip_atomic_load (T + ((ent_hash (key) >> 8) & tmask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    volatile LDS uint32_t* p = (volatile LDS uint32_t*)&S.ent[R.idx] + 1;
    if (st) *p = (*p & 0xfff00000u) | st;
    R.miss = false;
  }
}
// every DynProb of the cache to the spill table, the cache cleared.  Four entries per lane at a time: their home slots are read
// together and taken together (one memory round trip each instead of one per entry); what finds its home slot taken by another key
// goes the slow way.
__device__ __forceinline__ bool rs_flush (LDS ResolveLds& S, GLB u64* T, uint32_t tmask, int lane) {
  bool ok = true;
  for (int u0 = 0; u0 < R2_SLOTS / 64; u0 += 4) {
    u64 e[4], cur[4]; GLB u64* hp[4]; bool took[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { e[u] = S.ent[lane + 64 * (u0 + u)]; hp[u] = T + ((ent_hash (e[u]) >> 8) & tmask); }
#pragma unroll
    for (int u = 0; u < 4; u++) cur[u] = e[u] ? __hip_atomic_load (hp[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ~0ull;
#pragma unroll
    for (int u = 0; u < 4; u++) {
      took[u] = false;
      if (cur[u] == 0ull) { u64 expect = 0ull; took[u] = __hip_atomic_compare_exchange_strong (hp[u], &expect, e[u], __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); cur[u] = expect; }
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (e[u] && !took[u]) {
        if (((cur[u] ^ e[u]) & RS_KEYMASK) == 0ull) __hip_atomic_store (hp[u], e[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else ok = spill_put (T, tmask, e[u], 1u) && ok;
      }
      S.ent[lane + 64 * (u0 + u)] = 0ull;
    }
  }
  return ok;
}

// the rounds of a partition: its runs of decision words in the segments [g, g_end) of the stream, packed 64 to a round.  The run of
// the next segment is looked up (scalar loads from the segment tables) when the current one is opened, a run ahead of its use.
struct RoundGen {
  uint32_t g, g_end, part, pstride;
  u64 dbase, cur; uint32_t rem;
  uint32_t nx_o0, nx_o1, nx_doff;                         // segment g's table entries, requested ahead
  __device__ __forceinline__ void request (const uint32_t* __restrict__ seg_doff, const uint32_t* __restrict__ seg_part) {
    const uint32_t gg = g < g_end ? g : g_end - 1u;         // (always a valid segment: the stream has one when it has a round)
    nx_o0 = seg_part[(size_t)gg * pstride + part]; nx_o1 = seg_part[(size_t)gg * pstride + part + 1u]; nx_doff = seg_doff[gg];
  }
  // the words of the next round: index of this lane's word (any valid index for a lane beyond the count), the count (0: no more rounds)
  __device__ __forceinline__ uint32_t next (const uint32_t* __restrict__ seg_doff, const uint32_t* __restrict__ seg_part, int lane, u64& idx) {
    uint32_t c = min (rem, 64u);
    idx = cur + (uint32_t)lane;
    cur += c; rem -= c;
    while (c < 64u && g < g_end) {
      const uint32_t len = nx_o1 - nx_o0;
      const u64 b = dbase + nx_doff + nx_o0;
      g++;
      request (seg_doff, seg_part);
      if (len == 0u) continue;
      const uint32_t take = min (len, 64u - c);
      if ((uint32_t)lane >= c) idx = b + ((uint32_t)lane - c);
      cur = b + take; rem = len - take; c += take;
    }
    if ((uint32_t)lane >= c) idx = dbase;
    return c;
  }
};

__global__ void __launch_bounds__ (R2_WAVES * 64)
coder_resolve_kernel (const lh264_code_stream_t* __restrict__ streams, uint32_t* __restrict__ chain_info, const uint32_t* __restrict__ seg0,
                      const int32_t* __restrict__ chain_first, const uint32_t* __restrict__ seg_doff, const uint32_t* __restrict__ seg_part,
                      const uint64_t* __restrict__ D, uint16_t* __restrict__ Q, int n_chains, int log2p, uint32_t* __restrict__ progress, int window) {
  __shared__ ResolveLds Sg[R2_WAVES];
  const int lane = threadIdx.x & 63, wave = uniform ((int) (threadIdx.x >> 6));
  LDS ResolveLds& S = * (LDS ResolveLds*) (uintptr_t) (uint32_t) (uintptr_t)&Sg[wave];
  // (stream, partition) of this wave.  Workgroups go to the XCDs in turn (blockIdx.x mod 8), and every XCD has its own L2: the waves of
  // a stream all write the stream's tag lists, 2-byte entries that only become whole sectors when the entries of ALL partitions have
  // arrived - so a stream's workgroups are given block indices of one residue mod 8 and meet in one L2
#ifndef R2_PLAIN_MAP
  const uint32_t wgs = (1u << log2p) >= (uint32_t)R2_WAVES ? (1u << log2p) / R2_WAVES : 1u;      // workgroups per stream
  const uint32_t xj = blockIdx.x >> 3;
  const int chain = (int) ((xj / wgs) * 8u + (blockIdx.x & 7u));
  const uint32_t part = (xj % wgs) * R2_WAVES + (uint32_t)wave;
  if (chain >= n_chains || part >= (1u << log2p)) return;
#else
  const uint32_t wid = blockIdx.x * R2_WAVES + (uint32_t)wave;
  const int chain = (int) (wid >> log2p);
  const uint32_t part = wid & ((1u << log2p) - 1u);
  if (chain >= n_chains) return;
#endif
  uint32_t* I = chain_info + (size_t)chain * LH264_CODER_INFO_WORDS;
  // the partition's share of the stream's spill table (hash_cap cells of 64 bytes = 8 entries each)
  const uint32_t hc = streams[chain].hash_cap;
  if (hc == 0u || (hc & (hc - 1u)) != 0u || hc > (1u << 20) || ((hc * 8u) >> log2p) < 64u) {
    if (lane == 0) atomicOr (&I[LH264_CODER_INFO_STATUS], (uint32_t)LH264_CODER_ST_TABLE_FULL);
    return;
  }
  const uint32_t tsize = (hc * 8u) >> log2p, tmask = tsize - 1u;
  GLB u64* T = glb<u64> (streams[chain].hash_cells_dev) + (size_t)part * tsize;
  GLB uint16_t* Qc = glb<uint16_t> (Q) + ((unsigned long long)I[LH264_CODER_INFO_QBASE] | (unsigned long long)I[LH264_CODER_INFO_QBASE + 1] << 32);
  const GLB uint64_t* Dg = glb<const uint64_t> (D);
  for (int i = lane; i < R2_SLOTS; i += 64) S.ent[i] = 0ull;
  if (lane == 0) S.nres = 0;
  RoundGen G;
  G.g = (uint32_t)uniform ((int)seg0[chain_first[chain]]); G.g_end = (uint32_t)uniform ((int)seg0[chain_first[chain + 1]]); G.part = part; G.pstride = (1u << log2p) + 1u;
  G.dbase = (u64)I[LH264_CODER_INFO_DBASE] | (u64)I[LH264_CODER_INFO_DBASE + 1] << 32; G.cur = G.dbase; G.rem = 0;
  if (G.g >= G.g_end) return;
  G.request (seg_doff, seg_part);
  wsync();
  // Decision words travel HBM -> LDS by LDS-DMA, three rounds ahead of their use, so that neither the compiler's nor this code's
  // waits for OTHER memory operations ever have to wait for a word that was only just requested.  (Always issued - a lane without a
  // word reads the stream's first one - so that the counted wait below is right in the last rounds too.)
  const uint32_t my_ring = (uint32_t) (uintptr_t)&S.ring[0][0][0];
  auto dma = [&] (u64 idx, uint32_t slot) {
    const GLB uint32_t* src = (const GLB uint32_t*) (Dg + idx);
    // (as asm statements: hipcc would make every later LDS read wait for a load it knows to write LDS; M0 = the LDS address of lane 0's
    // dword, saved and restored inside the statement)
    const uint32_t dst = (uint32_t)uniform ((int) (my_ring + slot * 512u));
    uint32_t keep;
    asm volatile ("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" R2_DMA_MOD "\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dword %2, off" R2_DMA_MOD "\n\ts_mov_b32 m0, %0"
                  : "=&s"(keep) : "v"(src), "v"(src + 1), "s"(dst), "s"(dst + 256u) : "memory");
  };
  // rounds it (resolved in this iteration), it + 1 (looked up; a spilled answer is taken in this iteration), it + 2 (looked up in this
  // iteration), it + 3 .. it + 2 + R2_DEPTH (under way into the ring)
  u64 x0, x1, x2, xr;
  uint32_t n0 = G.next (seg_doff, seg_part, lane, x0), n1 = G.next (seg_doff, seg_part, lane, x1), n2 = G.next (seg_doff, seg_part, lane, x2);
  if (n0 == 0u) return;
  uint64_t w0 = Dg[x0], w1 = Dg[x1], w2 = Dg[x2];
  uint32_t nr[R2_DEPTH];                                   // words of the rounds in the ring (slot = round % R2_DEPTH)
#pragma unroll
  for (int k = 0; k < R2_DEPTH; k++) { nr[(3 + k) % R2_DEPTH] = G.next (seg_doff, seg_part, lane, xr); dma (xr, (uint32_t) ((3 + k) % R2_DEPTH)); }
  // (once: the counted wait in the loop relies on what each iteration issues.  As the builtin, so that the compiler knows the three
  // words loaded above have arrived: otherwise it puts "s_waitcnt vmcnt(0)" in front of their first use INSIDE the loop - the loop
  // carries them - and every round drains the requests under way, i.e. waits for memory once per round)
  __builtin_amdgcn_s_waitcnt (0x0F70);                     // vmcnt(0), gfx9 encoding
  bool spilled = false;
  SlotRef e0, e1;
  rs_lookup (S, T, tmask, (uint32_t)w0, (uint32_t) (w0 >> 32), (uint32_t)lane < n0, false, e0);
  rs_lookup (S, T, tmask, (uint32_t)w1, (uint32_t) (w1 >> 32), (uint32_t)lane < n1, false, e1);
  uint32_t nres = (uint32_t)__popcll (__ballot (e0.inserted)) + (uint32_t)__popcll (__ballot (e1.inserted));
  bool pend_ok = false; uint32_t pend_q = 0, pend_v = 0;
  // The waves of a stream keep near one another (a hint, never waited for beyond a bounded number of looks): a sector of a tag list is
  // complete when the entries of ALL partitions have been written, and only sectors that complete while they are still in the L2 leave
  // it whole - waves that drift apart by more than the L2 holds write every 2-byte entry to HBM as a partial sector of its own.
  // progress[stream][partition] = the segment the wave has reached (0xffffffff: done).
  GLB uint32_t* prog = (progress && window > 0 && (1 << log2p) <= 64) ? glb<uint32_t> (progress) + ((size_t)chain << log2p) : (GLB uint32_t*)0;
  RS_STAMP_DECL
  for (uint32_t it = 0; n0 != 0u; it++) {
    const bool v_cur = (uint32_t)lane < n0;
    if (prog && (it & (R2_SYNC_EVERY - 1u)) == 0u) {
      if (lane == 0) __hip_atomic_store (prog + part, G.g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int look = 0; look < R2_SYNC_LOOKS; look++) {
        uint32_t v = lane < (1 << log2p) ? __hip_atomic_load (prog + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffu;
        for (int m = 1; m < 64; m <<= 1) v = min (v, (uint32_t)__shfl_xor ((int)v, m));
        if (G.g <= v + (uint32_t)window) break;
        __builtin_amdgcn_s_sleep (64);
      }
      asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (pend_ok) Qc[pend_q] = (uint16_t)pend_v;          // the list entry of the round resolved in the last iteration
    pend_ok = false;
    const uint32_t w_hi = (uint32_t) (w0 >> 32);
    // the words of round it + 3: requested R2_DEPTH iterations ago.  Every iteration since has issued a store (the list entry) and two
    // requests, this one a store: that many younger operations may still be under way
    asm volatile ("s_waitcnt vmcnt(%0)" :: "n" (3 * (R2_DEPTH - 1) + 1) : "memory");
    const uint32_t slot = (it + 3u) % R2_DEPTH;
    const uint64_t w3 = (uint64_t) * (volatile LDS uint32_t*) (uintptr_t) (my_ring + slot * 512u + (uint32_t)lane * 4u) |
                        (uint64_t) * (volatile LDS uint32_t*) (uintptr_t) (my_ring + slot * 512u + 256u + (uint32_t)lane * 4u) << 32;
    uint32_t n3 = 0;
#pragma unroll
    for (int k = 0; k < R2_DEPTH; k++) if (slot == (uint32_t)k) n3 = nr[k];
    RS_STAMP (0)
    // ---- the entries of the round two ahead, first thing: spilled counters that are requested here are taken a whole round later
    // (inserting does not disturb the rounds in flight: they use entries they found earlier) ----------------------------------------------
    SlotRef e2;
    rs_lookup (S, T, tmask, (uint32_t)w2, (uint32_t) (w2 >> 32), (uint32_t)lane < n2, spilled, e2);
    nres += (uint32_t)__popcll (__ballot (e2.inserted));
    RS_STAMP (1)
#ifdef LH264_CODER_DEBUG
    { int mx = e2.iters; for (int m = 1; m < 64; m <<= 1) mx = max (mx, __shfl_xor (mx, m)); st_acc[9] += (unsigned long long)mx; }
#endif
    // ---- who shares my DynProb -----------------------------------------------------------------------------------------------------
    const int bit = (int) ((w_hi >> 4) & 1u);
    const unsigned long long valid = __ballot (v_cur);
    uint32_t slo, shi;
    wave_match<R2_LOG2_BUCKETS + 2> ((uint32_t)e0.idx, valid, slo, shi);
    const unsigned long long zm = __ballot (v_cur && bit == 0);
    const int rank = below (slo, shi), nn = __popc (slo) + __popc (shi);
    const int z = below (slo & (uint32_t)zm, shi & (uint32_t) (zm >> 32));
    const int head = slo ? __ffs ((int)slo) - 1 : 32 + __ffs ((int)shi) - 1;
    // spilled counters requested a round ago: into the entries now
    RS_STAMP (2)
    rs_land (S, T, tmask, (uint32_t)w1, (uint32_t) (w1 >> 32), e1);
    RS_STAMP (3)
    // ---- counters in, counters out -------------------------------------------------------------------------------------------------
    {
      volatile LDS uint32_t* sp = (volatile LDS uint32_t*)&S.ent[e0.idx] + 1;     // (the entry's high dword: place and counters)
      uint32_t st = 0;
      if (v_cur) st = *sp;
      asm volatile ("" ::: "memory");
      const uint32_t c0 = st & 1023u, c1 = (st >> 10) & 1023u;
      const bool lazy = c0 + c1 > 512u;                                     // the halving the last decision left to its successor
      const uint32_t f0 = lazy ? (c0 + 1u) >> 1 : c0, f1 = lazy ? (c1 + 1u) >> 1 : c1;
      // counters before this lane's decision: the group's earlier zeros and ones on top of the stored ones
      uint32_t b0 = f0 + (uint32_t)z, b1 = f1 + (uint32_t) (rank - z);
      uint32_t a0 = rank == 0 ? c0 : b0, a1 = rank == 0 ? c1 : b1;         // what the probability is computed from
      const int t = 512 - (int) (f0 + f1);          // the decision of this rank brings the sum to 513: halved before the decision after the next
      if (__ballot (v_cur && nn > t + 1)) {
        // a halving inside the group: rank t + 1 is still coded from the pair as it stands after rank t, but counts on from the halved
        // pair, as do the ranks behind it
        LDS uint32_t* sc = S.scratch;
        if (v_cur && rank == t + 1) sc[head] = (uint32_t)z;                 // zeros among ranks 0..t
        wsync();
        if (v_cur && rank > t) {
          const uint32_t zt = * (volatile LDS uint32_t*)&sc[head];
          const uint32_t h0 = (f0 + zt + 1u) >> 1, h1 = (f1 + (uint32_t) (t + 1) - zt + 1u) >> 1;
          b0 = h0 + ((uint32_t)z - zt); b1 = h1 + ((uint32_t) (rank - z) - ((uint32_t) (t + 1) - zt));
          if (rank > t + 1) { a0 = b0; a1 = b1; }
        }
        wsync();
      }
      if (v_cur && rank == nn - 1) *sp = (st & 0xfff00000u) | (b0 + (uint32_t) (bit ^ 1)) | (b1 + (uint32_t)bit) << 10;
      asm volatile ("" ::: "memory");
      // the list entry carries the probability of the bit that occurred (what the bool coder multiplies with, see code_step); stored at
      // the top of the next iteration (a store as the youngest memory operation at the loop's end would make the compiler's wait for
      // the spill-table answers wait for the store as well)
      const uint32_t prob = dp_ratio (a0, a1);
      pend_ok = v_cur; pend_q = w_hi >> 5; pend_v = (bit ? 256u - prob : prob) << 1 | (uint32_t)bit;
    }
    RS_STAMP (4)
    if ((it % R2_CHECK) == R2_CHECK - 1 && nres > R2_FLUSH) {
      // every DynProb to the spill table, then the cache starts over with the entries of the two rounds in flight
      rs_land (S, T, tmask, (uint32_t)w1, (uint32_t) (w1 >> 32), e1);
      rs_land (S, T, tmask, (uint32_t)w2, (uint32_t) (w2 >> 32), e2);
      wsync();
      if (!rs_flush (S, T, tmask, lane)) atomicOr (&I[LH264_CODER_INFO_STATUS], (uint32_t)LH264_CODER_ST_TABLE_FULL);
      asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
      wsync();
#ifdef LH264_CODER_DEBUG
      st_acc[8] += 1;
#endif
      spilled = true;
      rs_lookup (S, T, tmask, (uint32_t)w1, (uint32_t) (w1 >> 32), (uint32_t)lane < n1, true, e1);
      rs_lookup (S, T, tmask, (uint32_t)w2, (uint32_t) (w2 >> 32), (uint32_t)lane < n2, true, e2);
      nres = (uint32_t)__popcll (__ballot (e1.inserted)) + (uint32_t)__popcll (__ballot (e2.inserted));
      rs_land (S, T, tmask, (uint32_t)w1, (uint32_t) (w1 >> 32), e1);
      rs_land (S, T, tmask, (uint32_t)w2, (uint32_t) (w2 >> 32), e2);
      wsync();
    }
    RS_STAMP (5)
    asm volatile ("s_waitcnt lgkmcnt(0)" ::: "memory");      // the ring slot has been read: it takes the words of round it + 3 + R2_DEPTH
    const uint32_t nn6 = G.next (seg_doff, seg_part, lane, xr);
    dma (xr, slot);
#pragma unroll
    for (int k = 0; k < R2_DEPTH; k++) if (slot == (uint32_t)k) nr[k] = nn6;
    w0 = w1; w1 = w2; w2 = w3;
    n0 = n1; n1 = n2; n2 = n3;
    e0 = e1; e1 = e2;
    RS_STAMP (6)
#ifdef LH264_CODER_DEBUG
    st_acc[7] += 1;
#endif
  }
  if (pend_ok) Qc[pend_q] = (uint16_t)pend_v;
  if (prog && lane == 0) __hip_atomic_store (prog + part, 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  RS_STAMP_FLUSH
}
#ifdef LH264_CODER_DEBUG
void read_rs_stamps (unsigned long long* out, bool reset) {
  (void)hipMemcpyFromSymbol (out, HIP_SYMBOL (g_rs_stamps), sizeof (g_rs_stamps));
  if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol (HIP_SYMBOL (g_rs_stamps), z, sizeof (z)); }
}
#endif

// ---- kernels 6..9: the libvpx bool coder (vpx_writer, bitwriter.h:35-105; vpx_stop_encode bitwriter.cpp:17-37) ----------------
// vpx_write keeps (low, range, count).  `range` depends only on the decisions so far - a 7-bit state that does not forget where it
// started (two start values stay apart for hundreds of decisions: measured), so its recurrence has to be walked serially - per
// coarse chunk of a list, from checked candidate start states (kernel 7 below); but that walk is all that is serial.  What vpx_write
// finally writes is one big number: every decision with bit 1 adds its `split` (8 bits) at the bit position given by the shifts
// before it, and carries run towards the first byte.  So:
//   coder_chunkmap_kernel  chunks (CODE_CHUNK decisions) and coarse chunks per (stream, tag) pair, running sums
//   coder_range_*_kernel   the range recurrence alone (no low, no bytes) - mad, two shifts, count-leading-zeros, shift per decision -,
//                          noting range and bit position at the start of every chunk
//   coder_accum_kernel     one lane per chunk: the recurrence again from the noted state; the addends go into 32-bit sums per output
//                          byte position (a window in registers, collected in the wave's LDS, added to memory at the end)
//   coder_bytes_kernel     one wave per pair: carries from the last byte to the first, the bytes, vpx_stop_encode's padding byte
// The 32 "stop" decisions (bit 0, probability 128) are decisions n .. n+31 of a list.
// A list entry is e = q << 1 | bit with q = the probability of the bit that occurred, in 1/256 (bit 0: the decision's probability p,
// bit 1: 256 - p; written so by the resolve kernel).  With x = range - 1:  bit 0: what is left is split = (x p + 256) >> 8;  bit 1:
// range - split = x - (x p >> 8) = (x (256 - p) + 255) >> 8.  One multiply-add for both: rq = range q + (256 - bit - q) < 2^16,
// r = rq >> 8, and the normalising shift (vpx_norm[r]) is the number of leading zeros of rq << 16.
#define CODE_CHUNK ((uint32_t)LH264_CODER_CODE_CHUNK)
#define CODE_STOP_ENTRY (128u << 1)
#define CODE_COARSE ((uint32_t)LH264_CODER_CODE_COARSE)      // decisions per coarse chunk of the range walk (a multiple of CODE_CHUNK)
struct CodeStep { uint32_t add, shift; };
__device__ __forceinline__ CodeStep code_step (uint32_t& range, uint32_t e) {
  const uint32_t q = (e >> 1) & 0x1ffu;
  const uint32_t mask = 0u - (e & 1u);
  const uint32_t k = 256u + mask - q;
  uint32_t rq;                                             // one instruction on the serial chain (the compiler would fold k into it as two)
  asm ("v_mad_u32_u24 %0, %1, %2, %3" : "=v" (rq) : "v" (range), "v" (q), "v" (k));
  const uint32_t r = rq >> 8;
  CodeStep s;
  s.add = (range - r) & mask;
  s.shift = (uint32_t)__builtin_clz (rq << 16);            // 1 <= r <= 255
  range = r << s.shift;
  return s;
}
// entry i of a tag's list (16-bit entries behind `src`), the stop decisions behind the n real ones
struct ListReader {
  const GLB u32x4* src; uint32_t n; u32x4 cur; uint32_t have;       // `cur` holds entries have .. have + 7
  __device__ __forceinline__ void init (const GLB uint16_t* list, uint32_t n_) { src = (const GLB u32x4*)list; n = n_; have = 0xffffffffu; }
  __device__ __forceinline__ uint32_t at (uint32_t i) {
    if (i >= n) return CODE_STOP_ENTRY;
    if ((i & ~7u) != have) { have = i & ~7u; cur = src[i >> 3]; }
    const uint32_t w = (i & 4u) ? ((i & 2u) ? cur.w : cur.z) : ((i & 2u) ? cur.y : cur.x);
    return (w >> (16u * (i & 1u))) & 0xffffu;
  }
};
struct PairInfo { const GLB uint16_t* list; uint32_t n, total; bool used; unsigned long long acc0; };
__device__ __forceinline__ PairInfo pair_info (const uint32_t* chain_info, const uint16_t* Q, uint32_t pair) {
  const uint32_t chain = pair / LH264_N_TAG_SLOTS, slot = pair % LH264_N_TAG_SLOTS;
  const uint32_t* I = chain_info + (size_t)chain * LH264_CODER_INFO_WORDS;
  PairInfo P;
  P.n = slot < 35u ? I[LH264_CODER_INFO_TAGCNT + slot] : 0u;
  const unsigned long long tm = (unsigned long long)I[LH264_CODER_INFO_TOUCH] | (unsigned long long)I[LH264_CODER_INFO_TOUCH + 1] << 32;
  P.used = slot < 35u && (P.n > 0u || ((tm >> slot) & 1ull));
  P.total = P.used ? P.n + 32u : 0u;
  const unsigned long long q0 = ((unsigned long long)I[LH264_CODER_INFO_QBASE] | (unsigned long long)I[LH264_CODER_INFO_QBASE + 1] << 32) + I[LH264_CODER_INFO_TAGBASE + (slot < 35u ? slot : 0u)];
  P.list = glb<const uint16_t> (Q) + q0;
  P.acc0 = q0 + 48ull * pair;                        // the pair's sums: fewer than n + 40 positions (a decision shifts out 7 bits at most)
  return P;
}

// kernel 6: chunks per (stream, tag) pair and their running sum (one workgroup)
__global__ void __launch_bounds__ (CODER_ONE_WG)
coder_chunkmap_kernel (const uint32_t* __restrict__ chain_info, int n_pairs, uint32_t* __restrict__ pair_chunk0, uint32_t* __restrict__ pair_coarse0, uint32_t* __restrict__ cand_list) {
  if (threadIdx.x == 0) cand_list[0] = 0;                  // the walks from candidate start states the seed kernel will ask for
  __shared__ uint32_t wsum[16], wsum2[16];
  __shared__ uint32_t carry, carry2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) { carry = 0; carry2 = 0; }
  __syncthreads();
  for (int p0 = 0; p0 < n_pairs; p0 += CODER_ONE_WG) {
    const int p = p0 + tid;
    uint32_t v = 0, v2 = 0;
    if (p < n_pairs) {
      const uint32_t chain = (uint32_t)p / LH264_N_TAG_SLOTS, slot = (uint32_t)p % LH264_N_TAG_SLOTS;
      const uint32_t* I = chain_info + (size_t)chain * LH264_CODER_INFO_WORDS;
      const uint32_t n = slot < 35u ? I[LH264_CODER_INFO_TAGCNT + slot] : 0u;
      const unsigned long long tm = (unsigned long long)I[LH264_CODER_INFO_TOUCH] | (unsigned long long)I[LH264_CODER_INFO_TOUCH + 1] << 32;
      if (slot < 35u && (n > 0u || ((tm >> slot) & 1ull))) { v = (n + 32u + CODE_CHUNK - 1u) / CODE_CHUNK; v2 = (n + 32u + CODE_COARSE - 1u) / CODE_COARSE; }
    }
    const uint32_t incl = (uint32_t)wave_scan_add ((int)v), incl2 = (uint32_t)wave_scan_add ((int)v2);
    if (lane == 63) { wsum[wave] = incl; wsum2[wave] = incl2; }
    __syncthreads();
    uint32_t before = carry, before2 = carry2;
    for (int w = 0; w < wave; w++) { before += wsum[w]; before2 += wsum2[w]; }
    if (p < n_pairs) { pair_chunk0[p] = before + incl - v; pair_coarse0[p] = before2 + incl2 - v2; }
    __syncthreads();
    if (tid == CODER_ONE_WG - 1) { carry = before + incl; carry2 = before2 + incl2; }
    __syncthreads();
  }
  if (tid == 0) { pair_chunk0[n_pairs] = carry; pair_coarse0[n_pairs] = carry2; }
}

// kernel 7: the range recurrence, cut into coarse chunks of CODE_COARSE decisions so that a long list is walked by many lanes.
// `range` is a 7-bit state (128..255 after every decision) that does not forget its start - two start values stay apart for thousands
// of decisions - but every decision that shifts bits out merges states, and after 1,024 decisions at most 4 of the 128 possible states
// are left in 90 % of all cases, at most 8 in 96 % (measured on the bench streams, tools/range_probe.py; the rest are lists that
// only rotate the states, e.g. a constant run at probability 255).  So:
//   coder_range_seed_kernel   one wave per coarse chunk: walks the CODE_LOOKBACK decisions in front of the chunk from ALL 128 states
//                             (two per lane; the decision is wave-uniform) and notes the states that are left - the CANDIDATES for the
//                             chunk's start state, at most 8; a chunk with more is left "unresolved".  Chunk 0 of a list starts at 255.
//   coder_range_first_kernel   one lane per (coarse chunk, candidate): the recurrence from the candidate over the chunk (and on through
//                             unresolved chunks behind it) -> the state at the start of the next resolved chunk, IF the chunk starts there
//   coder_range_link_kernel   one lane per pair: from 255 at the list's start, chunk after chunk: which candidate is the true start state,
//                             what that makes the next chunk's
//   coder_range_walk_kernel   one lane per resolved coarse chunk: the recurrence from the true start state, noting {range | pair << 8,
//                             bits shifted out since the coarse chunk's start} at every chunk of CODE_CHUNK decisions, and the bits each
//                             coarse chunk shifts out
//   coder_range_scan_kernel   one wave per pair: running sum of those bits over the pair's coarse chunks
// Worst case (a list whose states never merge): one lane walks the whole list, as the one-lane-per-pair kernel of round 2 did for
// every list (178 ms for the 1080p batch).
// Loads the compiler does not know about: a loop that reads ahead through ordinary loads gets "s_waitcnt vmcnt(0)" at its head
// (the loop-carried loads are tracked conservatively), i.e. one memory round trip per iteration.  Here the load and the wait are
// written out: `code_ld16` starts a 16-byte load into v (tied: the register is not renamed, nothing copies it while the data is
// under way), `code_wait<N>` waits until at most N younger vector-memory operations are outstanding and is v's first reader.
__device__ __forceinline__ void code_ld16 (u32x4& v, const GLB u32x4* p) { asm volatile ("global_load_dwordx4 %0, %1, off" : "+v" (v) : "v" (p) : "memory"); }
template <int N> __device__ __forceinline__ void code_wait (u32x4& v) { asm volatile ("s_waitcnt vmcnt(%1)" : "+v" (v) : "n" (N) : "memory"); }

// which pair coarse chunk G belongs to: the largest p with pair_coarse0[p] <= G (pairs without chunks share their successor's start)
__device__ __forceinline__ uint32_t coarse_pair (const uint32_t* __restrict__ pair_coarse0, uint32_t n_pairs, uint32_t G) {
  uint32_t lo = 0, hi = n_pairs;
  while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (pair_coarse0[mid] <= G) lo = mid; else hi = mid; }
  return lo;
}

#define CODE_LOOKBACK 1024u
#define CODE_CANDS 8
#define CODE_MAPPED 0xffffffff00000000ull      // candidate words of a chunk with more than CODE_CANDS possible start states
__global__ void __launch_bounds__ (256)
coder_range_seed_kernel (const uint32_t* __restrict__ chain_info, const uint16_t* __restrict__ Q, const uint32_t* __restrict__ pair_coarse0, int n_pairs,
                         uint32_t* __restrict__ cand, uint32_t* __restrict__ cand_list, uint32_t long_list) {
  const uint32_t G = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
  if (G >= pair_coarse0[n_pairs]) return;
  const uint32_t pair = (uint32_t)uniform ((int)coarse_pair (pair_coarse0, (uint32_t)n_pairs, G));
  const uint32_t c = G - pair_coarse0[pair];
  if (c == 0u) { if (lane < 2u) cand[2 * (size_t)G + lane] = lane == 0u ? 255u : 0u; return; }
  const PairInfo P = pair_info (chain_info, Q, pair);
  // a list of moderate length (the host's choice, lh264_capi.hip) is walked whole by the lane of its first chunk - candidates cost
  // several walks per chunk, which only pays where one lane would take longer than the rest of the coder
  if (P.total <= long_list) { if (lane < 2u) cand[2 * (size_t)G + lane] = 0u; return; }
  const uint32_t b = c * CODE_COARSE;
  uint32_t r0 = 128u + lane, r1 = 192u + lane;
  for (uint32_t i0 = b - CODE_LOOKBACK; i0 < b; i0 += 64u) {
    const uint32_t idx = i0 + lane;
    const uint32_t e = idx < P.n ? (uint32_t)P.list[idx] : CODE_STOP_ENTRY;
#pragma unroll 8
    for (int i = 0; i < 64; i++) {
      const uint32_t ei = (uint32_t)__builtin_amdgcn_readlane ((int)e, i);
      const uint32_t q = (ei >> 1) & 0x1ffu, k = 256u - (ei & 1u) - q;
      const uint32_t a0 = r0 * q + k, a1 = r1 * q + k;
      r0 = (a0 >> 8) << __builtin_clz (a0 << 16);
      r1 = (a1 >> 8) << __builtin_clz (a1 << 16);
    }
  }
  // the states that are left, one after the other (8 bytes; 0: none - a state is at least 128)
  bool act0 = true, act1 = true;
  unsigned long long cs = 0;
  for (int k = 0; k < CODE_CANDS; k++) {
    const unsigned long long m0 = __ballot (act0), m1 = __ballot (act1);
    if ((m0 | m1) == 0ull) break;
    const uint32_t v = m0 ? (uint32_t)__builtin_amdgcn_readlane ((int)r0, (int)__ffsll ((long long)m0) - 1) : (uint32_t)__builtin_amdgcn_readlane ((int)r1, (int)__ffsll ((long long)m1) - 1);
    cs |= (unsigned long long)v << (8 * k);
    act0 = act0 && r0 != v; act1 = act1 && r1 != v;
  }
  if (__ballot (act0 || act1)) cs = CODE_MAPPED;       // more than CODE_CANDS states left: the chunk's whole state map is worked out (coder_range_first_kernel)
#ifdef LH264_RANGE_PROBE     // diagnostic build: how many distinct states are left, noted behind the candidates
  { uint32_t d = 0; for (uint32_t v = 128u; v < 256u; v++) d += __ballot (r0 == v || r1 == v) != 0ull; if (lane == 0u) cand[2 * (size_t)pair_coarse0[n_pairs] + G] = d; }
#endif
  if (lane < 2u) cand[2 * (size_t)G + lane] = (uint32_t) (cs >> (32 * lane));
  // several candidates: a walk from each - one dense list of (chunk, candidate) for coder_range_first_kernel (cand_list[0] = their number)
  if (cs != CODE_MAPPED && (cs >> 8) != 0ull) {
    int nc = 0;
    for (int k = 0; k < CODE_CANDS; k++) nc += ((cs >> (8 * k)) & 0xffull) != 0ull;
    uint32_t base = 0;
    if (lane == 0u) base = atomicAdd (cand_list, (uint32_t)nc);
    base = (uint32_t)__builtin_amdgcn_readfirstlane ((int)base);
    if (lane < (uint32_t)nc) cand_list[1u + base + lane] = G * CODE_CANDS + lane;
  }
}

struct RangeWalk {
  GLB uint32_t* rec; uint32_t range, pos, g0, pair;
  __device__ __forceinline__ void note (uint32_t i) { if (rec) { const size_t g = g0 + i / CODE_CHUNK; rec[2 * g] = range | pair << 8; rec[2 * g + 1] = pos; } }
  // one whole piece (8 decisions): no test per decision; a chunk starts on a piece
  __device__ __forceinline__ void piece (const u32x4 v, uint32_t c) {
    if ((c & (CODE_CHUNK / 8 - 1u)) == 0u) note (c * 8u);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 8; q++) pos += code_step (range, w[q >> 1] >> (16 * (q & 1))).shift;
  }
};
#define ACC_WIN 2048u         // positions of the accumulate kernel's LDS window per wave
#define CODE_AHEAD 8u        // 16-byte pieces of the list under way per lane
// the walk of coarse chunk G from state s0, on through the unresolved chunks behind it.  NOTES: the final walk (chunk_rec, coarse_bits);
// otherwise only the state the walk ends with is wanted.  resolved[2 * G] != 0 <=> chunk G has candidates (a lane of its own).
template <bool NOTES>
__device__ __forceinline__ uint32_t range_walk (const uint32_t* __restrict__ chain_info, const uint16_t* __restrict__ Q, const uint32_t* __restrict__ pair_chunk0,
                                                const uint32_t* __restrict__ pair_coarse0, int n_pairs, const uint32_t* __restrict__ resolved,
                                                uint32_t G, uint32_t s0, uint32_t* __restrict__ chunk_rec, uint32_t* __restrict__ coarse_bits) {
  RangeWalk A;
  A.pair = coarse_pair (pair_coarse0, (uint32_t)n_pairs, G);
  A.range = s0; A.pos = 0; A.rec = NOTES ? glb<uint32_t> (chunk_rec) : (GLB uint32_t*)nullptr;
  A.g0 = pair_chunk0[A.pair];
  const uint32_t G_end = pair_coarse0[A.pair + 1];
  const PairInfo P = pair_info (chain_info, Q, A.pair);
  const uint32_t n = P.n;
  const GLB u32x4* src = (const GLB u32x4*)P.list;
  for (uint32_t cc = G - pair_coarse0[A.pair]; ; cc++, G++) {
    // decisions [i0, i1) of the list (the 32 stop decisions lie behind the n real ones); pieces [c, whole) are whole and real
    const uint32_t i0 = cc * CODE_COARSE, i1 = min (i0 + CODE_COARSE, P.total);
    const uint32_t whole = min (i1, n) >> 3, last = n ? ((n + 7u) >> 3) - 1u : 0u;
    uint32_t c = i0 >> 3;
    A.pos = 0;
    // eight separate registers quadruples (an array would be one aggregate that the compiler copies around while loads are under way)
    u32x4 b0 = {0u, 0u, 0u, 0u}, b1 = b0, b2 = b0, b3 = b0, b4 = b0, b5 = b0, b6 = b0, b7 = b0;
    const bool any = c < whole || (whole << 3) < min (i1, n);      // the chunk reads the list at all
#define CODE_FIRST(B, J) if (any) code_ld16 (B, src + min (c + (uint32_t) (J), last));
    CODE_FIRST (b0, 0) CODE_FIRST (b1, 1) CODE_FIRST (b2, 2) CODE_FIRST (b3, 3) CODE_FIRST (b4, 4) CODE_FIRST (b5, 5) CODE_FIRST (b6, 6) CODE_FIRST (b7, 7)
#undef CODE_FIRST
    // CODE_AHEAD loads are under way all the time (behind the end of the list the last piece is read again), so the oldest one has
    // arrived when at most CODE_AHEAD - 1 are outstanding
#define CODE_TURN(B, J) code_wait<CODE_AHEAD - 1> (B); A.piece (B, c + (J)); code_ld16 (B, src + min (c + (J) + CODE_AHEAD, last));
    for (; c + CODE_AHEAD <= whole; c += CODE_AHEAD) {
      CODE_TURN (b0, 0u) CODE_TURN (b1, 1u) CODE_TURN (b2, 2u) CODE_TURN (b3, 3u) CODE_TURN (b4, 4u) CODE_TURN (b5, 5u) CODE_TURN (b6, 6u) CODE_TURN (b7, 7u)
    }
#undef CODE_TURN
    code_wait<0> (b0); code_wait<0> (b1); code_wait<0> (b2); code_wait<0> (b3); code_wait<0> (b4); code_wait<0> (b5); code_wait<0> (b6); code_wait<0> (b7);
    // fewer than CODE_AHEAD whole pieces are left: register j holds piece min (c + j, last); then the list's last, partial piece
    const uint32_t left = c < whole ? whole - c : 0u;
    u32x4 pv = b0;
#define CODE_LAST(B, J) if ((J) < left) A.piece (B, c + (J)); if (left == (J)) pv = B;
    CODE_LAST (b0, 0u) CODE_LAST (b1, 1u) CODE_LAST (b2, 2u) CODE_LAST (b3, 3u) CODE_LAST (b4, 4u) CODE_LAST (b5, 5u) CODE_LAST (b6, 6u) CODE_LAST (b7, 7u)
#undef CODE_LAST
    const uint32_t w[4] = {pv.x, pv.y, pv.z, pv.w};
    for (uint32_t i = max (i0, whole * 8u); i < i1; i++) {
      if ((i & (CODE_CHUNK - 1u)) == 0u) A.note (i);
      const uint32_t j = i & 7u;
      const uint32_t wj = (j & 4u) ? ((j & 2u) ? w[3] : w[2]) : ((j & 2u) ? w[1] : w[0]);
      A.pos += code_step (A.range, i < n ? wj >> (16u * (j & 1u)) : CODE_STOP_ENTRY).shift;
    }
    if (NOTES) coarse_bits[G] = A.pos;                  // the bits this coarse chunk shifts out
    if (G + 1u >= G_end || resolved[2 * (size_t) (G + 1u)] != 0u || resolved[2 * (size_t) (G + 1u) + 1u] != 0u) break;   // the next chunk has a lane of its own (or there is none)
  }
  return A.range;
}
// The walks that need nothing but the candidates, in ONE launch (each of them is a few hundred waves walking serially - one after the
// other they left the machine idle three times over; the launch's workgroups are dealt to the three in the order below, the longest first):
//   blocks [0, 35 groups)         one lane per list: its first chunk starts at 255 - the final walk at once; a wave = one tag slot of 64
//                                 consecutive streams (lists of about the same length in its lanes: a wave is as slow as its longest list).
//                                 Few waves, each a long serial walk: they start first and the rest fills the machine around them
//   blocks [.., + n_cand)         one lane per (coarse chunk with several candidates, candidate start state) of the seed kernel's list: where
//                                 the walk ends
//   the rest                      one lane per later coarse chunk with ONE candidate: that is its start state - the final walk at once
__global__ void __launch_bounds__ (64)
coder_range_first_kernel (const uint32_t* __restrict__ chain_info, const uint16_t* __restrict__ Q, const uint32_t* __restrict__ pair_chunk0,
                          const uint32_t* __restrict__ pair_coarse0, int n_pairs, int groups, unsigned n_cand, unsigned n_later, const uint32_t* __restrict__ cand, const uint32_t* __restrict__ cand_list,
                          uint8_t* __restrict__ cand_end, uint8_t* __restrict__ cmap, uint32_t* __restrict__ chunk_rec, uint32_t* __restrict__ coarse_bits) {
  const uint32_t n_first = (uint32_t)groups * 35u;
  uint32_t G;
  if (blockIdx.x < n_first) {
    const uint32_t slot = blockIdx.x / (uint32_t)groups, chain = (blockIdx.x % (uint32_t)groups) * 64u + threadIdx.x;
    const uint32_t pair = chain * LH264_N_TAG_SLOTS + slot;
    if (pair >= (uint32_t)n_pairs) return;
    G = pair_coarse0[pair];
    if (pair_coarse0[pair + 1] == G) return;             // the tag has no list
  } else if (blockIdx.x < n_first + n_cand) {
    const uint32_t idx = (blockIdx.x - n_first) * 64u + threadIdx.x;
    if (idx >= cand_list[0]) return;
    const uint32_t gk = cand_list[1u + idx], k = gk % CODE_CANDS;
    G = gk / CODE_CANDS;
    const uint32_t c0 = cand[2 * (size_t)G], c1 = cand[2 * (size_t)G + 1];
    const uint32_t s0 = ((k < 4u ? c0 : c1) >> (8 * (k & 3))) & 0xffu;
    cand_end[(size_t)G * CODE_CANDS + k] = (uint8_t)range_walk<false> (chain_info, Q, pair_chunk0, pair_coarse0, n_pairs, cand, G, s0, nullptr, nullptr);
    return;
  } else if (blockIdx.x < n_first + n_cand + n_later) {
    G = (blockIdx.x - n_first - n_cand) * 64u + threadIdx.x;
    if (G >= pair_coarse0[n_pairs]) return;
    if (G == pair_coarse0[coarse_pair (pair_coarse0, (uint32_t)n_pairs, G)]) return;      // a first chunk: walked above
  } else {
    // a wave per coarse chunk whose start state the lookback could not narrow down (a list that only rotates the states - a run of
    // near-certain decisions): where EVERY start state ends, two states per lane, the decision wave-uniform.  Twelve instructions a
    // decision for all 128 - a lane walking such a list on its own, chunk after chunk, was the longest wave of the launch (10.8 ms)
    G = blockIdx.x - n_first - n_cand - n_later;
    if (G >= pair_coarse0[n_pairs]) return;
    if (cand[2 * (size_t)G] != 0u || cand[2 * (size_t)G + 1] != (uint32_t) (CODE_MAPPED >> 32)) return;
    const uint32_t lane = threadIdx.x;
    const uint32_t pair = (uint32_t)uniform ((int)coarse_pair (pair_coarse0, (uint32_t)n_pairs, G));
    const PairInfo P = pair_info (chain_info, Q, pair);
    const uint32_t i0 = (G - pair_coarse0[pair]) * CODE_COARSE, i1 = min (i0 + CODE_COARSE, P.total);
    uint32_t r0 = 128u + lane, r1 = 192u + lane;
    uint32_t e_next = i0 + lane < P.n ? (uint32_t)P.list[i0 + lane] : CODE_STOP_ENTRY;
    for (uint32_t b = i0; b < i1; b += 64u) {
      const uint32_t e = e_next;
      e_next = b + 64u + lane < P.n ? (uint32_t)P.list[b + 64u + lane] : CODE_STOP_ENTRY;
      const int cnt = (int)min (64u, i1 - b);
      if (cnt == 64) {
#pragma unroll 8
        for (int i = 0; i < 64; i++) {
          const uint32_t ei = (uint32_t)__builtin_amdgcn_readlane ((int)e, i);
          const uint32_t q = (ei >> 1) & 0x1ffu, k = 256u - (ei & 1u) - q;
          const uint32_t a0 = r0 * q + k, a1 = r1 * q + k;
          r0 = (a0 >> 8) << __builtin_clz (a0 << 16);
          r1 = (a1 >> 8) << __builtin_clz (a1 << 16);
        }
      } else {
        for (int i = 0; i < cnt; i++) {
          const uint32_t ei = (uint32_t)__shfl ((int)e, i);
          const uint32_t q = (ei >> 1) & 0x1ffu, k = 256u - (ei & 1u) - q;
          const uint32_t a0 = r0 * q + k, a1 = r1 * q + k;
          r0 = (a0 >> 8) << __builtin_clz (a0 << 16);
          r1 = (a1 >> 8) << __builtin_clz (a1 << 16);
        }
      }
    }
    cmap[(size_t)G * 128u + lane] = (uint8_t)r0; cmap[(size_t)G * 128u + 64u + lane] = (uint8_t)r1;
    return;
  }
  const uint32_t c0 = cand[2 * (size_t)G], c1 = cand[2 * (size_t)G + 1];
  if (c0 == 0u || (c0 >> 8) != 0u || c1 != 0u) return;
  cand_end[(size_t)G * CODE_CANDS] = (uint8_t)range_walk<true> (chain_info, Q, pair_chunk0, pair_coarse0, n_pairs, cand, G, c0, chunk_rec, coarse_bits);
}
// one lane per pair: the true start state of every coarse chunk with several candidates (seed; 0 for the others: walked already, or
// walked by the lane of the chunk in front of them)
__global__ void __launch_bounds__ (64)
coder_range_link_kernel (const uint32_t* __restrict__ pair_coarse0, int n_pairs, const uint32_t* __restrict__ cand, const uint8_t* __restrict__ cand_end,
                         const uint8_t* __restrict__ cmap, uint32_t* __restrict__ seed, uint32_t* __restrict__ chain_info) {
  const uint32_t pair = blockIdx.x * 64u + threadIdx.x;
  if (pair >= (uint32_t)n_pairs) return;
  uint32_t s = 255u;
  bool lost = false;
  for (uint32_t G = pair_coarse0[pair]; G < pair_coarse0[pair + 1]; G++) {
    const unsigned long long cs = (unsigned long long)cand[2 * (size_t)G] | (unsigned long long)cand[2 * (size_t)G + 1] << 32;
    if (cs == 0ull) { seed[G] = 0u; continue; }
    if (cs == CODE_MAPPED) { seed[G] = s; s = cmap[(size_t)G * 128u + (s - 128u)]; continue; }      // every start state's end is known
    int k = -1;
    for (int q = CODE_CANDS - 1; q >= 0; q--) if (((cs >> (8 * q)) & 0xffull) == (unsigned long long)s) k = q;
    if (k < 0) { lost = true; k = 0; }                 // (cannot happen: the lookback starts from every state)
    seed[G] = (cs >> 8) == 0ull ? 0u : s;
    s = cand_end[(size_t)G * CODE_CANDS + k];
  }
  if (lost) atomicOr (&chain_info[(size_t) (pair / LH264_N_TAG_SLOTS) * LH264_CODER_INFO_WORDS + LH264_CODER_INFO_STATUS], (uint32_t)LH264_CODER_ST_HANDOFF);
}
__global__ void __launch_bounds__ (64)
coder_range_walk_kernel (const uint32_t* __restrict__ chain_info, const uint16_t* __restrict__ Q, const uint32_t* __restrict__ pair_chunk0,
                         const uint32_t* __restrict__ pair_coarse0, int n_pairs, const uint32_t* __restrict__ cand, const uint32_t* __restrict__ seed,
                         uint32_t* __restrict__ chunk_rec, uint32_t* __restrict__ coarse_bits) {
  const uint32_t G = blockIdx.x * 64u + threadIdx.x;
  if (G >= pair_coarse0[n_pairs]) return;
  const uint32_t s0 = seed[G];
  if (s0 == 0u) return;                                 // walked by the lane of the resolved chunk in front of it
  range_walk<true> (chain_info, Q, pair_chunk0, pair_coarse0, n_pairs, cand, G, s0, chunk_rec, coarse_bits);
}

// running sum of the bits over a pair's coarse chunks: coarse_bits[G] becomes the bits shifted out in front of coarse chunk G,
// pair_bits[pair] all the bits the list shifts out
__global__ void __launch_bounds__ (256)
coder_range_scan_kernel (const uint32_t* __restrict__ pair_coarse0, int n_pairs, uint32_t* __restrict__ coarse_bits, uint32_t* __restrict__ pair_bits,
                         const uint32_t* __restrict__ chain_info, const uint16_t* __restrict__ Q, uint32_t* __restrict__ acc) {
  const uint32_t pair = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
  if (pair >= (uint32_t)n_pairs) return;
  const uint32_t G0 = pair_coarse0[pair], G1 = pair_coarse0[pair + 1];
  uint32_t carry = 0;
  for (uint32_t g = G0; g < G1; g += 64u) {
    const uint32_t v = g + lane < G1 ? coarse_bits[g + lane] : 0u;
    const uint32_t incl = (uint32_t)wave_scan_add ((int)v);
    if (g + lane < G1) coarse_bits[g + lane] = carry + incl - v;
    carry += (uint32_t)__builtin_amdgcn_readlane ((int)incl, 63);
  }
  if (lane == 0u) pair_bits[pair] = carry;
  // the pair's sums start at zero - the positions its bits reach, not the whole bound of one position per list entry (that was a 23 GB
  // fill per call on the large batches: 4 ms)
  const PairInfo P = pair_info (chain_info, Q, pair);
  if (P.used) {
    GLB uint32_t* A = glb<uint32_t> (acc) + P.acc0;
    const uint32_t n = (carry >> 3) + 8u;
    for (uint32_t k = lane; k < n; k += 64u) A[k] = 0u;
  }
}

// kernel 8: the addends of every chunk into the sums of the output byte positions.  A position takes addends from the decisions that
// start in its byte or in the byte before; the positions strictly inside a chunk's span of bits belong to that chunk alone and are
// stored, the two at either end are shared with the neighbouring chunks and added atomically (the sums start out as zero).
__global__ void __launch_bounds__ (256)
coder_accum_kernel (const uint32_t* __restrict__ chain_info, const uint16_t* __restrict__ Q, const uint32_t* __restrict__ pair_chunk0,
                    const uint32_t* __restrict__ pair_coarse0, int n_pairs,
                    const uint32_t* __restrict__ chunk_rec, const uint32_t* __restrict__ coarse_bits, const uint32_t* __restrict__ pair_bits, uint32_t* __restrict__ acc) {
  // The sums of a wave's 64 chunks fall on a few thousand consecutive positions (the chunks follow one another in a list).  Stored one
  // by one they were 4-byte pieces a lane apart - 15 GB written for under 2 GB of sums; they are collected in a window of the wave's LDS
  // (positions relative to its first chunk's; what falls outside, or belongs to another list, goes to memory as before) and added to
  // memory at the end, 64 consecutive positions an instruction.
  __shared__ uint32_t wing[4][ACC_WIN];
  const uint32_t lane = threadIdx.x & 63u;
  LDS uint32_t* win = (LDS uint32_t*) (uintptr_t) (uint32_t) (uintptr_t)&wing[threadIdx.x >> 6][0];
  const uint32_t g_raw = blockIdx.x * 256u + threadIdx.x, n_chunks = pair_chunk0[n_pairs];
  if ((uint32_t)__builtin_amdgcn_readfirstlane ((int)g_raw) >= n_chunks) return;      // (lane 0 has the wave's first chunk: the whole wave is beyond the last one)
  for (uint32_t i = lane * 4u; i < ACC_WIN; i += 256u) { u32x4 z = {0u, 0u, 0u, 0u}; * (LDS u32x4*) (win + i) = z; }
  const bool live = g_raw < n_chunks;
  const uint32_t g = live ? g_raw : n_chunks - 1u;          // (a lane beyond the last chunk: reads that chunk's tables, adds nothing)
  uint32_t range = chunk_rec[2 * (size_t)g], t = chunk_rec[2 * (size_t)g + 1];
  const uint32_t pair = range >> 8;
  range &= 0xffu;
  const PairInfo P = pair_info (chain_info, Q, pair);
  const uint32_t c = g - pair_chunk0[pair], i0 = c * CODE_CHUNK, i1 = min (i0 + CODE_CHUNK, P.total);
  // (the noted bit positions count from the start of the coarse chunk: the bits in front of it on top)
  const uint32_t G0 = pair_coarse0[pair];
  t += coarse_bits[G0 + c / (CODE_COARSE / CODE_CHUNK)];
  const uint32_t t_end = g + 1u < pair_chunk0[pair + 1] ? chunk_rec[2 * (size_t)g + 3] + coarse_bits[G0 + (c + 1u) / (CODE_COARSE / CODE_CHUNK)]
                                                        : pair_bits[pair];     // where the next chunk starts
  const uint32_t own_lo = (t >> 3) + 2u, own_hi = t_end >> 3;            // positions own_lo .. own_hi - 1 are this chunk's alone
  GLB uint32_t* A = glb<uint32_t> (acc) + P.acc0;
  const uint32_t pair0 = (uint32_t)__builtin_amdgcn_readfirstlane ((int)pair), base0 = (uint32_t)__builtin_amdgcn_readfirstlane ((int) (t >> 3));
  const bool same = pair == pair0;
  wsync();
  auto put = [&] (uint32_t kpos, uint32_t v) {
    if (v == 0u || !live) return;
    const uint32_t rel = kpos - base0;
    if (same && rel < ACC_WIN) __hip_atomic_fetch_add (win + rel, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if (kpos >= own_lo && kpos < own_hi) A[kpos] = v;
    else atomicAdd ((uint32_t*) (uintptr_t) (A + kpos), v);
  };
  // window: the addends at byte position kb (bits 8 and up of w) and kb + 1 (bits 0..7); an addend starts t bits behind the list's first bit
  uint32_t kb = t >> 3, w = 0;
  auto one = [&] (uint32_t e) {
    const CodeStep s = code_step (range, e);
    if (s.add) {
      const uint32_t k = t >> 3;
      if (k != kb) {
        put (kb, w >> 8);
        if (k == kb + 1u) w = (w & 0xffu) << 8;
        else { put (kb + 1u, w & 0xffu); w = 0; }
        kb = k;
      }
      w += s.add << (8u - (t & 7u));
    }
    t += s.shift;
  };
  // whole pieces of the chunk, read four ahead (a chunk starts on a piece); then what is left of the list and the stop decisions
  const GLB u32x4* src = (const GLB u32x4*)P.list;
  const uint32_t c0 = i0 >> 3, c1 = min (i1, P.n) >> 3;         // pieces c0 .. c1-1 lie inside the chunk and inside the list
  // A lane owns 512 consecutive bytes of a list, its neighbour the next 512: every load of the wave touches 64 different lines.  The
  // lines are therefore taken WHOLE, eight 16-byte pieces = 128 bytes per lane at a time, one burst ahead of their use - taken a piece at
  // a time, a line was long gone from the L1 and the L2 when the lane came back for its next piece and crossed the fabric again
  // (76 GB fetched for 11.6 GB of lists).  (Ordinary loads: the atomics and stores between them are vector-memory operations too, a
  // counted wait would wait for nearly everything; with thousands of chunks per SIMD the latency is covered by other waves.)
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  u32x4 nx[8];
#pragma unroll
  for (int k = 0; k < 8; k++) nx[k] = c0 + (uint32_t)k < c1 ? src[c0 + (uint32_t)k] : zero4;
  for (uint32_t cb = c0; cb < c1; cb += 8u) {
    u32x4 cur[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { cur[k] = nx[k]; nx[k] = cb + 8u + (uint32_t)k < c1 ? src[cb + 8u + (uint32_t)k] : zero4; }
#pragma unroll
    for (int k = 0; k < 8; k++) {
      if (cb + (uint32_t)k >= c1) break;
      const uint32_t wd[4] = {cur[k].x, cur[k].y, cur[k].z, cur[k].w};
#pragma unroll
      for (int q = 0; q < 8; q++) one (wd[q >> 1] >> (16 * (q & 1)));
    }
  }
  ListReader L; L.init (P.list, P.n);
  for (uint32_t i = max (i0, c1 * 8u); i < i1; i++) one (L.at (i));
  put (kb, w >> 8);
  put (kb + 1u, w & 0xffu);
  // the window to memory (added: the positions at either end are shared with the neighbouring waves' windows)
  uint32_t top = same && live ? min (kb + 2u - base0, (uint32_t)ACC_WIN) : 0u;
  for (int m = 1; m < 64; m <<= 1) top = max (top, (uint32_t)__shfl_xor ((int)top, m));
  wsync();
  GLB uint32_t* A0 = glb<uint32_t> (acc) + ((unsigned long long) (uint32_t)__builtin_amdgcn_readfirstlane ((int) (uint32_t)P.acc0) |
                                             (unsigned long long) (uint32_t)__builtin_amdgcn_readfirstlane ((int) (uint32_t) (P.acc0 >> 32)) << 32) + base0;
  for (uint32_t i = lane; i < top; i += 64u) {
    const uint32_t v = * (volatile LDS uint32_t*) (win + i);
    if (v) atomicAdd ((uint32_t*) (uintptr_t) (A0 + i), v);
  }
}

// kernel 9: carries, bytes, lengths.  vpx_write puts a byte out whenever 8 more bits have been shifted out beyond the first 24:
// bytes = (bits - 24) / 8 + 1; byte k is byte position k of the sum.  vpx_stop_encode appends a zero byte behind a last byte 110xxxxx.
// One WAVE per pair, 256 positions per step from the last to the first (round 2: one lane per pair, a position at a time - 30 ms for the
// 1080p batch's megabyte tags).  A position's sum holds up to 32 bits, i.e. it reaches three positions up; two local steps bring
// every position down to a digit of at most 258 - B[k] = the bytes of A[k .. k+3] that fall on k (< 1024), C[k] = B[k] mod 256 + B[k+1]
// div 256 - and from there on a carry is one bit: a position generates one (C >= 256), passes one on (C == 255) or ends it.  The
// carries into the 64 positions of a step are then ONE 64-bit addition of the two lane masks (the adder's own carry chain does the
// work: carries = ((G | P) + G + carry_in) ^ P), and the carry out of the step goes into the next.
__global__ void __launch_bounds__ (256)
coder_bytes_kernel (const lh264_code_stream_t* __restrict__ streams, const uint32_t* __restrict__ chain_info, const uint16_t* __restrict__ Q,
                    const uint32_t* __restrict__ pair_bits, const uint32_t* __restrict__ acc, int n_pairs) {
  const uint32_t pair = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
  if (pair >= (uint32_t)n_pairs) return;
  const uint32_t chain = pair / LH264_N_TAG_SLOTS, slot = pair % LH264_N_TAG_SLOTS;
  if (slot >= 35u) return;                             // tag slots that do not exist (their lengths are cleared by the status kernel)
  const lh264_code_stream_t* S = streams + chain;
  GLB uint32_t* lens = glb<uint32_t> (S->out_len_dev);
  const PairInfo P = pair_info (chain_info, Q, pair);
  if (!P.used) { if (lane == 0u) lens[slot] = 0; return; }
  const uint32_t bits = pair_bits[pair], cap = S->out_cap;
  uint32_t nbytes = bits >= 24u ? ((bits - 24u) >> 3) + 1u : 0u;
  const GLB uint32_t* A = glb<const uint32_t> (acc) + P.acc0;
  GLB uint8_t* o = glb<uint8_t> (S->out_dev) + (size_t)slot * cap;
  const bool wide = (((uintptr_t)o) & 3u) == 0u;
  const uint32_t last = (bits >> 3) + 2u;                      // no addend lies behind this position (the sums behind it read zero)
  uint32_t cin = 0, final_byte = 0;
  // A lane holds FOUR consecutive positions (one dword of output), lane i of a step the positions k0 + 4 (63 - i) .. + 3: bit i of a lane
  // mask is then the bit a carry moves UP from towards bit i + 1.  Inside its four positions a lane works the carry out for both
  // cases - none coming in, one coming in - and the wave's two additions pick: 256 positions per step (a position per lane and step
  // took 10 ms for the 1080p batch's 850 KB tags: 13 k dependent steps of two ballots and three lane exchanges each).
  for (uint32_t k0 = last & ~255u; ; k0 -= 256u) {
    const uint32_t p0 = k0 + 4u * (63u - lane);
    // (loads of a clamped index, the value masked afterwards: a load under a branch is waited for before the next one is issued;
    // behind `last` the next pair's sums begin - read as zero)
    uint32_t a[8];
#pragma unroll
    for (int q = 0; q < 8; q++) a[q] = A[min (p0 + (uint32_t)q, last)];
#pragma unroll
    for (int q = 0; q < 8; q++) a[q] = p0 + (uint32_t)q <= last ? a[q] : 0u;
    uint32_t B[5];
#pragma unroll
    for (int q = 0; q < 5; q++) B[q] = (a[q] & 255u) + ((a[q + 1] >> 8) & 255u) + ((a[q + 2] >> 16) & 255u) + (a[q + 3] >> 24);
    uint32_t c0 = 0, c1 = 1, y0 = 0, y1 = 0;                     // the four bytes without / with a carry coming in, the carry going out
#pragma unroll
    for (int q = 3; q >= 0; q--) {
      const uint32_t C = (B[q] & 255u) + (B[q + 1] >> 8);
      const uint32_t t0 = C + c0, t1 = C + c1;
      y0 |= (t0 & 255u) << (8 * q); c0 = t0 >> 8;
      y1 |= (t1 & 255u) << (8 * q); c1 = t1 >> 8;
    }
    const unsigned long long G = __ballot (c0 != 0u), Pm = __ballot (c0 == 0u && c1 != 0u);
    const unsigned long long X = G | Pm, s1 = X + G, s2 = s1 + cin;
    const uint32_t cout = (s1 < X || s2 < s1) ? 1u : 0u;
    const unsigned long long carries = s2 ^ Pm;                // bit i: the carry INTO lane i's positions
    const uint32_t y = ((carries >> lane) & 1ull) ? y1 : y0;
    cin = cout;
    if (nbytes > 0u && nbytes - 1u - p0 < 4u) final_byte = (y >> (8u * (nbytes - 1u - p0))) & 255u;
    if (p0 < nbytes) {
      if (wide && p0 + 3u < nbytes && p0 + 3u < cap) * (GLB uint32_t*) (o + p0) = y;
      else {
#pragma unroll
        for (int q = 0; q < 4; q++) if (p0 + (uint32_t)q < nbytes && p0 + (uint32_t)q < cap) o[p0 + (uint32_t)q] = (uint8_t) (y >> (8 * q));
      }
    }
    if (k0 == 0u) break;
  }
  final_byte = (uint32_t)__builtin_amdgcn_readlane (wave_scan_add ((int)final_byte), 63);      // (one lane held it)
  if (lane == 0u) {
    if (nbytes > 0u && (final_byte & 0xe0u) == 0xc0u) { if (nbytes < cap) o[nbytes] = 0; nbytes++; }
    lens[slot] = nbytes;
    if (nbytes > cap) atomicOr ((uint32_t*) (uintptr_t) (lens + LH264_N_TAG_SLOTS), (uint32_t)LH264_CODER_ST_OUT_FULL);
  }
}

// the status word of every stream, before the coding kernel adds its own bit
__global__ void __launch_bounds__ (256)
coder_status_kernel (const lh264_code_stream_t* __restrict__ streams, const uint32_t* __restrict__ chain_info, int n_chains) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < n_chains) {
    GLB uint32_t* lens = glb<uint32_t> (streams[c].out_len_dev);
    uint32_t st = chain_info[(size_t)c * LH264_CODER_INFO_WORDS + LH264_CODER_INFO_STATUS];
    const uint32_t hc = streams[c].hash_cap;
    if (hc == 0u || (hc & (hc - 1u)) != 0u || hc > (1u << 20)) st |= LH264_CODER_ST_TABLE_FULL;      // a decision word carries 20 bits of table slot
    lens[LH264_N_TAG_SLOTS] = st;
    for (int t = 35; t < LH264_N_TAG_SLOTS; t++) lens[t] = 0;                                        // tag slots that do not exist
#ifdef LH264_CODER_DEBUG
    for (int q = 0; q < 5; q++) lens[35 + q] = chain_info[(size_t)c * LH264_CODER_INFO_WORDS + 90 + q];
    lens[34] = chain_info[(size_t)c * LH264_CODER_INFO_WORDS + 95];
#endif
  }
}

}  // namespace lh264
