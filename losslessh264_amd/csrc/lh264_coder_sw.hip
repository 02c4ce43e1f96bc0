// lh264_coder_sw.hip - the STREAM-PER-WORKGROUP form of the coder's first stages (round 2's kernels): binarisation over segments of 128
// macroblocks into decision words in CODING ORDER, and one workgroup of 8 waves per stream that resolves the adaptive probabilities 512
// decisions at a time, its waves taking turns on the stream's counters (a ticket in LDS).  With many small streams in the batch (hundreds
// of workgroups, each stream's state in one LDS cache) this is the faster form; lh264_coder.hip holds the form that scales INSIDE a stream
// (per-partition runs, one wave per (stream, partition)) for batches of few large streams, and the stages behind both (the bool coders).
// lh264_capi.hip chooses per call (code_binarise).  Same inputs, same tag lists out: the byte-identity tests run both.
// (the header comment of round 2's file follows)
// lh264_coder.hip - the recompressor's adaptive binary arithmetic coder on the device.
//
// The reference codes a stream strictly serially: symbol -> binarisation (emitInt / emitUEGkInt / Branch<n> /
// emitBitsZeroToPow2Inclusive, /root/reference/codec/decoder/core/inc/compression_stream.h:117-166,455-591) -> per decision an
// adaptive probability (DynProb :87-115) -> the libvpx bool coder of the decision's tag (bitwriter.h:35-105).  Only two things in
// that chain are really sequential: the state of ONE DynProb over the decisions made with it, and the state of ONE tag's bool coder
// over the decisions sent to it.  Everything else is data parallel, so the work is cut into kernels along those two lines:
//
//   coder_count_kernel    per segment (<= 128 consecutive macroblocks of a picture), a thread per symbol: the number of decisions
//                         per tag in closed form (sym_count), summed per segment                                   (parallel)
//   coder_scan_kernel     per stream: where each segment's decisions start; size of every tag's list               (small)
//   coder_bases_kernel    prefix over the streams; totals for the host                                              (small)
//   coder_emit_kernel     per segment, four waves over its flat symbol list: binarise, write one 64-bit word per decision in
//                         coding order (the key of the DynProb's cell, the place in it, bit, tag slot / raw bit)     (parallel)
//   coder_resolve_kernel  one workgroup of 8 waves per stream, 64 decisions per wave step: a DynProb is two counters, so the
//                         probability a decision is coded with follows from the counters before the step and PREFIX COUNTS of
//                         the earlier decisions of the step on the same DynProb.  Lanes holding the same DynProb find each other
//                         with ballots (no serial walk); the only serial part is a short ticketed section per step: read the
//                         counters, write them back.  The counters live in a keyed LDS cache in front of a spill table in HBM.
//                         Output: (probability of the bit that occurred, bit) appended to the list of the decision's tag.
//                                                                                                  (serial per stream, 512 wide)
//   (the bool coder behind them - range walk, sums, carries - is lh264_coder.hip's, shared by both forms)
//
// Halving is lazy: the table holds the un-halved pair of counters and the reader halves when their sum has passed 512, so the
// probability always follows from the pair; zero-filled memory is the initial state.  See DESIGN.md section 4.3.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/lh264.h"
#include "lh264_coder.h"
#undef LH264_CODER_SEG_MBS
#define LH264_CODER_SEG_MBS LH264_CODER_SW_SEG_MBS
#ifndef LH264_CODER_RESOLVE_WAVES
#define LH264_CODER_RESOLVE_WAVES 8
#endif
#define LH264_CODER_RESOLVE_THREADS (64 * LH264_CODER_RESOLVE_WAVES)

namespace lh264sw {

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

// ---- binarisation: a symbol becomes a short list of decisions ---------------------------------------------------------------
// Written once against a sink: sink.cell (key) names the 16-DynProb cell the following decisions use (priors that are trees of
// more than 16 nodes span several cells), sink.dec (j, bit, tag) is one decision on place j of that cell, j == 0xff a raw bit
// (coded with the shared TEST_PROB, compression_stream.h:363,441-448), sink.touch (tag) a stream that comes into existence.
template <class S> __device__ __forceinline__ void bz_unary (S& s, int data, int base, int n, int early, int tag) {   // emitUnary :465-474
  for (int i = 0; i < data; i++) {
    s.dec (base + (i < n - 1 ? i : n - 1), 1, tag);
    if (i == early - 1) return;
  }
  s.dec (base + (data < n - 1 ? data : n - 1), 0, tag);
}
// emitInt :523-572 with the prior's parts at fixed places of the cell (zero / sign < 0: the prior has none)
template <class S> __device__ __forceinline__ void bz_int (S& s, int data, int zero, int sign, int ebase, int E, int mbase, int M, int order,
                                                            int tag_exp, int tag_man, int tag_zero, int tag_sign) {
  if (zero >= 0) { s.dec (zero, data == 0, tag_zero); if (data == 0) return; }
  if (sign >= 0) { s.dec (sign, data > 0, tag_sign); if (data < 0) data = -data; }
  data--;
  const int data_high = 1 + (data >> order);
  const int log2 = 31 - __clz (data_high);               // largest l with (1 << l) <= data_high
  bz_unary (s, log2, ebase, E, -1, tag_exp);
  int lo = 0, hi = M;
  const int nb = log2 + order;
  for (int i = 0; i < nb; i++) {
    const int bit = i < log2 ? (data_high >> (log2 - 1 - i)) & 1 : (data >> (order - 1 - (i - log2))) & 1;
    if (hi > lo) {
      const int mid = (hi + lo) / 2;
      s.dec (mbase + mid, bit, tag_man);
      if (bit) lo = mid + 1; else hi = mid;
    } else s.dec (0xff, bit, tag_man);
  }
}
// emitUEGkInt :575-591; cell: zero 0, sign 1, first 2..2+M-1, second = {zero, exponent[E], mantissa[Mant]}
template <class S> __device__ __forceinline__ void bz_uegk (S& s, int data, int N, int M, int E, int Mant, int order, int tag_exp, int tag_man, int tag_zero, int tag_sign) {
  s.dec (0, data == 0, tag_zero);
  if (data == 0) return;
  s.dec (1, data < 0, tag_sign);
  if (data < 0) data = -data;
  bz_unary (s, data - 1, 2, M, N, tag_man);
  if (data - 1 >= N) bz_int (s, data - 1 - N, 2 + M, -1, 2 + M + 1, E, 2 + M + 1 + E, Mant, order, tag_exp, tag_man, tag_zero, tag_sign);
}
// Branch<nbits> (:117-166): a node's array = itself, its 0-subtree, its 1-subtree.  With more than 16 nodes, node n lives
// in cell index * groups + n / 16, place n % 16.
template <class S> __device__ __forceinline__ void bz_tree (S& s, uint32_t prior, int groups, unsigned off, unsigned data, int nbits, int tag) {
  const uint32_t index = prior & 0x7ffffffu;
  int cur_group = 0;
  for (int n = nbits; n >= 1; n--) {
    const int bit = (data >> (n - 1)) & 1;
    if (groups > 1 && (int) (off >> 4) != cur_group) { cur_group = (int) (off >> 4); s.cell ((prior & 0xf8000000u) | (index * (uint32_t)groups + (off >> 4))); }
    s.dec ((int) (off & 15u), bit, tag);
    off += bit ? 1u + ((1u << (n - 1)) - 1u) : 1u;
  }
}
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
template <class S> __device__ __forceinline__ void binarize (S& s, uint32_t prior, int value, int kind, int pad) {
  enum { T_LDC = 17, T_CRDC = 18, T_LAC_0_EOB = 19, T_LAC_N_EOB = 24, T_CRAC_EOB = 29 };
  const int table = (int) (prior >> 27);
  const uint32_t index = prior & 0x7ffffffu;
  switch (kind) {
  case LH264_SYM_LUMA_DC: case LH264_SYM_CHROMA_DC: {      // IntPrior<3,4>: exponent 0..2, mantissa 3..6, zero 7, sign 8
    const int t = kind == LH264_SYM_LUMA_DC ? T_LDC : T_CRDC;
    s.cell (LH264_PRIOR (kind == LH264_SYM_LUMA_DC ? LH264_TB_LDC : LH264_TB_CDC, prior));
    bz_int (s, value, 7, 8, 0, 3, 3, 4, 0, t, t, t, t);
    break; }
  case LH264_SYM_NZ4: case LH264_SYM_NZ8: {                 // UnsignedIntPrior<3,4>
    const int t = nz_tag (prior, pad);
    s.cell (LH264_PRIOR (kind == LH264_SYM_NZ4 ? LH264_TB_NZ4 : LH264_TB_NZ8, prior));
    bz_int (s, value, 7, -1, 0, 3, 3, 4, 0, t, t, t, t);
    break; }
  case LH264_SYM_AC4: case LH264_SYM_AC8: {                 // UEGkIntPrior<14,4,2,4,0>; tags by colour / first scan position (encode4x4)
    const int base = ac_tag_base (prior, kind, pad);
    s.touch (base + 2);                                      // encode4x4 bills to tag(..._EXP): the stream exists from then on
    s.cell (LH264_PRIOR (kind == LH264_SYM_AC4 ? LH264_TB_AC4 : LH264_TB_AC8, prior));
    bz_uegk (s, value, 14, 4, 2, 4, 0, base + 2, base + 3, base + 1, base + 4);
    break; }
  case LH264_SYM_BIT:
    s.cell (prior);
    s.dec (0, value != 0, pad);
    break;
  case LH264_SYM_RAW:
    for (int i = 0; i < (int)prior; i++) s.dec (0xff, (value >> ((int)prior - 1 - i)) & 1, pad);
    break;
  case LH264_SYM_MVD:                                       // UEGkIntPrior<9,4,3,4,3>
    s.cell (prior);
    bz_uegk (s, value, 9, 4, 3, 4, 3, pad, pad, pad, pad);
    break;
  case LH264_SYM_TREE: {
    int nbits = 4, groups = 1;
    if (table == LH264_TB_SKIPRUN) { nbits = 9; groups = 32; } else if (table == LH264_TB_SUBMB) { nbits = 8; groups = 16; }
    else if (table == LH264_TB_CBPC) nbits = 2;
    s.cell ((prior & 0xf8000000u) | (index * (uint32_t)groups));        // the cell of the tree's first 16 nodes
    bz_tree (s, prior, groups, 0, (unsigned) (uint16_t)value, nbits, pad);
    break; }
  case LH264_SYM_POW2: {                                    // emitBitsZeroToPow2Inclusive<nbits>: priors[0], then the tree in priors[1..]
    const bool qpl = table == LH264_TB_QPL;
    const int groups = qpl ? 8 : 1;
    const unsigned preferred = qpl ? 0u : index, data = (unsigned) (uint16_t)value;
    s.cell ((prior & 0xf8000000u) | (index * (uint32_t)groups));
    s.dec (0, data != preferred, pad);
    if (data != preferred) bz_tree (s, prior, groups, 1, data > preferred ? data - 1u : data, qpl ? 7 : 3, pad);
    break; }
  default: break;
  }
}

// ---- how many decisions a symbol becomes, per tag, without walking its binarisation (same cases as binarize above) -------------
struct SymCount { int n, s0, s1, s2, s3, n0, n1, n2, n3, tch; };      // at most four tags (slots; -1: unused), tch: tag brought into existence
// the tail of emitInt behind its zero flag / sign: exponent (unary) and mantissa decisions of data >= 1 (after the sign)
__device__ __forceinline__ void cnt_int_tail (int data, int order, int& ne, int& nm) {
  data--;
  const int l2 = 31 - __clz (1 + (data >> order));
  ne += l2 + 1; nm += l2 + order;
}
__device__ __forceinline__ SymCount sym_count (uint32_t prior, int value, int kind, int pad) {
  enum { T_LDC = 17, T_CRDC = 18, T_LAC_0_EOB = 19, T_LAC_N_EOB = 24, T_CRAC_EOB = 29 };
  SymCount c; c.n = 0; c.s0 = c.s1 = c.s2 = c.s3 = -1; c.n0 = c.n1 = c.n2 = c.n3 = 0; c.tch = -1;
  const int table = (int) (prior >> 27);
  switch (kind) {
  case LH264_SYM_LUMA_DC: case LH264_SYM_CHROMA_DC: case LH264_SYM_NZ4: case LH264_SYM_NZ8: {
    const bool dc = kind == LH264_SYM_LUMA_DC || kind == LH264_SYM_CHROMA_DC;
    const int t = kind == LH264_SYM_LUMA_DC ? T_LDC : kind == LH264_SYM_CHROMA_DC ? T_CRDC : nz_tag (prior, pad);
    int n = 1, ne = 0, nm = 0;                                   // the zero flag
    if (value != 0) { if (dc) n++; cnt_int_tail (value < 0 ? -value : value, 0, ne, nm); }      // sign (DC only), exponent, mantissa
    c.n = n + ne + nm; c.s0 = t; c.n0 = c.n;
    break; }
  case LH264_SYM_AC4: case LH264_SYM_AC8: {
    const int base = ac_tag_base (prior, kind, pad);
    c.tch = base + 2;
    int nz = 1, ns = 0, ne = 0, nm = 0;
    if (value != 0) {
      ns = 1;
      const int u = (value < 0 ? -value : value) - 1;
      nm = u >= 14 ? 14 : u + 1;
      if (u >= 14) { nz++; if (u - 14 != 0) cnt_int_tail (u - 14, 0, ne, nm); }
    }
    c.n = nz + ns + ne + nm;
    c.s0 = base + 1; c.n0 = nz;
    if (ns) { c.s1 = base + 4; c.n1 = ns; }
    if (nm) { c.s2 = base + 3; c.n2 = nm; }
    if (ne) { c.s3 = base + 2; c.n3 = ne; }
    break; }
  case LH264_SYM_BIT: c.n = 1; break;
  case LH264_SYM_RAW: c.n = (int)prior > 0 ? (int)prior : 0; break;
  case LH264_SYM_MVD: {                                         // UEGk<9,4,3,4,3>
    int n = 1, ne = 0, nm = 0;
    if (value != 0) {
      n++;
      const int u = (value < 0 ? -value : value) - 1;
      nm = u >= 9 ? 9 : u + 1;
      if (u >= 9) { n++; if (u - 9 != 0) cnt_int_tail (u - 9, 3, ne, nm); }
    }
    c.n = n + ne + nm;
    break; }
  case LH264_SYM_TREE: c.n = table == LH264_TB_SKIPRUN ? 9 : table == LH264_TB_SUBMB ? 8 : table == LH264_TB_CBPC ? 2 : 4; break;
  case LH264_SYM_POW2: {
    const bool qpl = table == LH264_TB_QPL;
    const unsigned preferred = qpl ? 0u : (prior & 0x7ffffffu), data = (unsigned) (uint16_t)value;
    c.n = 1 + (data != preferred ? (qpl ? 7 : 3) : 0);
    break; }
  default: break;
  }
  if (kind >= LH264_SYM_TREE && c.n > 0) { c.s0 = tag_slot (pad); c.n0 = c.n; }      // the host's symbols name their tag
  else { if (c.s0 >= 0) c.s0 = tag_slot (c.s0); if (c.s1 >= 0) c.s1 = tag_slot (c.s1); if (c.s2 >= 0) c.s2 = tag_slot (c.s2); if (c.s3 >= 0) c.s3 = tag_slot (c.s3); }
  if (c.tch >= 0) c.tch = tag_slot (c.tch);
  return c;
}

// a decision word (64 bits): the low dword is the key of the prior's cell (LH264_PRIOR form; 0 for a raw bit), the high dword
// bits 0..3 the place in the cell, bit 4 the bit, bits 5..10 the tag slot, bit 31 "raw bit" (coded with TEST_PROB)
struct EmitSink {
  GLB uint64_t* D; uint32_t pos, key;
  __device__ __forceinline__ void touch (int) {}
  __device__ __forceinline__ void cell (uint32_t k) { key = k; }
  __device__ __forceinline__ void dec (int j, int bit, int tag) {
    const uint32_t t = (uint32_t)tag_slot (tag) << 5 | (uint32_t) (bit & 1) << 4;
    const bool raw = (j & 0xff) == 0xff;
    D[pos++] = raw ? (uint64_t) (0x80000000u | t) << 32 : ((uint64_t) (t | (uint32_t) (j & 15)) << 32 | key);
  }
};

// ---- segments: the unit of the parallel binarisation -----------------------------------------------------------------------------
// A segment = up to CODER_SEG consecutive macroblocks of one picture.  Its symbols in coding order are, macroblock after macroblock,
// the host list with the coefficient symbols in place of the marker; the workgroup of a segment lays that order out once in LDS
// (where each macroblock's symbols start, where its marker is) and then walks the symbols 64 per wave step whatever macroblock they
// belong to - a wave per macroblock would idle most lanes on the many macroblocks with a handful of symbols.
#define CODER_SEG LH264_CODER_SEG_MBS
struct SegLds {
  uint32_t hoff[CODER_SEG + 1];      // host symbols of macroblock k start here (offsets into the picture's list)
  uint32_t sbase[CODER_SEG + 1];     // symbols of the segment before macroblock k, in coding order
  uint16_t mc[CODER_SEG];            // coefficient symbols of macroblock k
  uint16_t p[CODER_SEG];             // position of the marker in macroblock k's host list (0xffff: none)
  uint32_t wsum[4];
  uint32_t cnt[4][LH264_N_TAG_SLOTS + 2];      // per wave (the symbols of a step mostly count towards the same few tags)
};
struct Seg { const lh264_code_job_t* J; int job, k0, n; uint32_t total; };
// which picture and which macroblocks block `b` works on: seg0[] = segments before picture j
__device__ __forceinline__ bool seg_locate (const lh264_code_job_t* jobs, const uint32_t* seg0, int n_jobs, uint32_t b, Seg& S) {
  if (b >= seg0[n_jobs]) return false;
  uint32_t lo = 0, hi = (uint32_t)n_jobs;              // largest j with seg0[j] <= b
  while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (seg0[mid] <= b) lo = mid; else hi = mid; }
  S.job = (int)lo; S.J = jobs + lo;
  S.k0 = (int) (b - seg0[lo]) * CODER_SEG;
  S.n = min (CODER_SEG, S.J->n_mbs - S.k0);
  return S.n > 0;
}
// lay the segment out (all 256 threads); afterwards L.sbase[S.n] = S.total symbols
__device__ __forceinline__ void seg_layout (LDS SegLds& L, Seg& S, int tid) {
  const GLB uint32_t* off = glb<const uint32_t> (S.J->syn_off_dev) + S.k0;
  const GLB uint16_t* cn = glb<const uint16_t> (S.J->ctx_n_syms_dev) + S.k0;
  if (tid <= S.n) L.hoff[tid] = off[tid];
  if (tid < S.n) { L.mc[tid] = cn[tid]; L.p[tid] = 0xffffu; }
  __syncthreads();
  // the markers: every host symbol of the segment is looked at once
  const GLB uint64_t* hs = glb<const uint64_t> (S.J->syn_syms_dev);
  const uint32_t h0 = L.hoff[0], h1 = L.hoff[S.n];
  for (uint32_t h = h0 + (uint32_t)tid; h < h1; h += 256u) {
    if (((hs[h] >> 48) & 0xffull) == (unsigned long long)LH264_SYM_SPLICE) {
      uint32_t lo = 0, hi = (uint32_t)S.n;               // the macroblock whose list holds position h
      while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (L.hoff[mid] <= h) lo = mid; else hi = mid; }
      L.p[lo] = (uint16_t) (h - L.hoff[lo]);
    }
  }
  __syncthreads();
  // symbols per macroblock, running sum (the segment has at most 256 macroblocks: one per thread)
  uint32_t v = 0;
  if (tid < S.n) { const uint32_t nh = L.hoff[tid + 1] - L.hoff[tid]; v = L.p[tid] != 0xffffu ? nh - 1u + L.mc[tid] : nh; }
  const uint32_t incl = (uint32_t)wave_scan_add ((int)v);
  if ((tid & 63) == 63) L.wsum[tid >> 6] = incl;
  __syncthreads();
  uint32_t before = 0;
  for (int w = 0; w < (tid >> 6); w++) before += L.wsum[w];
  if (tid < S.n) L.sbase[tid] = before + incl - v;
  if (tid == 255) L.sbase[S.n] = before + incl;
  __syncthreads();
  S.total = L.sbase[S.n];
}
// where the coefficient symbols of macroblock k of a picture start (in symbols behind ctx_syms_dev): its fixed slot, or - compact
// layout - the picture's first symbol in the pool + the macroblock's offset
__device__ __forceinline__ size_t ctx_sym_at (const lh264_code_job_t* J, int k) {
  return J->ctx_sym_off_dev ? (size_t)*glb<const unsigned long long> (J->ctx_sym_base_dev) + glb<const uint32_t> (J->ctx_sym_off_dev)[k] : (size_t)k * LH264_CTX_MAX_SYMS;
}
// symbol s of the segment (coding order)
__device__ __forceinline__ uint64_t seg_symbol (const LDS SegLds& L, const Seg& S, uint32_t s) {
  uint32_t lo = 0, hi = (uint32_t)S.n;                   // the macroblock that holds symbol s
  while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (L.sbase[mid] <= s) lo = mid; else hi = mid; }
  const uint32_t i = s - L.sbase[lo], p = L.p[lo], mc = p != 0xffffu ? L.mc[lo] : 0u;
  const GLB uint64_t* hs = glb<const uint64_t> (S.J->syn_syms_dev) + L.hoff[lo];
  if (i < p || p == 0xffffu) return hs[i];
  if (i < p + mc) return glb<const uint64_t> (S.J->ctx_syms_dev)[ctx_sym_at (S.J, S.k0 + (int)lo) + (i - p)];
  return hs[i - mc + 1u];
}

// ---- kernel 0: segments before each picture; which stream a picture belongs to ----------------------------------------------------
__global__ void __launch_bounds__ (CODER_ONE_WG)
coder_jobs_kernel (const lh264_code_job_t* __restrict__ jobs, const int32_t* __restrict__ chain_first, int n_jobs, int n_chains,
                   uint32_t* __restrict__ seg0, uint32_t* __restrict__ job_chain, uint32_t* __restrict__ chain_info) {
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
    if (j < n_jobs) seg0[j] = before + (uint32_t) (incl - v);
    __syncthreads();
    if (tid == CODER_ONE_WG - 1) carry = before + (uint32_t)incl;
    __syncthreads();
  }
  if (tid == 0) seg0[n_jobs] = carry;
  for (int c = tid; c < n_chains; c += CODER_ONE_WG) {
    for (int j = chain_first[c]; j < chain_first[c + 1]; j++) job_chain[j] = (uint32_t)c;
    chain_info[(size_t)c * LH264_CODER_INFO_WORDS + LH264_CODER_INFO_STATUS] = 0;
    for (int q = 90; q < 96; q++) chain_info[(size_t)c * LH264_CODER_INFO_WORDS + q] = 0;
  }
}

// ---- kernel 1: decisions per tag of every segment ---------------------------------------------------------------------------------
// seg_cnt[segment][0 .. LH264_N_TAG_SLOTS-1] decisions per tag slot (bit 31: the segment brings the tag's stream into existence),
// [LH264_N_TAG_SLOTS] all decisions
__global__ void __launch_bounds__ (256)
coder_count_kernel (const lh264_code_job_t* __restrict__ jobs, const uint32_t* __restrict__ seg0, int n_jobs, uint32_t* __restrict__ seg_cnt) {
  __shared__ SegLds Lg;
  LDS SegLds& L = * (LDS SegLds*) (uintptr_t) (uint32_t) (uintptr_t)&Lg;
  Seg S;
  if (!seg_locate (jobs, seg0, n_jobs, blockIdx.x, S)) return;
  const int tid = threadIdx.x;
  for (int i = tid; i < 4 * (LH264_N_TAG_SLOTS + 2); i += 256) (&L.cnt[0][0])[i] = 0;
  seg_layout (L, S, tid);
  LDS uint32_t* cw = L.cnt[tid >> 6];
  uint32_t tot = 0;
  for (uint32_t s0 = 0; s0 < S.total; s0 += 256u) {
    const uint32_t s = s0 + (uint32_t)tid;
    if (s < S.total) {
      const uint64_t sym = seg_symbol (L, S, s);
      const uint32_t hi = (uint32_t) (sym >> 32);
      const SymCount c = sym_count ((uint32_t)sym, (int) (int16_t) (hi & 0xffffu), (int) ((hi >> 16) & 0xffu), (int) (hi >> 24));
      if (c.s0 >= 0) __hip_atomic_fetch_add (&cw[c.s0], (uint32_t)c.n0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (c.s1 >= 0) __hip_atomic_fetch_add (&cw[c.s1], (uint32_t)c.n1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (c.s2 >= 0) __hip_atomic_fetch_add (&cw[c.s2], (uint32_t)c.n2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (c.s3 >= 0) __hip_atomic_fetch_add (&cw[c.s3], (uint32_t)c.n3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (c.tch >= 0) __hip_atomic_fetch_or (&cw[c.tch], 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      tot += (uint32_t)c.n;
    }
  }
  tot = (uint32_t)__builtin_amdgcn_readlane (wave_scan_add ((int)tot), 63);
  if ((tid & 63) == 0) cw[LH264_N_TAG_SLOTS] = tot;
  __syncthreads();
  if (tid <= LH264_N_TAG_SLOTS) {
    const uint32_t a = L.cnt[0][tid], b = L.cnt[1][tid], c = L.cnt[2][tid], d = L.cnt[3][tid];
    seg_cnt[(size_t)blockIdx.x * LH264_CODER_CNT_STRIDE + tid] = ((a + b + c + d) & 0x7fffffffu) | ((a | b | c | d) & 0x80000000u);
  }
}

// ---- kernel 2: per stream, where each segment's decisions start and the size of every tag's list ---------------------------------------
__global__ void __launch_bounds__ (64)
coder_scan_kernel (const uint32_t* __restrict__ seg0, const int32_t* __restrict__ chain_first, const uint32_t* __restrict__ seg_cnt,
                   uint32_t* __restrict__ seg_doff, uint32_t* __restrict__ chain_info, int n_chains) {
  const int c = blockIdx.x, lane = threadIdx.x;
  if (c >= n_chains) return;
  const size_t m0 = seg0[chain_first[c]], m1 = seg0[chain_first[c + 1]];
  uint32_t acc = 0, touched = 0;
  bool big = false;
  const int t = lane <= LH264_N_TAG_SLOTS ? lane : LH264_N_TAG_SLOTS;
  const GLB uint32_t* p = glb<const uint32_t> (seg_cnt) + t;
  for (size_t g = m0; g < m1; g++) {
    const uint32_t v = p[g * LH264_CODER_CNT_STRIDE];
    if (lane == LH264_N_TAG_SLOTS) { seg_doff[g] = acc; big = big || acc + v < acc; acc += v; }
    else { acc += v & 0x7fffffffu; touched |= v >> 31; }
  }
  uint32_t* I = chain_info + (size_t)c * LH264_CODER_INFO_WORDS;
  // tag lists are padded to 8 entries (16 bytes): the coding kernel reads them 16 bytes at a time
  const uint32_t mine = lane < LH264_N_TAG_SLOTS ? ((acc + 7u) & ~7u) : 0u;
  const uint32_t incl = (uint32_t)wave_scan_add ((int)mine);
  if (lane < LH264_N_TAG_SLOTS) { I[LH264_CODER_INFO_TAGBASE + lane] = incl - mine; I[LH264_CODER_INFO_TAGCNT + lane] = acc; }
  const unsigned long long tm = __ballot (lane < LH264_N_TAG_SLOTS && touched != 0);
  if (lane == LH264_N_TAG_SLOTS) { I[LH264_CODER_INFO_NDEC] = acc; I[LH264_CODER_INFO_TOUCH] = (uint32_t)tm; I[LH264_CODER_INFO_TOUCH + 1] = (uint32_t) (tm >> 32); }
  if (lane == 63) I[LH264_CODER_INFO_NQ] = incl;
  if (__ballot (big) && lane == 0) atomicOr (&I[LH264_CODER_INFO_STATUS], (uint32_t)LH264_CODER_ST_COUNT);      // more than 2^32 decisions in a stream
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

// ---- kernel 4: the decision words, in coding order -------------------------------------------------------------------------
// The four waves of a segment's workgroup take a quarter of its symbols each: first how many decisions the quarter makes, then, behind
// a barrier, the words from where the quarters before it end.
__global__ void __launch_bounds__ (256)
coder_emit_kernel (const lh264_code_job_t* __restrict__ jobs, const uint32_t* __restrict__ seg0, const uint32_t* __restrict__ job_chain,
                   int n_jobs, const uint32_t* __restrict__ seg_doff, const uint32_t* __restrict__ chain_info, uint64_t* __restrict__ D) {
  __shared__ SegLds Lg;
  __shared__ uint32_t qtot[4];
  LDS SegLds& L = * (LDS SegLds*) (uintptr_t) (uint32_t) (uintptr_t)&Lg;
  Seg S;
  if (!seg_locate (jobs, seg0, n_jobs, blockIdx.x, S)) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  seg_layout (L, S, tid);
  const uint32_t* I = chain_info + (size_t)job_chain[S.job] * LH264_CODER_INFO_WORDS;
  const unsigned long long dbase = ((unsigned long long)I[LH264_CODER_INFO_DBASE] | (unsigned long long)I[LH264_CODER_INFO_DBASE + 1] << 32) + seg_doff[blockIdx.x];
  const uint32_t per = (S.total + 3u) >> 2, s_lo = min ((uint32_t)wave * per, S.total), s_hi = min (s_lo + per, S.total);
  uint32_t mine = 0;
  for (uint32_t s0 = s_lo; s0 < s_hi; s0 += 64u) {
    const uint32_t s = s0 + (uint32_t)lane;
    if (s < s_hi) {
      const uint64_t sym = seg_symbol (L, S, s);
      const uint32_t hi = (uint32_t) (sym >> 32);
      mine += (uint32_t)sym_count ((uint32_t)sym, (int) (int16_t) (hi & 0xffffu), (int) ((hi >> 16) & 0xffu), (int) (hi >> 24)).n;
    }
  }
  const uint32_t wtot = (uint32_t)__builtin_amdgcn_readlane (wave_scan_add ((int)mine), 63);
  if (lane == 0) qtot[wave] = wtot;
  __syncthreads();
  uint32_t running = 0;
  for (int w = 0; w < wave; w++) running += qtot[w];
  for (uint32_t s0 = s_lo; s0 < s_hi; s0 += 64u) {
    const uint32_t s = s0 + (uint32_t)lane;
    uint64_t sym = 0; uint32_t hi = 0; int n = 0;
    if (s < s_hi) {
      sym = seg_symbol (L, S, s);
      hi = (uint32_t) (sym >> 32);
      n = sym_count ((uint32_t)sym, (int) (int16_t) (hi & 0xffffu), (int) ((hi >> 16) & 0xffu), (int) (hi >> 24)).n;
    }
    const int incl = wave_scan_add (n);
    if (n > 0) {
      EmitSink es; es.D = glb<uint64_t> (D) + dbase; es.pos = running + (uint32_t) (incl - n); es.key = 0;
      binarize (es, (uint32_t)sym, (int) (int16_t) (hi & 0xffffu), (int) ((hi >> 16) & 0xffu), (int) (hi >> 24));
    }
    running += (uint32_t)__builtin_amdgcn_readlane (incl, 63);
  }
}

// ---- kernel 5: the probability every decision is coded with -----------------------------------------------------------------
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

// The counters of a DynProb as the resolve kernel keeps them: c0 | c1 << 10, NOT yet halved when their sum has reached 513 - the
// reference computes the next probability before it halves (DynProb::update, :101-113), so the probability of the next decision always
// follows from the stored pair, and the halving is done by the next reader.  An entry of the LDS cache (and of the spill table in HBM) is
// 64 bits: counters in bits 0..19, the DynProb's key (cell key << 4 | place, 36 bits) in bits 20..55, bit 63 set.  All zero = free.
#define RS_WAVES LH264_CODER_RESOLVE_WAVES
// diagnostic build (-DLH264_CODER_DEBUG): shader-clock stamps of the resolve kernel's phases, summed over the waves of a stream into
// chain_info words 90..94 (in units of 1024 cycles), reported in the unused length slots 35..39
#ifdef LH264_CODER_DEBUG
#define RS_STAMP_DECL unsigned long long st_t = __builtin_amdgcn_s_memtime(), st_acc[6] = {0, 0, 0, 0, 0, 0};
#define RS_STAMP(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_t; st_t = t_; }
#define RS_STAMP_FLUSH if (lane == 0) for (int q_ = 0; q_ < 6; q_++) atomicAdd (&chain_info[(size_t)chain * LH264_CODER_INFO_WORDS + 90 + q_], (uint32_t) (st_acc[q_] >> 10));
#else
#define RS_STAMP_DECL
#define RS_STAMP(i)
#define RS_STAMP_FLUSH
#endif
#define RS_LOG2_BUCKETS 11
#define RS_SLOTS (4 << RS_LOG2_BUCKETS)        // DynProbs in the LDS cache: 8192
#define RS_CHECK 4               // the fill of the cache is looked at every RS_CHECK workgroup steps (that many steps insert <= 2048)
#ifndef RS_FLUSH
#define RS_FLUSH 4600            // everything goes to the spill table and the cache starts over above this many
#endif
struct ResolveLds {
  unsigned long long ent[RS_SLOTS];
  uint32_t cursor[LH264_N_TAG_SLOTS];
  uint32_t test_prob;            // TEST_PROB: the DynProb shared by the raw bits of all tags
  uint32_t ticket;               // the next wave step allowed into the serial section
  uint32_t nres;                 // entries in the cache
  uint32_t flush_step;           // the last workgroup step at whose end the cache is (was) flushed
  uint32_t scratch[RS_WAVES][64];
  uint32_t ring[RS_WAVES][3][2][64];       // per wave: the decision words of three future rounds (low dwords, high dwords), filled by LDS-DMA
};
__device__ __forceinline__ unsigned long long rs_key (uint32_t lo, uint32_t hi) { return (unsigned long long)lo << 4 | (unsigned long long) (hi & 15u); }
#define RS_ENT_KEY(e) (((e) >> 20) & 0xfffffffffull)
#define RS_ENT_MAKE(key, st) (0x8000000000000000ull | (unsigned long long) (key) << 20 | (unsigned long long) (st))
__device__ __forceinline__ uint32_t rs_hash (unsigned long long key) { return ((uint32_t)key * 0x9E3779B1u) ^ ((uint32_t) (key >> 32) * 0x85EBCA6Bu); }

// the spill table: open addressing over the stream's `hash_cells_dev` memory (zero-filled by the caller), entries as above.  Only this
// workgroup touches it; its accesses go to L2 (agent scope), never through this CU's L1.
__device__ __forceinline__ bool spill_put (GLB unsigned long long* T, uint32_t tmask, unsigned long long key, uint32_t st) {
  const unsigned long long val = RS_ENT_MAKE (key, st);
  uint32_t h = rs_hash (key) >> 8;
  for (uint32_t tries = 0; tries <= tmask; tries++, h++) {
    GLB unsigned long long* p = T + (h & tmask);
    unsigned long long cur = __hip_atomic_load (p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == 0ull) { unsigned long long expect = 0ull; if (__hip_atomic_compare_exchange_strong (p, &expect, val, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return true; cur = expect; }
    if (RS_ENT_KEY (cur) == key) { __hip_atomic_store (p, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return true; }      // (a key sits in one cache entry: nobody else writes it now)
    if (tries >= 4096u) break;
  }
  return false;                  // the table is (as good as) full
}
__device__ __forceinline__ uint32_t spill_get (const GLB unsigned long long* T, uint32_t tmask, unsigned long long key, unsigned long long first) {
  uint32_t h = rs_hash (key) >> 8;
  unsigned long long cur = first;                   // the entry at the key's home slot, requested a step ago
  for (uint32_t tries = 0; tries <= tmask; tries++) {
    if (cur == 0ull) return 0u;
    if (RS_ENT_KEY (cur) == key) return (uint32_t)cur & 0xfffffu;
    h++;
    cur = __hip_atomic_load (T + (h & tmask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return 0u;
}

struct SlotRef { int idx; bool miss, inserted; unsigned long long first; };
// the cache entry of the DynProb of decision word (lo, hi): inserted with fresh counters if absent; if counters may have been spilled,
// the inserting lane asks the spill table (answer taken by rs_land)
__device__ __forceinline__ void rs_lookup (LDS ResolveLds& S, const GLB unsigned long long* T, uint32_t tmask, uint32_t lo, uint32_t hi, bool valid, bool spilled, SlotRef& R) {
  R.idx = 0; R.miss = false; R.inserted = false; R.first = 0ull;
  if (valid && !(hi & 0x80000000u)) {
    const unsigned long long key = rs_key (lo, hi), fresh = RS_ENT_MAKE (key, 0u);
    // buckets of four entries (32 bytes, read at once): a probe looks at a whole bucket, so the longest probe sequence among the 64
    // lanes of a wave - which is what the wave waits for - stays short
    uint32_t bkt = rs_hash (key) >> (32 - RS_LOG2_BUCKETS), h = 0;
    for (int tries = 0; tries < RS_SLOTS / 4; tries++, bkt++) {      // (the flush policy keeps the cache at most 7/8 full: bounded anyway)
      bkt &= RS_SLOTS / 4 - 1;
      const LDS u32x4* bp = (const LDS u32x4*)&S.ent[4u * bkt];
      const u32x4 a = * (volatile const LDS u32x4*)bp, b = * (volatile const LDS u32x4*) (bp + 1);
      const unsigned long long e[4] = {(unsigned long long)a.x | (unsigned long long)a.y << 32, (unsigned long long)a.z | (unsigned long long)a.w << 32,
                                       (unsigned long long)b.x | (unsigned long long)b.y << 32, (unsigned long long)b.z | (unsigned long long)b.w << 32};
      int found = -1, empty = -1;
#pragma unroll
      for (int q = 3; q >= 0; q--) { if (e[q] == 0ull) empty = q; if (e[q] != 0ull && RS_ENT_KEY (e[q]) == key) found = q; }
      if (found >= 0) { h = 4u * bkt + (uint32_t)found; break; }
      if (empty >= 0) {
        // take the first free entry of the bucket; if another lane is quicker, look at the bucket again (it may have put this very key there)
        unsigned long long expect = 0ull;
        if (__hip_atomic_compare_exchange_strong (&S.ent[4u * bkt + (uint32_t)empty], &expect, fresh, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
          h = 4u * bkt + (uint32_t)empty; R.miss = true; break;
        }
        bkt--;
      }
    }
    R.idx = (int)h;
    R.inserted = R.miss;
    R.miss = R.miss && spilled;                                // before the first flush a new DynProb is simply fresh
    if (R.miss) R.first = __hip_atomic_load (T + ((rs_hash (key) >> 8) & tmask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// the counters the spill table holds for an entry inserted by rs_lookup go into the entry (it has not been used yet)
__device__ __forceinline__ void rs_land (LDS ResolveLds& S, const GLB unsigned long long* T, uint32_t tmask, uint32_t lo, uint32_t hi, SlotRef& R) {
  if (R.miss) {
    const uint32_t st = spill_get (T, tmask, rs_key (lo, hi), R.first);
    volatile LDS uint32_t* p = (volatile LDS uint32_t*)&S.ent[R.idx];
    if (st) *p = (*p & 0xfff00000u) | st;
    R.miss = false;
  }
}
__device__ __forceinline__ void rs_count (LDS ResolveLds& S, bool inserted, int lane) {
  const unsigned long long mm = __ballot (inserted);
  if (mm && lane == 0) __hip_atomic_fetch_add (&S.nres, (uint32_t)__popcll (mm), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// LDS traffic of this wave done, then the workgroup barrier (not __syncthreads: it would also wait for the memory operations in flight)
__device__ __forceinline__ void rs_barrier() {
  asm volatile ("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile ("" ::: "memory");
}

// Workgroup step `it` = rounds it * RS_WAVES .. + RS_WAVES - 1 of 64 decisions, one per wave, through the ticketed serial section in
// round order; nothing else synchronises the waves of a step.  Software pipeline per wave: the decision words are fetched five steps
// ahead; the cache entries of a round are looked up two steps before the round is resolved (a DynProb that is not in the cache is
// inserted then, and its spilled counters requested); the answer goes into the entry one step later, BEFORE the wave's own turn of
// that step - every round that uses the entry comes later in ticket order than that turn, so it sees the counters.
__global__ void __launch_bounds__ (RS_WAVES * 64)
coder_resolve_kernel (const lh264_code_stream_t* __restrict__ streams, uint32_t* __restrict__ chain_info, const uint64_t* __restrict__ D,
                      uint16_t* __restrict__ Q, int n_chains) {
  __shared__ ResolveLds Sg;
  LDS ResolveLds& S = * (LDS ResolveLds*) (uintptr_t) (uint32_t) (uintptr_t)&Sg;
  const int chain = blockIdx.x;
  if (chain >= n_chains) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const uint32_t* I = chain_info + (size_t)chain * LH264_CODER_INFO_WORDS;
  const uint32_t n = I[LH264_CODER_INFO_NDEC];
  const GLB uint64_t* Dc = glb<const uint64_t> (D) + ((unsigned long long)I[LH264_CODER_INFO_DBASE] | (unsigned long long)I[LH264_CODER_INFO_DBASE + 1] << 32);
  GLB uint16_t* Qc = glb<uint16_t> (Q) + ((unsigned long long)I[LH264_CODER_INFO_QBASE] | (unsigned long long)I[LH264_CODER_INFO_QBASE + 1] << 32);
  GLB unsigned long long* T = glb<unsigned long long> (streams[chain].hash_cells_dev);
  const uint32_t hcap = streams[chain].hash_cap;
  if (hcap == 0u || (hcap & (hcap - 1u)) != 0u || hcap > (1u << 20)) {      // (the whole workgroup: nothing of this stream is coded)
    if (tid == 0) atomicOr (&chain_info[(size_t)chain * LH264_CODER_INFO_WORDS + LH264_CODER_INFO_STATUS], (uint32_t)LH264_CODER_ST_TABLE_FULL);
    return;
  }
  const uint32_t tmask = hcap * 8u - 1u;            // hash_cap cells of 64 bytes = 8 entries each
  for (int i = tid; i < RS_SLOTS; i += RS_WAVES * 64) S.ent[i] = 0ull;
  if (tid < LH264_N_TAG_SLOTS) S.cursor[tid] = I[LH264_CODER_INFO_TAGBASE + tid];
  if (tid == 0) { S.test_prob = 0; S.ticket = 0; S.nres = 0; S.flush_step = 0xffffffffu; }
  __syncthreads();
  const uint32_t n_rounds = (n + 63u) >> 6;
  const uint32_t n_iter = (n_rounds + RS_WAVES - 1) / RS_WAVES;
  uint32_t r = (uint32_t)wave;
  auto fetch = [&] (uint32_t round) -> uint64_t { const uint32_t i = round * 64u + (uint32_t)lane; return i < n ? Dc[i] : 0ull; };
  auto is_valid = [&] (uint32_t round) -> bool { return round * 64u + (uint32_t)lane < n; };
  // Decision words travel HBM -> LDS by LDS-DMA, three workgroup steps ahead of their use, so that neither the compiler's nor
  // this code's waits for OTHER memory operations ever have to wait for a word that was only just requested.  (Always issued - the
  // index is clamped - so that the counted wait below is right in the last steps too.)
  const uint32_t my_ring = (uint32_t) (uintptr_t)&S.ring[wave][0][0][0];
  auto dma = [&] (uint32_t round, uint32_t slot) {
    uint32_t i = round * 64u + (uint32_t)lane;
    if (i >= n) i = n - 1u;
    const GLB uint32_t* src = (const GLB uint32_t*) (Dc + i);
    // (as asm statements: hipcc would make every later LDS read wait for a load it knows to write LDS; M0 = the LDS address of lane 0's
    // dword, saved and restored inside the statement)
    const uint32_t dst = (uint32_t)uniform ((int) (my_ring + slot * 512u));
    uint32_t keep;
    asm volatile ("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dword %2, off\n\ts_mov_b32 m0, %0"
                  : "=&s"(keep) : "v"(src), "v"(src + 1), "s"(dst), "s"(dst + 256u) : "memory");
  };
  if (n == 0u) return;
  // rounds r (resolved in this step), r + W (looked up; a spilled answer is taken in this step), r + 2W (looked up in this step)
  uint64_t w0 = fetch (r), w1 = fetch (r + RS_WAVES), w2 = fetch (r + 2 * RS_WAVES);
  dma (r + 3 * RS_WAVES, 0u); dma (r + 4 * RS_WAVES, 1u); dma (r + 5 * RS_WAVES, 2u);
  bool spilled = false;
  SlotRef e0, e1;
  rs_lookup (S, T, tmask, (uint32_t)w0, (uint32_t) (w0 >> 32), is_valid (r), false, e0);
  rs_lookup (S, T, tmask, (uint32_t)w1, (uint32_t) (w1 >> 32), is_valid (r + RS_WAVES), false, e1);
  rs_count (S, e0.inserted, lane); rs_count (S, e1.inserted, lane);
  __syncthreads();
  RS_STAMP_DECL
  bool pend_ok = false; uint32_t pend_q = 0, pend_v = 0;
  for (uint32_t it = 0; it < n_iter; it++) {
    const bool v_cur = is_valid (r), round_ok = r < n_rounds;
    // every RS_CHECK steps, and only while every wave of the step has a round
    const bool check = (it % RS_CHECK) == RS_CHECK - 1 && (it + 1u) * RS_WAVES <= n_rounds;
    bool do_flush = false;
    if (pend_ok) Qc[pend_q] = (uint16_t)pend_v;          // the list entry of the round resolved in the last step
    pend_ok = false;
    const uint32_t w_hi = (uint32_t) (w0 >> 32);
    // the words of round r + 3W: requested three steps ago; the four requests behind them may still be under way
    asm volatile ("s_waitcnt vmcnt(4)" ::: "memory");
    const uint32_t slot = it % 3u;
    const uint64_t w3 = (uint64_t) * (volatile LDS uint32_t*) (uintptr_t) (my_ring + slot * 512u + (uint32_t)lane * 4u) |
                        (uint64_t) * (volatile LDS uint32_t*) (uintptr_t) (my_ring + slot * 512u + 256u + (uint32_t)lane * 4u) << 32;
    RS_STAMP (5)
    // ---- the entries of the round two steps ahead, first thing: spilled counters that are requested here are taken a whole step later
    // (inserting does not disturb the rounds in flight: they use entries they found earlier) ----------------------------------------------
    SlotRef e2;
    rs_lookup (S, T, tmask, (uint32_t)w2, (uint32_t) (w2 >> 32), is_valid (r + 2 * RS_WAVES), spilled, e2);
    rs_count (S, e2.inserted, lane);
    RS_STAMP (4)
    // ---- what does not depend on the adaptive state: who shares my DynProb, who shares my tag ----------------------------------------
    const bool raw = (w_hi & 0x80000000u) != 0;
    const int bit = (int) ((w_hi >> 4) & 1u), tag = (int) ((w_hi >> 5) & 63u);
    const unsigned long long valid = __ballot (v_cur);
    const uint32_t dkey = raw ? 0x3fffu : (uint32_t)e0.idx;                  // 13 bits of cache entry; 0x3fff: TEST_PROB
    uint32_t slo, shi, tlo, thi;
    wave_match<14> (dkey, valid, slo, shi);
    wave_match<6> ((uint32_t)tag, valid, tlo, thi);
    const unsigned long long zm = __ballot (v_cur && bit == 0);
    const int rank = below (slo, shi), nn = __popc (slo) + __popc (shi);
    const int z = below (slo & (uint32_t)zm, shi & (uint32_t) (zm >> 32));
    const int trank = below (tlo, thi), tn = __popc (tlo) + __popc (thi);
    const int head = slo ? __ffs ((int)slo) - 1 : 32 + __ffs ((int)shi) - 1;
    // spilled counters requested a step ago: into the entries now (before this wave's turn, see above)
    rs_land (S, T, tmask, (uint32_t)w1, (uint32_t) (w1 >> 32), e1);
    RS_STAMP (0)
    // ---- the serial section: counters in, counters out -------------------------------------------------------------------------------
    if (round_ok) {
      volatile LDS uint32_t* sp = raw ? &S.test_prob : (volatile LDS uint32_t*)&S.ent[e0.idx];
      volatile LDS uint32_t* cp = &S.cursor[tag];
      {   // (the spin is bounded so that a broken hand-off ends as a wrong result with a status bit, not as a hung GPU)
        volatile LDS uint32_t* tk = &S.ticket;
        uint32_t spins = 0;
        // the wave that waits for the ticket is the stream's critical path: it polls at raised priority (measured: 5.8 -> 5.45 ms)
        __builtin_amdgcn_s_setprio (2);
        for (; *tk != r && spins < (1u << 20); spins++) { }
        if (spins >= (1u << 20) && lane == 0) atomicOr (&chain_info[(size_t)chain * LH264_CODER_INFO_WORDS + LH264_CODER_INFO_STATUS], (uint32_t)LH264_CODER_ST_HANDOFF);
      }
      uint32_t st = 0, cb = 0;
      if (v_cur) { st = *sp; cb = *cp; }
      __builtin_amdgcn_s_setprio (3);                // the waves behind this one are waiting for exactly this section
      asm volatile ("" ::: "memory");
      RS_STAMP (1)
      if (check) {
        // is the cache filling up?  The first wave of the step decides inside its turn, the others read the decision inside theirs
        // (later in ticket order): no barrier unless there is something to flush
        volatile LDS uint32_t* fs = &S.flush_step;
        if (wave == 0 && * (volatile LDS uint32_t*)&S.nres > RS_FLUSH) *fs = it;
        do_flush = *fs == it;
      }
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
        LDS uint32_t* sc = S.scratch[wave];
        if (v_cur && rank == t + 1) sc[head] = (uint32_t)z;                 // zeros among ranks 0..t
        __builtin_amdgcn_wave_barrier();
        if (v_cur && rank > t) {
          const uint32_t zt = * (volatile LDS uint32_t*)&sc[head];
          const uint32_t h0 = (f0 + zt + 1u) >> 1, h1 = (f1 + (uint32_t) (t + 1) - zt + 1u) >> 1;
          b0 = h0 + ((uint32_t)z - zt); b1 = h1 + ((uint32_t) (rank - z) - ((uint32_t) (t + 1) - zt));
          if (rank > t + 1) { a0 = b0; a1 = b1; }
        }
        __builtin_amdgcn_wave_barrier();
      }
      if (v_cur && rank == nn - 1) *sp = (st & 0xfff00000u) | (b0 + (uint32_t) (bit ^ 1)) | (b1 + (uint32_t)bit) << 10;
      if (v_cur && trank == tn - 1) *cp = cb + (uint32_t)tn;
      asm volatile ("" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      if (lane == 0) { volatile LDS uint32_t* tk = &S.ticket; *tk = r + 1u; }
      __builtin_amdgcn_s_setprio (0);
      RS_STAMP (2)
      // ---- afterwards: the probability, and the entry of the tag's list ------------------------------------------------------------------
      // (stored at the top of the next step: a store as the youngest memory operation at the loop's end would make the compiler's
      // wait for the spill-table answers wait for the store as well)
      // the list entry carries the probability of the bit that occurred (what the bool coder multiplies with, see bc_step)
      const uint32_t prob = dp_ratio (a0, a1);
      pend_ok = v_cur; pend_q = cb + (uint32_t)trank; pend_v = (bit ? 256u - prob : prob) << 1 | (uint32_t)bit;
    }
    RS_STAMP (3)
    {
      if (do_flush) {
        // every DynProb to the spill table, then the cache starts over with the entries of the two rounds in flight
        rs_land (S, T, tmask, (uint32_t)w1, (uint32_t) (w1 >> 32), e1);
        rs_land (S, T, tmask, (uint32_t)w2, (uint32_t) (w2 >> 32), e2);
        rs_barrier();
        for (int i = tid; i < RS_SLOTS; i += RS_WAVES * 64) {
          const unsigned long long e = S.ent[i];
          if (e) {
            if (!spill_put (T, tmask, RS_ENT_KEY (e), (uint32_t)e & 0xfffffu))
              atomicOr (&chain_info[(size_t)chain * LH264_CODER_INFO_WORDS + LH264_CODER_INFO_STATUS], (uint32_t)LH264_CODER_ST_TABLE_FULL);
            S.ent[i] = 0ull;
          }
        }
        if (tid == 0) S.nres = 0;
        asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
        rs_barrier();
        spilled = true;
        rs_lookup (S, T, tmask, (uint32_t)w1, (uint32_t) (w1 >> 32), is_valid (r + RS_WAVES), true, e1);
        rs_lookup (S, T, tmask, (uint32_t)w2, (uint32_t) (w2 >> 32), is_valid (r + 2 * RS_WAVES), true, e2);
        rs_count (S, e1.inserted, lane); rs_count (S, e2.inserted, lane);
        rs_land (S, T, tmask, (uint32_t)w1, (uint32_t) (w1 >> 32), e1);
        rs_land (S, T, tmask, (uint32_t)w2, (uint32_t) (w2 >> 32), e2);
        rs_barrier();
      }
    }
    asm volatile ("s_waitcnt lgkmcnt(0)" ::: "memory");      // the ring slot has been read: it takes the words of round r + 6W
    dma (r + 6 * RS_WAVES, slot);
    r += RS_WAVES;
    w0 = w1; w1 = w2; w2 = is_valid (r + 2 * RS_WAVES) ? w3 : 0ull;
    e0 = e1; e1 = e2;
  }
  if (pend_ok) Qc[pend_q] = (uint16_t)pend_v;
  RS_STAMP_FLUSH
}

}  // namespace lh264sw
