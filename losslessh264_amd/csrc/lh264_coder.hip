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
//   coder_range_kernel    one lane per (stream, tag): the bool coder's range recurrence over that tag's list       (serial per tag)
//   coder_accum_kernel, coder_bytes_kernel   the bool coder's `low`: addends summed per output byte position by chunks of the list,
//                         then the carries and the bytes                                                            (parallel)
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
    const int t = ((prior / 27u) % 3u) ? T_CRAC_EOB : T_LAC_0_EOB;
    s.cell (LH264_PRIOR (kind == LH264_SYM_NZ4 ? LH264_TB_NZ4 : LH264_TB_NZ8, prior));
    bz_int (s, value, 7, -1, 0, 3, 3, 4, 0, t, t, t, t);
    break; }
  case LH264_SYM_AC4: case LH264_SYM_AC8: {                 // UEGkIntPrior<14,4,2,4,0>; tags by colour / first scan position (encode4x4)
    const uint32_t nco = kind == LH264_SYM_AC4 ? 16u : 64u;
    const uint32_t outer = prior / 3125u;
    const int emitted = (int) (outer % nco), color = (int) ((outer / nco) % 3u), code = (int) ((outer / nco / 3u) % 16u);
    const int first = color == 0 && emitted == 0 && code != 1;
    const int base = color ? T_CRAC_EOB : (first ? T_LAC_0_EOB : T_LAC_N_EOB);
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
    const int t = kind == LH264_SYM_LUMA_DC ? T_LDC : kind == LH264_SYM_CHROMA_DC ? T_CRDC : (((prior / 27u) % 3u) ? T_CRAC_EOB : T_LAC_0_EOB);
    int n = 1, ne = 0, nm = 0;                                   // the zero flag
    if (value != 0) { if (dc) n++; cnt_int_tail (value < 0 ? -value : value, 0, ne, nm); }      // sign (DC only), exponent, mantissa
    c.n = n + ne + nm; c.s0 = t; c.n0 = c.n;
    break; }
  case LH264_SYM_AC4: case LH264_SYM_AC8: {
    const uint32_t nco = kind == LH264_SYM_AC4 ? 16u : 64u;
    const uint32_t outer = prior / 3125u;
    const int emitted = (int) (outer % nco), color = (int) ((outer / nco) % 3u), code = (int) ((outer / nco / 3u) % 16u);
    const int first = color == 0 && emitted == 0 && code != 1;
    const int base = color ? T_CRAC_EOB : (first ? T_LAC_0_EOB : T_LAC_N_EOB);
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
// symbol s of the segment (coding order)
__device__ __forceinline__ uint64_t seg_symbol (const LDS SegLds& L, const Seg& S, uint32_t s) {
  uint32_t lo = 0, hi = (uint32_t)S.n;                   // the macroblock that holds symbol s
  while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (L.sbase[mid] <= s) lo = mid; else hi = mid; }
  const uint32_t i = s - L.sbase[lo], p = L.p[lo], mc = p != 0xffffu ? L.mc[lo] : 0u;
  const GLB uint64_t* hs = glb<const uint64_t> (S.J->syn_syms_dev) + L.hoff[lo];
  if (i < p || p == 0xffffu) return hs[i];
  if (i < p + mc) return glb<const uint64_t> (S.J->ctx_syms_dev)[(size_t) (S.k0 + (int)lo) * LH264_CTX_MAX_SYMS + (i - p)];
  return hs[i - mc + 1u];
}

// ---- kernel 0: segments before each picture; which stream a picture belongs to ----------------------------------------------------
__global__ void __launch_bounds__ (1024)
coder_jobs_kernel (const lh264_code_job_t* __restrict__ jobs, const int32_t* __restrict__ chain_first, int n_jobs, int n_chains,
                   uint32_t* __restrict__ seg0, uint32_t* __restrict__ job_chain, uint32_t* __restrict__ chain_info) {
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int j0 = 0; j0 < n_jobs; j0 += 1024) {
    const int j = j0 + tid;
    const int v = j < n_jobs ? (max (jobs[j].n_mbs, 0) + CODER_SEG - 1) / CODER_SEG : 0;
    const int incl = wave_scan_add (v);
    if (lane == 63) wsum[wave] = (uint32_t)incl;
    __syncthreads();
    uint32_t before = carry;
    for (int w = 0; w < wave; w++) before += wsum[w];
    if (j < n_jobs) seg0[j] = before + (uint32_t) (incl - v);
    __syncthreads();
    if (tid == 1023) carry = before + (uint32_t)incl;
    __syncthreads();
  }
  if (tid == 0) seg0[n_jobs] = carry;
  for (int c = tid; c < n_chains; c += 1024) {
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
__global__ void __launch_bounds__ (1024)
coder_bases_kernel (uint32_t* __restrict__ chain_info, int n_chains, unsigned long long* __restrict__ totals) {
  __shared__ unsigned long long wd[16], wq[16];
  __shared__ unsigned long long cd, cq;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) { cd = 0; cq = 0; }
  __syncthreads();
  for (int c0 = 0; c0 < n_chains; c0 += 1024) {
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
    if (tid == 1023) { cd = bd + sd; cq = bq + sq; }
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
  const uint32_t tmask = streams[chain].hash_cap * 8u - 1u;            // hash_cap cells of 64 bytes = 8 entries each
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

// ---- kernels 6..9: the libvpx bool coder (vpx_writer, bitwriter.h:35-105; vpx_stop_encode bitwriter.cpp:17-37) ----------------
// vpx_write keeps (low, range, count).  `range` depends only on the decisions so far - a 7-bit state that does not forget where it
// started (two start values stay apart for hundreds of decisions: measured), so its recurrence is walked once, serially, per
// (stream, tag); but that walk is all that is serial.  What vpx_write finally writes is one big number: every decision with bit 1
// adds its `split` (8 bits) at the bit position given by the shifts before it, and carries run towards the first byte.  So:
//   coder_chunkmap_kernel  chunks (CODE_CHUNK decisions) per (stream, tag) pair, running sum
//   coder_range_kernel     one lane per pair: the range recurrence alone (no low, no bytes) - mad, two shifts, count-leading-zeros,
//                          shift per decision -, noting range and bit position at the start of every chunk
//   coder_accum_kernel     one lane per chunk: the recurrence again from the noted state; the addends go into 32-bit sums per output
//                          byte position (a window in registers, atomic adds when it moves on)
//   coder_bytes_kernel     one lane per pair: carries from the last byte to the first, the bytes, vpx_stop_encode's padding byte
// The 32 "stop" decisions (bit 0, probability 128) are decisions n .. n+31 of a list.
// A list entry is e = q << 1 | bit with q = the probability of the bit that occurred, in 1/256 (bit 0: the decision's probability p,
// bit 1: 256 - p; written so by the resolve kernel).  With x = range - 1:  bit 0: what is left is split = (x p + 256) >> 8;  bit 1:
// range - split = x - (x p >> 8) = (x (256 - p) + 255) >> 8.  One multiply-add for both: rq = range q + (256 - bit - q) < 2^16,
// r = rq >> 8, and the normalising shift (vpx_norm[r]) is the number of leading zeros of rq << 16.
#define CODE_CHUNK ((uint32_t)LH264_CODER_CODE_CHUNK)
#define CODE_STOP_ENTRY (128u << 1)
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
__global__ void __launch_bounds__ (1024)
coder_chunkmap_kernel (const uint32_t* __restrict__ chain_info, int n_pairs, uint32_t* __restrict__ pair_chunk0) {
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int p0 = 0; p0 < n_pairs; p0 += 1024) {
    const int p = p0 + tid;
    uint32_t v = 0;
    if (p < n_pairs) {
      const uint32_t chain = (uint32_t)p / LH264_N_TAG_SLOTS, slot = (uint32_t)p % LH264_N_TAG_SLOTS;
      const uint32_t* I = chain_info + (size_t)chain * LH264_CODER_INFO_WORDS;
      const uint32_t n = slot < 35u ? I[LH264_CODER_INFO_TAGCNT + slot] : 0u;
      const unsigned long long tm = (unsigned long long)I[LH264_CODER_INFO_TOUCH] | (unsigned long long)I[LH264_CODER_INFO_TOUCH + 1] << 32;
      if (slot < 35u && (n > 0u || ((tm >> slot) & 1ull))) v = (n + 32u + CODE_CHUNK - 1u) / CODE_CHUNK;
    }
    const uint32_t incl = (uint32_t)wave_scan_add ((int)v);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t before = carry;
    for (int w = 0; w < wave; w++) before += wsum[w];
    if (p < n_pairs) pair_chunk0[p] = before + incl - v;
    __syncthreads();
    if (tid == 1023) carry = before + incl;
    __syncthreads();
  }
  if (tid == 0) pair_chunk0[n_pairs] = carry;
}

// kernel 7: per pair, the range recurrence over the whole list; per chunk {range | pair << 8, bits shifted out} at its first decision.
// A wave = one tag slot of 64 consecutive streams: lists of about the same length in its lanes.  (Two lists per lane, their steps
// interleaved in one instruction stream, were slower - 4.2 vs 3.1 ms: a lone wave is bound by the number of instructions it issues,
// not by the latency of the recurrence.)
// Loads the compiler does not know about: a loop that reads ahead through ordinary loads gets "s_waitcnt vmcnt(0)" at its head
// (the loop-carried loads are tracked conservatively), i.e. one memory round trip per iteration.  Here the load and the wait are
// written out: `code_ld16` starts a 16-byte load into v (tied: the register is not renamed, nothing copies it while the data is
// under way), `code_wait<N>` waits until at most N younger vector-memory operations are outstanding and is v's first reader.
__device__ __forceinline__ void code_ld16 (u32x4& v, const GLB u32x4* p) { asm volatile ("global_load_dwordx4 %0, %1, off" : "+v" (v) : "v" (p) : "memory"); }
template <int N> __device__ __forceinline__ void code_wait (u32x4& v) { asm volatile ("s_waitcnt vmcnt(%1)" : "+v" (v) : "n" (N) : "memory"); }

struct RangeWalk {
  GLB uint32_t* rec; uint32_t range, pos, g0, pair;
  __device__ __forceinline__ void note (uint32_t i) { const size_t g = g0 + i / CODE_CHUNK; rec[2 * g] = range | pair << 8; rec[2 * g + 1] = pos; }
  // one whole piece (8 decisions): no test per decision; a chunk starts on a piece
  __device__ __forceinline__ void piece (const u32x4 v, uint32_t c) {
    if ((c & (CODE_CHUNK / 8 - 1u)) == 0u) note (c * 8u);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 8; q++) pos += code_step (range, w[q >> 1] >> (16 * (q & 1))).shift;
  }
};
#define CODE_AHEAD 8u        // 16-byte pieces of the list under way per lane
__global__ void __launch_bounds__ (64)
coder_range_kernel (const uint32_t* __restrict__ chain_info, const uint16_t* __restrict__ Q, const uint32_t* __restrict__ pair_chunk0, int n_chains,
                    int groups, uint32_t* __restrict__ chunk_rec, uint32_t* __restrict__ pair_bits) {
  const uint32_t slot = blockIdx.x / (uint32_t)groups, chain = (blockIdx.x % (uint32_t)groups) * 64u + threadIdx.x;
  if (chain >= (uint32_t)n_chains) return;
  RangeWalk A;
  A.pair = chain * LH264_N_TAG_SLOTS + slot; A.range = 255u; A.pos = 0; A.rec = glb<uint32_t> (chunk_rec);
  A.g0 = pair_chunk0[A.pair];
  if (pair_chunk0[A.pair + 1] != A.g0) {
    const PairInfo P = pair_info (chain_info, Q, A.pair);
    const uint32_t n = P.n, pieces = (n + 7u) >> 3, whole = n >> 3, last = pieces ? pieces - 1u : 0u;
    const GLB u32x4* src = (const GLB u32x4*)P.list;
    // eight separate registers quadruples (an array would be one aggregate that the compiler copies around while loads are under way)
    u32x4 b0 = {0u, 0u, 0u, 0u}, b1 = b0, b2 = b0, b3 = b0, b4 = b0, b5 = b0, b6 = b0, b7 = b0;
#define CODE_FIRST(B, J) if (pieces) code_ld16 (B, src + min ((uint32_t) (J), last));
    CODE_FIRST (b0, 0) CODE_FIRST (b1, 1) CODE_FIRST (b2, 2) CODE_FIRST (b3, 3) CODE_FIRST (b4, 4) CODE_FIRST (b5, 5) CODE_FIRST (b6, 6) CODE_FIRST (b7, 7)
#undef CODE_FIRST
    uint32_t c = 0;
    // CODE_AHEAD loads are under way all the time (behind the end of the list the last piece is read again), so the oldest one has
    // arrived when at most CODE_AHEAD - 1 are outstanding
#define CODE_TURN(B, J) code_wait<CODE_AHEAD - 1> (B); A.piece (B, c + (J)); code_ld16 (B, src + min (c + (J) + CODE_AHEAD, last));
    for (; c + CODE_AHEAD <= whole; c += CODE_AHEAD) {
      CODE_TURN (b0, 0u) CODE_TURN (b1, 1u) CODE_TURN (b2, 2u) CODE_TURN (b3, 3u) CODE_TURN (b4, 4u) CODE_TURN (b5, 5u) CODE_TURN (b6, 6u) CODE_TURN (b7, 7u)
    }
#undef CODE_TURN
    code_wait<0> (b0); code_wait<0> (b1); code_wait<0> (b2); code_wait<0> (b3); code_wait<0> (b4); code_wait<0> (b5); code_wait<0> (b6); code_wait<0> (b7);
    // fewer than CODE_AHEAD whole pieces are left: register j holds piece min (c + j, last); then the last, partial piece
    const uint32_t left = whole - c;
    u32x4 pv = b0;
#define CODE_LAST(B, J) if ((J) < left) A.piece (B, c + (J)); if (left == (J)) pv = B;
    CODE_LAST (b0, 0u) CODE_LAST (b1, 1u) CODE_LAST (b2, 2u) CODE_LAST (b3, 3u) CODE_LAST (b4, 4u) CODE_LAST (b5, 5u) CODE_LAST (b6, 6u) CODE_LAST (b7, 7u)
#undef CODE_LAST
    const uint32_t w[4] = {pv.x, pv.y, pv.z, pv.w};
    for (uint32_t i = whole * 8u; i < P.total; i++) {
      if ((i & (CODE_CHUNK - 1u)) == 0u) A.note (i);
      const uint32_t j = i & 7u;
      const uint32_t wj = (j & 4u) ? ((j & 2u) ? w[3] : w[2]) : ((j & 2u) ? w[1] : w[0]);
      A.pos += code_step (A.range, i < n ? wj >> (16u * (j & 1u)) : CODE_STOP_ENTRY).shift;
    }
  }
  pair_bits[A.pair] = A.pos;                            // all the bits the list shifts out
}

// kernel 8: the addends of every chunk into the sums of the output byte positions.  A position takes addends from the decisions that
// start in its byte or in the byte before; the positions strictly inside a chunk's span of bits belong to that chunk alone and are
// stored, the two at either end are shared with the neighbouring chunks and added atomically (the sums start out as zero).
__global__ void __launch_bounds__ (256)
coder_accum_kernel (const uint32_t* __restrict__ chain_info, const uint16_t* __restrict__ Q, const uint32_t* __restrict__ pair_chunk0, int n_pairs,
                    const uint32_t* __restrict__ chunk_rec, const uint32_t* __restrict__ pair_bits, uint32_t* __restrict__ acc) {
  const uint32_t g = blockIdx.x * 256u + threadIdx.x;
  if (g >= pair_chunk0[n_pairs]) return;
  uint32_t range = chunk_rec[2 * (size_t)g], t = chunk_rec[2 * (size_t)g + 1];
  const uint32_t pair = range >> 8;
  range &= 0xffu;
  const PairInfo P = pair_info (chain_info, Q, pair);
  const uint32_t c = g - pair_chunk0[pair], i0 = c * CODE_CHUNK, i1 = min (i0 + CODE_CHUNK, P.total);
  const uint32_t t_end = g + 1u < pair_chunk0[pair + 1] ? chunk_rec[2 * (size_t)g + 3] : pair_bits[pair];     // where the next chunk starts
  const uint32_t own_lo = (t >> 3) + 2u, own_hi = t_end >> 3;            // positions own_lo .. own_hi - 1 are this chunk's alone
  GLB uint32_t* A = glb<uint32_t> (acc) + P.acc0;
  auto put = [&] (uint32_t kpos, uint32_t v) {
    if (v == 0u) return;
#ifndef LH264_ABL_CODE_NOATOMIC     // timing ablation only (wrong output)
    if (kpos >= own_lo && kpos < own_hi) A[kpos] = v;
    else atomicAdd ((uint32_t*) (uintptr_t) (A + kpos), v);
#endif
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
  // (ordinary loads here: the atomics and stores between them are vector-memory operations too, a counted wait would wait for
  // nearly everything; with thousands of chunks per SIMD the latency is covered by other waves)
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  u32x4 p0 = c0 < c1 ? src[c0] : zero4, p1 = c0 + 1 < c1 ? src[c0 + 1] : zero4, p2 = c0 + 2 < c1 ? src[c0 + 2] : zero4, p3 = c0 + 3 < c1 ? src[c0 + 3] : zero4;
  for (uint32_t cc = c0; cc < c1; cc++) {
    const u32x4 v = p0;
    p0 = p1; p1 = p2; p2 = p3; p3 = cc + 4 < c1 ? src[cc + 4] : zero4;
    const uint32_t wd[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 8; q++) one (wd[q >> 1] >> (16 * (q & 1)));
  }
  ListReader L; L.init (P.list, P.n);
  for (uint32_t i = max (i0, c1 * 8u); i < i1; i++) one (L.at (i));
  put (kb, w >> 8);
  put (kb + 1u, w & 0xffu);
}

// kernel 9: carries, bytes, lengths.  vpx_write puts a byte out whenever 8 more bits have been shifted out beyond the first 24:
// bytes = (bits - 24) / 8 + 1; byte k is byte position k of the sum.  vpx_stop_encode appends a zero byte behind a last byte 110xxxxx.
__global__ void __launch_bounds__ (64)
coder_bytes_kernel (const lh264_code_stream_t* __restrict__ streams, const uint32_t* __restrict__ chain_info, const uint16_t* __restrict__ Q,
                    const uint32_t* __restrict__ pair_bits, const uint32_t* __restrict__ acc, int n_pairs) {
  const uint32_t pair = blockIdx.x * 64u + threadIdx.x;
  if (pair >= (uint32_t)n_pairs) return;
  const uint32_t chain = pair / LH264_N_TAG_SLOTS, slot = pair % LH264_N_TAG_SLOTS;
  if (slot >= 35u) return;                             // tag slots that do not exist (their lengths are cleared by the status kernel)
  const lh264_code_stream_t* S = streams + chain;
  GLB uint32_t* lens = glb<uint32_t> (S->out_len_dev);
  const PairInfo P = pair_info (chain_info, Q, pair);
  if (!P.used) { lens[slot] = 0; return; }
  const uint32_t bits = pair_bits[pair], cap = S->out_cap;
  uint32_t nbytes = bits >= 24u ? ((bits - 24u) >> 3) + 1u : 0u;
  const GLB uint32_t* A = glb<const uint32_t> (acc) + P.acc0;
  GLB uint8_t* o = glb<uint8_t> (S->out_dev) + (size_t)slot * cap;
  // sums behind the last byte that is put out carry into it as well
  uint32_t carry = 0;
  const uint32_t last = (bits >> 3) + 2u;                      // no addend lies behind this position
  for (uint32_t k = last; k + 1u > nbytes; k--) carry = (A[k] + carry) >> 8;           // positions last .. nbytes
  uint32_t final_byte = 0;
  uint32_t k = nbytes;
  for (; (k & 3u) != 0u; ) {                          // down to a multiple of four positions, then four sums (and four bytes) at a time
    k--;
    const uint32_t v = A[k] + carry;
    carry = v >> 8;
    if (k < cap) o[k] = (uint8_t)v;
    if (k == nbytes - 1u) final_byte = v & 0xffu;
  }
  const bool wide = (((uintptr_t)o) & 3u) == 0u;
  // four positions (and four bytes) per step; the sums are read four steps ahead of the carry chain (code_ld16 / code_wait: the loop
  // would otherwise wait for every load it has just issued)
  auto four = [&] (const u32x4 q) {
    k -= 4u;
    const uint32_t v3 = q.w + carry, v2 = q.z + (v3 >> 8), v1 = q.y + (v2 >> 8), v0 = q.x + (v1 >> 8);
    carry = v0 >> 8;
    if (k + 3u == nbytes - 1u) final_byte = v3 & 0xffu;
    const uint32_t word = (v0 & 0xffu) | (v1 & 0xffu) << 8 | (v2 & 0xffu) << 16 | (v3 & 0xffu) << 24;
    if (wide && k + 3u < cap) * (GLB uint32_t*) (o + k) = word;
    else { if (k < cap) o[k] = (uint8_t)v0; if (k + 1u < cap) o[k + 1] = (uint8_t)v1; if (k + 2u < cap) o[k + 2] = (uint8_t)v2; if (k + 3u < cap) o[k + 3] = (uint8_t)v3; }
  };
  u32x4 f0 = {0u, 0u, 0u, 0u}, f1 = f0, f2 = f0, f3 = f0;
  code_ld16 (f0, (const GLB u32x4*) (A + (k >= 4u ? k - 4u : 0u)));
  code_ld16 (f1, (const GLB u32x4*) (A + (k >= 8u ? k - 8u : 0u)));
  code_ld16 (f2, (const GLB u32x4*) (A + (k >= 12u ? k - 12u : 0u)));
  code_ld16 (f3, (const GLB u32x4*) (A + (k >= 16u ? k - 16u : 0u)));
#define CODE_TURN(F) code_wait<3> (F); four (F); code_ld16 (F, (const GLB u32x4*) (A + (k >= 16u ? k - 16u : 0u)));
  while (k >= 16u) { CODE_TURN (f0) CODE_TURN (f1) CODE_TURN (f2) CODE_TURN (f3) }
#undef CODE_TURN
  code_wait<0> (f0); code_wait<0> (f1); code_wait<0> (f2); code_wait<0> (f3);
  if (k >= 4u) four (f0);
  if (k >= 4u) four (f1);
  if (k >= 4u) four (f2);
  if (nbytes > 0u && (final_byte & 0xe0u) == 0xc0u) { if (nbytes < cap) o[nbytes] = 0; nbytes++; }
  lens[slot] = nbytes;
  if (nbytes > cap) atomicOr ((uint32_t*) (uintptr_t) (lens + LH264_N_TAG_SLOTS), (uint32_t)LH264_CODER_ST_OUT_FULL);
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
