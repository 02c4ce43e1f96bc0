// lh264_kernels.hip - gfx950 (MI355X, wave64) kernels of the decode-reconstruct hot path.
//
// One workgroup per chain of frames (the frames of one stream, in decode order), one wavefront
// per macroblock row (row r is served by wave r % NW), row r running at least two macroblocks
// behind row r-1 -- the raster dependency of intra prediction and of the in-loop filter
// (reference loop order: WelsTargetSliceConstruction decode_slice.cpp:110-206, WelsDeblockingFilterSlice
// deblocking.cpp:872-934).  ALL hand-offs between rows go through LDS:
//   * `line`  : the unfiltered bottom sample row of each macroblock row  (intra prediction neighbours)
//   * `fline` : the filtered bottom 4 (chroma 2) sample rows              (deblocking neighbours)
//   * progress counters
// so no wave ever waits for its HBM stores inside a frame.  Per macroblock a wave
//   0. stages the prefetched record (and the record of the MB above) into LDS and prefetches the next one,
//   1. copies the unfiltered neighbours into its private LDS tile,
//   2. inverse-transforms the 384 coefficients (coalesced 768-byte read, prefetched one MB ahead; 4x4
//      butterflies across lane quads with DPP; 8x8 / DC transforms through LDS),
//   3. predicts (intra from the tile, inter = quarter-pel MC straight from the padded reference planes)
//      and adds the residual                                       [RecI*/GetInterPred rec_mb.cpp],
//   4. publishes the unfiltered bottom row / right column,
//   5. deblocks the macroblock's edges inside the tile              [WelsDeblockingMb deblocking.cpp:815],
//   6. writes a 16x16 window displaced by (-4,-3) -- the samples that became final in this step: every
//      output byte is written exactly once -- and publishes the filtered bottom rows.
// Frames of a chain are PIPELINED through the same wavefront: the rows of all frames of the chain form one sequence
// (global row g is served by wave g % NW), there is no barrier between frames.  A wave that finishes a row also pads
// the band of picture rows that just became final (ExpandReferencingPicture, row by row) and publishes a per-wave
// "stored" counter once those HBM stores have completed; a macroblock of the next frame whose motion vectors reach
// into the frame still in flight waits for exactly the rows it reads.  At most two frames are in flight.
//
// Integer work on u8/int16: no MFMA.  Bounds: HBM traffic (1,280 B / intra MB algorithmic) and VALU
// issue; see DESIGN.md.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/lh264.h"

namespace lh264 {

// Explicit address spaces.  HIP pointers are generic by default and hipcc only recovers LDS/global for accesses
// it can trace to a __shared__ variable or a kernel argument; pointers that pass through structs, noinline
// calls or memory become FLAT accesses (slower, and every wait becomes vmcnt(0)&lgkmcnt(0)).
#ifndef LH264_PHASE
#define LH264_PHASE __noinline__
#endif
#ifndef LH264_MIN_WAVES
#define LH264_MIN_WAVES 1
#endif
// diagnostic build (-DLH264_STAMP): cycle stamps per phase, summed over all waves into g_stamps (never in product builds)
#ifdef LH264_STAMP
__device__ unsigned long long g_stamps[16];
#define STAMP_DECL unsigned long long st_t0 = __builtin_amdgcn_s_memtime(); unsigned long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define STAMP(i) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_t0; st_t0 = t_; } while (0)
#define STAMP_FLUSH do { if (lane == 0) for (int i_ = 0; i_ < 12; i_++) atomicAdd (&g_stamps[i_], st_acc[i_]); } while (0)
#else
#define STAMP_DECL
#define STAMP(i) do {} while (0)
#define STAMP_FLUSH do {} while (0)
#endif
#define LDS __attribute__ ((address_space (3)))
#define GLB __attribute__ ((address_space (1)))
typedef int v2i __attribute__ ((ext_vector_type (2)));
__device__ __forceinline__ v2i mk2 (int a, int b) { v2i r; r.x = a; r.y = b; return r; }
template <typename T> __device__ __forceinline__ GLB T* to_glb (const void* p) { return (GLB T*) (uintptr_t)p; }

// ---- tables (H.264 Tables 8-16/8-17; reference copies: deblocking.cpp:89-125) ----------------
__constant__ uint8_t kTables[52 + 52 + 52 * 4] = {
  // alpha
  0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 4, 4, 5, 6, 7, 8, 9, 10, 12, 13,
  15, 17, 20, 22, 25, 28, 32, 36, 40, 45, 50, 56, 63, 71, 80, 90, 101, 113, 127, 144, 162, 182, 203, 226, 255, 255,
  // beta
  0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 6, 6, 7, 7,
  8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 14, 14, 15, 15, 16, 16, 17, 17, 18, 18,
  // tc0[indexA][bS] (bS 0 unused)
  0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
  0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1,
  0, 0, 0, 1, 0, 0, 1, 1, 0, 0, 1, 1, 0, 1, 1, 1, 0, 1, 1, 1, 0, 1, 1, 1, 0, 1, 1, 1, 0, 1, 1, 2, 0, 1, 1, 2, 0, 1, 1, 2,
  0, 1, 1, 2, 0, 1, 2, 3, 0, 1, 2, 3, 0, 2, 2, 3, 0, 2, 2, 4, 0, 2, 3, 4, 0, 2, 3, 4, 0, 3, 3, 5, 0, 3, 4, 6, 0, 3, 4, 6,
  0, 4, 5, 7, 0, 4, 5, 8, 0, 4, 6, 9, 0, 5, 7, 10, 0, 6, 8, 11, 0, 6, 8, 13, 0, 7, 10, 14, 0, 8, 11, 16, 0, 9, 12, 18, 0, 10, 13, 20,
  0, 11, 15, 23, 0, 13, 17, 25
};
#define TAB_ALPHA 0
#define TAB_BETA 52
#define TAB_TC0 104

// ---- workgroup-shared LDS ----------------------------------------------------------------------
struct WgLds {
  int      progress[16];   // per wave: (rows it has started - 1) << 12 | macroblocks finished in its current row
  int      stored[16];     // per wave: index (among its own rows) of the last row whose HBM stores, including the
                           // padding that row is responsible for, have completed; -1 = none
  uint8_t  tab[52 + 52 + 208];
};

// ---- per-wave LDS ------------------------------------------------------------------------------
struct WaveLds {
  uint8_t  T[20 * 32];       // luma tile: rows -4..15, cols -4..27      idx (r+4)*32 + (c+4)
  uint8_t  C[2][10 * 16];    // chroma tiles: rows -2..7, cols -4..11    idx (r+2)*16 + (c+4)
  int16_t  R[384];           // residual, layout [block b (z-order)][row][col] ; chroma blocks 16..23
  uint32_t rec[2][32];       // current / previous (= left) macroblock record, ping-pong
  uint32_t trec[32];         // record of the macroblock above
  uint32_t slc[58];          // cached slice record (lh264_slice_t)
  uint8_t  leftY[16];        // unfiltered right column of the previous MB (intra neighbours)
  uint8_t  leftC[2][8];
  uint32_t lfY[20];          // filtered right 4 columns of the previous MB, rows -4..15
  uint32_t lfC[2][10];       //                                            rows -2..7
  int32_t  mvi[16][4];       // per 4x4 block motion info
  uint8_t  bs[32];           // boundary strengths [dir][edge][segment]
  uint8_t  E[32];            // filtered I8x8 edge
  int32_t  S[64];            // DC transform scratch
  uint64_t refp[LH264_MAX_REFS][3];   // reference plane pointers of this wave's current job
  int32_t  known_prefix;     // every global row <= known_prefix is known to be stored (cache of wait_prefix)
};

__device__ __forceinline__ int tY (int r, int c) { return (r + 4) * 32 + (c + 4); }
__device__ __forceinline__ int tC (int r, int c) { return (r + 2) * 16 + (c + 4); }
__device__ __forceinline__ int clip_u8 (int v) { return min (max (v, 0), 255); }
__device__ __forceinline__ int clip3 (int v, int lo, int hi) { return min (max (v, lo), hi); }
__device__ __forceinline__ int zidx (int bx, int by) { return (bx & 1) | ((by & 1) << 1) | ((bx >> 1) << 2) | ((by >> 1) << 3); }
__device__ __forceinline__ int tab_idx (int v) { return clip3 (v, 0, 51); }
__device__ __forceinline__ int uni (int v) { return __builtin_amdgcn_readfirstlane (v); }

// order LDS traffic between the lanes of one wave.  The DS instructions of a wave execute in issue order, so
// a later ds_read observes an earlier ds_write of any lane; all that is needed is to stop the COMPILER from
// moving memory operations across this point.  (A wavefront-scope fence would do that too, but hipcc lowers
// it to "s_waitcnt vmcnt(0) lgkmcnt(0)", which drains the prefetches and the output stores at every phase.)
__device__ __forceinline__ void wsync() {
  asm volatile ("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

template <int K> __device__ __forceinline__ int quad_bcast (int v) {
  return __builtin_amdgcn_mov_dpp (v, K * 0x55, 0xf, 0xf, true);
}
__device__ __forceinline__ int sext16 (int v) { return (int) (short)v; }

// 4x4 inverse transform over a lane quad: lane r of the quad holds row r (4 coefficients) of the block and
// receives row r of the residual ((x+32)>>6).  Rows first with int16 intermediates, then columns
// (IdctResAddPred_c, decode_mb_aux.cpp:42-77).
__device__ __forceinline__ void idct4x4_quad (int a, int b, int c, int d, int r, int out[4]) {
  const int e0 = a + c, e1 = a - c, e2 = (b >> 1) - d, e3 = b + (d >> 1);
  int t[4] = {sext16 (e0 + e3), sext16 (e1 + e2), sext16 (e1 - e2), sext16 (e0 - e3)};
  const bool outer = (r == 0) || (r == 3);
  const bool plus = r < 2;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int s0 = quad_bcast<0> (t[i]), s1 = quad_bcast<1> (t[i]), s2 = quad_bcast<2> (t[i]), s3 = quad_bcast<3> (t[i]);
    const int f0 = s0 + s2, f1 = s0 - s2, f2 = (s1 >> 1) - s3, f3 = s1 + (s3 >> 1);
    const int A = outer ? f0 : f1, B = outer ? f3 : f2;
    out[i] = (32 + (plus ? A + B : A - B)) >> 6;
  }
}

// one 8-point butterfly of IdctResAddPred8x8_c (decode_mb_aux.cpp:79-167), int16 temporaries
__device__ __forceinline__ void idct8_1d (const int p[8], int o[8]) {
  int a0 = sext16 (p[0] + p[4]), a1 = sext16 (p[0] - p[4]);
  int a2 = sext16 (p[6] - (p[2] >> 1)), a3 = sext16 (p[2] + (p[6] >> 1));
  const int b0 = sext16 (a0 + a3), b2 = sext16 (a1 - a2), b4 = sext16 (a1 + a2), b6 = sext16 (a0 - a3);
  a0 = sext16 (-p[3] + p[5] - p[7] - (p[7] >> 1));
  a1 = sext16 (p[1] + p[7] - p[3] - (p[3] >> 1));
  a2 = sext16 (-p[1] + p[7] + p[5] + (p[5] >> 1));
  a3 = sext16 (p[3] + p[5] + p[1] + (p[1] >> 1));
  const int b1 = sext16 (a0 + (a3 >> 2)), b3 = sext16 (a1 + (a2 >> 2));
  const int b5 = sext16 (a2 - (a1 >> 2)), b7 = sext16 (a3 - (a0 >> 2));
  o[0] = sext16 (b0 + b7); o[1] = sext16 (b2 - b5); o[2] = sext16 (b4 + b3); o[3] = sext16 (b6 + b1);
  o[4] = sext16 (b6 - b1); o[5] = sext16 (b4 - b3); o[6] = sext16 (b2 + b5); o[7] = sext16 (b0 - b7);
}

__device__ __forceinline__ int tap6 (int a, int b, int c, int d, int e, int f) { return a - 5 * b + 20 * c + 20 * d - 5 * e + f; }

__device__ __forceinline__ int sum_xor (int v, int width) {   // butterfly sum inside aligned groups of `width` lanes
  for (int m = 1; m < width; m <<= 1) v += __shfl_xor (v, m);
  return v;
}

// typed views of a macroblock record staged in LDS (offsets == lh264_mb_t)
struct RecView {
  const LDS uint8_t* p;
  __device__ __forceinline__ int mb_type() const { return * (const LDS uint16_t*)p; }
  __device__ __forceinline__ int cbp() const { return p[2]; }
  __device__ __forceinline__ int qp_y() const { return p[3]; }
  __device__ __forceinline__ int qp_c (int i) const { return p[4 + i]; }
  __device__ __forceinline__ int flags() const { return p[6]; }
  __device__ __forceinline__ int intra_avail() const { return p[7]; }
  __device__ __forceinline__ int intra_mode (int i) const { return (int8_t)p[8 + i]; }
  __device__ __forceinline__ int chroma_mode() const { return (int8_t)p[24]; }
  __device__ __forceinline__ int slice_id() const { return * (const LDS uint16_t*) (p + 26); }
  __device__ __forceinline__ int sub_type (int i) const { return p[28 + i]; }
  __device__ __forceinline__ int ref_idx (int i) const { return (int8_t)p[32 + i]; }
  __device__ __forceinline__ int nzc (int i) const { return p[36 + i]; }
  __device__ __forceinline__ int mvx (int i) const { return * (const LDS int16_t*) (p + 60 + 4 * i); }
  __device__ __forceinline__ int mvy (int i) const { return * (const LDS int16_t*) (p + 62 + 4 * i); }
  __device__ __forceinline__ int ref4 (int b) const { return ref_idx (((b >> 3) << 1) + ((b & 3) >> 1)); }
  __device__ __forceinline__ int nz8 (int o) const { return nzc (o) | nzc (o + 1) | nzc (o + 4) | nzc (o + 5); }
};
// cached slice record (offsets == lh264_slice_t)
struct SliceView {
  const LDS uint8_t* p;
  __device__ __forceinline__ int slice_type() const { return p[8]; }
  __device__ __forceinline__ int deblock_idc() const { return p[9]; }
  __device__ __forceinline__ int alpha_off() const { return (int8_t)p[10]; }
  __device__ __forceinline__ int beta_off() const { return (int8_t)p[11]; }
  __device__ __forceinline__ int weighted() const { return p[12]; }
  __device__ __forceinline__ int luma_denom() const { return p[13]; }
  __device__ __forceinline__ int chroma_denom() const { return p[14]; }
  __device__ __forceinline__ int luma_weight (int i) const { return * (const LDS int16_t*) (p + 16 + 2 * i); }
  __device__ __forceinline__ int luma_offset (int i) const { return * (const LDS int16_t*) (p + 48 + 2 * i); }
  __device__ __forceinline__ int chroma_weight (int i, int c) const { return * (const LDS int16_t*) (p + 80 + 4 * i + 2 * c); }
  __device__ __forceinline__ int chroma_offset (int i, int c) const { return * (const LDS int16_t*) (p + 144 + 4 * i + 2 * c); }
  __device__ __forceinline__ int ref_slot (int i) const { return (int8_t)p[208 + i]; }
  __device__ __forceinline__ int luma_dc_weight() const { return p[224]; }
};

struct FrameCtx {
  const GLB lh264_mb_t* mbs; const GLB int16_t* coeffs; const GLB lh264_slice_t* slices;
  GLB uint8_t* dy; GLB uint8_t* du; GLB uint8_t* dv;
  int mb_w, mb_h, sy, sc, flags;
  // the previous job of the chain (the only other frame that can still be in flight)
  uint64_t prev_dy;          // its luma plane pointer (0: none)
  int prev_base, prev_h;     // global index of its row 0, its height in macroblock rows
  int nw;                    // waves in the workgroup
};

// Block until every global row <= g_need has been stored (rows are owned round-robin: the next unfinished row of wave
// w is w + nw * (stored[w] + 1)).
__device__ __forceinline__ void wait_prefix (const LDS WgLds& G, LDS WaveLds& L, int g_need, int nw) {
  if (__builtin_amdgcn_readfirstlane (L.known_prefix) >= g_need) return;
  volatile const LDS int* st = G.stored;
  for (;;) {
    int pre = 0x7fffffff;
    for (int w = 0; w < nw; w++) pre = min (pre, w + nw * (st[w] + 1));
    pre = __builtin_amdgcn_readfirstlane (pre) - 1;
    if (pre >= g_need) { L.known_prefix = pre; break; }
    __builtin_amdgcn_s_sleep (2);
  }
  __builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "workgroup");
}


struct Pref { v2i l, c; uint32_t rec; };   // what is fetched one macroblock ahead (per lane)

__device__ __forceinline__ Pref prefetch_mb (const FrameCtx& F, int k, bool has_top, int lane) {
  Pref p;
  const GLB int16_t* cf = F.coeffs + (size_t)k * 384;
  p.l = * (const GLB v2i*) (cf + 4 * lane);
  p.c = mk2 (0, 0);
  p.rec = 0;
  if (lane < 32) {
    p.c = * (const GLB v2i*) (cf + 256 + 4 * lane);
    p.rec = ((const GLB uint32_t*) (F.mbs + k))[lane];
  } else if (has_top) p.rec = ((const GLB uint32_t*) (F.mbs + k - F.mb_w))[lane - 32];
  return p;
}

// ------------------------------------------------------------------------------------------------
// residual: coefficients (registers) -> R[] in LDS as int16, layout [block][row][col]
// ------------------------------------------------------------------------------------------------
__device__ LH264_PHASE void residual_phase (LDS WaveLds& L, v2i v, v2i cv, int mb_type, int cbp, bool t8, int qp, int dcw, int lane) {
  const int b = lane >> 2, r = lane & 3;
  int res[4] = {0, 0, 0, 0}, cres[4] = {0, 0, 0, 0};
  const bool i16 = mb_type == LH264_MB_I16x16;
  int c0 = sext16 (v.x), c1 = v.x >> 16, c2 = sext16 (v.y), c3 = v.y >> 16;
  int d0 = sext16 (cv.x), d1 = cv.x >> 16, d2 = sext16 (cv.y), d3 = cv.y >> 16;
  const bool have_c = (cbp >> 4) != 0;
  if (i16 || have_c) {
    // DC transforms through LDS (WelsLumaDcDequantIdct decode_slice.cpp:271-311, WelsChromaDcIdct :375-396)
    if (r == 0) { L.S[b] = c0; if (lane < 32) L.S[16 + (lane >> 2)] = d0; }
    wsync();
    if (i16 && lane < 16) {
      const int ox = (lane & 1) | ((lane >> 2) & 1) << 1, oy = ((lane >> 1) & 1) | ((lane >> 3) & 1) << 1;  // z-order -> (x,y)
      int f = 0;
#pragma unroll
      for (int jj = 0; jj < 4; jj++) {
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
          // Hadamard sign patterns: row 0 ++++, 1 ++--, 2 +--+, 3 +-+-
          const int sy_ = (0xA6C0 >> (oy * 4 + jj)) & 1, sx_ = (0xA6C0 >> (ox * 4 + kk)) & 1;
          const int mv = L.S[zidx (kk, jj)];
          f += (sy_ ^ sx_) ? -mv : mv;
        }
      }
      const int na = (int) ((0x12100E0D0B0Aull >> (8 * (qp % 6))) & 0xff);      // 10 11 13 14 16 18
      const int dq = na << (qp / 6);
      const int qmul = dcw == 16 ? dq : ((dcw * dq) >> 4);
      L.S[32 + lane] = sext16 ((f * qmul + 2) >> 2);
    }
    if (have_c && lane >= 16 && lane < 24) {
      const int p = (lane - 16) >> 2, i = lane & 3;
      const int a = L.S[16 + p * 4], bb = L.S[16 + p * 4 + 1], c = L.S[16 + p * 4 + 2], d = L.S[16 + p * 4 + 3];
      const int s0 = a + bb, dd0 = a - bb, s1 = c + d, dd1 = c - d;
      const int o = i == 0 ? s0 + s1 : i == 1 ? dd0 + dd1 : i == 2 ? s0 - s1 : dd0 - dd1;
      L.S[48 + (lane - 16)] = sext16 (o >> 1);
    }
    wsync();
    if (r == 0) { if (i16) c0 = L.S[32 + b]; if (have_c && lane < 32) d0 = L.S[48 + (lane >> 2)]; }
    wsync();
  }
  if (!t8) idct4x4_quad (c0, c1, c2, c3, r, res);
  else {
    // 8x8 transform through LDS: the MB's luma coefficients already sit row-major per 8x8 block
    * (LDS v2i*) (L.R + 4 * lane) = v;
    wsync();
    if (lane < 32) {              // rows
      int p[8], o[8];
#pragma unroll
      for (int i = 0; i < 8; i++) p[i] = L.R[lane * 8 + i];
      idct8_1d (p, o);
#pragma unroll
      for (int i = 0; i < 8; i++) L.R[lane * 8 + i] = (int16_t)o[i];
    }
    wsync();
    if (lane < 32) {              // columns
      const int blk = lane >> 3, col = lane & 7;
      int p[8], o[8];
#pragma unroll
      for (int i = 0; i < 8; i++) p[i] = L.R[blk * 64 + i * 8 + col];
      idct8_1d (p, o);
#pragma unroll
      for (int i = 0; i < 8; i++) L.R[blk * 64 + i * 8 + col] = (int16_t)o[i];
    }
    wsync();
    const int bx = (b & 1) | ((b >> 2) & 1) << 1, by = ((b >> 1) & 1) | ((b >> 3) & 1) << 1;
    const int ly = 4 * by + r, lx0 = 4 * bx;
    const int i8 = (ly >> 3) * 2 + (lx0 >> 3);
#pragma unroll
    for (int i = 0; i < 4; i++) res[i] = (32 + L.R[i8 * 64 + (ly & 7) * 8 + (lx0 & 7) + i]) >> 6;
    wsync();
  }
  if (have_c) idct4x4_quad (d0, d1, d2, d3, r, cres);
  * (LDS v2i*) (L.R + 4 * lane) = mk2 ((res[0] & 0xffff) | (res[1] << 16), (res[2] & 0xffff) | (res[3] << 16));
  if (lane < 32) * (LDS v2i*) (L.R + 256 + 4 * lane) = mk2 ((cres[0] & 0xffff) | (cres[1] << 16), (cres[2] & 0xffff) | (cres[3] << 16));
  wsync();
}

__device__ __forceinline__ void zero_residual (LDS WaveLds& L, int lane) {
  * (LDS v2i*) (L.R + 4 * lane) = mk2 (0, 0);
  if (lane < 32) * (LDS v2i*) (L.R + 256 + 4 * lane) = mk2 (0, 0);
  wsync();
}

// residual of this lane's 4-sample strip (luma layout: lane = 4*b + r ; chroma layout lanes 0..31)
__device__ __forceinline__ void load_res4 (const LDS int16_t* R, int lane, int out[4]) {
  const v2i v = * (const LDS v2i*) (R + 4 * lane);
  out[0] = sext16 (v.x); out[1] = v.x >> 16; out[2] = sext16 (v.y); out[3] = v.y >> 16;
}
__device__ __forceinline__ uint32_t pack_add4 (const int pr[4], const int rs[4]) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) o |= (uint32_t)clip_u8 (pr[i] + rs[i]) << (8 * i);
  return o;
}

// ------------------------------------------------------------------------------------------------
// intra prediction
// ------------------------------------------------------------------------------------------------
// I4x4 directional prediction of one sample from the edge e[0..12] = L3 L2 L1 L0 TL T0..T7
// (get_intra_predictor.cpp:54-380).  hi = 12, or 8 for the *_TOP variants (p[3,-1] replicated).
__device__ __forceinline__ int edge4_load (const LDS uint8_t* T, int bx, int by, int j, int hi) {
  j = clip3 (j, 0, hi);
  const int r = (j < 4) ? (4 * by + 3 - j) : (4 * by - 1);
  const int c = (j < 4) ? (4 * bx - 1) : (4 * bx + j - 5);
  return T[tY (r, c)];
}

__device__ __forceinline__ int pred4_dir (const LDS uint8_t* T, int bx, int by, int mode, int x, int y) {
  int k, three = 1, hi = 12;
  switch (mode) {
  case LH264_I4_DDL_TOP: hi = 8;    // fall through
  case LH264_I4_DDL: k = x + y + 6; break;
  case LH264_I4_DDR: k = 4 + x - y; break;
  case LH264_I4_VR: {
    const int z = 2 * x - y;
    if (z >= 0) { k = 4 + x - (y >> 1); three = z & 1; }
    else if (z == -1) k = 4;
    else k = 5 - y;
    break;
  }
  case LH264_I4_HD: {
    const int z = 2 * y - x;
    if (z >= 0) { k = 4 - (y - (x >> 1)); three = z & 1; if (!three) k -= 1; }
    else if (z == -1) k = 4;
    else k = 3 + x;
    break;
  }
  case LH264_I4_VL_TOP: hi = 8;     // fall through
  case LH264_I4_VL: { const int kk = x + (y >> 1); three = y & 1; k = three ? kk + 6 : kk + 5; break; }
  default: /* HU */ { const int z = x + 2 * y; const int a = z >> 1; three = z & 1; k = 2 - a; break; }
  }
  if (three) {
    const int e0 = edge4_load (T, bx, by, k - 1, hi), e1 = edge4_load (T, bx, by, k, hi), e2 = edge4_load (T, bx, by, k + 1, hi);
    return (e0 + 2 * e1 + e2 + 2) >> 2;
  }
  const int e0 = edge4_load (T, bx, by, k, hi), e1 = edge4_load (T, bx, by, k + 1, hi);
  return (e0 + e1 + 1) >> 1;
}

// RecI4x4Luma rec_mb.cpp:124-157, one lane per sample.  The reference walks the 16 blocks in z-order; a block reads its left, top,
// top-left and - where H.264 calls it available, i.e. where it comes earlier in z-order - top-right neighbour block.  All of those have a
// smaller bx + 2 by, so the blocks of one anti-diagonal bx + 2 by = t are independent of each other: 10 steps of one or two blocks (lanes
// 0..15 and 16..31) instead of 16 steps of one.  (A block whose top-right neighbour is NOT available was given a *_TOP mode by the
// parser, parse_mb_syn_cavlc.cpp:519-611, and does not look at it - that it may already be there in this order does not matter.)
__device__ LH264_PHASE void intra4x4_phase (LDS WaveLds& L, RecView m, int lane) {
  LDS uint8_t* T = L.T;
  const int half = (lane >> 4) & 1, x = lane & 3, y = (lane >> 2) & 3;
  for (int t = 0; t < 10; t++) {
    // blocks on this anti-diagonal: by from max (0, ceil ((t - 3) / 2)) to min (3, t / 2); at most two
    const int by0 = t <= 3 ? 0 : (t - 2) >> 1;
    const int by1 = min (3, t >> 1);
    const int qy = by0 + half, qx = t - 2 * qy;
    const bool act = lane < 32 && qy <= by1 && qx >= 0 && qx <= 3;
    if (act) {
      const int mode = m.intra_mode (qy * 4 + qx);
      int v;
      if (mode == LH264_I4_V) v = T[tY (4 * qy - 1, 4 * qx + x)];
      else if (mode == LH264_I4_H) v = T[tY (4 * qy + y, 4 * qx - 1)];
      else if (mode == LH264_I4_DC || mode == LH264_I4_DC_L || mode == LH264_I4_DC_T) {
        int s = 0;
        if (mode != LH264_I4_DC_L) {
          const uint32_t tw = * (const LDS uint32_t*)&T[tY (4 * qy - 1, 4 * qx)];
          s += (tw & 0xff) + ((tw >> 8) & 0xff) + ((tw >> 16) & 0xff) + (tw >> 24);
        }
        if (mode != LH264_I4_DC_T)
          s += (int)T[tY (4 * qy, 4 * qx - 1)] + (int)T[tY (4 * qy + 1, 4 * qx - 1)] + (int)T[tY (4 * qy + 2, 4 * qx - 1)] + (int)T[tY (4 * qy + 3, 4 * qx - 1)];
        v = (mode == LH264_I4_DC) ? (s + 4) >> 3 : (s + 2) >> 2;
      } else if (mode == LH264_I4_DC_128) v = 128;
      else v = pred4_dir (T, qx, qy, mode, x, y);
      const int rr = L.R[zidx (qx, qy) * 16 + y * 4 + x];
      T[tY (4 * qy + y, 4 * qx + x)] = (uint8_t)clip_u8 (v + rr);
    }
    wsync();
  }
}

// RecI8x8Luma rec_mb.cpp:70-115 (reference-sample low-pass 8.3.2.2.1; get_intra_predictor.cpp:382-880)
__device__ LH264_PHASE void intra8x8_phase (LDS WaveLds& L, RecView m, int lane) {
  LDS uint8_t* T = L.T;
  const int av = uni (m.intra_avail());
  for (int i8 = 0; i8 < 4; i8++) {
    const int ox = (i8 & 1) * 8, oy = (i8 >> 1) * 8;
    const int mode = uni (m.intra_mode (((i8 >> 1) << 3) + ((i8 & 1) << 1)));
    int tl = i8 == 0 ? !! (av & LH264_AVAIL_TL) : i8 == 1 ? !! (av & LH264_AVAIL_T) : i8 == 2 ? !! (av & LH264_AVAIL_L) : 1;
    const int tr = i8 == 0 ? !! (av & LH264_AVAIL_T) : i8 == 1 ? !! (av & LH264_AVAIL_TR) : i8 == 2 ? 1 : 0;
    const bool need_top = !(mode == LH264_I4_H || mode == LH264_I4_DC_L || mode == LH264_I4_DC_128 || mode == LH264_I4_HU);
    const bool need_left = !(mode == LH264_I4_V || mode == LH264_I4_DC_T || mode == LH264_I4_DC_128 || mode == LH264_I4_DDL ||
                             mode == LH264_I4_DDL_TOP || mode == LH264_I4_VL || mode == LH264_I4_VL_TOP);
    const bool full_tr = (mode == LH264_I4_DDL || mode == LH264_I4_VL);
    const bool corner = (mode == LH264_I4_DDR || mode == LH264_I4_VR || mode == LH264_I4_HD);
    const bool top_var = (mode == LH264_I4_DDL_TOP || mode == LH264_I4_VL_TOP);
    if (corner) tl = 1;
    // filtered edge E[0..24] = L'7..L'0, TL', T'0..T'15
    if (lane < 25) {
      int val = 0;
      const int TLv = tl ? T[tY (oy - 1, ox - 1)] : 0;
      if (lane < 8) {                 // L'[7-lane]
        const int i = 7 - lane;
        if (need_left) {
          const int l0 = T[tY (oy + i, ox - 1)];
          const int lm = i > 0 ? T[tY (oy + i - 1, ox - 1)] : TLv;
          const int lp = i < 7 ? T[tY (oy + i + 1, ox - 1)] : l0;
          if (i == 0) val = tl ? (TLv + 2 * l0 + lp + 2) >> 2 : (3 * l0 + lp + 2) >> 2;
          else val = (lm + 2 * l0 + lp + 2) >> 2;      // i == 7: lp == l0 -> (L6 + 3 L7 + 2) >> 2
        }
      } else if (lane == 8) {
        if (corner) val = ((int)T[tY (oy, ox - 1)] + 2 * TLv + (int)T[tY (oy - 1, ox)] + 2) >> 2;
      } else if (need_top) {
        const int i = lane - 9;       // T'[i]
        const bool t8real = tr && !top_var;
        // raw top samples with the reference's substitution rules
        auto rawT = [&] (int q) -> int {
          if (q < 8) return T[tY (oy - 1, ox + q)];
          if (full_tr) return T[tY (oy - 1, ox + q)];
          if (t8real && q == 8) return T[tY (oy - 1, ox + 8)];
          return T[tY (oy - 1, ox + 7)];
        };
        if (full_tr) {
          if (i == 0) val = tl ? (TLv + 2 * rawT (0) + rawT (1) + 2) >> 2 : (3 * rawT (0) + rawT (1) + 2) >> 2;
          else if (i == 15) val = (rawT (14) + 3 * rawT (15) + 2) >> 2;
          else val = (rawT (i - 1) + 2 * rawT (i) + rawT (i + 1) + 2) >> 2;
        } else {
          if (i == 0) val = tl ? (TLv + 2 * rawT (0) + rawT (1) + 2) >> 2 : (3 * rawT (0) + rawT (1) + 2) >> 2;
          else if (i < 7) val = (rawT (i - 1) + 2 * rawT (i) + rawT (i + 1) + 2) >> 2;
          else if (i == 7) val = t8real ? (rawT (6) + 2 * rawT (7) + rawT (8) + 2) >> 2 : (rawT (6) + 3 * rawT (7) + 2) >> 2;
          else val = rawT (7);
        }
      }
      L.E[lane] = (uint8_t)val;
    }
    wsync();
    {
      const int x = lane & 7, y = lane >> 3;
      const LDS uint8_t* E = L.E;
      int v;
      auto f3 = [&] (int kk) -> int { const int lo = clip3 (kk - 1, 0, 24), mid = clip3 (kk, 0, 24), hi = clip3 (kk + 1, 0, 24); return (E[lo] + 2 * E[mid] + E[hi] + 2) >> 2; };
      auto f2 = [&] (int kk) -> int { return (E[clip3 (kk, 0, 24)] + E[clip3 (kk + 1, 0, 24)] + 1) >> 1; };
      switch (mode) {
      case LH264_I4_V: v = E[9 + x]; break;
      case LH264_I4_H: v = E[7 - y]; break;
      case LH264_I4_DC: case LH264_I4_DC_L: case LH264_I4_DC_T: {
        int term = 0;
        if (lane < 8) term = (mode == LH264_I4_DC_T) ? 0 : E[lane];
        else if (lane < 16) term = (mode == LH264_I4_DC_L) ? 0 : E[9 + lane - 8];
        term = sum_xor (term, 16);
        const int s = __builtin_amdgcn_readlane (term, 0);
        v = (mode == LH264_I4_DC) ? (s + 8) >> 4 : (s + 4) >> 3;
        break;
      }
      case LH264_I4_DC_128: v = 128; break;
      case LH264_I4_DDL: case LH264_I4_DDL_TOP: v = f3 (9 + x + y + 1); break;
      case LH264_I4_DDR: v = f3 (8 + x - y); break;
      case LH264_I4_VR: {
        const int z = 2 * x - y;
        if (z >= 0) { const int kk = 8 + x - (y >> 1); v = (z & 1) ? f3 (kk) : f2 (kk); }
        else if (z == -1) v = f3 (8);
        else v = f3 (9 - y + 2 * x);
        break;
      }
      case LH264_I4_HD: {
        const int z = 2 * y - x;
        if (z >= 0) { const int kk = 8 - (y - (x >> 1)); v = (z & 1) ? f3 (kk) : f2 (kk - 1); }
        else if (z == -1) v = f3 (8);
        else v = f3 (7 + x - 2 * y);
        break;
      }
      case LH264_I4_VL: case LH264_I4_VL_TOP: { const int kk = 9 + x + (y >> 1); v = (y & 1) ? f3 (kk + 1) : f2 (kk); break; }
      default: { const int z = x + 2 * y, a = z >> 1; v = (z & 1) ? f3 (6 - a) : f2 (6 - a); break; }   // HU
      }
      const int X = ox + x, Y = oy + y;
      const int rr = L.R[zidx (X >> 2, Y >> 2) * 16 + (Y & 3) * 4 + (X & 3)];
      wsync();
      T[tY (Y, X)] = (uint8_t)clip_u8 (v + rr);
    }
    wsync();
  }
}

// RecI16x16Mb rec_mb.cpp:179-230 (luma part), lane = 4*b + r strip layout
__device__ LH264_PHASE void intra16_phase (LDS WaveLds& L, int mode, int lane) {
  LDS uint8_t* T = L.T;
  const int b = lane >> 2, r = lane & 3;
  const int bx = (b & 1) | ((b >> 2) & 1) << 1, by = ((b >> 1) & 1) | ((b >> 3) & 1) << 1;
  const int ly = 4 * by + r, lx0 = 4 * bx;
  int pr[4], res[4];
  load_res4 (L.R, lane, res);
  if (mode == LH264_I16_V) {
    const uint32_t t = * (const LDS uint32_t*)&T[tY (-1, lx0)];
#pragma unroll
    for (int i = 0; i < 4; i++) pr[i] = (t >> (8 * i)) & 0xff;
  } else if (mode == LH264_I16_H) {
    const int l = T[tY (ly, -1)];
#pragma unroll
    for (int i = 0; i < 4; i++) pr[i] = l;
  } else if (mode == LH264_I16_P) {
    int term = 0;
    if (lane < 8) term = (lane + 1) * ((int)T[tY (-1, 8 + lane)] - (int)T[tY (-1, 6 - lane)]);
    else if (lane < 16) { const int i = lane - 8; term = (i + 1) * ((int)T[tY (8 + i, -1)] - (int)T[tY (6 - i, -1)]); }
    term = sum_xor (term, 8);
    const int H = __builtin_amdgcn_readlane (term, 0), V = __builtin_amdgcn_readlane (term, 8);
    const int a = ((int)T[tY (15, -1)] + (int)T[tY (-1, 15)]) << 4;
    const int bb = (5 * H + 32) >> 6, c = (5 * V + 32) >> 6;
#pragma unroll
    for (int i = 0; i < 4; i++) pr[i] = clip_u8 ((a + bb * (lx0 + i - 7) + c * (ly - 7) + 16) >> 5);
  } else {
    int v = 128;
    if (mode != LH264_I16_DC_128) {
      int term = 0;
      if (lane < 16) term = (mode == LH264_I16_DC_L) ? 0 : T[tY (-1, lane)];
      else if (lane < 32) term = (mode == LH264_I16_DC_T) ? 0 : T[tY (lane - 16, -1)];
      term = sum_xor (term, 32);
      const int s = __builtin_amdgcn_readlane (term, 0);
      v = (mode == LH264_I16_DC) ? (s + 16) >> 5 : (s + 8) >> 4;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) pr[i] = v;
  }
  const uint32_t o = pack_add4 (pr, res);
  wsync();
  * (LDS uint32_t*)&T[tY (ly, lx0)] = o;
}

// RecI4x4Chroma rec_mb.cpp:160-177 (prediction) + RecChroma :547-575 (residual): lanes 0..31
__device__ LH264_PHASE void intra_chroma_phase (LDS WaveLds& L, int mode, int lane) {
  const int r = lane & 3;
  const int cp = (lane >> 4) & 1, cj = (lane >> 2) & 3;
  const int cy = 4 * (cj >> 1) + r, cx0 = 4 * (cj & 1);
  const LDS uint8_t* Cc = L.C[cp];
  int pr[4] = {128, 128, 128, 128}, cres[4];
  load_res4 (L.R + 256, lane & 31, cres);
  const int q = lane & 15;
  if (mode == LH264_C_V) {
    const uint32_t t = * (const LDS uint32_t*)&Cc[tC (-1, cx0)];
#pragma unroll
    for (int i = 0; i < 4; i++) pr[i] = (t >> (8 * i)) & 0xff;
  } else if (mode == LH264_C_H) {
    const int l = Cc[tC (cy, -1)];
#pragma unroll
    for (int i = 0; i < 4; i++) pr[i] = l;
  } else if (mode == LH264_C_P) {
    int term = 0;
    if (q < 4) term = (q + 1) * ((int)Cc[tC (-1, 4 + q)] - (int)Cc[tC (-1, 2 - q)]);
    else if (q < 8) { const int i = q - 4; term = (i + 1) * ((int)Cc[tC (4 + i, -1)] - (int)Cc[tC (2 - i, -1)]); }
    term = sum_xor (term, 4);
    const int H = __shfl (term, (lane & 48) + 0), V = __shfl (term, (lane & 48) + 4);
    const int a = ((int)Cc[tC (7, -1)] + (int)Cc[tC (-1, 7)]) << 4;
    const int bb = (17 * H + 16) >> 5, c = (17 * V + 16) >> 5;
#pragma unroll
    for (int i = 0; i < 4; i++) pr[i] = clip_u8 ((a + bb * (cx0 + i - 3) + c * (cy - 3) + 16) >> 5);
  } else if (mode != LH264_C_DC_128) {
    int term = (q < 8) ? (int)Cc[tC (-1, q)] : (int)Cc[tC (q - 8, -1)];
    term = sum_xor (term, 4);
    const int base = lane & 48;
    const int t0 = __shfl (term, base + 0), t1 = __shfl (term, base + 4), l0 = __shfl (term, base + 8), l1 = __shfl (term, base + 12);
    int v;
    const bool rightq = cx0 >= 4, lowq = cy >= 4;
    if (mode == LH264_C_DC) v = !rightq && !lowq ? (t0 + l0 + 4) >> 3 : rightq && !lowq ? (t1 + 2) >> 2 : !rightq ? (l1 + 2) >> 2 : (t1 + l1 + 4) >> 3;
    else if (mode == LH264_C_DC_L) v = lowq ? (l1 + 2) >> 2 : (l0 + 2) >> 2;
    else v = rightq ? (t1 + 2) >> 2 : (t0 + 2) >> 2;
#pragma unroll
    for (int i = 0; i < 4; i++) pr[i] = v;
  }
  const uint32_t o = pack_add4 (pr, cres);
  wsync();
  if (lane < 32) * (LDS uint32_t*)&L.C[cp][tC (cy, cx0)] = o;
}

// ------------------------------------------------------------------------------------------------
// motion compensation straight from the padded reference planes (mc.cpp:142-380)
// ------------------------------------------------------------------------------------------------
// Reference fetches are issued as whole 16-byte (luma) / 8-byte (chroma) aligned rows up front, for every row the
// macroblock can need, and the interpolation works on registers afterwards: one exposed memory latency per macroblock
// instead of one per filter stage.  Row12 = 12 consecutive bytes starting at an unaligned address.
typedef uint32_t v4u __attribute__ ((ext_vector_type (4)));
typedef uint32_t v2u __attribute__ ((ext_vector_type (2)));
struct Row12 { uint32_t w0, w1, w2; };
__device__ __forceinline__ Row12 align12 (v4u d, int sh) {
  Row12 o;
  o.w0 = __builtin_amdgcn_alignbyte (d.y, d.x, sh);
  o.w1 = __builtin_amdgcn_alignbyte (d.z, d.y, sh);
  o.w2 = __builtin_amdgcn_alignbyte (d.w, d.z, sh);
  return o;
}
#define BYTE(w, i) (int) (((w) >> (8 * (i))) & 0xff)
// byte i (0..11, compile-time) of a Row12
#define RB(R, i) ((i) < 4 ? BYTE ((R).w0, (i)) : (i) < 8 ? BYTE ((R).w1, (i) - 4) : BYTE ((R).w2, (i) - 8))

// The 6-tap filter (1,-5,20,20,-5,1) on bytes without unpacking them one by one.
// Horizontal: the six samples of an output are consecutive bytes - two signed 4x8-bit dot products (v_dot4c_i32_i8) on the row
// with 128 subtracted from every sample (the taps sum to 32, so 4096 goes back in).  Vertical: the same byte position of six
// rows - the rows are split into two 2x16-bit words each (even / odd bytes) and the filter runs on 16-bit pairs (v_pk_*); the
// sums stay within int16 (|sum| <= 10,710).
typedef short s16x2 __attribute__ ((ext_vector_type (2)));
__device__ __forceinline__ s16x2 as_s2 (uint32_t v) { return __builtin_bit_cast (s16x2, v); }
__device__ __forceinline__ uint32_t as_u (s16x2 v) { return __builtin_bit_cast (uint32_t, v); }
// horizontal taps at outputs 0..3 of a row (output i = samples i..i+5)
__device__ __forceinline__ void htap4 (const Row12 R, int out[4]) {
  const uint32_t s0 = R.w0 ^ 0x80808080u, s1 = R.w1 ^ 0x80808080u, s2 = R.w2 ^ 0x80808080u;
  const int ca = 0x1414FB01, cb = 0x000001FB;       // bytes (1, -5, 20, 20) and (-5, 1, 0, 0)
  out[0] = __builtin_amdgcn_sdot4 ((int)s0, ca, __builtin_amdgcn_sdot4 ((int)s1, cb, 4096, false), false);
#pragma unroll
  for (int i = 1; i < 4; i++)
    out[i] = __builtin_amdgcn_sdot4 ((int)__builtin_amdgcn_alignbyte (s1, s0, i), ca,
                                     __builtin_amdgcn_sdot4 ((int)__builtin_amdgcn_alignbyte (s2, s1, i), cb, 4096, false), false);
}
// vertical taps of the byte positions of one word of six rows: even bytes (0, 2) in e, odd bytes (1, 3) in o, as 16-bit pairs
__device__ __forceinline__ void vtap_word (uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3, uint32_t r4, uint32_t r5, s16x2& e, s16x2& o) {
  const uint32_t M = 0x00ff00ffu;
  const s16x2 k20 = {20, 20}, k5 = {5, 5};
  e = (as_s2 (r2 & M) + as_s2 (r3 & M)) * k20 + (as_s2 (r0 & M) + as_s2 (r5 & M)) - (as_s2 (r1 & M) + as_s2 (r4 & M)) * k5;
  o = (as_s2 ((r2 >> 8) & M) + as_s2 ((r3 >> 8) & M)) * k20 + (as_s2 ((r0 >> 8) & M) + as_s2 ((r5 >> 8) & M)) - (as_s2 ((r1 >> 8) & M) + as_s2 ((r4 >> 8) & M)) * k5;
}

// one 4x1 luma strip from the fetched rows: R[k] holds samples -2..9 of row k-2 relative to the strip's first sample
// (McLuma_c / McHorVer* common/src/mc.cpp:142-380).  With fy == 0 only R[2] is defined.
__device__ __forceinline__ void mc_luma_rows (const Row12 R[6], int fx, int fy, int out[4]) {
  if ((fx | fy) == 0) {
#pragma unroll
    for (int i = 0; i < 4; i++) out[i] = RB (R[2], i + 2);
    return;
  }
  if (fy == 0) {                      // a, b, c : one row
    int t[4];
    htap4 (R[2], t);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int b = clip_u8 ((t[i] + 16) >> 5);
      out[i] = fx == 2 ? b : (b + (fx == 1 ? RB (R[2], i + 2) : RB (R[2], i + 3)) + 1) >> 1;
    }
    return;
  }
  if (fx == 0) {                      // d, h, n : six rows, samples 0..3 = bytes 2..5
    s16x2 e, o;
    vtap_word (__builtin_amdgcn_alignbyte (R[0].w1, R[0].w0, 2), __builtin_amdgcn_alignbyte (R[1].w1, R[1].w0, 2), __builtin_amdgcn_alignbyte (R[2].w1, R[2].w0, 2),
               __builtin_amdgcn_alignbyte (R[3].w1, R[3].w0, 2), __builtin_amdgcn_alignbyte (R[4].w1, R[4].w0, 2), __builtin_amdgcn_alignbyte (R[5].w1, R[5].w0, 2), e, o);
    const int t[4] = {e.x, o.x, e.y, o.y};
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int h = clip_u8 ((t[i] + 16) >> 5);
      out[i] = fy == 2 ? h : (h + (fy == 1 ? RB (R[2], i + 2) : RB (R[3], i + 2)) + 1) >> 1;
    }
    return;
  }
  // both fractions: vertical 6-tap sums of the 9 columns -2..6 (int16, as the reference keeps them)
  int vs[9];
  {
    s16x2 e0, o0, e1, o1, e2, o2;
    vtap_word (R[0].w0, R[1].w0, R[2].w0, R[3].w0, R[4].w0, R[5].w0, e0, o0);
    vtap_word (R[0].w1, R[1].w1, R[2].w1, R[3].w1, R[4].w1, R[5].w1, e1, o1);
    vtap_word (R[0].w2, R[1].w2, R[2].w2, R[3].w2, R[4].w2, R[5].w2, e2, o2);
    vs[0] = e0.x; vs[1] = o0.x; vs[2] = e0.y; vs[3] = o0.y; vs[4] = e1.x; vs[5] = o1.x; vs[6] = e1.y; vs[7] = o1.y; vs[8] = e2.x;
  }
  int hh[4];                          // horizontal half-sample of window row 2 (fy==1,2) or 3 (fy==3)
  Row12 H = R[2];
  if (fy == 3) H = R[3];
  htap4 (H, hh);
#pragma unroll
  for (int i = 0; i < 4; i++) hh[i] = clip_u8 ((hh[i] + 16) >> 5);
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int c = i + 2;
    const int vv0 = clip_u8 ((vs[c] + 16) >> 5), vv1 = clip_u8 ((vs[c + 1] + 16) >> 5);
    int v;
    if (fx == 2 || fy == 2) {
      const int j = clip_u8 ((tap6 (vs[c - 2], vs[c - 1], vs[c], vs[c + 1], vs[c + 2], vs[c + 3]) + 512) >> 10);
      if (fx == 2 && fy == 2) v = j;
      else if (fx == 2) v = (j + hh[i] + 1) >> 1;          // f, q
      else v = (j + (fx == 1 ? vv0 : vv1) + 1) >> 1;       // i, k
    } else v = (hh[i] + (fx == 1 ? vv0 : vv1) + 1) >> 1;   // e, g, p, r
    out[i] = v;
  }
}

// two horizontally adjacent chroma samples sharing one motion vector (a 4x4 luma block = 2x2 chroma); a / b = the
// 4 bytes starting at the pair's first sample in rows 0 / 1 (McChroma_c mc.cpp)
__device__ __forceinline__ void mc_chroma_rows (uint32_t a, uint32_t b, int dx, int dy, int& o0, int& o1) {
  if ((dx | dy) == 0) { o0 = BYTE (a, 0); o1 = BYTE (a, 1); return; }
  // the four weights (each <= 64) as bytes, the four samples of an output as bytes: one unsigned 4x8-bit dot product each
  const uint32_t wgt = (uint32_t) ((8 - dx) * (8 - dy)) | (uint32_t) (dx * (8 - dy)) << 8 | (uint32_t) ((8 - dx) * dy) << 16 | (uint32_t) (dx * dy) << 24;
  o0 = (int) (__builtin_amdgcn_udot4 ((a & 0xffffu) | (b << 16), wgt, 32u, false) >> 6);
  o1 = (int) (__builtin_amdgcn_udot4 (((a >> 8) & 0xffffu) | ((b >> 8) << 16), wgt, 32u, false) >> 6);
}

// partition geometry of the 4x4 block (bx,by): origin and size of the motion partition that holds it
__device__ __forceinline__ void partition_of (int mb_type, RecView m, int bx, int by, int& ox, int& oy, int& pw, int& ph) {
  ox = oy = 0; pw = ph = 16;
  if (mb_type == LH264_MB_P16x8) { ph = 8; oy = (by >> 1) << 3; }
  else if (mb_type == LH264_MB_P8x16) { pw = 8; ox = (bx >> 1) << 3; }
  else if (mb_type == LH264_MB_P8x8 || mb_type == LH264_MB_P8x8REF0) {
    const int q = ((by >> 1) << 1) | (bx >> 1);
    const int st = m.sub_type (q);
    ox = (bx >> 1) << 3; oy = (by >> 1) << 3; pw = ph = 8;
    if (st == LH264_SUB_8x4) { ph = 4; oy += (by & 1) << 2; }
    else if (st == LH264_SUB_4x8) { pw = 4; ox += (bx & 1) << 2; }
    else if (st == LH264_SUB_4x4) { pw = ph = 4; ox += (bx & 1) << 2; oy += (by & 1) << 2; }
  }
}

// what the inter phase needs of the frame context, passed BY VALUE: a reference to the caller's FrameCtx would turn
// every field access in this non-inlined function into a scratch load followed by "s_waitcnt vmcnt(0)", which also
// drains the reference fetches in flight.
struct InterCtx { uint64_t prev_dy; int sy, sc, mb_w, mb_h, prev_base, prev_h, nw; };

// GetInterPred rec_mb.cpp:344-545, BaseMC :247-274, WeightPrediction :276-341 (+ residual add)
__device__ LH264_PHASE void inter_phase (const InterCtx F, LDS WaveLds& L, const LDS WgLds& G, RecView m, SliceView sl, int mb_type,
                                          int mbx, int mby, bool has_res, int lane) {
  const int b = lane >> 2, r = lane & 3;
  const int bx = (b & 1) | ((b >> 2) & 1) << 1, by = ((b >> 1) & 1) | ((b >> 3) & 1) << 1;
  const int ly = 4 * by + r, lx0 = 4 * bx;
  const int cp = (lane >> 4) & 1, cj = (lane >> 2) & 3;
  const int cy = 4 * (cj >> 1) + r, cx0 = 4 * (cj & 1);
  int need = -0x7fffffff;
  if (lane < 16) {
    const int qx = lane & 3, qy = lane >> 2;      // raster 4x4 block
    int ox, oy, pw, ph;
    partition_of (mb_type, m, qx, qy, ox, oy, pw, ph);
    const int pb = (oy >> 2) * 4 + (ox >> 2);     // raster index of the partition's first block (its MV)
    const int W = F.mb_w * 16, Hh = F.mb_h * 16;
    int fx = ((mbx * 16 + ox) << 2) + m.mvx (pb);
    int fy = ((mby * 16 + oy) << 2) + m.mvy (pb);
    fx = clip3 (fx, (-LH264_PAD_LUMA + 2) * 4, (W + LH264_PAD_LUMA - 19) * 4);
    fy = clip3 (fy, (-LH264_PAD_LUMA + 2) * 4, (Hh + LH264_PAD_LUMA - 19) * 4);
    const int ridx = m.ref_idx (((qy >> 1) << 1) + (qx >> 1));
    int slot = (ridx >= 0 && ridx < LH264_MAX_REFS) ? sl.ref_slot (ridx) : -1;
    if (slot < 0) slot = sl.ref_slot (0);          // rec_mb.cpp:238-242: missing picture -> list entry 0
    if (slot < 0) slot = 0;
    // integer sample position of this 4x4 block's first pixel in the reference planes
    const int sx = (fx >> 2) + (qx * 4 - ox), syy = (fy >> 2) + (qy * 4 - oy);
    const int cxs = (fx >> 3) + ((qx * 4 - ox) >> 1), cys = (fy >> 3) + ((qy * 4 - oy) >> 1);
    L.mvi[lane][0] = syy * F.sy + sx;
    L.mvi[lane][1] = cys * F.sc + cxs;
    L.mvi[lane][2] = (fx & 3) | (fy & 3) << 2 | (fx & 7) << 4 | (fy & 7) << 8 | slot << 12 | (ridx < 0 ? 0 : ridx) << 16;
    L.mvi[lane][3] = ox | oy << 8 | pw << 16 | ph << 24;
    // last luma row this block can touch in the frame that may still be in flight: its 4 rows, 3 more under a
    // vertical 6-tap; chroma: 2 rows, 1 more under a vertical fraction (in luma units)
    if (L.refp[slot][0] == F.prev_dy) need = max (syy + 3 + ((fy & 3) ? 3 : 0), 2 * (cys + 1 + ((fy & 7) ? 1 : 0)) + 1);
  }
  if (F.prev_dy) {
#pragma unroll
    for (int msk = 1; msk < 16; msk <<= 1) need = max (need, __shfl_xor (need, msk));
    need = uni (need);
    if (need != -0x7fffffff) {
      const int rneed = clip3 (need >> 4, 0, F.prev_h - 1);
      wait_prefix (G, L, F.prev_base + min (rneed + 1, F.prev_h - 1), F.nw);
    }
  }
  wsync();
  const bool wp = sl.weighted() != 0;
  // ---- issue every reference fetch of the macroblock ------------------------------------------------------------
  const int lrb = by * 4 + bx;                      // raster index of this lane's luma block
  const int linfo = L.mvi[lrb][2];
  const int lfx = linfo & 3, lfy = (linfo >> 2) & 3;
  const bool any_fy = __ballot (lfy != 0) != 0;     // wave-uniform: does any strip need the vertical neighbours?
  v4u lrow[6];
  int lsh;
  {
    const GLB uint8_t* plane = (const GLB uint8_t*)L.refp[(linfo >> 12) & 15][0];
    const uintptr_t a0 = (uintptr_t) (plane + L.mvi[lrb][0] + r * F.sy - 2);
    lsh = (int) (a0 & 3);
    const GLB uint8_t* q = (const GLB uint8_t*) (a0 & ~ (uintptr_t)3);
#ifdef LH264_ABL_NOMC      // timing ablation only (wrong pictures): no reference fetch
#pragma unroll
    for (int k = 0; k < 6; k++) lrow[k] = (v4u) ((uint32_t) (uintptr_t)q);
#else
    lrow[2] = * (const GLB v4u*)q;
    if (any_fy) {
      lrow[0] = * (const GLB v4u*) (q - 2 * F.sy); lrow[1] = * (const GLB v4u*) (q - F.sy);
      lrow[3] = * (const GLB v4u*) (q + F.sy); lrow[4] = * (const GLB v4u*) (q + 2 * F.sy); lrow[5] = * (const GLB v4u*) (q + 3 * F.sy);
    } else lrow[0] = lrow[1] = lrow[3] = lrow[4] = lrow[5] = (v4u) (0u);     // never read (every strip has fy == 0)
#endif
  }
  v2u crow[2][2];                                   // [pair h][row]
  int cinfo[2], csh[2], crb[2];
  if (lane < 32) {
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int cx = cx0 + 2 * h;
      crb[h] = (cy >> 1) * 4 + (cx >> 1);
      cinfo[h] = L.mvi[crb[h]][2];
      const GLB uint8_t* plane = (const GLB uint8_t*)L.refp[(cinfo[h] >> 12) & 15][1 + cp];
      const uintptr_t a0 = (uintptr_t) (plane + L.mvi[crb[h]][1] + (cy & 1) * F.sc);
      csh[h] = (int) (a0 & 3);
      const GLB uint8_t* q = (const GLB uint8_t*) (a0 & ~ (uintptr_t)3);
#ifdef LH264_ABL_NOMC
      crow[h][0] = crow[h][1] = (v2u) ((uint32_t) (uintptr_t)q);
#else
      crow[h][0] = * (const GLB v2u*)q;
      crow[h][1] = * (const GLB v2u*) (q + F.sc);
#endif
    }
  }
  // ---- luma ------------------------------------------------------------------------------------------------------
  int pr[4], res[4];
  {
    Row12 R[6];
#pragma unroll
    for (int k = 0; k < 6; k++) R[k] = align12 (lrow[k], lsh);
    mc_luma_rows (R, lfx, lfy, pr);
    if (wp) {
      const int ri = (linfo >> 16) & 15, ld = sl.luma_denom(), wt = sl.luma_weight (ri), of = sl.luma_offset (ri);
#pragma unroll
      for (int i = 0; i < 4; i++) pr[i] = clip_u8 (ld >= 1 ? ((pr[i] * wt + (1 << (ld - 1))) >> ld) + of : pr[i] * wt + of);
    }
  }
  res[0] = res[1] = res[2] = res[3] = 0;
  if (has_res) load_res4 (L.R, lane, res);
  * (LDS uint32_t*)&L.T[tY (ly, lx0)] = pack_add4 (pr, res);
  // ---- chroma strip: plane cp, row cy, cols cx0..cx0+3 = two luma 4x4 blocks wide ----------------------------------
  if (lane < 32) {
    int cpr[4], cres[4];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int cx = cx0 + 2 * h;
      const int info = cinfo[h];
      const uint32_t a = __builtin_amdgcn_alignbyte (crow[h][0].y, crow[h][0].x, csh[h]);
      const uint32_t bb = __builtin_amdgcn_alignbyte (crow[h][1].y, crow[h][1].x, csh[h]);
      mc_chroma_rows (a, bb, (info >> 4) & 7, (info >> 8) & 7, cpr[2 * h], cpr[2 * h + 1]);
      if (wp) {
        // reference quirk (rec_mb.cpp:309-311): only the top-left (w>>2)x(h>>2) samples of the chroma block are weighted
        const int geo = L.mvi[crb[h]][3];
        const int pox = (geo & 0xff) >> 1, poy = ((geo >> 8) & 0xff) >> 1, pw = (geo >> 16) & 0xff, ph = (geo >> 24) & 0xff;
        const int ri = (info >> 16) & 15, ld = sl.chroma_denom(), wt = sl.chroma_weight (ri, cp), of = sl.chroma_offset (ri, cp);
#pragma unroll
        for (int i = 0; i < 2; i++) {
          if (cx + i - pox < (pw >> 2) && cy - poy < (ph >> 2)) {
            const int v = cpr[2 * h + i];
            cpr[2 * h + i] = clip_u8 (ld >= 1 ? ((v * wt + (1 << (ld - 1))) >> ld) + of : v * wt + of);
          }
        }
      }
    }
    cres[0] = cres[1] = cres[2] = cres[3] = 0;
    if (has_res) load_res4 (L.R + 256, lane, cres);
    * (LDS uint32_t*)&L.C[cp][tC (cy, cx0)] = pack_add4 (cpr, cres);
  }
}

// ------------------------------------------------------------------------------------------------
// deblocking (deblocking.cpp / deblocking_common.cpp)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int bs_mv (RecView a, int ia, RecView b, int ib) {
  const int dx = abs (a.mvx (ia) - b.mvx (ib)), dy = abs (a.mvy (ia) - b.mvy (ib));
  return (a.ref4 (ia) != b.ref4 (ib)) || dx >= 4 || dy >= 4;   // MB_BS_MV deblocking.cpp:58-63: reference INDICES
}

// boundary strength of segment `seg` of edge `e` in direction dir (0: vertical edges, 1: horizontal);
// DeblockingBsMarginalMBAvcbase deblocking.cpp:273-352, DeblockingBSInsideMB* :160-271, WelsDeblockingMb :815-862
__device__ __forceinline__ int compute_bs (RecView m, RecView nb, bool have_nb, int dir, int e, int seg, bool intra) {
  const int t8 = m.flags() & LH264_MBF_T8x8;
  const int mtype = m.mb_type();
  if (e == 0) {
    if (!have_nb) return 0;
    if (intra || (nb.mb_type() & LH264_MB_INTRA)) return 4;
    const int t8n = nb.flags() & LH264_MBF_T8x8;
    const int bc = dir == 0 ? seg * 4 : seg;
    const int bn = dir == 0 ? seg * 4 + 3 : 12 + seg;
    int nzc_c = m.nzc (bc), nzc_n = nb.nzc (bn);
    if (t8) nzc_c = m.nz8 (((bc >> 3) << 3) + ((bc & 3) >> 1) * 2);
    if (t8n) nzc_n = nb.nz8 (((bn >> 3) << 3) + ((bn & 3) >> 1) * 2);
    if (nzc_c | nzc_n) return 2;
    int mc_ = bc, mn_ = bn;
    if (t8) mc_ = dir == 0 ? (seg >> 1) * 8 : (seg >> 1) * 2;
    if (t8n) mn_ = dir == 0 ? (seg >> 1) * 8 + 2 : 8 + (seg >> 1) * 2;
    return bs_mv (m, mc_, nb, mn_);
  }
  if (intra) return 3;
  if (mtype == LH264_MB_SKIP) return 0;
  if (t8 && e != 2) return 0;
  const bool mv_too = mtype != LH264_MB_P16x16;
  if (t8) {
    const int r = seg >> 1;
    int a, b;
    if (dir == 0) { a = r * 8; b = r * 8 + 2; } else { a = r * 2; b = 8 + r * 2; }
    if (m.nz8 (a) | m.nz8 (b)) return 2;
    return mv_too ? bs_mv (m, b, m, a) : 0;
  }
  const int a = dir == 0 ? seg * 4 + e - 1 : (e - 1) * 4 + seg;
  const int b = dir == 0 ? seg * 4 + e : e * 4 + seg;
  if (m.nzc (a) | m.nzc (b)) return 2;
  return mv_too ? bs_mv (m, b, m, a) : 0;
}

// Filter one line across one edge, samples in registers.  Luma: DeblockLumaLt4_c / DeblockLumaEq4_c
// (deblocking_common.cpp:5-83); chroma: DeblockChromaLt4_c / DeblockChromaEq4_c (:85-140), which only touch p0/q0 and
// use tc0+1.  One routine for both so that the luma lanes (0..15) and the chroma lanes (16..31) of a wave run the same
// instruction stream.  Every formula uses the samples as they were before this edge.
// Written without divergent branches: the lanes of a wave hold different lines with different strengths, so a branch per condition
// is a chain of exec-mask saves and restores around a few instructions each; here the normal filter is computed for every lane and
// selected, the strong one (bS 4: macroblock edges of intra macroblocks only) behind one wave-uniform branch.
__device__ __forceinline__ int absd (int a, int b) { return (int)__builtin_amdgcn_sad_u16 ((uint32_t)a, (uint32_t)b, 0u); }    // samples: 0..255
__device__ __forceinline__ void filter_edge (int& rp3, int& rp2, int& rp1, int& rp0, int& rq0, int& rq1, int& rq2, int& rq3,
                                             int bs, int alpha, int beta, int tc0, bool chroma) {
  const int p3 = rp3, p2 = rp2, p1 = rp1, p0 = rp0, q0 = rq0, q1 = rq1, q2 = rq2, q3 = rq3;
  const int d = absd (p0, q0);
  const bool on = bs != 0 && d < alpha && absd (p1, p0) < beta && absd (q1, q0) < beta;
  const bool ap = !chroma && absd (p2, p0) < beta, aq = !chroma && absd (q2, q0) < beta;
  // bS < 4
  const int t = chroma ? tc0 + 1 : tc0 + (ap ? 1 : 0) + (aq ? 1 : 0);
  const int dl = clip3 ((((q0 - p0) << 2) + (p1 - q1) + 4) >> 3, -t, t);
  const int avg = (p0 + q0 + 1) >> 1;
  int np0 = clip_u8 (p0 + dl), nq0 = clip_u8 (q0 - dl);
  int np1 = ap ? p1 + clip3 ((p2 + avg - (p1 << 1)) >> 1, -tc0, tc0) : p1;
  int nq1 = aq ? q1 + clip3 ((q2 + avg - (q1 << 1)) >> 1, -tc0, tc0) : q1;
  int np2 = p2, nq2 = q2;
  if (__ballot (on && bs == 4)) {                   // wave-uniform
    const bool small = !chroma && d < ((alpha >> 2) + 2);
    const bool sp = small && ap, sq = small && aq;
    const int s4 = bs == 4;
    const int wp0 = (2 * p1 + p0 + q1 + 2) >> 2, wq0 = (2 * q1 + q0 + p1 + 2) >> 2;
    const int xp0 = sp ? (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3 : wp0;
    const int xq0 = sq ? (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3 : wq0;
    const int xp1 = sp ? (p2 + p1 + p0 + q0 + 2) >> 2 : p1, xq1 = sq ? (p0 + q0 + q1 + q2 + 2) >> 2 : q1;
    const int xp2 = sp ? (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3 : p2, xq2 = sq ? (2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3 : q2;
    np0 = s4 ? xp0 : np0; nq0 = s4 ? xq0 : nq0; np1 = s4 ? xp1 : np1; nq1 = s4 ? xq1 : nq1; np2 = s4 ? xp2 : np2; nq2 = s4 ? xq2 : nq2;
  }
  rp0 = on ? np0 : p0; rq0 = on ? nq0 : q0; rp1 = on ? np1 : p1; rq1 = on ? nq1 : q1; rp2 = on ? np2 : p2; rq2 = on ? nq2 : q2;
}

// WelsDeblockingMb (deblocking.cpp:815-862) + FilteringEdgeLumaHV / FilteringEdgeChromaHV (:568-700): the macroblock's
// vertical edges then its horizontal edges, inside the tile.  Lane i < 16 owns luma line i (a row for the vertical
// edges, a column for the horizontal ones), lanes 16..31 own the 8 lines of Cb and Cr.  A line is loaded once, all its
// edges are filtered in registers, and it is written back once.
__device__ LH264_PHASE void deblock_phase (LDS WaveLds& L, const LDS WgLds& G, RecView m, RecView lm, RecView tm, SliceView sl,
                                            bool left_av, bool top_av, int lane) {
  const int mtype = uni (m.mb_type());
  const bool mintra = (mtype == LH264_MB_I4x4 || mtype == LH264_MB_I8x8 || mtype == LH264_MB_I16x16 || mtype == LH264_MB_IPCM);
  int my_bs = 0;
  if (lane < 32) {
    const int dir = lane >> 4, e = (lane >> 2) & 3, seg = lane & 3;
    // with the 8x8 transform the luma edges 1 and 3 do not exist (deblocking.cpp:836-849; chroma never uses them)
    const bool t8 = (m.flags() & LH264_MBF_T8x8) != 0;
    my_bs = (t8 && (e & 1)) ? 0 : compute_bs (m, dir == 0 ? lm : tm, dir == 0 ? left_av : top_av, dir, e, seg, mintra);
    L.bs[lane] = (uint8_t)my_bs;
  }
  // which of the 2 x 4 edges have any strength at all: one scalar mask (bit 16 dir + 4 edge + segment), so that an edge nobody
  // filters costs a scalar branch and nothing else
  const uint32_t live = (uint32_t)__ballot (my_bs != 0);
  wsync();
  const v4u bs0 = * (const LDS v4u*)&L.bs[0], bs1 = * (const LDS v4u*)&L.bs[16];     // [edge] = 4 segment bytes
  const bool any0 = (live & 0xffffu) != 0, any1 = (live >> 16) != 0;
  if ((!any0 && !any1) || lane >= 32) return;
#ifdef LH264_ABL_BSONLY      // timing ablation only (wrong pictures): boundary strengths, no filtering
  return;
#endif
  const bool chroma = lane >= 16;
  const int cpl = (lane >> 3) & 1;                  // chroma plane of lanes 16..31
  const int li = chroma ? (lane & 7) : lane;        // line inside the plane
  const int sh8 = 8 * (chroma ? (li >> 1) : (li >> 2));   // this line's segment byte inside a bS dword
  // alpha / beta / tc0 for the three QP cases: inner edges, left macroblock edge, top macroblock edge
  const LDS uint8_t* tab = G.tab;
  const int ao = sl.alpha_off(), bo = sl.beta_off();
  const int qc = chroma ? m.qp_c (cpl) : m.qp_y();
  const int ql = (qc + (chroma ? lm.qp_c (cpl) : lm.qp_y()) + 1) >> 1;
  const int qt = (qc + (chroma ? tm.qp_c (cpl) : tm.qp_y()) + 1) >> 1;
  const int iac = tab_idx (qc + ao), ial = tab_idx (ql + ao), iat = tab_idx (qt + ao);
  const int alc = tab[TAB_ALPHA + iac], all_ = tab[TAB_ALPHA + ial], alt = tab[TAB_ALPHA + iat];
  const int bec = tab[TAB_BETA + tab_idx (qc + bo)], bel = tab[TAB_BETA + tab_idx (ql + bo)], bet = tab[TAB_BETA + tab_idx (qt + bo)];
  const uint32_t tcc = * (const LDS uint32_t*)&tab[TAB_TC0 + 4 * iac], tcl = * (const LDS uint32_t*)&tab[TAB_TC0 + 4 * ial],
                 tct = * (const LDS uint32_t*)&tab[TAB_TC0 + 4 * iat];
  // Reference behaviour kept bit for bit: for an INTRA macroblock whose Cb and Cr QPs differ, FilteringEdgeChromaHV looks tc0 up
  // in its "chroma v" loop only; its "chroma h" loop (deblocking.cpp:786-807) filters the inner edge of each plane with what that
  // loop left behind - Cr's tc0 whenever Cr's alpha | beta is nonzero.  So Cb's inner horizontal edge takes Cr's tc0 then.
  uint32_t tcc_h = tcc;
  if (mintra && uni (m.qp_c (0)) != uni (m.qp_c (1))) {
    const int qr = m.qp_c (1), iar = tab_idx (qr + ao);
    if (chroma && cpl == 0 && (tab[TAB_ALPHA + iar] | tab[TAB_BETA + tab_idx (qr + bo)])) tcc_h = * (const LDS uint32_t*)&tab[TAB_TC0 + 4 * iar];
  }
  int v[20];
  if (any0 && (live & 0xfff0u) == 0) {
    // ---- vertical edges, only the macroblock edge is live: samples -4..3 of the row ------------------------------------------------
    LDS uint32_t* rowp = (LDS uint32_t*) (chroma ? &L.C[cpl][tC (li, -4)] : &L.T[tY (li, -4)]);
    const uint32_t w0 = rowp[0], w1 = rowp[1];
    int p3 = BYTE (w0, 0), p2 = BYTE (w0, 1), p1 = BYTE (w0, 2), p0 = BYTE (w0, 3), q0 = BYTE (w1, 0), q1 = BYTE (w1, 1), q2 = BYTE (w1, 2), q3 = BYTE (w1, 3);
    const int bs = (int) ((bs0.x >> sh8) & 0xff);
    filter_edge (p3, p2, p1, p0, q0, q1, q2, q3, (all_ | bel) ? bs : 0, all_, bel, (int) ((tcl >> (8 * (bs & 3))) & 0xff), chroma);
    rowp[0] = (uint32_t)p3 | (uint32_t)p2 << 8 | (uint32_t)p1 << 16 | (uint32_t)p0 << 24;
    rowp[1] = (uint32_t)q0 | (uint32_t)q1 << 8 | (uint32_t)q2 << 16 | (uint32_t)q3 << 24;
  } else if (any0) {
    // ---- vertical edges: line = row li, samples -4..15 (chroma -4..7) ------------------------------------------
    LDS uint32_t* rowp = (LDS uint32_t*) (chroma ? &L.C[cpl][tC (li, -4)] : &L.T[tY (li, -4)]);
    uint32_t w[5];
#pragma unroll
    for (int k = 0; k < 4; k++) w[k] = rowp[k];
    w[4] = chroma ? 0u : rowp[4];
#pragma unroll
    for (int i = 0; i < 20; i++) v[i] = BYTE (w[i >> 2], i & 3);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      // (scalar; the chroma lines take the strengths of luma edge 2 for their edge 1)
      if ((((k == 1 ? live | live >> 4 : live) >> (4 * k)) & 15u) == 0) continue;
      // chroma edge k (k < 2) lies on luma edge 2k; chroma has no edges 2, 3
      const uint32_t bl = k == 0 ? bs0.x : k == 1 ? bs0.y : k == 2 ? bs0.z : bs0.w;
      const uint32_t bc = k == 0 ? bs0.x : k == 1 ? bs0.z : 0u;
      const int bs = (int) (((chroma ? bc : bl) >> sh8) & 0xff);
      const int alpha = k == 0 ? all_ : alc, beta = k == 0 ? bel : bec;
      const uint32_t tcw = k == 0 ? tcl : tcc;
      filter_edge (v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3], v[4 * k + 4], v[4 * k + 5], v[4 * k + 6], v[4 * k + 7],
                   (alpha | beta) ? bs : 0, alpha, beta, (int) ((tcw >> (8 * (bs & 3))) & 0xff), chroma);
    }
#pragma unroll
    for (int k = 0; k < 5; k++) w[k] = (uint32_t)v[4 * k] | (uint32_t)v[4 * k + 1] << 8 | (uint32_t)v[4 * k + 2] << 16 | (uint32_t)v[4 * k + 3] << 24;
#pragma unroll
    for (int k = 0; k < 3; k++) rowp[k] = w[k];
    if (!chroma) { rowp[3] = w[3]; rowp[4] = w[4]; }
  }
  wsync();
  if (any1 && (live & 0xfff00000u) == 0) {
    // ---- horizontal edges, only the macroblock edge is live: rows -4..3 of the column (chroma: -2..1) --------------------------------
    int p3 = 0, p2 = 0, p1, p0, q0, q1, q2 = 0, q3 = 0;
    const int bs = (int) ((bs1.x >> sh8) & 0xff);
    const int tc0 = (int) ((tct >> (8 * (bs & 3))) & 0xff);
    if (chroma) {
      LDS uint8_t* cp = &L.C[cpl][tC (-2, li)];
      p1 = cp[0]; p0 = cp[16]; q0 = cp[32]; q1 = cp[48];
      filter_edge (p3, p2, p1, p0, q0, q1, q2, q3, (alt | bet) ? bs : 0, alt, bet, tc0, true);
      cp[16] = (uint8_t)p0; cp[32] = (uint8_t)q0;
    } else {
      LDS uint8_t* cp = &L.T[tY (-4, li)];
      p3 = cp[0]; p2 = cp[32]; p1 = cp[64]; p0 = cp[96]; q0 = cp[128]; q1 = cp[160]; q2 = cp[192]; q3 = cp[224];
      filter_edge (p3, p2, p1, p0, q0, q1, q2, q3, (alt | bet) ? bs : 0, alt, bet, tc0, false);
      cp[32] = (uint8_t)p2; cp[64] = (uint8_t)p1; cp[96] = (uint8_t)p0; cp[128] = (uint8_t)q0; cp[160] = (uint8_t)q1; cp[192] = (uint8_t)q2;
    }
  } else if (any1) {
    // ---- horizontal edges: line = column li, luma rows -4..15 = v[0..19], chroma rows -2..7 = v[2..11] -----------
    LDS uint8_t* colp = chroma ? &L.C[cpl][tC (-2, li)] : &L.T[tY (-4, li)];
#pragma unroll
    for (int i = 0; i < 20; i++) {
      const int oc = (i < 2 ? 0 : i > 11 ? 9 : i - 2) * 16;
      v[i] = colp[chroma ? oc : i * 32];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if ((((k == 1 ? live | live >> 4 : live) >> (16 + 4 * k)) & 15u) == 0) continue;      // (scalar)
      const uint32_t bl = k == 0 ? bs1.x : k == 1 ? bs1.y : k == 2 ? bs1.z : bs1.w;
      const uint32_t bc = k == 0 ? bs1.x : k == 1 ? bs1.z : 0u;
      const int bs = (int) (((chroma ? bc : bl) >> sh8) & 0xff);
      const int alpha = k == 0 ? alt : alc, beta = k == 0 ? bet : bec;
      const uint32_t tcw = k == 0 ? tct : tcc_h;
      filter_edge (v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3], v[4 * k + 4], v[4 * k + 5], v[4 * k + 6], v[4 * k + 7],
                   (alpha | beta) ? bs : 0, alpha, beta, (int) ((tcw >> (8 * (bs & 3))) & 0xff), chroma);
    }
#pragma unroll
    for (int i = 1; i < 19; i++) {
      if (i >= 2 && i <= 11) colp[chroma ? (i - 2) * 16 : i * 32] = (uint8_t)v[i];
      else if (!chroma) colp[i * 32] = (uint8_t)v[i];
    }
  }
  wsync();
}

// ------------------------------------------------------------------------------------------------
// one macroblock: reconstruct (WelsTargetMbConstruction) + deblock (WelsDeblockingMb) + write back
// ------------------------------------------------------------------------------------------------
struct RowBufs {
  LDS uint8_t* lineCur; const LDS uint8_t* lineTop;     // unfiltered bottom rows
  LDS uint8_t* fCur; const LDS uint8_t* fTop;           // filtered bottom 4 / 2 rows
  int LY, LC, FW;                               // layout of a slot
};

// Wait until the wave of the row above has finished `need` (encoded as in WgLds::progress); nothing to wait for in the first row.
__device__ __forceinline__ void wait_row_above (volatile LDS int* prog_above, int need, bool have_row_above) {
  if (have_row_above) {
    while ((int) (*prog_above - need) < 0) __builtin_amdgcn_s_sleep (1);
    wsync();
  }
}

__device__ __forceinline__ void process_mb (const FrameCtx& F, LDS WaveLds& L, const LDS WgLds& G, const RowBufs& B, const Pref& pf,
                                            int mbx, int mby, int par, int& slc_id, bool pub_line, bool pub_left, int lane,
                                            volatile LDS int* prog_above, int need_above
#ifdef LH264_STAMP
                                            , unsigned long long& st_t0, unsigned long long* st_acc
#endif
                                           ) {
  LDS uint8_t* T = L.T;
  // The row above must be two macroblocks ahead (top-right neighbour of intra prediction; raster order of the in-loop filter).  Only
  // an INTRA macroblock needs that before it starts: inter prediction and the residual read nothing of the row above, so every other
  // macroblock predicts first and waits where its filter takes the neighbours' filtered samples - the wait overlaps its own work.
  const bool early_wait = (uni ((int)pf.rec) & LH264_MB_INTRA) != 0;         // lane 0 holds the record's first dword: mb_type in the low half
  if (early_wait) wait_row_above (prog_above, need_above, mby > 0);
  // ---- 0. stage the prefetched records ----------------------------------------------------------
  if (lane < 32) L.rec[par][lane] = pf.rec; else L.trec[lane - 32] = pf.rec;
  wsync();
  const RecView m = { (const LDS uint8_t*)L.rec[par]}, lm = { (const LDS uint8_t*)L.rec[par ^ 1]}, tm = { (const LDS uint8_t*)L.trec};
  const int mb_type = uni (m.mb_type());
  const int cbp = uni (m.cbp());
  const int mflags = uni (m.flags());
  const int sid = uni (m.slice_id());
  if (sid != slc_id) {              // refresh the cached slice record
    if (lane < 58) L.slc[lane] = ((const GLB uint32_t*) (F.slices + sid))[lane];
    slc_id = sid;
    wsync();
  }
  const SliceView sl = { (const LDS uint8_t*)L.slc};
  const bool t8 = mflags & LH264_MBF_T8x8;
  const bool covered = mb_type != 0;
  const bool intra = (mb_type & LH264_MB_INTRA) != 0;

  const int b = lane >> 2, r = lane & 3;
  const int bx = (b & 1) | ((b >> 2) & 1) << 1, by = ((b >> 1) & 1) | ((b >> 3) & 1) << 1;
  const int ly = 4 * by + r, lx0 = 4 * bx;
  const int cp = (lane >> 4) & 1, cj = (lane >> 2) & 3;
  const int cy = 4 * (cj >> 1) + r, cx0 = 4 * (cj & 1);

  STAMP (0);
  // ---- 1. unfiltered neighbours -> tile (only intra prediction reads them) -------------------------
  if (intra && mby > 0) {
    if (lane < 8) * (LDS uint32_t*)&T[tY (-1, -4 + 4 * lane)] = * (const LDS uint32_t*)&B.lineTop[16 + 16 * mbx - 4 + 4 * lane];
    else if (lane < 16) {
      const int p = (lane - 8) >> 2, q = (lane - 8) & 3;
      * (LDS uint32_t*)&L.C[p][tC (-1, -4 + 4 * q)] = * (const LDS uint32_t*)&B.lineTop[B.LY + p * B.LC + 8 + 8 * mbx - 4 + 4 * q];
    }
  }
  if (intra && mbx > 0) {
    if (lane >= 16 && lane < 32) T[tY (lane - 16, -1)] = L.leftY[lane - 16];
    else if (lane >= 32 && lane < 48) { const int p = (lane - 32) >> 3, q = (lane - 32) & 7; L.C[p][tC (q, -1)] = L.leftC[p][q]; }
  }
  wsync();

  STAMP (1);
  if (covered) {
    // ---- 2. residual -> R ------------------------------------------------------------------------
    const bool i16 = mb_type == LH264_MB_I16x16;
    const bool has_res = cbp != 0 || i16;
    if (mb_type != LH264_MB_IPCM) {
      if (has_res) {
        const int dcw = sl.luma_dc_weight();
#ifndef LH264_ABL_NORES      // timing ablation only (wrong pictures)
        residual_phase (L, pf.l, pf.c, mb_type, cbp, t8, m.qp_y(), dcw ? dcw : 16, lane);
#endif
      } else if (intra) zero_residual (L, lane);
    }
    STAMP (2);
    // ---- 3. prediction + residual -> tile -----------------------------------------------------
    if (mb_type == LH264_MB_IPCM) {
      // samples travel in the coefficient slot, row-major (decode_slice.cpp:3213-3263 copies them at parse time)
      const v2i v = pf.l;
      const int yy = lane >> 2, xx = 4 * (lane & 3);
      * (LDS uint32_t*)&T[tY (yy, xx)] = (v.x & 0xff) | ((v.x >> 16) & 0xff) << 8 | (v.y & 0xff) << 16 | ((v.y >> 16) & 0xff) << 24;
      if (lane < 32) {
        const v2i cv = pf.c;
        const int p = lane >> 4, yy2 = (lane >> 1) & 7, xx2 = 4 * (lane & 1);
        * (LDS uint32_t*)&L.C[p][tC (yy2, xx2)] = (cv.x & 0xff) | ((cv.x >> 16) & 0xff) << 8 | (cv.y & 0xff) << 16 | ((cv.y >> 16) & 0xff) << 24;
      }
    } else if (intra) {
      if (i16) intra16_phase (L, uni (m.intra_mode (0)), lane);
      else if (mb_type == LH264_MB_I8x8) intra8x8_phase (L, m, lane);
      else intra4x4_phase (L, m, lane);
      intra_chroma_phase (L, uni (m.chroma_mode()), lane);
    } else if (mb_type & LH264_MB_INTER) {
      InterCtx ic;
      ic.prev_dy = F.prev_dy; ic.sy = F.sy; ic.sc = F.sc; ic.mb_w = F.mb_w; ic.mb_h = F.mb_h;
      ic.prev_base = F.prev_base; ic.prev_h = F.prev_h; ic.nw = F.nw;
#ifndef LH264_ABL_NOINTER    // timing ablation only (wrong pictures)
      inter_phase (ic, L, G, m, sl, mb_type, mbx, mby, has_res, lane);
#endif
    }
  } else {
    // macroblock not covered by any slice (lost data): pass the picture's current samples through
    * (LDS uint32_t*)&T[tY (ly, lx0)] = * (const GLB uint32_t*) (F.dy + (size_t) (mby * 16 + ly) * F.sy + mbx * 16 + lx0);
    if (lane < 32) * (LDS uint32_t*)&L.C[cp][tC (cy, cx0)] = * (const GLB uint32_t*) ((cp ? F.dv : F.du) + (size_t) (mby * 8 + cy) * F.sc + mbx * 8 + cx0);
  }
  wsync();

  STAMP (3);
  // ---- 4. publish unfiltered bottom row / right column, when an intra macroblock will read them -----
  if (pub_line) {
    if (lane < 4) * (LDS uint32_t*)&B.lineCur[16 + 16 * mbx + 4 * lane] = * (const LDS uint32_t*)&T[tY (15, 4 * lane)];
    else if (lane < 8) { const int p = (lane - 4) >> 1, q = (lane - 4) & 1; * (LDS uint32_t*)&B.lineCur[B.LY + p * B.LC + 8 + 8 * mbx + 4 * q] = * (const LDS uint32_t*)&L.C[p][tC (7, 4 * q)]; }
  }
  if (pub_left) {
    if (lane >= 16 && lane < 32) L.leftY[lane - 16] = T[tY (lane - 16, 15)];
    else if (lane >= 32 && lane < 48) { const int p = (lane - 32) >> 3, q = (lane - 32) & 7; L.leftC[p][q] = L.C[p][tC (q, 7)]; }
  }
  if (pub_line || pub_left) wsync();

  STAMP (4);
  // ---- 5. in-loop filter inside the tile ---------------------------------------------------------
  const int stype = sl.slice_type();
  const int idc = sl.deblock_idc();
  const bool filt = covered && !(F.flags & LH264_JOB_NO_DEBLOCK) && idc != 1 && (stype == 0 || stype == 2);
  bool left_av = mbx > 0, top_av = mby > 0;
  if (filt && idc == 2) {           // DeblockingAvailableNoInterlayer deblocking.cpp:354-369
    if (left_av) left_av = lm.slice_id() == sid;
    if (top_av) top_av = tm.slice_id() == sid;
  }
  if (!early_wait) wait_row_above (prog_above, need_above, mby > 0);
  // neighbours' filtered samples: left 4 columns carried from the previous step, top rows from `fline`
  if (mbx > 0) {
    if (lane < 20) * (LDS uint32_t*)&T[tY (lane - 4, -4)] = L.lfY[lane];
    else if (lane < 40) { const int p = (lane - 20) / 10, q = (lane - 20) % 10; * (LDS uint32_t*)&L.C[p][tC (q - 2, -4)] = L.lfC[p][q]; }
  }
  if (mby > 0) {
    if (lane >= 40 && lane < 56) {
      const int rr = (lane - 40) >> 2, q = lane & 3;
      * (LDS uint32_t*)&T[tY (-4 + rr, 4 * q)] = * (const LDS uint32_t*)&B.fTop[rr * B.FW + mbx * 16 + 4 * q];
    } else if (lane >= 56) {
      const int p = (lane - 56) >> 2, rr = (lane >> 1) & 1, q = lane & 1;
      * (LDS uint32_t*)&L.C[p][tC (-2 + rr, 4 * q)] = * (const LDS uint32_t*)&B.fTop[4 * B.FW + (p * 2 + rr) * (B.FW >> 1) + mbx * 8 + 4 * q];
    }
  }
  wsync();
  STAMP (5);
  if (filt) deblock_phase (L, G, m, lm, tm, sl, left_av, top_av, lane);
  STAMP (6);
  // carry the (filtered) right 4 columns to the next macroblock of this row
  if (lane < 20) L.lfY[lane] = * (const LDS uint32_t*)&T[tY (lane - 4, 12)];
  else if (lane < 40) { const int p = (lane - 20) / 10, q = (lane - 20) % 10; L.lfC[p][q] = * (const LDS uint32_t*)&L.C[p][tC (q - 2, 4)]; }

  // ---- 6. write the samples that became final: 16x16 window at (-4,-3), chroma 8x8 at (-4,-1) --------
  const bool last_x = mbx == F.mb_w - 1, last_y = mby == F.mb_h - 1;
#ifndef LH264_ABL_NOSTORE    // timing ablation only (no pictures)
  {
    const int rr = -3 + (lane >> 2), cc = -4 + 4 * (lane & 3);
    if ((rr >= 0 || mby > 0) && (cc >= 0 || mbx > 0))
      * (GLB uint32_t*) (F.dy + (ptrdiff_t) (mby * 16 + rr) * F.sy + mbx * 16 + cc) = * (const LDS uint32_t*)&T[tY (rr, cc)];
    if (lane < 32) {
      const int p = lane >> 4, r2 = -1 + ((lane >> 1) & 7), c2 = -4 + 4 * (lane & 1);
      if ((r2 >= 0 || mby > 0) && (c2 >= 0 || mbx > 0))
        * (GLB uint32_t*) ((p ? F.dv : F.du) + (ptrdiff_t) (mby * 8 + r2) * F.sc + mbx * 8 + c2) = * (const LDS uint32_t*)&L.C[p][tC (r2, c2)];
    }
  }
  if (last_x) {                     // right-most columns never get a successor
    if (lane < 16) {
      const int rr = -3 + lane;
      if (rr >= 0 || mby > 0) * (GLB uint32_t*) (F.dy + (ptrdiff_t) (mby * 16 + rr) * F.sy + mbx * 16 + 12) = * (const LDS uint32_t*)&T[tY (rr, 12)];
    } else if (lane < 32) {
      const int p = (lane - 16) >> 3, r2 = -1 + (lane & 7);
      if (r2 >= 0 || mby > 0) * (GLB uint32_t*) ((p ? F.dv : F.du) + (ptrdiff_t) (mby * 8 + r2) * F.sc + mbx * 8 + 4) = * (const LDS uint32_t*)&L.C[p][tC (r2, 4)];
    }
  }
  if (last_y) {                     // bottom rows never get a successor
    if (lane < 15) {
      const int rr = 13 + lane / 5, cc = -4 + 4 * (lane % 5);
      if ((cc >= 0 || mbx > 0) && (cc < 12 || last_x))
        * (GLB uint32_t*) (F.dy + (ptrdiff_t) (mby * 16 + rr) * F.sy + mbx * 16 + cc) = * (const LDS uint32_t*)&T[tY (rr, cc)];
    } else if (lane >= 16 && lane < 22) {
      const int p = (lane - 16) / 3, c2 = -4 + 4 * ((lane - 16) % 3);
      if ((c2 >= 0 || mbx > 0) && (c2 < 4 || last_x))
        * (GLB uint32_t*) ((p ? F.dv : F.du) + (ptrdiff_t) (mby * 8 + 7) * F.sc + mbx * 8 + c2) = * (const LDS uint32_t*)&L.C[p][tC (7, c2)];
    }
  }
#endif
  STAMP (7);
  // filtered bottom rows for the row below: tile rows 12..15 x cols [-4,11] (+ [12,15] at the last MB)
  if (!last_y) {
    if (lane < 20) {
      const int rr = lane / 5, q = lane % 5, cc = -4 + 4 * q;
      if ((cc >= 0 || mbx > 0) && (cc < 12 || last_x))
        * (LDS uint32_t*)&B.fCur[rr * B.FW + mbx * 16 + cc] = * (const LDS uint32_t*)&T[tY (12 + rr, cc)];
    } else if (lane >= 32 && lane < 44) {
      const int i = lane - 32, p = i / 6, rr = (i % 6) / 3, q = i % 3, c2 = -4 + 4 * q;
      if ((c2 >= 0 || mbx > 0) && (c2 < 4 || last_x))
        * (LDS uint32_t*)&B.fCur[4 * B.FW + (p * 2 + rr) * (B.FW >> 1) + mbx * 8 + c2] = * (const LDS uint32_t*)&L.C[p][tC (6 + rr, c2)];
    }
  }
  wsync();
}

// ------------------------------------------------------------------------------------------------
// ExpandReferencingPicture (expand_pic.cpp:145-174): every padding sample = nearest picture sample.
// Done band by band by single waves as the rows of a picture become final.
// ------------------------------------------------------------------------------------------------
// left and right padding of picture rows [r0, r1): every lane replicates one edge sample into 16 (PADW == 32: luma)
// or 16 (PADW == 16: chroma, the whole side) bytes
template <int PADW>
__device__ __forceinline__ void expand_rows (GLB uint8_t* p, int stride, int w, int r0, int r1, int lane) {
  constexpr int UPS = PADW / 16;                       // 16-byte units per side
  const int n = (r1 - r0) * 2 * UPS;
  for (int i = lane; i < n; i += 64) {
    const int rr = r0 + i / (2 * UPS), j = i % (2 * UPS);
    const bool right = j >= UPS;
    GLB uint8_t* row = p + (ptrdiff_t)rr * stride;
    const uint32_t v = 0x01010101u * row[right ? w - 1 : 0];
    GLB uint8_t* d = right ? row + w + 16 * (j - UPS) : row - PADW + 16 * j;
    * (GLB v4u*)d = (v4u) (v);
  }
}
// the band of PADW rows above (bottom == false) or below the picture, full padded width.  Each lane owns a column
// group of sizeof (U) bytes of the source row (replicated edge samples in the corners) and stores it to the band rows.
template <int PADW, typename U>
__device__ __forceinline__ void expand_band (GLB uint8_t* p, int stride, int w, int h, bool bottom, int lane) {
  constexpr int UB = (int)sizeof (U);
  const int nu = (w + 2 * PADW) / UB;                  // units per padded row
  const GLB uint8_t* src = p + (ptrdiff_t) (bottom ? h - 1 : 0) * stride;
  GLB uint8_t* dst0 = p + (ptrdiff_t) (bottom ? h : -PADW) * stride - PADW;
  const int rpi = nu >= 64 ? 1 : 64 / nu;              // band rows covered by one wave-wide store
  for (int c0 = 0; c0 < nu; c0 += 64) {
    const int cu = c0 + (rpi > 1 ? lane % nu : lane);   // this lane's unit
    const int rsub = rpi > 1 ? lane / nu : 0;
    const bool act = rpi > 1 ? lane < rpi * nu : cu < nu;
    U v = (U) (0u);
    if (act) {
      const int col = cu * UB - PADW;                  // first picture column of the unit
      if (col < 0) v = (U) (0x01010101u * src[0]);
      else if (col >= w) v = (U) (0x01010101u * src[w - 1]);
      else v = * (const GLB U*) (src + col);
    }
    for (int rr = rsub; rr < PADW; rr += rpi)
      if (act) * (GLB U*) (dst0 + (ptrdiff_t)rr * stride + cu * UB) = v;
  }
}
struct PadCtx { GLB uint8_t* dy; GLB uint8_t* du; GLB uint8_t* dv; int sy, sc, mb_w, mb_h; };
// padding that depends on macroblock row r only (its samples must be final and visible)
__device__ __noinline__ void pad_mb_row (const PadCtx F, int r, int lane) {
  const int W = F.mb_w * 16, H = F.mb_h * 16;
  expand_rows<LH264_PAD_LUMA> (F.dy, F.sy, W, 16 * r, 16 * r + 16, lane);
  expand_rows<LH264_PAD_CHROMA> (F.du, F.sc, W >> 1, 8 * r, 8 * r + 8, lane);
  expand_rows<LH264_PAD_CHROMA> (F.dv, F.sc, W >> 1, 8 * r, 8 * r + 8, lane);
  if (r == 0) {
    expand_band<LH264_PAD_LUMA, v4u> (F.dy, F.sy, W, H, false, lane);
    expand_band<LH264_PAD_CHROMA, v2u> (F.du, F.sc, W >> 1, H >> 1, false, lane);
    expand_band<LH264_PAD_CHROMA, v2u> (F.dv, F.sc, W >> 1, H >> 1, false, lane);
  }
  if (r == F.mb_h - 1) {
    expand_band<LH264_PAD_LUMA, v4u> (F.dy, F.sy, W, H, true, lane);
    expand_band<LH264_PAD_CHROMA, v2u> (F.du, F.sc, W >> 1, H >> 1, true, lane);
    expand_band<LH264_PAD_CHROMA, v2u> (F.dv, F.sc, W >> 1, H >> 1, true, lane);
  }
}

__global__ void __launch_bounds__ (512, LH264_MIN_WAVES)
recon_chain_kernel (const lh264_frame_job_t* __restrict__ jobs, const int32_t* __restrict__ chain_first, int n_chains,
                    int slot_bytes) {
  extern __shared__ __attribute__ ((aligned (16))) uint8_t smem_generic[];
  LDS uint8_t* smem = (LDS uint8_t*) (uintptr_t) (uint32_t) (uintptr_t)smem_generic;
  const int NW = blockDim.x >> 6;
  const int tid = threadIdx.x;
  const int wave = uni (tid >> 6);
  const int lane = tid & 63;
  LDS WgLds& G = * (LDS WgLds*)smem;
  LDS WaveLds* wl = (LDS WaveLds*) (smem + ((sizeof (WgLds) + 15) & ~15));
  LDS uint8_t* slots = (LDS uint8_t*) (wl + NW);                           // [NW + 1][slot_bytes]
  LDS WaveLds& L = wl[wave];
  const int NL = NW + 1;
  volatile LDS int* progress = G.progress;
  volatile LDS int* stored = G.stored;

  for (int i = tid; i < (int)sizeof (G.tab); i += blockDim.x) G.tab[i] = kTables[i];
  if (tid < 16) { progress[tid] = 0; stored[tid] = -1; }
  if (lane == 0) L.known_prefix = -1;
  __syncthreads();

  const int chain = blockIdx.x;
  if (chain >= n_chains) return;
  const int first = chain_first[chain], last = chain_first[chain + 1];

  // this wave's cursor over the chain's rows: global row g = base + row of job ji; the wave owns g == wave (mod NW)
  int ji = first, row = wave, base = 0, cur_ji = -1;
  int jw = 0;                               // rows this wave has started so far
  bool overlap_ok = false;
  int slc_id = -1;
  FrameCtx F;
  F.nw = NW; F.prev_dy = 0; F.prev_base = 0; F.prev_h = 1;
  RowBufs B;
  int lu = 0;
  STAMP_DECL
  for (;;) {
    while (ji < last) {
      const int h = jobs[ji].mb_h;
      if (row < h) break;
      row -= h; base += h; ji++;
    }
    if (ji >= last) break;
    if (ji != cur_ji) {                      // entering a new frame
      const lh264_frame_job_t* J = jobs + ji;
      F.mbs = to_glb<const lh264_mb_t> (J->mbs_dev); F.coeffs = to_glb<const int16_t> (J->coeffs_dev); F.slices = to_glb<const lh264_slice_t> (J->slices_dev);
      F.dy = to_glb<uint8_t> (J->dst.y_dev); F.du = to_glb<uint8_t> (J->dst.u_dev); F.dv = to_glb<uint8_t> (J->dst.v_dev);
      F.mb_w = J->mb_w; F.mb_h = J->mb_h; F.sy = J->stride_y; F.sc = J->stride_c; F.flags = J->flags;
      B.LY = F.mb_w * 16 + 48; B.LC = F.mb_w * 8 + 24; B.FW = F.mb_w * 16;
      lu = B.LY + 2 * B.LC;
      wsync();
      if (lane < LH264_MAX_REFS * 3) L.refp[lane / 3][lane % 3] = ((const uint64_t*)&J->ref[lane / 3])[lane % 3];
      overlap_ok = false;
      F.prev_dy = 0;
      if (ji > first) {
        const lh264_frame_job_t* P = J - 1;
        F.prev_dy = (uint64_t) (uintptr_t)P->dst.y_dev; F.prev_h = P->mb_h; F.prev_base = base - P->mb_h;
#ifndef LH264_NO_OVERLAP
        // two frames may be in flight only if this frame's picture is not one the previous frame still reads
        overlap_ok = true;
        for (int k = 0; k < LH264_MAX_REFS; k++) overlap_ok = overlap_ok && P->ref[k].y_dev != J->dst.y_dev;
        overlap_ok = overlap_ok && P->dst.y_dev != J->dst.y_dev;
#endif
      }
      slc_id = -1;
      cur_ji = ji;
      wsync();
    }
    // everything older than the previous frame must be complete (and the previous frame too if pictures alias)
    if (ji > first) wait_prefix (G, L, (overlap_ok ? F.prev_base : base) - 1, NW);

    STAMP (10);
    const int g = base + row;
    LDS uint8_t* cur = slots + (g % NL) * slot_bytes;
    const LDS uint8_t* top = slots + ((g + NL - 1) % NL) * slot_bytes;
    B.lineCur = cur; B.lineTop = top; B.fCur = cur + lu; B.fTop = top + lu;
    const int wprev = (wave + NW - 1) % NW;
    const int jprev = row > 0 ? (g - 1 - wprev) / NW : 0;
    Pref pf = prefetch_mb (F, row * F.mb_w, row > 0, lane);
    unsigned long long below = 0;            // bit i: macroblock 64*(x/64)+i of the row below is intra
    unsigned long long ahead = 0;            // bit i: macroblock 64*(x/64)+i+1 of this row is intra
    for (int x = 0; x < F.mb_w; x++) {
      if ((x & 63) == 0) {                   // (before the prefetch: these two loads are waited for at once)
        int t = 0, u = 0;
        if (row + 1 < F.mb_h && x + lane < F.mb_w) t = * (const GLB uint16_t*) (F.mbs + (size_t) (row + 1) * F.mb_w + x + lane);
        if (x + 1 + lane < F.mb_w) u = * (const GLB uint16_t*) (F.mbs + (size_t)row * F.mb_w + x + 1 + lane);
        below = __ballot ((t & LH264_MB_INTRA) != 0);
        ahead = __ballot ((u & LH264_MB_INTRA) != 0);
      }
      Pref nx = pf;
      if (x + 1 < F.mb_w) nx = prefetch_mb (F, row * F.mb_w + x + 1, row > 0, lane);
      // does an intra macroblock read this one's unfiltered bottom row (below-left, below, below-right) / right column?
      // (the type of the next macroblock comes from the row's mask, not from the record just requested: looking at that one here
      // would wait for the prefetch the moment it is issued)
      const int xi = x & 63;
      const bool pub_line = row + 1 < F.mb_h && (xi == 0 || xi >= 62 || ((below >> (xi - 1)) & 7) != 0);
      const bool pub_left = x + 1 < F.mb_w && ((ahead >> xi) & 1) != 0;
      const int need = (jprev << 12) | min (x + 2, F.mb_w);
      STAMP (8);
      // The lane index goes in opaque: otherwise every lane-dependent address and mask of the macroblock's steps is hoisted out of
      // this loop, and at 128 registers 47 of them live in scratch (12.8 KB per wave, more than the L2 holds for a full chip: the
      // reloads were most of the kernel's HBM traffic).  Recomputing them per macroblock costs a few dozen VALU instructions.
#ifndef LH264_HOIST_LANE
      int lane_mb = lane;
      asm volatile ("" : "+v" (lane_mb));
#else
      const int lane_mb = lane;
#endif
#ifdef LH264_STAMP
      process_mb (F, L, G, B, pf, x, row, x & 1, slc_id, pub_line, pub_left, lane_mb, &progress[wprev], need, st_t0, st_acc);
#else
      process_mb (F, L, G, B, pf, x, row, x & 1, slc_id, pub_line, pub_left, lane_mb, &progress[wprev], need);
#endif
      if (lane == 0) progress[wave] = (jw << 12) | (x + 1);
      pf = nx;
    }
    // ---- row end: wait for this row's stores, pad what became final, publish ----------------------------------
    __builtin_amdgcn_fence (__ATOMIC_RELEASE, "workgroup");
    if (!(F.flags & LH264_JOB_NO_EXPAND)) {
      PadCtx pc;
      pc.dy = F.dy; pc.du = F.du; pc.dv = F.dv; pc.sy = F.sy; pc.sc = F.sc; pc.mb_w = F.mb_w; pc.mb_h = F.mb_h;
      if (row > 0) {                         // the row above is final now (its last 3 sample rows were written by this wave)
        while ((int) (stored[wprev] - jprev) < 0) __builtin_amdgcn_s_sleep (1);
        __builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "workgroup");
        pad_mb_row (pc, row - 1, lane);
      }
      if (row == F.mb_h - 1) pad_mb_row (pc, row, lane);
      __builtin_amdgcn_fence (__ATOMIC_RELEASE, "workgroup");
    }
    if (lane == 0) stored[wave] = jw;
    STAMP (9);
    jw++;
    row += NW;
  }
  STAMP_FLUSH;
}

#ifdef LH264_STAMP
void read_stamps (unsigned long long* out, bool reset) {
  (void)hipMemcpyFromSymbol (out, HIP_SYMBOL (g_stamps), sizeof (g_stamps));
  if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol (HIP_SYMBOL (g_stamps), z, sizeof (z)); }
}
#endif
size_t wave_lds_bytes() { return sizeof (WaveLds); }
size_t wg_lds_bytes() { return (sizeof (WgLds) + 15) & ~ (size_t)15; }

}  // namespace lh264
