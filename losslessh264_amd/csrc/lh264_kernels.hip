// lh264_kernels.hip - gfx950 (MI355X, wave64) kernels of the decode-reconstruct hot path.
//
// One workgroup per chain of frames (the frames of one stream, in decode order), one wavefront
// per macroblock row (row r is served by wave r % NW), row r running at least two macroblocks
// behind row r-1 -- the raster dependency of intra prediction and of the in-loop filter
// (reference loop order: WelsTargetSliceConstruction decode_slice.cpp:110-206, WelsDeblockingFilterSlice
// deblocking.cpp:872-934).  Per macroblock a wave
//   1. stages the unfiltered neighbour samples (line buffers in LDS) into its private LDS tile,
//   2. inverse-transforms the 384 coefficients (coalesced 768-byte read; 4x4 butterflies across
//      lane quads with DPP; 8x8 / DC transforms through LDS),
//   3. predicts (intra from the tile, inter = quarter-pel MC straight from the padded reference
//      planes) and adds the residual                      [RecI*/GetInterPred rec_mb.cpp],
//   4. publishes the unfiltered bottom row / right column for later intra prediction,
//   5. deblocks the macroblock's left and top edges inside the tile (top rows come back from HBM,
//      written two steps earlier by the wave of the row above)          [WelsDeblockingMb],
//   6. writes the macroblock back and releases its progress counter.
// After the last row the workgroup pads the picture (ExpandReferencingPicture) and moves on to the
// next frame of the chain, which may use it as a reference.
//
// Integer work on u8/int16: no MFMA.  Bounds: HBM traffic (1,280 B / intra MB algorithmic) and
// VALU issue; see DESIGN.md.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/lh264.h"

namespace lh264 {

// ---- tables (H.264 Tables 8-16/8-17; reference copies: deblocking.cpp:89-125) ----------------
__constant__ uint8_t kAlpha[52] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 4, 4, 5, 6, 7, 8, 9, 10, 12, 13,
                                   15, 17, 20, 22, 25, 28, 32, 36, 40, 45, 50, 56, 63, 71, 80, 90, 101, 113, 127, 144, 162, 182, 203, 226, 255, 255
                                  };
__constant__ uint8_t kBeta[52] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 6, 6, 7, 7,
                                  8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 14, 14, 15, 15, 16, 16, 17, 17, 18, 18
                                 };
__constant__ uint8_t kTc0[52][4] = {
  {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0},
  {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 1}, {0, 0, 0, 1}, {0, 0, 0, 1},
  {0, 0, 0, 1}, {0, 0, 1, 1}, {0, 0, 1, 1}, {0, 1, 1, 1}, {0, 1, 1, 1}, {0, 1, 1, 1}, {0, 1, 1, 1}, {0, 1, 1, 2}, {0, 1, 1, 2}, {0, 1, 1, 2},
  {0, 1, 1, 2}, {0, 1, 2, 3}, {0, 1, 2, 3}, {0, 2, 2, 3}, {0, 2, 2, 4}, {0, 2, 3, 4}, {0, 2, 3, 4}, {0, 3, 3, 5}, {0, 3, 4, 6}, {0, 3, 4, 6},
  {0, 4, 5, 7}, {0, 4, 5, 8}, {0, 4, 6, 9}, {0, 5, 7, 10}, {0, 6, 8, 11}, {0, 6, 8, 13}, {0, 7, 10, 14}, {0, 8, 11, 16}, {0, 9, 12, 18}, {0, 10, 13, 20},
  {0, 11, 15, 23}, {0, 13, 17, 25}
};
__constant__ uint8_t kNormAdjust0[6] = {10, 11, 13, 14, 16, 18};

// ---- per-wave LDS ------------------------------------------------------------------------------
struct WaveLds {
  uint8_t  T[20 * 32];       // luma tile: rows -4..15, cols -4..27      idx (r+4)*32 + (c+4)
  uint8_t  C[2][10 * 16];    // chroma tiles: rows -2..7, cols -4..11    idx (r+2)*16 + (c+4)
  int16_t  R[384];           // coefficient / residual scratch
  uint8_t  leftY[16];        // unfiltered right column of the previous MB (intra neighbours)
  uint8_t  leftC[2][8];
  uint32_t lfY[16];          // filtered right 4 columns of the previous MB (deblock neighbours)
  uint32_t lfC[2][8];
  int32_t  mvi[16][4];       // per 4x4 block: {luma src offset, chroma src offset, fractions|slot, weight info}
  uint8_t  bs[32];           // boundary strengths [dir][edge][segment]
  uint8_t  E[32];            // filtered I8x8 edge
  int32_t  S[64];            // DC transform scratch
};

__device__ __forceinline__ int tY (int r, int c) { return (r + 4) * 32 + (c + 4); }
__device__ __forceinline__ int tC (int r, int c) { return (r + 2) * 16 + (c + 4); }
__device__ __forceinline__ int clip_u8 (int v) { return min (max (v, 0), 255); }
__device__ __forceinline__ int clip3 (int v, int lo, int hi) { return min (max (v, lo), hi); }
__device__ __forceinline__ int zidx (int bx, int by) { return (bx & 1) | ((by & 1) << 1) | ((bx >> 1) << 2) | ((by >> 1) << 3); }
__device__ __forceinline__ int tab_idx (int v) { return clip3 (v, 0, 51); }

// order LDS traffic between the lanes of one wave (DS ops of a wave execute in order; this only
// stops the compiler from moving them)
__device__ __forceinline__ void wsync() {
  __builtin_amdgcn_fence (__ATOMIC_ACQ_REL, "wavefront");
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

struct FrameCtx {
  const lh264_mb_t* mbs; const int16_t* coeffs; const lh264_slice_t* slices;
  uint8_t* dy; uint8_t* du; uint8_t* dv;
  const lh264_frame_job_t* job;
  int mb_w, mb_h, sy, sc, flags;
};

// ------------------------------------------------------------------------------------------------
// I4x4 directional prediction of one sample from the edge e[0..12] = L3 L2 L1 L0 TL T0..T7
// (get_intra_predictor.cpp:54-380).  hi = 12, or 8 for the *_TOP variants (p[3,-1] replicated).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int edge4_load (const uint8_t* T, int bx, int by, int j, int hi) {
  j = clip3 (j, 0, hi);
  const int r = (j < 4) ? (4 * by + 3 - j) : (4 * by - 1);
  const int c = (j < 4) ? (4 * bx - 1) : (4 * bx + j - 5);
  return T[tY (r, c)];
}

__device__ __forceinline__ int pred4_dir (const uint8_t* T, int bx, int by, int mode, int x, int y) {
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

// ------------------------------------------------------------------------------------------------
// motion compensation of one 4x1 luma strip / chroma sample straight from the padded reference
// (mc.cpp:142-380).  src points at the integer sample of the strip's first pixel.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void mc_luma_strip (const uint8_t* __restrict__ src, int st, int fx, int fy, int out[4]) {
  if ((fx | fy) == 0) {
#pragma unroll
    for (int i = 0; i < 4; i++) out[i] = src[i];
    return;
  }
  // window rows -2..3, cols -2..6 (only rows 0/1 when there is no vertical fraction)
  int w[6][9];
  const bool nv = fy != 0;
#pragma unroll
  for (int rr = 0; rr < 6; rr++) {
#pragma unroll
    for (int cc = 0; cc < 9; cc++) w[rr][cc] = (nv || rr == 2 || rr == 3) ? (int)src[(rr - 2) * st + cc - 2] : 0;
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int c = i + 2;              // column of the pixel in the window (compile-time)
    const int hh2 = clip_u8 ((tap6 (w[2][c - 2], w[2][c - 1], w[2][c], w[2][c + 1], w[2][c + 2], w[2][c + 3]) + 16) >> 5);
    const int hh3 = clip_u8 ((tap6 (w[3][c - 2], w[3][c - 1], w[3][c], w[3][c + 1], w[3][c + 2], w[3][c + 3]) + 16) >> 5);
    int v;
    if (fy == 0) {                    // a, b, c
      v = fx == 2 ? hh2 : (hh2 + (fx == 1 ? w[2][c] : w[2][c + 1]) + 1) >> 1;
    } else {
      const int vv0 = clip_u8 ((tap6 (w[0][c], w[1][c], w[2][c], w[3][c], w[4][c], w[5][c]) + 16) >> 5);
      const int vv1 = clip_u8 ((tap6 (w[0][c + 1], w[1][c + 1], w[2][c + 1], w[3][c + 1], w[4][c + 1], w[5][c + 1]) + 16) >> 5);
      if (fx == 0) {                  // d, h, n
        v = fy == 2 ? vv0 : (vv0 + (fy == 1 ? w[2][c] : w[3][c]) + 1) >> 1;
      } else if (fx == 2 || fy == 2) {  // f, i, j, k, q : need the centre sample
        int t[6];
#pragma unroll
        for (int k = 0; k < 6; k++) t[k] = sext16 (tap6 (w[0][c - 2 + k], w[1][c - 2 + k], w[2][c - 2 + k], w[3][c - 2 + k], w[4][c - 2 + k], w[5][c - 2 + k]));
        const int j = clip_u8 ((tap6 (t[0], t[1], t[2], t[3], t[4], t[5]) + 512) >> 10);
        if (fx == 2 && fy == 2) v = j;
        else if (fx == 2) v = (j + (fy == 1 ? hh2 : hh3) + 1) >> 1;
        else v = (j + (fx == 1 ? vv0 : vv1) + 1) >> 1;
      } else {                        // e, g, p, r
        v = ((fy == 1 ? hh2 : hh3) + (fx == 1 ? vv0 : vv1) + 1) >> 1;
      }
    }
    out[i] = v;
  }
}

__device__ __forceinline__ int mc_chroma_px (const uint8_t* __restrict__ s, int st, int dx, int dy) {
  if ((dx | dy) == 0) return s[0];
  const int A = (8 - dx) * (8 - dy), B = dx * (8 - dy), Cc = (8 - dx) * dy, D = dx * dy;
  return (A * s[0] + B * s[1] + Cc * s[st] + D * s[st + 1] + 32) >> 6;
}

// partition geometry of the 4x4 block (bx,by): origin and size of the motion partition that holds it
__device__ __forceinline__ void partition_of (int mb_type, const uint8_t* sub_type, int bx, int by,
                                              int& ox, int& oy, int& pw, int& ph) {
  ox = oy = 0; pw = ph = 16;
  if (mb_type == LH264_MB_P16x8) { ph = 8; oy = (by >> 1) << 3; }
  else if (mb_type == LH264_MB_P8x16) { pw = 8; ox = (bx >> 1) << 3; }
  else if (mb_type == LH264_MB_P8x8 || mb_type == LH264_MB_P8x8REF0) {
    const int q = ((by >> 1) << 1) | (bx >> 1);
    const int st = sub_type[q];
    ox = (bx >> 1) << 3; oy = (by >> 1) << 3; pw = ph = 8;
    if (st == LH264_SUB_8x4) { ph = 4; oy += (by & 1) << 2; }
    else if (st == LH264_SUB_4x8) { pw = 4; ox += (bx & 1) << 2; }
    else if (st == LH264_SUB_4x4) { pw = ph = 4; ox += (bx & 1) << 2; oy += (by & 1) << 2; }
  }
}

// ------------------------------------------------------------------------------------------------
// deblocking helpers (deblocking.cpp / deblocking_common.cpp)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int mb_ref4 (const lh264_mb_t* m, int b) { return m->ref_idx[((b >> 3) << 1) + ((b & 3) >> 1)]; }
__device__ __forceinline__ int bs_mv (const lh264_mb_t* a, int ia, const lh264_mb_t* b, int ib) {
  const int dx = abs ((int)a->mv[ia][0] - (int)b->mv[ib][0]), dy = abs ((int)a->mv[ia][1] - (int)b->mv[ib][1]);
  return (mb_ref4 (a, ia) != mb_ref4 (b, ib)) || dx >= 4 || dy >= 4;
}
__device__ __forceinline__ int nz8 (const lh264_mb_t* m, int o) { return m->nzc[o] | m->nzc[o + 1] | m->nzc[o + 4] | m->nzc[o + 5]; }

// boundary strength of segment `seg` of edge `e` in direction dir (0: vertical edges, 1: horizontal);
// DeblockingBsMarginalMBAvcbase deblocking.cpp:273-352, DeblockingBSInsideMB* :160-271, WelsDeblockingMb :815-862
__device__ int compute_bs (const lh264_mb_t* m, const lh264_mb_t* nb, int dir, int e, int seg, bool intra) {
  const int t8 = m->flags & LH264_MBF_T8x8;
  if (e == 0) {
    if (!nb) return 0;
    if (intra || (nb->mb_type & LH264_MB_INTRA)) return 4;
    const int t8n = nb->flags & LH264_MBF_T8x8;
    const int bc = dir == 0 ? seg * 4 : seg;
    const int bn = dir == 0 ? seg * 4 + 3 : 12 + seg;
    int nzc_c = m->nzc[bc], nzc_n = nb->nzc[bn];
    if (t8) nzc_c = nz8 (m, ((bc >> 3) << 3) + ((bc & 3) >> 1) * 2);
    if (t8n) nzc_n = nz8 (nb, ((bn >> 3) << 3) + ((bn & 3) >> 1) * 2);
    if (nzc_c | nzc_n) return 2;
    int mc_ = bc, mn_ = bn;
    if (t8) mc_ = dir == 0 ? (seg >> 1) * 8 : (seg >> 1) * 2;
    if (t8n) mn_ = dir == 0 ? (seg >> 1) * 8 + 2 : 8 + (seg >> 1) * 2;
    return bs_mv (m, mc_, nb, mn_);
  }
  if (intra) return 3;
  if (m->mb_type == LH264_MB_SKIP) return 0;
  if (t8 && e != 2) return 0;
  const bool mv_too = m->mb_type != LH264_MB_P16x16;
  if (t8) {
    const int r = seg >> 1;
    int a, b;
    if (dir == 0) { a = r * 8; b = r * 8 + 2; } else { a = r * 2; b = 8 + r * 2; }
    if (nz8 (m, a) | nz8 (m, b)) return 2;
    return mv_too ? bs_mv (m, b, m, a) : 0;
  }
  const int a = dir == 0 ? seg * 4 + e - 1 : (e - 1) * 4 + seg;
  const int b = dir == 0 ? seg * 4 + e : e * 4 + seg;
  if (m->nzc[a] | m->nzc[b]) return 2;
  return mv_too ? bs_mv (m, b, m, a) : 0;
}

// filter one line across an edge.  p points at q0, xs = distance between samples across the edge.
__device__ __forceinline__ void filter_luma_line (uint8_t* p, int xs, int bs, int alpha, int beta, int tc0) {
  const int p0 = p[-xs], p1 = p[-2 * xs], p2 = p[-3 * xs], q0 = p[0], q1 = p[xs], q2 = p[2 * xs];
  const int d = abs (p0 - q0);
  if (!(d < alpha && abs (p1 - p0) < beta && abs (q1 - q0) < beta)) return;
  if (bs == 4) {                                 // DeblockLumaEq4_c
    if (d < ((alpha >> 2) + 2)) {
      if (abs (p2 - p0) < beta) {
        const int p3 = p[-4 * xs];
        p[-xs] = (uint8_t) ((p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3);
        p[-2 * xs] = (uint8_t) ((p2 + p1 + p0 + q0 + 2) >> 2);
        p[-3 * xs] = (uint8_t) ((2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3);
      } else p[-xs] = (uint8_t) ((2 * p1 + p0 + q1 + 2) >> 2);
      if (abs (q2 - q0) < beta) {
        const int q3 = p[3 * xs];
        p[0] = (uint8_t) ((p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3);
        p[xs] = (uint8_t) ((p0 + q0 + q1 + q2 + 2) >> 2);
        p[2 * xs] = (uint8_t) ((2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3);
      } else p[0] = (uint8_t) ((2 * q1 + q0 + p1 + 2) >> 2);
    } else {
      p[-xs] = (uint8_t) ((2 * p1 + p0 + q1 + 2) >> 2);
      p[0] = (uint8_t) ((2 * q1 + q0 + p1 + 2) >> 2);
    }
  } else {                                       // DeblockLumaLt4_c
    int t = tc0;
    if (abs (p2 - p0) < beta) { p[-2 * xs] = (uint8_t) (p1 + clip3 ((p2 + ((p0 + q0 + 1) >> 1) - (p1 << 1)) >> 1, -tc0, tc0)); t++; }
    if (abs (q2 - q0) < beta) { p[xs] = (uint8_t) (q1 + clip3 ((q2 + ((p0 + q0 + 1) >> 1) - (q1 << 1)) >> 1, -tc0, tc0)); t++; }
    const int dl = clip3 ((((q0 - p0) << 2) + (p1 - q1) + 4) >> 3, -t, t);
    p[-xs] = (uint8_t)clip_u8 (p0 + dl);
    p[0] = (uint8_t)clip_u8 (q0 - dl);
  }
}
__device__ __forceinline__ void filter_chroma_line (uint8_t* p, int xs, int bs, int alpha, int beta, int tc) {
  const int p0 = p[-xs], p1 = p[-2 * xs], q0 = p[0], q1 = p[xs];
  if (!(abs (p0 - q0) < alpha && abs (p1 - p0) < beta && abs (q1 - q0) < beta)) return;
  if (bs == 4) {
    p[-xs] = (uint8_t) ((2 * p1 + p0 + q1 + 2) >> 2);
    p[0] = (uint8_t) ((2 * q1 + q0 + p1 + 2) >> 2);
  } else {
    const int dl = clip3 ((((q0 - p0) << 2) + (p1 - q1) + 4) >> 3, -tc, tc);
    p[-xs] = (uint8_t)clip_u8 (p0 + dl);
    p[0] = (uint8_t)clip_u8 (q0 - dl);
  }
}

// ------------------------------------------------------------------------------------------------
// one macroblock: reconstruct (WelsTargetMbConstruction) + deblock (WelsDeblockingMb) + write back
// ------------------------------------------------------------------------------------------------
__device__ void process_mb (const FrameCtx& F, WaveLds& L, uint8_t* lineCur, const uint8_t* lineTop,
                            int LY, int LC, int mbx, int mby, int lane) {
  const int k = mby * F.mb_w + mbx;
  const lh264_mb_t* __restrict__ m = F.mbs + k;
  const int mb_type = __builtin_amdgcn_readfirstlane ((int)m->mb_type);
  const int cbp = __builtin_amdgcn_readfirstlane ((int)m->cbp);
  const int mflags = __builtin_amdgcn_readfirstlane ((int)m->flags);
  const lh264_slice_t* __restrict__ sl = F.slices + __builtin_amdgcn_readfirstlane ((int)m->slice_id);
  const bool t8 = mflags & LH264_MBF_T8x8;
  const bool covered = mb_type != 0;
  const bool intra = (mb_type & LH264_MB_INTRA) != 0;

  uint8_t* T = L.T;
  // luma lane layout: lane = 4*b + r, b = z-order 4x4 block, r = row inside the block
  const int b = lane >> 2, r = lane & 3;
  const int bx = (b & 1) | ((b >> 2) & 1) << 1, by = ((b >> 1) & 1) | ((b >> 3) & 1) << 1;
  const int ly = 4 * by + r, lx0 = 4 * bx;
  // chroma lane layout (lanes 0..31): plane p, block j, row r
  const int cp = (lane >> 4) & 1, cj = (lane >> 2) & 3;
  const int cy = 4 * (cj >> 1) + r, cx0 = 4 * (cj & 1);

  // ---- 1. unfiltered neighbours -> tile ------------------------------------------------------
  if (mby > 0) {
    if (lane < 8) * (uint32_t*)&T[tY (-1, -4 + 4 * lane)] = * (const uint32_t*)&lineTop[16 + 16 * mbx - 4 + 4 * lane];
    else if (lane < 16) {
      const int p = (lane - 8) >> 2, q = (lane - 8) & 3;
      * (uint32_t*)&L.C[p][tC (-1, -4 + 4 * q)] = * (const uint32_t*)&lineTop[LY + p * LC + 8 + 8 * mbx - 4 + 4 * q];
    }
  }
  if (mbx > 0) {
    if (lane >= 16 && lane < 32) T[tY (lane - 16, -1)] = L.leftY[lane - 16];
    else if (lane >= 32 && lane < 48) { const int p = (lane - 32) >> 3, q = (lane - 32) & 7; L.C[p][tC (q, -1)] = L.leftC[p][q]; }
  }
  wsync();

  if (covered) {
    // ---- 2. residual -------------------------------------------------------------------------
    int res[4] = {0, 0, 0, 0}, cres[4] = {0, 0, 0, 0};
    const bool i16 = mb_type == LH264_MB_I16x16;
    const bool have_res = (cbp != 0 || i16) && mb_type != LH264_MB_IPCM;
    const int16_t* __restrict__ cf = F.coeffs + (size_t)k * 384;
    if (have_res) {
      const int2 v = * (const int2*) (cf + 4 * lane);
      int c0 = sext16 (v.x), c1 = v.x >> 16, c2 = sext16 (v.y), c3 = v.y >> 16;
      int d0 = 0, d1 = 0, d2 = 0, d3 = 0;
      const bool have_c = (cbp >> 4) != 0;
      if (have_c && lane < 32) {
        const int2 cv = * (const int2*) (cf + 256 + 4 * lane);
        d0 = sext16 (cv.x); d1 = cv.x >> 16; d2 = sext16 (cv.y); d3 = cv.y >> 16;
      }
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
          const int qp = m->qp_y;
          const int wgt = sl->luma_dc_weight ? sl->luma_dc_weight : 16;
          const int dq = kNormAdjust0[qp % 6] << (qp / 6);
          const int qmul = wgt == 16 ? dq : ((wgt * dq) >> 4);
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
        * (int2*) (L.R + 4 * lane) = v;
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
        const int i8 = (ly >> 3) * 2 + (lx0 >> 3);
#pragma unroll
        for (int i = 0; i < 4; i++) res[i] = (32 + L.R[i8 * 64 + (ly & 7) * 8 + (lx0 & 7) + i]) >> 6;
        wsync();
      }
      if (have_c) idct4x4_quad (d0, d1, d2, d3, r, cres);
    }

    // ---- 3. prediction + residual ------------------------------------------------------------
    if (mb_type == LH264_MB_IPCM) {
      // samples travel in the coefficient slot, row-major (decode_slice.cpp:3213-3263 copies them at parse time)
      const int2 v = * (const int2*) (cf + 4 * lane);
      const int yy = lane >> 2, xx = 4 * (lane & 3);
      * (uint32_t*)&T[tY (yy, xx)] = (v.x & 0xff) | ((v.x >> 16) & 0xff) << 8 | (v.y & 0xff) << 16 | ((v.y >> 16) & 0xff) << 24;
      if (lane < 32) {
        const int2 cv = * (const int2*) (cf + 256 + 4 * lane);
        const int p = lane >> 4, yy2 = (lane >> 1) & 7, xx2 = 4 * (lane & 1);
        * (uint32_t*)&L.C[p][tC (yy2, xx2)] = (cv.x & 0xff) | ((cv.x >> 16) & 0xff) << 8 | (cv.y & 0xff) << 16 | ((cv.y >> 16) & 0xff) << 24;
      }
    } else if (intra) {
      // ---- luma
      if (i16) {                                                   // RecI16x16Mb rec_mb.cpp:179-230
        const int mode = __builtin_amdgcn_readfirstlane ((int)m->intra_mode[0]);
        int pr[4];
        if (mode == LH264_I16_V) {
          const uint32_t t = * (const uint32_t*)&T[tY (-1, lx0)];
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
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) o |= (uint32_t)clip_u8 (pr[i] + res[i]) << (8 * i);
        wsync();
        * (uint32_t*)&T[tY (ly, lx0)] = o;
      } else if (mb_type == LH264_MB_I8x8) {                       // RecI8x8Luma rec_mb.cpp:70-115
        * (int2*) (L.R + 4 * lane) = make_int2 ((res[0] & 0xffff) | (res[1] << 16), (res[2] & 0xffff) | (res[3] << 16));
        wsync();
        const int av = m->intra_avail;
        for (int i8 = 0; i8 < 4; i8++) {
          const int ox = (i8 & 1) * 8, oy = (i8 >> 1) * 8;
          const int mode = __builtin_amdgcn_readfirstlane ((int)m->intra_mode[((i8 >> 1) << 3) + ((i8 & 1) << 1)]);
          int tl = i8 == 0 ? !! (av & LH264_AVAIL_TL) : i8 == 1 ? !! (av & LH264_AVAIL_T) : i8 == 2 ? !! (av & LH264_AVAIL_L) : 1;
          const int tr = i8 == 0 ? !! (av & LH264_AVAIL_T) : i8 == 1 ? !! (av & LH264_AVAIL_TR) : i8 == 2 ? 1 : 0;
          const bool need_top = !(mode == LH264_I4_H || mode == LH264_I4_DC_L || mode == LH264_I4_DC_128 || mode == LH264_I4_HU);
          const bool need_left = !(mode == LH264_I4_V || mode == LH264_I4_DC_T || mode == LH264_I4_DC_128 || mode == LH264_I4_DDL ||
                                   mode == LH264_I4_DDL_TOP || mode == LH264_I4_VL || mode == LH264_I4_VL_TOP);
          const bool full_tr = (mode == LH264_I4_DDL || mode == LH264_I4_VL);
          const bool corner = (mode == LH264_I4_DDR || mode == LH264_I4_VR || mode == LH264_I4_HD);
          const bool top_var = (mode == LH264_I4_DDL_TOP || mode == LH264_I4_VL_TOP);
          if (corner) tl = 1;
          // filtered edge E[0..24] = L'7..L'0, TL', T'0..T'15   (8.3.2.2.1; get_intra_predictor.cpp:382-880)
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
            const uint8_t* E = L.E;
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
            // residual of pixel (ox+x, oy+y) in (block,row,col) layout
            const int X = ox + x, Y = oy + y;
            const int rr = L.R[zidx (X >> 2, Y >> 2) * 16 + (Y & 3) * 4 + (X & 3)];
            wsync();
            T[tY (Y, X)] = (uint8_t)clip_u8 (v + rr);
          }
          wsync();
        }
      } else {                                                      // RecI4x4Luma rec_mb.cpp:124-157
        * (int2*) (L.R + 4 * lane) = make_int2 ((res[0] & 0xffff) | (res[1] << 16), (res[2] & 0xffff) | (res[3] << 16));
        wsync();
        const uint2 ma = * (const uint2*)m->intra_mode, mb2 = * (const uint2*) (m->intra_mode + 8);   // 16 final modes, raster 4x4
        for (int blk = 0; blk < 16; blk++) {
          const int qx = (blk & 1) | ((blk >> 2) & 1) << 1, qy = ((blk >> 1) & 1) | ((blk >> 3) & 1) << 1;
          const int ri = qy * 4 + qx;
          const uint32_t mw = ri < 4 ? ma.x : ri < 8 ? ma.y : ri < 12 ? mb2.x : mb2.y;
          const int mode = __builtin_amdgcn_readfirstlane ((int) ((mw >> (8 * (ri & 3))) & 0xff));
          if (lane < 16) {
            const int x = lane & 3, y = lane >> 2;
            int v;
            switch (mode) {
            case LH264_I4_V: v = T[tY (4 * qy - 1, 4 * qx + x)]; break;
            case LH264_I4_H: v = T[tY (4 * qy + y, 4 * qx - 1)]; break;
            case LH264_I4_DC: case LH264_I4_DC_L: case LH264_I4_DC_T: {
              int s = 0;
              if (mode != LH264_I4_DC_L) {
                const uint32_t t = * (const uint32_t*)&T[tY (4 * qy - 1, 4 * qx)];
                s += (t & 0xff) + ((t >> 8) & 0xff) + ((t >> 16) & 0xff) + (t >> 24);
              }
              if (mode != LH264_I4_DC_T)
                s += (int)T[tY (4 * qy, 4 * qx - 1)] + (int)T[tY (4 * qy + 1, 4 * qx - 1)] + (int)T[tY (4 * qy + 2, 4 * qx - 1)] + (int)T[tY (4 * qy + 3, 4 * qx - 1)];
              v = (mode == LH264_I4_DC) ? (s + 4) >> 3 : (s + 2) >> 2;
              break;
            }
            case LH264_I4_DC_128: v = 128; break;
            default: v = pred4_dir (T, qx, qy, mode, x, y); break;
            }
            const int rr = L.R[blk * 16 + y * 4 + x];
            T[tY (4 * qy + y, 4 * qx + x)] = (uint8_t)clip_u8 (v + rr);
          }
          wsync();
        }
      }
      // ---- chroma (RecI4x4Chroma rec_mb.cpp:160-177): lanes 0..31
      {
        const int mode = __builtin_amdgcn_readfirstlane ((int)m->chroma_mode);
        const uint8_t* Cc = L.C[cp];
        int pr[4] = {128, 128, 128, 128};
        const int q = lane & 15;
        if (mode == LH264_C_V) {
          const uint32_t t = * (const uint32_t*)&Cc[tC (-1, cx0)];
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
        // RecChroma: residual only when cbp_c is 1 or 2 (cres is zero otherwise)
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) o |= (uint32_t)clip_u8 (pr[i] + cres[i]) << (8 * i);
        wsync();
        if (lane < 32) * (uint32_t*)&L.C[cp][tC (cy, cx0)] = o;
      }
    } else if (mb_type & LH264_MB_INTER) {
      // ---- inter prediction (GetInterPred rec_mb.cpp:344-545, BaseMC :247-274)
      if (lane < 16) {
        const int qx = lane & 3, qy = lane >> 2;      // raster 4x4 block
        int ox, oy, pw, ph;
        partition_of (mb_type, m->sub_type, qx, qy, ox, oy, pw, ph);
        const int pb = (oy >> 2) * 4 + (ox >> 2);     // raster index of the partition's first block (its MV)
        const int W = F.mb_w * 16, Hh = F.mb_h * 16;
        int fx = ((mbx * 16 + ox) << 2) + m->mv[pb][0];
        int fy = ((mby * 16 + oy) << 2) + m->mv[pb][1];
        fx = clip3 (fx, (-LH264_PAD_LUMA + 2) * 4, (W + LH264_PAD_LUMA - 19) * 4);
        fy = clip3 (fy, (-LH264_PAD_LUMA + 2) * 4, (Hh + LH264_PAD_LUMA - 19) * 4);
        int ridx = m->ref_idx[((qy >> 1) << 1) + (qx >> 1)];
        int slot = (ridx >= 0 && ridx < LH264_MAX_REFS) ? sl->ref_slot[ridx] : -1;
        if (slot < 0) slot = sl->ref_slot[0];
        if (slot < 0) slot = 0;
        // integer sample position of this 4x4 block's first pixel in the reference planes
        const int sx = (fx >> 2) + (qx * 4 - ox), syy = (fy >> 2) + (qy * 4 - oy);
        const int cxs = (fx >> 3) + ((qx * 4 - ox) >> 1), cys = (fy >> 3) + ((qy * 4 - oy) >> 1);
        L.mvi[lane][0] = syy * F.sy + sx;
        L.mvi[lane][1] = cys * F.sc + cxs;
        L.mvi[lane][2] = (fx & 3) | (fy & 3) << 2 | (fx & 7) << 4 | (fy & 7) << 8 | slot << 12 | (ridx < 0 ? 0 : ridx) << 16;
        // weighted prediction covers the partition (luma) and -- reference quirk, rec_mb.cpp:309-311 -- only the
        // top-left (w>>2)x(h>>2) samples of its chroma block
        L.mvi[lane][3] = ox | oy << 8 | pw << 16 | ph << 24;
      }
      wsync();
      int pr[4], cpr[4] = {0, 0, 0, 0};
      {
        const int rb = by * 4 + bx;                     // raster index of this lane's luma block
        const int info = L.mvi[rb][2];
        const lh264_pic_t* rp = &F.job->ref[(info >> 12) & 15];
        const uint8_t* src = rp->y_dev + L.mvi[rb][0] + r * F.sy;
        mc_luma_strip (src, F.sy, info & 3, (info >> 2) & 3, pr);
        if (sl->weighted_pred) {
          const int ri = (info >> 16) & 15, ld = sl->luma_log2_denom, wt = sl->luma_weight[ri], of = sl->luma_offset[ri];
#pragma unroll
          for (int i = 0; i < 4; i++) pr[i] = clip_u8 (ld >= 1 ? ((pr[i] * wt + (1 << (ld - 1))) >> ld) + of : pr[i] * wt + of);
        }
      }
      if (lane < 32) {
        // chroma strip: plane cp, row cy, cols cx0..cx0+3 ; two luma 4x4 blocks wide
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int cx = cx0 + i;
          const int rb = (cy >> 1) * 4 + (cx >> 1);
          const int info = L.mvi[rb][2];
          const lh264_pic_t* rp = &F.job->ref[(info >> 12) & 15];
          const uint8_t* plane = cp ? rp->v_dev : rp->u_dev;
          const uint8_t* s = plane + L.mvi[rb][1] + (cy & 1) * F.sc + (cx & 1);
          int v = mc_chroma_px (s, F.sc, (info >> 4) & 7, (info >> 8) & 7);
          if (sl->weighted_pred) {
            const int geo = L.mvi[rb][3];
            const int pox = (geo & 0xff) >> 1, poy = ((geo >> 8) & 0xff) >> 1, pw = (geo >> 16) & 0xff, ph = (geo >> 24) & 0xff;
            if (cx - pox < (pw >> 2) && cy - poy < (ph >> 2)) {
              const int ri = (info >> 16) & 15, ld = sl->chroma_log2_denom, wt = sl->chroma_weight[ri][cp], of = sl->chroma_offset[ri][cp];
              v = clip_u8 (ld >= 1 ? ((v * wt + (1 << (ld - 1))) >> ld) + of : v * wt + of);
            }
          }
          cpr[i] = v;
        }
      }
      uint32_t o = 0, oc = 0;
#pragma unroll
      for (int i = 0; i < 4; i++) { o |= (uint32_t)clip_u8 (pr[i] + res[i]) << (8 * i); oc |= (uint32_t)clip_u8 (cpr[i] + cres[i]) << (8 * i); }
      * (uint32_t*)&T[tY (ly, lx0)] = o;
      if (lane < 32) * (uint32_t*)&L.C[cp][tC (cy, cx0)] = oc;
    }
  }
  else {
    // macroblock not covered by any slice (lost data): pass the picture's current samples through
    * (uint32_t*)&T[tY (ly, lx0)] = * (const uint32_t*) (F.dy + (size_t) (mby * 16 + ly) * F.sy + mbx * 16 + lx0);
    if (lane < 32) * (uint32_t*)&L.C[cp][tC (cy, cx0)] = * (const uint32_t*) ((cp ? F.dv : F.du) + (size_t) (mby * 8 + cy) * F.sc + mbx * 8 + cx0);
  }
  wsync();

  // ---- 4. publish unfiltered bottom row / right column for later intra prediction ---------------
  if (lane < 4) * (uint32_t*)&lineCur[16 + 16 * mbx + 4 * lane] = * (const uint32_t*)&T[tY (15, 4 * lane)];
  else if (lane < 8) { const int p = (lane - 4) >> 1, q = (lane - 4) & 1; * (uint32_t*)&lineCur[LY + p * LC + 8 + 8 * mbx + 4 * q] = * (const uint32_t*)&L.C[p][tC (7, 4 * q)]; }
  else if (lane >= 16 && lane < 32) L.leftY[lane - 16] = T[tY (lane - 16, 15)];
  else if (lane >= 32 && lane < 48) { const int p = (lane - 32) >> 3, q = (lane - 32) & 7; L.leftC[p][q] = L.C[p][tC (q, 7)]; }
  wsync();

  // ---- 5. in-loop filter inside the tile ---------------------------------------------------------
  const bool filt = covered && !(F.flags & LH264_JOB_NO_DEBLOCK) && sl->deblock_idc != 1 && (sl->slice_type == 0 || sl->slice_type == 2);
  bool left_av = mbx > 0, top_av = mby > 0;
  if (filt && sl->deblock_idc == 2) {
    if (left_av) left_av = m[-1].slice_id == m->slice_id;
    if (top_av) top_av = m[-F.mb_w].slice_id == m->slice_id;
  }
  // neighbours' filtered samples: left 4 columns from the previous step, top 4 rows from HBM
  if (mbx > 0) {
    if (lane < 16) * (uint32_t*)&T[tY (lane, -4)] = L.lfY[lane];
    else if (lane < 32) { const int p = (lane - 16) >> 3, q = lane & 7; * (uint32_t*)&L.C[p][tC (q, -4)] = L.lfC[p][q]; }
  }
  if (mby > 0 && filt && top_av) {
    if (lane < 16) {
      const int rr = lane >> 2, q = lane & 3;
      * (uint32_t*)&T[tY (-4 + rr, 4 * q)] = * (const uint32_t*) (F.dy + (size_t) (mby * 16 - 4 + rr) * F.sy + mbx * 16 + 4 * q);
    } else if (lane < 24) {
      const int p = (lane - 16) >> 2, rr = (lane >> 1) & 1, q = lane & 1;
      const uint8_t* pl = p ? F.dv : F.du;
      * (uint32_t*)&L.C[p][tC (-2 + rr, 4 * q)] = * (const uint32_t*) (pl + (size_t) (mby * 8 - 2 + rr) * F.sc + mbx * 8 + 4 * q);
    }
  }
  wsync();
  if (filt) {
    const bool mintra = (mb_type == LH264_MB_I4x4 || mb_type == LH264_MB_I8x8 || mb_type == LH264_MB_I16x16 || mb_type == LH264_MB_IPCM);
    if (lane < 32) {
      const int dir = lane >> 4, e = (lane >> 2) & 3, seg = lane & 3;
      const lh264_mb_t* nb = dir == 0 ? (left_av ? m - 1 : nullptr) : (top_av ? m - F.mb_w : nullptr);
      L.bs[lane] = (uint8_t)compute_bs (m, nb, dir, e, seg, mintra);
    }
    wsync();
    const int qp = m->qp_y, qpc0 = m->qp_c[0], qpc1 = m->qp_c[1];
    const int ao = sl->alpha_c0_offset, bo = sl->beta_offset;
    for (int dir = 0; dir < 2; dir++) {
      const bool have_nb = dir == 0 ? left_av : top_av;
      const lh264_mb_t* nb = dir == 0 ? m - 1 : m - F.mb_w;
      for (int e = 0; e < 4; e++) {
        if (e == 0 && !have_nb) continue;
        if ((e & 1) && t8) continue;
        const uint32_t bs4 = * (const uint32_t*)&L.bs[dir * 16 + e * 4];
        if (__builtin_amdgcn_readfirstlane ((int)bs4) == 0) continue;
        int q = qp, qc0 = qpc0, qc1 = qpc1;
        if (e == 0) { q = (qp + nb->qp_y + 1) >> 1; qc0 = (qpc0 + nb->qp_c[0] + 1) >> 1; qc1 = (qpc1 + nb->qp_c[1] + 1) >> 1; }
        if (lane < 16) {
          const int bs = (bs4 >> (8 * (lane >> 2))) & 0xff;
          const int ia = q + ao;
          const int alpha = kAlpha[tab_idx (ia)], beta = kBeta[tab_idx (q + bo)];
          if (bs && (alpha | beta)) {
            uint8_t* p = dir == 0 ? &T[tY (lane, 4 * e)] : &T[tY (4 * e, lane)];
            filter_luma_line (p, dir == 0 ? 1 : 32, bs, alpha, beta, kTc0[tab_idx (ia)][bs & 3]);
          }
        } else if (lane < 32 && !(e & 1)) {
          const int p_ = (lane - 16) >> 3, i = lane & 7;
          const int bs = (bs4 >> (8 * (i >> 1))) & 0xff;
          const int qq = p_ ? qc1 : qc0;
          const int ia = qq + ao;
          const int alpha = kAlpha[tab_idx (ia)], beta = kBeta[tab_idx (qq + bo)];
          if (bs && (alpha | beta)) {
            uint8_t* p = dir == 0 ? &L.C[p_][tC (i, 2 * e)] : &L.C[p_][tC (2 * e, i)];
            filter_chroma_line (p, dir == 0 ? 1 : 16, bs, alpha, beta, kTc0[tab_idx (ia)][bs & 3] + 1);
          }
        }
        wsync();
      }
    }
  }
  // keep the (filtered) right 4 columns for the next macroblock of this row
  if (lane < 16) L.lfY[lane] = * (const uint32_t*)&T[tY (lane, 12)];
  else if (lane < 32) { const int p = (lane - 16) >> 3, q = lane & 7; L.lfC[p][q] = * (const uint32_t*)&L.C[p][tC (q, 4)]; }

  // ---- 6. write back: rows 0..15 x cols -4..15 (+ top rows -3..-1 x cols 0..15 when filtered) ------
  {
    // luma: 16 rows x 5 dwords = 80 dwords; chroma: 2 x 8 rows x 3 dwords = 48
    for (int i = lane; i < 80; i += 64) {
      const int rr = i / 5, q = i % 5;
      if (q == 0 && mbx == 0) continue;
      * (uint32_t*) (F.dy + (size_t) (mby * 16 + rr) * F.sy + mbx * 16 - 4 + 4 * q) = * (const uint32_t*)&T[tY (rr, -4 + 4 * q)];
    }
    if (lane < 48) {
      const int p = lane / 24, i = lane % 24, rr = i / 3, q = i % 3;
      if (!(q == 0 && mbx == 0)) {
        uint8_t* pl = p ? F.dv : F.du;
        * (uint32_t*) (pl + (size_t) (mby * 8 + rr) * F.sc + mbx * 8 - 4 + 4 * q) = * (const uint32_t*)&L.C[p][tC (rr, -4 + 4 * q)];
      }
    }
    if (mby > 0 && filt && top_av) {
      if (lane < 12) {
        const int rr = lane >> 2, q = lane & 3;
        * (uint32_t*) (F.dy + (size_t) (mby * 16 - 3 + rr) * F.sy + mbx * 16 + 4 * q) = * (const uint32_t*)&T[tY (-3 + rr, 4 * q)];
      } else if (lane >= 16 && lane < 20) {
        const int p = (lane - 16) >> 1, q = lane & 1;
        uint8_t* pl = p ? F.dv : F.du;
        * (uint32_t*) (pl + (size_t) (mby * 8 - 1) * F.sc + mbx * 8 + 4 * q) = * (const uint32_t*)&L.C[p][tC (-1, 4 * q)];
      }
    }
  }
  wsync();
}

// ------------------------------------------------------------------------------------------------
// ExpandReferencingPicture (expand_pic.cpp:145-174): every padding sample = nearest picture sample
// ------------------------------------------------------------------------------------------------
__device__ void expand_plane (uint8_t* p, int stride, int w, int h, int pad, int tid, int nthreads) {
  // work in dwords: padded rows are (w + 2*pad) wide, pad and w are multiples of 4
  const int wd = (w + 2 * pad) >> 2, pd = pad >> 2;
  const int total_rows = h + 2 * pad;
  // (a) left/right borders of picture rows + (b) full top/bottom rows
  for (int i = tid; i < total_rows * wd; i += nthreads) {
    const int rr = i / wd - pad, cd = i % wd - pd;           // row, dword column relative to pixel (0,0)
    const bool inside_row = rr >= 0 && rr < h;
    if (inside_row && cd >= 0 && cd < (w >> 2)) continue;    // picture interior
    const int sr = clip3 (rr, 0, h - 1);
    uint32_t v;
    if (cd < 0) v = 0x01010101u * p[(size_t)sr * stride];
    else if (cd >= (w >> 2)) v = 0x01010101u * p[(size_t)sr * stride + w - 1];
    else v = * (const uint32_t*) (p + (size_t)sr * stride + 4 * cd);
    * (uint32_t*) (p + (ptrdiff_t)rr * stride + 4 * cd) = v;
  }
}

__global__ void __launch_bounds__ (1024)
recon_chain_kernel (const lh264_frame_job_t* __restrict__ jobs, const int32_t* __restrict__ chain_first, int n_chains,
                    int line_bytes) {
  extern __shared__ __attribute__ ((aligned (16))) uint8_t smem[];
  const int NW = blockDim.x >> 6;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane (tid >> 6);
  const int lane = tid & 63;
  volatile int* progress = (volatile int*)smem;                           // [16]
  WaveLds* wl = (WaveLds*) (smem + 64);
  uint8_t* lines = smem + 64 + sizeof (WaveLds) * NW;                     // [NW + 1][line_bytes]
  WaveLds& L = wl[wave];
  const int NL = NW + 1;

  const int chain = blockIdx.x;
  if (chain >= n_chains) return;
  const int first = chain_first[chain], last = chain_first[chain + 1];
  for (int ji = first; ji < last; ji++) {
    const lh264_frame_job_t* J = jobs + ji;
    FrameCtx F;
    F.job = J; F.mbs = J->mbs_dev; F.coeffs = J->coeffs_dev; F.slices = J->slices_dev;
    F.dy = J->dst.y_dev; F.du = J->dst.u_dev; F.dv = J->dst.v_dev;
    F.mb_w = J->mb_w; F.mb_h = J->mb_h; F.sy = J->stride_y; F.sc = J->stride_c; F.flags = J->flags;
    const int LY = F.mb_w * 16 + 48, LC = F.mb_w * 8 + 24;
    if (tid < 16) progress[tid] = 0;
    __syncthreads();
    int jrow = 0;
    for (int row = wave; row < F.mb_h; row += NW, jrow++) {
      uint8_t* lineCur = lines + (size_t) (row % NL) * line_bytes;
      const uint8_t* lineTop = lines + (size_t) ((row + NL - 1) % NL) * line_bytes;
      const int wprev = (wave + NW - 1) % NW;
      const int jprev = (row - 1) / NW;
      for (int x = 0; x < F.mb_w; x++) {
        if (row > 0) {
          const int need = jprev * F.mb_w + min (x + 2, F.mb_w);
          while (progress[wprev] < need) __builtin_amdgcn_s_sleep (1);
          __builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "workgroup");
        }
        process_mb (F, L, lineCur, lineTop, LY, LC, x, row, lane);
        __builtin_amdgcn_fence (__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) progress[wave] = jrow * F.mb_w + x + 1;
      }
    }
    __syncthreads();
    __builtin_amdgcn_fence (__ATOMIC_ACQ_REL, "workgroup");
    if (!(F.flags & LH264_JOB_NO_EXPAND)) {
      expand_plane (F.dy, F.sy, F.mb_w * 16, F.mb_h * 16, LH264_PAD_LUMA, tid, blockDim.x);
      expand_plane (F.du, F.sc, F.mb_w * 8, F.mb_h * 8, LH264_PAD_CHROMA, tid, blockDim.x);
      expand_plane (F.dv, F.sc, F.mb_w * 8, F.mb_h * 8, LH264_PAD_CHROMA, tid, blockDim.x);
    }
    __builtin_amdgcn_fence (__ATOMIC_ACQ_REL, "workgroup");
    __syncthreads();
  }
}

size_t wave_lds_bytes() { return sizeof (WaveLds); }

}  // namespace lh264
