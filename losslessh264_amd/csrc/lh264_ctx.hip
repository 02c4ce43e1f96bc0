// lh264_ctx.hip - per-coefficient context-model index kernels.
//
// Pass 1a ctx_nnz_kernel       : one wave per coded macroblock, all frames of all streams at once: reads the 768 B of
//         levels coalesced (lane = 4 coefficients of one 4x4 row), counts nonzeros per 4x4 block across the lane quad
//         and writes the 24-byte entry of the "nnz image".
// Pass 1b ctx_inherit_chain_kernel : one workgroup per stream, frames in order: a skipped macroblock inherits the
//         PAST entry (FreqImage semantics, decode_slice.cpp:3104-3108) - 24 bytes per skipped macroblock.
// Pass 2  ctx_symbols_kernel   : one wave per macroblock, all frames of all streams at once (context INDICES do not
//         depend on the adaptive state).  Levels are staged in LDS; one lane per 4x4 block (or per 8x8 block) pulls its
//         levels into registers in scan order (static zig-zag offsets), finds the last nonzero one with a bit mask,
//         and walks encode4x4's loop (decode_slice.cpp:2059-2094) producing (prior index, value) pairs straight
//         into the macroblock's emission-order buffer in LDS, which the wave then copies out coalesced.
// HBM-bound: 768 B + 128 B + 72 B read, ~8 B per coded symbol written.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/lh264.h"

namespace lh264 {

#define LDS __attribute__ ((address_space (3)))
#define GLB __attribute__ ((address_space (1)))
typedef int v2i __attribute__ ((ext_vector_type (2)));
template <typename T> __device__ __forceinline__ GLB T* as_glb (const void* p) { return (GLB T*) (uintptr_t)p; }

__device__ __forceinline__ int mb_type_code (int t) {       // MacroblockModel::encodeMacroblockType, macroblock_model.cpp:647-679
  switch (t) {
  case LH264_MB_I4x4: return 0;  case LH264_MB_I16x16: return 1;  case LH264_MB_I8x8: return 2;
  case LH264_MB_P16x16: return 3;  case LH264_MB_P16x8: return 4;  case LH264_MB_P8x16: return 5;
  case LH264_MB_P8x8: return 6;  case LH264_MB_P8x8REF0: return 7;  case LH264_MB_IPCM: return 8;
  case 0x400: return 9;  case 0x4000: return 10;
  default: return 11;
  }
}


// ---- pass 1a: nonzero counts of every coded macroblock, all frames of all streams at once -----------------------------
__global__ void __launch_bounds__ (256)
ctx_nnz_kernel (const lh264_ctx_job_t* __restrict__ jobs, int n_jobs, int blocks_per_job) {
  const int ji = blockIdx.x / blocks_per_job;
  if (ji >= n_jobs) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const lh264_ctx_job_t* J = jobs + ji;
  const int n = J->mb_w * J->mb_h;
  const int k = (blockIdx.x % blocks_per_job) * 4 + wave;
  if (k >= n) return;
  // the macroblock type and the levels are requested together (one memory round trip, not two)
  const GLB int16_t* lv = as_glb<const int16_t> (J->levels_dev) + (size_t)k * 384;
  GLB uint8_t* cur = as_glb<uint8_t> (J->nnz_cur_dev) + (size_t)k * 24;
  const int type_l = as_glb<const lh264_mb_t> (J->mbs_dev)[k].mb_type;
  const v2i a = * (const GLB v2i*) (lv + 4 * lane);
  v2i b = {0, 0};
  if (lane < 32) b = * (const GLB v2i*) (lv + 256 + 4 * lane);
  asm volatile ("" : : "v"(a.x), "v"(a.y), "v"(b.x), "v"(b.y));      // keeps the loads above the branch
  const int type = __builtin_amdgcn_readfirstlane (type_l);
  if (type == LH264_MB_SKIP || type == 0) return;       // inherited from PAST by pass 1b
  int c = ((a.x & 0xffff) != 0) + ((a.x >> 16) != 0) + ((a.y & 0xffff) != 0) + ((a.y >> 16) != 0);
  c += __shfl_xor (c, 1); c += __shfl_xor (c, 2);       // luma block = lane >> 2
  int d = ((b.x & 0xffff) != 0) + ((b.x >> 16) != 0) + ((b.y & 0xffff) != 0) + ((b.y >> 16) != 0);
  d += __shfl_xor (d, 1); d += __shfl_xor (d, 2);
  // gather the 24 counts into 6 dwords: lane 16*j + 4*i holds count 4*j + i
  const int c0 = __shfl (c, (lane & 3) * 16), c1 = __shfl (c, (lane & 3) * 16 + 4), c2 = __shfl (c, (lane & 3) * 16 + 8), c3 = __shfl (c, (lane & 3) * 16 + 12);
  const int d0 = __shfl (d, (lane & 1) * 16), d1 = __shfl (d, (lane & 1) * 16 + 4), d2 = __shfl (d, (lane & 1) * 16 + 8), d3 = __shfl (d, (lane & 1) * 16 + 12);
  if (lane < 4) ((GLB uint32_t*)cur)[lane] = (uint32_t)c0 | (uint32_t)c1 << 8 | (uint32_t)c2 << 16 | (uint32_t)c3 << 24;
  else if (lane < 6) ((GLB uint32_t*)cur)[lane] = (uint32_t)d0 | (uint32_t)d1 << 8 | (uint32_t)d2 << 16 | (uint32_t)d3 << 24;
}

// ---- pass 1b: a skipped macroblock inherits the PAST entry (FreqImage semantics, decode_slice.cpp:3104-3108).  The only
// step that is sequential over the frames of a stream; it moves 24 bytes per skipped macroblock. -------------------------
__global__ void __launch_bounds__ (256)
ctx_inherit_chain_kernel (const lh264_ctx_job_t* __restrict__ jobs, const int32_t* __restrict__ chain_first, int n_chains) {
  const int chain = blockIdx.x;
  if (chain >= n_chains) return;
  for (int ji = chain_first[chain]; ji < chain_first[chain + 1]; ji++) {
    const lh264_ctx_job_t* J = jobs + ji;
    const GLB lh264_mb_t* mbs = as_glb<const lh264_mb_t> (J->mbs_dev);
    const GLB uint8_t* past = as_glb<const uint8_t> (J->nnz_past_dev);
    GLB uint8_t* cur = as_glb<uint8_t> (J->nnz_cur_dev);
    const int n = J->mb_w * J->mb_h;
    for (int k = threadIdx.x; k < n; k += blockDim.x) {
      const int type = mbs[k].mb_type;
      if (type == LH264_MB_SKIP || type == 0) {
        GLB uint32_t* d = (GLB uint32_t*) (cur + (size_t)k * 24);
        if (J->nnz_past_dev) {
          const GLB uint32_t* sp = (const GLB uint32_t*) (past + (size_t)k * 24);
          const uint32_t v0 = sp[0], v1 = sp[1], v2 = sp[2], v3 = sp[3], v4 = sp[4], v5 = sp[5];
          d[0] = v0; d[1] = v1; d[2] = v2; d[3] = v3; d[4] = v4; d[5] = v5;
        } else { d[0] = d[1] = d[2] = d[3] = d[4] = d[5] = 0u; }
      }
    }
    __builtin_amdgcn_fence (__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "workgroup");
  }
}

// ---- pass 2 ---------------------------------------------------------------------------------------------------
struct CtxWave {
  int16_t  lv[384];
  uint8_t  nz[4][24];          // cur, left, above, past
  uint64_t osym[LH264_CTX_MAX_SYMS];   // the macroblock's symbols in emission order, copied out coalesced
};

constexpr uint8_t cZz16[16] = {0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15};
constexpr uint8_t cZz64[64] = {
  0, 1, 5, 6, 14, 15, 27, 28, 2, 4, 7, 13, 16, 26, 29, 42, 3, 8, 12, 17, 25, 30, 41, 43, 9, 11, 18, 24, 31, 40, 44, 53,
  10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60, 21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63
};

__device__ __forceinline__ uint64_t mk_sym (uint32_t prior, int value, int kind) {
  // {u32 prior, i16 value, u8 kind, u8 pad}
  return (uint64_t)prior | ((uint64_t) (uint16_t)value << 32) | ((uint64_t) (uint8_t)kind << 48);
}
__device__ __forceinline__ int min2 (int v) { return v < 2 ? v : 2; }
__device__ __forceinline__ int clamp04 (int v) { return v < 0 ? 0 : (v > 4 ? 4 : v); }

// 16 levels of a block in scan order (chunk CH of the 8x8 scan when NCO == 64); static LDS offsets
template <int NCO, int CH>
__device__ __forceinline__ void load_scan16 (const LDS int16_t* ac, int c[16]) {
#pragma unroll
  for (int i = 0; i < 16; i++) c[i] = ac[NCO == 16 ? cZz16[i] : cZz64[CH * 16 + i]];
}

struct WalkState { int left_nz, prev, prev2, emitted; };

// encode4x4's coefficient loop (decode_slice.cpp:2059-2094) over scan positions [16*CH, 16*CH+16): every coefficient
// up to the last nonzero one is a symbol whose prior depends on the two previous levels and the nonzeros left
// wmax: the largest `last` of the wave (uniform): positions beyond it are skipped four at a time without being looked at
template <int CH>
__device__ __forceinline__ void walk16 (const int c[16], int start, int last, int wmax, uint32_t outer0, int kind, WalkState& w, LDS uint64_t* dst) {
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const int pos = CH * 16 + i;
    if ((i & 3) == 0 && wmax < pos) return;
    if (pos >= start && pos <= last) {
      const uint32_t inner = (uint32_t) ((((min (4, w.left_nz) * 5 + clamp04 (w.prev + 2)) * 5 + clamp04 (w.prev2 + 2)) * 5 + 2) * 5 + 2);
      dst[w.emitted] = mk_sym ((outer0 + w.emitted) * 3125u + inner, c[i], kind);
      w.prev2 = w.prev; w.prev = c[i]; w.emitted++;
      if (c[i]) w.left_nz--;
    }
  }
}

__device__ __forceinline__ uint32_t nzmask16 (const int c[16]) {
  uint32_t m = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) m |= (uint32_t) (c[i] != 0) << i;
  return m;
}

__global__ void __launch_bounds__ (256)
ctx_symbols_kernel (const lh264_ctx_job_t* __restrict__ jobs, int n_jobs, int blocks_per_job) {
  __shared__ CtxWave sm[4];
  const int ji = blockIdx.x / blocks_per_job;
  if (ji >= n_jobs) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const lh264_ctx_job_t* J = jobs + ji;
  const int mb_w = J->mb_w, n = mb_w * J->mb_h;
  const int k = (blockIdx.x % blocks_per_job) * 4 + wave;
  if (k >= n) return;
  LDS CtxWave& W = * (LDS CtxWave*) (uintptr_t) (uint32_t) (uintptr_t)&sm[wave];
  const GLB lh264_mb_t* m = as_glb<const lh264_mb_t> (J->mbs_dev) + k;
  const GLB int16_t* lv = as_glb<const int16_t> (J->levels_dev) + (size_t)k * 384;
  const GLB uint8_t* cur = as_glb<const uint8_t> (J->nnz_cur_dev);
  GLB uint64_t* out = (GLB uint64_t*) (as_glb<lh264_ctx_sym_t> (J->syms_dev) + (size_t)k * LH264_CTX_MAX_SYMS);
  GLB uint16_t* nout = as_glb<uint16_t> (J->n_syms_dev) + k;

  // Every global read of the macroblock is issued before anything is looked at: the record (lanes 0..7 take a dword each),
  // the levels and the four nnz entries travel together, one memory round trip instead of a chain of three (a wave has
  // nothing else to hide them behind; for a skipped macroblock the levels are read for nothing)
  const int mbx = k % mb_w;
  const uint32_t rec = lane < 8 ? ((const GLB uint32_t*)m)[lane] : 0u;
  const v2i l0 = * (const GLB v2i*) (lv + 4 * lane);
  v2i l1 = {0, 0};
  if (lane < 32) l1 = * (const GLB v2i*) (lv + 256 + 4 * lane);
  const int who = lane / 6, q = lane % 6;
  uint32_t nzv = 0;
  if (lane < 24) {
    if (who == 0) nzv = ((const GLB uint32_t*) (cur + (size_t)k * 24))[q];
    else if (who == 1) { if (mbx > 0) nzv = ((const GLB uint32_t*) (cur + (size_t) (k - 1) * 24))[q]; }
    else if (who == 2) { if (k >= mb_w) nzv = ((const GLB uint32_t*) (cur + (size_t) (k - mb_w) * 24))[q]; }
    else if (J->nnz_past_dev) nzv = ((const GLB uint32_t*) (as_glb<const uint8_t> (J->nnz_past_dev) + (size_t)k * 24))[q];
  }
  // stage levels and the four nnz entries
  * (LDS v2i*) (W.lv + 4 * lane) = l0;
  if (lane < 32) * (LDS v2i*) (W.lv + 256 + 4 * lane) = l1;
  if (lane < 24) ((LDS uint32_t*)W.nz[who])[q] = nzv;
  const uint32_t head = (uint32_t)__builtin_amdgcn_readlane ((int)rec, 0);      // mb_type | cbp << 16 | qp << 24
  const int type = head & 0xffff;
  if (type == LH264_MB_SKIP || type == LH264_MB_IPCM || type == 0) {     // no coefficient symbols (writeBlock false / PCM)
    if (lane == 0) *nout = 0;
    return;
  }
  const int cbp = (head >> 16) & 0xff;
  const int t8 = (__builtin_amdgcn_readlane ((int)rec, 1) >> 16) & LH264_MBF_T8x8;                 // flags: byte 6
  const int sid = (int) ((uint32_t)__builtin_amdgcn_readlane ((int)rec, 6) >> 16);                // slice_id: bytes 26..27
  const int st = as_glb<const lh264_slice_t> (J->slices_dev)[sid].slice_type;
  const int mbc = mb_type_code (type);
  asm volatile ("" ::: "memory");
  __builtin_amdgcn_wave_barrier();

  const int cbpl = cbp & 15, cbpc = cbp >> 4;
  const bool i16 = type == LH264_MB_I16x16;
  const bool cdc = cbpc == 1 || cbpc == 2;
  // one lane per block: 4x4 blocks -> lanes 0..23 ; with the 8x8 transform luma uses lanes 0,4,8,12
  const int b = lane;
  const bool luma = b < 16;
  bool coded = lane < 24 && (luma ? ((cbpl >> (b >> 2)) & 1) : (cbpc == 2));
  const bool big = luma && t8;
  if (big) coded = coded && ((b & 3) == 0);
  const bool emit_dc = luma ? !i16 : !cdc;
  const int start = emit_dc ? 0 : 1;
  LDS const int16_t* ac = W.lv + (lane < 24 ? b * 16 : 0);
  int c[16];
  int cnt = 0, nonzeros = 0, last = -1;
  if (coded) {
    if (!big) {
      load_scan16<16, 0> (ac, c);
      const uint32_t mk = nzmask16 (c) & (emit_dc ? 0xffffu : 0xfffeu);
      nonzeros = __popc (mk);
      last = 31 - __clz ((int)mk);                     // -1 when mk == 0
    } else {
      uint64_t mk = 0;
      load_scan16<64, 0> (ac, c); mk |= (uint64_t)nzmask16 (c);
      load_scan16<64, 1> (ac, c); mk |= (uint64_t)nzmask16 (c) << 16;
      load_scan16<64, 2> (ac, c); mk |= (uint64_t)nzmask16 (c) << 32;
      load_scan16<64, 3> (ac, c); mk |= (uint64_t)nzmask16 (c) << 48;
      if (!emit_dc) mk &= ~1ull;
      nonzeros = __popcll (mk);
      last = 63 - __clzll ((long long)mk);
    }
    cnt = 1 + (nonzeros ? last - start + 1 : 0);
  }
  // emission order: luma DC run, chroma DC run, blocks 0..23
  const int ndc_l = i16 ? 16 : 0, ndc_c = cdc ? 8 : 0;
  int incl = cnt;                                    // inclusive scan over lanes 0..23
  for (int d = 1; d < 32; d <<= 1) { const int t = __shfl_up (incl, d); if (lane >= d) incl += t; }
  const int total = __shfl (incl, 23) + ndc_l + ndc_c;
  const int my_off = ndc_l + ndc_c + incl - cnt;
  int wmax = coded ? last : -1;                     // largest scan position any block of the macroblock reaches
  for (int d = 32; d; d >>= 1) wmax = max (wmax, __shfl_xor (wmax, d));
  wmax = __builtin_amdgcn_readfirstlane (wmax);
  if (coded) {
    const int color = luma ? 0 : (b < 20 ? 1 : 2);
    LDS const uint8_t* C = W.nz[0], *Lf = W.nz[1], *Ab = W.nz[2], *Pa = W.nz[3];
    int past, left, above;
    if (big) {
      const int s = b >> 2;
      auto c8 = [] (LDS const uint8_t* p, int i) { return p[i] + p[i + 1] + p[i + 2] + p[i + 3]; };
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
    LDS uint64_t* dst = W.osym + my_off;
    const int nco = big ? 64 : 16;
    dst[0] = mk_sym ((uint32_t) ((((((st * 16 + mbc) * 3 + color) * 3 + min2 (past)) * 3 + min2 (left)) * 3) + min2 (above)), nonzeros,
                     big ? LH264_SYM_NZ8 : LH264_SYM_NZ4);
    const uint32_t outer0 = (uint32_t) (((st * 16 + mbc) * 3 + color) * nco);
    WalkState w = {nonzeros, 0, 0, 0};
    if (!big) walk16<0> (c, start, last, wmax, outer0, LH264_SYM_AC4, w, dst + 1);
    else {
      load_scan16<64, 0> (ac, c); walk16<0> (c, start, last, wmax, outer0, LH264_SYM_AC8, w, dst + 1);
      load_scan16<64, 1> (ac, c); walk16<1> (c, start, last, wmax, outer0, LH264_SYM_AC8, w, dst + 1);
      load_scan16<64, 2> (ac, c); walk16<2> (c, start, last, wmax, outer0, LH264_SYM_AC8, w, dst + 1);
      load_scan16<64, 3> (ac, c); walk16<3> (c, start, last, wmax, outer0, LH264_SYM_AC8, w, dst + 1);
    }
  }
  if (lane >= 32 && lane < 48) {
    if (i16) {                                       // getLumaDCIntPrior: lumaDCIntPriors[i][slice][mbtype]
      const int i = lane - 32;
      W.osym[i] = mk_sym ((uint32_t) ((i * 5 + st) * 16 + mbc), W.lv[i * 16], LH264_SYM_LUMA_DC);
    }
  } else if (lane >= 48 && lane < 56) {
    if (cdc) {
      const int i = lane - 48;
      W.osym[ndc_l + i] = mk_sym ((uint32_t) ((i * 5 + st) * 16 + mbc), W.lv[256 + i * 16], LH264_SYM_CHROMA_DC);
    }
  }
  asm volatile ("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  for (int i = lane; i < total; i += 64) out[i] = W.osym[i];
  if (lane == 0) *nout = (uint16_t)total;
}

}  // namespace lh264
