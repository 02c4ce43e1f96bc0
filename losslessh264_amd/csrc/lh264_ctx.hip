// lh264_ctx.hip - per-coefficient context-model index kernels.
//
// Pass 1  ctx_nnz_chain_kernel : one workgroup per stream, frames in order.  Per macroblock one wave reads the 768 B
//         of levels coalesced (lane = 4 coefficients of one 4x4 row, as in the reconstruct kernel), counts nonzeros
//         per 4x4 block across the lane quad and writes the 24-byte entry of the "nnz image"; a skipped macroblock
//         inherits the PAST entry (FreqImage semantics, decode_slice.cpp:3104-3108).
// Pass 2  ctx_symbols_kernel   : one wave per macroblock, all frames of all streams at once (context INDICES do not
//         depend on the adaptive state).  Levels are staged in LDS; one lane per 4x4 block (or per 8x8 block) walks
//         the zig-zag scan (encode4x4, decode_slice.cpp:2059-2094) producing (prior index, value) pairs, which are
//         compacted into emission order and written out.
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

__constant__ uint8_t kZz16[16] = {0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15};
__constant__ uint8_t kZz64[64] = {
  0, 1, 5, 6, 14, 15, 27, 28, 2, 4, 7, 13, 16, 26, 29, 42, 3, 8, 12, 17, 25, 30, 41, 43, 9, 11, 18, 24, 31, 40, 44, 53,
  10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60, 21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63
};

// ---- pass 1 ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__ (256)
ctx_nnz_chain_kernel (const lh264_ctx_job_t* __restrict__ jobs, const int32_t* __restrict__ chain_first, int n_chains) {
  const int chain = blockIdx.x;
  if (chain >= n_chains) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  for (int ji = chain_first[chain]; ji < chain_first[chain + 1]; ji++) {
    const lh264_ctx_job_t* J = jobs + ji;
    const GLB lh264_mb_t* mbs = as_glb<const lh264_mb_t> (J->mbs_dev);
    const GLB int16_t* levels = as_glb<const int16_t> (J->levels_dev);
    const GLB uint8_t* past = as_glb<const uint8_t> (J->nnz_past_dev);
    GLB uint8_t* cur = as_glb<uint8_t> (J->nnz_cur_dev);
    const int n = J->mb_w * J->mb_h;
    for (int k = wave; k < n; k += nw) {
      const int type = __builtin_amdgcn_readfirstlane ((int)mbs[k].mb_type);
      if (type == LH264_MB_SKIP || type == 0) {           // rtd.isSkipped: the FreqImage entry is inherited from PAST
        if (lane < 6) ((GLB uint32_t*) (cur + (size_t)k * 24))[lane] = J->nnz_past_dev ? ((const GLB uint32_t*) (past + (size_t)k * 24))[lane] : 0u;
        continue;
      }
      const GLB int16_t* lv = levels + (size_t)k * 384;
      const v2i a = * (const GLB v2i*) (lv + 4 * lane);
      int c = ((a.x & 0xffff) != 0) + ((a.x >> 16) != 0) + ((a.y & 0xffff) != 0) + ((a.y >> 16) != 0);
      c += __shfl_xor (c, 1); c += __shfl_xor (c, 2);     // luma block = lane >> 2
      int d = 0;
      if (lane < 32) {
        const v2i b = * (const GLB v2i*) (lv + 256 + 4 * lane);
        d = ((b.x & 0xffff) != 0) + ((b.x >> 16) != 0) + ((b.y & 0xffff) != 0) + ((b.y >> 16) != 0);
      }
      d += __shfl_xor (d, 1); d += __shfl_xor (d, 2);
      if ((lane & 3) == 0) {
        cur[(size_t)k * 24 + (lane >> 2)] = (uint8_t)c;
        if (lane < 32) cur[(size_t)k * 24 + 16 + (lane >> 2)] = (uint8_t)d;
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
  uint32_t sp[24 * 17];        // staged prior per block (8x8 block i uses the 68 slots of blocks 4i..4i+3)
  int16_t  sv[24 * 17];
  int32_t  off[26];            // output offset of: block 0..23, luma DC run (24), chroma DC run (25)
};

__device__ __forceinline__ void put_sym (GLB lh264_ctx_sym_t* dst, uint32_t prior, int value, int kind) {
  // {u32 prior, i16 value, u8 kind, u8 pad} as one 8-byte store
  * (GLB uint64_t*)dst = (uint64_t)prior | ((uint64_t) (uint16_t)value << 32) | ((uint64_t) (uint8_t)kind << 48);
}
__device__ __forceinline__ int min2 (int v) { return v < 2 ? v : 2; }
__device__ __forceinline__ int clamp04 (int v) { return v < 0 ? 0 : (v > 4 ? 4 : v); }

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
  GLB lh264_ctx_sym_t* out = as_glb<lh264_ctx_sym_t> (J->syms_dev) + (size_t)k * LH264_CTX_MAX_SYMS;
  GLB uint16_t* nout = as_glb<uint16_t> (J->n_syms_dev) + k;

  const int type = __builtin_amdgcn_readfirstlane ((int)m->mb_type);
  if (type == LH264_MB_SKIP || type == LH264_MB_IPCM || type == 0) {     // no coefficient symbols (writeBlock false / PCM)
    if (lane == 0) *nout = 0;
    return;
  }
  const int cbp = __builtin_amdgcn_readfirstlane ((int)m->cbp);
  const int t8 = __builtin_amdgcn_readfirstlane ((int)m->flags) & LH264_MBF_T8x8;
  const int sid = __builtin_amdgcn_readfirstlane ((int)m->slice_id);
  const int st = as_glb<const lh264_slice_t> (J->slices_dev)[sid].slice_type;
  const int mbc = mb_type_code (type);
  const int mbx = k % mb_w;
  // stage levels and the four nnz entries
  * (LDS v2i*) (W.lv + 4 * lane) = * (const GLB v2i*) (lv + 4 * lane);
  if (lane < 32) * (LDS v2i*) (W.lv + 256 + 4 * lane) = * (const GLB v2i*) (lv + 256 + 4 * lane);
  if (lane < 24) {
    const int who = lane / 6, q = lane % 6;
    uint32_t v = 0;
    if (who == 0) v = ((const GLB uint32_t*) (cur + (size_t)k * 24))[q];
    else if (who == 1) { if (mbx > 0) v = ((const GLB uint32_t*) (cur + (size_t) (k - 1) * 24))[q]; }
    else if (who == 2) { if (k >= mb_w) v = ((const GLB uint32_t*) (cur + (size_t) (k - mb_w) * 24))[q]; }
    else if (J->nnz_past_dev) v = ((const GLB uint32_t*) (as_glb<const uint8_t> (J->nnz_past_dev) + (size_t)k * 24))[q];
    ((LDS uint32_t*)W.nz[who])[q] = v;
  }
  asm volatile ("" ::: "memory");
  __builtin_amdgcn_wave_barrier();

  const int cbpl = cbp & 15, cbpc = cbp >> 4;
  const bool i16 = type == LH264_MB_I16x16;
  const bool cdc = cbpc == 1 || cbpc == 2;
  // one lane per block: 4x4 blocks -> lanes 0..23 ; with the 8x8 transform luma uses lanes 0,4,8,12
  int cnt = 0;
  if (lane < 24) {
    const int b = lane;
    const bool luma = b < 16;
    bool coded = luma ? ((cbpl >> (b >> 2)) & 1) : (cbpc == 2);
    int nco = 16, base = b * 16;
    if (luma && t8) { coded = coded && ((b & 3) == 0); nco = 64; }
    if (coded) {
      const int color = luma ? 0 : (b < 20 ? 1 : 2);
      const bool emit_dc = luma ? !i16 : !cdc;
      LDS const uint8_t* C = W.nz[0], *Lf = W.nz[1], *Ab = W.nz[2], *Pa = W.nz[3];
      int past, left, above;
      if (luma && t8) {
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
      LDS const int16_t* ac = W.lv + base;
      int nonzeros = 0;
      for (int i = emit_dc ? 0 : 1; i < nco; i++) nonzeros += ac[nco == 16 ? kZz16[i] : kZz64[i]] != 0;
      LDS uint32_t* sp = W.sp + b * 17;
      LDS int16_t* sv = W.sv + b * 17;
      sp[0] = (uint32_t) ((((((st * 16 + mbc) * 3 + color) * 3 + min2 (past)) * 3 + min2 (left)) * 3) + min2 (above));
      sv[0] = (int16_t)nonzeros;
      cnt = 1;
      int left_nz = nonzeros, prev = 0, prev2 = 0, emitted = 0;
      const uint32_t outer0 = (uint32_t) (((st * 16 + mbc) * 3 + color) * nco);
      for (int i = emit_dc ? 0 : 1; i < nco && left_nz > 0; i++) {
        const int c = ac[nco == 16 ? kZz16[i] : kZz64[i]];
        const uint32_t inner = (uint32_t) ((((min (4, left_nz) * 5 + clamp04 (prev + 2)) * 5 + clamp04 (prev2 + 2)) * 5 + 2) * 5 + 2);
        sp[cnt] = (outer0 + emitted) * 3125u + inner;
        sv[cnt] = (int16_t)c;
        cnt++;
        prev2 = prev; prev = c; emitted++;
        if (c) left_nz--;
      }
    }
  }
  // emission order: luma DC run, chroma DC run, blocks 0..23
  const int ndc_l = i16 ? 16 : 0, ndc_c = cdc ? 8 : 0;
  int incl = cnt;                                    // inclusive scan over lanes 0..23
  for (int d = 1; d < 32; d <<= 1) { const int t = __shfl_up (incl, d); if (lane >= d) incl += t; }
  const int total = __shfl (incl, 23) + ndc_l + ndc_c;
  const int my_off = ndc_l + ndc_c + incl - cnt;
  const int st16 = st;
  if (lane < 24) {
    const int b = lane;
    const uint8_t kind_nz = (b < 16 && t8) ? LH264_SYM_NZ8 : LH264_SYM_NZ4, kind_ac = (b < 16 && t8) ? LH264_SYM_AC8 : LH264_SYM_AC4;
    LDS const uint32_t* sp = W.sp + b * 17;
    LDS const int16_t* sv = W.sv + b * 17;
    for (int j = 0; j < cnt; j++) {
      put_sym (out + my_off + j, sp[j], sv[j], j == 0 ? kind_nz : kind_ac);
    }
  } else if (lane >= 32 && lane < 48) {
    if (i16) {                                       // getLumaDCIntPrior: lumaDCIntPriors[i][slice][mbtype]
      const int i = lane - 32;
      put_sym (out + i, (uint32_t) ((i * 5 + st16) * 16 + mbc), W.lv[i * 16], LH264_SYM_LUMA_DC);
    }
  } else if (lane >= 48 && lane < 56) {
    if (cdc) {
      const int i = lane - 48;
      put_sym (out + ndc_l + i, (uint32_t) ((i * 5 + st16) * 16 + mbc), W.lv[256 + i * 16], LH264_SYM_CHROMA_DC);
    }
  }
  if (lane == 0) *nout = (uint16_t)total;
}

}  // namespace lh264
