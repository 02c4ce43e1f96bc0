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


// position of raster coefficient i of a 4x4 / 8x8 block in scan order (the inverses of the zig-zag tables of pass 2), four to a word: a
// lane holds four consecutive raster coefficients and takes its word with one load
__constant__ uint32_t kInvZz16x4[4] = {0x06050100u, 0x0c070402u, 0x0d0b0803u, 0x0f0e0a09u};
__constant__ uint32_t kInvZz64x4[16] = {0x10080100u, 0x0a030209u, 0x19201811u, 0x05040b12u, 0x211a130cu, 0x22293028u, 0x060d141bu, 0x1c150e07u, 0x38312a23u, 0x242b3239u, 0x170f161du, 0x332c251eu, 0x2d343b3au, 0x2e271f26u, 0x363d3c35u, 0x3f3e372fu};

// lane exchanges inside quads / rows of 16 / the wave as DPP modifiers of a move (a __shfl_xor goes through the LDS crossbar)
template <int CTRL> __device__ __forceinline__ int dpp_mov (int v) { return __builtin_amdgcn_mov_dpp (v, CTRL, 0xf, 0xf, true); }
template <int CTRL, int ROWS> __device__ __forceinline__ int dpp_add0 (int x) { return __builtin_amdgcn_update_dpp (0, x, CTRL, ROWS, 0xf, false); }
__device__ __forceinline__ int quad_max (int v) { v = max (v, dpp_mov<0xB1> (v)); return max (v, dpp_mov<0x4E> (v)); }       // quad_perm [1,0,3,2], [2,3,0,1]
__device__ __forceinline__ int row16_max_of_quads (int v) { v = max (v, dpp_mov<0x124> (v)); return max (v, dpp_mov<0x128> (v)); }   // row_ror 4, 8
__device__ __forceinline__ int wave_sum (int x) {          // the sum over the 64 lanes (in lane 63, read back to all)
  x += dpp_add0<0x111, 0xf> (x); x += dpp_add0<0x112, 0xf> (x); x += dpp_add0<0x114, 0xf> (x); x += dpp_add0<0x118, 0xf> (x);
  x += dpp_add0<0x142, 0xa> (x); x += dpp_add0<0x143, 0xc> (x);
  return __builtin_amdgcn_readlane (x, 63);
}

// ---- pass 1a: nonzero counts of every coded macroblock, all frames of all streams at once; and HOW MANY symbols pass 2 will write
// for it (the compact layout places a macroblock's symbols behind its predecessors': a running sum of these counts) -------------
// symbols of a macroblock = 16 luma DC (Intra16x16) + 8 chroma DC (cbp chroma 1, 2) + per coded block 1 (its nonzero count) + the scan
// positions from the block's first (0, or 1 where the DC went separately) to its last nonzero one (encode4x4, decode_slice.cpp:2059-2094)
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
  GLB uint16_t* nout = as_glb<uint16_t> (J->n_syms_dev) + k;
  const GLB uint32_t* rec = (const GLB uint32_t*) (as_glb<const lh264_mb_t> (J->mbs_dev) + k);
  const uint32_t head_l = rec[0], flags_l = rec[1];             // mb_type | cbp << 16 | qp << 24 ; qp_c | flags << 16 | intra_avail << 24
  const v2i a = * (const GLB v2i*) (lv + 4 * lane);
  v2i b = {0, 0};
  if (lane < 32) b = * (const GLB v2i*) (lv + 256 + 4 * lane);
  asm volatile ("" : : "v"(a.x), "v"(a.y), "v"(b.x), "v"(b.y));      // keeps the loads above the branch
  const uint32_t head = (uint32_t)__builtin_amdgcn_readfirstlane ((int)head_l);
  const int type = head & 0xffff;
  if (type == LH264_MB_SKIP || type == 0) { if (lane == 0) *nout = 0; return; }       // inherited from PAST by pass 1b; no symbols
  if (((head >> 16) & 0xff) == 0 && type != LH264_MB_I16x16 && type != LH264_MB_IPCM) {
    // nothing coded (cbp 0): every count is zero, there is no symbol (the levels were requested with the record: one round trip)
    if (lane < 6) ((GLB uint32_t*)cur)[lane] = 0u;
    if (lane == 0) *nout = 0;
    return;
  }
  const int a0 = (a.x & 0xffff) != 0, a1 = (a.x >> 16) != 0, a2 = (a.y & 0xffff) != 0, a3 = (a.y >> 16) != 0;
  const int b0 = (b.x & 0xffff) != 0, b1 = (b.x >> 16) != 0, b2 = (b.y & 0xffff) != 0, b3 = (b.y >> 16) != 0;
  int c = a0 + a1 + a2 + a3;
  c += __shfl_xor (c, 1); c += __shfl_xor (c, 2);       // luma block = lane >> 2
  int d = b0 + b1 + b2 + b3;
  d += __shfl_xor (d, 1); d += __shfl_xor (d, 2);
  // gather the 24 counts into 6 dwords: lane 16*j + 4*i holds count 4*j + i
  const int c0 = __shfl (c, (lane & 3) * 16), c1 = __shfl (c, (lane & 3) * 16 + 4), c2 = __shfl (c, (lane & 3) * 16 + 8), c3 = __shfl (c, (lane & 3) * 16 + 12);
  const int d0 = __shfl (d, (lane & 1) * 16), d1 = __shfl (d, (lane & 1) * 16 + 4), d2 = __shfl (d, (lane & 1) * 16 + 8), d3 = __shfl (d, (lane & 1) * 16 + 12);
  if (lane < 4) ((GLB uint32_t*)cur)[lane] = (uint32_t)c0 | (uint32_t)c1 << 8 | (uint32_t)c2 << 16 | (uint32_t)c3 << 24;
  else if (lane < 6) ((GLB uint32_t*)cur)[lane] = (uint32_t)d0 | (uint32_t)d1 << 8 | (uint32_t)d2 << 16 | (uint32_t)d3 << 24;
  // ---- the symbol count (same cases as ctx_symbols_kernel below) -------------------------------------------------------------------
  if (type == LH264_MB_IPCM) { if (lane == 0) *nout = 0; return; }
  const int cbp = (head >> 16) & 0xff, cbpl = cbp & 15, cbpc = cbp >> 4;
  const bool t8 = (((uint32_t)__builtin_amdgcn_readfirstlane ((int)flags_l) >> 16) & LH264_MBF_T8x8) != 0;
  const bool i16 = type == LH264_MB_I16x16, cdc = cbpc == 1 || cbpc == 2;
  // luma: lane = 4 consecutive coefficients; 4x4: block lane >> 2 (z-order), raster 4 (lane & 3) + j; 8x8: block lane >> 4, raster 4 (lane & 15) + j
  const int ls = i16 ? 1 : 0;                                   // first scan position of a luma block's coefficient symbols
  // this lane's last nonzero scan position (-1: none).  Only raster coefficient 0 has scan position 0: where the DC went separately
  // (ls == 1) it is the block's head lane that leaves its first coefficient out
  const uint32_t pk4 = kInvZz16x4[lane & 3];
  const bool lhead = t8 ? (lane & 15) == 0 : (lane & 3) == 0;   // one lane per block
  int cnt = 0;
  if (cbpl) {
    const uint32_t pk = t8 ? kInvZz64x4[lane & 15] : pk4;
    const int v0 = (a0 && !(ls && lhead)) ? (int) (pk & 0xffu) : -1, v1 = a1 ? (int) ((pk >> 8) & 0xffu) : -1,
              v2 = a2 ? (int) ((pk >> 16) & 0xffu) : -1, v3 = a3 ? (int) (pk >> 24) : -1;
    int last = max (max (v0, v1), max (v2, v3));
    last = quad_max (last);
    if (t8) last = row16_max_of_quads (last);
    const bool lcoded = ((cbpl >> (lane >> 4)) & 1) != 0;      // the 8x8 quadrant of this lane's block (lane >> 4 either way)
    cnt = (lcoded && lhead) ? 1 + (last >= ls ? last - ls + 1 : 0) : 0;
  }
  // chroma AC blocks (cbp chroma 2): lanes 0..31, block lane >> 2, their DC went separately
  if (cbpc == 2) {
    const bool chead = (lane & 3) == 0;
    const int v0 = (b0 && !chead) ? (int) (pk4 & 0xffu) : -1, v1 = b1 ? (int) ((pk4 >> 8) & 0xffu) : -1,
              v2 = b2 ? (int) ((pk4 >> 16) & 0xffu) : -1, v3 = b3 ? (int) (pk4 >> 24) : -1;
    const int clast = quad_max (max (max (v0, v1), max (v2, v3)));
    if (lane < 32 && chead) cnt += 1 + (clast >= 1 ? clast : 0);
  }
  cnt = wave_sum (cnt);
  if (lane == 0) *nout = (uint16_t) (cnt + (i16 ? 16 : 0) + (cdc ? 8 : 0));
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

__device__ __forceinline__ uint64_t mk_sym (uint32_t prior, int value, int kind, int tag) {
  // {u32 prior, i16 value, u8 kind, u8 pad}; pad: the symbol's (first) tag, so that the coder does not have to take the prior apart again
  return (uint64_t)prior | ((uint64_t) (uint16_t)value << 32) | ((uint64_t) (uint8_t)kind << 48) | ((uint64_t) (uint8_t)tag << 56);
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
__device__ __forceinline__ void walk16 (const int c[16], int start, int last, int wmax, uint32_t outer0, int kind, int tag0, int tagn, WalkState& w, LDS uint64_t* dst) {
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const int pos = CH * 16 + i;
    if ((i & 3) == 0 && wmax < pos) return;
    if (pos >= start && pos <= last) {
      const uint32_t inner = (uint32_t) ((((min (4, w.left_nz) * 5 + clamp04 (w.prev + 2)) * 5 + clamp04 (w.prev2 + 2)) * 5 + 2) * 5 + 2);
      dst[w.emitted] = mk_sym ((outer0 + w.emitted) * 3125u + inner, c[i], kind, w.emitted ? tagn : tag0);
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

// ---- the compact layout: where each macroblock's symbols go ----------------------------------------------------------------------
// ctx_offsets_kernel: one workgroup per picture - running sum of n_syms over its macroblocks -> sym_off_dev[k], the picture's total
// ctx_bases_kernel:   one workgroup - running sum of the pictures' totals -> *sym_base_dev of every picture, the grand total
__global__ void __launch_bounds__ (256)
ctx_offsets_kernel (const lh264_ctx_job_t* __restrict__ jobs, int n_jobs, unsigned long long* __restrict__ job_total) {
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t carry;
  const int ji = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (ji >= n_jobs) return;
  const lh264_ctx_job_t* J = jobs + ji;
  if (tid == 0) carry = 0;
  __syncthreads();
  if (!J->sym_off_dev) { if (tid == 0) job_total[ji] = 0; return; }
  const int n = J->mb_w * J->mb_h;
  const GLB uint16_t* ns = as_glb<const uint16_t> (J->n_syms_dev);
  GLB uint32_t* off = as_glb<uint32_t> (J->sym_off_dev);
  for (int k0 = 0; k0 < n; k0 += 256) {
    const int k = k0 + tid;
    const uint32_t v = k < n ? ns[k] : 0u;
    uint32_t incl = v;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t t = (uint32_t)__shfl_up ((int)incl, d); if (lane >= d) incl += t; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t before = carry;
    for (int w = 0; w < wave; w++) before += wsum[w];
    if (k < n) off[k] = before + incl - v;
    __syncthreads();
    if (tid == 255) carry = before + incl;
    __syncthreads();
  }
  if (tid == 0) job_total[ji] = carry;
}
// (a wave per SIMD: beside the reconstruct kernel of the next batch there is room for one more wave per SIMD, and this kernel sits in
// the middle of the chain - with 1,024 threads it waited for a CU to come free)
#define CTX_ONE_WG 256
__global__ void __launch_bounds__ (CTX_ONE_WG)
ctx_bases_kernel (int n_jobs, unsigned long long* __restrict__ job_total, unsigned long long* __restrict__ total) {
  __shared__ unsigned long long wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // every thread a run of consecutive pictures: its sum, the running sum over the threads, then the run again.  job_total[j] becomes
  // the symbols in front of picture j (ctx_scatter_kernel hands them to the jobs)
  const int per = (n_jobs + CTX_ONE_WG - 1) / CTX_ONE_WG, j0 = min (tid * per, n_jobs), j1 = min (j0 + per, n_jobs);
  unsigned long long mine = 0;
  for (int j = j0; j < j1; j++) mine += job_total[j];
  unsigned long long incl = mine;
  for (int d = 1; d < 64; d <<= 1) { const unsigned long long t = (unsigned long long)__shfl_up ((long long)incl, d); if (lane >= d) incl += t; }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  unsigned long long at = incl - mine;
  for (int w = 0; w < wave; w++) at += wsum[w];
  for (int j = j0; j < j1; j++) { const unsigned long long v = job_total[j]; job_total[j] = at; at += v; }
  if (tid == CTX_ONE_WG - 1) { job_total[n_jobs] = at; if (total) *total = at; }
}
__global__ void __launch_bounds__ (256)
ctx_scatter_kernel (const lh264_ctx_job_t* __restrict__ jobs, int n_jobs, const unsigned long long* __restrict__ job_total) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j < n_jobs && jobs[j].sym_base_dev) *as_glb<unsigned long long> (jobs[j].sym_base_dev) = job_total[j];
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
  GLB uint16_t* nout = as_glb<uint16_t> (J->n_syms_dev) + k;
  // where the macroblock's symbols go: its fixed slot, or (compact layout) behind its predecessors' in the pool - the count pass has
  // said how many there are (n_syms), the running sums where (sym_off, sym_base); a pool that is too small is not written to.
  // (requested here with the other reads of the macroblock, looked at when the symbols are copied out)
  const bool compact = J->sym_off_dev != nullptr;
  unsigned long long sym_base_v = 0; uint32_t sym_off_v = 0, n_syms_v = 0;
  if (compact) { sym_base_v = *as_glb<const unsigned long long> (J->sym_base_dev); sym_off_v = as_glb<const uint32_t> (J->sym_off_dev)[k]; n_syms_v = *nout; }

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
                     big ? LH264_SYM_NZ8 : LH264_SYM_NZ4, color ? 29 : 19);
    const int tagn = color ? 29 : 24, tag0 = (color || mbc == 1) ? tagn : 19;      // encode4x4's tags: chroma / the first luma coefficient / the others
    const uint32_t outer0 = (uint32_t) (((st * 16 + mbc) * 3 + color) * nco);
    WalkState w = {nonzeros, 0, 0, 0};
    if (!big) walk16<0> (c, start, last, wmax, outer0, LH264_SYM_AC4, tag0, tagn, w, dst + 1);
    else {
      load_scan16<64, 0> (ac, c); walk16<0> (c, start, last, wmax, outer0, LH264_SYM_AC8, tag0, tagn, w, dst + 1);
      load_scan16<64, 1> (ac, c); walk16<1> (c, start, last, wmax, outer0, LH264_SYM_AC8, tag0, tagn, w, dst + 1);
      load_scan16<64, 2> (ac, c); walk16<2> (c, start, last, wmax, outer0, LH264_SYM_AC8, tag0, tagn, w, dst + 1);
      load_scan16<64, 3> (ac, c); walk16<3> (c, start, last, wmax, outer0, LH264_SYM_AC8, tag0, tagn, w, dst + 1);
    }
  }
  if (lane >= 32 && lane < 48) {
    if (i16) {                                       // getLumaDCIntPrior: lumaDCIntPriors[i][slice][mbtype]
      const int i = lane - 32;
      W.osym[i] = mk_sym ((uint32_t) ((i * 5 + st) * 16 + mbc), W.lv[i * 16], LH264_SYM_LUMA_DC, 17);
    }
  } else if (lane >= 48 && lane < 56) {
    if (cdc) {
      const int i = lane - 48;
      W.osym[ndc_l + i] = mk_sym ((uint32_t) ((i * 5 + st) * 16 + mbc), W.lv[256 + i * 16], LH264_SYM_CHROMA_DC, 18);
    }
  }
  asm volatile ("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  GLB uint64_t* out;
  bool room = true;
  if (compact) {
    const unsigned long long at = sym_base_v + sym_off_v;
    room = at + n_syms_v <= J->syms_cap && (int)n_syms_v == total;
    out = (GLB uint64_t*) (as_glb<lh264_ctx_sym_t> (J->syms_dev) + at);
  } else out = (GLB uint64_t*) (as_glb<lh264_ctx_sym_t> (J->syms_dev) + (size_t)k * LH264_CTX_MAX_SYMS);
  if (room) for (int i = lane; i < total; i += 64) out[i] = W.osym[i];
  if (lane == 0 && (!compact || !room)) *nout = room ? (uint16_t)total : (uint16_t)0;       // (compact layout: the count pass has written it)
}

}  // namespace lh264
