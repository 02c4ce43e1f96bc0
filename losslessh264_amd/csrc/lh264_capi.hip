// lh264_capi.hip - the C ABI declared in include/lh264.h (host side of the hot path).
// No CPU fallback: every compute entry point fails with LH264_E_NODEVICE when no HIP device is
// visible.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <atomic>
#include <thread>
#include <vector>
#include "../../include/lh264.h"
#include "host/h264_parser.h"
#include "host/capi_internal.h"
#include "host/pip_restore.h"
#include "lh264_coder.h"
#include <mutex>

namespace lh264 {
__global__ void recon_chain_kernel (const lh264_frame_job_t* jobs, const int32_t* chain_first, int n_chains, int line_bytes);
__global__ void ctx_nnz_kernel (const lh264_ctx_job_t* jobs, int n_jobs, int blocks_per_job);
__global__ void ctx_inherit_chain_kernel (const lh264_ctx_job_t* jobs, const int32_t* chain_first, int n_chains);
__global__ void ctx_symbols_kernel (const lh264_ctx_job_t* jobs, int n_jobs, int blocks_per_job);
__global__ void ctx_offsets_kernel (const lh264_ctx_job_t* jobs, int n_jobs, unsigned long long* job_total);
__global__ void ctx_bases_kernel (int n_jobs, unsigned long long* job_total, unsigned long long* total);
__global__ void ctx_scatter_kernel (const lh264_ctx_job_t* jobs, int n_jobs, const unsigned long long* job_total);
__global__ void coder_jobs_kernel (const lh264_code_job_t* jobs, const int32_t* chain_first, int n_jobs, int n_chains, unsigned seg_bound, uint32_t* seg0, uint32_t* seg_job, uint32_t* job_chain, uint32_t* chain_info);
__global__ void coder_count_kernel (const lh264_code_job_t* jobs, const uint32_t* seg0, const uint32_t* seg_job, int n_jobs, uint32_t* seg_cnt, uint32_t* seg_bkt);
__global__ void coder_balance_kernel (const uint32_t* seg0, const int32_t* chain_first, const uint32_t* seg_bkt, int n_chains, int log2p, uint8_t* chain_map);
__global__ void coder_partoff_kernel (const uint32_t* seg0, const uint32_t* seg_job, const uint32_t* job_chain, int n_jobs, int log2p, const uint32_t* seg_bkt,
                                      const uint8_t* chain_map, uint32_t* seg_part);
__global__ void coder_scan_kernel (const uint32_t* seg0, const int32_t* chain_first, uint32_t* seg_cnt, uint32_t* seg_doff, uint32_t* chain_info, int n_chains);
__global__ void coder_bases_kernel (uint32_t* chain_info, int n_chains, unsigned long long* totals);
__global__ void coder_emit_kernel (const lh264_code_job_t* jobs, const uint32_t* seg0, const uint32_t* seg_job, const uint32_t* job_chain, int n_jobs, int log2p,
                                   const uint32_t* seg_doff, const uint32_t* seg_cnt, const uint32_t* seg_part, const uint8_t* chain_map, const uint32_t* chain_info, uint64_t* D);
__global__ void coder_resolve_kernel (const lh264_code_stream_t* streams, uint32_t* chain_info, const uint32_t* seg0, const int32_t* chain_first,
                                      const uint32_t* seg_doff, const uint32_t* seg_part, const uint64_t* D, uint16_t* Q, int n_chains, int log2p, uint32_t* progress, int window);
__global__ void coder_chunkmap_kernel (const uint32_t* chain_info, int n_pairs, uint32_t* pair_chunk0, uint32_t* pair_coarse0, uint32_t* cand_list);
__global__ void coder_range_seed_kernel (const uint32_t* chain_info, const uint16_t* Q, const uint32_t* pair_coarse0, int n_pairs, uint32_t* cand, uint32_t* cand_list, uint32_t long_list);
__global__ void coder_range_first_kernel (const uint32_t* chain_info, const uint16_t* Q, const uint32_t* pair_chunk0, const uint32_t* pair_coarse0, int n_pairs, int groups,
                                          unsigned n_cand, unsigned n_later, const uint32_t* cand, const uint32_t* cand_list, uint8_t* cand_end, uint8_t* cmap, uint32_t* chunk_rec, uint32_t* coarse_bits);
__global__ void coder_range_link_kernel (const uint32_t* pair_coarse0, int n_pairs, const uint32_t* cand, const uint8_t* cand_end, const uint8_t* cmap, uint32_t* seed, uint32_t* chain_info);
__global__ void coder_range_walk_kernel (const uint32_t* chain_info, const uint16_t* Q, const uint32_t* pair_chunk0, const uint32_t* pair_coarse0, int n_pairs,
                                         const uint32_t* cand, const uint32_t* seed, uint32_t* chunk_rec, uint32_t* coarse_bits);
__global__ void coder_range_scan_kernel (const uint32_t* pair_coarse0, int n_pairs, uint32_t* coarse_bits, uint32_t* pair_bits, const uint32_t* chain_info,
                                         const uint16_t* Q, uint32_t* acc);
__global__ void coder_accum_kernel (const uint32_t* chain_info, const uint16_t* Q, const uint32_t* pair_chunk0, const uint32_t* pair_coarse0, int n_pairs,
                                    const uint32_t* chunk_rec, const uint32_t* coarse_bits, const uint32_t* pair_bits, uint32_t* acc);
__global__ void coder_bytes_kernel (const lh264_code_stream_t* streams, const uint32_t* chain_info, const uint16_t* Q, const uint32_t* pair_bits,
                                    const uint32_t* acc, int n_pairs);
__global__ void coder_status_kernel (const lh264_code_stream_t* streams, const uint32_t* chain_info, int n_chains);
size_t wave_lds_bytes();
size_t wg_lds_bytes();
#ifdef LH264_CODER_DEBUG
void read_rs_stamps (unsigned long long* out, bool reset);
#endif
#ifdef LH264_STAMP
void read_stamps (unsigned long long* out, bool reset);
#endif
}

// the stream-per-workgroup form of the coder's first stages (lh264_coder_sw.hip)
namespace lh264sw {
__global__ void coder_jobs_kernel (const lh264_code_job_t* jobs, const int32_t* chain_first, int n_jobs, int n_chains, uint32_t* seg0, uint32_t* job_chain, uint32_t* chain_info);
__global__ void coder_count_kernel (const lh264_code_job_t* jobs, const uint32_t* seg0, int n_jobs, uint32_t* seg_cnt);
__global__ void coder_scan_kernel (const uint32_t* seg0, const int32_t* chain_first, const uint32_t* seg_cnt, uint32_t* seg_doff, uint32_t* chain_info, int n_chains);
__global__ void coder_bases_kernel (uint32_t* chain_info, int n_chains, unsigned long long* totals);
__global__ void coder_emit_kernel (const lh264_code_job_t* jobs, const uint32_t* seg0, const uint32_t* job_chain, int n_jobs,
                                   const uint32_t* seg_doff, const uint32_t* chain_info, uint64_t* D);
__global__ void coder_resolve_kernel (const lh264_code_stream_t* streams, uint32_t* chain_info, const uint64_t* D, uint16_t* Q, int n_chains);
}

static thread_local std::string g_err;
static int fail (int code, const char* what, hipError_t e = hipSuccess) {
  char buf[256];
  if (e != hipSuccess) snprintf (buf, sizeof (buf), "%s: %s", what, hipGetErrorString (e));
  else snprintf (buf, sizeof (buf), "%s", what);
  g_err = buf;
  return code;
}
#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail (LH264_E_HIP, #call, e_); } while (0)

extern "C" {

int lh264_abi_version (void) { return LH264_ABI_VERSION; }
#ifndef LH264_BUILD_ID
#define LH264_BUILD_ID "unknown"
#endif
const char* lh264_build_id (void) { return LH264_BUILD_ID; }
const char* lh264_last_error (void) { return g_err.c_str(); }

int lh264_device_count (void) {
  int n = 0;
  if (hipGetDeviceCount (&n) != hipSuccess) return 0;
  return n;
}
int lh264_set_device (int device) {
  if (lh264_device_count() <= 0) return fail (LH264_E_NODEVICE, "no HIP device visible");
  HIPCHK (hipSetDevice (device));
  return LH264_OK;
}

void* lh264_dev_malloc (size_t bytes) {
  void* p = nullptr;
  if (lh264_device_count() <= 0) { fail (LH264_E_NODEVICE, "no HIP device visible"); return nullptr; }
  hipError_t e = hipMalloc (&p, bytes);
  if (e != hipSuccess) { fail (LH264_E_HIP, "hipMalloc", e); return nullptr; }
  return p;
}
int lh264_dev_free (void* p) { HIPCHK (hipFree (p)); return LH264_OK; }
int lh264_memcpy_h2d (void* dst, const void* src, size_t bytes, void* stream) {
  HIPCHK (hipMemcpyAsync (dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  return LH264_OK;
}
int lh264_memcpy_d2h (void* dst, const void* src, size_t bytes, void* stream) {
  HIPCHK (hipMemcpyAsync (dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
  return LH264_OK;
}
int lh264_dev_memset (void* dst, int value, size_t bytes, void* stream) {
  HIPCHK (hipMemsetAsync (dst, value, bytes, (hipStream_t)stream));
  return LH264_OK;
}
int lh264_stream_sync (void* stream) { HIPCHK (hipStreamSynchronize ((hipStream_t)stream)); return LH264_OK; }

// the reference's picture layout (pic_queue.cpp:62-112): 32 / 16 samples of padding, luma stride aligned to 32
size_t lh264_pic_bytes (int mb_w, int mb_h, int* stride_y, int* stride_c, size_t* off_y, size_t* off_u, size_t* off_v) {
  const int w = mb_w * 16, h = mb_h * 16;
  const int sy = (w + 2 * LH264_PAD_LUMA + 31) & ~31, sc = sy >> 1;
  const size_t hy = (size_t)h + 2 * LH264_PAD_LUMA, hc = (size_t) (h >> 1) + 2 * LH264_PAD_CHROMA;
  if (stride_y) *stride_y = sy;
  if (stride_c) *stride_c = sc;
  if (off_y) *off_y = (size_t)LH264_PAD_LUMA * sy + LH264_PAD_LUMA;
  if (off_u) *off_u = (size_t)sy * hy + (size_t)LH264_PAD_CHROMA * sc + LH264_PAD_CHROMA;
  if (off_v) *off_v = (size_t)sy * hy + (size_t)sc * hc + (size_t)LH264_PAD_CHROMA * sc + LH264_PAD_CHROMA;
  return (size_t)sy * hy + 2 * (size_t)sc * hc + 64;   // + slack for dword reads at the very end
}

// launch geometry: one wave per macroblock row, at most 8 waves; a wave64 pool larger than the number of rows
// that can be in flight (min(rows, ceil(w/2))) is wasted.  LDS: per-wave tiles + (NW+1) row slots holding the
// unfiltered (32 B/MB + 96) and filtered (96 B/MB) bottom rows of a macroblock row.
static int pick_waves (int max_mb_w, int max_mb_h, int slot_bytes, size_t* lds_out) {
  int inflight = (max_mb_w + 1) / 2;
  if (inflight > max_mb_h) inflight = max_mb_h;
  int nw = 1;
  while (nw < inflight && nw < 8) nw <<= 1;
  if (const char* e = getenv ("LH264_WAVES")) { const int v = atoi (e); if (v >= 1 && v <= 8) nw = v; }   // tuning experiments
  for (;;) {
    const size_t lds = lh264::wg_lds_bytes() + lh264::wave_lds_bytes() * nw + (size_t) (nw + 1) * slot_bytes;
    if (lds <= 160 * 1024 || nw == 1) { *lds_out = lds; return nw; }
    nw >>= 1;
  }
}

static int launch_chains (const lh264_frame_job_t* jobs_dev, const int32_t* chain_first_dev, int n_chains,
                          int max_mb_w, int max_mb_h, hipStream_t st) {
  if (lh264_device_count() <= 0) return fail (LH264_E_NODEVICE, "no HIP device visible");
  if (!jobs_dev || !chain_first_dev || n_chains < 0 || max_mb_w <= 0 || max_mb_h <= 0) return fail (LH264_E_ARG, "bad argument");
  if (n_chains == 0) return LH264_OK;
  const int slot_bytes = 128 * max_mb_w + 96;
  size_t lds = 0;
  const int nw = pick_waves (max_mb_w, max_mb_h, slot_bytes, &lds);
  if (lds > 160 * 1024) return fail (LH264_E_UNSUPPORTED, "picture too wide for the LDS line buffers");
  static bool attr_set = false;
  if (!attr_set) {
    HIPCHK (hipFuncSetAttribute ((const void*)lh264::recon_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL (lh264::recon_chain_kernel, dim3 (n_chains), dim3 (nw * 64), lds, st, jobs_dev, chain_first_dev, n_chains, slot_bytes);
  HIPCHK (hipGetLastError());
  return LH264_OK;
}

int lh264_recon_chains (const lh264_frame_job_t* jobs_dev, const int32_t* chain_first_dev, int n_chains,
                        int max_mb_w, int max_mb_h, void* stream) {
  return launch_chains (jobs_dev, chain_first_dev, n_chains, max_mb_w, max_mb_h, (hipStream_t)stream);
}

int lh264_recon_frames (const lh264_frame_job_t* jobs_dev, int n_jobs, int max_mb_w, int max_mb_h, void* stream) {
  if (lh264_device_count() <= 0) return fail (LH264_E_NODEVICE, "no HIP device visible");
  if (n_jobs <= 0) return n_jobs == 0 ? LH264_OK : fail (LH264_E_ARG, "bad argument");
  // independent frames = chains of length one; the index table lives in a small cached device buffer
  static int32_t* idx_dev = nullptr;
  static int idx_cap = 0;
  hipStream_t st = (hipStream_t)stream;
  if (n_jobs + 1 > idx_cap) {
    if (idx_dev) hipFree (idx_dev);
    idx_cap = n_jobs + 1 + 1024;
    HIPCHK (hipMalloc (&idx_dev, sizeof (int32_t) * idx_cap));
    int32_t* h = new int32_t[idx_cap];
    for (int i = 0; i < idx_cap; i++) h[i] = i;
    hipError_t e = hipMemcpy (idx_dev, h, sizeof (int32_t) * idx_cap, hipMemcpyHostToDevice);
    delete[] h;
    if (e != hipSuccess) return fail (LH264_E_HIP, "hipMemcpy", e);
  }
  return launch_chains (jobs_dev, idx_dev, n_jobs, max_mb_w, max_mb_h, st);
}

// per device: the pictures' symbol totals of the compact layout (n_jobs + 1 words), kept between calls
namespace {
struct CtxWs { std::mutex mu; void* totals = nullptr; size_t cap = 0; };
CtxWs g_ctx_ws[16];
}
static int ctx_passes (const lh264_ctx_job_t* jobs_dev, const int32_t* chain_first_dev, int n_chains, int n_jobs, int max_mbs_per_frame,
                       unsigned long long* total_dev, bool symbols, hipStream_t st) {
  if (lh264_device_count() <= 0) return fail (LH264_E_NODEVICE, "no HIP device visible");
  if (!jobs_dev || !chain_first_dev || n_chains < 0 || n_jobs < 0 || max_mbs_per_frame <= 0) return fail (LH264_E_ARG, "bad argument");
  if (n_chains == 0 || n_jobs == 0) { if (total_dev) HIPCHK (hipMemsetAsync (total_dev, 0, 8, st)); return LH264_OK; }
  int dev = 0;
  HIPCHK (hipGetDevice (&dev));
  if (dev < 0 || dev >= 16) return fail (LH264_E_ARG, "device index out of range");
  CtxWs& W = g_ctx_ws[dev];
  std::lock_guard<std::mutex> lock (W.mu);
  const size_t need = ((size_t)n_jobs + 2) * 8;
  if (need > W.cap) {
    if (W.totals) (void)hipFree (W.totals);            // (synchronises the device: nobody reads the old one any more)
    W.totals = nullptr; W.cap = 0;
    HIPCHK (hipMalloc (&W.totals, need + 4096));
    W.cap = need + 4096;
  }
  const int bpj = (max_mbs_per_frame + 3) / 4;
  hipLaunchKernelGGL (lh264::ctx_nnz_kernel, dim3 ((unsigned)n_jobs * bpj), dim3 (256), 0, st, jobs_dev, n_jobs, bpj);
  HIPCHK (hipGetLastError());
  hipLaunchKernelGGL (lh264::ctx_inherit_chain_kernel, dim3 (n_chains), dim3 (256), 0, st, jobs_dev, chain_first_dev, n_chains);
  HIPCHK (hipGetLastError());
  // the compact layout: where every macroblock's symbols go (pictures in the fixed layout count as empty)
  hipLaunchKernelGGL (lh264::ctx_offsets_kernel, dim3 (n_jobs), dim3 (256), 0, st, jobs_dev, n_jobs, (unsigned long long*)W.totals);
  HIPCHK (hipGetLastError());
  hipLaunchKernelGGL (lh264::ctx_bases_kernel, dim3 (1), dim3 (256), 0, st, n_jobs, (unsigned long long*)W.totals, total_dev);
  HIPCHK (hipGetLastError());
  hipLaunchKernelGGL (lh264::ctx_scatter_kernel, dim3 ((unsigned) ((n_jobs + 255) / 256)), dim3 (256), 0, st, jobs_dev, n_jobs, (const unsigned long long*)W.totals);
  HIPCHK (hipGetLastError());
  if (symbols) {
    hipLaunchKernelGGL (lh264::ctx_symbols_kernel, dim3 ((unsigned)n_jobs * bpj), dim3 (256), 0, st, jobs_dev, n_jobs, bpj);
    HIPCHK (hipGetLastError());
  }
  return LH264_OK;
}
int lh264_ctx_index_chains (const lh264_ctx_job_t* jobs_dev, const int32_t* chain_first_dev, int n_chains,
                            int n_jobs, int max_mbs_per_frame, void* stream) {
  return ctx_passes (jobs_dev, chain_first_dev, n_chains, n_jobs, max_mbs_per_frame, nullptr, true, (hipStream_t)stream);
}
int lh264_ctx_count_chains (const lh264_ctx_job_t* jobs_dev, const int32_t* chain_first_dev, int n_chains,
                            int n_jobs, int max_mbs_per_frame, unsigned long long* total_dev, void* stream) {
  if (!total_dev) return fail (LH264_E_ARG, "bad argument");
  return ctx_passes (jobs_dev, chain_first_dev, n_chains, n_jobs, max_mbs_per_frame, total_dev, false, (hipStream_t)stream);
}

// Work memory of the coder stages, kept between calls and grown on demand: one set per device.  The lock only serialises the host
// side (the enqueue); on the device the calls are ordered by an event: every coder call records `done` on its stream when it has
// enqueued its last kernel, and a call on ANOTHER stream first makes its stream wait for it - the decision words, tag lists and sums
// of the previous call are still being read until then.  (Calls on one stream are ordered by the stream.)
namespace {
struct CoderWs {
  std::mutex mu;
  hipEvent_t done = nullptr; hipStream_t last_stream = nullptr; bool busy = false;
  // before the first kernel of a coder call on `st`
  int enter (hipStream_t st) {
    if (busy && st != last_stream) { hipError_t e = hipStreamWaitEvent (st, done, 0); if (e != hipSuccess) return (int)e; }
    return 0;
  }
  // behind its last kernel
  int leave (hipStream_t st) {
    if (!done) { hipError_t e = hipEventCreateWithFlags (&done, hipEventDisableTiming); if (e != hipSuccess) return (int)e; }
    hipError_t e = hipEventRecord (done, st);
    if (e != hipSuccess) return (int)e;
    last_stream = st; busy = true;
    return 0;
  }
  void* small = nullptr; size_t small_cap = 0;     // job / macroblock / stream tables
  void* big = nullptr; size_t big_cap = 0;         // decision words + tag lists
  unsigned long long* totals_host = nullptr;       // page-locked, 2 x u64
  unsigned long long last_words = 0, last_q = 0;   // of the last call: decisions (incl. per-stream padding to 64), list entries
  // what lh264_code_binarise_chains leaves for lh264_code_finish_chains
  int ready_chains = -1, n_pairs = 0; size_t chunk_bound = 0; const lh264_code_stream_t* ready_streams = nullptr;
  uint32_t* info = nullptr; uint64_t* D = nullptr; uint16_t* Q = nullptr;
  uint32_t* pair_chunk0 = nullptr; uint32_t* pair_bits = nullptr; uint32_t* chunk_rec = nullptr; uint32_t* acc = nullptr;
  uint32_t* pair_coarse0 = nullptr; uint32_t* seed = nullptr; uint32_t* coarse_bits = nullptr; size_t coarse_bound = 0; int n_pairs_last = 0;
  uint32_t* cand = nullptr; uint8_t* cand_end = nullptr; uint8_t* cmap = nullptr; uint32_t* cand_list = nullptr;
  const uint32_t* seg0 = nullptr; const uint32_t* seg_doff = nullptr; const uint32_t* seg_part = nullptr; const int32_t* chain_first = nullptr; int log2p = 3; bool sw = false;
  uint32_t* progress = nullptr; int window = 0;    // the resolve kernel's waves of a stream keep within `window` segments of one another
};
CoderWs g_coder_ws[16];
int grow (void** p, size_t* cap, size_t need) {
  if (need <= *cap) return LH264_OK;
  if (*p) { (void)hipFree (*p); *p = nullptr; *cap = 0; }
  const size_t want = need + need / 8 + 4096;
  hipError_t e = hipMalloc (p, want);
  if (e != hipSuccess) return fail (LH264_E_HIP, "hipMalloc (coder work memory)", e);
  *cap = want;
  return LH264_OK;
}
size_t up256 (size_t v) { return (v + 255) & ~ (size_t)255; }
}

// the two halves of lh264_code_chains: binarise (count, scan, bases, one synchronisation for the sizes, emit) and code (resolve, range,
// accumulate, bytes).  What the second half needs of the first is kept in the device's work space.
static int code_binarise (CoderWs& W, const lh264_code_job_t* jobs_dev, const int32_t* chain_first_dev, const lh264_code_stream_t* streams_dev,
                          int n_chains, int n_jobs, long long total_mbs, hipStream_t st) {
  W.ready_chains = -1;
  if (W.enter (st)) return fail (LH264_E_HIP, "hipStreamWaitEvent (coder work memory)");
  if (!W.totals_host) HIPCHK (hipHostMalloc ((void**)&W.totals_host, 2 * sizeof (unsigned long long), hipHostMallocDefault));
  // Two forms of the first stages (binarisation, adaptive probabilities), chosen per call:
  //   wave per (stream, partition) (lh264_coder.hip): scales inside a stream; with balanced partitions and the waves of a stream kept
  //     together it is the faster one on nearly every batch measured (1,024 - 2,048 QCIF streams, the mixed batch, 1,024 CIF streams,
  //     the 720p and 1080p batches: by 2 - 15 %)
  //   stream per workgroup (lh264_coder_sw.hip): a stream's decisions in coding order, one workgroup of 8 waves resolves them - round
  //     2's form; still 2 % ahead where a batch is a few hundred small streams of about the same size (512 QCIF streams: 26.6 against
  //     27.2 ms a step beside the reconstruct kernel - it leaves that kernel more of the vector ALUs)
  // LH264_CODER_PATH=sw|wave overrides, for the tests (both forms must give the same bytes) and for experiments.
  bool sw = n_chains >= 384 && n_chains < 1024 && total_mbs / n_chains <= 12288;
  if (const char* e = getenv ("LH264_CODER_PATH")) { if (!strcmp (e, "sw")) sw = true; else if (!strcmp (e, "wave")) sw = false; }
  W.sw = sw;
  // the partitions of a stream's DynProbs (each resolved by a wave of its own): as few as fill the machine with waves - a partition's
  // runs of decision words get shorter with their number, and a run costs its wave a look at the segment tables
  // (LH264_CODER_LOG2P overrides, for experiments)
  // (measured: 256 x 16 beats 256 x 8 and x 32 - 40 against 60 and 54 ms on the 1080p batch; 512 x 8 beats 512 x 16 on QCIF streams -
  // 4.9 against 6.6 ms -, 1,024 x 8 beats x 16 on CIF streams - 4.6 against 5.6 -, but 1,024 x 16 beats x 8 on 720p streams - 33 against 36)
  int log2p = (n_chains >= 512 && total_mbs / n_chains <= 12288) ? 3 : 4;
  while (log2p < LH264_CODER_MAX_LOG2P && ((long long)n_chains << log2p) < 2048) log2p++;
  if (const char* e = getenv ("LH264_CODER_LOG2P")) { const int v = atoi (e); if (v >= 0 && v <= LH264_CODER_MAX_LOG2P) log2p = v; }
  W.log2p = log2p;
  const size_t pstride = ((size_t)1 << log2p) + 1;
  // small tables.  A picture of n macroblocks is cut into ceil (n / segment size) segments
  const size_t seg_mbs = sw ? LH264_CODER_SW_SEG_MBS : LH264_CODER_SEG_MBS;
  const size_t seg_bound = (size_t)total_mbs / seg_mbs + (size_t)n_jobs + 1;
  const size_t o_seg0 = 0, o_jobchain = up256 ((size_t) (n_jobs + 1) * 4), o_info = o_jobchain + up256 ((size_t) (n_jobs + 1) * 4),
               o_totals = o_info + up256 ((size_t)n_chains * LH264_CODER_INFO_WORDS * 4), o_doff = o_totals + 256,
               o_sjob = o_doff + up256 (seg_bound * 4 + 4), o_cnt = o_sjob + up256 (seg_bound * 4 + 4), o_part = o_cnt + up256 (seg_bound * LH264_CODER_CNT_STRIDE * 4 + 4),
               o_bkt = o_part + up256 (sw ? 4 : seg_bound * pstride * 4 + 4), o_map = o_bkt + up256 (sw ? 4 : seg_bound * LH264_CODER_MAX_PARTS * 4),
               o_prog = o_map + up256 ((size_t)n_chains * LH264_CODER_MAX_PARTS), small_need = o_prog + up256 ((size_t)n_chains * LH264_CODER_MAX_PARTS * 4);
  if (int rc = grow (&W.small, &W.small_cap, small_need)) return rc;
  uint8_t* sm = (uint8_t*)W.small;
  uint32_t* seg0 = (uint32_t*) (sm + o_seg0); uint32_t* job_chain = (uint32_t*) (sm + o_jobchain); uint32_t* info = (uint32_t*) (sm + o_info);
  unsigned long long* totals = (unsigned long long*) (sm + o_totals); uint32_t* seg_doff = (uint32_t*) (sm + o_doff); uint32_t* seg_cnt = (uint32_t*) (sm + o_cnt);
  uint32_t* seg_part = (uint32_t*) (sm + o_part); uint32_t* seg_job = (uint32_t*) (sm + o_sjob);
  uint32_t* seg_bkt = (uint32_t*) (sm + o_bkt); uint8_t* chain_map = sm + o_map;     // decisions per bucket of cells; bucket -> partition
  const unsigned seg_blocks = (unsigned) ((seg_bound + LH264_CODER_WG_WAVES - 1) / LH264_CODER_WG_WAVES), seg_threads = 64 * LH264_CODER_WG_WAVES;
  if (sw) {
    hipLaunchKernelGGL (lh264sw::coder_jobs_kernel, dim3 (1), dim3 (CODER_ONE_WG), 0, st, jobs_dev, chain_first_dev, n_jobs, n_chains, seg0, job_chain, info);
    HIPCHK (hipGetLastError());
    if (n_jobs > 0 && total_mbs > 0) {
      hipLaunchKernelGGL (lh264sw::coder_count_kernel, dim3 ((unsigned)seg_bound), dim3 (256), 0, st, jobs_dev, seg0, n_jobs, seg_cnt);
      HIPCHK (hipGetLastError());
    }
    hipLaunchKernelGGL (lh264sw::coder_scan_kernel, dim3 (n_chains), dim3 (64), 0, st, seg0, chain_first_dev, seg_cnt, seg_doff, info, n_chains);
    HIPCHK (hipGetLastError());
    hipLaunchKernelGGL (lh264sw::coder_bases_kernel, dim3 (1), dim3 (CODER_ONE_WG), 0, st, info, n_chains, totals);
    HIPCHK (hipGetLastError());
  } else {
    hipLaunchKernelGGL (lh264::coder_jobs_kernel, dim3 (1), dim3 (CODER_ONE_WG), 0, st, jobs_dev, chain_first_dev, n_jobs, n_chains, (unsigned)seg_bound, seg0, seg_job, job_chain, info);
    HIPCHK (hipGetLastError());
    if (n_jobs > 0 && total_mbs > 0) {
      hipLaunchKernelGGL (lh264::coder_count_kernel, dim3 (seg_blocks), dim3 (seg_threads), 0, st, jobs_dev, seg0, seg_job, n_jobs, seg_cnt, seg_bkt);
      HIPCHK (hipGetLastError());
    }
    // the stream's partitions: its buckets of cells dealt out evenly; then where each partition's run starts in every segment
    hipLaunchKernelGGL (lh264::coder_balance_kernel, dim3 (n_chains), dim3 (64), 0, st, seg0, chain_first_dev, seg_bkt, n_chains, log2p, chain_map);
    HIPCHK (hipGetLastError());
    if (n_jobs > 0 && total_mbs > 0) {
      hipLaunchKernelGGL (lh264::coder_partoff_kernel, dim3 (seg_blocks), dim3 (seg_threads), 0, st, seg0, seg_job, job_chain, n_jobs, log2p, seg_bkt, chain_map, seg_part);
      HIPCHK (hipGetLastError());
    }
    hipLaunchKernelGGL (lh264::coder_scan_kernel, dim3 (n_chains), dim3 (64), 0, st, seg0, chain_first_dev, seg_cnt, seg_doff, info, n_chains);
    HIPCHK (hipGetLastError());
    hipLaunchKernelGGL (lh264::coder_bases_kernel, dim3 (1), dim3 (CODER_ONE_WG), 0, st, info, n_chains, totals);
    HIPCHK (hipGetLastError());
  }
  // the sizes of the decision words and of the tag lists are only known now
  HIPCHK (hipMemcpyAsync (W.totals_host, totals, 2 * sizeof (unsigned long long), hipMemcpyDeviceToHost, st));
  HIPCHK (hipStreamSynchronize (st));
  const unsigned long long n_words = W.totals_host[0], n_q = W.totals_host[1];
  // the bool coder's tables: per (stream, tag slot) pair the first chunk (256 decisions) and the bits shifted out, per chunk its state at
  // the first decision, per output byte position a 32-bit sum (one per list entry + 48 per pair bounds them)
  const int n_pairs = n_chains * LH264_N_TAG_SLOTS;
  if (n_pairs >= (1 << 24)) return fail (LH264_E_ARG, "too many streams in one call");
  const size_t chunk_bound = (size_t)n_q / LH264_CODER_CODE_CHUNK + 2 * (size_t)n_pairs + 1;
  const size_t n_acc = (size_t)n_q + 48 * (size_t)n_pairs + 64;
  // coarse chunks of the range walk (LH264_CODER_CODE_COARSE decisions): per pair the first one; per chunk its candidate start states (8
  // bytes; + 4 more words in diagnostic builds), where a walk from each ends, its true start state, the bits it shifts out
  const size_t coarse_bound = (size_t)n_q / LH264_CODER_CODE_COARSE + 2 * (size_t)n_pairs + 1;
  const size_t o_q = up256 ((size_t)n_words * 8 + 512), o_pc0 = o_q + up256 ((size_t)n_q * 2 + 256), o_pbits = o_pc0 + up256 ((size_t) (n_pairs + 1) * 4),
               o_crec = o_pbits + up256 ((size_t)n_pairs * 4), o_pco0 = o_crec + up256 (chunk_bound * 8), o_seed = o_pco0 + up256 ((size_t) (n_pairs + 1) * 4),
               o_cand = o_seed + up256 (coarse_bound * 4), o_cend = o_cand + up256 (coarse_bound * 12), o_cmap = o_cend + up256 (coarse_bound * 8), o_clist = o_cmap + up256 (coarse_bound * 128), o_cbits = o_clist + up256 (coarse_bound * 32 + 4),
               o_acc = o_cbits + up256 (coarse_bound * 4 + 4);
  if (int rc = grow (&W.big, &W.big_cap, o_acc + n_acc * 4 + 256)) return rc;
  uint8_t* bg = (uint8_t*)W.big;
  uint64_t* D = (uint64_t*)bg; uint16_t* Q = (uint16_t*) (bg + o_q);
  uint32_t* pair_chunk0 = (uint32_t*) (bg + o_pc0); uint32_t* pair_bits = (uint32_t*) (bg + o_pbits);
  uint32_t* chunk_rec = (uint32_t*) (bg + o_crec); uint32_t* acc = (uint32_t*) (bg + o_acc);
  W.pair_coarse0 = (uint32_t*) (bg + o_pco0); W.seed = (uint32_t*) (bg + o_seed); W.coarse_bits = (uint32_t*) (bg + o_cbits); W.coarse_bound = coarse_bound; W.n_pairs_last = n_pairs;
  W.cand = (uint32_t*) (bg + o_cand); W.cand_end = (uint8_t*) (bg + o_cend); W.cmap = bg + o_cmap; W.cand_list = (uint32_t*) (bg + o_clist);
  W.seg0 = seg0; W.seg_doff = seg_doff; W.seg_part = seg_part; W.chain_first = chain_first_dev;
  W.progress = (uint32_t*) (sm + o_prog); W.window = 3;
  if (const char* e = getenv ("LH264_CODER_WINDOW")) W.window = atoi (e);      // (experiments; 0: the waves run free)
  if (n_jobs > 0 && total_mbs > 0) {
    if (sw) hipLaunchKernelGGL (lh264sw::coder_emit_kernel, dim3 ((unsigned)seg_bound), dim3 (256), 0, st, jobs_dev, seg0, job_chain, n_jobs, seg_doff, info, D);
    else hipLaunchKernelGGL (lh264::coder_emit_kernel, dim3 (seg_blocks), dim3 (seg_threads), 0, st, jobs_dev, seg0, seg_job, job_chain, n_jobs, log2p, seg_doff, seg_cnt, seg_part, chain_map, info, D);
    HIPCHK (hipGetLastError());
  }
  W.info = info; W.D = D; W.Q = Q; W.pair_chunk0 = pair_chunk0; W.pair_bits = pair_bits; W.chunk_rec = chunk_rec; W.acc = acc;
  W.chunk_bound = chunk_bound; W.n_pairs = n_pairs; W.ready_chains = n_chains; W.ready_streams = streams_dev;
  W.last_words = n_words; W.last_q = n_q;
  if (W.leave (st)) return fail (LH264_E_HIP, "hipEventRecord (coder work memory)");
  return LH264_OK;
}

static int code_finish (CoderWs& W, const lh264_code_stream_t* streams_dev, int n_chains, hipStream_t st) {
  if (W.ready_chains != n_chains || W.ready_streams != streams_dev) return fail (LH264_E_ARG, "lh264_code_finish_chains without the matching lh264_code_binarise_chains");
  uint32_t* info = W.info; uint64_t* D = W.D; uint16_t* Q = W.Q; uint32_t* pair_chunk0 = W.pair_chunk0; uint32_t* pair_bits = W.pair_bits;
  uint32_t* chunk_rec = W.chunk_rec; uint32_t* acc = W.acc;
  const size_t chunk_bound = W.chunk_bound; const int n_pairs = W.n_pairs;
  W.ready_chains = -1;
  if (W.enter (st)) return fail (LH264_E_HIP, "hipStreamWaitEvent (coder work memory)");
  if (!W.sw && W.progress) HIPCHK (hipMemsetAsync (W.progress, 0, ((size_t)n_chains << W.log2p) * 4, st));
  if (W.sw) hipLaunchKernelGGL (lh264sw::coder_resolve_kernel, dim3 (n_chains), dim3 (LH264_CODER_RESOLVE_THREADS), 0, st, streams_dev, info, D, Q, n_chains);
  else hipLaunchKernelGGL (lh264::coder_resolve_kernel, dim3 ((unsigned) ((((size_t)n_chains + 7) / 8) * 8 * (((size_t)1 << W.log2p) >= LH264_CODER_WG_WAVES ? ((size_t)1 << W.log2p) / LH264_CODER_WG_WAVES : 1))), dim3 (64 * LH264_CODER_WG_WAVES), 0, st,
                           streams_dev, info, W.seg0, W.chain_first,
                           W.seg_doff, W.seg_part, D, Q, n_chains, W.log2p, W.progress, W.window);
  HIPCHK (hipGetLastError());
  hipLaunchKernelGGL (lh264::coder_status_kernel, dim3 ((n_chains + 255) / 256), dim3 (256), 0, st, streams_dev, info, n_chains);
  HIPCHK (hipGetLastError());
  hipLaunchKernelGGL (lh264::coder_chunkmap_kernel, dim3 (1), dim3 (CODER_ONE_WG), 0, st, info, n_pairs, pair_chunk0, W.pair_coarse0, W.cand_list);
  HIPCHK (hipGetLastError());
  // the bool coders' range recurrence in coarse chunks: start states by lookback, the walk, the running sum of the bits.
  // Lists up to long_list decisions are walked whole by one lane: with many moderate streams (the QCIF batch: 0.7 M decisions a stream)
  // every list is one of thousands and candidates are wasted walks; with large streams (5 - 23 M decisions) the lists of 65 - 262 k
  // decisions were the longest lanes of the launch (1080p batch 18.4 -> 14.5 ms, QCIF batch 1.2 -> 1.4 ms the other way)
  const uint32_t long_list = n_chains > 0 && W.last_q / (unsigned long long)n_chains > 4000000ull ? (uint32_t)LH264_CODER_CODE_COARSE : 262144u;
  hipLaunchKernelGGL (lh264::coder_range_seed_kernel, dim3 ((unsigned) ((W.coarse_bound + 3) / 4)), dim3 (256), 0, st, info, Q, W.pair_coarse0, n_pairs, W.cand, W.cand_list, long_list);
  HIPCHK (hipGetLastError());
  const int groups = (n_chains + 63) / 64;
  {
    const unsigned n_cand = (unsigned) ((W.coarse_bound + 7) / 8), n_later = (unsigned) ((W.coarse_bound + 63) / 64);
    hipLaunchKernelGGL (lh264::coder_range_first_kernel, dim3 (n_cand + (unsigned)groups * 35 + n_later + (unsigned)W.coarse_bound), dim3 (64), 0, st, info, Q, pair_chunk0, W.pair_coarse0,
                        n_pairs, groups, n_cand, n_later, W.cand, W.cand_list, W.cand_end, W.cmap, chunk_rec, W.coarse_bits);
    HIPCHK (hipGetLastError());
  }
  hipLaunchKernelGGL (lh264::coder_range_link_kernel, dim3 ((unsigned) ((n_pairs + 63) / 64)), dim3 (64), 0, st, W.pair_coarse0, n_pairs, W.cand, W.cand_end, W.cmap, W.seed, info);
  HIPCHK (hipGetLastError());
  hipLaunchKernelGGL (lh264::coder_range_walk_kernel, dim3 ((unsigned) ((W.coarse_bound + 63) / 64)), dim3 (64), 0, st, info, Q, pair_chunk0, W.pair_coarse0, n_pairs,
                      W.cand, W.seed, chunk_rec, W.coarse_bits);
  HIPCHK (hipGetLastError());
  hipLaunchKernelGGL (lh264::coder_range_scan_kernel, dim3 ((unsigned) ((n_pairs + 3) / 4)), dim3 (256), 0, st, W.pair_coarse0, n_pairs, W.coarse_bits, pair_bits, info, Q, acc);
  HIPCHK (hipGetLastError());
  hipLaunchKernelGGL (lh264::coder_accum_kernel, dim3 ((unsigned) ((chunk_bound + 255) / 256)), dim3 (256), 0, st, info, Q, pair_chunk0, W.pair_coarse0, n_pairs,
                      chunk_rec, W.coarse_bits, pair_bits, acc);
  HIPCHK (hipGetLastError());
  hipLaunchKernelGGL (lh264::coder_bytes_kernel, dim3 ((unsigned) ((n_pairs + 3) / 4)), dim3 (256), 0, st, streams_dev, info, Q, pair_bits, acc, n_pairs);
  HIPCHK (hipGetLastError());
  // tag slots 35 .. LH264_N_TAG_SLOTS-1 do not exist: their lengths read 0
  if (W.leave (st)) return fail (LH264_E_HIP, "hipEventRecord (coder work memory)");
  return LH264_OK;
}


static int coder_ws (CoderWs** W) {
  if (lh264_device_count() <= 0) return fail (LH264_E_NODEVICE, "no HIP device visible");
  int dev = 0;
  HIPCHK (hipGetDevice (&dev));
  if (dev < 0 || dev >= 16) return fail (LH264_E_ARG, "device index out of range");
  *W = &g_coder_ws[dev];
  return LH264_OK;
}
static bool code_args_ok (const void* jobs_dev, const void* chain_first_dev, const void* streams_dev, int n_chains, int n_jobs, long long total_mbs, int max_mbs_per_frame) {
  return jobs_dev && chain_first_dev && streams_dev && n_chains >= 0 && n_jobs >= 0 && total_mbs >= 0 && max_mbs_per_frame > 0;
}

int lh264_code_chains (const lh264_code_job_t* jobs_dev, const int32_t* chain_first_dev, const lh264_code_stream_t* streams_dev,
                       int n_chains, int n_jobs, long long total_mbs, int max_mbs_per_frame, void* stream) {
  CoderWs* W = nullptr;
  if (int rc = coder_ws (&W)) return rc;
  if (!code_args_ok (jobs_dev, chain_first_dev, streams_dev, n_chains, n_jobs, total_mbs, max_mbs_per_frame)) return fail (LH264_E_ARG, "bad argument");
  if (n_chains == 0) return LH264_OK;
  std::lock_guard<std::mutex> lock (W->mu);
  if (int rc = code_binarise (*W, jobs_dev, chain_first_dev, streams_dev, n_chains, n_jobs, total_mbs, (hipStream_t)stream)) return rc;
  return code_finish (*W, streams_dev, n_chains, (hipStream_t)stream);
}
int lh264_code_binarise_chains (const lh264_code_job_t* jobs_dev, const int32_t* chain_first_dev, const lh264_code_stream_t* streams_dev,
                                int n_chains, int n_jobs, long long total_mbs, int max_mbs_per_frame, void* stream) {
  CoderWs* W = nullptr;
  if (int rc = coder_ws (&W)) return rc;
  if (!code_args_ok (jobs_dev, chain_first_dev, streams_dev, n_chains, n_jobs, total_mbs, max_mbs_per_frame)) return fail (LH264_E_ARG, "bad argument");
  if (n_chains == 0) return LH264_OK;
  std::lock_guard<std::mutex> lock (W->mu);
  return code_binarise (*W, jobs_dev, chain_first_dev, streams_dev, n_chains, n_jobs, total_mbs, (hipStream_t)stream);
}
int lh264_code_finish_chains (const lh264_code_stream_t* streams_dev, int n_chains, void* stream) {
  CoderWs* W = nullptr;
  if (int rc = coder_ws (&W)) return rc;
  if (!streams_dev || n_chains < 0) return fail (LH264_E_ARG, "bad argument");
  if (n_chains == 0) return LH264_OK;
  std::lock_guard<std::mutex> lock (W->mu);
  return code_finish (*W, streams_dev, n_chains, (hipStream_t)stream);
}

int lh264_code_last_totals (unsigned long long* decision_words, unsigned long long* list_entries) {
  int dev = 0;
  if (lh264_device_count() <= 0 || hipGetDevice (&dev) != hipSuccess || dev < 0 || dev >= 16) return fail (LH264_E_NODEVICE, "no HIP device visible");
  CoderWs& W = g_coder_ws[dev];
  std::lock_guard<std::mutex> lock (W.mu);
  if (decision_words) *decision_words = W.last_words;
  if (list_entries) *list_entries = W.last_q;
  return LH264_OK;
}

double lh264_time_recon_chains (const lh264_frame_job_t* jobs_dev, const int32_t* chain_first_dev, int n_chains,
                                int max_mb_w, int max_mb_h, int iters, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  hipEvent_t a, b;
  if (hipEventCreate (&a) != hipSuccess || hipEventCreate (&b) != hipSuccess) { fail (LH264_E_HIP, "hipEventCreate"); return -1.0; }
  if (launch_chains (jobs_dev, chain_first_dev, n_chains, max_mb_w, max_mb_h, st) != LH264_OK) return -1.0;   // warm-up
  hipEventRecord (a, st);
  for (int i = 0; i < iters; i++)
    if (launch_chains (jobs_dev, chain_first_dev, n_chains, max_mb_w, max_mb_h, st) != LH264_OK) return -1.0;
  hipEventRecord (b, st);
  if (hipEventSynchronize (b) != hipSuccess) { fail (LH264_E_HIP, "hipEventSynchronize"); return -1.0; }
  float ms = 0.f;
  hipEventElapsedTime (&ms, a, b);
  hipEventDestroy (a); hipEventDestroy (b);
  return iters > 0 ? (double)ms / iters : 0.0;
}

// ---- host front end ------------------------------------------------------------------------------------------------
lh264_parser_t* lh264_parser_create (void) { return new lh264_parser(); }
void lh264_parser_destroy (lh264_parser_t* p) { delete p; }
int lh264_parser_feed (lh264_parser_t* p, const uint8_t* data, size_t len, int flush) {
  if (!p) return LH264_E_ARG;
  int rc = (data && len) ? p->p.feed (data, len) : 0;
  if (flush) p->p.flush();
  return rc < 0 ? LH264_E_UNSUPPORTED : LH264_OK;
}
static thread_local std::string g_restore_err;
int lh264_pip_restore (const uint8_t* main_stream, size_t main_len, const uint8_t* const* tags, const size_t* tag_len, int n_tags,
                       uint8_t* out, size_t out_cap, size_t* out_len) {
  if (!main_stream || !tags || !tag_len || !out_len || n_tags < 0) return LH264_E_ARG;
  std::vector<uint8_t> o;
  if (lh264host::pip_restore (main_stream, main_len, tags, tag_len, n_tags, o, g_restore_err) < 0) return LH264_E_UNSUPPORTED;
  *out_len = o.size();
  if (o.size() > out_cap || (!out && o.size())) { g_restore_err = "output buffer too small"; return LH264_E_ARG; }
  if (o.size()) memcpy (out, o.data(), o.size());
  return LH264_OK;
}
const char* lh264_restore_error (void) { return g_restore_err.c_str(); }
static void put32 (uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t) (v >> 8); p[2] = (uint8_t) (v >> 16); p[3] = (uint8_t) (v >> 24); }
static uint32_t get32 (const uint8_t* p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
static const char kPipMagic[8] = {'L', 'H', 'P', 'I', 'P', '1', 0, 0};
size_t lh264_pip_pack_bound (size_t main_len, const size_t* tag_len, int n_tags) {
  size_t n = 16 + 8 + main_len;
  for (int t = 0; tag_len && t < n_tags; t++) n += 8 + tag_len[t];
  return n;
}
int lh264_pip_pack (const uint8_t* main_stream, size_t main_len, const uint8_t* const* tags, const size_t* tag_len, int n_tags,
                    uint32_t flags, uint8_t* out, size_t out_cap, size_t* out_len) {
  if ((!main_stream && main_len) || !out_len || n_tags < 0 || (n_tags && (!tags || !tag_len))) return LH264_E_ARG;
  int n = 1;
  size_t total = main_len;
  for (int t = 0; t < n_tags; t++) if (tags[t]) { n++; total += tag_len[t]; if (tag_len[t] > 0xffffffffull) return LH264_E_ARG; }
  if (main_len > 0xffffffffull) return LH264_E_ARG;
  const size_t need = 16 + 8 * (size_t)n + total;
  *out_len = need;
  if (!out || out_cap < need) return LH264_E_ARG;
  memcpy (out, kPipMagic, 8); put32 (out + 8, flags); put32 (out + 12, (uint32_t)n);
  uint8_t* h = out + 16; uint8_t* d = out + 16 + 8 * (size_t)n;
  put32 (h, 0x7fffffffu); put32 (h + 4, (uint32_t)main_len); h += 8;
  if (main_len) memcpy (d, main_stream, main_len);
  d += main_len;
  for (int t = 0; t < n_tags; t++) if (tags[t]) {
      put32 (h, (uint32_t)t); put32 (h + 4, (uint32_t)tag_len[t]); h += 8;
      if (tag_len[t]) memcpy (d, tags[t], tag_len[t]);
      d += tag_len[t];
    }
  return LH264_OK;
}
int lh264_pip_restore_file (const uint8_t* file, size_t len, uint8_t* out, size_t out_cap, size_t* out_len) {
  if (!file || !out_len) return LH264_E_ARG;
  if (len < 16 || memcmp (file, kPipMagic, 8) != 0) { g_restore_err = "not a LHPIP1 container"; return LH264_E_UNSUPPORTED; }
  const uint32_t flags = get32 (file + 8), n = get32 (file + 12);
  if (n < 1 || (size_t)n > (len - 16) / 8) { g_restore_err = "corrupt container"; return LH264_E_UNSUPPORTED; }
  const uint8_t* d = file + 16 + 8 * (size_t)n;
  size_t left = len - 16 - 8 * (size_t)n;
  const uint8_t* main_stream = nullptr; size_t main_len = 0;
  const uint8_t* tags[72]; size_t tag_len[72];
  for (int t = 0; t < 72; t++) { tags[t] = nullptr; tag_len[t] = 0; }
  for (uint32_t i = 0; i < n; i++) {
    const uint32_t id = get32 (file + 16 + 8 * (size_t)i), ln = get32 (file + 20 + 8 * (size_t)i);
    if (ln > left) { g_restore_err = "corrupt container"; return LH264_E_UNSUPPORTED; }
    if (id == 0x7fffffffu) { main_stream = d; main_len = ln; }
    else if (id < 72) { tags[id] = ln ? d : (const uint8_t*)""; tag_len[id] = ln; }
    d += ln; left -= ln;
  }
  if (!main_stream && main_len == 0 && !(flags & LH264_PIP_VERBATIM)) main_stream = (const uint8_t*)"";
  if (flags & LH264_PIP_VERBATIM) {
    *out_len = main_len;
    if (main_len > out_cap || (!out && main_len)) { g_restore_err = "output buffer too small"; return LH264_E_ARG; }
    if (main_len) memcpy (out, main_stream, main_len);
    return LH264_OK;
  }
  return lh264_pip_restore (main_stream, main_len, tags, tag_len, 72, out, out_cap, out_len);
}
int lh264_parse_batch (const uint8_t* const* data, const size_t* len, int n, int threads, lh264_parser_t** parsers_out) {
  if (!data || !len || !parsers_out || n < 0) return LH264_E_ARG;
  for (int i = 0; i < n; i++) { parsers_out[i] = new lh264_parser(); parsers_out[i]->p.set_stream_arena (true); }
  run_parallel (n, threads, [&] (int i) { if (data[i] || !len[i]) parsers_out[i]->p.feed_file (data[i], len[i]); });
  return LH264_OK;
}
int lh264_parse_batch_discard (const uint8_t* const* data, const size_t* len, int n, int threads, int64_t* pictures_out) {
  if (!data || !len || !pictures_out || n < 0) return LH264_E_ARG;
  run_parallel (n, threads, [&] (int i) {
    lh264host::Parser p;
    p.set_keep_frames (false);
    if (data[i] || !len[i]) p.feed_file (data[i], len[i]);
    pictures_out[i] = p.pictures_done();
  });
  return LH264_OK;
}
int lh264_pip_restore_batch (lh264_restore_item_t* items, int n, int threads) {
  if (!items || n < 0) return LH264_E_ARG;
  run_parallel (n, threads, [&] (int i) {
    lh264_restore_item_t& it = items[i];
    it.status = lh264_pip_restore (it.main_stream, it.main_len, it.tags, it.tag_len, it.n_tags, it.out, it.out_cap, &it.out_len);
  });
  return LH264_OK;
}
int lh264_parser_feed_file (lh264_parser_t* p, const uint8_t* data, size_t len) {
  if (!p || (!data && len)) return LH264_E_ARG;
  return p->p.feed_file (data, len) < 0 ? LH264_E_UNSUPPORTED : LH264_OK;
}
const uint8_t* lh264_parser_main_stream (const lh264_parser_t* p, size_t* len) {
  if (!p) return nullptr;
  if (len) *len = p->p.main_stream().size();
  return p->p.main_stream().data();
}
const uint8_t* lh264_parser_pcm_samples (const lh264_parser_t* p, size_t* len) {
  if (!p) return nullptr;
  if (len) *len = p->p.pcm_samples().size();
  return p->p.pcm_samples().data();
}
int lh264_parser_frame_count (const lh264_parser_t* p) { return p ? (int)const_cast<lh264_parser_t*> (p)->p.frames().size() : 0; }
static const lh264host::FrameOut* pf (const lh264_parser_t* p, int idx) {
  if (!p) return nullptr;
  auto& f = const_cast<lh264_parser_t*> (p)->p.frames();
  return (idx >= 0 && idx < (int)f.size()) ? f[idx].get() : nullptr;
}
int lh264_parser_frame_info (const lh264_parser_t* p, int idx, lh264_frame_info_t* o) {
  const lh264host::FrameOut* f = pf (p, idx);
  if (!f || !o) return LH264_E_ARG;
  o->id = f->id; o->mb_w = f->mb_w; o->mb_h = f->mb_h; o->n_slices = (int)f->slices.size(); o->n_refs = (int)f->ref_ids.size();
  o->frame_num = f->frame_num; o->crop_x = f->crop_x; o->crop_y = f->crop_y; o->crop_w = f->crop_w; o->crop_h = f->crop_h;
  o->is_ref = f->is_ref; o->idr = f->idr;
  for (int i = 0; i < LH264_MAX_REFS; i++) o->ref_ids[i] = i < (int)f->ref_ids.size() ? f->ref_ids[i] : -1;
  return LH264_OK;
}
const lh264_mb_t* lh264_parser_frame_mbs (const lh264_parser_t* p, int idx) { auto f = pf (p, idx); return f ? f->mbs.data() : nullptr; }
const int16_t* lh264_parser_frame_coeffs (const lh264_parser_t* p, int idx) { auto f = pf (p, idx); return f ? f->coeffs.data() : nullptr; }
const int16_t* lh264_parser_frame_levels (const lh264_parser_t* p, int idx) { auto f = pf (p, idx); return f ? f->levels.data() : nullptr; }
const lh264_slice_t* lh264_parser_frame_slices (const lh264_parser_t* p, int idx) { auto f = pf (p, idx); return f ? f->slices.data() : nullptr; }
const uint8_t* lh264_parser_frame_covered (const lh264_parser_t* p, int idx) { auto f = pf (p, idx); return f ? f->covered.data() : nullptr; }
static_assert (sizeof (lh264_mbsyn_t) == sizeof (lh264host::MbSyn), "lh264_mbsyn_t layout");
const lh264_mbsyn_t* lh264_parser_frame_syntax (const lh264_parser_t* p, int idx) { auto f = pf (p, idx); return f ? (const lh264_mbsyn_t*)f->syn.data() : nullptr; }
const int32_t* lh264_parser_frame_slice_syntax (const lh264_parser_t* p, int idx) { auto f = pf (p, idx); return f ? (const int32_t*)f->slice_syn.data() : nullptr; }
const lh264_ctx_sym_t* lh264_parser_frame_syn_symbols (const lh264_parser_t* p, int idx, int* count) {
  auto f = pf (p, idx);
  if (count) *count = f ? (int)f->syn_syms.size() : 0;
  return f ? f->syn_syms.data() : nullptr;
}
const uint32_t* lh264_parser_frame_syn_offsets (const lh264_parser_t* p, int idx) { auto f = pf (p, idx); return f ? f->syn_off.data() : nullptr; }
const char* lh264_parser_error (const lh264_parser_t* p) { return p ? const_cast<lh264_parser_t*> (p)->p.error().c_str() : ""; }

#ifdef LH264_CODER_DEBUG
void lh264_debug_read_rs_stamps (unsigned long long* out16, int reset) { lh264::read_rs_stamps (out16, reset != 0); }
#endif
#ifdef LH264_RANGE_PROBE
// diagnostic builds only: the seed words of the last coder call on the current device
long long lh264_debug_coder_seeds (uint32_t* out, long long cap) {
  int dev = 0; (void)hipGetDevice (&dev);
  CoderWs& W = g_coder_ws[dev];
  uint32_t total = 0;                   // coarse chunks of the call: the probe words lie behind the seeds
  if (hipMemcpy (&total, W.pair_coarse0 + W.n_pairs_last, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  const long long n = (long long)total < cap ? (long long)total : cap;
  if (hipMemcpy (out, W.cand + 2 * (size_t)total, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return n;
}
#endif
// tuning aid (not declared in lh264.h): decisions per partition of stream `chain` of the last wave-form coder call on the current device;
// returns the number of partitions, -1 if there is nothing to read
int lh264_debug_coder_parts (int chain, unsigned long long* out, int cap) {
  int dev = 0; (void)hipGetDevice (&dev);
  CoderWs& W = g_coder_ws[dev];
  if (W.sw || !W.seg_part || !W.seg0 || !W.chain_first) return -1;
  (void)hipDeviceSynchronize();
  int32_t cf[2]; uint32_t sg[2];
  if (hipMemcpy (cf, W.chain_first + chain, 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  if (hipMemcpy (&sg[0], W.seg0 + cf[0], 4, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy (&sg[1], W.seg0 + cf[1], 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  const int P = 1 << W.log2p;
  if (P > cap) return -1;
  std::vector<uint32_t> rows ((size_t) (sg[1] - sg[0]) * (size_t) (P + 1));
  if (hipMemcpy (rows.data(), W.seg_part + (size_t)sg[0] * (size_t) (P + 1), rows.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  for (int p = 0; p < P; p++) out[p] = 0;
  for (size_t g = 0; g < sg[1] - sg[0]; g++) for (int p = 0; p < P; p++) out[p] += rows[g * (P + 1) + p + 1] - rows[g * (P + 1) + p];
  return P;
}
#ifdef LH264_STAMP
// diagnostic builds only (not declared in lh264.h)
void lh264_debug_read_stamps (unsigned long long* out16, int reset) { lh264::read_stamps (out16, reset != 0); }
#endif

}  // extern "C"
