// lh264_compress.hip - the compress direction behind one C call (include/lh264.h: lh264_compress_batch): the host
// orchestration the reference does inside its decoder loop (decode_slice.cpp:3085-3112 per macroblock, flushToWriter at
// the end), here per batch of independent streams: parse on host threads, stage records and symbol lists in HBM, one
// launch of the context-index kernels and one of the coder kernel, copy the tagged streams back.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <memory>
#include <mutex>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <thread>
#include <string>
#include <vector>
#include "../../include/lh264.h"
#include "host/h264_parser.h"
#include "host/capi_internal.h"

struct lh264_compressed {
  int status = LH264_OK;
  std::string error;
  std::vector<uint8_t> main_stream;
  std::vector<uint8_t> tag[72];
  bool has_tag[72] = {false};
  int pictures = 0;
};

// gathers the tagged streams of a group (one workgroup per stream and tag) into one buffer: one download instead of thousands
struct PackItem { uint64_t src, dst; uint32_t len, pad; };
__global__ void __launch_bounds__ (256) pack_tags_kernel (const PackItem* __restrict__ items, const uint8_t* __restrict__ out, uint8_t* __restrict__ packed) {
  const PackItem it = items[blockIdx.x];
  const uint8_t* s = out + it.src; uint8_t* d = packed + it.dst;
  for (uint32_t i = threadIdx.x; i < it.len; i += blockDim.x) d[i] = s[i];
}

// the raw levels travel as a list of the nonzero ones: (index into the group's level planes) << 16 | level
__global__ void __launch_bounds__ (256) expand_levels_kernel (const uint64_t* __restrict__ ents, size_t n, int16_t* __restrict__ dense) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { const uint64_t e = ents[i]; dense[e >> 16] = (int16_t) (uint16_t) (e & 0xffffu); }
}

namespace {

static bool trace_on() { static const bool t = getenv ("LH264_TRACE_COMPRESS") != nullptr; return t; }
static double now_s() { return std::chrono::duration<double> (std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct DevBuf {                       // device memory, grown when a group needs more, reused otherwise
  void* p = nullptr; size_t cap = 0;
  ~DevBuf() { if (p) hipFree (p); }
  bool alloc (size_t bytes, bool zero) {
    if (bytes < 16) bytes = 16;
    if (bytes > cap) {
      if (p) { hipFree (p); p = nullptr; cap = 0; }
      const size_t want = bytes + bytes / 8;
      if (hipMalloc (&p, want) != hipSuccess) { p = nullptr; return false; }
      cap = want;
    }
    if (zero && hipMemsetAsync (p, 0, bytes, nullptr) != hipSuccess) return false;
    return true;
  }
  template <typename T> T* as() const { return (T*)p; }
};
struct PinBuf {                       // page-locked staging memory (the upload runs at PCIe speed and asynchronously)
  void* p = nullptr; size_t cap = 0;
  ~PinBuf() { if (p) hipHostFree (p); }
  bool alloc (size_t bytes) {
    if (bytes < 16) bytes = 16;
    if (bytes <= cap) return true;
    if (p) { hipHostFree (p); p = nullptr; cap = 0; }
    const size_t want = bytes + bytes / 8;
    if (hipHostMalloc (&p, want, hipHostMallocDefault) != hipSuccess) { p = nullptr; return false; }
    cap = want;
    return true;
  }
  template <typename T> T* as() const { return (T*)p; }
};
struct Arena {
  DevBuf d_mbs, d_lev, d_sl, d_nnz, d_syms, d_nsyms, d_symoff, d_symbase, d_cj, d_first, d_syn, d_off, d_kj, d_st, d_keys, d_cells, d_out, d_len, d_items, d_packed;
  PinBuf h_mbs, h_sparse, h_sl, h_syn, h_off, h_packed;
  DevBuf d_sparse;
};

// which earlier picture the reference's FreqImage holds as PAST (decoded_macroblock.h:119-123): two buffers, flipped when
// frame_num changes; -1 = none
void past_policy (const std::vector<std::unique_ptr<lh264host::FrameOut>>& fr, std::vector<int>& past) {
  int cur = 0, last_fn = 0, slot[2] = {-1, -1};
  past.resize (fr.size());
  for (size_t i = 0; i < fr.size(); i++) {
    if (fr[i]->frame_num != last_fn) { cur ^= 1; last_fn = fr[i]->frame_num; }
    past[i] = slot[1 - cur];
    slot[cur] = (int)i;
  }
}

void fail_all (lh264_compressed_t** out, const std::vector<int>& idx, int code, const char* what) {
  for (int i : idx) { out[i]->status = code; out[i]->error = what; }
}

// one sub-batch: streams idx[0..] of `parsers`, all parsed without error
void compress_group (Arena& A, std::vector<std::unique_ptr<lh264host::Parser>>& parsers, const std::vector<int>& idx, const size_t* len,
                     lh264_compressed_t** out, int threads) {
  using lh264host::FrameOut;
  const int n_chains = (int)idx.size();
  // where every stream's records go
  std::vector<size_t> mb0 (n_chains + 1, 0), sl0 (n_chains + 1, 0), sy0 (n_chains + 1, 0), of0 (n_chains + 1, 0), jb0 (n_chains + 1, 0), sp0 (n_chains + 1, 0);
  int max_mbs = 1;
  for (int c = 0; c < n_chains; c++) {
    size_t m = 0, sl = 0, sy = 0, of = 0, jb = 0, sp = 0;
    for (auto& f : parsers[idx[c]]->frames()) {
      const size_t n = (size_t)f->mb_w * f->mb_h;
      m += n; sl += f->slices.size(); sy += f->syn_syms.size(); of += n + 1; jb++; sp += f->sparse.size();
      max_mbs = std::max (max_mbs, (int)n);
    }
    sp0[c + 1] = sp0[c] + sp;
    mb0[c + 1] = mb0[c] + m; sl0[c + 1] = sl0[c] + sl; sy0[c + 1] = sy0[c] + sy; of0[c + 1] = of0[c] + of; jb0[c + 1] = jb0[c] + jb;
  }
  const size_t n_sparse = sp0[n_chains];
  const size_t n_mbs = mb0[n_chains], n_slices = sl0[n_chains], n_syn = sy0[n_chains], n_off = of0[n_chains], n_jobs = jb0[n_chains];
  if (n_jobs == 0) return;
  std::vector<lh264_ctx_job_t> h_cj (n_jobs);
  std::vector<lh264_code_job_t> h_kj (n_jobs);
  std::vector<int32_t> h_first (n_chains + 1);
  std::vector<lh264_code_stream_t> h_st (n_chains);
  std::vector<uint32_t> hash_cap (n_chains), out_cap (n_chains);
  std::vector<size_t> key0 (n_chains + 1, 0), out0 (n_chains + 1, 0);
  for (int c = 0; c < n_chains; c++) {
    const size_t mbs = mb0[c + 1] - mb0[c];
    uint32_t hc = 1u << 13;                                   // 8 spill entries per cell: four entries per input byte (a stream touches
    while ((size_t)hc * 2 < len[idx[c]] && hc < (1u << 20)) hc <<= 1;      // 0.2 .. 0.5 adaptive probabilities per byte); status 1 reports a full table
    hash_cap[c] = hc; key0[c + 1] = key0[c] + hc;
    out_cap[c] = (uint32_t)std::max<size_t> (1u << 16, 2 * len[idx[c]] + 4096);
    out0[c + 1] = out0[c] + (size_t)LH264_N_TAG_SLOTS * out_cap[c];
  }
  const size_t keys_total = key0[n_chains], out_total = out0[n_chains];
  const double t_a = now_s();
  const bool ok = A.d_mbs.alloc (n_mbs * sizeof (lh264_mb_t), false) && A.d_lev.alloc (n_mbs * 768, true) && A.d_sparse.alloc (n_sparse * 8, false) && A.d_sl.alloc (n_slices * sizeof (lh264_slice_t), false) &&
                  A.d_nnz.alloc (n_mbs * 24, true) && A.d_nsyms.alloc (n_mbs * 2, true) && A.d_symoff.alloc (n_mbs * 4, false) && A.d_symbase.alloc ((n_jobs + 2) * 8, false) &&
                  A.d_cj.alloc (n_jobs * sizeof (lh264_ctx_job_t), false) && A.d_first.alloc ((n_chains + 1) * 4, false) && A.d_syn.alloc (n_syn * sizeof (lh264_ctx_sym_t), false) &&
                  A.d_off.alloc (n_off * 4, false) && A.d_kj.alloc (n_jobs * sizeof (lh264_code_job_t), false) && A.d_st.alloc (n_chains * sizeof (lh264_code_stream_t), false) &&
                  A.d_keys.alloc (256, false) && A.d_cells.alloc (keys_total * 64, true) && A.d_out.alloc (out_total, false) &&
                  A.d_len.alloc ((size_t)n_chains * (LH264_N_TAG_SLOTS + 1) * 4, true) &&
                  A.h_mbs.alloc (n_mbs * sizeof (lh264_mb_t)) && A.h_sparse.alloc (n_sparse * 8) && A.h_sl.alloc (n_slices * sizeof (lh264_slice_t)) &&
                  A.h_syn.alloc (n_syn * sizeof (lh264_ctx_sym_t)) && A.h_off.alloc (n_off * 4);
  if (!ok) { fail_all (out, idx, LH264_E_HIP, "device allocation failed"); return; }
  const double t_b = now_s();
  lh264_mb_t* h_mbs = A.h_mbs.as<lh264_mb_t>(); uint64_t* h_sparse = A.h_sparse.as<uint64_t>(); lh264_slice_t* h_sl = A.h_sl.as<lh264_slice_t>();
  lh264_ctx_sym_t* h_syn = A.h_syn.as<lh264_ctx_sym_t>(); uint32_t* h_off = A.h_off.as<uint32_t>();
  // staging: every stream copies its pictures to its place (host threads), and writes its job records
  run_parallel (n_chains, threads, [&] (int c) {
    auto& fr = parsers[idx[c]]->frames();
    size_t mo = mb0[c], so = sl0[c], yo = sy0[c], oo = of0[c], j = jb0[c], po = sp0[c];
    h_first[c] = (int32_t)j;
    std::vector<int> past;
    past_policy (fr, past);
    std::vector<size_t> mb_at (fr.size());
    for (size_t i = 0; i < fr.size(); i++) {
      FrameOut& f = *fr[i];
      const size_t n = (size_t)f.mb_w * f.mb_h;
      mb_at[i] = mo;
      memcpy (&h_mbs[mo], f.mbs.data(), n * sizeof (lh264_mb_t));
      { const uint64_t add = (uint64_t) (mo * 384) << 16; const size_t ns = f.sparse.size(); for (size_t q = 0; q < ns; q++) h_sparse[po + q] = f.sparse[q] + add; po += ns; }
      if (!f.slices.empty()) memcpy (&h_sl[so], f.slices.data(), f.slices.size() * sizeof (lh264_slice_t));
      if (!f.syn_syms.empty()) memcpy (&h_syn[yo], f.syn_syms.data(), f.syn_syms.size() * sizeof (lh264_ctx_sym_t));
      memcpy (&h_off[oo], f.syn_off.data(), (n + 1) * 4);
      lh264_ctx_job_t& cj = h_cj[j];
      cj.mbs_dev = A.d_mbs.as<lh264_mb_t>() + mo; cj.levels_dev = A.d_lev.as<int16_t>() + mo * 384; cj.slices_dev = A.d_sl.as<lh264_slice_t>() + so;
      cj.nnz_cur_dev = A.d_nnz.as<uint8_t>() + mo * 24;
      cj.nnz_past_dev = past[i] < 0 ? nullptr : A.d_nnz.as<uint8_t>() + mb_at[past[i]] * 24;
      // (the compact layout: the pool's address and size are set once the count pass has said how many symbols there are)
      cj.syms_dev = nullptr; cj.syms_cap = 0; cj.n_syms_dev = A.d_nsyms.as<uint16_t>() + mo;
      cj.sym_off_dev = A.d_symoff.as<uint32_t>() + mo; cj.sym_base_dev = A.d_symbase.as<uint64_t>() + j;
      cj.mb_w = f.mb_w; cj.mb_h = f.mb_h;
      lh264_code_job_t& kj = h_kj[j];
      kj.syn_syms_dev = A.d_syn.as<lh264_ctx_sym_t>() + yo; kj.syn_off_dev = A.d_off.as<uint32_t>() + oo;
      kj.ctx_syms_dev = nullptr; kj.ctx_n_syms_dev = cj.n_syms_dev; kj.n_mbs = (int32_t)n; kj.reserved = 0;
      kj.ctx_sym_off_dev = cj.sym_off_dev; kj.ctx_sym_base_dev = cj.sym_base_dev;
      mo += n; so += f.slices.size(); yo += f.syn_syms.size(); oo += n + 1; j++;
    }
    lh264_code_stream_t& st = h_st[c];
    st.hash_keys_dev = A.d_keys.as<uint32_t>(); st.hash_cells_dev = A.d_cells.as<uint32_t>() + key0[c] * 16;
    st.out_dev = A.d_out.as<uint8_t>() + out0[c]; st.out_len_dev = A.d_len.as<uint32_t>() + (size_t)c * (LH264_N_TAG_SLOTS + 1);
    st.hash_cap = hash_cap[c]; st.out_cap = out_cap[c];
  });
  h_first[n_chains] = (int32_t)n_jobs;
  const double t_c = now_s();
  auto up = [] (DevBuf& d, const void* s, size_t bytes) { return bytes == 0 || hipMemcpyAsync (d.p, s, bytes, hipMemcpyHostToDevice, nullptr) == hipSuccess; };
  if (!(up (A.d_mbs, h_mbs, n_mbs * sizeof (lh264_mb_t)) && up (A.d_sparse, h_sparse, n_sparse * 8) && up (A.d_sl, h_sl, n_slices * sizeof (lh264_slice_t)) &&
        up (A.d_syn, h_syn, n_syn * sizeof (lh264_ctx_sym_t)) && up (A.d_off, h_off, n_off * 4) && up (A.d_cj, h_cj.data(), n_jobs * sizeof (lh264_ctx_job_t)) &&
        up (A.d_kj, h_kj.data(), n_jobs * sizeof (lh264_code_job_t)) && up (A.d_first, h_first.data(), (n_chains + 1) * 4) && up (A.d_st, h_st.data(), n_chains * sizeof (lh264_code_stream_t)))) {
    fail_all (out, idx, LH264_E_HIP, "upload failed"); return;
  }
  if (n_sparse) hipLaunchKernelGGL (expand_levels_kernel, dim3 ((unsigned) ((n_sparse + 255) / 256)), dim3 (256), 0, nullptr, A.d_sparse.as<uint64_t>(), n_sparse, A.d_lev.as<int16_t>());
  if (trace_on()) hipDeviceSynchronize();
  const double t_d = now_s();
  // the symbol pool: the count pass says how many symbols the group's pictures have (8 bytes each; the fixed layout took 3,456 bytes per
  // macroblock), then the job tables get the pool's address
  unsigned long long n_syms_total = 0;
  int rc = lh264_ctx_count_chains (A.d_cj.as<lh264_ctx_job_t>(), A.d_first.as<int32_t>(), n_chains, (int)n_jobs, max_mbs, A.d_symbase.as<unsigned long long>() + n_jobs + 1, nullptr);
  if (rc == LH264_OK && hipMemcpy (&n_syms_total, A.d_symbase.as<unsigned long long>() + n_jobs + 1, 8, hipMemcpyDeviceToHost) != hipSuccess) rc = LH264_E_HIP;
  if (rc == LH264_OK && !A.d_syms.alloc ((size_t)n_syms_total * sizeof (lh264_ctx_sym_t), false)) rc = LH264_E_HIP;
  if (rc == LH264_OK) {
    for (size_t j = 0; j < n_jobs; j++) { h_cj[j].syms_dev = A.d_syms.as<lh264_ctx_sym_t>(); h_cj[j].syms_cap = n_syms_total; h_kj[j].ctx_syms_dev = A.d_syms.as<lh264_ctx_sym_t>(); }
    if (hipMemcpy (A.d_cj.p, h_cj.data(), n_jobs * sizeof (lh264_ctx_job_t), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy (A.d_kj.p, h_kj.data(), n_jobs * sizeof (lh264_code_job_t), hipMemcpyHostToDevice) != hipSuccess) rc = LH264_E_HIP;
  }
  if (rc == LH264_OK) rc = lh264_ctx_index_chains (A.d_cj.as<lh264_ctx_job_t>(), A.d_first.as<int32_t>(), n_chains, (int)n_jobs, max_mbs, nullptr);
  if (rc == LH264_OK) rc = lh264_code_chains (A.d_kj.as<lh264_code_job_t>(), A.d_first.as<int32_t>(), A.d_st.as<lh264_code_stream_t>(), n_chains, (int)n_jobs, (long long)n_mbs, max_mbs, nullptr);
  if (rc != LH264_OK || hipDeviceSynchronize() != hipSuccess) { fail_all (out, idx, rc != LH264_OK ? rc : LH264_E_HIP, "kernel launch failed"); return; }
  const double t_e = now_s();
  std::vector<uint32_t> lens ((size_t)n_chains * (LH264_N_TAG_SLOTS + 1));
  if (hipMemcpy (lens.data(), A.d_len.p, lens.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { fail_all (out, idx, LH264_E_HIP, "download failed"); return; }
  std::vector<PackItem> items;
  size_t packed_bytes = 0;
  for (int c = 0; c < n_chains; c++) {
    lh264_compressed_t& r = *out[idx[c]];
    const uint32_t* L = &lens[(size_t)c * (LH264_N_TAG_SLOTS + 1)];
    if (L[LH264_N_TAG_SLOTS] != 0) { r.status = LH264_E_HIP; r.error = "device coder status " + std::to_string (L[LH264_N_TAG_SLOTS]) + " (bits - 1: prior table full or invalid, 4: output overflow, 8: counter overflow, 16: internal hand-off; include/lh264.h)"; continue; }
    for (int slot = 0; slot < 35; slot++) if (L[slot]) {
        PackItem it; it.src = out0[c] + (size_t)slot * out_cap[c]; it.dst = packed_bytes; it.len = L[slot]; it.pad = (uint32_t)c << 8 | (uint32_t)slot;
        items.push_back (it);
        packed_bytes += (L[slot] + 15u) & ~15u;
      }
  }
  if (!items.empty()) {
    if (!(A.d_items.alloc (items.size() * sizeof (PackItem), false) && A.d_packed.alloc (packed_bytes, false) && A.h_packed.alloc (packed_bytes)) ||
        hipMemcpyAsync (A.d_items.p, items.data(), items.size() * sizeof (PackItem), hipMemcpyHostToDevice, nullptr) != hipSuccess) { fail_all (out, idx, LH264_E_HIP, "download failed"); return; }
    hipLaunchKernelGGL (pack_tags_kernel, dim3 ((unsigned)items.size()), dim3 (256), 0, nullptr, A.d_items.as<PackItem>(), A.d_out.as<uint8_t>(), A.d_packed.as<uint8_t>());
    if (hipMemcpy (A.h_packed.p, A.d_packed.p, packed_bytes, hipMemcpyDeviceToHost) != hipSuccess) { fail_all (out, idx, LH264_E_HIP, "download failed"); return; }
    const uint8_t* hp = A.h_packed.as<uint8_t>();
    run_parallel ((int)items.size(), threads, [&] (int k) {
      const PackItem& it = items[k];
      lh264_compressed_t& r = *out[idx[it.pad >> 8]];
      const int slot = (int) (it.pad & 0xff), tag = slot == 34 ? 69 : slot;
      r.tag[tag].assign (hp + it.dst, hp + it.dst + it.len); r.has_tag[tag] = true;
    });
  }
  if (trace_on()) fprintf (stderr, "[lh264 compress] group of %d streams, %zu MBs: alloc+clear %.3f s, staging %.3f, upload %.3f, kernels %.3f, download %.3f\n", n_chains, n_mbs,
                           t_b - t_a, t_c - t_b, t_d - t_c, t_e - t_d, now_s() - t_e);
}

enum { kMaxDevices = 16 };
std::unique_ptr<Arena> g_arena[kMaxDevices];      // one per device: device and page-locked buffers kept between calls
std::mutex g_arena_mutex[kMaxDevices];             // one compress call at a time per device

}  // namespace

extern "C" {

int lh264_compress_batch (const uint8_t* const* data, const size_t* len, int n, int threads, lh264_compressed_t** out) {
  if (!data || !len || !out || n < 0) return LH264_E_ARG;
  for (int i = 0; i < n; i++) out[i] = new lh264_compressed();
  if (lh264_device_count() <= 0) { for (int i = 0; i < n; i++) { out[i]->status = LH264_E_NODEVICE; out[i]->error = "no HIP device visible"; } return LH264_E_NODEVICE; }
  if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
  if (threads < 1) threads = 1;
  int device = 0;
  hipGetDevice (&device);
  // Streams are parsed in waves on the host threads; parsed streams collect into a group until the group is worth a launch
  // (bounded by macroblock count: the symbol buffer takes 3.4 KB per macroblock); the group's staging, upload, kernels and
  // download run on their own host thread while the next wave is being parsed.
  // the device stage of a group is bound by the per-stream serial chains of the coder (about 15 ms for 100 QCIF pictures, whatever the
  // number of streams above a few hundred), its parse by the host threads (0.2 us per macroblock and thread): groups of this size keep
  // both sides busy, so that a batch of a few hundred streams already overlaps parsing with the device stage
  const size_t kBudget = 1300000;
  const int kWave = std::max (8, 4 * threads);
  // device and page-locked buffers live across calls (allocating and releasing ~20 GB costs more than a whole batch):
  // one arena per process, one compress call at a time; lh264_compress_release() gives the memory back
  if (device < 0 || device >= kMaxDevices) return LH264_E_ARG;
  std::lock_guard<std::mutex> arena_lock (g_arena_mutex[device]);
  if (!g_arena[device]) g_arena[device].reset (new Arena());
  Arena& arena = *g_arena[device];
  const double t_call = now_s();
  std::vector<std::unique_ptr<lh264host::Parser>> parsers (n);
  std::thread device_thread;
  std::vector<int> running;                       // the group the device thread works on (its parsers are released when it is done)
  double t_blocked = 0, t_serial = 0;
  auto launch = [&] (std::vector<int>& group) {
    const double t_j = now_s();
    if (device_thread.joinable()) device_thread.join();
    t_blocked += now_s() - t_j;
    running.swap (group); group.clear();
    if (running.empty()) return;
    device_thread = std::thread ([&, device] () {
      hipSetDevice (device);
      if (!getenv ("LH264_COMPRESS_PARSE_ONLY"))          // diagnostic: the host side alone
        compress_group (arena, parsers, running, len, out, std::max (1, threads / 2));
      for (int i : running) parsers[i].reset();          // the pictures go back to the pool while the next wave is parsed
    });
  };
  std::vector<int> group;
  size_t in_group = 0;
  for (int w0 = 0; w0 < n; w0 += kWave) {
    const int w1 = std::min (n, w0 + kWave);
    const double t_p = now_s();
    run_parallel (w1 - w0, threads, [&] (int k) {
      const int i = w0 + k;
      parsers[i].reset (new lh264host::Parser());
      parsers[i]->set_want_coeffs (false);
      parsers[i]->set_sparse_levels (true);
      parsers[i]->set_stream_arena (true);
      if (data[i] || !len[i]) parsers[i]->feed_file (data[i], len[i]);
    });
    if (trace_on()) fprintf (stderr, "[lh264 compress] wave of %d streams parsed in %.3f s\n", w1 - w0, now_s() - t_p);
    const double t_s = now_s();
    for (int i = w0; i < w1; i++) {
      lh264_compressed_t& r = *out[i];
      lh264host::Parser& P = *parsers[i];
      r.main_stream = P.main_stream();
      if (!P.pcm_samples().empty()) { r.tag[LH264_TAG_PCM] = P.pcm_samples(); r.has_tag[LH264_TAG_PCM] = true; }
      r.pictures = (int)P.frames().size();
      size_t mbs = 0;
      bool symbols = true;
      for (auto& f : P.frames()) { mbs += (size_t)f->mb_w * f->mb_h; symbols = symbols && f->syn_off.size() == (size_t)f->mb_w * f->mb_h + 1 && (f->syn_off.back() == f->syn_syms.size()); }
      if (!P.error().empty()) { r.status = LH264_E_UNSUPPORTED; r.error = P.error(); }
      else if (!symbols) { r.status = LH264_E_UNSUPPORTED; r.error = "a picture with an incomplete slice"; }
      else if (P.damaged()) { r.status = LH264_E_UNSUPPORTED; r.error = "a picture with macroblocks no slice covers: the reference conceals them, which is not modelled (the stream would not restore)"; }
      if (r.status != LH264_OK || mbs == 0) { parsers[i].reset(); continue; }
      if (in_group && in_group + mbs > kBudget) { launch (group); in_group = 0; }
      group.push_back (i); in_group += mbs;
    }
    t_serial += now_s() - t_s;
  }
  launch (group);
  if (device_thread.joinable()) device_thread.join();
  if (trace_on()) fprintf (stderr, "[lh264 compress] %d streams: %.3f s (between waves %.3f s, of which waiting for the device stage %.3f s)\n", n, now_s() - t_call, t_serial, t_blocked);
  return LH264_OK;
}
void lh264_compress_release (void) {
  int cur = 0;
  hipGetDevice (&cur);
  for (int d = 0; d < kMaxDevices; d++) {
    std::lock_guard<std::mutex> lock (g_arena_mutex[d]);
    if (g_arena[d]) { hipSetDevice (d); g_arena[d].reset(); }
  }
  hipSetDevice (cur);
}

// the same batch over several devices of the node: the streams are cut into contiguous shares of about equal input size, one
// host thread per share drives lh264_compress_batch on its device (streams are independent: no exchange between devices)
int lh264_compress_batch_devices (const uint8_t* const* data, const size_t* len, int n, int threads, const int* devices, int n_devices,
                                  lh264_compressed_t** out) {
  if (!data || !len || !out || n < 0 || !devices || n_devices < 1) return LH264_E_ARG;
  if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
  size_t total = 0;
  for (int i = 0; i < n; i++) total += len[i];
  std::vector<int> first (n_devices + 1, n);
  first[0] = 0;
  {
    size_t acc = 0; int s = 1;
    for (int i = 0; i < n && s < n_devices; i++) {
      acc += len[i];
      while (s < n_devices && acc * n_devices >= total * s) first[s++] = i + 1;
    }
  }
  std::vector<int> rcs (n_devices, LH264_OK);
  std::vector<std::thread> th;
  for (int s = 0; s < n_devices; s++) {
    th.emplace_back ([&, s] () {
      const int a = first[s], b = first[s + 1];
      if (hipSetDevice (devices[s]) != hipSuccess) { for (int i = a; i < b; i++) { out[i] = new lh264_compressed(); out[i]->status = LH264_E_HIP; out[i]->error = "hipSetDevice failed"; } rcs[s] = LH264_E_HIP; return; }
      rcs[s] = lh264_compress_batch (data + a, len + a, b - a, std::max (1, threads / n_devices), out + a);
    });
  }
  for (auto& t : th) t.join();
  for (int rc : rcs) if (rc != LH264_OK) return rc;
  return LH264_OK;
}
int lh264_compressed_status (const lh264_compressed_t* c) { return c ? c->status : LH264_E_ARG; }
const char* lh264_compressed_error (const lh264_compressed_t* c) { return c ? c->error.c_str() : ""; }
const uint8_t* lh264_compressed_main (const lh264_compressed_t* c, size_t* len) {
  if (!c) return nullptr;
  if (len) *len = c->main_stream.size();
  return c->main_stream.empty() ? (const uint8_t*)"" : c->main_stream.data();
}
const uint8_t* lh264_compressed_tag (const lh264_compressed_t* c, int tag, size_t* len) {
  if (!c || tag < 0 || tag >= 72 || !c->has_tag[tag]) { if (len) *len = 0; return nullptr; }
  if (len) *len = c->tag[tag].size();
  return c->tag[tag].data();
}
int lh264_compressed_pictures (const lh264_compressed_t* c) { return c ? c->pictures : 0; }
void lh264_compressed_free (lh264_compressed_t* c) { delete c; }

}
