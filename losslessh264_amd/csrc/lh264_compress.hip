// lh264_compress.hip - the compress direction behind one C call (include/lh264.h: lh264_compress_batch): the host
// orchestration the reference does inside its decoder loop (decode_slice.cpp:3085-3112 per macroblock, flushToWriter at
// the end), here per batch of independent streams: parse on host threads, stage records and symbol lists in HBM, one
// launch of the context-index kernels and one of the coder kernel, copy the tagged streams back.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <memory>
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

namespace {

struct DevBuf {
  void* p = nullptr;
  ~DevBuf() { if (p) hipFree (p); }
  bool alloc (size_t bytes, bool zero) {
    if (p) { hipFree (p); p = nullptr; }
    if (hipMalloc (&p, bytes ? bytes : 16) != hipSuccess) { p = nullptr; return false; }
    if (zero && hipMemset (p, 0, bytes ? bytes : 16) != hipSuccess) return false;
    return true;
  }
  template <typename T> T* as() const { return (T*)p; }
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
void compress_group (std::vector<lh264host::Parser*>& parsers, const std::vector<int>& idx, const size_t* len, lh264_compressed_t** out) {
  using lh264host::FrameOut;
  size_t n_mbs = 0, n_slices = 0, n_jobs = 0, n_syn = 0, n_off = 0;
  int max_mbs = 1;
  for (int i : idx) for (auto& f : parsers[i]->frames()) {
      const size_t n = (size_t)f->mb_w * f->mb_h;
      n_mbs += n; n_slices += f->slices.size(); n_jobs++; n_syn += f->syn_syms.size(); n_off += n + 1;
      max_mbs = std::max (max_mbs, (int)n);
    }
  const int n_chains = (int)idx.size();
  if (n_jobs == 0) return;
  // ---- host staging ------------------------------------------------------------------------------------------------------
  std::vector<lh264_mb_t> h_mbs (n_mbs);
  std::vector<int16_t> h_lev (n_mbs * 384);
  std::vector<lh264_slice_t> h_sl (n_slices);
  std::vector<lh264_ctx_sym_t> h_syn (std::max<size_t> (n_syn, 1));
  std::vector<uint32_t> h_off (n_off);
  std::vector<lh264_ctx_job_t> h_cj (n_jobs);
  std::vector<lh264_code_job_t> h_kj (n_jobs);
  std::vector<int32_t> h_first (n_chains + 1);
  std::vector<lh264_code_stream_t> h_st (n_chains);
  std::vector<uint32_t> hash_cap (n_chains), out_cap (n_chains);
  size_t keys_total = 0, out_total = 0;
  for (int c = 0; c < n_chains; c++) {
    size_t mbs = 0;
    for (auto& f : parsers[idx[c]]->frames()) mbs += (size_t)f->mb_w * f->mb_h;
    uint32_t hc = 1u << 16;
    while (hc < mbs * 8 && hc < (1u << 22)) hc <<= 1;       // cells touched grow far slower than macroblocks; status 1 reports a full table
    hash_cap[c] = hc; keys_total += hc;
    out_cap[c] = (uint32_t)std::max<size_t> (1u << 16, 2 * len[idx[c]] + 4096);
    out_total += (size_t)LH264_N_TAG_SLOTS * out_cap[c];
  }
  DevBuf d_mbs, d_lev, d_sl, d_nnz, d_syms, d_nsyms, d_cj, d_first, d_syn, d_off, d_kj, d_st, d_keys, d_cells, d_out, d_len;
  const bool ok = d_mbs.alloc (n_mbs * sizeof (lh264_mb_t), false) && d_lev.alloc (n_mbs * 768, false) && d_sl.alloc (n_slices * sizeof (lh264_slice_t), false) &&
                  d_nnz.alloc (n_mbs * 24, true) && d_syms.alloc (n_mbs * LH264_CTX_MAX_SYMS * sizeof (lh264_ctx_sym_t), false) && d_nsyms.alloc (n_mbs * 2, true) &&
                  d_cj.alloc (n_jobs * sizeof (lh264_ctx_job_t), false) && d_first.alloc ((n_chains + 1) * 4, false) && d_syn.alloc (h_syn.size() * sizeof (lh264_ctx_sym_t), false) &&
                  d_off.alloc (n_off * 4, false) && d_kj.alloc (n_jobs * sizeof (lh264_code_job_t), false) && d_st.alloc (n_chains * sizeof (lh264_code_stream_t), false) &&
                  d_keys.alloc (keys_total * 4, true) && d_cells.alloc (keys_total * 64, true) && d_out.alloc (out_total, false) &&
                  d_len.alloc ((size_t)n_chains * (LH264_N_TAG_SLOTS + 1) * 4, true);
  if (!ok) { fail_all (out, idx, LH264_E_HIP, "device allocation failed"); return; }
  size_t mo = 0, so = 0, yo = 0, oo = 0, j = 0, ko = 0, uo = 0;
  std::vector<int> past;
  for (int c = 0; c < n_chains; c++) {
    auto& fr = parsers[idx[c]]->frames();
    h_first[c] = (int32_t)j;
    past_policy (fr, past);
    std::vector<size_t> mb_at (fr.size());
    for (size_t i = 0; i < fr.size(); i++) {
      FrameOut& f = *fr[i];
      const size_t n = (size_t)f.mb_w * f.mb_h;
      mb_at[i] = mo;
      memcpy (&h_mbs[mo], f.mbs.data(), n * sizeof (lh264_mb_t));
      memcpy (&h_lev[mo * 384], f.levels.data(), n * 768);
      if (!f.slices.empty()) memcpy (&h_sl[so], f.slices.data(), f.slices.size() * sizeof (lh264_slice_t));
      if (!f.syn_syms.empty()) memcpy (&h_syn[yo], f.syn_syms.data(), f.syn_syms.size() * sizeof (lh264_ctx_sym_t));
      memcpy (&h_off[oo], f.syn_off.data(), (n + 1) * 4);
      lh264_ctx_job_t& cj = h_cj[j];
      cj.mbs_dev = d_mbs.as<lh264_mb_t>() + mo; cj.levels_dev = d_lev.as<int16_t>() + mo * 384; cj.slices_dev = d_sl.as<lh264_slice_t>() + so;
      cj.nnz_cur_dev = d_nnz.as<uint8_t>() + mo * 24;
      cj.nnz_past_dev = past[i] < 0 ? nullptr : d_nnz.as<uint8_t>() + mb_at[past[i]] * 24;
      cj.syms_dev = d_syms.as<lh264_ctx_sym_t>() + mo * LH264_CTX_MAX_SYMS; cj.n_syms_dev = d_nsyms.as<uint16_t>() + mo;
      cj.mb_w = f.mb_w; cj.mb_h = f.mb_h;
      lh264_code_job_t& kj = h_kj[j];
      kj.syn_syms_dev = d_syn.as<lh264_ctx_sym_t>() + yo; kj.syn_off_dev = d_off.as<uint32_t>() + oo;
      kj.ctx_syms_dev = cj.syms_dev; kj.ctx_n_syms_dev = cj.n_syms_dev; kj.n_mbs = (int32_t)n; kj.reserved = 0;
      mo += n; so += f.slices.size(); yo += f.syn_syms.size(); oo += n + 1; j++;
    }
    lh264_code_stream_t& st = h_st[c];
    st.hash_keys_dev = d_keys.as<uint32_t>() + ko; st.hash_cells_dev = d_cells.as<uint32_t>() + ko * 16;
    st.out_dev = d_out.as<uint8_t>() + uo; st.out_len_dev = d_len.as<uint32_t>() + (size_t)c * (LH264_N_TAG_SLOTS + 1);
    st.hash_cap = hash_cap[c]; st.out_cap = out_cap[c];
    ko += hash_cap[c]; uo += (size_t)LH264_N_TAG_SLOTS * out_cap[c];
  }
  h_first[n_chains] = (int32_t)j;
  auto up = [] (DevBuf& d, const void* s, size_t bytes) { return bytes == 0 || hipMemcpy (d.p, s, bytes, hipMemcpyHostToDevice) == hipSuccess; };
  if (!(up (d_mbs, h_mbs.data(), n_mbs * sizeof (lh264_mb_t)) && up (d_lev, h_lev.data(), n_mbs * 768) && up (d_sl, h_sl.data(), n_slices * sizeof (lh264_slice_t)) &&
        up (d_syn, h_syn.data(), n_syn * sizeof (lh264_ctx_sym_t)) && up (d_off, h_off.data(), n_off * 4) && up (d_cj, h_cj.data(), n_jobs * sizeof (lh264_ctx_job_t)) &&
        up (d_kj, h_kj.data(), n_jobs * sizeof (lh264_code_job_t)) && up (d_first, h_first.data(), (n_chains + 1) * 4) && up (d_st, h_st.data(), n_chains * sizeof (lh264_code_stream_t)))) {
    fail_all (out, idx, LH264_E_HIP, "upload failed"); return;
  }
  int rc = lh264_ctx_index_chains (d_cj.as<lh264_ctx_job_t>(), d_first.as<int32_t>(), n_chains, (int)n_jobs, max_mbs, nullptr);
  if (rc == LH264_OK) rc = lh264_code_chains (d_kj.as<lh264_code_job_t>(), d_first.as<int32_t>(), d_st.as<lh264_code_stream_t>(), n_chains, nullptr);
  if (rc != LH264_OK || hipDeviceSynchronize() != hipSuccess) { fail_all (out, idx, rc != LH264_OK ? rc : LH264_E_HIP, "kernel launch failed"); return; }
  std::vector<uint32_t> lens ((size_t)n_chains * (LH264_N_TAG_SLOTS + 1));
  if (hipMemcpy (lens.data(), d_len.p, lens.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { fail_all (out, idx, LH264_E_HIP, "download failed"); return; }
  uo = 0;
  for (int c = 0; c < n_chains; c++) {
    lh264_compressed_t& r = *out[idx[c]];
    const uint32_t* L = &lens[(size_t)c * (LH264_N_TAG_SLOTS + 1)];
    if (L[LH264_N_TAG_SLOTS] != 0) { r.status = LH264_E_HIP; r.error = "device coder status " + std::to_string (L[LH264_N_TAG_SLOTS]) + " (1: prior table full, 4: output overflow)"; }
    else for (int slot = 0; slot < 35; slot++) if (L[slot]) {
          const int tag = slot == 34 ? 69 : slot;
          r.tag[tag].resize (L[slot]); r.has_tag[tag] = true;
          if (hipMemcpy (r.tag[tag].data(), d_out.as<uint8_t>() + uo + (size_t)slot * out_cap[c], L[slot], hipMemcpyDeviceToHost) != hipSuccess) { r.status = LH264_E_HIP; r.error = "download failed"; }
        }
    uo += (size_t)LH264_N_TAG_SLOTS * out_cap[c];
  }
}

}  // namespace

extern "C" {

int lh264_compress_batch (const uint8_t* const* data, const size_t* len, int n, int threads, lh264_compressed_t** out) {
  if (!data || !len || !out || n < 0) return LH264_E_ARG;
  for (int i = 0; i < n; i++) out[i] = new lh264_compressed();
  if (lh264_device_count() <= 0) { for (int i = 0; i < n; i++) { out[i]->status = LH264_E_NODEVICE; out[i]->error = "no HIP device visible"; } return LH264_E_NODEVICE; }
  std::vector<lh264_parser_t*> ph (n, nullptr);
  const int rc = lh264_parse_batch (data, len, n, threads, ph.data());
  if (rc != LH264_OK) { for (auto p : ph) if (p) lh264_parser_destroy (p); return rc; }
  std::vector<lh264host::Parser*> parsers (n);
  for (int i = 0; i < n; i++) parsers[i] = lh264_parser_impl (ph[i]);
  // sub-batches bounded by macroblock count (the symbol buffer takes 3.4 KB per macroblock)
  const size_t kBudget = 1500000;
  std::vector<int> group;
  size_t in_group = 0;
  auto flush = [&] () { if (!group.empty()) compress_group (parsers, group, len, out); group.clear(); in_group = 0; };
  for (int i = 0; i < n; i++) {
    lh264_compressed_t& r = *out[i];
    lh264host::Parser& P = *parsers[i];
    r.main_stream = P.main_stream();
    r.pictures = (int)P.frames().size();
    if (!P.error().empty()) { r.status = LH264_E_UNSUPPORTED; r.error = P.error(); continue; }
    size_t mbs = 0;
    bool symbols = true;
    for (auto& f : P.frames()) { mbs += (size_t)f->mb_w * f->mb_h; symbols = symbols && f->syn_off.size() == (size_t)f->mb_w * f->mb_h + 1 && (f->syn_off.back() == f->syn_syms.size()); }
    if (!symbols) { r.status = LH264_E_UNSUPPORTED; r.error = "a picture with an incomplete slice"; continue; }
    if (mbs == 0) continue;
    if (in_group && in_group + mbs > kBudget) flush();
    group.push_back (i); in_group += mbs;
  }
  flush();
  for (auto p : ph) lh264_parser_destroy (p);
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
