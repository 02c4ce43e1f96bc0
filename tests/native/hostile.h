// hostile.h - hand-built H.264 streams for the sanitizer drivers (parser_stress.cpp, restore_stress.cpp).
#ifndef LH264_TESTS_HOSTILE_H_
#define LH264_TESTS_HOSTILE_H_
#include <stdint.h>
#include <vector>
// ---- hand-built hostile streams: Exp-Golomb codes with up to 31 leading zeros in every header field that is narrowed or used as
// an index (random byte flips never produce those) ----------------------------------------------------------------------------
struct BitW {
  std::vector<uint8_t> b; int n = 0;
  void bit (int v) { if ((n & 7) == 0) b.push_back (0); if (v) b.back() |= (uint8_t) (0x80 >> (n & 7)); n++; }
  void u (int bits, uint32_t v) { for (int i = bits - 1; i >= 0; i--) bit ((v >> i) & 1); }
  void ue (uint64_t v) { const uint64_t x = v + 1; int z = 0; while ((x >> (z + 1)) != 0) z++; for (int i = 0; i < z; i++) bit (0); for (int i = z; i >= 0; i--) bit ((int) ((x >> i) & 1)); }
  void se (int v) { ue (v > 0 ? 2 * (uint64_t)v - 1 : 2 * (uint64_t) (-v)); }
  void trail() { bit (1); while (n & 7) bit (0); }
};
static void put_nal (std::vector<uint8_t>& out, int hdr, const BitW& w) {
  const uint8_t sc[4] = {0, 0, 0, 1};
  out.insert (out.end(), sc, sc + 4);
  out.push_back ((uint8_t)hdr);
  int zeros = 0;
  for (uint8_t x : w.b) {
    if (zeros >= 2 && x <= 3) { out.push_back (3); zeros = 0; }
    out.push_back (x);
    zeros = x == 0 ? zeros + 1 : 0;
  }
}
struct Hostile { uint64_t sps_id = 0, l2fn = 0, l2poc = 0, nref = 1, mbw = 10, mbh = 8, pps_id = 0, pps_sps = 0, nidx = 0, first_mb = 0, slice_pps = 0, skip_run = 0, ovr_nidx = 0; bool cabac = false, have_ovr = false; };
static std::vector<uint8_t> hostile_stream (const Hostile& h, bool p_slice) {
  std::vector<uint8_t> out;
  { BitW w; w.u (8, 66); w.u (8, 0); w.u (8, 30); w.ue (h.sps_id); w.ue (h.l2fn); w.ue (0); w.ue (h.l2poc); w.ue (h.nref); w.bit (0); w.ue (h.mbw); w.ue (h.mbh);
    w.bit (1); w.bit (0); w.bit (0); w.bit (0); w.trail(); put_nal (out, 0x67, w); }
  { BitW w; w.ue (h.pps_id); w.ue (h.pps_sps); w.bit (h.cabac); w.bit (0); w.ue (0); w.ue (h.nidx); w.ue (0); w.bit (0); w.u (2, 0); w.se (0); w.se (0); w.se (0);
    w.bit (0); w.bit (0); w.bit (0); w.trail(); put_nal (out, 0x68, w); }
  { BitW w; w.ue (0); w.ue (7); w.ue (h.slice_pps); w.u (4, 0); w.ue (0); w.u (4, 0); w.bit (0); w.bit (0); w.se (0);            // an IDR I slice first
    for (int i = 0; i < 40; i++) w.ue (3);     // I16x16 macroblocks of some kind; whatever they parse as
    w.trail(); put_nal (out, 0x65, w); }
  if (p_slice) {
    BitW w; w.ue (h.first_mb); w.ue (5); w.ue (h.slice_pps); w.u (4, 1); w.u (4, 2);
    w.bit (h.have_ovr); if (h.have_ovr) w.ue (h.ovr_nidx);
    w.bit (0); w.bit (0); w.se (0);
    if (h.cabac) { w.ue (0); while (w.n & 7) w.bit (1); for (int i = 0; i < 64; i++) w.u (8, 0x5a); }
    else { w.ue (h.skip_run); w.ue (0); w.ue (0); w.se (0); }
    w.trail(); put_nal (out, 0x41, w);
  }
  return out;
}

static const uint64_t kHostileValues[] = {0xfffffffeull, 0x80000000ull, 0x7fffffffull, 0xffffull, 139263, 139264, 1u << 20};
// field 0..11: which header field carries the value
static Hostile hostile_case (uint64_t v, int field, bool cabac) {
  Hostile h; h.cabac = cabac;
  switch (field) {
  case 0: h.first_mb = v; break;
  case 1: h.skip_run = v; break;
  case 2: h.slice_pps = v; break;
  case 3: h.have_ovr = true; h.ovr_nidx = v; break;
  case 4: h.l2fn = v; break;
  case 5: h.l2poc = v; break;
  case 6: h.mbw = v; break;
  case 7: h.mbh = v; break;
  case 8: h.sps_id = v; break;
  case 9: h.pps_id = v; break;
  case 10: h.nidx = v; break;
  default: h.mbw = v & 1023; h.mbh = v & 1023; h.nref = v; break;
  }
  return h;
}
#endif
