// parser_stress.cpp - robustness driver for the host front end, built with -fsanitize=address,undefined by
// tests/test_parser_robust.py (CPU only).  Feeds every stream given on the command line to lh264host::Parser whole, in
// NAL-sized pieces, truncated at many points and with bytes corrupted by a fixed-seed generator; the parser may report
// errors but must never read or write out of bounds, overflow, or hang.
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../losslessh264_amd/csrc/host/h264_parser.h"

static uint32_t rng_state = 12345;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

// chunk 2: the configuration of the batch compressor (whole file at once, pictures in a stream arena, levels as a sparse list,
// no dequantised coefficients)
static long run (const std::vector<uint8_t>& bs, size_t chunk) {
  lh264host::Parser p;
  if (chunk == 2) {
    p.set_stream_arena (true); p.set_sparse_levels (true); p.set_want_coeffs (false);
    p.feed_file (bs.data(), bs.size());
    long mbs = 0;
    for (auto& f : p.frames()) {
      mbs += (long)f->mbs.size();
      if (f->levels.size() != 0 || f->coeffs.size() != 0) { fprintf (stderr, "planes allocated in sparse mode\n"); exit (3); }
      for (uint64_t e : f->sparse) if ((e >> 16) >= (uint64_t)f->mbs.size() * 384 || (e & 0xffff) == 0) { fprintf (stderr, "bad sparse entry\n"); exit (3); }
      if (f->syn_off.size() && f->syn_off.back() != f->syn_syms.size()) { fprintf (stderr, "symbol offsets inconsistent\n"); exit (3); }
    }
    return mbs;
  }
  if (chunk == 0) p.feed (bs.data(), bs.size());
  else {
    // split at start codes, like the console decoder
    size_t pos = 0;
    while (pos < bs.size()) {
      size_t next = pos + 3;
      for (; next + 3 <= bs.size(); next++) if (bs[next] == 0 && bs[next + 1] == 0 && bs[next + 2] == 1) break;
      if (next + 3 > bs.size()) next = bs.size();
      p.feed (&bs[pos], next - pos);
      pos = next;
    }
  }
  p.flush();
  long mbs = 0;
  for (auto& f : p.frames()) {
    mbs += (long)f->mbs.size();
    if (f->syn_off.size() && f->syn_off.back() != f->syn_syms.size()) { fprintf (stderr, "symbol offsets inconsistent\n"); exit (3); }
    for (auto& s : f->syn_syms) if (s.kind != LH264_SYM_SPLICE && s.kind != LH264_SYM_RAW && (s.prior >> 27) >= LH264_TB_COUNT) { fprintf (stderr, "bad table id\n"); exit (3); }
  }
  return mbs;
}


#include "hostile.h"
static int hostile_cases (long& total) {
  int cases = 0;
  for (uint64_t v : kHostileValues) for (int field = 0; field < 12; field++) for (int cab = 0; cab < 2; cab++) {
    const std::vector<uint8_t> bs = hostile_stream (hostile_case (v, field, cab != 0), true);
    for (size_t chunk = 0; chunk < 3; chunk++) { total += run (bs, chunk); cases++; }
  }
  return cases;
}

int main (int argc, char** argv) {
  long total = 0; int cases = 0;
  cases += hostile_cases (total);
  for (int a = 1; a < argc; a++) {
    FILE* f = fopen (argv[a], "rb");
    if (!f) { perror (argv[a]); return 2; }
    std::vector<uint8_t> bs; uint8_t tmp[65536]; size_t n;
    while ((n = fread (tmp, 1, sizeof (tmp), f)) > 0) bs.insert (bs.end(), tmp, tmp + n);
    fclose (f);
    total += run (bs, 0); total += run (bs, 1); total += run (bs, 2); cases += 3;
    for (int t = 0; t < 24; t++) {                 // truncations
      std::vector<uint8_t> c (bs.begin(), bs.begin() + (size_t) (rnd() % (bs.size() + 1)));
      total += run (c, t % 3); cases++;
    }
    for (int t = 0; t < 40; t++) {                 // corruptions: a few random bytes overwritten
      std::vector<uint8_t> c = bs;
      const int k = 1 + (int) (rnd() % 8);
      for (int i = 0; i < k; i++) c[rnd() % c.size()] = (uint8_t)rnd();
      total += run (c, t % 3); cases++;
    }
    {                                              // empty and tiny inputs
      std::vector<uint8_t> e; total += run (e, 0);
      std::vector<uint8_t> one = {0, 0, 1}; total += run (one, 0);
      std::vector<uint8_t> two = {0, 0, 1, 0x65}; total += run (two, 1); cases += 3;
    }
  }
  printf ("cases=%d macroblocks=%ld\n", cases, total);
  return 0;
}
