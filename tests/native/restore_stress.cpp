// restore_stress.cpp - robustness driver for the restore direction (csrc/host/pip_restore.cpp) and the default-stream
// writer (Parser::feed_file), built with -fsanitize=address,undefined by tests/test_parser_robust.py (CPU only).
// argv: base paths; for each, <base>.264 is the original, <base>.pip the default stream and <base>.pip.<tag> the tagged
// streams.  The clean restore must reproduce the original; damaged inputs may fail but must never touch memory they
// should not, overflow, or hang.
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "../../losslessh264_amd/csrc/host/h264_parser.h"
#include "../../losslessh264_amd/csrc/host/pip_restore.h"
#include "hostile.h"

static uint32_t rng_state = 777;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

static bool load (const std::string& p, std::vector<uint8_t>& v) {
  FILE* f = fopen (p.c_str(), "rb");
  if (!f) return false;
  fseek (f, 0, SEEK_END); long n = ftell (f); fseek (f, 0, SEEK_SET);
  v.resize ((size_t)n);
  if (n && fread (v.data(), 1, (size_t)n, f) != (size_t)n) { fclose (f); return false; }
  fclose (f);
  return true;
}

static int restore (const std::vector<uint8_t>& main, const std::vector<std::vector<uint8_t>>& tags, const std::vector<char>& have,
                    std::vector<uint8_t>& out) {
  const uint8_t* ptr[72]; size_t len[72];
  for (int t = 0; t < 72; t++) { ptr[t] = have[t] ? (tags[t].empty() ? (const uint8_t*)"" : tags[t].data()) : nullptr; len[t] = tags[t].size(); }
  std::string err;
  return lh264host::pip_restore (main.data(), main.size(), ptr, len, 72, out, err);
}

int main (int argc, char** argv) {
  int cases = 0, failed_ok = 0;
  {   // crafted default streams with no tagged streams: 32-bit Exp-Golomb values in every header field the restorer reads
    std::vector<std::vector<uint8_t>> none (72);
    std::vector<char> have (72, 0), all (72, 1);
    std::vector<uint8_t> out;
    for (uint64_t v : kHostileValues) for (int field = 0; field < 12; field++) for (int cab = 0; cab < 2; cab++) {
      const std::vector<uint8_t> m = hostile_stream (hostile_case (v, field, cab != 0), true);
      if (restore (m, none, have, out) != 0) failed_ok++;
      if (restore (m, none, all, out) != 0) failed_ok++;      // every tagged stream present and empty
      cases += 2;
    }
  }
  for (int a = 1; a < argc; a++) {
    const std::string base = argv[a];
    std::vector<uint8_t> orig, main;
    if (!load (base + ".264", orig) || !load (base + ".pip", main)) { fprintf (stderr, "cannot read %s\n", base.c_str()); return 2; }
    std::vector<std::vector<uint8_t>> tags (72);
    std::vector<char> have (72, 0);
    for (int t = 0; t < 72; t++) have[t] = load (base + ".pip." + std::to_string (t), tags[t]);
    std::vector<uint8_t> out;
    if (restore (main, tags, have, out) != 0 || out != orig) { fprintf (stderr, "%s: clean restore differs\n", base.c_str()); return 3; }
    {   // the default stream of the original, built again, equals the one given
      lh264host::Parser p;
      p.feed_file (orig.data(), orig.size());
      if (p.main_stream() != main) { fprintf (stderr, "%s: default stream differs\n", base.c_str()); return 3; }
    }
    cases++;
    for (int trial = 0; trial < 40; trial++) {
      std::vector<std::vector<uint8_t>> t2 = tags;
      std::vector<uint8_t> m2 = main;
      std::vector<char> h2 = have;
      const int what = trial % 5;
      if (what == 0) { for (int i = 0; i < 3 && !m2.empty(); i++) m2[rnd() % m2.size()] = (uint8_t)rnd(); }
      else if (what == 1) { m2.resize (rnd() % (m2.size() + 1)); }
      else if (what == 2) { for (int t = 0; t < 72; t++) if (have[t] && (rnd() & 3) == 0 && !t2[t].empty()) for (int i = 0; i < 4; i++) t2[t][rnd() % t2[t].size()] = (uint8_t)rnd(); }
      else if (what == 3) { for (int t = 0; t < 72; t++) if (have[t] && (rnd() & 3) == 0) t2[t].resize (rnd() % (t2[t].size() + 1)); }
      else { for (int t = 0; t < 72; t++) if (have[t] && (rnd() & 7) == 0) h2[t] = 0; }
      if (restore (m2, t2, h2, out) != 0) failed_ok++;
      cases++;
    }
    for (int trial = 0; trial < 10; trial++) {     // damaged originals through the default-stream writer
      std::vector<uint8_t> o2 = orig;
      if (trial & 1) o2.resize (rnd() % (o2.size() + 1));
      for (int i = 0; i < 6 && !o2.empty(); i++) o2[rnd() % o2.size()] = (uint8_t)rnd();
      lh264host::Parser p;
      p.feed_file (o2.data(), o2.size());
      cases++;
    }
  }
  printf ("cases=%d rejected=%d\n", cases, failed_ok);
  return 0;
}
