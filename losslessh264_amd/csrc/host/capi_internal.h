// capi_internal.h - what the C ABI's opaque handles are, for the translation units of the library that share them
#pragma once
#include <atomic>
#include <thread>
#include <vector>
#include "h264_parser.h"
struct lh264_parser { lh264host::Parser p; };
static inline lh264host::Parser* lh264_parser_impl (lh264_parser* h) { return h ? &h->p : nullptr; }

// n independent pieces of work on `threads` host threads (0 = one per hardware thread)
template <typename F> static void run_parallel (int n, int threads, F&& fn) {
  if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
  if (threads < 1) threads = 1;
  if (threads > n) threads = n;
  std::atomic<int> next (0);
  auto worker = [&] () { for (;;) { const int i = next.fetch_add (1); if (i >= n) break; fn (i); } };
  std::vector<std::thread> pool;
  for (int t = 1; t < threads; t++) pool.emplace_back (worker);
  worker();
  for (auto& t : pool) t.join();
}

