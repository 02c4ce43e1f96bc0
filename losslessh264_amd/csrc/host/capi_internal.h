// capi_internal.h - what the C ABI's opaque handles are, for the translation units of the library that share them
#pragma once
#include "h264_parser.h"
struct lh264_parser { lh264host::Parser p; };
static inline lh264host::Parser* lh264_parser_impl (lh264_parser* h) { return h ? &h->p : nullptr; }
