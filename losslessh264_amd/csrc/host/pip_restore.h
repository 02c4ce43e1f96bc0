// pip_restore.h - the restore direction of the recompressor, host side: given the default
// stream (the ".pip" file: the Annex-B input minus its slice data) and the tagged arithmetic-coded streams (".pip.<tag>"),
// rebuild the original H.264 byte stream.  Three pieces, all fresh code:
//   * the adaptive binary arithmetic DEcoder: the scan side of rows a9 (DynProb, Branch<n>, scanInt / scanUEGkInt /
//     scanBitsZeroToPow2Inclusive / scanUnary, ArithmeticCodedInput::scanBit, vpx_read / vpx_reader_fill;
//     decoder/core/inc/compression_stream.h:87-241,289-351,607-676, inc/bitreader.h:77-136, src/bitreader.cpp:43-108)
//   * the model in scan order: the inverse of csrc/host/pip_symbols.cpp (row a10) and of the device kernels of
//     csrc/lh264_ctx.hip (row a8): which prior decodes which value, in which order, from which tag
//     (WelsDecodeSliceForRecoding decode_slice.cpp:2476-2830, decode4x4 :2096-2124, macroblock_model.cpp:370-645)
//   * a CAVLC macroblock-layer writer straight from ITU-T H.264 7.3.5 / 9.2 (the reference borrows its encoder's writer,
//     encoder/core/src/svc_set_mb_syn_cavlc.cpp:266-320 via decoder/core/inc/encoder_from_decoder.h)
// The adaptive decode is serial by nature (every prior depends on the values decoded before it); this first version runs it
// on the host, one stream per thread.  CABAC slices go through a CABAC macroblock writer (9.3.2-9.3.4, the mirror image of the
// front end's CABAC parser).  I_PCM macroblocks are reported as unsupported.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string>
#include <vector>

namespace lh264host {

// tags[t] / tag_len[t] for t in 0 .. n_tags-1 indexed by the tag id of billing.h:6-55 (so n_tags >= 70 to carry the pad-bit
// tag 69); a null pointer = the stream does not exist.  Returns 0 and fills out, or < 0 with a message in err.
int pip_restore (const uint8_t* main_stream, size_t main_len, const uint8_t* const* tags, const size_t* tag_len, int n_tags,
                 std::vector<uint8_t>& out, std::string& err);

}  // namespace lh264host
