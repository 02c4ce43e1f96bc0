// pip_symbols.h - host side of SURVEY section 8 row a10: turns the parsed macroblock syntax of a stream into the ordered
// list of (prior table, prior index, value, tag) symbols the recompressor codes for everything that is not a residual
// coefficient.  The coefficient symbols (row a8) are produced on the device and spliced in at the marker.
// What must be reproduced of the reference: WHICH prior codes WHICH value in WHICH order and to which tag - the per-macroblock
// emit code of WelsDecodeSliceForNonRecoding (decoder/core/src/decode_slice.cpp:2174-2473) and the prior selection of
// MacroblockModel (decoder/core/src/macroblock_model.cpp:370-645), including its history image FreqImage
// (decoder/core/inc/decoded_macroblock.h:106-192).
#pragma once
#include <stdint.h>
#include <vector>
#include "../../../include/lh264.h"
#include "h264_parser.h"

namespace lh264host {

class Symbolizer {
 public:
  // appends the picture's symbols to f.syn_syms / f.syn_off (pictures of one stream, in decode order)
  void picture (FrameOut& f);

 private:
  struct Cell {                       // what the model remembers of a macroblock (DecodedMacroblock, decoded_macroblock.h:4-34)
    uint8_t initialized = 0, zeroed = 0, cbp_c = 0, cbp_l = 0, chroma_mode = 0, luma16_mode = 0;
    uint16_t cached_skips = 0;
    uint32_t mb_type = 0, num_ref = 0;
  };
  std::vector<Cell> img_[2];
  int img_w_ = 0, img_h_ = 0, cur_ = 0, last_frame_id_ = 0;
  std::vector<int8_t> ipm_;           // the decoder's pIntraPredMode[mb][0..6] (raw modes of the bottom row / right column)
  std::vector<uint8_t> nxn_;          // macroblock is I4x4 / I8x8
  std::vector<lh264_ctx_sym_t> flat_; std::vector<uint32_t> start_, cnt_;   // scratch of picture()
  void update_frame (int frame_id);
};

}  // namespace lh264host
