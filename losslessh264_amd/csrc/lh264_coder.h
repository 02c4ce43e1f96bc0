// lh264_coder.h - internal layout shared by the coder kernels (lh264_coder.hip) and their launcher (lh264_capi.hip).
#ifndef LH264_CODER_INTERNAL_H_
#define LH264_CODER_INTERNAL_H_
#include "../../include/lh264.h"

// the parallel binarisation works on segments of at most this many consecutive macroblocks of a picture, one wave each (<= 64: the wave
// lays a segment out with a lane per macroblock and 16-bit counters per lane)
#define LH264_CODER_SEG_MBS 64
// (the stream-per-workgroup form, lh264_coder_sw.hip: a workgroup of 256 threads per segment, a thread per macroblock)
#define LH264_CODER_SW_SEG_MBS 128
// per-segment decision counts (32-bit): [0 .. LH264_N_TAG_SLOTS-1] per tag slot (bit 31: the segment brings the tag's stream into
// existence), [LH264_N_TAG_SLOTS] all decisions of the segment
#define LH264_CODER_CNT_STRIDE (LH264_N_TAG_SLOTS + 1)
// the DynProbs of a stream are cut into P = 2^log2p partitions by a hash of their cell's key; per segment P + 1 words say where each
// partition's run of decision words starts inside the segment's words
#define LH264_CODER_MAX_LOG2P 7
#define LH264_CODER_MAX_PARTS (1 << LH264_CODER_MAX_LOG2P)

// per-stream record (32-bit words)
#define LH264_CODER_INFO_TAGBASE 0                                   /* [slot] first entry of the tag's list inside the stream's lists */
#define LH264_CODER_INFO_TAGCNT  LH264_N_TAG_SLOTS                   /* [slot] entries                                                 */
#define LH264_CODER_INFO_NDEC    (2 * LH264_N_TAG_SLOTS)             /* decisions of the stream                                        */
#define LH264_CODER_INFO_NQ      (2 * LH264_N_TAG_SLOTS + 1)         /* list entries incl. padding                                     */
#define LH264_CODER_INFO_TOUCH   (2 * LH264_N_TAG_SLOTS + 2)         /* 64-bit mask: tag streams that exist without a decision          */
#define LH264_CODER_INFO_DBASE   (2 * LH264_N_TAG_SLOTS + 4)         /* 64-bit: first decision word of the stream                      */
#define LH264_CODER_INFO_QBASE   (2 * LH264_N_TAG_SLOTS + 6)         /* 64-bit: first list entry of the stream                         */
#define LH264_CODER_INFO_STATUS  (2 * LH264_N_TAG_SLOTS + 8)
#define LH264_CODER_INFO_WORDS   96

// threads of a coder_resolve_kernel workgroup (one workgroup per stream)
#ifndef LH264_CODER_RESOLVE_WAVES
#define LH264_CODER_RESOLVE_WAVES 8
#endif
#define LH264_CODER_RESOLVE_THREADS (64 * LH264_CODER_RESOLVE_WAVES)

// the bool coder's output is summed up by chunks of this many decisions of a tag's list (a multiple of 8)
#ifndef LH264_CODER_CODE_CHUNK
#define LH264_CODER_CODE_CHUNK 256
#endif

// the range recurrence of a tag's list is walked in coarse chunks of this many decisions (a multiple of LH264_CODER_CODE_CHUNK), each
// from a start state found by looking back over the decisions in front of it (coder_range_seed_kernel)
#ifndef LH264_CODER_CODE_COARSE
#define LH264_CODER_CODE_COARSE 65536
#endif

// status bits reported in out_len_dev[LH264_N_TAG_SLOTS]
#define LH264_CODER_ST_TABLE_FULL 1
#define LH264_CODER_ST_OUT_FULL   4
#define LH264_CODER_ST_COUNT      8     /* more decisions in one macroblock than the counters hold */
#define LH264_CODER_ST_HANDOFF    16    /* internal: a wave step waited too long for its turn (the result is wrong) */

// waves per workgroup of the per-segment kernels (count, partition offsets, emit) and of the resolve kernel; their waves share nothing
// (every wave has its own LDS block).  Measured with 1, 2 and 4: alone the kernels are 3 - 10 % faster with 4, beside the reconstruct
// kernel of the next batch (which leaves 28 - 60 KB of a CU's LDS at 1080p / 720p) the step is the same within the run-to-run spread
#ifndef LH264_CODER_WG_WAVES
#define LH264_CODER_WG_WAVES 4
#endif

// threads of the one-workgroup kernels (running sums over pictures, streams, pairs).  A wave per SIMD, not four: these kernels sit in
// the middle of the coder's chain, and beside the reconstruct kernel of the next batch (four waves of 104 registers on every SIMD)
// there is room for ONE more wave per SIMD - a workgroup of 1,024 threads waited milliseconds for a CU to come free
#define CODER_ONE_WG 256

#endif
