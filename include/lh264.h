/*
 * lh264.h - C ABI of the MI355X-native decode-reconstruct hot path.
 *
 * This is the FINE drop-in boundary of the reference:
 * the reference reconstructs one slice at a time on the CPU through
 *     WelsTargetSliceConstruction(ctx)      codec/decoder/core/src/decode_slice.cpp:110-206
 *       -> WelsTargetMbConstruction         decode_slice.cpp:353-373
 *       -> WelsDeblockingFilterSlice        codec/decoder/core/src/deblocking.cpp:872-934
 *     ExpandReferencingPicture              codec/common/src/expand_pic.cpp:145-174
 * reading the per-macroblock arrays of SDqLayer (codec/decoder/core/inc/dec_frame.h:60-97).
 * Here the same per-macroblock state is handed over as flat records
 * (lh264_mb_t + 384 int16 coefficients per macroblock) and whole batches of
 * frames are reconstructed on the GPU by hand-written gfx950 kernels.
 *
 * Plain pointers and sizes only; no C++ or torch types cross this boundary.
 * Every pointer named *_dev is a device (HBM) address.
 */
#ifndef LH264_H_
#define LH264_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LH264_ABI_VERSION 3

/* ---- macroblock types: the reference's own flag values (codec/common/inc/wels_common_defs.h:264-281) */
#define LH264_MB_I4x4      0x0001
#define LH264_MB_I16x16    0x0002
#define LH264_MB_I8x8      0x0004
#define LH264_MB_P16x16    0x0008
#define LH264_MB_P16x8     0x0010
#define LH264_MB_P8x16     0x0020
#define LH264_MB_P8x8      0x0040
#define LH264_MB_P8x8REF0  0x0080
#define LH264_MB_SKIP      0x0100
#define LH264_MB_IPCM      0x0200
#define LH264_MB_INTRA     (LH264_MB_I4x4 | LH264_MB_I16x16 | LH264_MB_I8x8 | LH264_MB_IPCM)
#define LH264_MB_INTER     (LH264_MB_P16x16 | LH264_MB_P16x8 | LH264_MB_P8x16 | LH264_MB_P8x8 | LH264_MB_P8x8REF0 | LH264_MB_SKIP)
#define LH264_SUB_8x8 1
#define LH264_SUB_8x4 2
#define LH264_SUB_4x8 4
#define LH264_SUB_4x4 8

/* final (availability-resolved) intra modes, wels_common_defs.h:303-342 */
enum { LH264_I4_V = 0, LH264_I4_H, LH264_I4_DC, LH264_I4_DDL, LH264_I4_DDR, LH264_I4_VR, LH264_I4_HD, LH264_I4_VL,
       LH264_I4_HU, LH264_I4_DC_L, LH264_I4_DC_T, LH264_I4_DC_128, LH264_I4_DDL_TOP, LH264_I4_VL_TOP };
enum { LH264_I16_V = 0, LH264_I16_H, LH264_I16_DC, LH264_I16_P, LH264_I16_DC_L, LH264_I16_DC_T, LH264_I16_DC_128 };
enum { LH264_C_DC = 0, LH264_C_H, LH264_C_V, LH264_C_P, LH264_C_DC_L, LH264_C_DC_T, LH264_C_DC_128 };

#define LH264_MBF_T8x8   0x01   /* transform_size_8x8_flag (pTransformSize8x8Flag) */
#define LH264_MBF_PCM_IN_COEFF 0x02 /* I_PCM: the 384 samples are carried in the coefficient slot (low byte of each int16) */

/* intra_avail bits = pIntraNxNAvailFlag (decode_slice.cpp:567-569, rec_mb.cpp:88-96) */
#define LH264_AVAIL_T  0x1
#define LH264_AVAIL_TL 0x2
#define LH264_AVAIL_L  0x4
#define LH264_AVAIL_TR 0x8

#define LH264_MB_COEFFS 384     /* MB_COEFF_LIST_SIZE, codec/common/inc/wels_const_common.h:55 */
#define LH264_PAD_LUMA   32     /* PADDING_LENGTH, expand_pic.h:49 */
#define LH264_PAD_CHROMA 16
#define LH264_MAX_REFS   16

/*
 * One macroblock, 128 bytes. Field meaning == the SDqLayer array of the same
 * name at index iMbXy (dec_frame.h:60-97) at the moment the reference enters
 * WelsTargetMbConstruction, i.e. after parsing, before any in-place transform.
 */
typedef struct lh264_mb {
  uint16_t mb_type;         /* pMbType                                                   */
  uint8_t  cbp;             /* pCbp: luma bits 0-3, chroma (0..2) << 4                   */
  uint8_t  qp_y;            /* pLumaQp                                                   */
  uint8_t  qp_c[2];         /* pChromaQp[2] (Cb, Cr)                                     */
  uint8_t  flags;           /* LH264_MBF_*                                               */
  uint8_t  intra_avail;     /* pIntraNxNAvailFlag (only I8x8 reads it)                   */
  int8_t   intra_mode[16];  /* pIntra4x4FinalMode[16] (index = raster 4x4 block; I8x8 uses the
                               top-left 4x4 of each 8x8); I16x16: [0] = pIntraPredMode[7] */
  int8_t   chroma_mode;     /* pChromaPredMode                                           */
  uint8_t  reserved0;
  uint16_t slice_id;        /* index into the frame's lh264_slice_t table (pSliceIdc)    */
  uint8_t  sub_type[4];     /* pSubMbType[4]                                             */
  int8_t   ref_idx[4];      /* pRefIndex[LIST_0] of the four 8x8 quadrants               */
  uint8_t  nzc[24];         /* pNzc[24], the reference's raster layout (common_tables.cpp:39-47),
                               as left by the parser (the kernels only test != 0)        */
  int16_t  mv[16][2];       /* pMv[LIST_0][16][2], raster 4x4 blocks, quarter-pel        */
  uint8_t  reserved1[4];
} lh264_mb_t;

/* One slice of a frame (the SSliceHeader fields the hot path reads). */
typedef struct lh264_slice {
  int32_t  first_mb;            /* iFirstMbInSlice                                        */
  int32_t  n_mbs;               /* iTotalMbInCurSlice (consecutive raster MBs; no FMO)    */
  uint8_t  slice_type;          /* 0 = P, 2 = I (EWelsSliceType)                          */
  uint8_t  deblock_idc;         /* uiDisableDeblockingFilterIdc (0,1,2)                   */
  int8_t   alpha_c0_offset;     /* iSliceAlphaC0Offset (already x2, as the reference keeps it) */
  int8_t   beta_offset;         /* iSliceBetaOffset                                       */
  uint8_t  weighted_pred;       /* bUseWeightPredictionFlag                               */
  uint8_t  luma_log2_denom;     /* uiLumaLog2WeightDenom                                  */
  uint8_t  chroma_log2_denom;   /* uiChromaLog2WeightDenom                                */
  uint8_t  n_refs;              /* uiRefCount[0]                                          */
  int16_t  luma_weight[LH264_MAX_REFS];
  int16_t  luma_offset[LH264_MAX_REFS];
  int16_t  chroma_weight[LH264_MAX_REFS][2];
  int16_t  chroma_offset[LH264_MAX_REFS][2];
  int8_t   ref_slot[LH264_MAX_REFS]; /* ref_idx -> index into lh264_frame_job_t.ref (sRefPic.pRefList[LIST_0]) */
  uint8_t  luma_dc_weight;      /* Intra-Y 4x4 scaling-list entry [0] (16 = flat): feeds kiQMul of
                                   WelsLumaDcDequantIdct, decode_slice.cpp:272                      */
  uint8_t  reserved[7];
} lh264_slice_t;

/* A picture in HBM: three planes with the reference's padded layout
 * (pic_queue.cpp:62-112): stride = align32(W + 64) luma, half for chroma;
 * plane pointers address pixel (0,0), padding lies at negative offsets. */
typedef struct lh264_pic {
  uint8_t* y_dev;
  uint8_t* u_dev;
  uint8_t* v_dev;
} lh264_pic_t;

/* One frame to reconstruct (== one run of the reference's per-slice loop for every
 * slice NAL of an access unit + ExpandReferencingPicture). */
typedef struct lh264_frame_job {
  const lh264_mb_t*    mbs_dev;     /* mb_w*mb_h records, raster order                    */
  const int16_t*       coeffs_dev;  /* mb_w*mb_h*384, pScaledTCoeff layout*/
  const lh264_slice_t* slices_dev;  /* n_slices entries                                   */
  lh264_pic_t          dst;         /* picture being reconstructed                        */
  lh264_pic_t          ref[LH264_MAX_REFS]; /* reference pictures (deblocked + padded)    */
  int32_t  mb_w, mb_h;
  int32_t  stride_y, stride_c;
  int32_t  n_slices;
  int32_t  flags;                   /* LH264_JOB_* */
} lh264_frame_job_t;

#define LH264_JOB_NO_EXPAND   0x1   /* skip border replication (non-reference picture)    */
#define LH264_JOB_NO_DEBLOCK  0x2   /* debug: stop after reconstruction (pre-deblock planes) */

/* ---- library / device management ----------------------------------------- */
int         lh264_abi_version(void);
/* identifies the build: a hash of the sources the library was compiled from (set by the build recipe, __graft_entry__.build);
 * measurements kept beside the sources (profiles/traffic.json) name the build they were taken on */
const char* lh264_build_id(void);
const char* lh264_last_error(void);
/* number of visible HIP devices (<=0: none; every compute entry point then fails loudly) */
int         lh264_device_count(void);
int         lh264_set_device(int device);

void* lh264_dev_malloc(size_t bytes);
int   lh264_dev_free(void* p_dev);
int   lh264_memcpy_h2d(void* dst_dev, const void* src, size_t bytes, void* hip_stream);
int   lh264_memcpy_d2h(void* dst, const void* src_dev, size_t bytes, void* hip_stream);
int   lh264_dev_memset(void* dst_dev, int value, size_t bytes, void* hip_stream);
int   lh264_stream_sync(void* hip_stream);

/* bytes of one padded picture (all three planes, incl. padding) and the offsets
 * of pixel (0,0) of each plane inside such an allocation. */
size_t lh264_pic_bytes(int mb_w, int mb_h, int* stride_y, int* stride_c,
                       size_t* off_y, size_t* off_u, size_t* off_v);

/* ---- the hot path ---------------------------------------------------------
 * Reconstruct a batch of independent frames: intra/inter prediction + inverse
 * transforms (WelsTargetMbConstruction), in-loop deblocking
 * (WelsDeblockingFilterSlice) and border expansion (ExpandReferencingPicture).
 * jobs_dev: n_jobs descriptors resident in HBM. Frames in one call must not
 * reference each other (a P frame and its reference go in successive calls on
 * the same stream). Asynchronous on hip_stream (NULL = default stream).
 * Returns 0 or a negative LH264_E_* code. */
int lh264_recon_frames(const lh264_frame_job_t* jobs_dev, int n_jobs, int max_mb_w, int max_mb_h,
                       void* hip_stream);

/* Sequential chains: chain c reconstructs jobs_dev[chain_first[c] .. chain_first[c+1]-1]
 * in order inside one workgroup (frames of one stream: P frames may reference
 * earlier jobs of the same chain). chain_first_dev has n_chains+1 entries. */
int lh264_recon_chains(const lh264_frame_job_t* jobs_dev, const int32_t* chain_first_dev, int n_chains,
                       int max_mb_w, int max_mb_h, void* hip_stream);

/* timing helper for bench.py: time `iters` back-to-back invocations of the
 * dominant kernel with hipEvents on the launch stream; returns mean ms per
 * launch (<0 on error). */
double lh264_time_recon_chains(const lh264_frame_job_t* jobs_dev, const int32_t* chain_first_dev, int n_chains,
                               int max_mb_w, int max_mb_h, int iters, void* hip_stream);

/* ---- per-coefficient context-model index -------------------------------------------------
 * For every coefficient symbol of a macroblock the recompressor codes (luma/chroma DC, per-block nonzero count,
 * coefficients in zig-zag order) compute WHICH adaptive prior codes it: the flat index into the reference's
 * tables lumaDCIntPriors / chromaDCIntPriors / nonzerosPriors[8x8] / acPriors[8x8] (macroblock_model.h:45-53),
 * i.e. what getLumaDCIntPrior, getChromaDCIntPrior, getNonzerosPrior4x4/8x8 and getACPrior4x4/8x8
 * (macroblock_model.cpp:466-594) return, driven like encode4x4 (decode_slice.cpp:2059-2094, 2393-2434).
 * The adaptive update and the arithmetic coder (rows a9/a10) consume these symbols on the host side. */
typedef struct lh264_ctx_sym {
  uint32_t prior;     /* flat index into the table selected by `kind`                                  */
  int16_t  value;     /* the integer that is coded with that prior                                     */
  uint8_t  kind;      /* LH264_SYM_*                                                                   */
  uint8_t  pad;
} lh264_ctx_sym_t;
enum { LH264_SYM_LUMA_DC = 0, LH264_SYM_CHROMA_DC = 1, LH264_SYM_NZ4 = 2, LH264_SYM_AC4 = 3, LH264_SYM_NZ8 = 4, LH264_SYM_AC8 = 5,
       /* the non-coefficient syntax symbols (row a10), produced by the host front end; for these `prior` is
        * LH264_PRIOR(table, index) and `pad` the tag the decisions go to (billing.h:6-55) */
       LH264_SYM_TREE = 6,    /* value coded MSB first through a binary tree of priors (Branch<n>, compression_stream.h:117-166) */
       LH264_SYM_POW2 = 7,    /* emitBitsZeroToPow2Inclusive<n> (:455-463): flag "differs from the preferred value" + tree      */
       LH264_SYM_BIT = 8,     /* one decision with its own prior                                                             */
       LH264_SYM_RAW = 9,     /* `prior` raw bits of `value`, MSB first, through the shared adaptive TEST_PROB (:441-448)      */
       LH264_SYM_MVD = 10,    /* emitUEGkInt with MotionVectorDifferencePrior = UEGkIntPrior<9,4,3,4,3> (:575-591)            */
       LH264_SYM_SPLICE = 15  /* marker in a host list: the macroblock's coefficient symbols (rows a8) go here             */ };
#define LH264_CTX_MAX_SYMS 432   /* 16 + 8 + 24 + 384 symbols per macroblock at most */
/* prior tables of the reference's MacroblockModel (macroblock_model.h:36-136) */
enum { LH264_TB_MBTYPE = 0, LH264_TB_MVD, LH264_TB_MODE8, LH264_TB_LDC, LH264_TB_CDC, LH264_TB_NZ4, LH264_TB_NZ8, LH264_TB_AC4, LH264_TB_AC8,
       LH264_TB_SKIPRUN, LH264_TB_QPL, LH264_TB_SUBMB, LH264_TB_NUMREF, LH264_TB_CBPC, LH264_TB_CBPL, LH264_TB_STOP, LH264_TB_T8,
       LH264_TB_PREDMODE, LH264_TB_COUNT };
#define LH264_PRIOR(table, index) (((uint32_t)(table) << 27) | (uint32_t)(index))
#define LH264_MAX_SYN_SYMS 96    /* non-coefficient symbols of one macroblock at most (incl. the splice marker and a slice's pad bits) */

/* One frame of context-index work.  `levels` = the raw (not dequantised) coefficient levels in the
 * pScaledTCoeffQuant layout (what the reference copies into DecodedMacroblock::odata, decode_slice.cpp:69-79).
 * nnz images: 24 bytes per macroblock = per-4x4 nonzero counts (DecodedMacroblock::countSubblockNonzeros) of the
 * reference's FreqImage entry of that position: a skipped macroblock inherits the PAST entry
 * (decode_slice.cpp:3104-3108).  The host decides which earlier frame is PAST (frame_num flips, resolution
 * changes: decoded_macroblock.h:119-123, decode_slice.cpp:3032-3046) by pointing nnz_past_dev at its image. */
typedef struct lh264_ctx_job {
  const lh264_mb_t*    mbs_dev;       /* mb_w*mb_h records                                       */
  const int16_t*       levels_dev;    /* mb_w*mb_h*384                                           */
  const lh264_slice_t* slices_dev;
  const uint8_t*       nnz_past_dev;  /* mb_w*mb_h*24 or NULL (no PAST)                          */
  uint8_t*             nnz_cur_dev;   /* mb_w*mb_h*24, written by pass 1, read by pass 2 and later frames */
  lh264_ctx_sym_t*     syms_dev;      /* the symbols, emission order.  FIXED layout (sym_off_dev == NULL): mb_w*mb_h*LH264_CTX_MAX_SYMS slots,
                                         macroblock k at k*LH264_CTX_MAX_SYMS, first n_syms[k] valid.  COMPACT layout: see below       */
  uint16_t*            n_syms_dev;    /* mb_w*mb_h                                               */
  int32_t  mb_w, mb_h;
  /* COMPACT layout (ABI 3; sym_off_dev != NULL): the jobs of a call share ONE pool of symbols - syms_dev is the pool's first symbol in
   * every job, syms_cap its size in symbols -, macroblock k of this picture lies at syms_dev[*sym_base_dev + sym_off_dev[k]].  Both
   * are WRITTEN by the call (a count pass over the levels: 8 bytes per coded symbol, a macroblock's run padded to a multiple of 8
   * symbols = a 64-byte line, instead of 3,456 bytes per macroblock).  The pool is
   * sized from lh264_ctx_count_chains; a pool that is too small is not written to (every n_syms reads 0, *total_dev tells the need). */
  uint32_t*            sym_off_dev;   /* mb_w*mb_h offsets (symbols) behind the picture's first symbol              */
  uint64_t*            sym_base_dev;  /* this picture's first symbol in the pool (one word per job)                 */
  uint64_t             syms_cap;      /* symbols there is room for at syms_dev                                      */
} lh264_ctx_job_t;

/* chains as in lh264_recon_chains (frames of one stream in order: the nnz image of a frame may be the PAST of a
 * later one).  Pass 1 (nnz images) runs one workgroup per chain, pass 2 (symbols) one wave per macroblock. */
int lh264_ctx_index_chains (const lh264_ctx_job_t* jobs_dev, const int32_t* chain_first_dev, int n_chains,
                            int n_jobs, int max_mbs_per_frame, void* hip_stream);
/* COMPACT layout, first half: pass 1 and the count alone - fills n_syms_dev, sym_off_dev and *sym_base_dev of every job and
 * *total_dev = the symbols of all jobs (what the pool must hold).  lh264_ctx_index_chains repeats this on its own: the call exists so
 * that the caller can size the pool. */
int lh264_ctx_count_chains (const lh264_ctx_job_t* jobs_dev, const int32_t* chain_first_dev, int n_chains,
                            int n_jobs, int max_mbs_per_frame, unsigned long long* total_dev, void* hip_stream);

/* ---- rows a9 + a10 + f4: the adaptive binary arithmetic coder, on the device -----------------------------------
 * Consumes, per macroblock in coding order, the host list of syntax symbols (with the SPLICE marker where the
 * coefficient symbols of lh264_ctx_index_chains go) and produces the byte string of every tagged stream exactly as
 * the reference's compressor writes it to <out>.pip.<tag> (ArithmeticCodedOutput / vpx_writer,
 * compression_stream.h:353-487, bitwriter.h:35-105; DynProb :87-115; emitInt / emitUEGkInt :523-591).
 * The stream's symbols are binarised in parallel (one wave per macroblock); the adaptive probabilities are resolved by one
 * workgroup per stream, 64 decisions per wave step; one lane per (stream, tag) runs the bool coder.  The adaptive priors live in
 * the LDS of the stream's workgroup; what does not fit is spilled to a per-stream open-addressing table in HBM (hash_cells_dev:
 * hash_cap x 64 bytes = 8 x hash_cap entries of one DynProb each, zero-filled by the caller; a stream of N macroblocks touches
 * roughly 2 N DynProbs, QCIF ... 1080p content measured). */
#define LH264_N_TAG_SLOTS 40
typedef struct lh264_code_job {
  const lh264_ctx_sym_t* syn_syms_dev;   /* host symbols of the picture, macroblock after macroblock        */
  const uint32_t*        syn_off_dev;    /* n_mbs + 1 offsets into syn_syms_dev                             */
  const lh264_ctx_sym_t* ctx_syms_dev;   /* lh264_ctx_job_t.syms_dev: n_mbs * LH264_CTX_MAX_SYMS, or the pool    */
  const uint16_t*        ctx_n_syms_dev; /* n_mbs                                                           */
  int32_t n_mbs, reserved;
  const uint32_t*        ctx_sym_off_dev;  /* lh264_ctx_job_t.sym_off_dev (NULL: the fixed layout)          */
  const uint64_t*        ctx_sym_base_dev; /* lh264_ctx_job_t.sym_base_dev                                  */
} lh264_code_job_t;
typedef struct lh264_code_stream {
  uint32_t* hash_keys_dev;     /* not used (ABI 1 kept the keys of the table here)                          */
  uint32_t* hash_cells_dev;    /* hash_cap * 16 words, zero-filled: the spill table of the adaptive priors  */
  uint8_t*  out_dev;           /* LH264_N_TAG_SLOTS * out_cap bytes: slot t at t * out_cap                  */
  uint32_t* out_len_dev;       /* LH264_N_TAG_SLOTS lengths (0: tag never used); [LH264_N_TAG_SLOTS] = status (0 ok) */
  uint32_t  hash_cap;          /* power of two, at most 1 << 20                                             */
  uint32_t  out_cap;
} lh264_code_stream_t;
/* tag id (billing.h) <-> slot: slot = tag for tags < 34, slot 34 = tag 69 (pad bits).
 * chain_first_dev: n_chains + 1 entries (stream c codes jobs_dev[chain_first[c] .. chain_first[c+1]-1] in order); n_jobs = all
 * pictures, total_mbs = the sum of their n_mbs, max_mbs_per_frame = the largest n_mbs.  The call sizes its work memory (kept
 * between calls, per device) from a count pass, so it synchronises hip_stream once in the middle; the tagged streams are complete
 * when hip_stream has drained.  out_len_dev[LH264_N_TAG_SLOTS] != 0 reports (bits): 1 prior table full or hash_cap invalid (not a power of
 * two, zero, above 1 << 20, or - with the per-partition tables of large streams - below 8 entries x the partition count), 4 output
 * overflow (the lengths then say how much room is needed), 8 counter overflow (2^32 decisions or 2^27 list entries in one stream, or
 * total_mbs smaller than the pictures' macroblock counts), 16 internal (a hand-off between the waves of a stream timed out / the range
 * walk lost its state: the result is wrong; never seen outside broken development builds).  Any non-zero status: the stream's tagged
 * bytes must not be used.
 * Coder calls share one set of work memory per device: calls on different HIP streams are ordered on the device by an event (a later call
 * waits for the earlier one's kernels), calls on one stream by the stream. */
int lh264_code_chains (const lh264_code_job_t* jobs_dev, const int32_t* chain_first_dev, const lh264_code_stream_t* streams_dev,
                       int n_chains, int n_jobs, long long total_mbs, int max_mbs_per_frame, void* hip_stream);
/* The same in two calls, for a caller that wants to put other work between the halves (the second half - the adaptive probabilities
 * and the bool coders - is two serial chains whose waves mostly wait: it runs well beside a kernel that is bound by arithmetic,
 * e.g. lh264_recon_chains on another stream; the first half - binarisation - does not).  lh264_code_binarise_chains counts and
 * writes the decision words (it is the half that synchronises hip_stream once); lh264_code_finish_chains must follow on the same
 * device with the same streams_dev and n_chains, before any other coder call there. */
int lh264_code_binarise_chains (const lh264_code_job_t* jobs_dev, const int32_t* chain_first_dev, const lh264_code_stream_t* streams_dev,
                                int n_chains, int n_jobs, long long total_mbs, int max_mbs_per_frame, void* hip_stream);
int lh264_code_finish_chains (const lh264_code_stream_t* streams_dev, int n_chains, void* hip_stream);
/* sizes of the last lh264_code_chains call on the current device: 64-bit decision words written and read between its stages (one
 * per binary decision, each stream's count rounded up to 64) and 16-bit tag-list entries (one per decision, each tag's list padded
 * to 8): what the coder's memory traffic is computed from (bench.py). */
int lh264_code_last_totals (unsigned long long* decision_words, unsigned long long* list_entries);

/* ---- host front end: Annex-B bitstream -> macroblock records ------------------------------
 * Replaces, for the records the hot path needs, the reference's WelsDecodeBs / ParseNonVclNal / slice-header parse /
 * CAVLC macroblock parse (decoder.cpp:658-860, au_parser.cpp, decode_slice.cpp:3173-3984, parse_mb_syn_cavlc.cpp).
 * Pure host code.  Pictures come out in decode order (I and P slices only, as in the reference). */
typedef struct lh264_parser lh264_parser_t;
typedef struct lh264_frame_info {
  int32_t id, mb_w, mb_h, n_slices, n_refs, frame_num;
  int32_t crop_x, crop_y, crop_w, crop_h;      /* cropped output window in luma samples (SBufferInfo iWidth/iHeight) */
  int32_t is_ref, idr;
  int32_t ref_ids[LH264_MAX_REFS];             /* picture ids behind this picture's job ref slots */
} lh264_frame_info_t;
lh264_parser_t* lh264_parser_create (void);
void  lh264_parser_destroy (lh264_parser_t* p);
/* feed Annex-B bytes ending on a NAL boundary; flush != 0 completes the picture in progress. <0 on a parse error
 * (lh264_parser_error gives the text; pictures parsed so far stay available) */
int   lh264_parser_feed (lh264_parser_t* p, const uint8_t* data, size_t len, int flush);
/* a whole Annex-B file, cut into chunks and fed the way the reference's console application does (h264dec.cpp:246-272,
 * one ISVCDecoder::DecodeFrameNoDelay per start-code-delimited chunk), flush included.  Besides the pictures this builds
 * the recompressor's default stream (stream id 0x7fffffff, the ".pip" file itself: the input minus its slice data,
 * decoder.cpp:658-860, au_parser.cpp:143,588, decode_slice.cpp:2974-2980), returned by lh264_parser_main_stream */
int   lh264_parser_feed_file (lh264_parser_t* p, const uint8_t* data, size_t len);
const uint8_t* lh264_parser_main_stream (const lh264_parser_t* p, size_t* len);
/* the samples of the stream's I_PCM macroblocks (384 bytes each, decoding order): stream LH264_TAG_PCM of the container, see
 * lh264_pip_restore */
const uint8_t* lh264_parser_pcm_samples (const lh264_parser_t* p, size_t* len);
int   lh264_parser_frame_count (const lh264_parser_t* p);
int   lh264_parser_frame_info (const lh264_parser_t* p, int idx, lh264_frame_info_t* out);
const lh264_mb_t*    lh264_parser_frame_mbs (const lh264_parser_t* p, int idx);
const int16_t*       lh264_parser_frame_coeffs (const lh264_parser_t* p, int idx);
const int16_t*       lh264_parser_frame_levels (const lh264_parser_t* p, int idx);
const lh264_slice_t* lh264_parser_frame_slices (const lh264_parser_t* p, int idx);
const uint8_t*       lh264_parser_frame_covered (const lh264_parser_t* p, int idx);
/* row a10: the macroblock syntax the recompressor codes beyond lh264_mb_t (the reference's DecodedMacroblock fields,
 * decoded_macroblock.h:12-34): one packed 116-byte lh264_mbsyn_t per macroblock, and per slice 4 x int32
 * {alignment bit count after the stop bit, their value, PPS transform_8x8_mode_flag,
 * flags: bit 0 entropy_coding_mode_flag, bit 1 constrained_intra_pred_flag} */
typedef struct lh264_mbsyn {
  uint8_t  have, slice_type, t8, cbp_c, cbp_l, chroma_mode, luma16_mode, luma_qp;
  uint8_t  mb_type[4], num_ref_idx_l0[4], skip_run[4];   /* little-endian u32 / u32 / i32 (the struct is byte-packed) */
  int8_t   ref_idx[4];
  uint8_t  sub_type[4];
  int8_t   pred_mode[16];
  uint8_t  mvd[64];                                      /* int16 [16][2], little-endian */
  uint8_t  delta_qp[4], last_mb_qp[4];                   /* i32 */
} lh264_mbsyn_t;
const lh264_mbsyn_t* lh264_parser_frame_syntax (const lh264_parser_t* p, int idx);
const int32_t*       lh264_parser_frame_slice_syntax (const lh264_parser_t* p, int idx);
/* the row-a10 symbols of the picture (LH264_SYM_TREE .. LH264_SYM_SPLICE) and mb_w*mb_h + 1 offsets into them */
const lh264_ctx_sym_t* lh264_parser_frame_syn_symbols (const lh264_parser_t* p, int idx, int* count);
const uint32_t*      lh264_parser_frame_syn_offsets (const lh264_parser_t* p, int idx);
const char*          lh264_parser_error (const lh264_parser_t* p);

/* ---- restore direction, host side ------------------------------------------------------------
 * The inverse of compress: the default stream (".pip") plus the tagged arithmetic-coded streams (".pip.<tag>") -> the
 * original Annex-B bytes (what `h264dec in.pip out.264` does in the reference: decode_slice.cpp:2476-2936, decoder.cpp:658-860).
 * tags[t] / tag_len[t] are indexed by tag id (billing.h:6-55), n_tags >= 70 to include the pad-bit tag 69; NULL = no such
 * stream.  *out_len receives the restored size; LH264_E_ARG when out_cap is too small (then *out_len = the size needed).
 * I_PCM macroblocks: the reference's representation does not carry their samples (its own restore aborts on them); ours adds one
 * stream, tags[LH264_TAG_PCM] = the 384 samples of every I_PCM macroblock in decoding order, stored as they are.  With it streams
 * with I_PCM macroblocks restore (CAVLC: 7.3.5; CABAC: the engine is flushed before the samples and restarted behind them,
 * 9.3.1.2); without it the call gives LH264_E_UNSUPPORTED (lh264_restore_error: the text). */
#define LH264_TAG_PCM 70
int lh264_pip_restore (const uint8_t* main_stream, size_t main_len, const uint8_t* const* tags, const size_t* tag_len, int n_tags,
                       uint8_t* out, size_t out_cap, size_t* out_len);
const char* lh264_restore_error (void);       /* message of the calling thread's last failed lh264_pip_restore */

/* ---- single-file container: the default stream and the tagged streams in one file, or - flag VERBATIM - the
 * input itself for streams the round trip cannot carry (damaged streams, syntax the front end does not parse) or does not shrink, so that
 * every input restores.  Layout: "LHPIP1\0\0", u32 flags, u32 n, n x {u32 stream id (0x7fffffff = default stream, else the
 * tag id), u32 length}, the payloads in that order; little endian.  The reference has no counterpart (it writes one file per
 * stream, h264dec.cpp:79-104, and aborts on what it cannot restore). */
#define LH264_PIP_VERBATIM 1u
size_t lh264_pip_pack_bound (size_t main_len, const size_t* tag_len, int n_tags);
int lh264_pip_pack (const uint8_t* main_stream, size_t main_len, const uint8_t* const* tags, const size_t* tag_len, int n_tags,
                    uint32_t flags, uint8_t* out, size_t out_cap, size_t* out_len);
/* restore from a container: lh264_pip_restore on its streams, or a copy of the payload when it is VERBATIM */
int lh264_pip_restore_file (const uint8_t* file, size_t len, uint8_t* out, size_t out_cap, size_t* out_len);

/* ---- the whole compress direction behind one call (host orchestration in C++; what the Python sessions of this repository do)
 * n independent Annex-B files -> per stream the default stream and the tagged streams, exactly the files the reference's console
 * application writes (h264dec.cpp:79-121): host front end on `threads` host threads (0 = all), one lh264_ctx_index_chains and
 * one lh264_code_chains launch per sub-batch on the current device, results copied back.  Every out[i] is a handle to free;
 * lh264_compressed_status tells whether stream i compressed (LH264_OK) or why not (LH264_E_UNSUPPORTED: syntax the front end
 * does not parse; LH264_E_HIP ...), lh264_compressed_error gives the text. */
typedef struct lh264_compressed lh264_compressed_t;
int lh264_compress_batch (const uint8_t* const* data, const size_t* len, int n, int threads, lh264_compressed_t** out);
/* the same over several devices of one node: contiguous shares of about equal input size, one host thread per device driving
 * lh264_compress_batch there (the streams are independent: nothing is exchanged between devices) */
int lh264_compress_batch_devices (const uint8_t* const* data, const size_t* len, int n, int threads, const int* devices, int n_devices,
                                  lh264_compressed_t** out);
int lh264_compressed_status (const lh264_compressed_t* c);
const char* lh264_compressed_error (const lh264_compressed_t* c);
const uint8_t* lh264_compressed_main (const lh264_compressed_t* c, size_t* len);
const uint8_t* lh264_compressed_tag (const lh264_compressed_t* c, int tag, size_t* len);     /* NULL: the stream does not exist */
int lh264_compressed_pictures (const lh264_compressed_t* c);
void lh264_compressed_free (lh264_compressed_t* c);
void lh264_compress_release (void);       /* frees the device and page-locked buffers lh264_compress_batch keeps between calls */

/* ---- batches of independent streams on the host cores (streams are independent, one thread each) --
 * lh264_parse_batch: n Annex-B files -> n parsers (lh264_parser_feed_file each), `threads` worker threads (0 = one per
 * hardware thread).  parsers_out[i] is always a valid handle to destroy; its error text tells whether the stream parsed.
 * lh264_pip_restore_batch: n restores as lh264_pip_restore, item by item; item.status receives the return code. */
int lh264_parse_batch (const uint8_t* const* data, const size_t* len, int n, int threads, lh264_parser_t** parsers_out);
/* the same work with every picture released as soon as it is complete (what a pipeline does once the records are on their way
 * to the device): pictures_out[i] = pictures parsed of stream i.  The steady-state throughput probe of the front end. */
int lh264_parse_batch_discard (const uint8_t* const* data, const size_t* len, int n, int threads, int64_t* pictures_out);
typedef struct lh264_restore_item {
  const uint8_t* main_stream; size_t main_len;
  const uint8_t* const* tags; const size_t* tag_len; int32_t n_tags;
  int32_t status;                 /* out: LH264_OK / LH264_E_*                       */
  uint8_t* out; size_t out_cap;   /* caller's buffer                                 */
  size_t out_len;                 /* out: restored size (or the size needed)         */
} lh264_restore_item_t;
int lh264_pip_restore_batch (lh264_restore_item_t* items, int n, int threads);

#define LH264_OK            0
#define LH264_E_NODEVICE   -1
#define LH264_E_ARG        -2
#define LH264_E_HIP        -3
#define LH264_E_UNSUPPORTED -4

#ifdef __cplusplus
}
#endif
#endif /* LH264_H_ */
