/*
 * lambda_ext.h -- C ABI of the MI355X seed-extension engine (liblambda_ext.so).
 *
 * This is the drop-in boundary for lambda3's per-hit gapped extension.  The reference has no
 * FFI; its seam is the C++ template call
 *
 *     _performAlignment(depSetH, depSetV, blastMatches, lH, bool_constant<withTrace>, bsDirection)
 *         /root/reference/src/search_algo.hpp:1070-1076, called at :1246 (score) and :1296 (trace)
 *
 * whose inputs are N (query slice, subject slice) pairs plus the scoring scheme of the strand
 * direction, and whose outputs are alignmentScore per pair (:1129) or the gapped rows (:1127).
 * The entry points below replace exactly that call (lx_score_batch / lx_align_batch) and, one level
 * up, the whole of iterateMatchesFullSimd (:1177-1332) (lx_iterate_matches).  INTEGRATION.md shows
 * the reference-side binding.
 *
 * Conventions
 *   - residues are 1 byte each, already in the rank encoding the reference feeds to SeqAn
 *     (src/seqan2_to_biocpp.hpp:382-395): SeqAn AminoAcid order for proteins, BioC++ rank for
 *     match/mismatch-scored nucleotides, SeqAn Dna5 order in bisulfite mode.
 *   - query = horizontal / outer sequence, subject = vertical / inner (src/search_algo.hpp:1058-1059)
 *   - gap_open is SeqAn's scoreGapOpen, i.e. lambda's gapOpen + gapExtend (src/search_algo.hpp:226-230)
 *   - every function returns LX_OK (0) or a negative LX_E* code; lx_last_error() gives the text.
 *     The C++ wrapper (lambda_amd/csrc/host/lambda_ext.hpp) rethrows std::runtime_error, which is what
 *     the reference does on failure (src/search.cpp:98-125).
 *   - caller owns every buffer; nothing allocated here crosses the boundary.
 *   - a handle is bound to one device and one HIP stream; use one handle per host thread, as the
 *     reference uses one LocalDataHolder per OpenMP thread (src/search.cpp:379-381).
 *   - there is NO CPU fallback: if no gfx950 device / kernel image is usable every call fails loudly.
 */
#ifndef LAMBDA_EXT_H
#define LAMBDA_EXT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LX_ABI_VERSION 1
#define LX_ALPH 32 /* matrix stride; alphabet_size <= 31, rank 31 is reserved as padding */

enum
{
    LX_OK        = 0,
    LX_EINVAL    = -1, /* bad argument                         */
    LX_ENODEV    = -2, /* no usable gfx950 device              */
    LX_ENOMEM    = -3, /* device or host allocation failed     */
    LX_EHIP      = -4, /* HIP runtime error                    */
    LX_EOVERFLOW = -5, /* workspace / output capacity exceeded */
    LX_ESTATE    = -6  /* call sequence error                  */
};

/* Scoring scheme of one strand direction: TScoreSchemeAlign, src/search_datastructures.hpp:360-365;
 * filled like prepareScoring(), src/search_algo.hpp:166-234. */
typedef struct lx_scoring
{
    int32_t alphabet_size;            /* number of valid ranks (27 aa, 5 nucleotide)              */
    int32_t gap_open;                 /* cost of the first gap character  (negative)               */
    int32_t gap_extend;               /* cost of each further gap character (negative)             */
    int32_t reserved;
    int8_t  matrix[LX_ALPH * LX_ALPH]; /* matrix[query_rank * 32 + subject_rank]                   */
} lx_scoring;

/* One (query slice, subject slice) pair = one element of depSetH/depSetV (src/search_algo.hpp:1056-1060). */
typedef struct lx_extension
{
    uint64_t q_off; /* byte offset of the query slice in the query residue buffer    */
    uint64_t s_off; /* byte offset of the subject slice in the subject residue buffer */
    uint32_t q_len;
    uint32_t s_len;
} lx_extension;

/* Result of the traceback pass: what _adaptTraceSegmentsTo + beginPosition/endPosition give
 * (src/search_algo.hpp:1127, :1032-1035) and what computeAlignmentStats derives (:1308). */
typedef struct lx_hsp
{
    int32_t score;
    int32_t q_begin, q_end; /* 0-based half-open, relative to the query slice   */
    int32_t s_begin, s_end; /* 0-based half-open, relative to the subject slice */
    int32_t n_ops;          /* alignment columns written to the ops buffer      */
    int32_t num_matches, num_mismatches, num_positives;
    int32_t num_gap_opens, num_gap_extensions;
    int32_t reserved;
} lx_hsp;

/* src/search_datastructures.hpp:46-61 */
typedef struct lx_match
{
    uint64_t qryId, subjId, qryStart, qryEnd, subjStart, subjEnd;
} lx_match;

typedef struct lx_handle lx_handle;

/* ---- lifetime ------------------------------------------------------------------------------- */
int          lx_abi_version(void);
int          lx_device_count(void);
int          lx_create(int device_id, lx_handle ** out);
void         lx_destroy(lx_handle * h);
char const * lx_last_error(lx_handle const * h); /* h may be NULL: error of the failed lx_create */

/* Tuning knobs.  They never change results, only which kernel geometry is launched:
 *   LX_OPT_MAX_QLEN        longest query slice the *_dev calls will see (0 = unknown -> generic geometry)
 *   LX_OPT_QUERY_RUN       promise for the *_dev calls: extensions come in runs of this many consecutive entries
 *                          that share one query slice (must be a multiple of 8; 0 = no promise).  Lets a wavefront
 *                          build one LDS profile instead of one per extension.  A violated promise is detected on
 *                          the device and reported as LX_ESTATE by lx_synchronize().
 *   LX_OPT_WORKSPACE_BYTES carry workspace for queries wider than one panel in the *_dev calls (default 64 MiB) */
enum
{
    LX_OPT_MAX_QLEN        = 1,
    LX_OPT_QUERY_RUN       = 2,
    LX_OPT_WORKSPACE_BYTES = 3
};
int lx_set_option(lx_handle * h, int option, uint64_t value);

/* slot 0 = forward scheme, slot 1 = bisulfite reverse scheme (scoringSchemeAlignBSRev,
 * src/search_algo.hpp:1097-1098).  Must be called before any batch call using that slot. */
int lx_set_scoring(lx_handle * h, int slot, lx_scoring const * sc);

/* Fills *sc with a built-in scheme the way prepareScoring() does: scoring_method 45/62/80 = BLOSUM
 * (protein), 0 = match/mismatch (nucleotide), -1 = bisulfite forward, -2 = bisulfite reverse.
 * gap_open_lambda / gap_extend are lambda's options (e.g. -11/-1), NOT SeqAn's. */
int lx_builtin_scoring(int scoring_method, int match, int mismatch, int gap_open_lambda, int gap_extend,
                       lx_scoring * sc);

/* ---- pass 1: score only  (replaces _performAlignment<false>, src/search_algo.hpp:1246) ------- */
/* Host buffers in, host buffers out; copies through pinned staging, runs on the handle's stream, returns
 * after the results have landed. out_score[i] is what the reference stores at :1129. */
int lx_score_batch(lx_handle * h, int slot, uint8_t const * q_res, uint64_t q_bytes, uint8_t const * s_res,
                   uint64_t s_bytes, lx_extension const * ext, uint64_t n, int32_t * out_score);

/* Device-resident variant: every pointer is a device pointer on the handle's device; asynchronous on
 * `stream` (a hipStream_t; NULL = the handle's own stream).  This is what bench.py times. */
int lx_score_batch_dev(lx_handle * h, int slot, void const * d_q_res, void const * d_s_res, void const * d_ext,
                       uint64_t n, void * d_out_score, void * stream);

/* ---- pass 2: traceback  (replaces _performAlignment<true>, src/search_algo.hpp:1296) --------- */
/* ops: one byte per alignment column ('M','D','I'; 'D' = gap in the query row), extension i writes
 * out_hsp[i].n_ops bytes starting at out_ops + ops_off[i]; the caller sizes that slot to q_len+s_len. */
int lx_align_batch(lx_handle * h, int slot, uint8_t const * q_res, uint64_t q_bytes, uint8_t const * s_res,
                   uint64_t s_bytes, lx_extension const * ext, uint64_t n, lx_hsp * out_hsp, uint8_t * out_ops,
                   uint64_t const * ops_off);

int lx_align_batch_dev(lx_handle * h, int slot, void const * d_q_res, void const * d_s_res, void const * d_ext,
                       uint64_t n, void * d_out_hsp, void * d_out_ops, void const * d_ops_off, void * stream);

/* ---- pre-extension filter (seedLooksPromising, src/search_algo.hpp:426-481) ------------------ */
/* One diagonal per item; out_keep[i] = 1 if the ungapped max-segment score reaches the threshold. */
typedef struct lx_seed
{
    uint64_t q_off, s_off; /* start of the whole (frame) query / subject sequence */
    uint32_t q_len, s_len; /* their full lengths                                 */
    uint32_t qry_start, qry_end, subj_start, reserved;
} lx_seed;
int lx_prefilter_batch(lx_handle * h, int slot, uint8_t const * q_res, uint64_t q_bytes, uint8_t const * s_res,
                       uint64_t s_bytes, lx_seed const * seeds, uint64_t n, uint32_t seed_length,
                       int32_t pre_scoring, double pre_scoring_thresh, uint8_t * out_keep);

/* ---- misc ------------------------------------------------------------------------------------ */
/* Blocks until everything queued on the handle's stream has finished. */
int lx_synchronize(lx_handle * h);
/* Duration in ms of the most recent score / align kernel launch sequence on this handle, measured with HIP
 * events on the launch stream (valid after lx_synchronize or a host-buffer call). */
int lx_last_kernel_ms(lx_handle * h, float * ms);

#ifdef __cplusplus
}
#endif
#endif /* LAMBDA_EXT_H */
